// Bilinear x2 up-sampling (+ concat with the skip tensor), forward and backward, NHWC fp32 / bf16 (gfx950).  HBM-bound.
//
// north_star names the decoder's up-sampling "bilinear"; the reference's traced model uses NEAREST (five
// aten::upsample_nearest2d in the add_graph fixture), so nearest stays the parity default and is fused into the consumer
// convolution's gather (conv_igemm.hip).  This file is the alternate mode Unet(..., upsample="bilinear"):
//   torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
// followed by torch.cat([up, skip], 1), as one pass.  With scale 2 and align_corners=False the source coordinate of output
// row oy is oy/2 - 0.25 clamped at 0, so every output is a fixed 0.75 / 0.25 blend of two neighbouring rows and columns
// (edges clamp):   out[2k] = 0.25 in[max(k-1, 0)] + 0.75 in[k],   out[2k+1] = 0.75 in[k] + 0.25 in[min(k+1, h-1)].
// Backward is in gather form (each input pixel sums its <= 4x4 dependants with the transposed weights): no atomics.
#include "common.h"

namespace udaseg {

template <bool BF>
struct Vec16 {   // one 16-byte vector: 4 fp32 or 8 bf16 channels, widened to fp32 for arithmetic
  static constexpr int N = BF ? 8 : 4;
  float v[N];
  __device__ __forceinline__ static Vec16 load(const void* base, int64_t idx) {
    Vec16 r;
    const f32x4 raw = static_cast<const f32x4*>(base)[idx];
    if constexpr (BF) {
      const unsigned* u = reinterpret_cast<const unsigned*>(&raw);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        r.v[2 * i] = __builtin_bit_cast(float, u[i] << 16);
        r.v[2 * i + 1] = __builtin_bit_cast(float, u[i] & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) r.v[i] = raw[i];
    }
    return r;
  }
  __device__ __forceinline__ void store(void* base, int64_t idx) const {
    f32x4 raw;
    if constexpr (BF) {
      unsigned* u = reinterpret_cast<unsigned*>(&raw);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const __bf16 lo = (__bf16)v[2 * i], hi = (__bf16)v[2 * i + 1];     // round to nearest even
        u[i] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) raw[i] = v[i];
    }
    static_cast<f32x4*>(base)[idx] = raw;
  }
};

// source rows / columns and the weight of the SECOND one for output index o of a length-2*len axis
__device__ __forceinline__ void bilinear_src(int o, int len, int& i0, int& i1, float& l1) {
  const int k = o >> 1;
  if (o & 1) {
    i0 = k;
    i1 = min(k + 1, len - 1);
    l1 = 0.25f;
  } else {
    i0 = max(k - 1, 0);
    i1 = k;
    l1 = k == 0 ? 0.f : 0.75f;      // torch clamps the source coordinate at 0: the first output row is row 0 itself
    if (k == 0) i1 = min(1, len - 1);
  }
}

template <bool BF>
__global__ void bilinear_upcat_fwd_kernel(const void* __restrict__ a, const void* __restrict__ skip, void* __restrict__ out,
                                          int n, int h, int w, int caq, int cbq) {
  const int ctq = caq + cbq, H = 2 * h, W = 2 * w;
  const int64_t total = (int64_t)n * H * W * ctq;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += T) {
    const int q = (int)(i % ctq);
    const int64_t r = i / ctq;      // output pixel
    if (q >= caq) {
      Vec16<BF>::load(skip, r * cbq + (q - caq)).store(out, i);
      continue;
    }
    const int x = (int)(r % W);
    const int64_t r2 = r / W;
    const int y = (int)(r2 % H), ni = (int)(r2 / H);
    int y0, y1, x0, x1;
    float hl1, wl1;
    bilinear_src(y, h, y0, y1, hl1);
    bilinear_src(x, w, x0, x1, wl1);
    const float hl0 = 1.f - hl1, wl0 = 1.f - wl1;
    const int64_t b = (int64_t)ni * h;
    const Vec16<BF> p00 = Vec16<BF>::load(a, ((b + y0) * w + x0) * caq + q), p01 = Vec16<BF>::load(a, ((b + y0) * w + x1) * caq + q);
    const Vec16<BF> p10 = Vec16<BF>::load(a, ((b + y1) * w + x0) * caq + q), p11 = Vec16<BF>::load(a, ((b + y1) * w + x1) * caq + q);
    Vec16<BF> o;
#pragma unroll
    for (int e = 0; e < Vec16<BF>::N; ++e)
      o.v[e] = hl0 * (wl0 * p00.v[e] + wl1 * p01.v[e]) + hl1 * (wl0 * p10.v[e] + wl1 * p11.v[e]);
    o.store(out, i);
  }
}

// weights with which input index k receives the gradients of outputs 2k-1 .. 2k+2 (transpose of bilinear_src)
__device__ __forceinline__ void bilinear_adj(int k, int len, float (&wt)[4]) {
  wt[0] = 0.25f; wt[1] = 0.75f; wt[2] = 0.75f; wt[3] = 0.25f;
  if (k == 0) { wt[0] = 0.f; wt[1] = 1.f; }                // output 0 is input 0 itself
  if (k == len - 1) { wt[3] = 0.f; wt[2] = 1.f; }          // output 2*len-1 blends row len-1 with its clamped self
}

template <bool BF>
__global__ void bilinear_upcat_bwd_a_kernel(const void* __restrict__ dout, void* __restrict__ da, int n, int h, int w, int caq,
                                            int ctq, int accumulate) {
  const int W = 2 * w;
  const int64_t total = (int64_t)n * h * w * caq;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += T) {
    const int q = (int)(i % caq);
    int64_t r = i / caq;
    const int x = (int)(r % w);
    r /= w;
    const int y = (int)(r % h), ni = (int)(r / h);
    float wy[4], wx[4];
    bilinear_adj(y, h, wy);
    bilinear_adj(x, w, wx);
    Vec16<BF> g;
#pragma unroll
    for (int e = 0; e < Vec16<BF>::N; ++e) g.v[e] = 0.f;
#pragma unroll
    for (int jy = 0; jy < 4; ++jy) {
      if (wy[jy] == 0.f) continue;
      const int64_t row = ((int64_t)ni * 2 * h + (2 * y - 1 + jy)) * W;
#pragma unroll
      for (int jx = 0; jx < 4; ++jx) {
        if (wx[jx] == 0.f) continue;
        const Vec16<BF> d = Vec16<BF>::load(dout, (row + (2 * x - 1 + jx)) * ctq + q);
        const float wgt = wy[jy] * wx[jx];
#pragma unroll
        for (int e = 0; e < Vec16<BF>::N; ++e) g.v[e] += wgt * d.v[e];
      }
    }
    if (accumulate) {
      const Vec16<BF> old = Vec16<BF>::load(da, i);
#pragma unroll
      for (int e = 0; e < Vec16<BF>::N; ++e) g.v[e] += old.v[e];
    }
    g.store(da, i);
  }
}

template <bool BF>
__global__ void upcat_bwd_skip_any_kernel(const void* __restrict__ dout, void* __restrict__ dskip, int64_t pixels, int caq,
                                          int cbq, int accumulate) {
  const int ctq = caq + cbq;
  const int64_t total = pixels * cbq;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += T) {
    const int q = (int)(i % cbq);
    const int64_t p = i / cbq;
    Vec16<BF> g = Vec16<BF>::load(dout, p * ctq + caq + q);
    if (accumulate) {
      const Vec16<BF> old = Vec16<BF>::load(dskip, i);
#pragma unroll
      for (int e = 0; e < Vec16<BF>::N; ++e) g.v[e] += old.v[e];
    }
    g.store(dskip, i);
  }
}

static inline int grid_items(int64_t items) {
  int64_t g = (items + 255) / 256;
  if (g > 8192) g = 8192;
  return g < 1 ? 1 : (int)g;
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_upsample2x_bilinear_concat_fwd(const void* a, const void* skip, void* out, int n, int h, int w, int ca,
                                                     int cb, int bf16, void* stream) {
  const int g = bf16 ? 8 : 4;
  UDASEG_CHECK_ARG(a && out && n > 0 && h > 0 && w > 0 && ca > 0 && ca % g == 0 && cb >= 0 && cb % g == 0,
                   "upsample2x_bilinear_concat_fwd: bad arguments (channels must be multiples of %d)", g);
  UDASEG_CHECK_ARG(cb == 0 || skip, "upsample2x_bilinear_concat_fwd: skip is NULL but cb > 0");
  const int64_t total = (int64_t)n * 4 * h * w * ((ca + cb) / g);
  if (bf16)
    hipLaunchKernelGGL(bilinear_upcat_fwd_kernel<true>, dim3(grid_items(total)), dim3(256), 0, as_stream(stream), a, skip, out, n, h,
                       w, ca / g, cb / g);
  else
    hipLaunchKernelGGL(bilinear_upcat_fwd_kernel<false>, dim3(grid_items(total)), dim3(256), 0, as_stream(stream), a, skip, out, n,
                       h, w, ca / g, cb / g);
  UDASEG_LAUNCH_CHECK("upsample2x_bilinear_concat_fwd launch");
  return UDASEG_OK;
}

extern "C" int udaseg_upsample2x_bilinear_concat_bwd(const void* dout, void* da, void* dskip, int n, int h, int w, int ca, int cb,
                                                     int accumulate_da, int accumulate_dskip, int bf16, void* stream) {
  const int g = bf16 ? 8 : 4;
  UDASEG_CHECK_ARG(dout && n > 0 && h > 0 && w > 0 && ca > 0 && ca % g == 0 && cb >= 0 && cb % g == 0,
                   "upsample2x_bilinear_concat_bwd: bad arguments (channels must be multiples of %d)", g);
  hipStream_t st = as_stream(stream);
  if (da) {
    const int64_t total = (int64_t)n * h * w * (ca / g);
    if (bf16)
      hipLaunchKernelGGL(bilinear_upcat_bwd_a_kernel<true>, dim3(grid_items(total)), dim3(256), 0, st, dout, da, n, h, w, ca / g,
                         (ca + cb) / g, accumulate_da);
    else
      hipLaunchKernelGGL(bilinear_upcat_bwd_a_kernel<false>, dim3(grid_items(total)), dim3(256), 0, st, dout, da, n, h, w, ca / g,
                         (ca + cb) / g, accumulate_da);
    UDASEG_LAUNCH_CHECK("upsample2x_bilinear_concat_bwd(a) launch");
  }
  if (dskip && cb > 0) {
    const int64_t pixels = (int64_t)n * 4 * h * w;
    if (bf16)
      hipLaunchKernelGGL(upcat_bwd_skip_any_kernel<true>, dim3(grid_items(pixels * (cb / g))), dim3(256), 0, st, dout, dskip,
                         pixels, ca / g, cb / g, accumulate_dskip);
    else
      hipLaunchKernelGGL(upcat_bwd_skip_any_kernel<false>, dim3(grid_items(pixels * (cb / g))), dim3(256), 0, st, dout, dskip,
                         pixels, ca / g, cb / g, accumulate_dskip);
    UDASEG_LAUNCH_CHECK("upsample2x_bilinear_concat_bwd(skip) launch");
  }
  return UDASEG_OK;
}
