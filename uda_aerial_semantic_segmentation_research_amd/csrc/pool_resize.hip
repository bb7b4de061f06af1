// Layout, pooling and resize glue of the encoder-decoder, NHWC fp32 (gfx950).  All HBM-bound streams.
//
// Replaces, inside smp.Unet.forward / autograd (reference src/models/train.py:341,343):
//   max_pool2d(3,2,1) of the ResNet stem, F.interpolate(scale_factor=2, mode='nearest') + torch.cat([up, skip], 1)
//   of every decoder block (trace fixture: 5 upsample_nearest2d, 4 cat, order [upsampled, skip]),
// and the NCHW batch the DataLoader hands over (train.py:337).
#include "common.h"

namespace udaseg {

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int c, int64_t hw, int cpad,
                                    int64_t total_pix) {
  // one thread per pixel: c strided plane reads (coalesced across lanes), one contiguous cpad-float write
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total_pix; p += T) {
    const int64_t n = p / hw, s = p - n * hw;
    const float* src = x + n * c * hw + s;
    float* dst = y + p * cpad;
    for (int k = 0; k < cpad; k += 4) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (k + e) < c ? src[(int64_t)(k + e) * hw] : 0.f;
      *reinterpret_cast<f32x4*>(dst + k) = v;
    }
  }
}

// max_pool2d(kernel 3, stride 2, pad 1).  idx = first maximal tap in row-major window order (torch CPU kernel's rule:
// update on (val > max) || isnan(val)).
__global__ void maxpool_fwd_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, uint32_t* __restrict__ idx, int n,
                                   int h, int w, int c4, int ho, int wo) {
  const int64_t total = (int64_t)n * ho * wo * c4;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += T) {
    const int q = (int)(i % c4);
    int64_t r = i / c4;
    const int ox = (int)(r % wo);
    r /= wo;
    const int oy = (int)(r % ho);
    const int ni = (int)(r / ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    uint32_t bi[4] = {0, 0, 0, 0};
    bool first = true;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)w) continue;
        const f32x4 v = x[((int64_t)(ni * h + iy) * w + ix) * c4 + q];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (first || v[e] > best[e] || v[e] != v[e]) {
            best[e] = v[e];
            bi[e] = (uint32_t)(ky * 3 + kx);
          }
        }
        first = false;
      }
    }
    y[i] = best;
    idx[i] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
  }
}

// gather form: every input element collects from the <= 4 windows that contain it; no atomics.
// One image row per blockIdx.y step (row index math is wave-uniform), 32-bit column math: the flat 64-bit index form spent its
// time in three emulated 64-bit divisions per element (100 us on the 8 x 256^2 x 64 stem gradient against ~40 us of HBM traffic).
__global__ void maxpool_bwd_kernel(const f32x4* __restrict__ dy, const uint32_t* __restrict__ idx, f32x4* __restrict__ dx,
                                   int n, int h, int w, int c4, int ho, int wo, int accumulate) {
  const int rowlen = w * c4;
  const int col = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (col >= rowlen) return;
  const int ix = col / c4, q = col - ix * c4;
  for (int row = (int)blockIdx.y; row < n * h; row += (int)gridDim.y) {
    const int ni = row / h, iy = row - ni * h;
    f32x4 g = {0, 0, 0, 0};
    // windows oy with oy*2-1+ky == iy, ky in [0,3)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = iy + 1 - ky;
      if (ty < 0 || (ty & 1)) continue;
      const int oy = ty >> 1;
      if (oy >= ho) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = ix + 1 - kx;
        if (tx < 0 || (tx & 1)) continue;
        const int ox = tx >> 1;
        if (ox >= wo) continue;
        const int64_t o = ((int64_t)(ni * ho + oy) * wo + ox) * c4 + q;
        const uint32_t id = idx[o];
        const f32x4 d = dy[o];
        const uint32_t me = (uint32_t)(ky * 3 + kx);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (((id >> (8 * e)) & 0xffu) == me) g[e] += d[e];
      }
    }
    const int64_t i = (int64_t)row * rowlen + col;
    if (accumulate) g += dx[i];
    dx[i] = g;
  }
}

__global__ void upcat_fwd_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ skip, f32x4* __restrict__ out, int n,
                                 int h, int w, int ca4, int cb4) {
  const int ct4 = ca4 + cb4, H = 2 * h, W = 2 * w;
  const int64_t total = (int64_t)n * H * W * ct4;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += T) {
    const int q = (int)(i % ct4);
    int64_t r = i / ct4;  // output pixel index
    f32x4 v;
    if (q < ca4) {
      const int x = (int)(r % W);
      int64_t r2 = r / W;
      const int y = (int)(r2 % H);
      const int ni = (int)(r2 / H);
      v = a[((int64_t)(ni * h + (y >> 1)) * w + (x >> 1)) * ca4 + q];
    } else {
      v = skip[r * cb4 + (q - ca4)];
    }
    out[i] = v;
  }
}

// one row of da per blockIdx.y step, 32-bit column math (see maxpool_bwd_kernel)
__global__ void upcat_bwd_a_kernel(const f32x4* __restrict__ dout, f32x4* __restrict__ da, int n, int h, int w, int ca4,
                                   int ct4, int accumulate) {
  const int W = 2 * w, rowlen = w * ca4;
  const int col = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (col >= rowlen) return;
  const int x = col / ca4, q = col - x * ca4;
  for (int row = (int)blockIdx.y; row < n * h; row += (int)gridDim.y) {
    const int64_t p00 = (int64_t)(2 * row) * W + 2 * x;      // row = ni * h + y -> output row ni * 2h + 2y
    f32x4 g = dout[p00 * ct4 + q];
    g += dout[(p00 + 1) * ct4 + q];
    g += dout[(p00 + W) * ct4 + q];
    g += dout[(p00 + W + 1) * ct4 + q];
    const int64_t i = (int64_t)row * rowlen + col;
    if (accumulate) g += da[i];
    da[i] = g;
  }
}

__global__ void upcat_bwd_skip_kernel(const f32x4* __restrict__ dout, f32x4* __restrict__ dskip, int64_t pixels, int ca4,
                                      int cb4, int accumulate) {
  const int ct4 = ca4 + cb4;
  const int64_t total = pixels * cb4;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += T) {
    const int q = (int)(i % cb4);
    const int64_t p = i / cb4;
    f32x4 g = dout[p * ct4 + ca4 + q];
    if (accumulate) g += dskip[i];
    dskip[i] = g;
  }
}

__global__ void fill_kernel(float* __restrict__ p, int64_t n, float v) {
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += T) p[i] = v;
}
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n, float alpha) {
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += T) y[i] += alpha * x[i];
}

__global__ void scale_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float alpha) {
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += T) y[i] = alpha * x[i];
}

__global__ void add_i64_kernel(long long* __restrict__ p, int64_t n, long long v) {
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += T) p[i] += v;
}

static inline int grid_for(int64_t items, int per_thread = 2) {
  int64_t g = (items + 256LL * per_thread - 1) / (256LL * per_thread);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_nchw_to_nhwc(const float* x, float* y, int n, int c, int h, int w, int cpad, void* stream) {
  UDASEG_CHECK_ARG(x && y && n > 0 && c > 0 && h > 0 && w > 0, "nchw_to_nhwc: bad arguments");
  UDASEG_CHECK_ARG(cpad >= c && cpad % 4 == 0, "nchw_to_nhwc: cpad must be a multiple of 4 and >= c");
  const int64_t hw = (int64_t)h * w, total = hw * n;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total, 1)), dim3(256), 0, as_stream(stream), x, y, c, hw, cpad, total);
  UDASEG_LAUNCH_CHECK("nchw_to_nhwc launch");
  return UDASEG_OK;
}

extern "C" int udaseg_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int n, int h, int w, int c, void* stream) {
  UDASEG_CHECK_ARG(x && y && idx && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "maxpool_fwd: bad arguments");
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)n * ho * wo * (c / 4);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total, 1)), dim3(256), 0, as_stream(stream), (const f32x4*)x, (f32x4*)y,
                     (uint32_t*)idx, n, h, w, c / 4, ho, wo);
  UDASEG_LAUNCH_CHECK("maxpool_fwd launch");
  return UDASEG_OK;
}

extern "C" int udaseg_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int n, int h, int w, int c,
                                       int accumulate, void* stream) {
  UDASEG_CHECK_ARG(dy && idx && dx && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "maxpool_bwd: bad arguments");
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  UDASEG_CHECK_ARG((int64_t)w * (c / 4) < (1LL << 30) && (int64_t)n * h < (1LL << 30), "maxpool_bwd: extent");
  const unsigned gx = (unsigned)(((int64_t)w * (c / 4) + 255) / 256);
  const unsigned gy = (unsigned)((int64_t)n * h < 65535 ? (int64_t)n * h : 65535);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(gx, gy), dim3(256), 0, as_stream(stream), (const f32x4*)dy,
                     (const uint32_t*)idx, (f32x4*)dx, n, h, w, c / 4, ho, wo, accumulate);
  UDASEG_LAUNCH_CHECK("maxpool_bwd launch");
  return UDASEG_OK;
}

extern "C" int udaseg_upsample2x_concat_fwd(const float* a, const float* skip, float* out, int n, int h, int w, int ca,
                                            int cb, void* stream) {
  UDASEG_CHECK_ARG(a && out && n > 0 && h > 0 && w > 0 && ca > 0 && ca % 4 == 0 && cb >= 0 && cb % 4 == 0,
                   "upsample2x_concat_fwd: bad arguments");
  UDASEG_CHECK_ARG(cb == 0 || skip, "upsample2x_concat_fwd: skip is NULL but cb > 0");
  const int64_t total = (int64_t)n * 4 * h * w * ((ca + cb) / 4);
  hipLaunchKernelGGL(upcat_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const f32x4*)a,
                     (const f32x4*)skip, (f32x4*)out, n, h, w, ca / 4, cb / 4);
  UDASEG_LAUNCH_CHECK("upsample2x_concat_fwd launch");
  return UDASEG_OK;
}

extern "C" int udaseg_upsample2x_concat_bwd(const float* dout, float* da, float* dskip, int n, int h, int w, int ca,
                                            int cb, int accumulate_da, int accumulate_dskip, void* stream) {
  UDASEG_CHECK_ARG(dout && n > 0 && h > 0 && w > 0 && ca > 0 && ca % 4 == 0 && cb >= 0 && cb % 4 == 0,
                   "upsample2x_concat_bwd: bad arguments");
  hipStream_t st = as_stream(stream);
  if (da) {
    UDASEG_CHECK_ARG((int64_t)w * (ca / 4) < (1LL << 30) && (int64_t)n * h < (1LL << 30), "upsample2x_concat_bwd: extent");
    const unsigned gx = (unsigned)(((int64_t)w * (ca / 4) + 255) / 256);
    const unsigned gy = (unsigned)((int64_t)n * h < 65535 ? (int64_t)n * h : 65535);
    hipLaunchKernelGGL(upcat_bwd_a_kernel, dim3(gx, gy), dim3(256), 0, st, (const f32x4*)dout, (f32x4*)da, n, h, w,
                       ca / 4, (ca + cb) / 4, accumulate_da);
    UDASEG_LAUNCH_CHECK("upsample2x_concat_bwd(a) launch");
  }
  if (dskip && cb > 0) {
    const int64_t pixels = (int64_t)n * 4 * h * w;
    hipLaunchKernelGGL(upcat_bwd_skip_kernel, dim3(grid_for(pixels * (cb / 4))), dim3(256), 0, st, (const f32x4*)dout,
                       (f32x4*)dskip, pixels, ca / 4, cb / 4, accumulate_dskip);
    UDASEG_LAUNCH_CHECK("upsample2x_concat_bwd(skip) launch");
  }
  return UDASEG_OK;
}

extern "C" int udaseg_fill_f32(float* p, int64_t count, float value, void* stream) {
  UDASEG_CHECK_ARG(p && count >= 0, "fill_f32: bad arguments");
  if (count == 0) return UDASEG_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(count, 4)), dim3(256), 0, as_stream(stream), p, count, value);
  UDASEG_LAUNCH_CHECK("fill launch");
  return UDASEG_OK;
}

extern "C" int udaseg_axpy_f32(float* y, const float* x, int64_t count, float alpha, void* stream) {
  UDASEG_CHECK_ARG(y && x && count >= 0, "axpy_f32: bad arguments");
  if (count == 0) return UDASEG_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(count, 4)), dim3(256), 0, as_stream(stream), y, x, count, alpha);
  UDASEG_LAUNCH_CHECK("axpy launch");
  return UDASEG_OK;
}

/* p[i] += value on int64 (nn.BatchNorm2d's num_batches_tracked += 1 for every BatchNorm of a network in ONE launch: the buffers are
 * views of one arena; a torch add_ costs ~38 us of host time, this 4; reference src/models/train.py:341 in training mode) */
extern "C" int udaseg_add_i64(int64_t* p, int64_t count, int64_t value, void* stream) {
  UDASEG_CHECK_ARG(p && count >= 0, "add_i64: bad arguments");
  if (count == 0) return UDASEG_OK;
  hipLaunchKernelGGL(add_i64_kernel, dim3(grid_for(count, 4)), dim3(256), 0, as_stream(stream), reinterpret_cast<long long*>(p), count,
                     (long long)value);
  UDASEG_LAUNCH_CHECK("add_i64 launch");
  return UDASEG_OK;
}

/* y = alpha * x (gradient reversal: alpha = -lambda; reference src/models/uda.py:99-111) */
extern "C" int udaseg_scale_f32(const float* x, float* y, int64_t count, float alpha, void* stream) {
  UDASEG_CHECK_ARG(y && x && count >= 0, "scale_f32: bad arguments");
  if (count == 0) return UDASEG_OK;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(count, 4)), dim3(256), 0, as_stream(stream), x, y, count, alpha);
  UDASEG_LAUNCH_CHECK("scale launch");
  return UDASEG_OK;
}
