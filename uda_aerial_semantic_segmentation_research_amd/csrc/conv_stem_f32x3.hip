// The encoder's stem (round 5): 7x7 / stride 2 / pad 3 convolution of a 4-channel (3 + padding) fp32 image to 64 channels -- torchvision
// ResNet.conv1 inside smp.Unet (reference src/test_system.py:90-95; called src/models/train.py:341; trace fixture: the first
// aten::_convolution, 3 -> 64, k 7 / s 2 / p 3).  Forward only (the image needs no gradient; the weight gradient stays on the split-K
// kernel).  fp32 tensors, the exact three-term bf16 split, six v_mfma_f32_32x32x16_bf16 products per operand pair.
//
// Before: conv_igemm_kernel X3 with the generic (non-uniform) gather -- 4 channels per tap are no K tile, so every K step paid the
// tap-table address arithmetic and the operand split for its rows AND the weights: 157 us for 9.9 GFLOP (84 TFLOP/s).
// Here the im2col never exists: for one kernel ROW ky the 7 taps x 4 channels of an output pixel are 28 CONTIGUOUS floats of input
// row 2 oy + ky - 3 (NHWC with C = 4), starting 8 floats further for every output pixel.  A block stages a band of input rows once
// (fp32 -> three bf16 planes, origin shifted so that output pixel p's window starts at element 8 p: 16-byte aligned), and K runs as
// 7 kernel rows x 32 (28 + 4 zero-weight columns that read the neighbouring pixel): the pixel fragment of (ky, 16-wide half h) is ONE
// ds_read_b128 per lane at  row(ky) + 16 p + 32 h + 16 (lane >> 5)  -- consecutive pixels 16 bytes apart, conflict-free.  Weights are
// pre-split and pre-packed in fragment order (udaseg_pack_up_batched_f32x3 mode 8) and stay in LDS for the life of the block (84 KB);
// the kernel is persistent over 8 x 32 output-pixel tiles with the next tile's band in flight during the current tile's MFMAs
// (conv_n16_f32x3.hip's scheme); each of the 8 waves owns one output row x 64 channels.  Epilogue: 16-byte stores, BatchNorm statistics
// of the output summed over the block's tiles, one set of f64 atomics per block.
#include <stdlib.h>

#include "common.h"
#include "halo_common.h"

namespace udaseg {

struct StemArgs {
  const float* x;       // [n][h][w][4]
  const void* wf;       // [3][7 * 2 * 2 * 512] bf16: plane[p][ky][h][cb][lane][8]
  float* y;             // [n][h/2][w/2][64]
  int n, h, w;          // input extents (even)
  double* stats;        // [R][2][64] f64 or null
  int ntx, nty;
  int q1, q3;           // (ky, h) groups [q1, q3) run on negated weights and a negated accumulator
  unsigned x_bytes, w_plane_bytes, y_bytes;
};

struct StemCfg {
  static constexpr int NT = 512, NW = 8;
  static constexpr int TH = 8, TW = 32;                   // output pixels per tile: one row per wave
  static constexpr int IR = 2 * TH + 5;                   // input rows of the band
  static constexpr int NP = 35;                           // 2-pixel pieces per band row (69 pixels + 1)
  static constexpr int ROWB = NP * 16;                    // bytes of a band row in one bf16 plane (8 bf16 per piece)
  static constexpr int PLANE = IR * ROWB;
  static constexpr int LDS_BAND = 3 * PLANE;
  static constexpr int NPIECE = IR * NP;
  static constexpr int NI = (NPIECE + NT - 1) / NT;
  static constexpr int NFRAG = 7 * 2 * 2;                 // weight fragments per plane: [ky][h][cb]
  static constexpr int LDS_W = 3 * NFRAG * 1024;
  static constexpr int LDS = LDS_BAND + LDS_W;
};

__global__ __launch_bounds__(512, 1) void conv_stem_f32x3_kernel(const StemArgs a) {
  using C = StemCfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wlds = smem + C::LDS_BAND;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lp = lane & 31, lh = lane >> 5;
  const int H = a.h, W = a.w, HO = a.h >> 1, WO = a.w >> 1;
  const int ntiles = a.n * a.nty * a.ntx;

  unsigned voff[C::NI][2];
  int img = 0, y0 = 0, x0 = 0;
  auto tile_setup = [&](int tl) {
    const int tx = tl % a.ntx;
    const int t2 = tl / a.ntx;
    const int ty = t2 % a.nty;
    img = t2 / a.nty;
    y0 = ty * C::TH;
    x0 = tx * C::TW;
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
      const int piece = tid + i * C::NT;
      const int r = piece / C::NP, q = piece - r * C::NP;
      const int iy = 2 * y0 - 3 + r, ix = 2 * x0 - 3 + 2 * q;
      const bool rok = piece < C::NPIECE && (unsigned)iy < (unsigned)H;
#pragma unroll
      for (int e = 0; e < 2; ++e)
        voff[i][e] = (rok && (unsigned)(ix + e) < (unsigned)W) ? (unsigned)(((img * H + iy) * W + ix + e) * 16) : 0x80000000u;
    }
  };
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wf), 0, (int)(3u * a.w_plane_bytes), 0x00020000);
  u32x4 stage[C::NI][2];
  auto load_band = [&]() {
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
      stage[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)voff[i][0], 0, 0);
      stage[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)voff[i][1], 0, 0);
    }
  };
  auto store_band = [&]() {
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
      const int piece = tid + i * C::NT;
      if (i < C::NI - 1 || piece < C::NPIECE) {
        u32x4 p0, p1, p2;
        split3(stage[i][0], stage[i][1], p0, p1, p2);
        *reinterpret_cast<u32x4*>(smem + piece * 16) = p0;
        *reinterpret_cast<u32x4*>(smem + C::PLANE + piece * 16) = p1;
        *reinterpret_cast<u32x4*>(smem + 2 * C::PLANE + piece * 16) = p2;
      }
    }
  };

  // the weights, resident for the block: LDS [fragment (ky, h, cb)][plane][lane][16 bytes]
  for (int i = tid; i < 3 * C::NFRAG * 64; i += C::NT) {
    const int pl = i / (C::NFRAG * 64), r = i - pl * (C::NFRAG * 64);
    const int f = r >> 6, ln = r & 63;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(pl * a.w_plane_bytes + r * 16), 0, 0);
    *reinterpret_cast<u32x4*>(wlds + (f * 3 + pl) * 1024 + ln * 16) = v;
  }

  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, 0x00020000);
  const bool want_stats = a.stats != nullptr;
  float sA[32], sB[32];               // [cb * 16 + v]
#pragma unroll
  for (int v = 0; v < 32; ++v) sA[v] = sB[v] = 0.f;
  const int pbase = (2 * wave) * C::ROWB + 16 * lp + 16 * lh;      // this wave's output row reads band rows 2 * wave + ky

  int tile = blockIdx.x;
  if (tile < ntiles) {
    tile_setup(tile);
    load_band();
  }
  for (; tile < ntiles; tile += gridDim.x) {
    const int cimg = img, cy0 = y0, cx0 = x0;
    store_band();
    __syncthreads();                   // the band (and, the first time, the weights) are visible
    if (tile + (int)gridDim.x < ntiles) {
      tile_setup(tile + gridDim.x);
      load_band();
    }
    f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[cb][v] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int G = 2 * ky + h;
        if (G == a.q1 || G == a.q3) {
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) acc[cb] = -acc[cb];
        }
        u32x4 B[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) B[pl] = *reinterpret_cast<const u32x4*>(smem + pl * C::PLANE + pbase + ky * C::ROWB + 32 * h);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          u32x4 A[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) A[pl] = *reinterpret_cast<const u32x4*>(wlds + (((ky * 2 + h) * 2 + cb) * 3 + pl) * 1024 + lane * 16);
#pragma unroll
          for (int ij = 2; ij >= 0; --ij)
#pragma unroll
            for (int i = 0; i <= ij; ++i)
              acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[i]), __builtin_bit_cast(bf16x8, B[ij - i]), acc[cb], 0, 0, 0);
        }
      }
    __syncthreads();                   // every wave is done with the band before the next tile overwrites it

    // epilogue: acc[cb][v] of lane (lp, lh): channel cb * 32 + (v & 3) + 8 (v >> 2) + 4 lh of output pixel (cy0 + wave, cx0 + lp)
    const int oy = cy0 + wave, ox = cx0 + lp;
    const bool cv = oy < HO && ox < WO;
    const unsigned pixoff = (unsigned)((cimg * HO + oy) * WO + ox);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned off = cv ? (pixoff * 64u + (unsigned)(cb * 32 + 8 * g + 4 * lh)) * 4u : 0x80000000u;
        float val[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = acc[cb][4 * g + e];
        if (want_stats) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float q = cv ? val[e] : 0.f;
            sA[cb * 16 + 4 * g + e] += q;
            sB[cb * 16 + 4 * g + e] = __builtin_fmaf(q, q, sB[cb * 16 + 4 * g + e]);
          }
        }
        u32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = __builtin_bit_cast(unsigned, val[e]);
        __builtin_amdgcn_raw_buffer_store_b128(d, rs_y, (int)off, 0, 0);
      }
  }
  if (want_stats) {                    // once per block
    asm volatile("s_nop 1");
    halfwave_sum_n(sA);
    halfwave_sum_n(sB);
    asm volatile("s_nop 1");
    float* red = reinterpret_cast<float*>(smem);   // [2][8 waves][64]; the tile loop ended with a barrier
    if (lp == 31) {
#pragma unroll
      for (int v = 0; v < 32; ++v) {
        const int ch = (v >> 4) * 32 + (v & 3) + 8 * ((v & 15) >> 2) + 4 * lh;
        red[wave * 64 + ch] = sA[v];
        red[C::NW * 64 + wave * 64 + ch] = sB[v];
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, ch = tid & 63;
      float t = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < C::NW; ++w8) t += red[which * C::NW * 64 + w8 * 64 + ch];
      atomicAdd(a.stats + (size_t)(blockIdx.x % HALO_STATS_REPLICAS) * 128 + which * 64 + ch, (double)t);
    }
  }
}

static bool stem_applicable(const udaseg_conv_desc* d) {
  if (!d || !f32_halo_enabled()) return false;
  if (d->kh != 7 || d->kw != 7 || d->stride != 2 || d->pad != 3 || d->ci != 4 || d->co != 64) return false;
  if (d->n <= 0 || d->hi < 2 || d->wi < 2 || d->hi % 2 != 0 || d->wi % 2 != 0 || d->ho * 2 != d->hi || d->wo * 2 != d->wi) return false;
  const long long pin = (long long)d->n * d->hi * d->wi, pout = (long long)d->n * d->ho * d->wo;
  return pin * 16 < (1LL << 31) && pout * 256 < (1LL << 31);
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_conv_stem_f32x3_ok(const udaseg_conv_desc* d) { return stem_applicable(d) ? 1 : 0; }

// y[n][hi/2][wi/2][64] = conv7x7 / stride 2 / pad 3 (x[n][hi][wi][4]); stats: BatchNorm statistics of y.  wfrag: mode-8 packing
// (3 planes of 7 * 2 * 2 * 512 bf16).
extern "C" int udaseg_conv2d_fwd_stem_f32x3(const udaseg_conv_desc* d, const float* x, const void* wfrag, float* y, double* stats,
                                            void* stream) {
  UDASEG_CHECK_ARG(d && x && wfrag && y, "conv2d_fwd_stem_f32x3: NULL pointer");
  if (!stem_applicable(d)) {
    set_error("conv2d_fwd_stem_f32x3: geometry not supported (7x7 / stride 2 / pad 3, 4 -> 64 channels, even extents; ask "
              "udaseg_conv_stem_f32x3_ok first)");
    return UDASEG_E_UNSUPPORTED;
  }
  using C = StemCfg;
  StemArgs a = {};
  a.x = x; a.wf = wfrag; a.y = y;
  a.n = d->n; a.h = d->hi; a.w = d->wi;
  a.stats = stats;
  a.ntx = cdiv(d->wo, C::TW);
  a.nty = cdiv(d->ho, C::TH);
  a.q1 = a.q3 = -1;
  if (f3_signs_on()) {
    a.q1 = (14 + 2) / 4;
    a.q3 = 14 - a.q1;
  }
  a.x_bytes = (unsigned)((long long)d->n * d->hi * d->wi * 16);
  a.w_plane_bytes = (unsigned)(C::NFRAG * 1024);
  a.y_bytes = (unsigned)((long long)d->n * d->ho * d->wo * 256);
  const long long ntiles = (long long)d->n * a.nty * a.ntx;
  if (ntiles <= 0) return UDASEG_OK;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
              ? pr.multiProcessorCount : 256;
  }
  const long long blocks = ntiles < cus ? ntiles : cus;      // one 8-wave block per CU (119 KB of LDS)
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_f32x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       C::LDS);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv_stem_f32x3)");
    attr_done = true;
  }
  static std::atomic<int> kid{-1};
  if (kid < 0) kid = kprof_id("conv_stem_f32x3_kernel");
  hipStream_t st = as_stream(stream);
  prof_begin(0, st);
  hipEvent_t ev = kprof_begin(st);
  hipLaunchKernelGGL(conv_stem_f32x3_kernel, dim3((unsigned)blocks), dim3(C::NT), C::LDS, st, a);
  kprof_end(kid, ev, st, udaseg_conv_flops(d));
  prof_end(0, st, udaseg_conv_flops(d), 0, d);
  UDASEG_LAUNCH_CHECK("conv_stem_f32x3 launch");
  return UDASEG_OK;
}
