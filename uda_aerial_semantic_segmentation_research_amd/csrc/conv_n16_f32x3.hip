// Sixteen produced channels on a sixteen-wide matrix tile (round 5): the full-resolution decoder tail of smp.Unet -- decoder block 4
// conv2 (16 -> 16 at 512^2), its data gradient, and the segmentation head's data gradient (24 -> 16) -- reference
// src/test_system.py:90-95 (decoder_channels (256, 128, 64, 32, 16), head Conv2d(16, classes, 3)), called src/models/train.py:341,
// differentiated :343.  fp32 tensors, the exact three-term bf16 split of conv_halo_f32x3.hip, fp32 accumulation.
//
// On conv3x3_f32x3_kernel<4, 1, 2> these layers fill 16 of the 32 rows of every v_mfma_f32_32x32x16_bf16: 116 GFLOP of matrix-pipe
// work for a 9.66 GFLOP layer, and that -- not HBM -- is what bounded them (profiles/r03_f32x3.txt).  Here the products run on
// v_mfma_f32_16x16x32_bf16: A = weights (16 channels x K = 32), B = pixels (K = 32 x 16 pixels), K = 32 = TWO TAPS of a 16-channel
// chunk -- lane l holds A[channel l & 15][k = 8 (l >> 4) .. + 7] and B[k][pixel l & 15] with k = 16 t + c, t the tap of the pair,
// c the channel of the chunk; the nine taps go as the pairs (0,1) (2,3) (4,5) (6,7) (8,-): five MFMA sets per chunk and
// 16 x 16 tile (the last one half empty) instead of nine 32-row ones = 0.28 of the pipe cycles.  The accumulator lane holds 4
// consecutive channels of one pixel (row = 4 (l >> 4) + reg, column = l & 15): 16-byte stores, 1 KB contiguous per tile.
// The two taps of a pair read two different halo pixels: lanes 0-31 at tap 2j, lanes 32-63 at tap 2j + 1 -- per-lane LDS
// addresses, 32-byte pixel rows with NO swizzle (ds_read_b128 serves lanes {0-3, 12-15, 20-27} together: pixels 0-3 / 12-15 at
// octet 0 with pixels 4-11 at octet 1 -- sixteen distinct 16-byte slots of a 256-byte line for any column shift).
// One-role kernel (every wave loads, splits, stages and multiplies), PERSISTENT over 8 x 32 pixel tiles: the weights of the whole layer
// stay in LDS (15 KB per chunk), the next tile's fp32 halo is in flight during the current tile's MFMAs and epilogue, the statistics
// leave the block once.  Options: the unwritten BatchNorm activation as input
// (F3Args::in_scale of conv_halo_f32x3.hip), BatchNorm statistics of the output (forward), BatchNorm-backward sums of the producing
// layer (data gradient).  Weight fragments: udaseg_pack_up_batched_f32x3 modes 4 / 5, plane[p][chunk][pair][lane][8].
#include <stdlib.h>

#include "common.h"
#include "halo_common.h"

namespace udaseg {

struct N16Args {
  const float* x;       // gathered tensor [n][h][w][ci]
  const void* wf;       // [3][nk16 * 5 * 512] bf16
  float* y;             // produced tensor [n][h][w][16]
  int n, h, w, ci;
  double* stats;        // [R][2][16] f64: statistics of y, or the bnb_* sums
  double* sscr;
  const float* bnb_y;   // data gradient: conv output [n][h][w][16] of the producing conv+BN+activation layer
  const float* bnb_mean;
  const float* bnb_rstd;
  const float* bnb_gamma;
  const float* bnb_beta;
  int bnb_act;
  float bnb_slope;
  const float* in_scale;   // x is an unwritten BatchNorm activation: act(fma(x, in_scale[c], in_shift[c])) while staging
  const float* in_shift;
  int in_act;
  float in_slope;
  int ntx, nty, nk16;
  int q1, q3;           // (chunk, pair) groups [q1, q3) run on negated weights and a negated accumulator
  unsigned x_bytes, w_plane_bytes, y_bytes;
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct N16Cfg {
  static constexpr int NT = 256, WM = 4, RPW = 2;
  static constexpr int TH = WM * RPW, TW = 32;
  static constexpr int HR = TH + 2, HWD = TW + 2;
  static constexpr int PLANE = HR * HWD * 32;
  static constexpr int LDS_HALO = 3 * PLANE;
  static constexpr int NPIECE = HR * HWD * 2;
  static constexpr int NI = (NPIECE + NT - 1) / NT;
  static constexpr int WCHUNK = 5 * 3 * 1024;             // weight fragments of a 16-channel chunk: [pair][plane] x 1 KB
  static constexpr int NB = 2 * RPW;                      // 16-pixel blocks per wave
};

// sum over the 16 lanes of a DPP row (the 16 pixels of an accumulator column block); every lane of the row ends with the total
template <int N>
__device__ __forceinline__ void row16_sum_n(float (&x)[N]) {
#pragma unroll
  for (int v = 0; v < N; ++v) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x[v]));
  asm volatile("s_nop 1");
#pragma unroll
  for (int v = 0; v < N; ++v) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(x[v]));
  asm volatile("s_nop 1");
#pragma unroll
  for (int v = 0; v < N; ++v) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf" : "+v"(x[v]));
  asm volatile("s_nop 1");
#pragma unroll
  for (int v = 0; v < N; ++v) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(x[v]));
  asm volatile("s_nop 1");
}

template <bool XF>
__global__ __launch_bounds__(256, 2) void conv3x3_n16_f32x3_kernel(const N16Args a) {
  using C = N16Cfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wlds = smem + C::LDS_HALO;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;          // pixel of the 16-pixel block / K slice (g >> 1: tap of the pair, g & 1: octet)

  const int H = a.h, W = a.w;
  const int ntiles = a.n * a.nty * a.ntx;
  // persistent: block b takes tiles b, b + gridDim.x, ...; the fp32 halo of the NEXT tile is requested before the MFMAs and the
  // epilogue of the current one (with one or two chunks of K a tile has nothing to pipeline inside itself, and these layers move
  // 0.27-0.4 GB per launch: what bounds them is bytes in flight, not the matrix pipe any more)
  const int oct = tid & 1;
  unsigned voff[C::NI], soffl[C::NI];
  unsigned inmask = 0;                             // XF: bit i = piece i of the staged chunk lies inside the image
#pragma unroll
  for (int i = 0; i < C::NI; ++i) soffl[i] = (unsigned)(((tid + i * C::NT) >> 1) * 32 + oct * 16);
  int img = 0, y0 = 0, x0 = 0;                     // the tile the loads were last issued for
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
      const int pix = piece >> 1;
      const int hy = pix / C::HWD, hx = pix - hy * C::HWD;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      const bool ok = piece < C::NPIECE && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      voff[i] = ok ? (unsigned)((((img * H + iy) * W + ix) * a.ci + oct * 8) * 4) : 0x80000000u;
    }
  };
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wf), 0, (int)(3u * a.w_plane_bytes), 0x00020000);

  u32x4 stage[C::NI][2];
  auto load_chunk = [&](int c) {
    const int soff = c * 64;
    const unsigned kill = (c * 16 + oct * 8 < a.ci) ? 0u : 0x80000000u;
    inmask = 0;
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
      stage[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(voff[i] | kill), soff, 0);
      stage[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(voff[i] | kill), soff + 16, 0);
      inmask |= (voff[i] != 0x80000000u ? 1u : 0u) << i;
    }
  };
  const bool xf_relu = a.in_act == UDASEG_ACT_LEAKY && a.in_slope == 0.f;
  auto store_chunk = [&](int c) {
    f32x4 sc0 = {0.f, 0.f, 0.f, 0.f}, sc1 = sc0, sh0 = sc0, sh1 = sc0;
    if constexpr (XF) {
      const int ch = c * 16 + oct * 8;
      if (ch < a.ci) {
        sc0 = *reinterpret_cast<const f32x4*>(a.in_scale + ch);
        sh0 = *reinterpret_cast<const f32x4*>(a.in_shift + ch);
        if (ch + 4 < a.ci) {
          sc1 = *reinterpret_cast<const f32x4*>(a.in_scale + ch + 4);
          sh1 = *reinterpret_cast<const f32x4*>(a.in_shift + ch + 4);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
      if (i < C::NI - 1 || tid + i * C::NT < C::NPIECE) {
        u32x4 lo = stage[i][0], hi = stage[i][1];
        if constexpr (XF) {
          const bool inside = (inmask >> i) & 1u;            // zero padding is padding of the ACTIVATION: stays zero
          f32x4 l = __builtin_bit_cast(f32x4, lo), h = __builtin_bit_cast(f32x4, hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t0 = __builtin_fmaf(l[e], sc0[e], sh0[e]), t1 = __builtin_fmaf(h[e], sc1[e], sh1[e]);
            if (xf_relu) {
              t0 = t0 > 0.f ? t0 : 0.f;
              t1 = t1 > 0.f ? t1 : 0.f;
            } else {
              t0 = act_apply(t0, a.in_act, a.in_slope);
              t1 = act_apply(t1, a.in_act, a.in_slope);
            }
            l[e] = inside ? t0 : 0.f;
            h[e] = inside ? t1 : 0.f;
          }
          lo = __builtin_bit_cast(u32x4, l);
          hi = __builtin_bit_cast(u32x4, h);
        }
        u32x4 p0, p1, p2;
        split3(lo, hi, p0, p1, p2);
        *reinterpret_cast<u32x4*>(smem + soffl[i]) = p0;
        *reinterpret_cast<u32x4*>(smem + C::PLANE + soffl[i]) = p1;
        *reinterpret_cast<u32x4*>(smem + 2 * C::PLANE + soffl[i]) = p2;
      }
    }
  };

  // the layer's weight fragments, resident for the block: LDS [chunk][pair][plane][lane][16 bytes]
  {
    const int npc = a.nk16 * 5 * 64;                // 16-byte pieces per plane
    for (int i = tid; i < 3 * npc; i += C::NT) {
      const int pl = i / npc, r = i - pl * npc;
      const int f = r >> 6, ln = r & 63;            // fragment (chunk * 5 + pair), lane
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(pl * a.w_plane_bytes + r * 16), 0, 0);
      *reinterpret_cast<u32x4*>(wlds + (f * 3 + pl) * 1024 + ln * 16) = v;
    }
  }

  // per-lane halo offsets of the five tap pairs: lanes with g >> 1 == 0 read tap 2j, the others tap 2j + 1 (the ninth tap's partner
  // carries zero weights: any valid address)
  int toff[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    int tap = 2 * j + (g >> 1);
    if (tap > 8) tap = 8;
    toff[j] = ((tap / 3) * C::HWD + (tap % 3)) * 32;
  }
  const int pbase = ((wave * C::RPW) * C::HWD + p) * 32 + (g & 1) * 16;

  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, 0x00020000);
  const bool want_stats = a.stats != nullptr && a.bnb_y == nullptr;
  const bool want_bnb = a.bnb_y != nullptr;
  __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(want_bnb ? a.bnb_y : a.x), 0,
                                                                  (int)(want_bnb ? a.y_bytes : 0u), 0x00020000);
  float s[8];                        // [0..3]: sums, [4..7]: second sums, of this lane's 4 channels, over all tiles of the block
#pragma unroll
  for (int v = 0; v < 8; ++v) s[v] = 0.f;
  f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc, mu = sc, rsd = sc;
  if (want_bnb) {
    mu = *reinterpret_cast<const f32x4*>(a.bnb_mean + 4 * g);
    rsd = *reinterpret_cast<const f32x4*>(a.bnb_rstd + 4 * g);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(a.bnb_gamma + 4 * g), bt = *reinterpret_cast<const f32x4*>(a.bnb_beta + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sc[e] = gm[e] * rsd[e];                     // as bn_apply forms them
      sh[e] = bt[e] - mu[e] * sc[e];
    }
  }

  int tile = blockIdx.x;
  if (tile < ntiles) {
    tile_setup(tile);
    load_chunk(0);
  }
  for (; tile < ntiles; tile += gridDim.x) {
    const int cimg = img, cy0 = y0, cx0 = x0;      // this tile (tile_setup moves on to the next one inside the loop)
    f32x4v acc[C::NB];
#pragma unroll
    for (int b = 0; b < C::NB; ++b) acc[b] = f32x4v{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < a.nk16; ++c) {
      store_chunk(c);
      __syncthreads();               // the chunk's halo (and, the first time, the weights) are visible
      if (c + 1 < a.nk16) {
        load_chunk(c + 1);
      } else if (tile + (int)gridDim.x < ntiles) {
        tile_setup(tile + gridDim.x);
        load_chunk(0);
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int G = 5 * c + j;
        if (G == a.q1 || G == a.q3) {
#pragma unroll
          for (int b = 0; b < C::NB; ++b) acc[b] = -acc[b];
        }
        u32x4 A[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) A[pl] = *reinterpret_cast<const u32x4*>(wlds + ((c * 5 + j) * 3 + pl) * 1024 + lane * 16);
#pragma unroll
        for (int b = 0; b < C::NB; ++b) {
          const int r = b >> 1, bx = b & 1;
          u32x4 B[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            B[pl] = *reinterpret_cast<const u32x4*>(smem + pl * C::PLANE + pbase + toff[j] + (r * C::HWD + bx * 16) * 32);
          // smallest terms first (weight piece i x pixel piece ij - i, i + j <= 2)
#pragma unroll
          for (int ij = 2; ij >= 0; --ij)
#pragma unroll
            for (int i = 0; i <= ij; ++i)
              acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[i]), __builtin_bit_cast(bf16x8, B[ij - i]), acc[b], 0, 0, 0);
        }
      }
      __syncthreads();               // every wave is done with the halo before the next chunk / tile overwrites it
    }

    // ---- epilogue: acc[b][e] of lane (p, g): channel 4 g + e of pixel (row wave * RPW + (b >> 1), column 16 (b & 1) + p)
#pragma unroll
    for (int b = 0; b < C::NB; ++b) {
      const int oy = cy0 + wave * C::RPW + (b >> 1), ox = cx0 + 16 * (b & 1) + p;
      const bool cv = oy < H && ox < W;
      const unsigned off = cv ? (((unsigned)((cimg * H + oy) * W + ox)) * 16u + 4u * (unsigned)g) * 4u : 0x80000000u;
      // (whole-vector cast: the element-wise form, bit_cast(unsigned, acc[b][e]) in a loop, compiled to FOUR COPIES OF ELEMENT 0 --
      // hipcc 7.2; the same trap as the accumulate load in conv_halo_f32x3_epilogue.inc)
      const u32x4 d = __builtin_bit_cast(u32x4, acc[b]);
      __builtin_amdgcn_raw_buffer_store_b128(d, rs_y, (int)off, 0, 0);
      if (want_stats) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float q = cv ? acc[b][e] : 0.f;
          s[e] += q;
          s[4 + e] = __builtin_fmaf(q, q, s[4 + e]);
        }
      }
      if (want_bnb) {
        const f32x4 yv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, (int)off, 0, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float yy = yv[e];
          const float gg = cv ? acc[b][e] * act_grad(__builtin_fmaf(yy, sc[e], sh[e]), a.bnb_act, a.bnb_slope) : 0.f;
          s[e] += gg;
          s[4 + e] = __builtin_fmaf(gg, (yy - mu[e]) * rsd[e], s[4 + e]);
        }
      }
    }
  }
  if (want_stats || want_bnb) {      // once per block: the sums of all its tiles
    asm volatile("s_nop 1");
    row16_sum_n(s);
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][2][16]; the tile loop ended with a barrier
    if (p == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[wave * 32 + 4 * g + e] = s[e];
        red[wave * 32 + 16 + 4 * g + e] = s[4 + e];
      }
    }
    __syncthreads();
    if (tid < 32) {
      const float tot = red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid];      // tid < 16: sums, else second sums
      atomicAdd(a.stats + (size_t)(blockIdx.x % HALO_STATS_REPLICAS) * 32 + tid, (double)tot);
    }
  }
}

static bool n16_applicable(const udaseg_conv_desc* d, int dgrad) {
  if (!d || !f32_halo_enabled()) return false;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->ho != d->hi || d->wo != d->wi) return false;
  const int gathered = dgrad ? d->co : d->ci, produced = dgrad ? d->ci : d->co;
  if (produced != 16 || gathered % 8 != 0 || gathered < 8 || gathered > 32 || d->n <= 0 || d->hi <= 0 || d->wi < 16) return false;
  const long long px = (long long)d->n * d->hi * d->wi;
  return px * 32 * 4 < (1LL << 31);
}

static int launch_n16(N16Args a, hipStream_t s, double flops) {
  using C = N16Cfg;
  a.ntx = cdiv(a.w, C::TW);
  a.nty = cdiv(a.h, C::TH);
  a.nk16 = (a.ci + 15) / 16;
  a.q1 = a.q3 = -1;
  if (f3_signs_on()) {
    const int ng = 5 * a.nk16;
    a.q1 = (ng + 2) / 4;
    a.q3 = ng - a.q1;
  }
  const int lds = C::LDS_HALO + a.nk16 * C::WCHUNK;
  const long long ntiles = (long long)a.n * a.nty * a.ntx;
  if (ntiles <= 0) return UDASEG_OK;
  if (ntiles >= (1LL << 31)) return UDASEG_E_UNSUPPORTED;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
              ? pr.multiProcessorCount : 256;
  }
  const long long resident = (long long)cus * (a.nk16 == 1 ? 3 : 2);      // blocks that fit the chip at once (48 / 63 KB of LDS each)
  const long long blocks = ntiles < resident ? ntiles : resident;
  a.sscr = nullptr;
  const bool xf = a.in_scale != nullptr;
  auto kern = xf ? conv3x3_n16_f32x3_kernel<true> : conv3x3_n16_f32x3_kernel<false>;
  static std::atomic<bool> attr_done[2] = {{false}, {false}};
  if (!attr_done[xf]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       C::LDS_HALO + 2 * C::WCHUNK);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3x3_n16_f32x3)");
    attr_done[xf] = true;
  }
  static std::atomic<int> kid[2] = {{-1}, {-1}};
  if (kid[xf] < 0) kid[xf] = kprof_id(xf ? "conv3x3_n16_f32x3_kernel<true>" : "conv3x3_n16_f32x3_kernel<false>");
  hipEvent_t ev = kprof_begin(s);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::NT), lds, s, a);
  kprof_end(kid[xf], ev, s, flops);
  UDASEG_LAUNCH_CHECK("conv3x3_n16_f32x3 launch");
  return UDASEG_OK;
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_conv_n16_f32x3_ok(const udaseg_conv_desc* d, int dgrad) { return n16_applicable(d, dgrad) ? 1 : 0; }

// y[n][h][w][16] = conv3x3(x) on the sixteen-wide tile; in_scale != NULL: x is the RAW conv output of a conv + BatchNorm + activation
// layer whose activation was never written (udaseg_conv2d_fwd_f32x3_bnin's contract); stats: BatchNorm statistics of y
extern "C" int udaseg_conv2d_fwd_n16_f32x3(const udaseg_conv_desc* d, const float* x, const float* in_scale, const float* in_shift,
                                           int in_act, float in_slope, const void* wfrag, float* y, double* stats, void* stream) {
  UDASEG_CHECK_ARG(d && x && wfrag && y, "conv2d_fwd_n16_f32x3: NULL pointer");
  UDASEG_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "conv2d_fwd_n16_f32x3: in_scale and in_shift come together");
  UDASEG_CHECK_ARG(in_act == UDASEG_ACT_NONE || in_act == UDASEG_ACT_LEAKY, "conv2d_fwd_n16_f32x3: unknown activation %d", in_act);
  if (!n16_applicable(d, 0)) {
    set_error("conv2d_fwd_n16_f32x3: geometry not supported (ask udaseg_conv_n16_f32x3_ok first)");
    return UDASEG_E_UNSUPPORTED;
  }
  N16Args a = {};
  a.x = x; a.wf = wfrag; a.y = y;
  a.n = d->n; a.h = d->hi; a.w = d->wi; a.ci = d->ci;
  a.stats = stats;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act; a.in_slope = in_slope;
  const long long px = (long long)d->n * d->hi * d->wi;
  a.x_bytes = (unsigned)(px * d->ci * 4);
  a.w_plane_bytes = (unsigned)(((d->ci + 15) / 16) * 5 * 1024);
  a.y_bytes = (unsigned)(px * 16 * 4);
  hipStream_t st = as_stream(stream);
  prof_begin(0, st);
  const int rc = launch_n16(a, st, udaseg_conv_flops(d));
  prof_end(0, st, udaseg_conv_flops(d), 0, d);
  return rc;
}

// dx[n][h][w][16] = conv_transpose(dy, w) for a convolution with 16 input channels; prev_y != NULL: also the BatchNorm-backward
// sums of the layer that produced the convolution's input (udaseg_conv2d_dgrad_f32x3's contract)
extern "C" int udaseg_conv2d_dgrad_n16_f32x3(const udaseg_conv_desc* d, const float* dy, const void* wfrag_t, float* dx,
                                             const float* prev_y, const float* save_mean, const float* save_rstd, const float* gamma,
                                             const float* beta, int bn_act, float bn_slope, double* bsums, void* stream) {
  UDASEG_CHECK_ARG(d && dy && wfrag_t && dx, "conv2d_dgrad_n16_f32x3: NULL pointer");
  const bool bn = prev_y != nullptr;
  UDASEG_CHECK_ARG(!bn || (save_mean && save_rstd && gamma && beta && bsums), "conv2d_dgrad_n16_f32x3: the BatchNorm-backward sums "
                   "need mean, rstd, gamma, beta, bsums");
  UDASEG_CHECK_ARG(bn || bsums == nullptr, "conv2d_dgrad_n16_f32x3: bsums without prev_y");
  if (!n16_applicable(d, 1)) {
    set_error("conv2d_dgrad_n16_f32x3: geometry not supported (ask udaseg_conv_n16_f32x3_ok first)");
    return UDASEG_E_UNSUPPORTED;
  }
  N16Args a = {};
  a.x = dy; a.wf = wfrag_t; a.y = dx;
  a.n = d->n; a.h = d->hi; a.w = d->wi; a.ci = d->co;
  if (bn) {
    a.bnb_y = prev_y; a.bnb_mean = save_mean; a.bnb_rstd = save_rstd; a.bnb_gamma = gamma; a.bnb_beta = beta;
    a.bnb_act = bn_act; a.bnb_slope = bn_slope; a.stats = bsums;
  }
  const long long px = (long long)d->n * d->hi * d->wi;
  a.x_bytes = (unsigned)(px * d->co * 4);
  a.w_plane_bytes = (unsigned)(((d->co + 15) / 16) * 5 * 1024);
  a.y_bytes = (unsigned)(px * 16 * 4);
  hipStream_t st = as_stream(stream);
  prof_begin(0, st);
  const int rc = launch_n16(a, st, udaseg_conv_flops(d));
  prof_end(0, st, udaseg_conv_flops(d), 1, d);
  return rc;
}
