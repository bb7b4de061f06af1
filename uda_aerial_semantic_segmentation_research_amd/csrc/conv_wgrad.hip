// Convolution weight gradient, NHWC fp32, split-K implicit GEMM on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Replaces autograd's conv weight-gradient reached by loss.backward() (reference src/models/train.py:343,
// src/models/adversarial_trainer.py:97,113).
//
// GEMM view:  dW[co][j] = sum_m dy[m][co] * X[m][j],   j = tap*ci + c  (the OHWI weight row, J = taps*ci),
//             m over the N*Ho*Wo output pixels, X[m][j] = x[n, oy*s - p + r, ox*s - p + q, c] (0 outside).
//   GEMM-M = co, GEMM-N = j, GEMM-K = pixels.  Both operands are pixel-major in memory, so a K-tile of 32
//   pixels is staged as LDS rows [pixel][channel] and MFMA operands are read with ds_read_b32 (lane = channel).
// grid.x = (co tiles) * (j tiles), grid.y = K splits; each block reduces its pixel range in registers and
// adds its tile to dW with global_atomic_add_f32 (one 128-B row segment per half-wave = full atomic rate),
// or stores directly when there is a single split.
//
// Measured (profiles/r01_pmc_traffic.json): the L2 fetches 4-5x the algorithmic bytes here, because the tap tiles of one
// pixel slab are dealt round-robin to the 8 XCD L2s.  An XCD-aware remap that keeps a slab's tiles on one L2 was tried
// and ran 5 % SLOWER (80 -> 76 TFLOP/s): the re-fetches are served by the Infinity Cache and do not bound the kernel.
// Round 2 built and measured three alternatives on the r18 step (profiles/r02_wgrad_row_tiles.txt), all removed again:
//  * kernel-row tiles (one block per 64 co x 64 ci x three taps sharing one staged x row and the dy fragment: operand
//    traffic / 3, 4 LDS reads per 3 MFMAs, 48 MFMAs per barrier, XCD-aware order): 77-83 TFLOP/s against 95 here.  The
//    partial tiles of the pixel splits are reduced with fp32 atomics, 50 MB per launch for either tiling (38 us at the
//    chip's ~1.3 TB/s atomic rate): this kernel's 3072 small blocks run in three rounds and hide them (a timing-only build
//    with plain stores instead: 99.9 vs 95.0 TFLOP/s, the atomics cost 5 %), 1024 three-tap blocks finish together and
//    expose them;
//  * transposed staging (pixel-contiguous swizzled LDS rows through a register transpose, 16-byte fragment reads feeding
//    four MFMAs each): 95.4-95.6 vs 95.6 TFLOP/s -- the fragment-read width is not the bound;
//  * the same with loads two K-tiles ahead: 89.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "halo_common.h"

namespace udaseg {

struct WgradArgs {
  const void* x;    // fp32, or bf16 for conv_wgrad_bf16_kernel
  const void* dy;
  float* dw;
  int hi, wi, ci, ho, wo, co;
  int kw, stride, pad;
  int M, J;          // pixels, taps*ci
  int kchunk;        // pixels per split (multiple of 32)
  int use_atomic;
  float inv_ci, inv_kw, inv_wo, inv_ho;
  int row_uniform;   // host-side selector of the row-uniform gather (see conv_wgrad_kernel)
  unsigned x_bytes, dy_bytes;
  // A launch may produce only a channel SLICE of dW -- the weight gradient of a convolution over a virtual concatenation
  // (fused decoder input cat([up(a), skip])) is one launch per source: x is that source ([..][ci] with ci = ITS channel
  // count), its columns land at dw[co][tap * ci_full + c_off + c], dW rows are J_ld = taps * ci_full long.
  // up: the source sits at half resolution behind a nearest x2 up-sampling: pixel (iy, ix) reads (iy >> 1, ix >> 1).
  int ci_full, c_off, J_ld, up;
  // bf16 kernel only: x is a convolution output whose training-mode BatchNorm + activation was never written (its consumer applied
  // it while staging: udaseg_conv2d_fwd_frag_bf16 in_scale / in_shift); the gather applies the same transform
  // v -> act(fma(v, in_scale[c], in_shift[c])) rounded to bf16, padded taps stay zero.  null: x is used as it is.
  const float* in_scale;
  const float* in_shift;
  int in_act;
  float in_slope;
  // bf16 kernel: 1-D grid of tiles * splits blocks, remapped so that every (co tile, tap / channel tile) block of ONE pixel
  // split runs on the same XCD (blocks b and b + 8 share one): the nine tap tiles of a 3x3 layer gather the same x rows and
  // the same dy rows, and spread over eight L2s each XCD fetched its own copy -- 181 MB per launch at the fabric for 34 MB of
  // operands (PMC, profiles/r03_bf16_cfg3_pmc_traffic.txt), which at bf16 rates IS the launch time.  0: two-dimensional grid.
  int xcd_tiles, xcd_splits;
};

constexpr int WBK = 32;  // pixels per K-tile

// ROWU = true: row-uniform gather.  When the output width is a multiple of the rows one load pass covers, the pixels of a
// pass share their image row: (image, oy, first ox) are SCALARS updated once per K-tile, a thread's gather address is
// scalar + per-thread constant, and the bounds test is two adds and two compares; loads are range-checked buffer loads
// (invalid lanes get an out-of-range offset and read zeros).  The generic loop spends two divisions and a 64-bit address
// per gathered row per K-tile in the VALU, which on gfx950 shares its ALUs with the fp32 MFMA.
template <int BMW, int BNW, int WAVES_M, int WAVES_N, bool ROWU>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int WM = BMW / WAVES_M, WN = BNW / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int AQ = BMW / 4, BQ = BNW / 4;           // float4 columns per row
  constexpr int A_ROWS = 256 / AQ, B_ROWS = 256 / BQ;  // rows covered per pass
  constexpr int A_PASS = WBK / A_ROWS, B_PASS = WBK / B_ROWS;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(A_PASS >= 1 && B_PASS >= 1, "tile too wide");

  __shared__ __attribute__((aligned(16))) float As[2][WBK][BMW];
  __shared__ __attribute__((aligned(16))) float Bs[2][WBK][BNW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;

  const int ntj = (a.J + BNW - 1) / BNW;
  const int co0 = (blockIdx.x / ntj) * BMW;
  const int j0 = (blockIdx.x % ntj) * BNW;
  const int kbeg = blockIdx.y * a.kchunk;
  const int kend = min(a.M, kbeg + a.kchunk);

  // A (dy) slot: column aq (4 channels), rows arow + A_ROWS*p
  const int aq = tid % AQ, arow = tid / AQ;
  const int a_co = co0 + aq * 4;
  const bool a_ok = a_co < a.co;
  // B (x gather) slot: column bq -> fixed (tap, c)
  const int bq = tid % BQ, brow = tid / BQ;
  const int j = j0 + bq * 4;
  const bool b_ok = j < a.J;
  int b_dy = 0, b_dx = 0, b_c = 0;
  if (b_ok) {
    const int tap = fast_div(j, a.ci, a.inv_ci);
    b_c = j - tap * a.ci;
    const int r = fast_div(tap, a.kw, a.inv_kw);
    b_dy = r - a.pad;
    b_dx = (tap - r * a.kw) - a.pad;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][jn][v] = 0.f;

  f32x4 ra[A_PASS], rb[B_PASS];
  const int nkt = (kend - kbeg + WBK - 1) / WBK;

  // row-uniform state: per-thread constants and per-pass scalars of the NEXT tile to load (tiles are loaded in order)
  unsigned u_ac[A_PASS], u_bc = 0;
  int u_bdy = 0, u_bx = 0;
  int s_ni[B_PASS], s_oy[B_PASS], s_ox[B_PASS];
  unsigned s_dyoff = 0;
  __amdgpu_buffer_rsrc_t rsrc_x, rsrc_dy;
  if constexpr (ROWU) {
    rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);
#pragma unroll
    for (int p = 0; p < A_PASS; ++p)
      u_ac[p] = a_ok ? (unsigned)((arow + A_ROWS * p) * a.co + a_co) * 4u : 0x80000000u;
    u_bdy = b_dy;
    u_bx = brow * a.stride + b_dx;
    // up-sampled source: the column part of (ix >> 1) splits into scalar + constant because a pass starts at an even ox;
    // the row part (iy >> 1) depends on the parity of the scalar oy and is formed per load
    u_bc = !b_ok ? 0x80000000u
                 : a.up ? (unsigned)((u_bx >> 1) * a.ci + b_c) * 4u
                        : (unsigned)((b_dy * a.wi + brow * a.stride + b_dx) * a.ci + b_c) * 4u;
    s_dyoff = (unsigned)kbeg * (unsigned)a.co * 4u;
#pragma unroll
    for (int p = 0; p < B_PASS; ++p) {
      const int m = kbeg + B_ROWS * p;          // uniform
      const int t1 = m / a.wo;
      s_ox[p] = m - t1 * a.wo;
      s_ni[p] = t1 / a.ho;
      s_oy[p] = t1 - s_ni[p] * a.ho;
    }
  }

  auto load_tile = [&](int kt) {
    if constexpr (ROWU) {
#pragma unroll
      for (int p = 0; p < A_PASS; ++p)
        ra[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, (int)(u_ac[p] + s_dyoff), 0, 0));
      s_dyoff += (unsigned)(WBK * 4) * (unsigned)a.co;
#pragma unroll
      for (int p = 0; p < B_PASS; ++p) {
        const int oys = s_oy[p] * a.stride, oxs = s_ox[p] * a.stride;                                   // scalars
        const bool ok = (unsigned)(oys + u_bdy) < (unsigned)a.hi && (unsigned)(oxs + u_bx) < (unsigned)a.wi;
        unsigned voff;
        if (a.up) {   // uniform branch
          const int w2 = a.wi >> 1;
          const unsigned s_off = (unsigned)((s_ni[p] * (a.hi >> 1) * w2 + (oxs >> 1)) * a.ci) * 4u;   // scalar
          voff = u_bc + s_off + (unsigned)(((oys + u_bdy) >> 1) * w2 * a.ci) * 4u;
        } else {
          const unsigned s_off = (unsigned)(((s_ni[p] * a.hi + oys) * a.wi + oxs) * a.ci) * 4u;        // scalar
          voff = u_bc + s_off;
        }
        voff = ok ? voff : 0x80000000u;
        rb[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)voff, 0, 0));
        // advance this pass by one K-tile (32 pixels); wo is a multiple of B_ROWS, so a pass never straddles image rows
        s_ox[p] += WBK;
        while (s_ox[p] >= a.wo) {
          s_ox[p] -= a.wo;
          if (++s_oy[p] == a.ho) {
            s_oy[p] = 0;
            ++s_ni[p];
          }
        }
      }
      return;
    }
    const int mb = kbeg + kt * WBK;
#pragma unroll
    for (int p = 0; p < A_PASS; ++p) {
      const int m = mb + arow + A_ROWS * p;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (a_ok && m < kend) v = *reinterpret_cast<const f32x4*>(static_cast<const float*>(a.dy) + (size_t)m * a.co + a_co);
      ra[p] = v;
    }
#pragma unroll
    for (int p = 0; p < B_PASS; ++p) {
      const int m = mb + brow + B_ROWS * p;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (b_ok && m < kend) {
        const int t1 = fast_div(m, a.wo, a.inv_wo);
        const int ox = m - t1 * a.wo;
        const int ni = fast_div(t1, a.ho, a.inv_ho);
        const int oy = t1 - ni * a.ho;
        const int iy = oy * a.stride + b_dy, ix = ox * a.stride + b_dx;
        if ((unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi) {
          const size_t pix = a.up ? (size_t)(ni * (a.hi >> 1) + (iy >> 1)) * (a.wi >> 1) + (ix >> 1) : (size_t)(ni * a.hi + iy) * a.wi + ix;
          v = *reinterpret_cast<const f32x4*>(static_cast<const float*>(a.x) + pix * (size_t)a.ci + b_c);
        }
      }
      rb[p] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int p = 0; p < A_PASS; ++p) *reinterpret_cast<f32x4*>(&As[buf][arow + A_ROWS * p][aq * 4]) = ra[p];
#pragma unroll
    for (int p = 0; p < B_PASS; ++p) *reinterpret_cast<f32x4*>(&Bs[buf][brow + B_ROWS * p][bq * 4]) = rb[p];
  };

  if (nkt > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = (kt + 1) < nkt;
    if (more) load_tile(kt + 1);
#pragma unroll
    for (int k2 = 0; k2 < WBK / 2; ++k2) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = As[cur][2 * k2 + lh][wm + i * 32 + lr];
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) bf[jn] = Bs[cur][2 * k2 + lh][wn + jn * 32 + lr];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[jn], acc[i][jn], 0, 0, 0);
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // D reg v of lane (lr, lh): row co = (v&3) + 8*(v>>2) + 4*lh, col j = lr
  int col[TN];   // this lane's dW column per tile: (tap, c) of the launch's source -> tap * ci_full + c_off + c
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int jj = j0 + wn + jn * 32 + lr;
    const int tap = fast_div(jj, a.ci, a.inv_ci);
    col[jn] = jj < a.J ? tap * a.ci_full + a.c_off + (jj - tap * a.ci) : -1;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int co = co0 + wm + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
      if (co >= a.co) continue;
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        if (col[jn] >= 0) {
          float* dst = a.dw + (size_t)co * a.J_ld + col[jn];
          if (a.use_atomic) atomicAdd(dst, acc[i][jn][v]);
          else *dst = acc[i][jn][v];
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------- bf16 storage
// Same GEMM with bf16 x / dy and fp32 dW (master gradients).  GEMM-K is the pixel axis = the STRIDED axis of both operands
// ([pixel][channel] in memory and in LDS), while v_mfma_f32_32x32x16_bf16 wants 8 consecutive K values per lane: fragments
// are read with ds_read_b64_tr_b16 (per 16 lanes: a 4-pixel x 16-channel block delivered channel-major), two reads per
// fragment.  LDS rows carry a 16-byte pad so the 4 rows of a block fall on distinct banks.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
constexpr int WBKB = 64;  // pixels per K-tile (four 16-deep MFMA steps per barrier; 32 left the kernel barrier-bound)

__device__ __forceinline__ bf16x8w tr_fragment(const unsigned short* row0, int ld) {
  // row0: this lane's address in the first 4-row block; the second block is 4 rows further down
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(row0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(row0 + 4 * ld));
  const s16x8 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8w, f);
}

template <int BMW, int BNW, int WAVES_M, int WAVES_N, bool ROWU>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(const WgradArgs a) {
  constexpr int WM = BMW / WAVES_M, WN = BNW / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDA = BMW + 8, LDB = BNW + 8;         // elements per LDS row (16-byte pad)
  constexpr int AQ = BMW / 8, BQ = BNW / 8;            // 16-byte vectors per row
  constexpr int A_ROWS = 256 / AQ, B_ROWS = 256 / BQ;
  constexpr int A_PASS = (WBKB + A_ROWS - 1) / A_ROWS, B_PASS = (WBKB + B_ROWS - 1) / B_ROWS;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");

  __shared__ __attribute__((aligned(16))) unsigned short As[2][WBKB][LDA];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[2][WBKB][LDB];
  const unsigned short* xg = reinterpret_cast<const unsigned short*>(a.x);
  const unsigned short* dyg = reinterpret_cast<const unsigned short*>(a.dy);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;
  // transposed-read roles inside the 16-lane group
  const int grp = lane >> 4, cb = 16 * (grp & 1), hk = grp >> 1, tq = (lane & 15) >> 2, tp = lane & 3;

  const int ntj = (a.J + BNW - 1) / BNW;
  int tile_id = blockIdx.x, split_id = blockIdx.y;
  if (a.xcd_tiles > 0) {        // see WgradArgs::xcd_tiles
    const int L = blockIdx.x, xcd = L & 7, idx = L >> 3;
    split_id = xcd + 8 * (idx / a.xcd_tiles);
    tile_id = idx % a.xcd_tiles;
  }
  const int co0 = (tile_id / ntj) * BMW;
  const int j0 = (tile_id % ntj) * BNW;
  const int kbeg = split_id * a.kchunk;
  const int kend = min(a.M, kbeg + a.kchunk);
  if (kbeg >= a.M) return;          // an empty split of a rounded-up split count (block-uniform, before any barrier)

  const int aq = tid % AQ, arow = tid / AQ;
  const int a_co = co0 + aq * 8;
  const bool a_ok = a_co < a.co;
  const int bq = tid % BQ, brow = tid / BQ;
  const int j = j0 + bq * 8;
  const bool b_ok = j < a.J;
  int b_dy = 0, b_dx = 0, b_c = 0;
  if (b_ok) {
    const int tap = fast_div(j, a.ci, a.inv_ci);
    b_c = j - tap * a.ci;
    const int r = fast_div(tap, a.kw, a.inv_kw);
    b_dy = r - a.pad;
    b_dx = (tap - r * a.kw) - a.pad;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][jn][v] = 0.f;

  f32x4 ra[A_PASS], rb[B_PASS];
  const int nkt = (kend - kbeg + WBKB - 1) / WBKB;
  // producer's BatchNorm + activation on the gathered operand (WgradArgs::in_scale): this thread's 8 channels are fixed
  const bool xform = a.in_scale != nullptr;
  const bool x_relu = a.in_act == UDASEG_ACT_LEAKY && a.in_slope == 0.f;
  float x_sc[8], x_sh[8];
  unsigned x_ok = 0;          // bit p: pass p of the tile in the registers is inside the image
  if (xform) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      x_sc[e] = b_ok ? a.in_scale[b_c + e] : 0.f;
      x_sh[e] = b_ok ? a.in_shift[b_c + e] : 0.f;
    }
  }

  // row-uniform gather (see conv_wgrad_kernel): a pass covers B_ROWS consecutive pixels of one image row
  static_assert(!ROWU || (WBKB % A_ROWS == 0 && WBKB % B_ROWS == 0), "row-uniform passes must tile the K-tile");
  unsigned u_ac[A_PASS], u_bc = 0;
  int u_bdy = 0, u_bx = 0;
  int s_ni[B_PASS], s_oy[B_PASS], s_ox[B_PASS];
  unsigned s_dyoff = 0;
  __amdgpu_buffer_rsrc_t rsrc_x, rsrc_dy;
  if constexpr (ROWU) {
    rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);
#pragma unroll
    for (int p = 0; p < A_PASS; ++p)
      u_ac[p] = a_ok ? (unsigned)((arow + A_ROWS * p) * a.co + a_co) * 2u : 0x80000000u;
    u_bdy = b_dy;
    u_bx = brow * a.stride + b_dx;
    // up-sampled source: the column part of (ix >> 1) splits into scalar + constant because a pass starts at an even ox;
    // the row part (iy >> 1) depends on the parity of the scalar oy and is formed per load
    u_bc = !b_ok ? 0x80000000u
                 : a.up ? (unsigned)((u_bx >> 1) * a.ci + b_c) * 2u
                        : (unsigned)((b_dy * a.wi + brow * a.stride + b_dx) * a.ci + b_c) * 2u;
    s_dyoff = (unsigned)kbeg * (unsigned)a.co * 2u;
#pragma unroll
    for (int p = 0; p < B_PASS; ++p) {
      const int m = kbeg + B_ROWS * p;
      const int t1 = m / a.wo;
      s_ox[p] = m - t1 * a.wo;
      s_ni[p] = t1 / a.ho;
      s_oy[p] = t1 - s_ni[p] * a.ho;
    }
  }

  auto load_tile = [&](int kt) {
    if constexpr (ROWU) {
#pragma unroll
      for (int p = 0; p < A_PASS; ++p)
        ra[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, (int)(u_ac[p] + s_dyoff), 0, 0));
      s_dyoff += (unsigned)(WBKB * 2) * (unsigned)a.co;
#pragma unroll
      for (int p = 0; p < B_PASS; ++p) {
        const int oys = s_oy[p] * a.stride, oxs = s_ox[p] * a.stride;
        const bool ok = (unsigned)(oys + u_bdy) < (unsigned)a.hi && (unsigned)(oxs + u_bx) < (unsigned)a.wi;
        unsigned voff;
        if (a.up) {   // uniform branch
          const int w2 = a.wi >> 1;
          const unsigned s_off = (unsigned)((s_ni[p] * (a.hi >> 1) * w2 + (oxs >> 1)) * a.ci) * 2u;
          voff = u_bc + s_off + (unsigned)(((oys + u_bdy) >> 1) * w2 * a.ci) * 2u;
        } else {
          const unsigned s_off = (unsigned)(((s_ni[p] * a.hi + oys) * a.wi + oxs) * a.ci) * 2u;
          voff = u_bc + s_off;
        }
        voff = ok ? voff : 0x80000000u;
        x_ok = (x_ok & ~(1u << p)) | ((ok && b_ok ? 1u : 0u) << p);
        rb[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)voff, 0, 0));
        s_ox[p] += WBKB;
        while (s_ox[p] >= a.wo) {
          s_ox[p] -= a.wo;
          if (++s_oy[p] == a.ho) {
            s_oy[p] = 0;
            ++s_ni[p];
          }
        }
      }
      return;
    }
    const int mb = kbeg + kt * WBKB;
#pragma unroll
    for (int p = 0; p < A_PASS; ++p) {
      const int row = arow + A_ROWS * p, m = mb + row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (a_ok && row < WBKB && m < kend) v = *reinterpret_cast<const f32x4*>(dyg + (size_t)m * a.co + a_co);
      ra[p] = v;
    }
#pragma unroll
    for (int p = 0; p < B_PASS; ++p) {
      const int row = brow + B_ROWS * p, m = mb + row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      bool inside = false;
      if (b_ok && row < WBKB && m < kend) {
        const int t1 = fast_div(m, a.wo, a.inv_wo);
        const int ox = m - t1 * a.wo;
        const int ni = fast_div(t1, a.ho, a.inv_ho);
        const int oy = t1 - ni * a.ho;
        const int iy = oy * a.stride + b_dy, ix = ox * a.stride + b_dx;
        if ((unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi) {
          const size_t pix = a.up ? (size_t)(ni * (a.hi >> 1) + (iy >> 1)) * (a.wi >> 1) + (ix >> 1) : (size_t)(ni * a.hi + iy) * a.wi + ix;
          v = *reinterpret_cast<const f32x4*>(xg + pix * (size_t)a.ci + b_c);
          inside = true;
        }
      }
      x_ok = (x_ok & ~(1u << p)) | ((inside ? 1u : 0u) << p);
      rb[p] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int p = 0; p < A_PASS; ++p)
      if (arow + A_ROWS * p < WBKB) *reinterpret_cast<f32x4*>(&As[buf][arow + A_ROWS * p][aq * 8]) = ra[p];
    if (xform) {
#pragma unroll
      for (int p = 0; p < B_PASS; ++p) {
        typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
        const u32x4w wv = __builtin_bit_cast(u32x4w, rb[p]);
        u32x4w dv;
        const bool ok = (x_ok >> p) & 1u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned w = wv[e];
          float t0 = __builtin_fmaf(__builtin_bit_cast(float, w << 16), x_sc[2 * e], x_sh[2 * e]);
          float t1 = __builtin_fmaf(__builtin_bit_cast(float, w & 0xffff0000u), x_sc[2 * e + 1], x_sh[2 * e + 1]);
          if (x_relu) {               // uniform: ReLU is one v_max (the general form is compare + multiply + select)
            t0 = t0 > 0.f ? t0 : 0.f;
            t1 = t1 > 0.f ? t1 : 0.f;
          } else {
            t0 = act_apply(t0, a.in_act, a.in_slope);
            t1 = act_apply(t1, a.in_act, a.in_slope);
          }
          dv[e] = ok ? ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)t0) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)t1) << 16))
                     : 0u;
        }
        rb[p] = __builtin_bit_cast(f32x4, dv);
      }
    }
#pragma unroll
    for (int p = 0; p < B_PASS; ++p)
      if (brow + B_ROWS * p < WBKB) *reinterpret_cast<f32x4*>(&Bs[buf][brow + B_ROWS * p][bq * 8]) = rb[p];
  };

  if (nkt > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = (kt + 1) < nkt;
    if (more) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < WBKB / 16; ++ks) {
      bf16x8w af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = tr_fragment(&As[cur][16 * ks + 8 * hk + tq][wm + 32 * i + cb + 4 * tp], LDA);
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) bfr[jn] = tr_fragment(&Bs[cur][16 * ks + 8 * hk + tq][wn + 32 * jn + cb + 4 * tp], LDB);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[jn], acc[i][jn], 0, 0, 0);
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  int col[TN];   // this lane's dW column per tile: (tap, c) of the launch's source -> tap * ci_full + c_off + c
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int jj = j0 + wn + jn * 32 + lr;
    const int tap = fast_div(jj, a.ci, a.inv_ci);
    col[jn] = jj < a.J ? tap * a.ci_full + a.c_off + (jj - tap * a.ci) : -1;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int co = co0 + wm + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
      if (co >= a.co) continue;
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        if (col[jn] >= 0) {
          float* dst = a.dw + (size_t)co * a.J_ld + col[jn];
          if (a.use_atomic) atomicAdd(dst, acc[i][jn][v]);
          else *dst = acc[i][jn][v];
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------- fp32 on the bf16 pipe
// The same GEMM with fp32 x / dy / dW and the products on v_mfma_f32_32x32x16_bf16 (round 4): a staged 8-channel piece (two 16-byte
// fp32 loads) is split exactly into three bf16 terms (halo_common.h split3) and written to three LDS planes laid out like the bf16
// kernel's tiles; fragments are the bf16 kernel's transposed reads, once per plane, six MFMAs per fragment pair (smallest terms first).
// For the weight gradients conv_wgrad_halo2.hip does not take: stride 2, the 7x7 stem, 1x1 / stride 2, produced channels that 32 does
// not divide.  K-tile: 32 pixels (two 16-deep steps = 12 MFMAs per wave and barrier; 55 KB of LDS for the 64 x 64 tile).
constexpr int WBKX = 32;

template <int BMW, int BNW, int WAVES_M, int WAVES_N, bool ROWU>
__global__ __launch_bounds__(256) void conv_wgrad_x3_kernel(const WgradArgs a) {
  constexpr int WM = BMW / WAVES_M, WN = BNW / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDA = BMW + 8, LDB = BNW + 8;         // bf16 elements per LDS row (16-byte pad)
  constexpr int AQ = BMW / 8, BQ = BNW / 8;            // 8-channel pieces per row
  constexpr int A_ROWS = 256 / AQ, B_ROWS = 256 / BQ;
  constexpr int A_PASS = (WBKX + A_ROWS - 1) / A_ROWS, B_PASS = (WBKX + B_ROWS - 1) / B_ROWS;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(!ROWU || (WBKX % A_ROWS == 0 && WBKX % B_ROWS == 0), "row-uniform passes must tile the K-tile");

  __shared__ __attribute__((aligned(16))) unsigned short As[3][2][WBKX][LDA];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[3][2][WBKX][LDB];
  const float* xg = reinterpret_cast<const float*>(a.x);
  const float* dyg = reinterpret_cast<const float*>(a.dy);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;
  const int grp = lane >> 4, cb = 16 * (grp & 1), hk = grp >> 1, tq = (lane & 15) >> 2, tp = lane & 3;

  const int ntj = (a.J + BNW - 1) / BNW;
  const int co0 = (blockIdx.x / ntj) * BMW;
  const int j0 = (blockIdx.x % ntj) * BNW;
  const int kbeg = blockIdx.y * a.kchunk;
  const int kend = min(a.M, kbeg + a.kchunk);
  if (kbeg >= a.M) return;

  const int aq = tid % AQ, arow = tid / AQ;
  const int a_co = co0 + aq * 8;
  const bool a_ok = a_co < a.co;
  const int bq = tid % BQ, brow = tid / BQ;
  const int j = j0 + bq * 8;
  const bool b_ok = j < a.J;
  int b_dy = 0, b_dx = 0, b_c = 0;
  if (b_ok) {
    const int tap = fast_div(j, a.ci, a.inv_ci);
    b_c = j - tap * a.ci;
    const int r = fast_div(tap, a.kw, a.inv_kw);
    b_dy = r - a.pad;
    b_dx = (tap - r * a.kw) - a.pad;
  }
  // an 8-wide piece of the J axis may straddle two taps when ci is not a multiple of 8 (the stem's 4 padded channels): its two
  // 4-channel halves are gathered separately
  const bool b_two = (a.ci & 7) != 0;
  int b2_dy = 0, b2_dx = 0, b2_c = 0;
  const bool b2_ok = b_two && (j + 4) < a.J;
  if (b2_ok) {
    const int tap = fast_div(j + 4, a.ci, a.inv_ci);
    b2_c = (j + 4) - tap * a.ci;
    const int r = fast_div(tap, a.kw, a.inv_kw);
    b2_dy = r - a.pad;
    b2_dx = (tap - r * a.kw) - a.pad;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][jn][v] = 0.f;

  // odd K-tiles (LDS buffer 1) carry -dy and accumulate into acc2, dW = acc - acc2: the bf16 MFMA adder truncates toward minus
  // infinity, and a bias over up to 2^19 pixels is what this sum must not have (conv_igemm.hip, X3)
  f32x16 acc2[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc2[i][jn][v] = 0.f;

  u32x4 ra[A_PASS][2], rb[B_PASS][2];
  const int nkt = (kend - kbeg + WBKX - 1) / WBKX;

  unsigned u_ac[A_PASS], u_bc = 0;
  int u_bdy = 0, u_bx = 0;
  int s_ni[B_PASS], s_oy[B_PASS], s_ox[B_PASS];
  unsigned s_dyoff = 0;
  __amdgpu_buffer_rsrc_t rsrc_x, rsrc_dy;
  if constexpr (ROWU) {
    rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);
#pragma unroll
    for (int p = 0; p < A_PASS; ++p)
      u_ac[p] = a_ok ? (unsigned)((arow + A_ROWS * p) * a.co + a_co) * 4u : 0x80000000u;
    u_bdy = b_dy;
    u_bx = brow * a.stride + b_dx;
    u_bc = !b_ok ? 0x80000000u
                 : a.up ? (unsigned)((u_bx >> 1) * a.ci + b_c) * 4u
                        : (unsigned)((b_dy * a.wi + brow * a.stride + b_dx) * a.ci + b_c) * 4u;
    s_dyoff = (unsigned)kbeg * (unsigned)a.co * 4u;
#pragma unroll
    for (int p = 0; p < B_PASS; ++p) {
      const int m = kbeg + B_ROWS * p;
      const int t1 = m / a.wo;
      s_ox[p] = m - t1 * a.wo;
      s_ni[p] = t1 / a.ho;
      s_oy[p] = t1 - s_ni[p] * a.ho;
    }
  }

  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load_tile = [&](int kt) {
    if constexpr (ROWU) {      // host: ci a multiple of 8 (a piece never straddles taps)
#pragma unroll
      for (int p = 0; p < A_PASS; ++p) {
        ra[p][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, (int)(u_ac[p] + s_dyoff), 0, 0);
        ra[p][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, (int)(u_ac[p] + s_dyoff), 16, 0);
      }
      s_dyoff += (unsigned)(WBKX * 4) * (unsigned)a.co;
#pragma unroll
      for (int p = 0; p < B_PASS; ++p) {
        const int oys = s_oy[p] * a.stride, oxs = s_ox[p] * a.stride;
        const bool ok = (unsigned)(oys + u_bdy) < (unsigned)a.hi && (unsigned)(oxs + u_bx) < (unsigned)a.wi;
        unsigned voff;
        if (a.up) {   // uniform branch
          const int w2 = a.wi >> 1;
          const unsigned s_off = (unsigned)((s_ni[p] * (a.hi >> 1) * w2 + (oxs >> 1)) * a.ci) * 4u;
          voff = u_bc + s_off + (unsigned)(((oys + u_bdy) >> 1) * w2 * a.ci) * 4u;
        } else {
          const unsigned s_off = (unsigned)(((s_ni[p] * a.hi + oys) * a.wi + oxs) * a.ci) * 4u;
          voff = u_bc + s_off;
        }
        voff = ok ? voff : 0x80000000u;
        rb[p][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)voff, 0, 0);
        rb[p][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)voff, 16, 0);
        s_ox[p] += WBKX;
        while (s_ox[p] >= a.wo) {
          s_ox[p] -= a.wo;
          if (++s_oy[p] == a.ho) {
            s_oy[p] = 0;
            ++s_ni[p];
          }
        }
      }
      return;
    }
    const int mb = kbeg + kt * WBKX;
#pragma unroll
    for (int p = 0; p < A_PASS; ++p) {
      const int row = arow + A_ROWS * p, m = mb + row;
      u32x4 v0 = zero4, v1 = zero4;
      if (a_ok && row < WBKX && m < kend) {
        const u32x4* src = reinterpret_cast<const u32x4*>(dyg + (size_t)m * a.co + a_co);
        v0 = src[0];
        if (a_co + 4 < a.co) v1 = src[1];
      }
      ra[p][0] = v0;
      ra[p][1] = v1;
    }
#pragma unroll
    for (int p = 0; p < B_PASS; ++p) {
      const int row = brow + B_ROWS * p, m = mb + row;
      u32x4 v0 = zero4, v1 = zero4;
      if (b_ok && row < WBKX && m < kend) {
        const int t1 = fast_div(m, a.wo, a.inv_wo);
        const int ox = m - t1 * a.wo;
        const int ni = fast_div(t1, a.ho, a.inv_ho);
        const int oy = t1 - ni * a.ho;
        {
          const int iy = oy * a.stride + b_dy, ix = ox * a.stride + b_dx;
          if ((unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi) {
            const size_t pix = a.up ? (size_t)(ni * (a.hi >> 1) + (iy >> 1)) * (a.wi >> 1) + (ix >> 1) : (size_t)(ni * a.hi + iy) * a.wi + ix;
            const u32x4* src = reinterpret_cast<const u32x4*>(xg + pix * (size_t)a.ci + b_c);
            v0 = src[0];
            if (!b_two && j + 4 < a.J) v1 = src[1];
          }
        }
        if (b2_ok) {
          const int iy = oy * a.stride + b2_dy, ix = ox * a.stride + b2_dx;
          if ((unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi) {
            const size_t pix = a.up ? (size_t)(ni * (a.hi >> 1) + (iy >> 1)) * (a.wi >> 1) + (ix >> 1) : (size_t)(ni * a.hi + iy) * a.wi + ix;
            v1 = *reinterpret_cast<const u32x4*>(xg + pix * (size_t)a.ci + b2_c);
          }
        }
      }
      rb[p][0] = v0;
      rb[p][1] = v1;
    }
  };
  auto store_tile = [&](auto BUF) {
    constexpr int buf = decltype(BUF)::value;
#pragma unroll
    for (int p = 0; p < A_PASS; ++p)
      if (arow + A_ROWS * p < WBKX) {
        u32x4 p0, p1, p2;
        split3(ra[p][0], ra[p][1], p0, p1, p2);
        if constexpr (buf != 0) {
          p0 ^= 0x80008000u;
          p1 ^= 0x80008000u;
          p2 ^= 0x80008000u;
        }
        *reinterpret_cast<u32x4*>(&As[0][buf][arow + A_ROWS * p][aq * 8]) = p0;
        *reinterpret_cast<u32x4*>(&As[1][buf][arow + A_ROWS * p][aq * 8]) = p1;
        *reinterpret_cast<u32x4*>(&As[2][buf][arow + A_ROWS * p][aq * 8]) = p2;
      }
#pragma unroll
    for (int p = 0; p < B_PASS; ++p)
      if (brow + B_ROWS * p < WBKX) {
        u32x4 p0, p1, p2;
        split3(rb[p][0], rb[p][1], p0, p1, p2);
        *reinterpret_cast<u32x4*>(&Bs[0][buf][brow + B_ROWS * p][bq * 8]) = p0;
        *reinterpret_cast<u32x4*>(&Bs[1][buf][brow + B_ROWS * p][bq * 8]) = p1;
        *reinterpret_cast<u32x4*>(&Bs[2][buf][brow + B_ROWS * p][bq * 8]) = p2;
      }
  };
  // one K-tile: request the next tile, multiply the one in LDS buffer CUR (even tiles -> acc, odd tiles -> acc2), stage the next
  auto step = [&](auto CUR, int kt) {
    constexpr int cur = decltype(CUR)::value;
    const bool more = (kt + 1) < nkt;
    if (more) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < WBKX / 16; ++ks) {
      bf16x8w af[3][TM], bfr[3][TN];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[pl][i] = tr_fragment(&As[pl][cur][16 * ks + 8 * hk + tq][wm + 32 * i + cb + 4 * tp], LDA);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) bfr[pl][jn] = tr_fragment(&Bs[pl][cur][16 * ks + 8 * hk + tq][wn + 32 * jn + cb + 4 * tp], LDB);
      }
#pragma unroll
      for (int pq = 2; pq >= 0; --pq)
#pragma unroll
        for (int pa = 0; pa <= pq; ++pa)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
              if constexpr (cur != 0)
                acc2[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa][i], bfr[pq - pa][jn], acc2[i][jn], 0, 0, 0);
              else
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa][i], bfr[pq - pa][jn], acc[i][jn], 0, 0, 0);
            }
    }
    if (more) store_tile(std::integral_constant<int, cur ^ 1>{});
    __syncthreads();
  };

  if (nkt > 0) {
    load_tile(0);
    store_tile(std::integral_constant<int, 0>{});
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; kt += 2) {
    step(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nkt) step(std::integral_constant<int, 1>{}, kt + 1);
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) acc[i][jn] -= acc2[i][jn];
  int col[TN];
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int jj = j0 + wn + jn * 32 + lr;
    const int tap = fast_div(jj, a.ci, a.inv_ci);
    col[jn] = jj < a.J ? tap * a.ci_full + a.c_off + (jj - tap * a.ci) : -1;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int co = co0 + wm + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
      if (co >= a.co) continue;
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        if (col[jn] >= 0) {
          float* dst = a.dw + (size_t)co * a.J_ld + col[jn];
          if (a.use_atomic) atomicAdd(dst, acc[i][jn][v]);
          else *dst = acc[i][jn][v];
        }
      }
    }
}

// blocks of an X3 launch: the MFMA phase of a split is ~2.5x shorter than on the fp32 pipe, the fp32 atomics of the partial tiles
// are not -- fewer, longer splits than launch_wgrad's 3072 (UDASEG_WGRAD_X3_BLOCKS; profiles/r04_igemm_x3.txt: 512 / 1024 / 2048 / 3072 blocks gave the same step)
template <int BMW, int BNW, int WAVES_M, int WAVES_N>
static int launch_wgrad_x3(WgradArgs a, int accumulate, hipStream_t s) {
  const int tiles = cdiv(a.co, BMW) * cdiv(a.J, BNW);
  int target = opt_get(UDASEG_OPT_WGRAD_X3_BLOCKS);
  if (target < 64) target = 1024;
  int splits = cdiv(target, tiles);
  const int max_splits = cdiv(a.M, 256);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  const int kchunk = cdiv(cdiv(a.M, splits), WBKX) * WBKX;
  splits = cdiv(a.M, kchunk);
  a.kchunk = kchunk;
  a.use_atomic = (splits > 1 || accumulate) ? 1 : 0;
  if (a.use_atomic && !accumulate) {
    if (a.J != a.J_ld) {
      set_error("conv2d_wgrad_part: a channel slice of dW must be accumulated onto a caller-zeroed gradient");
      return UDASEG_E_BADARG;
    }
    hipError_t e = hipMemsetAsync(a.dw, 0, (size_t)a.co * a.J * sizeof(float), s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(dw)");
  }
  dim3 grid((unsigned)tiles, (unsigned)splits), block(256);
  a.xcd_tiles = a.xcd_splits = 0;
  constexpr int b_rows = 256 / (BNW / 8);
  const long long xb = (long long)a.M / (a.ho * a.wo) * (a.up ? (a.hi >> 1) * (a.wi >> 1) : a.hi * a.wi) * a.ci * 4,
                  dyb = (long long)a.M * a.co * 4;
  a.row_uniform = !opt_get(UDASEG_OPT_WGRAD_GENERIC) && a.wo % b_rows == 0 && a.ci % 8 == 0 && a.co % 8 == 0 &&
                  xb <= (1LL << 30) && dyb <= (1LL << 30);
  a.x_bytes = (unsigned)xb;
  a.dy_bytes = (unsigned)dyb;
  hipEvent_t ev = kprof_begin(s);
  if (a.row_uniform)
    hipLaunchKernelGGL((conv_wgrad_x3_kernel<BMW, BNW, WAVES_M, WAVES_N, true>), grid, block, 0, s, a);
  else
    hipLaunchKernelGGL((conv_wgrad_x3_kernel<BMW, BNW, WAVES_M, WAVES_N, false>), grid, block, 0, s, a);
  static std::atomic<int> kidx[2] = {{-1}, {-1}};
  std::atomic<int>& kid = kidx[a.row_uniform ? 1 : 0];
  if (kid < 0) {
    char nm[96];
    snprintf(nm, sizeof(nm), "conv_wgrad_x3_kernel<%d, %d, %d, %d, %s>", BMW, BNW, WAVES_M, WAVES_N, a.row_uniform ? "true" : "false");
    kid = kprof_id(nm);
  }
  kprof_end(kid, ev, s, 2.0 * (double)a.M * a.co * a.J);
  UDASEG_LAUNCH_CHECK("conv_wgrad_x3 launch");
  return UDASEG_OK;
}

template <int BMW, int BNW, int WAVES_M, int WAVES_N>
static int launch_wgrad(WgradArgs a, int accumulate, hipStream_t s, bool bf16 = false) {
  const int tiles = cdiv(a.co, BMW) * cdiv(a.J, BNW);
  // enough blocks for ~4 per CU, at least 256 pixels per split (UDASEG_WGRAD_BLOCKS: tuning aid)
  int target = opt_get(UDASEG_OPT_WGRAD_BLOCKS);      // 0: the defaults below
  const bool target_from_env = target >= 64;
  if (!target_from_env) target = 3072;  // measured sweep 512..8192: 68.7 / 79.4 / 80.3 / 85.9 / 87.0 / 89.1 / 88.8 / 88.1 TFLOP/s
  // bf16: the MFMA phase of a split is 16x shorter, so the fp32 atomics of the partial tiles (blocks x 16 KB at ~1.3 TB/s)
  // are the launch: 512 blocks measured best (r18 8x512^2 step: 256 / 512 / 768 / 1024 / 2048 / 3072 blocks ->
  // 1285 / 1398 / 1366 / 1353 / 1326 / 1285 images/s; r50 768^2: 337 / 344 / 338 / 335 / 336 / 332)
  // (the <= 32-output-channel full-resolution layers -- 32 x 128 tile, 16 KB of partials per block, >= 1 M pixels -- are the
  // exception: latency-bound at two blocks per CU, 2048 blocks: 185 -> 152, 138 -> 120, 128 -> 109 us at 512^2)
  const int bf16_target = (BMW == 32 && a.M >= (1 << 20)) ? 2048 : 512;
  int splits = cdiv(bf16 && !target_from_env ? bf16_target : target, tiles);
  const int max_splits = cdiv(a.M, 256);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int ktile = bf16 ? WBKB : WBK;
  int kchunk = cdiv(cdiv(a.M, splits), ktile) * ktile;
  splits = cdiv(a.M, kchunk);
  if (bf16 && splits >= 16 && tiles > 1 && splits % 8 != 0) {
    // the XCD-aware block order (WgradArgs::xcd_tiles) needs a multiple of 8 splits: take the nearest one that still tiles M
    // (a split may come out empty when M / ktile is not divisible: its block returns at once)
    const int s8 = ((splits + 4) / 8) * 8;
    const int kc = cdiv(cdiv(a.M, s8), ktile) * ktile;
    if ((long long)kc * (s8 - 8) < a.M) {     // at most the last few splits are empty
      kchunk = kc;
      splits = s8;
    }
  }
  a.kchunk = kchunk;
  a.use_atomic = (splits > 1 || accumulate) ? 1 : 0;
  if (a.use_atomic && !accumulate) {
    if (a.J != a.J_ld) {
      set_error("conv2d_wgrad_part: a channel slice of dW must be accumulated onto a caller-zeroed gradient");
      return UDASEG_E_BADARG;
    }
    hipError_t e = hipMemsetAsync(a.dw, 0, (size_t)a.co * a.J * sizeof(float), s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(dw)");
  }
  dim3 grid((unsigned)tiles, (unsigned)splits), block(256);
  a.xcd_tiles = a.xcd_splits = 0;
  if (bf16 && splits % 8 == 0 && tiles > 1) {
    if (!opt_get(UDASEG_OPT_WGRAD_NO_XCD)) {      // 1: A/B
      a.xcd_tiles = tiles;
      a.xcd_splits = splits;
      grid = dim3((unsigned)(tiles * splits), 1u);
    }
  }
  hipEvent_t ev = kprof_begin(s);
  if (bf16) {
    constexpr int b_rows16 = 256 / (BNW / 8);
    const long long xb16 = (long long)a.M / (a.ho * a.wo) * (a.up ? (a.hi >> 1) * (a.wi >> 1) : a.hi * a.wi) * a.ci * 2,
                    dyb16 = (long long)a.M * a.co * 2;
    a.row_uniform = !opt_get(UDASEG_OPT_WGRAD_GENERIC) && a.wo % b_rows16 == 0 && xb16 <= (1LL << 30) && dyb16 <= (1LL << 30);
    a.x_bytes = (unsigned)xb16;
    a.dy_bytes = (unsigned)dyb16;
    if (a.row_uniform)
      hipLaunchKernelGGL((conv_wgrad_bf16_kernel<BMW, BNW, WAVES_M, WAVES_N, true>), grid, block, 0, s, a);
    else
      hipLaunchKernelGGL((conv_wgrad_bf16_kernel<BMW, BNW, WAVES_M, WAVES_N, false>), grid, block, 0, s, a);
    static std::atomic<int> kid16[2] = {{-1}, {-1}};     // one id per rocprofv3 symbol (round 2 lumped the bf16 instantiations into one)
    std::atomic<int>& kid = kid16[a.row_uniform ? 1 : 0];
    if (kid < 0) {
      char nm[96];
      snprintf(nm, sizeof(nm), "conv_wgrad_bf16_kernel<%d, %d, %d, %d, %s>", BMW, BNW, WAVES_M, WAVES_N, a.row_uniform ? "true" : "false");
      kid = kprof_id(nm);
    }
    kprof_end(kid, ev, s, 2.0 * (double)a.M * a.co * a.J);
  } else {
    constexpr int b_rows = 256 / (BNW / 4);
    const long long xb = (long long)a.M / (a.ho * a.wo) * (a.up ? (a.hi >> 1) * (a.wi >> 1) : a.hi * a.wi) * a.ci * 4,
                    dyb = (long long)a.M * a.co * 4;
    a.row_uniform = !opt_get(UDASEG_OPT_WGRAD_GENERIC) && a.wo % b_rows == 0 && xb <= (1LL << 30) && dyb <= (1LL << 30);
    a.x_bytes = (unsigned)xb;
    a.dy_bytes = (unsigned)dyb;
    if (a.row_uniform)
      hipLaunchKernelGGL((conv_wgrad_kernel<BMW, BNW, WAVES_M, WAVES_N, true>), grid, block, 0, s, a);
    else
      hipLaunchKernelGGL((conv_wgrad_kernel<BMW, BNW, WAVES_M, WAVES_N, false>), grid, block, 0, s, a);
    int kid32 = (a.row_uniform ? 18 : 7) + (BMW == 64 ? 0 : 1);      // the fixed table knows the 64 x 64 and 32 x 128 symbols
    if (BMW == 128) {
      static std::atomic<int> kid128[2] = {{-1}, {-1}};
      std::atomic<int>& k = kid128[a.row_uniform ? 1 : 0];
      if (k < 0) {
        char nm[96];
        snprintf(nm, sizeof(nm), "conv_wgrad_kernel<%d, %d, %d, %d, %s>", BMW, BNW, WAVES_M, WAVES_N, a.row_uniform ? "true" : "false");
        k = kprof_id(nm);
      }
      kid32 = k;
    }
    kprof_end(kid32, ev, s, 2.0 * (double)a.M * a.co * a.J);
  }
  UDASEG_LAUNCH_CHECK("conv_wgrad launch");
  return UDASEG_OK;
}


// The halo-resident weight gradients (bf16 and the fp32 three-term split) live in conv_wgrad_halo2.hip since round 4; round 3's
// kernels (conv_wgrad_halo_bf16_kernel, conv_wgrad_halo_f32x3_kernel: 64 x 64 blocks only, one x fragment read per tap, a 144-byte LDS
// pitch with two-way bank conflicts on every transposed read) were removed when the second form had replaced them on every shape.
}  // namespace udaseg

using namespace udaseg;

static bool wgrad_x3_on() {      // UDASEG_OPT_WGRAD_X3 = 0 (A/B) or UDASEG_OPT_F32_SPLIT = 0: the fp32-pipe kernel
  return udaseg::opt_get(UDASEG_OPT_WGRAD_X3) != 0 && udaseg::f32_split_enabled();
}

// One implementation behind the four entry points.  src_c / c_off / up describe a channel slice of dW (WgradArgs::ci_full):
// the whole gradient is src_c == d->ci, c_off == 0, up == 0.
static int conv2d_wgrad_impl(const udaseg_conv_desc* d, const void* x, int src_c, int c_off, int up, const void* dy, float* dw,
                             int accumulate, void* stream, int bf16, const float* in_scale = nullptr, const float* in_shift = nullptr,
                             int in_act = UDASEG_ACT_NONE, float in_slope = 0.f) {
  UDASEG_CHECK_ARG(d && x && dy && dw, "conv2d_wgrad: NULL pointer");
  const int g = bf16 ? 8 : 4;
  UDASEG_CHECK_ARG(d->ci % g == 0 && d->co % g == 0 && d->ci > 0 && d->co > 0, "conv2d_wgrad: channels must be multiples of %d", g);
  UDASEG_CHECK_ARG(d->kh > 0 && d->kw > 0 && d->stride >= 1 && d->pad >= 0, "conv2d_wgrad: bad kernel/stride");
  // found by the sanitizer build (make asan): an empty batch / extent reached the split-K planner and divided by zero
  UDASEG_CHECK_ARG(d->n > 0 && d->hi > 0 && d->wi > 0 && d->ho > 0 && d->wo > 0, "conv2d_wgrad: non-positive extent");
  UDASEG_CHECK_ARG(d->ho == (d->hi + 2 * d->pad - d->kh) / d->stride + 1 && d->wo == (d->wi + 2 * d->pad - d->kw) / d->stride + 1,
                   "conv2d_wgrad: output extent %dx%d inconsistent with input %dx%d k%d s%d p%d", d->ho, d->wo, d->hi, d->wi,
                   d->kh, d->stride, d->pad);
  UDASEG_CHECK_ARG((long long)d->n * d->hi * d->wi * d->ci < (1LL << 31) && (long long)d->n * d->ho * d->wo * d->co < (1LL << 31),
                   "conv2d_wgrad: tensor exceeds 2^31 elements");
  UDASEG_CHECK_ARG(src_c > 0 && src_c % g == 0 && c_off >= 0 && c_off % g == 0 && c_off + src_c <= d->ci,
                   "conv2d_wgrad_part: channel slice [%d, %d) of %d", c_off, c_off + src_c, d->ci);
  UDASEG_CHECK_ARG(!up || (d->stride == 1 && d->hi % 2 == 0 && d->wi % 2 == 0), "conv2d_wgrad_part: an up-sampled source needs stride 1 and even extents");
  const bool whole = src_c == d->ci;
  UDASEG_CHECK_ARG(whole || accumulate, "conv2d_wgrad_part: a channel slice of dW must be accumulated onto a caller-zeroed gradient");
  WgradArgs a = {};
  a.x = x; a.dy = dy; a.dw = dw;
  a.hi = d->hi; a.wi = d->wi; a.ci = src_c; a.ho = d->ho; a.wo = d->wo; a.co = d->co;
  a.kw = d->kw; a.stride = d->stride; a.pad = d->pad;
  a.M = d->n * d->ho * d->wo;
  a.J = d->kh * d->kw * src_c;
  a.ci_full = d->ci; a.c_off = c_off; a.J_ld = d->kh * d->kw * d->ci; a.up = up;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act; a.in_slope = in_slope;
  a.inv_ci = 1.0f / src_c; a.inv_kw = 1.0f / d->kw; a.inv_wo = 1.0f / d->wo; a.inv_ho = 1.0f / d->ho;
  if (d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && !up) {
    // a 1x1 / stride 1 convolution gathers pixel m from pixel m: the batch is ONE image row of M pixels as far as the gather is
    // concerned, so the row-uniform loop applies whatever the image width (r50's 48- and 24-pixel-wide stages took the generic
    // loop -- two divisions per gathered row -- only because their width is not a multiple of the 32-pixel load pass)
    a.hi = 1; a.wi = a.M; a.ho = 1; a.wo = a.M;
    a.inv_wo = 1.0f / (float)a.M; a.inv_ho = 1.0f;
  }
  hipStream_t st = as_stream(stream);
  prof_begin(1, st);
  int rc;
  udaseg_conv_desc dp = *d;     // FLOPs of this launch: the slice's share
  dp.ci = src_c;
  if (bf16)   // co <= 32: the 32 x 128 tile, as in fp32 (three J-tiles instead of five re-reading dy, no half-empty co tile)
    // >= 128 produced channels: 128 x 64 tiles (a wave 64 x 32: two MFMAs per 16-pixel step and fragment pair instead of one, half
    // the gathered bytes per FLOP).  Same box, single-stream ms per step, 128 x 64 / 64 x 64: cfg 5 (r50's 1x1 and strided layers)
    // 1.94 / 2.17, cfg 3 (the discriminator's 4x4 / stride 2 layers) 1.06 / 1.24 -- cfg 3 1028 against 982 images/s
    // (profiles/r04_wgrad_tile.txt)
    rc = d->co >= 128 ? launch_wgrad<128, 64, 2, 2>(a, accumulate, st, true)
       : d->co > 32   ? launch_wgrad<64, 64, 2, 2>(a, accumulate, st, true)
                      : launch_wgrad<32, 128, 1, 4>(a, accumulate, st, true);
  else if (whole && d->kh == d->kw &&
           small_wgrad_applicable(d->kh, d->stride, d->pad, d->ci, d->co, d->n * cdiv(d->hi, 16) * cdiv(d->wi, 16)))
    rc = launch_small_wgrad(static_cast<const float*>(x), static_cast<const float*>(dy), dw, d->n, d->hi, d->wi, d->ci, d->co,
                            accumulate, st, up);
  // (fp32: the 128 x 64 tile buys nothing -- 0.449 against 0.433 ms per cfg 2 step for the seven layers left here)
  // X3: per call, us, fp32 pipe / X3 (bench.py --layer-table, cfg 2, profiles/r04_igemm_x3.txt): 3x3 / stride 2 74 / 69, 69 / 58, 66 / 72;
  // 1x1 / stride 2 unchanged; the 7x7 stem (4 padded channels: two gathers per 8-wide piece, generic loop) 174 / 215 -- it stays on the
  // fp32 pipe.  What the split buys here is mostly the error (sign-alternated accumulators: l2 2.9e-7 against 3.5e-7 of the fp32 pipe)
  else if (d->co > 32 && src_c % 8 == 0 && d->co % 8 == 0 && wgrad_x3_on()) rc = launch_wgrad_x3<64, 64, 2, 2>(a, accumulate, st);
  else if (d->co > 32) rc = launch_wgrad<64, 64, 2, 2>(a, accumulate, st);
  else rc = launch_wgrad<32, 128, 1, 4>(a, accumulate, st);
  prof_end(1, st, udaseg_conv_flops(&dp), 2, &dp);
  return rc;
}

extern "C" int udaseg_conv2d_wgrad(const udaseg_conv_desc* d, const float* x, const float* dy, float* dw,
                                   int accumulate, void* stream) {
  UDASEG_CHECK_ARG(d != nullptr, "conv2d_wgrad: NULL desc");
  return conv2d_wgrad_impl(d, x, d->ci, 0, 0, dy, dw, accumulate, stream, 0);
}

extern "C" int udaseg_conv2d_wgrad_bf16(const udaseg_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate,
                                        void* stream) {
  UDASEG_CHECK_ARG(d != nullptr, "conv2d_wgrad_bf16: NULL desc");
  return conv2d_wgrad_impl(d, x, d->ci, 0, 0, dy, dw, accumulate, stream, 1);
}

extern "C" int udaseg_conv2d_wgrad_bnin_bf16(const udaseg_conv_desc* d, const void* x, const float* in_scale, const float* in_shift,
                                             int in_act, float in_slope, const void* dy, float* dw, int accumulate, void* stream) {
  UDASEG_CHECK_ARG(d != nullptr && in_scale != nullptr && in_shift != nullptr, "conv2d_wgrad_bnin_bf16: NULL desc / scale / shift");
  return conv2d_wgrad_impl(d, x, d->ci, 0, 0, dy, dw, accumulate, stream, 1, in_scale, in_shift, in_act, in_slope);
}

// halo-resident weight gradients (conv_wgrad_halo2.hip)
namespace udaseg {
bool wgrad_h2_applicable(const udaseg_conv_desc* d, int up_ca, bool f32);
int launch_wgrad_h2(const udaseg_conv_desc* d, const void* x, const void* x2, int up_ca, const void* dy, float* dw, bool f32, hipStream_t s,
                    const float* in_scale = nullptr, const float* in_shift = nullptr, int in_act = 0, float in_slope = 0.f,
                    int ldw = 0, int dw_coff = 0);
static bool wgrad_halo_off(bool f32) {
  // UDASEG_OPT_NO_WGRAD_HALO = 1: the per-tap split-K kernels everywhere; UDASEG_OPT_F32_HALO = 0: no halo-resident fp32 split
  const bool off = opt_get(UDASEG_OPT_NO_WGRAD_HALO) != 0;
  return f32 ? (off || !f32_halo_enabled()) : off;
}
}  // namespace udaseg

// fp32, the gathered operand an unwritten BatchNorm activation (engine.LazyAct on fp32, round 4): the small-channel direct kernel
// (<= 32 channels, also behind a nearest x2 up-sampling: up) or the halo-resident split kernel (plain source) apply the transform
// while they stage x
extern "C" int udaseg_conv2d_wgrad_bnin_ok(const udaseg_conv_desc* d, int up) {
  if (!d || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->n <= 0 || d->hi <= 0 || d->wi <= 0 || d->ci % 8 != 0 ||
      d->co % 4 != 0)
    return 0;
  if (small_wgrad_applicable(d->kh, d->stride, d->pad, d->ci, d->co, d->n * cdiv(d->hi, 16) * cdiv(d->wi, 16))) return 1;
  return !up && !wgrad_halo_off(true) && wgrad_h2_applicable(d, 0, true) ? 1 : 0;
}

extern "C" int udaseg_conv2d_wgrad_bnin(const udaseg_conv_desc* d, const float* x, int up, const float* in_scale, const float* in_shift,
                                        int in_act, float in_slope, const float* dy, float* dw, int accumulate, void* stream) {
  UDASEG_CHECK_ARG(d && x && in_scale && in_shift && dy && dw, "conv2d_wgrad_bnin: NULL pointer");
  UDASEG_CHECK_ARG(in_act == UDASEG_ACT_NONE || in_act == UDASEG_ACT_LEAKY, "conv2d_wgrad_bnin: unknown activation %d", in_act);
  UDASEG_CHECK_ARG(d->ho == d->hi && d->wo == d->wi, "conv2d_wgrad_bnin: stride-1 'same' convolutions only");
  UDASEG_CHECK_ARG(!up || (d->hi % 2 == 0 && d->wi % 2 == 0), "conv2d_wgrad_bnin: an up-sampled source needs even extents");
  if (!udaseg_conv2d_wgrad_bnin_ok(d, up)) {
    set_error("conv2d_wgrad_bnin: geometry not supported (ask udaseg_conv2d_wgrad_bnin_ok)");
    return UDASEG_E_UNSUPPORTED;
  }
  hipStream_t st = as_stream(stream);
  const bool small = small_wgrad_applicable(d->kh, d->stride, d->pad, d->ci, d->co, d->n * cdiv(d->hi, 16) * cdiv(d->wi, 16));
  if (!small && !accumulate) {          // before prof_begin: an early return must not leave the family's timing record open
    hipError_t e = hipMemsetAsync(dw, 0, (size_t)d->co * 9 * d->ci * sizeof(float), st);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(dw)");
  }
  prof_begin(1, st);
  int rc;
  if (small) {
    rc = launch_small_wgrad(x, dy, dw, d->n, d->hi, d->wi, d->ci, d->co, accumulate, st, up, in_scale, in_shift, in_act, in_slope);
  } else {
    rc = launch_wgrad_h2(d, x, nullptr, 0, dy, dw, true, st, in_scale, in_shift, in_act, in_slope);
  }
  prof_end(1, st, udaseg_conv_flops(d), 2, d);
  return rc;
}

extern "C" int udaseg_conv2d_wgrad_halo_bf16_ok(const udaseg_conv_desc* d, int up_ca) {
  return d != nullptr && !wgrad_halo_off(false) && wgrad_h2_applicable(d, up_ca, false) ? 1 : 0;
}

extern "C" int udaseg_conv2d_wgrad_halo_bf16(const udaseg_conv_desc* d, const void* x, const void* skip, int up_ca, const void* dy,
                                             float* dw, void* stream) {
  UDASEG_CHECK_ARG(d && x && dy && dw, "conv2d_wgrad_halo_bf16: NULL pointer");
  UDASEG_CHECK_ARG(d->n > 0 && d->hi > 0 && d->wi > 0 && d->ho == d->hi && d->wo == d->wi, "conv2d_wgrad_halo_bf16: bad extents");
  UDASEG_CHECK_ARG(up_ca == 0 ? skip == nullptr : skip != nullptr, "conv2d_wgrad_halo_bf16: up_ca and skip come together");
  if (!udaseg_conv2d_wgrad_halo_bf16_ok(d, up_ca)) {
    set_error("conv2d_wgrad_halo_bf16: geometry not supported (ask udaseg_conv2d_wgrad_halo_bf16_ok first)");
    return UDASEG_E_UNSUPPORTED;
  }
  hipStream_t st = as_stream(stream);
  prof_begin(1, st);
  const int rc = launch_wgrad_h2(d, x, skip, up_ca, dy, dw, false, st);
  prof_end(1, st, udaseg_conv_flops(d), 2, d);
  return rc;
}

extern "C" int udaseg_conv2d_wgrad_halo_f32x3_ok(const udaseg_conv_desc* d, int up_ca) {
  return d != nullptr && !wgrad_halo_off(true) && wgrad_h2_applicable(d, up_ca, true) ? 1 : 0;
}

extern "C" int udaseg_conv2d_wgrad_halo_f32x3(const udaseg_conv_desc* d, const float* x, const float* skip, int up_ca, const float* dy,
                                              float* dw, void* stream) {
  UDASEG_CHECK_ARG(d && x && dy && dw, "conv2d_wgrad_halo_f32x3: NULL pointer");
  UDASEG_CHECK_ARG(d->n > 0 && d->hi > 0 && d->wi > 0 && d->ho == d->hi && d->wo == d->wi, "conv2d_wgrad_halo_f32x3: bad extents");
  UDASEG_CHECK_ARG((up_ca > 0) == (skip != nullptr), "conv2d_wgrad_halo_f32x3: up_ca=%d, skip %s", up_ca, skip ? "given" : "NULL");
  if (!udaseg_conv2d_wgrad_halo_f32x3_ok(d, up_ca)) {
    set_error("conv2d_wgrad_halo_f32x3: geometry not supported (ask udaseg_conv2d_wgrad_halo_f32x3_ok first)");
    return UDASEG_E_UNSUPPORTED;
  }
  hipStream_t st = as_stream(stream);
  prof_begin(1, st);
  const int rc = launch_wgrad_h2(d, x, skip, up_ca, dy, dw, true, st);
  prof_end(1, st, udaseg_conv_flops(d), 2, d);
  return rc;
}

// The same kernel filling a channel SLICE of a wider layer's gradient: d describes the slice (ci = the slice's channels = x's
// channels), dW rows are ldw channels long and the slice starts at c_off.  The skip half of a decoder conv1 whose up-sampled half
// runs in the phase form (udaseg_conv2d_wgrad_up_f32x3).
extern "C" int udaseg_conv2d_wgrad_halo_slice_f32x3(const udaseg_conv_desc* d, const float* x, const float* dy, float* dw, int ldw,
                                                    int c_off, void* stream) {
  UDASEG_CHECK_ARG(d && x && dy && dw, "conv2d_wgrad_halo_slice_f32x3: NULL pointer");
  UDASEG_CHECK_ARG(d->n > 0 && d->hi > 0 && d->wi > 0 && d->ho == d->hi && d->wo == d->wi, "conv2d_wgrad_halo_slice_f32x3: bad extents");
  UDASEG_CHECK_ARG(ldw >= d->ci && c_off >= 0 && c_off + d->ci <= ldw && c_off % 4 == 0 && ldw % 4 == 0,
                   "conv2d_wgrad_halo_slice_f32x3: slice [%d, %d) of rows of %d channels", c_off, c_off + d->ci, ldw);
  if (!udaseg_conv2d_wgrad_halo_f32x3_ok(d, 0)) {
    set_error("conv2d_wgrad_halo_slice_f32x3: geometry not supported (ask udaseg_conv2d_wgrad_halo_f32x3_ok on the slice's descriptor)");
    return UDASEG_E_UNSUPPORTED;
  }
  hipStream_t st = as_stream(stream);
  prof_begin(1, st);
  const int rc = launch_wgrad_h2(d, x, nullptr, 0, dy, dw, true, st, nullptr, nullptr, 0, 0.f, ldw, c_off);
  prof_end(1, st, udaseg_conv_flops(d), 2, d);
  return rc;
}

extern "C" int udaseg_conv2d_wgrad_part(const udaseg_conv_desc* d, const float* src, int src_c, int c_off, int up,
                                        const float* dy, float* dw, int accumulate, void* stream) {
  UDASEG_CHECK_ARG(d != nullptr, "conv2d_wgrad_part: NULL desc");
  return conv2d_wgrad_impl(d, src, src_c, c_off, up, dy, dw, accumulate, stream, 0);
}

extern "C" int udaseg_conv2d_wgrad_part_bf16(const udaseg_conv_desc* d, const void* src, int src_c, int c_off, int up,
                                             const void* dy, float* dw, int accumulate, void* stream) {
  UDASEG_CHECK_ARG(d != nullptr, "conv2d_wgrad_part_bf16: NULL desc");
  return conv2d_wgrad_impl(d, src, src_c, c_off, up, dy, dw, accumulate, stream, 1);
}

// ---- weight repack for dgrad: w[co][t][ci] -> w_t[ci][t][co] ------------------------------------------------
namespace udaseg {
__global__ void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wt, int co, int T, int ci) {
  const long long total = (long long)co * T * ci;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    // i indexes the destination [ci][T][co]
    const int o = (int)(i % co);
    const long long r = i / co;
    const int t = (int)(r % T);
    const int c = (int)(r / T);
    wt[i] = w[((long long)o * T + t) * ci + c];
  }
}
}  // namespace udaseg

// All convolutions of a network in ONE launch: table[i] = {src offset, dst offset, co, taps, ci} (float offsets into the
// parameter arena / the packed-weight scratch), blockIdx.y = table row.
namespace udaseg {
__global__ void pack_dgrad_batched_kernel(const float* __restrict__ arena, float* __restrict__ packed,
                                          const int* __restrict__ table) {
  const int* e = table + 5 * blockIdx.y;
  const float* w = arena + e[0];
  float* wt = packed + e[1];
  if ((e[2] | e[4]) & 3) {       // channel counts that are not multiples of 4 (not produced by the arena layout): element-wise
    const int co = e[2], T = e[3], ci = e[4];
    const int total = co * T * ci;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
      const int o = i % co;
      const int r = i / co;
      const int t = r % T;
      const int c = r / T;
      wt[i] = w[(o * T + t) * ci + c];
    }
    return;
  }
  pack_dgrad_tiles<float>(w, wt, e[2], e[3], e[4]);
}
}  // namespace udaseg

extern "C" int udaseg_pack_dgrad_batched(const float* arena, float* packed, const int* table, int entries, void* stream) {
  UDASEG_CHECK_ARG(arena && packed && table && entries > 0, "pack_dgrad_batched: bad arguments");
  hipLaunchKernelGGL(pack_dgrad_batched_kernel, dim3(PACK_DGRAD_GRID_X, entries), dim3(256), 0, as_stream(stream), arena, packed, table);
  UDASEG_LAUNCH_CHECK("pack_dgrad_batched launch");
  return UDASEG_OK;
}

extern "C" int udaseg_pack_dgrad_weights(const udaseg_conv_desc* d, const float* w, float* w_t, void* stream) {
  UDASEG_CHECK_ARG(d && w && w_t, "pack_dgrad_weights: NULL pointer");
  const long long total = (long long)d->co * d->kh * d->kw * d->ci;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, w_t, d->co, d->kh * d->kw, d->ci);
  UDASEG_LAUNCH_CHECK("pack_dgrad launch");
  return UDASEG_OK;
}
