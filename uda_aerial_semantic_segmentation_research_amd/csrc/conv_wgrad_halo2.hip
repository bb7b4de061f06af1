// Halo-resident weight gradient, second form (round 4): stride-1 3x3 / pad 1 layers, bf16 tensors or fp32 tensors with the exact
// three-term split of conv_halo_f32x3.hip (six v_mfma_f32_32x32x16_bf16 products per operand pair), dW in fp32.
//
//   dW[co][tap][ci] += sum over pixels p of dy[p][co] * x[p + tap][ci]        (reference: loss.backward(), src/models/train.py:343)
//
// What round 3's kernels (conv_wgrad_halo_bf16_kernel / conv_wgrad_halo_f32x3_kernel, removed from conv_wgrad.hip) left on the table, measured
// (profiles/r04_wgrad_v2.txt): (1) the 4-row tile's 144-byte LDS pitch put two of the four pixel rows of every transposed fragment
// read on the same banks -- rocprofv3 SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.447 on the dominant symbol; (2) every tap
// re-read its own x fragment: 36 LDS reads for 30 MFMAs per 16-pixel K step; (3) the two tap groups of a wave pair shared one
// accumulator array across a wave-uniform branch inside the K loop, which the compiler reconciled with 40 v_mov_b64 + s_nop 10 per
// unrolled iteration.  Here:
//   * LDS holds 32-channel SUB-planes with 64-byte pixel rows: the four pixel rows x 64 bytes one half-wave of a transposed read
//     touches are 256 CONTIGUOUS bytes = every bank once, for any pixel alignment (the tap shift), no padding, no swizzle; sub-plane
//     strides are 64 (mod 128) bytes so that the 16-byte staging stores of a pixel's two channel halves do not collide either;
//   * a wave owns one 32 x 32 (co, ci) quadrant and the taps of ONE OR TWO kernel columns (waves 0-3: dx 0 and 1, waves 4-7: dx 2
//     and all of the staging): the x fragment of halo row r, column shift dx feeds the three taps (dy, dx) of the output rows
//     r, r - 1, r - 2 -- whose dy fragments wait in a three-row register ring -- so a 16-pixel step costs 6 + 6 reads for up to
//     18 / 36 MFMAs (bf16: 2 + 2 for 3 / 6) instead of 6 per tap;
//   * each role runs its own copy of the tile loop (no join inside it): accumulators stay where they are;
//   * blocks smaller than 64 x 64 channels (32 produced channels: the decoder's fourth block; reference smp.Unet decoder_channels
//     (256, 128, 64, 32, 16)) give the spare waves other K steps of the same quadrant (the 16-pixel halves of a row, then row
//     ranges), each wave adding its partial tile with its own atomics;
//   * 16-pixel-wide images (r18 layer4 at 512^2) take 8 x 16 pixel tiles (one K step per row) -- round 3 left them on the fp32
//     split-K kernel.
// Blocks are few and long-lived as before (one set of fp32 atomics per wave per launch, ~120 blocks beside the main stream's chain).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "halo_common.h"

namespace udaseg {

typedef short h2_s16x4 __attribute__((ext_vector_type(4)));
typedef short h2_s16x8 __attribute__((ext_vector_type(8)));

struct WgradH2Args {
  const void* x;      // [n][h][w][cx], or the half-resolution a [n][h/2][w/2][up_ca] of a fused decoder input
  const void* x2;     // fused decoder input: the skip tensor [n][h][w][ci - up_ca]
  const void* dy;     // [n][h][w][co]
  float* dw;          // [co][9][ldw] fp32, accumulated onto: channels [dw_coff, dw_coff + ci) of every row (ldw = ci, dw_coff = 0: the whole row)
  int ldw, dw_coff;
  int n, h, w, ci, co, up_ca;
  int ntx, nty, ntiles, ncib, pairs, P;
  unsigned x_bytes, x2_bytes, dy_bytes;
  // fp32, single plain source: x is the raw output of a conv + BatchNorm + activation layer whose activation was never written
  // (engine.LazyAct); the staging applies act(fma(x, in_scale[c], in_shift[c])) -- bn_apply's arithmetic -- to the pieces inside
  // the image before the split.  null: x is used as it is.
  const float* in_scale;
  const float* in_shift;
  int in_act;
  float in_slope;
  unsigned long long* timeline;   // diagnosis (udaseg_debug_set_timeline, TL instantiation only): per (block, role) 8 x u64
};
extern unsigned long long* g_timeline;      // conv_igemm.hip
extern int g_timeline_blocks;

constexpr int h2_pad(int bytes) { return (bytes / 64) % 2 == 0 ? bytes + 64 : bytes; }     // sub-plane stride: 64 (mod 128)

template <int PL_, int COQ_, int CIQ_, int TR_, int TWK_, int NSTW_, bool DB_ = false>
struct H2 {
  static constexpr int PL = PL_, COQ = COQ_, CIQ = CIQ_, TR = TR_, TWK = TWK_, NSTW = NSTW_;
  static constexpr bool DB = DB_;                                  // two LDS buffers: tile t + 1 is staged during the MFMAs of tile t
  static constexpr int NT = 512;
  static constexpr int COB = 32 * COQ, CIB = 32 * CIQ, TW = 16 * TWK;
  static constexpr int HR = TR + 2, HWD = TW + 2, HP = HR * HWD, TP = TR * TW;
  static constexpr int NQ = COQ * CIQ, KS = 4 / NQ;                // waves per tap group that share a quadrant (K split)
  static constexpr int KSH = KS < TWK ? KS : TWK;                  // ... over the 16-pixel halves of a row
  static constexpr int KSR = KS / KSH;                             // ... and over row ranges
  static constexpr int RW = TR / KSR, NHF = TWK / KSH;
  static constexpr int XSP = h2_pad(HP * 64), DSP = h2_pad(TP * 64);          // bytes per 32-channel sub-plane
  static constexpr int XPLANE = CIQ * XSP, DPLANE = COQ * DSP;
  static constexpr int BUF = PL * (XPLANE + DPLANE);               // one buffer: the split planes of an x halo and a dy tile
  static constexpr int LDS = (DB ? 2 : 1) * BUF;
  static constexpr int ES = PL == 3 ? 4 : 2;                       // bytes per element in HBM
  static constexpr int NST = 64 * NSTW;                            // staging threads: the LAST NSTW waves
  static constexpr int XO = CIB / 8, DO = COB / 8;                 // 8-channel pieces per pixel
  static constexpr int XPC = HP * XO, DPC = TP * DO;
  static constexpr int NX = (XPC + NST - 1) / NST, ND = (DPC + NST - 1) / NST;
  static constexpr int LPP = PL == 3 ? 2 : 1;                      // 16-byte loads per piece
  static_assert(NQ == 1 || NQ == 2 || NQ == 4, "1, 2 or 4 quadrants");
  static_assert(TR % KSR == 0 && TWK % KSH == 0 && KSH * KSR == KS, "K split tiles the tile");
  static_assert(NSTW == 4 || NSTW == 8, "staging by waves 4-7 or by all");
  static_assert(LDS <= 160 * 1024, "LDS");
};

__device__ __forceinline__ bf16x8 h2_tr_fragment(const char* p) {
  // this lane's address in the first 4-pixel block of a 16-pixel K step; the second block is 4 pixel rows (256 bytes) further
  const h2_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((h2_s16x4 __attribute__((address_space(3)))*)(p));
  const h2_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((h2_s16x4 __attribute__((address_space(3)))*)(p + 256));
  const h2_s16x8 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, f);
}

// One role of a block: the tile loop of the waves that own kernel columns DX0 .. DX0 + NDX - 1; STG: these waves also stage.
template <typename C, int DX0, int NDX, bool STG, bool TL, bool XFT = false>
__device__ __forceinline__ void wgrad_h2_role(const WgradH2Args& a, char* smem, int tid, int lane, int wave) {
  constexpr int PL = C::PL;
  char* const Xs = smem;
  char* const Ds = smem + PL * C::XPLANE;
  const int lr = lane & 31, lh = lane >> 5;
  const int grp = lane >> 4, cb = 16 * (grp & 1), hk = grp >> 1, tq = (lane & 15) >> 2, tp = lane & 3;
  const int q = wave & 3;
  const int qi = q % C::NQ, ks = q / C::NQ;
  const int coq = qi / C::CIQ, ciq = qi % C::CIQ;
  const int ksh = ks % C::KSH, r0 = (ks / C::KSH) * C::RW;
  const int pair = (int)blockIdx.x % a.pairs, split = (int)blockIdx.x / a.pairs;
  const int cob = pair / a.ncib, cib = pair % a.ncib;
  const int H = a.h, W = a.w;

  // ---- staging (register-staged: the next tile's loads are in flight during the MFMA phase)
  const bool UPC = a.up_ca > 0;
  const bool second = UPC && cib * C::CIB >= a.up_ca;
  const int cx = UPC ? (second ? a.ci - a.up_ca : a.up_ca) : a.ci;          // channels of the tensor read
  const int c0 = second ? cib * C::CIB - a.up_ca : cib * C::CIB;
  const bool half_res = UPC && !second;
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(second ? a.x2 : a.x), 0,
                                                                  (int)(second ? a.x2_bytes : a.x_bytes), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);
  const int st = tid - (C::NT - C::NST);
  u32x4 sx[STG ? C::NX : 1][C::LPP], sd[STG ? C::ND : 1][C::LPP];
  // in-staging transform of x (WgradH2Args::in_scale): a staging thread's 8-channel piece is the same for every piece it owns
  // (NST is a multiple of the pieces per pixel), so its coefficients live in registers; x_in: bit i = piece i is inside the image
  constexpr bool XF = XFT && PL == 3 && STG;          // its own instantiation: the plain launches carry none of this
  static_assert(C::NST % C::XO == 0, "one channel octet per staging thread");
  unsigned x_in = 0;
  f32x4 xsc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, xsh[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if constexpr (XF) {
    const int ch = c0 + (st % C::XO) * 8;
    xsc[0] = *reinterpret_cast<const f32x4*>(a.in_scale + ch);
    xsc[1] = *reinterpret_cast<const f32x4*>(a.in_scale + ch + 4);
    xsh[0] = *reinterpret_cast<const f32x4*>(a.in_shift + ch);
    xsh[1] = *reinterpret_cast<const f32x4*>(a.in_shift + ch + 4);
  }
  auto load_tile = [&](int tile) {
    if constexpr (STG) {
      x_in = 0;
      const int tx = tile % a.ntx;
      const int t2 = tile / a.ntx;
      const int ty = t2 % a.nty, img = t2 / a.nty;
      const int y0 = ty * C::TR, x0 = tx * C::TW;
#pragma unroll
      for (int i = 0; i < C::NX; ++i) {
        const int pc = st + i * C::NST;
        const int pix = pc / C::XO, oct = pc % C::XO;
        const int hy = pix / C::HWD, hx = pix - hy * C::HWD;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const bool ok = pc < C::XPC && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const int p = half_res ? (img * (H >> 1) + (iy >> 1)) * (W >> 1) + (ix >> 1) : (img * H + iy) * W + ix;
        const unsigned off = ok ? (unsigned)((p * cx + c0 + oct * 8) * C::ES) : 0x80000000u;
        if constexpr (XF) x_in |= (ok ? 1u : 0u) << i;
#pragma unroll
        for (int l = 0; l < C::LPP; ++l) sx[i][l] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 16 * l, 0);
      }
#pragma unroll
      for (int i = 0; i < C::ND; ++i) {
        const int pc = st + i * C::NST;
        const int pix = pc / C::DO, oct = pc % C::DO;
        const int oy = y0 + pix / C::TW, ox = x0 + pix % C::TW;
        const bool ok = pc < C::DPC && oy < H && ox < W;
        const unsigned off = ok ? (unsigned)((((img * H + oy) * W + ox) * a.co + cob * C::COB + oct * 8) * C::ES) : 0x80000000u;
#pragma unroll
        for (int l = 0; l < C::LPP; ++l) sd[i][l] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, (int)off, 16 * l, 0);
      }
    }
  };
  auto store_piece = [&](char* dst, int plane_bytes, const u32x4 (&v)[C::LPP], unsigned sgn = 0u) {
    if constexpr (PL == 3) {
      u32x4 p0, p1, p2;
      split3(v[0] ^ sgn, v[C::LPP - 1] ^ sgn, p0, p1, p2);      // sgn = 0x80000000: the terms of -v are minus the terms of v
      *reinterpret_cast<u32x4*>(dst) = p0;
      *reinterpret_cast<u32x4*>(dst + plane_bytes) = p1;
      *reinterpret_cast<u32x4*>(dst + 2 * plane_bytes) = p2;
    } else {
      *reinterpret_cast<u32x4*>(dst) = v[0];
    }
  };
  // fp32 (PL == 3): the tiles [q1, q3) of this block's sequence are staged as -dy and accumulated on a negated accumulator, + - - +
  // like the forward kernels' K loop (halo_common.h f3_negated_groups: the bf16 MFMA adder truncates toward minus infinity, and a
  // weight gradient sums ~1000 pixels per block on top of the atomics)
  const int nseq = split < a.ntiles ? (a.ntiles - split + a.P - 1) / a.P : 0;
  const int sq1 = PL == 3 ? (nseq + 2) / 4 : 0, sq3 = PL == 3 ? nseq - sq1 : 0;
  auto store_tile = [&](int bo, int seq) {          // bo: byte offset of the LDS buffer; seq: the staged tile's place in the sequence
    const unsigned dsgn = (seq >= sq1 && seq < sq3) ? 0x80000000u : 0u;
    if constexpr (STG) {
#pragma unroll
      for (int i = 0; i < C::NX; ++i) {
        const int pc = st + i * C::NST;
        const int pix = pc / C::XO, oct = pc % C::XO;
        if constexpr (XF) {
          {
            const bool in = (x_in >> i) & 1u;
#pragma unroll
            for (int l = 0; l < 2; ++l) {
              f32x4 v = __builtin_bit_cast(f32x4, sx[i][l]);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float t = act_apply(__builtin_fmaf(v[e], xsc[l][e], xsh[l][e]), a.in_act, a.in_slope);
                v[e] = in ? t : 0.f;
              }
              sx[i][l] = __builtin_bit_cast(u32x4, v);
            }
          }
        }
        if (i < C::NX - 1 || pc < C::XPC) store_piece(Xs + bo + (oct >> 2) * C::XSP + pix * 64 + (oct & 3) * 16, C::XPLANE, sx[i]);
      }
#pragma unroll
      for (int i = 0; i < C::ND; ++i) {
        const int pc = st + i * C::NST;
        const int pix = pc / C::DO, oct = pc % C::DO;
        if (i < C::ND - 1 || pc < C::DPC) store_piece(Ds + bo + (oct >> 2) * C::DSP + pix * 64 + (oct & 3) * 16, C::DPLANE, sd[i], dsgn);
      }
    }
  };

  f32x16 acc[3 * NDX];        // [dy][dxi]
#pragma unroll
  for (int t = 0; t < 3 * NDX; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

  // this lane's byte address inside a 16-pixel K step (transposed reads), at this wave's first row / first half
  const int lane_off = (8 * hk + tq) * 64 + (cb + 4 * tp) * 2;
  const char* const dbase0 = Ds + coq * C::DSP + (r0 * C::TW + 16 * ksh) * 64 + lane_off;
  const char* const xbase0 = Xs + ciq * C::XSP + (r0 * C::HWD + 16 * ksh + DX0) * 64 + lane_off;

  // TL: shader-clock cycles this wave spends per phase, summed over its tiles (s_memtime at the phase boundaries, where the
  // LDS counter is drained anyway), and the launch's wall-clock ticks for the clock the chip held
  unsigned long long c_store = 0, c_b1 = 0, c_mfma = 0, c_b2 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0, tw0 = 0, tc0 = 0;
  int ntl = 0;
  if constexpr (TL) {
    tw0 = __builtin_amdgcn_s_memrealtime();
    tc0 = __builtin_amdgcn_s_memtime();
  }
  int cur = 0;                     // DB: byte offset of the buffer the MFMAs of this iteration read
  if (split < a.ntiles) load_tile(split);
  if constexpr (C::DB) {
    // two buffers: the staging waves split tile t + 1 into the OTHER buffer while every wave's MFMAs read tile t (their vector
    // instructions issue beside the older waves' MFMAs), and request tile t + 2; ONE barrier per tile
    store_tile(0, 0);
    if (split + a.P < a.ntiles) load_tile(split + a.P);
    __syncthreads();
  }
  int seq = 0;
  bool negated = false;
  for (int tile = split; tile < a.ntiles; tile += a.P, ++seq) {
    if constexpr (TL) t0 = __builtin_amdgcn_s_memtime();
    if constexpr (PL == 3) {
      if ((seq >= sq1 && seq < sq3) != negated) {        // uniform; twice per block
        negated = !negated;
#pragma unroll
        for (int t = 0; t < 3 * NDX; ++t) acc[t] = -acc[t];
      }
    }
    if constexpr (C::DB) {
      if (tile + a.P < a.ntiles) {
        store_tile(cur ^ C::BUF, seq + 1);
        if (tile + 2 * a.P < a.ntiles) load_tile(tile + 2 * a.P);
      }
      if constexpr (TL) t1 = t2 = __builtin_amdgcn_s_memtime();
    } else {
      store_tile(0, seq);
      if constexpr (TL) t1 = __builtin_amdgcn_s_memtime();
      __syncthreads();
      if constexpr (TL) t2 = __builtin_amdgcn_s_memtime();
      if (tile + a.P < a.ntiles) load_tile(tile + a.P);
    }
    const char* const dbase = dbase0 + cur;
    const char* const xbase = xbase0 + cur;
    if constexpr (C::DB) cur ^= C::BUF;
#pragma unroll
    for (int hfi = 0; hfi < C::NHF; ++hfi) {
      bf16x8 A[3][PL];
#pragma unroll
      for (int rr = 0; rr < C::RW + 2; ++rr) {
        if (rr < C::RW) {
#pragma unroll
          for (int pl = 0; pl < PL; ++pl)
            A[rr % 3][pl] = h2_tr_fragment(dbase + pl * C::DPLANE + (rr * C::TW + 16 * C::KSH * hfi) * 64);
        }
        bf16x8 B[NDX][PL];
#pragma unroll
        for (int dxi = 0; dxi < NDX; ++dxi)
#pragma unroll
          for (int pl = 0; pl < PL; ++pl)
            B[dxi][pl] = h2_tr_fragment(xbase + pl * C::XPLANE + (rr * C::HWD + 16 * C::KSH * hfi + dxi) * 64);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int j = rr - dy;               // output row (relative to r0) whose tap (dy, dx) reads halo row rr
          if (j >= 0 && j < C::RW) {
#pragma unroll
            for (int dxi = 0; dxi < NDX; ++dxi) {
              if constexpr (PL == 3) {
                // smallest terms first: (dy piece i) x (x piece ij - i), i + j <= 2
#pragma unroll
                for (int ij = 2; ij >= 0; --ij)
#pragma unroll
                  for (int i = 0; i <= ij; ++i)
                    acc[dy * NDX + dxi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j % 3][i], B[dxi][ij - i], acc[dy * NDX + dxi], 0, 0, 0);
              } else {
                acc[dy * NDX + dxi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j % 3][0], B[dxi][0], acc[dy * NDX + dxi], 0, 0, 0);
              }
            }
          }
        }
      }
    }
    if constexpr (TL) t3 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if constexpr (TL) {
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      c_store += t1 - t0; c_b1 += t2 - t1; c_mfma += t3 - t2; c_b2 += t4 - t3;
      ++ntl;
    }
  }
  if constexpr (TL) {
    if (a.timeline != nullptr && lane == 0 && (wave & 3) == 0) {
      unsigned long long* t = a.timeline + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 8;
      t[0] = c_store; t[1] = c_b1; t[2] = c_mfma; t[3] = c_b2;
      t[4] = __builtin_amdgcn_s_memtime() - tc0; t[5] = __builtin_amdgcn_s_memrealtime() - tw0; t[6] = (unsigned long long)ntl;
      t[7] = 1;
    }
  }

  if (negated) {      // a sequence too short for the pattern to close (one or two tiles)
#pragma unroll
    for (int t = 0; t < 3 * NDX; ++t) acc[t] = -acc[t];
  }
  // dW[co][tap][ci] += acc: register v of a tap = one co row, 32 consecutive ci per half-wave (two 128-byte segments per
  // wave-instruction: the full-rate shape of the memory-side float atomics)
  const int ci_g = cib * C::CIB + ciq * 32 + lr;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dxi = 0; dxi < NDX; ++dxi)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int co_g = cob * C::COB + coq * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
        atomicAdd(a.dw + ((size_t)co_g * 9 + dy * 3 + DX0 + dxi) * a.ldw + a.dw_coff + ci_g, acc[dy * NDX + dxi][v]);
      }
}

template <int PL, int COQ, int CIQ, int TR, int TWK, int NSTW, bool DB, bool TL = false, bool XF = false>
__global__ __launch_bounds__(512, 2) void conv_wgrad_h2_kernel(const WgradH2Args a) {
  using C = H2<PL, COQ, CIQ, TR, TWK, NSTW, DB>;
  extern __shared__ __attribute__((aligned(16))) char h2smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4) wgrad_h2_role<C, 0, 2, NSTW == 8, TL, XF>(a, h2smem, tid, lane, wave);
  else wgrad_h2_role<C, 2, 1, true, TL, XF>(a, h2smem, tid, lane, wave);
}

// which configuration serves (d, up_ca): 0 = none
//   1: 64 x 64 channel blocks, 4 x 32 pixel tiles (f32x3) / 8 x 32 (bf16)      2: 64 x 64, 8 x 16 pixel tiles (16-pixel-wide images)
//   3: 32 produced x 64 gathered channels, 4 x 32 (f32x3) / 8 x 32 (bf16)      4: 32 x 32 channels
static int h2_config(const udaseg_conv_desc* d, int up_ca, bool f32) {
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1) return 0;
  if (d->n <= 0 || d->hi <= 0 || d->wi <= 0 || d->ho != d->hi || d->wo != d->wi || d->ci <= 0 || d->co <= 0) return 0;
  const long long px = (long long)d->n * d->hi * d->wi;
  const int es = f32 ? 4 : 2;
  if (px * d->ci * es >= (1LL << 31) || px * d->co * es >= (1LL << 31)) return 0;
  int cfg = 0;
  if (d->ci % 64 == 0 && d->co % 64 == 0) cfg = d->wi < 32 ? 2 : 1;
  else if (d->ci % 64 == 0 && d->co % 32 == 0) cfg = 3;
  else if (d->ci % 32 == 0 && d->co % 32 == 0) cfg = 4;
  if (cfg == 0 || d->wi < 16) return 0;
  if (cfg != 2 && d->wi < 32) return 0;
  if (up_ca > 0) {
    if (up_ca >= d->ci || d->hi % 2 != 0 || d->wi % 2 != 0) return 0;
    if (up_ca % 64 != 0) {            // the source boundary must fall between two gathered-channel blocks: 32-channel blocks
      if (up_ca % 32 != 0 || d->wi < 32) return 0;
      cfg = 4;
    }
  }
  return cfg;
}

bool wgrad_h2_applicable(const udaseg_conv_desc* d, int up_ca, bool f32) { return d != nullptr && h2_config(d, up_ca, f32) != 0; }

struct H2In {        // WgradH2Args::in_*
  const float* scale = nullptr;
  const float* shift = nullptr;
  int act = 0;
  float slope = 0.f;
  int ldw = 0, dw_coff = 0;      // > 0: dW rows are ldw channels long and this launch fills [dw_coff, dw_coff + ci) (a slice of a wider layer)
};

template <int PL, int COQ, int CIQ, int TR, int TWK, int NSTW, bool DB = false>
static int launch_h2_t(const udaseg_conv_desc* d, const void* x, const void* x2, int up_ca, const void* dy, float* dw, hipStream_t s,
                       int target, const H2In& in = H2In()) {
  using C = H2<PL, COQ, CIQ, TR, TWK, NSTW, DB>;
  WgradH2Args a = {};
  a.x = x; a.x2 = x2; a.dy = dy; a.dw = dw;
  a.ldw = in.ldw > 0 ? in.ldw : d->ci; a.dw_coff = in.ldw > 0 ? in.dw_coff : 0;
  a.in_scale = in.scale; a.in_shift = in.shift; a.in_act = in.act; a.in_slope = in.slope;
  a.n = d->n; a.h = d->hi; a.w = d->wi; a.ci = d->ci; a.co = d->co; a.up_ca = up_ca;
  a.ntx = cdiv(d->wi, C::TW); a.nty = cdiv(d->hi, C::TR); a.ntiles = d->n * a.ntx * a.nty;
  a.ncib = d->ci / C::CIB; a.pairs = a.ncib * (d->co / C::COB);
  int P = target / a.pairs;
  if (P < 1) P = 1;
  if (P > a.ntiles) P = a.ntiles;
  a.P = P;
  const long long px = (long long)d->n * d->hi * d->wi;
  a.x_bytes = (unsigned)(up_ca > 0 ? (long long)d->n * (d->hi / 2) * (d->wi / 2) * up_ca * C::ES : px * d->ci * C::ES);
  a.x2_bytes = (unsigned)(up_ca > 0 ? px * (d->ci - up_ca) * C::ES : 0);
  a.dy_bytes = (unsigned)(px * d->co * C::ES);
  auto kern = conv_wgrad_h2_kernel<PL, COQ, CIQ, TR, TWK, NSTW, DB>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv_wgrad_h2)");
    attr_done = true;
  }
  static std::atomic<int> kid{-1};
  if (kid < 0) {
    char nm[96];
    snprintf(nm, sizeof(nm), "conv_wgrad_h2_kernel<%d, %d, %d, %d, %d, %d, %s, false, false>", PL, COQ, CIQ, TR, TWK, NSTW, DB ? "true" : "false");
    kid = kprof_id(nm);
  }
  hipEvent_t ev = kprof_begin(s);
  if constexpr (PL == 3 && COQ == 2 && CIQ == 2 && TWK == 2) {      // the dominant configuration has a stamped twin (diagnosis only)
    // (never for a launch that transforms x while staging: the twin has no XF form and would multiply the RAW tensor)
    if (g_timeline != nullptr && a.in_scale == nullptr && (long long)a.pairs * P * 16 <= (long long)g_timeline_blocks * 6) {
      auto kern_tl = conv_wgrad_h2_kernel<PL, COQ, CIQ, TR, TWK, NSTW, DB, true>;
      static std::atomic<bool> tl_attr{false};
      if (!tl_attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern_tl), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv_wgrad_h2 timeline twin)");
        tl_attr = true;
      }
      a.timeline = g_timeline;
      hipLaunchKernelGGL(kern_tl, dim3((unsigned)(a.pairs * P)), dim3(C::NT), C::LDS, s, a);
      kprof_end(kid, ev, s, 2.0 * (double)px * d->co * 9.0 * d->ci);
      UDASEG_LAUNCH_CHECK("conv_wgrad_h2 (timeline) launch");
      return UDASEG_OK;
    }
  }
  if constexpr (PL == 3) {
    if (a.in_scale != nullptr) {          // the instantiation that transforms x while it stages (WgradH2Args::in_scale)
      auto kern_xf = conv_wgrad_h2_kernel<PL, COQ, CIQ, TR, TWK, NSTW, DB, false, true>;
      static std::atomic<bool> xf_attr{false};
      if (!xf_attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern_xf), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv_wgrad_h2, in-staging transform)");
        xf_attr = true;
      }
      static std::atomic<int> kid_xf{-1};
      if (kid_xf < 0) {
        char nm[112];
        snprintf(nm, sizeof(nm), "conv_wgrad_h2_kernel<%d, %d, %d, %d, %d, %d, %s, false, true>", PL, COQ, CIQ, TR, TWK, NSTW, DB ? "true" : "false");
        kid_xf = kprof_id(nm);
      }
      hipLaunchKernelGGL(kern_xf, dim3((unsigned)(a.pairs * P)), dim3(C::NT), C::LDS, s, a);
      kprof_end(kid_xf, ev, s, 2.0 * (double)px * d->co * 9.0 * d->ci);
      UDASEG_LAUNCH_CHECK("conv_wgrad_h2 (in-staging transform) launch");
      return UDASEG_OK;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.pairs * P)), dim3(C::NT), C::LDS, s, a);
  kprof_end(kid, ev, s, 2.0 * (double)px * d->co * 9.0 * d->ci);
  UDASEG_LAUNCH_CHECK("conv_wgrad_h2 launch");
  return UDASEG_OK;
}

int launch_wgrad_h2(const udaseg_conv_desc* d, const void* x, const void* x2, int up_ca, const void* dy, float* dw, bool f32,
                    hipStream_t s, const float* in_scale, const float* in_shift, int in_act, float in_slope, int ldw, int dw_coff) {
  H2In in;
  in.scale = in_scale; in.shift = in_shift; in.act = in_act; in.slope = in_slope;
  in.ldw = ldw; in.dw_coff = dw_coff;
  if (in_scale != nullptr && (!f32 || up_ca != 0 || in_shift == nullptr)) {
    set_error("conv2d_wgrad_halo: the in-staging transform needs fp32 operands and a single plain source");
    return UDASEG_E_UNSUPPORTED;
  }
  const int cfg = h2_config(d, up_ca, f32);
  if (cfg == 0) {
    set_error("conv2d_wgrad_halo: geometry not supported (ask the _ok query first)");
    return UDASEG_E_UNSUPPORTED;
  }
  // blocks per launch: few and long-lived (one set of atomics per wave; the kernel runs beside the main stream's chain).
  // UDASEG_WGRAD_F3_BLOCKS / UDASEG_WGRAD_HALO_BLOCKS: tuning aids
  int tf3 = opt_get(UDASEG_OPT_WGRAD_F3_BLOCKS), tbf = opt_get(UDASEG_OPT_WGRAD_HALO_BLOCKS);
  const int tdeep_o = opt_get(UDASEG_OPT_WGRAD_DEEP_BLOCKS);
  const int tdeep = tdeep_o >= 1 ? tdeep_o : 256;        // 16-pixel-wide images, bf16
  const int tdeep3 = tdeep_o >= 1 ? tdeep_o : 128;       // ... fp32 split: 64 / 96 / 128 / 160 / 256 / 384 blocks -> 979.3 / 980.6 / 987.3 /
                                                         // 985.5 / 983.5 images/s on one box, 942.1 (128) / 942.3 (192) / 937.1 / 936.9 on another
  if (tf3 < 1) tf3 = 120;
  if (tbf < 1) tbf = 96;
  const int db = opt_get(UDASEG_OPT_WGRAD_DB);      // 0: the single-buffer 4-row form of the 64 x 64 fp32 configuration (A/B)
  if (f32) {
    if (cfg == 1 && db) return launch_h2_t<3, 2, 2, 2, 2, 4, true>(d, x, x2, up_ca, dy, dw, s, tf3, in);
    if (cfg == 1) return launch_h2_t<3, 2, 2, 4, 2, 4>(d, x, x2, up_ca, dy, dw, s, tf3, in);
    if (cfg == 2) return launch_h2_t<3, 2, 2, 8, 1, 4>(d, x, x2, up_ca, dy, dw, s, tdeep3, in);
    if (cfg == 3) return launch_h2_t<3, 1, 2, 4, 2, 4>(d, x, x2, up_ca, dy, dw, s, tf3, in);
    return launch_h2_t<3, 1, 1, 4, 2, 4>(d, x, x2, up_ca, dy, dw, s, tf3, in);
  }
  if (cfg == 1) return launch_h2_t<1, 2, 2, 8, 2, 8>(d, x, x2, up_ca, dy, dw, s, tbf);
  if (cfg == 2) return launch_h2_t<1, 2, 2, 8, 1, 8>(d, x, x2, up_ca, dy, dw, s, tdeep);
  if (cfg == 3) return launch_h2_t<1, 1, 2, 8, 2, 8>(d, x, x2, up_ca, dy, dw, s, tbf);
  return launch_h2_t<1, 1, 1, 8, 2, 8>(d, x, x2, up_ca, dy, dw, s, tbf);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Round 5: the weight gradient of a decoder conv1's UP-SAMPLED half in the phase form of conv_up_f32x3.hip.
//   conv3x3(nearest_x2(a)) = four 2x2 phase convolutions of a with pre-summed weights W'_{py,px}[u][v], so
//   T_{py,px}[u][v][co][ci] = sum_{q,r} dy[2q+py, 2r+px][co] * a[q + py - 1 + u, r + px - 1 + v][ci]          (16 correlations at a's size)
//   dW[ky][kx] += T_{py,px}[u][v]  for every (py, u) with ky in Ky(py, u) and (px, v) with kx in Kx(px, v)      (each tap: 4 of the 16)
// -- 16 tap evaluations per pixel of `a` instead of 4 x 9 = 36 (reference: loss.backward(), src/models/train.py:343, through
// aten::upsample_nearest2d + aten::_convolution of smp's DecoderBlock).  The h2 scheme in a's coordinates: a block owns a
// (32 COQ) x (32 CIQ) block of dW and ONE column phase px, and walks tiles of TR rows x 16 TWK pixels of `a`; it stages the a-halo
// once per tile and the 2 TR rows x TW pixels of dy that are the phases (0, px), (1, px) of the tile (pixels two apart in memory:
// the same 16-byte pieces as any other gather); waves 0-3 own the halo column offset ex = px (v = 0), waves 4-7 ex = px + 1 (v = 1),
// each its (py, u) in {0,1}^2 accumulators of a 32 x 32 quadrant: an x fragment of halo row rr feeds the dy rows 2 (rr - py - u) + py.
// All eight waves stage (the MFMA work per staged byte is 4/9 of the nine-tap kernel's); two LDS buffers, one barrier per tile.
// The accumulators leave as fp32 atomics into the one, two or four taps their phase tap stands for.
struct WgradUpArgs {
  const float* x;     // a [n][h][w][ca]
  const float* dy;    // [n][2h][2w][co]
  float* dw;          // [co][9][ldw] fp32, accumulated onto (channels [0, ca) of the rows)
  int n, h, w, ca, co, ldw;
  int ntx, nty, ntiles, ncib, pairs, P;
  unsigned x_bytes, dy_bytes;
};

template <int COQ_, int CIQ_, int TR_, int TWK_>
struct H2Up {
  static constexpr int PL = 3, COQ = COQ_, CIQ = CIQ_, TR = TR_, TWK = TWK_;
  static constexpr int NT = 512;
  static constexpr int COB = 32 * COQ, CIB = 32 * CIQ, TW = 16 * TWK;
  static constexpr int HR = TR + 2, HWD = TW + 2, HP = HR * HWD, TP = 2 * TR * TW;       // dy: rows 2 q + py of the tile
  static constexpr int NQ = COQ * CIQ, KS = 4 / NQ;
  static constexpr int KSH = KS < TWK ? KS : TWK, KSR = KS / KSH;
  static constexpr int RW = TR / KSR, NHF = TWK / KSH;
  static constexpr int XSP = h2_pad(HP * 64), DSP = h2_pad(TP * 64);
  static constexpr int XPLANE = CIQ * XSP, DPLANE = COQ * DSP;
  static constexpr int BUF = PL * (XPLANE + DPLANE);
  static constexpr int LDS = 2 * BUF;
  static constexpr int XO = CIB / 8, DO = COB / 8;
  static constexpr int XPC = HP * XO, DPC = TP * DO;
  static constexpr int NX = (XPC + NT - 1) / NT, ND = (DPC + NT - 1) / NT;
  static_assert(NQ == 2 || NQ == 4, "2 or 4 quadrants");
  static_assert(TR % KSR == 0 && TWK % KSH == 0 && KSH * KSR == KS, "K split tiles the tile");
  static_assert(LDS <= 160 * 1024, "LDS");
};

template <typename C, int V>
__device__ __forceinline__ void wgrad_up_role(const WgradUpArgs& a, char* smem, int tid, int lane, int wave) {
  constexpr int PL = 3;
  char* const Xs = smem;
  char* const Ds = smem + PL * C::XPLANE;
  const int lr = lane & 31, lh = lane >> 5;
  const int grp = lane >> 4, cb = 16 * (grp & 1), hk = grp >> 1, tq = (lane & 15) >> 2, tp = lane & 3;
  const int q = wave & 3;
  const int qi = q % C::NQ, ks = q / C::NQ;
  const int coq = qi / C::CIQ, ciq = qi % C::CIQ;
  const int ksh = ks % C::KSH, r0 = (ks / C::KSH) * C::RW;
  int bidx = (int)blockIdx.x;
  const int px = bidx & 1;                 // this block's column phase
  bidx >>= 1;
  const int pair = bidx % a.pairs, split = bidx / a.pairs;
  const int cob = pair / a.ncib, cib = pair % a.ncib;
  const int H = a.h, W = a.w;
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);
  u32x4 sx[C::NX][2], sd[C::ND][2];
  auto load_tile = [&](int tile) {
    const int tx = tile % a.ntx;
    const int t2 = tile / a.ntx;
    const int ty = t2 % a.nty, img = t2 / a.nty;
    const int y0 = ty * C::TR, x0 = tx * C::TW;
#pragma unroll
    for (int i = 0; i < C::NX; ++i) {
      const int pc = tid + i * C::NT;
      const int pix = pc / C::XO, oct = pc % C::XO;
      const int hy = pix / C::HWD, hx = pix - hy * C::HWD;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      const bool ok = pc < C::XPC && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const unsigned off = ok ? (unsigned)((((img * H + iy) * W + ix) * a.ca + cib * C::CIB + oct * 8) * 4) : 0x80000000u;
#pragma unroll
      for (int l = 0; l < 2; ++l) sx[i][l] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 16 * l, 0);
    }
#pragma unroll
    for (int i = 0; i < C::ND; ++i) {
      const int pc = tid + i * C::NT;
      const int pix = pc / C::DO, oct = pc % C::DO;
      const int row2 = pix / C::TW, col = pix % C::TW;          // row2 = 2 (row of the tile) + py
      const int qy = y0 + (row2 >> 1), qx = x0 + col;
      const bool ok = pc < C::DPC && qy < H && qx < W;
      const unsigned off = ok ? (unsigned)((((img * 2 * H + 2 * qy + (row2 & 1)) * (2 * W) + 2 * qx + px) * a.co + cob * C::COB + oct * 8) * 4)
                              : 0x80000000u;
#pragma unroll
      for (int l = 0; l < 2; ++l) sd[i][l] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, (int)off, 16 * l, 0);
    }
  };
  auto store_piece = [&](char* dst, int plane_bytes, const u32x4 (&v)[2], unsigned sgn) {
    u32x4 p0, p1, p2;
    split3(v[0] ^ sgn, v[1] ^ sgn, p0, p1, p2);
    *reinterpret_cast<u32x4*>(dst) = p0;
    *reinterpret_cast<u32x4*>(dst + plane_bytes) = p1;
    *reinterpret_cast<u32x4*>(dst + 2 * plane_bytes) = p2;
  };
  const int nseq = split < a.ntiles ? (a.ntiles - split + a.P - 1) / a.P : 0;
  const int sq1 = (nseq + 2) / 4, sq3 = nseq - sq1;             // + - - + over the block's tile sequence (see wgrad_h2_role)
  auto store_tile = [&](int bo, int seq) {
    const unsigned dsgn = (seq >= sq1 && seq < sq3) ? 0x80000000u : 0u;
#pragma unroll
    for (int i = 0; i < C::NX; ++i) {
      const int pc = tid + i * C::NT;
      const int pix = pc / C::XO, oct = pc % C::XO;
      if (i < C::NX - 1 || pc < C::XPC) store_piece(Xs + bo + (oct >> 2) * C::XSP + pix * 64 + (oct & 3) * 16, C::XPLANE, sx[i], 0u);
    }
#pragma unroll
    for (int i = 0; i < C::ND; ++i) {
      const int pc = tid + i * C::NT;
      const int pix = pc / C::DO, oct = pc % C::DO;
      if (i < C::ND - 1 || pc < C::DPC) store_piece(Ds + bo + (oct >> 2) * C::DSP + pix * 64 + (oct & 3) * 16, C::DPLANE, sd[i], dsgn);
    }
  };

  f32x16 acc[4];        // [py * 2 + u]
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

  const int lane_off = (8 * hk + tq) * 64 + (cb + 4 * tp) * 2;
  const char* const dbase0 = Ds + coq * C::DSP + (2 * r0 * C::TW + 16 * ksh) * 64 + lane_off;
  const char* const xbase0 = Xs + ciq * C::XSP + (r0 * C::HWD + 16 * ksh + px + V) * 64 + lane_off;

  int cur = 0;
  if (split < a.ntiles) load_tile(split);
  store_tile(0, 0);
  if (split + a.P < a.ntiles) load_tile(split + a.P);
  __syncthreads();
  int seq = 0;
  bool negated = false;
  for (int tile = split; tile < a.ntiles; tile += a.P, ++seq) {
    if ((seq >= sq1 && seq < sq3) != negated) {
      negated = !negated;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = -acc[t];
    }
    if (tile + a.P < a.ntiles) {
      store_tile(cur ^ C::BUF, seq + 1);
      if (tile + 2 * a.P < a.ntiles) load_tile(tile + 2 * a.P);
    }
    const char* const dbase = dbase0 + cur;
    const char* const xbase = xbase0 + cur;
    cur ^= C::BUF;
#pragma unroll
    for (int hfi = 0; hfi < C::NHF; ++hfi) {
      // dy fragments: D0[j] = phase row py = 0 of tile row j (LDS row 2 j), D1[j] = py = 1 (LDS row 2 j + 1); at halo row rr the
      // taps (py, u) read D_py[rr - py - u]: D0[rr] and D1[rr - 1] are new, D0[rr - 1] and D1[rr - 2] come from the step before
      bf16x8 D0[2][PL], D1[2][PL];
#pragma unroll
      for (int rr = 0; rr < C::RW + 2; ++rr) {
        if (rr < C::RW) {
#pragma unroll
          for (int pl = 0; pl < PL; ++pl)
            D0[rr & 1][pl] = h2_tr_fragment(dbase + pl * C::DPLANE + ((2 * rr) * C::TW + 16 * C::KSH * hfi) * 64);
        }
        if (rr >= 1 && rr - 1 < C::RW) {
#pragma unroll
          for (int pl = 0; pl < PL; ++pl)
            D1[(rr - 1) & 1][pl] = h2_tr_fragment(dbase + pl * C::DPLANE + ((2 * (rr - 1) + 1) * C::TW + 16 * C::KSH * hfi) * 64);
        }
        bf16x8 B[PL];
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) B[pl] = h2_tr_fragment(xbase + pl * C::XPLANE + (rr * C::HWD + 16 * C::KSH * hfi) * 64);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int py = t >> 1, u = t & 1;
          const int j = rr - py - u;           // tile row whose tap (py, u) reads halo row rr
          if (j >= 0 && j < C::RW) {
#pragma unroll
            for (int ij = 2; ij >= 0; --ij)
#pragma unroll
              for (int i = 0; i <= ij; ++i)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(py ? D1[j & 1][i] : D0[j & 1][i], B[ij - i], acc[t], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }
  if (negated) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = -acc[t];
  }
  // dW[co][ky][kx][ci] += T: the kernel rows Ky(py, u) x columns Kx(px, V) this phase tap stands for
  const int ci_g = cib * C::CIB + ciq * 32 + lr;
  const int kx0 = px == 0 ? (V ? 1 : 0) : (V ? 2 : 0), kx1 = px == 0 ? (V ? 2 : 0) : (V ? 2 : 1);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int py = t >> 1, u = t & 1;
    const int ky0 = py == 0 ? (u ? 1 : 0) : (u ? 2 : 0), ky1 = py == 0 ? (u ? 2 : 0) : (u ? 2 : 1);
#pragma unroll
    for (int ky = ky0; ky <= ky1; ++ky)
      for (int kx = kx0; kx <= kx1; ++kx)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int co_g = cob * C::COB + coq * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
          atomicAdd(a.dw + ((size_t)co_g * 9 + ky * 3 + kx) * a.ldw + ci_g, acc[t][v]);
        }
  }
}

template <int COQ, int CIQ, int TR, int TWK>
__global__ __launch_bounds__(512, 2) void conv_wgrad_up_kernel(const WgradUpArgs a) {
  using C = H2Up<COQ, CIQ, TR, TWK>;
  extern __shared__ __attribute__((aligned(16))) char h2smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4) wgrad_up_role<C, 0>(a, h2smem, tid, lane, wave);
  else wgrad_up_role<C, 1>(a, h2smem, tid, lane, wave);
}

// 0: not served; 1 / 2: 64 x 64 channel blocks, 2 x 16 a-pixel tiles (a 1 x 32 tile form for wide images was measured and dropped: its
// three-row halo for one row of pixels triples the staged x bytes); 3: 32 produced x 64 gathered channels, 1 x 32 tiles
static int h2up_config(const udaseg_conv_desc* d, int up_ca) {
  if (!d || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1) return 0;
  if (d->n <= 0 || d->hi < 2 || d->wi < 2 || d->ho != d->hi || d->wo != d->wi || d->hi % 2 || d->wi % 2) return 0;
  if (up_ca <= 0 || up_ca > d->ci || up_ca % 64 != 0 || d->co % 32 != 0) return 0;
  const long long px = (long long)d->n * d->hi * d->wi;
  if (px * d->co * 4 >= (1LL << 31) || px / 4 * up_ca * 4 >= (1LL << 31)) return 0;
  const int wa = d->wi / 2;
  if (wa < 16) return 0;
  if (d->co % 64 != 0) return wa >= 32 ? 3 : 0;
  return wa >= 32 ? 1 : 2;
}

template <int COQ, int CIQ, int TR, int TWK>
static int launch_h2up_t(const udaseg_conv_desc* d, const float* a_, int up_ca, const float* dy, float* dw, hipStream_t s, int target) {
  using C = H2Up<COQ, CIQ, TR, TWK>;
  WgradUpArgs a = {};
  a.x = a_; a.dy = dy; a.dw = dw;
  a.n = d->n; a.h = d->hi / 2; a.w = d->wi / 2; a.ca = up_ca; a.co = d->co; a.ldw = d->ci;
  a.ntx = cdiv(a.w, C::TW); a.nty = cdiv(a.h, C::TR); a.ntiles = a.n * a.ntx * a.nty;
  a.ncib = up_ca / C::CIB; a.pairs = a.ncib * (d->co / C::COB);
  int P = target / (2 * a.pairs);
  if (P < 1) P = 1;
  if (P > a.ntiles) P = a.ntiles;
  a.P = P;
  const long long pa = (long long)a.n * a.h * a.w;
  a.x_bytes = (unsigned)(pa * up_ca * 4);
  a.dy_bytes = (unsigned)(4 * pa * d->co * 4);
  auto kern = conv_wgrad_up_kernel<COQ, CIQ, TR, TWK>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv_wgrad_up)");
    attr_done = true;
  }
  static std::atomic<int> kid{-1};
  if (kid < 0) {
    char nm[96];
    snprintf(nm, sizeof(nm), "conv_wgrad_up_kernel<%d, %d, %d, %d>", COQ, CIQ, TR, TWK);
    kid = kprof_id(nm);
  }
  hipEvent_t ev = kprof_begin(s);
  hipLaunchKernelGGL(kern, dim3((unsigned)(2 * a.pairs * P)), dim3(C::NT), C::LDS, s, a);
  kprof_end(kid, ev, s, 2.0 * (double)pa * d->co * 16.0 * up_ca);
  UDASEG_LAUNCH_CHECK("conv_wgrad_up launch");
  return UDASEG_OK;
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_conv2d_wgrad_up_f32x3_ok(const udaseg_conv_desc* d, int up_ca) {
  return f32_halo_enabled() && h2up_config(d, up_ca) != 0 ? 1 : 0;
}

extern "C" int udaseg_wgrad_up_set_blocks(int blocks) {
  UDASEG_CHECK_ARG(blocks >= 0 && blocks <= 4096, "wgrad_up_set_blocks: 0 (default) .. 4096");
  return udaseg_set_option(UDASEG_OPT_WGRAD_UP_BLOCKS, blocks);
}

// dW[co][9][ci] (rows of d->ci channels, the first up_ca of them) += the weight gradient of conv3x3(nearest_x2(a)) in the phase form
extern "C" int udaseg_conv2d_wgrad_up_f32x3(const udaseg_conv_desc* d, const float* a, int up_ca, const float* dy, float* dw,
                                            void* stream) {
  UDASEG_CHECK_ARG(d && a && dy && dw, "conv2d_wgrad_up_f32x3: NULL pointer");
  const int cfg = f32_halo_enabled() ? h2up_config(d, up_ca) : 0;
  if (cfg == 0) {
    set_error("conv2d_wgrad_up_f32x3: geometry not supported (ask udaseg_conv2d_wgrad_up_f32x3_ok first)");
    return UDASEG_E_UNSUPPORTED;
  }
  // blocks per launch: few and long-lived like the nine-tap kernel.  Same-box sweep inside the overlapped step (images/s, phase weight
  // gradient off 983.7 / 996.7): 2 x 16 tiles at 64 / 128 / 256 blocks 995.2 / 990.1-994.4 / 969.7, 1 x 32 tiles at 128 / 256 blocks
  // 984.8 / 966.1 (profiles/r05_up_phase.txt) -- the 16-pixel tile everywhere, 96 blocks
  const int target = opt_get(UDASEG_OPT_WGRAD_UP_BLOCKS) > 0 ? opt_get(UDASEG_OPT_WGRAD_UP_BLOCKS) : 96;
  hipStream_t st = as_stream(stream);
  udaseg_conv_desc dd = *d;
  dd.ci = up_ca;
  const double flops = 2.0 * (double)d->n * (d->hi / 2) * (d->wi / 2) * d->co * 16.0 * up_ca;
  prof_begin(1, st);
  int rc;
  if (cfg == 1 || cfg == 2) rc = launch_h2up_t<2, 2, 2, 1>(d, a, up_ca, dy, dw, st, target);
  else rc = launch_h2up_t<1, 2, 1, 2>(d, a, up_ca, dy, dw, st, target);
  prof_end(1, st, flops, 2, &dd);
  return rc;
}
