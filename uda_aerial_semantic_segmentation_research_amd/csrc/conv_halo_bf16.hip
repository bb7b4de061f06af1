// bf16-first convolution kernels (round 3): stride-1 3x3 / pad 1 and 1x1 / pad 0 convolutions, forward and data gradient, NHWC
// bf16 storage, fp32 accumulation on v_mfma_f32_32x32x16_bf16 (gfx950).
//
// Replaces, for BASELINE configs 3 and 5 (bf16), the shared fp32 / bf16 implicit-GEMM source (conv_igemm.hip) on the layers that
// torch.nn.functional.conv2d / its autograd reach from smp.Unet.forward (reference src/models/train.py:341,343;
// src/models/adversarial_trainer.py:104,113).  Why a second kernel: at bf16 MFMA rates the 64x64-tile gather re-reads every
// input pixel nine times through L2 (32 FLOP per L2 byte; measured 0.12 of the MFMA peak AND 0.12 of HBM, bound by neither:
// profiles/r02_kernel_stats_bf16_*.csv).  Here
//   * a block owns a TH x TW patch of output pixels and stages the (TH+2) x (TW+2) input HALO of a channel chunk ONCE into LDS
//     (padded pixel rows: conflict-free ds_read_b128); the nine taps are nine shifted reads of that halo;
//   * the weights never touch LDS: they arrive pre-packed in MFMA-fragment order (udaseg_pack_frag_batched_bf16), one
//     1 KB coalesced load per fragment straight into registers, and a fragment is used for every row block the wave owns;
//   * operands are swapped (weights = MFMA A, pixels = MFMA B): an accumulator lane then holds 4 consecutive CHANNELS of one
//     pixel, the epilogue exchanges one register pair with the lane's partner (v_permlane32_swap) and writes 16-byte
//     row-contiguous bf16 stores straight from registers -- no LDS round trip, no barrier;
//   * the fused decoder input (cat([nearest_x2(a), skip])) is a chunk-uniform choice of source during staging, the data
//     gradient's split output a block-uniform choice of destination;
//   * training-mode BatchNorm + activation of the PRODUCER can be applied while the halo is staged (in_scale / in_shift): the
//     normalised activation of a single-consumer layer is then never written to HBM.
// FLOP per L2 byte rises from 32 to 160-200; the kernels are then bound by HBM (most layers) or by the weight stream from L2
// (the deep, low-resolution layers).
//
// Round 4: the discriminator's 4 x 4 / stride 2 / pad 1 convolutions (reference src/models/discriminator.py:15-34, called
// src/models/adversarial_trainer.py:87,88,108) run here too, as the KS = 2 window of the same kernel:
//   forward      y[oy][ox] = sum_{a,b in {0,1}} sum_{p,q in {0,1}} W[2a+p][2b+q] . x[2(oy-1+a)+1+p][2(ox-1+b)+1+q]
//                -- a stride-1 2 x 2 window (offsets -1, 0) over the four PARITY PHASES of the input, which the staging loop
//                gathers as 4 ci "virtual" channels (phase-major): the halo of a chunk is de-interleaved by input parity while it
//                is staged, so every fragment read is the conflict-free stride-1 read of the 3 x 3 kernel;
//   data grad    the four parity classes of dx are four stride-1 2 x 2 windows over dy (offsets -1, 0 for even input rows /
//                columns, 0, +1 for odd ones) written with stride 2 into dx: four launches of the same instantiation.
#include <stdlib.h>

#include "common.h"
#include "halo_common.h"

namespace udaseg {


struct HaloArgs {
  const void* x;        // gathered tensor [n][h][w][cx] bf16, or the HALF-resolution tensor a [n][h/2][w/2][up_ca] (up_ca > 0)
  const void* x2;       // up_ca > 0: the full-resolution skip tensor [n][h][w][ci - up_ca] (null when ci == up_ca)
  const void* wf;       // fragment-packed weights (see pack_frag_batched_bf16_kernel)
  const float* bias;    // [co] or null
  void* y;              // [n][h][w][co] bf16 (fp32 when out_f32), or channels [0, split_n) of a split output
  void* y2;             // channels [split_n, co) of a split output
  int n, h, w;          // output = input extent (stride 1, "same" padding)
  int ci, co;           // gathered / produced channel counts
  int up_ca, split_n;
  int accumulate;       // bf16 output only: y += result (one rounding of the fp32 sum), the identity branch's gradient in dx
  int out_f32, act;
  float slope;
  double* stats;        // [R][2][co] f64: BatchNorm statistics of the output, or the bnb_* sums
  double* sscr;         // launches of > 1024 blocks: f64 partial sums [HALO_SCR_REPLICAS][2][co] (zeroed scratch), folded into `stats`
                        // by halo_stats_fold_kernel -- 8192 blocks on 16 replicas put 512 same-address f64 atomics (~0.1 us each,
                        // serialised at the memory side) on every accumulator of a 16-channel layer: the atomics WERE the launch
  // BatchNorm-backward reductions of the layer behind a data gradient (see IgemmArgs::bnb_* in conv_igemm.hip)
  const void* bnb_y;
  const float* bnb_mean;
  const float* bnb_rstd;
  const float* bnb_gamma;
  const float* bnb_beta;
  int bnb_act;
  float bnb_slope;
  // producer's BatchNorm + activation applied to the gathered tensor while it is staged: v -> act(v * in_scale[c] + in_shift[c])
  const float* in_scale;
  const float* in_shift;
  int in_act;
  float in_slope;
  int ntx, nty, ncb;    // tiles along x / y, channel blocks of 32 * WN
  int nk16;             // ci / 16
  unsigned x_bytes, x2_bytes, w_bytes, y_bytes, y2_bytes, bnb_bytes;
  unsigned long long* timeline;   // udaseg_debug_set_timeline: per block {entry, first chunk staged, end of K loop, exit} + HW_ID + XCC_ID
  // KS = 2 launches only (4 x 4 / stride 2 convolutions):
  int s2;               // forward: the gathered tensor is the REAL input [n][2h][2w][cr]; ci = 4 cr virtual channels (phase-major)
  int cr, cr_log2;      // real gathered channels (a power of two)
  int pady, padx;       // window offset of tap 0 along y / x: 1 (rows o - 1, o) or 0 (rows o, o + 1)
  int out_h, out_w, out_sy, out_oy, out_sx, out_ox;   // produced pixel (oy, ox) is stored at (oy out_sy + out_oy, ox out_sx + out_ox) of [n][out_h][out_w]
  int ncls;             // 4: the four parity classes of a stride-2 data gradient in ONE launch (class = block index % 4: window offsets,
  unsigned cls_wbytes;  //    output offsets and the class's fragment packing, cls_wbytes apart, follow from it)
};
extern unsigned long long* g_timeline;
extern int g_timeline_blocks;


template <int KS, int CK, int WM, int WN, int RPW, int TW>
struct HaloCfg {
  static constexpr int NT = 64 * WM * WN;
  static constexpr int RB = 32 / TW;               // image rows per 32-pixel MFMA block
  static constexpr int TH = WM * RPW * RB;
  static constexpr int PAD = KS / 2;
  static constexpr int HR = TH + KS - 1, HWD = TW + KS - 1;
  static constexpr int OCT = CK / 8;               // 16-byte pieces per pixel per chunk
  static constexpr int STRIDE = CK * 2 + 16;       // padded pixel row in LDS (bytes)
  static constexpr int NPIECE = HR * HWD * OCT;
  static constexpr int NI = (NPIECE + NT - 1) / NT;
  static constexpr int LDS_HALO = HR * HWD * STRIDE;
  static constexpr int S = RB * (RPW - 1) + KS;   // distinct start rows of pixel fragments per (dx, k16)
  static constexpr int GPC = KS * (CK / 16);       // (dx, k16) groups per chunk
  static constexpr int FPC = GPC * KS;             // weight fragments per chunk per 32-channel block
  static constexpr int NW = WM * WN;               // waves
  static constexpr int NWI = (WN * FPC + NW - 1) / NW;   // weight fragments a wave stages per chunk
  static constexpr int LDS_W = WN * FPC * 1024;
  static constexpr int LDS = LDS_HALO + LDS_W;     // (the statistics reduction reuses the halo region: 2 * NW * 32 floats)
  static_assert(LDS_HALO >= 2 * NW * 32 * 4 && LDS_HALO % 16 == 0, "reduction scratch fits the halo region");
};

template <int KS, int CK, int WM, int WN, int RPW, int TW>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_halo_bf16_kernel(const HaloArgs a) {
  using C = HaloCfg<KS, CK, WM, WN, RPW, TW>;
  static_assert(C::NT % C::OCT == 0, "a thread stages one fixed channel octet");
  static_assert(TW == 32 || TW == 16, "tile width");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: everything derived from it stays scalar
  const int lp = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  unsigned long long tl0 = 0, tl1 = 0, tl2 = 0;
  if (a.timeline) tl0 = wall_clock64();

  // ---- block -> (image, tile, channel block); XCD-aware: blocks sharing blockIdx % 8 take a contiguous run of tiles
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int pady = a.pady, padx = a.padx, out_oy = a.out_oy, out_ox = a.out_ox;
  unsigned wcls = 0;
  if constexpr (KS == 2) {
    if (a.ncls > 1) {
      const int e = bid % a.ncls;
      bid /= a.ncls;
      pady = (e >> 1) ? 0 : 1; padx = (e & 1) ? 0 : 1;      // even input rows read output rows (i - 1, i), odd ones (i, i + 1)
      out_oy = e >> 1; out_ox = e & 1;
      wcls = (unsigned)e * a.cls_wbytes;
    }
  }
  const int cb = bid % a.ncb;
  int t = bid / a.ncb;
  const int tx = t % a.ntx;
  t /= a.ntx;
  const int ty = t % a.nty;
  const int img = t / a.nty;
  const int y0 = ty * C::TH, x0 = tx * TW;
  const int H = a.h, W = a.w;

  // ---- staging slots: piece = tid + i * NT -> (halo pixel, octet)
  const int oct = tid % C::OCT;
  unsigned voff[C::NI], voff2[C::NI];
  unsigned s2ok[KS == 2 ? C::NI : 1];
  unsigned okbits = 0;
  const bool UPC = a.up_ca > 0;
  const int cx = UPC ? a.up_ca : a.ci, cx2 = a.ci - a.up_ca;
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    const int piece = tid + i * C::NT;
    const int pix = piece / C::OCT;
    const int hy = pix / C::HWD, hx = pix - hy * C::HWD;
    const int iy = y0 + hy - (KS == 2 ? pady : C::PAD), ix = x0 + hx - (KS == 2 ? padx : C::PAD);
    bool ok = piece < C::NPIECE && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    if constexpr (KS == 2) {
      if (a.s2) {
        // phase (p, q) of virtual pixel (iy, ix) is the real pixel (2 iy + 1 + p, 2 ix + 1 + q): the offset of phase (0, 0) and one
        // validity bit per phase (virtual row -1 is real row 0 for p = 1; virtual row h - 1 is past the end for p = 1)
        const int ry = 2 * iy + 1, rx = 2 * ix + 1, HR2 = 2 * H, WR2 = 2 * W;
        unsigned m = 0;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
          m |= (piece < C::NPIECE && (unsigned)(ry + (ph >> 1)) < (unsigned)HR2 && (unsigned)(rx + (ph & 1)) < (unsigned)WR2 ? 1u : 0u) << ph;
        s2ok[i] = m;
        voff[i] = (unsigned)((((img * HR2 + ry) * WR2 + rx) * a.cr) * 2);      // may wrap for ry = -1: only used with its phase offset added
        voff2[i] = 0x80000000u;
        ok = m != 0;
        okbits |= (ok ? 1u : 0u) << i;
        continue;
      }
    }
    okbits |= (ok ? 1u : 0u) << i;
    if (UPC) {
      voff[i] = ok ? (unsigned)((((img * (H >> 1) + (iy >> 1)) * (W >> 1) + (ix >> 1)) * cx + oct * 8) * 2) : 0x80000000u;
      voff2[i] = ok ? (unsigned)((((img * H + iy) * W + ix) * cx2 + oct * 8) * 2) : 0x80000000u;
    } else {
      voff[i] = ok ? (unsigned)((((img * H + iy) * W + ix) * cx + oct * 8) * 2) : 0x80000000u;
      voff2[i] = 0x80000000u;
    }
  }
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(UPC && a.x2 ? a.x2 : a.x), 0,
                                                                   (int)(UPC && a.x2 ? a.x2_bytes : 0u), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wf), 0, (int)a.w_bytes, 0x00020000);

  // ---- this wave's channel block
  const int nblocks32 = (a.co + 31) >> 5;
  int nb = cb * WN + wn;
  const bool wave_live = nb < nblocks32;          // a dead wave (co not a multiple of 32 * WN) computes on zeros and stores nothing
  if (!wave_live) nb = 0;
  // ---- weight staging: the block's WN x FPC fragments of a chunk (1 KB each, fragment order [dx][k16][dy]) are loaded by the
  // waves round-robin, lane-linear (one coalesced 1 KB load per fragment), one chunk AHEAD of their use -- a fragment fetched
  // right before its MFMAs costs an L2 round trip per (dx, k16) group: 35 us instead of ~8 for a 64-channel layer (round 3)
  const int frag_per_nb = KS * a.nk16 * KS;       // fragments of one 32-channel block
  int wvoff[C::NWI];                              // scalar part of a fragment's offset (the lane's 16 bytes go in the vector offset)
  const unsigned wlane16 = (unsigned)lane * 16u;
#pragma unroll
  for (int i = 0; i < C::NWI; ++i) {
    const int q = wave + C::NW * i;               // fragment slot of the block: (channel block q / FPC, fragment q % FPC)
    const int wq = q / C::FPC, fi = q - wq * C::FPC;
    const int g = fi / KS, dy = fi - g * KS;
    const int dx = g / (CK / 16), k = g - dx * (CK / 16);
    int nbq = cb * WN + wq;
    const bool live = q < WN * C::FPC && nbq < nblocks32;
    wvoff[i] = live ? (int)wcls + (nbq * frag_per_nb + (dx * a.nk16 + k) * KS + dy) * 1024 : -1;
  }
  char* wlds = smem + C::LDS_HALO;
  const int wrd = (wn * C::FPC) * 1024 + lane * 16;     // this wave's fragments in the weight region

  // pixel fragment base address in the halo: lane pixel (ly, lx), K half lh
  const int ly = lp / TW, lx = lp % TW;
  const int pbase = ((wm * RPW * C::RB + ly) * C::HWD + lx) * C::STRIDE + lh * 16;

  f32x16 acc[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[r][v] = 0.f;

  const int nchunk = (a.ci + CK - 1) / CK;
  u32x4 stage[C::NI];
  auto load_chunk = [&](int c) {
    const int cbeg = c * CK;
    const bool second = UPC && cbeg >= a.up_ca;
    const int soff = (second ? cbeg - a.up_ca : cbeg) * 2;
    // channel tail (gathered channels not a multiple of the chunk, e.g. the 24-channel logits gradient): octets past the last
    // channel read zeros instead of the next pixel (their weights are zero, but 0 x NaN is not)
    const unsigned kill = (cbeg + oct * 8 < a.ci) ? 0u : 0x80000000u;
    if constexpr (KS == 2) {
      if (a.s2) {
        // this thread's octet of the chunk: virtual channel -> (phase, real channel); one phase per thread per chunk
        const int vch = cbeg + oct * 8, ph = vch >> a.cr_log2, cc = vch - (ph << a.cr_log2);
        const unsigned poff = (unsigned)((((ph >> 1) * 2 * W + (ph & 1)) * a.cr + cc) * 2);
#pragma unroll
        for (int i = 0; i < C::NI; ++i)
          stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)((((s2ok[i] >> ph) & 1u) ? voff[i] + poff : 0x80000000u) | kill), 0, 0);
        return;
      }
    }
    if (second) {
#pragma unroll
      for (int i = 0; i < C::NI; ++i) stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x2, (int)(voff2[i] | kill), soff, 0);
    } else {
#pragma unroll
      for (int i = 0; i < C::NI; ++i) stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(voff[i] | kill), soff, 0);
    }
  };
  auto store_chunk = [&](int c) {
    if (a.in_scale != nullptr) {
      // the producer's BatchNorm + activation, applied in registers; halo pixels outside the image stay zero (the padding is
      // applied to the ACTIVATION, after the transform)
      const int ch = c * CK + oct * 8;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(a.in_scale + ch), s1 = *reinterpret_cast<const f32x4*>(a.in_scale + ch + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(a.in_shift + ch), h1 = *reinterpret_cast<const f32x4*>(a.in_shift + ch + 4);
#pragma unroll
      for (int i = 0; i < C::NI; ++i) {
        u32x4 d = stage[i];
        const bool ok = (okbits >> i) & 1u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sc0 = e < 2 ? s0[2 * e] : s1[2 * e - 4], sc1 = e < 2 ? s0[2 * e + 1] : s1[2 * e - 3];
          const float sh0 = e < 2 ? h0[2 * e] : h1[2 * e - 4], sh1 = e < 2 ? h0[2 * e + 1] : h1[2 * e - 3];
          float t0 = __builtin_fmaf(bf_lo(d[e]), sc0, sh0), t1 = __builtin_fmaf(bf_hi(d[e]), sc1, sh1);
          if (a.in_act == UDASEG_ACT_LEAKY && a.in_slope == 0.f) {      // uniform: ReLU is one v_max
            t0 = t0 > 0.f ? t0 : 0.f;
            t1 = t1 > 0.f ? t1 : 0.f;
          } else {
            t0 = act_apply(t0, a.in_act, a.in_slope);
            t1 = act_apply(t1, a.in_act, a.in_slope);
          }
          d[e] = ok ? pack_bf16x2(t0, t1) : 0u;
        }
        stage[i] = d;
      }
    }
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
      const int piece = tid + i * C::NT;
      if (i < C::NI - 1 || piece < C::NPIECE) {
        const int pix = piece / C::OCT;
        *reinterpret_cast<u32x4*>(smem + pix * C::STRIDE + oct * 16) = stage[i];
      }
    }
  };

  u32x4 wstage[C::NWI];
  auto load_w = [&](int c) {
    const int soff = c * (CK / 16) * KS * 1024;   // k16 advances by CK/16 per chunk
#pragma unroll
    for (int i = 0; i < C::NWI; ++i)
      wstage[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wvoff[i] < 0 ? 0x80000000u : wlane16), wvoff[i] < 0 ? 0 : wvoff[i] + soff, 0);
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < C::NWI; ++i) {
      const int q = wave + C::NW * i;
      if (i < C::NWI - 1 || q < WN * C::FPC) *reinterpret_cast<u32x4*>(wlds + q * 1024 + lane * 16) = wstage[i];
    }
  };

  load_w(0);
  load_chunk(0);
  for (int c = 0; c < nchunk; ++c) {
    store_chunk(c);
    store_w();
    __syncthreads();
    if (a.timeline && c == 0) tl1 = wall_clock64();
    if (c + 1 < nchunk) {
      load_w(c + 1);
      load_chunk(c + 1);
    }
#pragma unroll
    for (int g = 0; g < C::GPC; ++g) {
      const int dx = g / (CK / 16), k = g % (CK / 16);
      u32x4 bf[KS];
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) bf[dy] = *reinterpret_cast<const u32x4*>(wlds + wrd + (g * KS + dy) * 1024);
#pragma unroll
      for (int s = 0; s < C::S; ++s) {
        const u32x4 pf = *reinterpret_cast<const u32x4*>(smem + pbase + (s * C::HWD + dx) * C::STRIDE + k * 32);
#pragma unroll
        for (int dy = 0; dy < KS; ++dy) {
          if ((s - dy) >= 0 && (s - dy) % C::RB == 0 && (s - dy) / C::RB < RPW) {
            const int r = (s - dy) / C::RB;
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[dy]), __builtin_bit_cast(bf16x8, pf),
                                                             acc[r], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }

  if (a.timeline) tl2 = wall_clock64();
  // ---- epilogue.  acc[r][v] of lane (lp, lh): channel nb*32 + (v&3) + 8*(v>>2) + 4*lh of pixel (row block r, lane pixel lp)
  const int cbase = nb * 32;
  void* yb = a.y;
  int ldc = a.co, csub = 0;
  unsigned ybytes = a.y_bytes;
  if (a.split_n > 0) {
    if (cbase >= a.split_n) { yb = a.y2; ldc = a.co - a.split_n; csub = a.split_n; ybytes = a.y2_bytes; }
    else ldc = a.split_n;
  }
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(yb, 0, (int)ybytes, 0x00020000);
  const bool want_stats = a.stats != nullptr && a.bnb_y == nullptr;
  const bool want_bnb = a.bnb_y != nullptr;
  float sA[16], sB[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) sA[v] = sB[v] = 0.f;
  float bv[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) bv[v] = 0.f;
  if (a.bias != nullptr) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int c = cbase + (v & 3) + 8 * (v >> 2) + 4 * lh;
      bv[v] = c < a.co ? a.bias[c] : 0.f;
    }
  }
  __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(want_bnb ? a.bnb_y : a.x), 0,
                                                                  (int)(want_bnb ? a.bnb_bytes : 0u), 0x00020000);
  const int es = a.out_f32 ? 4 : 2;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int oy = y0 + (wm * RPW + r) * C::RB + ly, ox = x0 + lx;
    const bool pv = wave_live && oy < H && ox < W;
    const unsigned pixoff = KS == 2 ? (unsigned)((img * a.out_h + oy * a.out_sy + out_oy) * a.out_w + ox * a.out_sx + out_ox)
                                    : (unsigned)((img * H + oy) * W + ox);
    unsigned dw[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = cbase + 8 * g + 4 * lh;      // this lane's 4 consecutive channels of group g
      const bool cv = pv && c0 < a.co;
      float val[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) val[e] = acc[r][4 * g + e] + bv[4 * g + e];
      if (want_stats) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float q = cv ? val[e] : 0.f;
          sA[4 * g + e] += q;
          sB[4 * g + e] = __builtin_fmaf(q, q, sB[4 * g + e]);
        }
      }
      if (a.act != UDASEG_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = act_apply(val[e], a.act, a.slope);
      }
      if (a.accumulate) {     // host-checked: bf16 output, no split
        const unsigned ooff = cv ? (pixoff * (unsigned)ldc + (unsigned)(c0 - csub)) * 2u : 0x80000000u;
        const u32x2 old = __builtin_amdgcn_raw_buffer_load_b64(rs_y, (int)ooff, 0, 0);
        val[0] += bf_lo(old[0]); val[1] += bf_hi(old[0]); val[2] += bf_lo(old[1]); val[3] += bf_hi(old[1]);
      }
      if (a.out_f32) {
        const unsigned off = cv ? (pixoff * (unsigned)ldc + (unsigned)(c0 - csub)) * 4u : 0x80000000u;
        u32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = __builtin_bit_cast(unsigned, val[e]);
        __builtin_amdgcn_raw_buffer_store_b128(d, rs_y, (int)off, 0, 0);
      } else {
        dw[2 * g] = pack_bf16x2(val[0], val[1]);
        dw[2 * g + 1] = pack_bf16x2(val[2], val[3]);
        if (want_bnb) {
          // g from the bf16-ROUNDED gradient (what a stand-alone bn_bwd_reduce would read back), the activation's argument
          // re-evaluated from the producer's conv output with bn_apply's own fused multiply-add
          const unsigned poff = cv ? (pixoff * (unsigned)a.co + (unsigned)c0) * 2u : 0x80000000u;
          const u32x2 yv = __builtin_amdgcn_raw_buffer_load_b64(rs_p, (int)poff, 0, 0);
          const f32x4 mu = *reinterpret_cast<const f32x4*>(a.bnb_mean + (c0 < a.co ? c0 : 0));
          const f32x4 rsd = *reinterpret_cast<const f32x4*>(a.bnb_rstd + (c0 < a.co ? c0 : 0));
          const f32x4 gm = *reinterpret_cast<const f32x4*>(a.bnb_gamma + (c0 < a.co ? c0 : 0));
          const f32x4 bt = *reinterpret_cast<const f32x4*>(a.bnb_beta + (c0 < a.co ? c0 : 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float yy = (e & 1) ? bf_hi(yv[e >> 1]) : bf_lo(yv[e >> 1]);
            const float gr = (e & 1) ? bf_hi(dw[2 * g + (e >> 1)]) : bf_lo(dw[2 * g + (e >> 1)]);
            const float sc = gm[e] * rsd[e], sh = bt[e] - mu[e] * sc;      // as bn_apply / bn_finalize form them
            const float gg = cv ? gr * act_grad(__builtin_fmaf(yy, sc, sh), a.bnb_act, a.bnb_slope) : 0.f;
            sA[4 * g + e] += gg;
            sB[4 * g + e] = __builtin_fmaf(gg, (yy - mu[e]) * rsd[e], sB[4 * g + e]);
          }
        }
      }
    }
    if (!a.out_f32) {
      // lanes (lp, 0) and (lp, 1) hold channels 8g + {0..3} and 8g + {4..7}: after the swap the low lane owns all 8 channels of
      // the even group, the high lane those of the odd group -> two 16-byte stores per lane per row block
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(dw[4 * gp], dw[4 * gp + 2], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(dw[4 * gp + 1], dw[4 * gp + 3], false, false);
        const u32x4 d = {s0[0], s1[0], s0[1], s1[1]};
        const int c8 = cbase + 8 * (2 * gp + lh);
        const unsigned off = (pv && c8 < a.co) ? (pixoff * (unsigned)ldc + (unsigned)(c8 - csub)) * 2u : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(d, rs_y, (int)off, 0, 0);
      }
    }
  }
  (void)es;

  if (want_stats || want_bnb) {
    // per-channel totals: 32 lanes of a half-wave hold the same channels -> DPP reduction over lane bits 0..4, then the waves
    // that share a channel block fold through LDS (the halo is free: the K loop ended with a barrier), one f64 atomic per
    // (channel, statistic) per block into replica blockIdx % R
    asm volatile("s_nop 1");      // the last VALU writes of sA / sB are at least two wait states behind
    halfwave_sum_n(sA);
    halfwave_sum_n(sB);
    asm volatile("s_nop 1");      // ... and so are the ordinary reads of the DPP results
    float* red = reinterpret_cast<float*>(smem);   // [2][waves][32]
    if (lp == 31) {                                // lanes 31 and 63 hold the totals of their half-wave
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int cl = (v & 3) + 8 * (v >> 2) + 4 * lh;
        red[wave * 32 + cl] = wave_live ? sA[v] : 0.f;
        red[C::NW * 32 + wave * 32 + cl] = wave_live ? sB[v] : 0.f;
      }
    }
    __syncthreads();
    if (tid < 32 * WN) {
      const int wc = tid >> 5, cl = tid & 31;
      const int c = (cb * WN + wc) * 32 + cl;
      if (c < a.co) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
          t1 += red[(m * WN + wc) * 32 + cl];
          t2 += red[C::NW * 32 + (m * WN + wc) * 32 + cl];
        }
        if (a.sscr != nullptr) {     // f64 like the accumulators themselves: the order of the adds must not show (the forward
          double* rep = a.sscr + (size_t)(blockIdx.x % HALO_SCR_REPLICAS) * 2 * a.co;     // is bitwise reproducible)
          atomicAdd(rep + c, (double)t1);
          atomicAdd(rep + a.co + c, (double)t2);
        } else {
          double* rep = a.stats + (size_t)(blockIdx.x % HALO_STATS_REPLICAS) * 2 * a.co;
          atomicAdd(rep + c, (double)t1);
          atomicAdd(rep + a.co + c, (double)t2);
        }
      }
    }
  }
  if (a.timeline && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): the block's stores have left
    unsigned long long* t = a.timeline + (size_t)blockIdx.x * 6;
    t[0] = tl0; t[1] = tl1; t[2] = tl2; t[3] = wall_clock64();
    t[4] = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    t[5] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
  }
}

// ------------------------------------------------------------------------------------------------ streaming 1x1
// 1x1 / stride 1 convolutions (r50's bottleneck projections, BASELINE cfg 5) are GEMMs over M = n*h*w contiguous pixel rows
// with K = 64 .. 2048: at bf16 MFMA rates every one of them is HBM-bound (arithmetic intensity 50-200 FLOP/B against a machine
// balance of ~400), and what the tile kernel above measured on them was latency, not bandwidth: one or two resident blocks per
// CU, each loading, computing (0.7 us) and storing (6 us) strictly in turn -- 2.4 TB/s on the 64 -> 256 layer at 192^2.
// This kernel is a PERSISTENT streamer: 8 waves per block, one block per CU, 256-pixel tiles of the flat pixel axis (no partial
// tiles on 48- and 24-pixel-wide images), the next tile's rows requested before the current tile's epilogue so that loads,
// MFMAs and the 16-byte stores of neighbouring tiles overlap; weights of a single-chunk layer (K <= 64) stay in LDS for the
// block's whole life; the per-channel statistics of all of a block's tiles are summed in LDS and leave as one set of f64
// atomics.  Same fragment packing, same accumulator layout and register-to-global epilogue as conv_halo_bf16_kernel.
template <int CK, int WM, int WN>
struct StreamCfg {
  static constexpr int NT = 512, TP = 256;         // threads, pixels per tile
  static constexpr int RPW = 8 / WM;               // 32-pixel row blocks per wave
  static constexpr int OCT = CK / 8, STRIDE = CK * 2 + 16;
  static constexpr int NPIECE = TP * OCT, NI = NPIECE / NT;
  static constexpr int FPC = CK / 16, NW = 8;
  static constexpr int NWI = (WN * FPC + NW - 1) / NW;
  static constexpr int LDS_A = TP * STRIDE, LDS_W = WN * FPC * 1024, LDS_STATS = NW * 2 * 32 * 4;   // one slot per wave
  static constexpr int LDS = LDS_A + LDS_W + LDS_STATS;
  static_assert(WM * WN == 8 && NPIECE % NT == 0 && NI >= 1, "8 waves; whole staging passes");
};

template <int CK, int WM, int WN>
__global__ __launch_bounds__(512, 2) void conv1x1_stream_bf16_kernel(const HaloArgs a) {
  using C = StreamCfg<CK, WM, WN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lp = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int M = a.n * a.h * a.w;
  const int ntiles = (M + C::TP - 1) / C::TP;
  const int cb = (int)blockIdx.x % a.ncb, P = (int)gridDim.x / a.ncb, slot = (int)blockIdx.x / a.ncb;

  // staging slots: piece = tid + i * 512 -> (tile pixel, octet); a tile's rows are one contiguous block of memory
  const int oct = tid % C::OCT;
  unsigned voffc[C::NI];
#pragma unroll
  for (int i = 0; i < C::NI; ++i) voffc[i] = (unsigned)((((tid + i * C::NT) / C::OCT) * a.ci + oct * 8) * 2);
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wf), 0, (int)a.w_bytes, 0x00020000);

  const int nblocks32 = (a.co + 31) >> 5;
  int nb = cb * WN + wn;
  const bool wave_live = nb < nblocks32;
  if (!wave_live) nb = 0;
  const int frag_per_nb = a.nk16;                  // 1x1: fragments of a 32-channel block = k16 steps
  int wvoff[C::NWI];
  const unsigned wlane16 = (unsigned)lane * 16u;
#pragma unroll
  for (int i = 0; i < C::NWI; ++i) {
    const int q = wave + C::NW * i;
    const int wq = q / C::FPC, k = q - wq * C::FPC;
    const int nbq = cb * WN + wq;
    wvoff[i] = (q < WN * C::FPC && nbq < nblocks32) ? (nbq * frag_per_nb + k) * 1024 : -1;
  }
  char* wlds = smem + C::LDS_A;
  float* sred = reinterpret_cast<float*>(smem + C::LDS_A + C::LDS_W);
  const int wrd = (wn * C::FPC) * 1024 + lane * 16;
  const int pbase = ((wm * C::RPW * 32) + lp) * C::STRIDE + lh * 16;

  const int nchunk = (a.ci + CK - 1) / CK;
  u32x4 stage[C::NI], wstage[C::NWI];
  unsigned okbits = 0;
  auto load_chunk = [&](int tile, int c) {
    const int cbeg = c * CK;
    const unsigned kill = (cbeg + oct * 8 < a.ci) ? 0u : 0x80000000u;
    const int soff = tile * C::TP * a.ci * 2 + cbeg * 2;
    const int left = M - tile * C::TP;             // pixels of this tile that exist
    okbits = 0;
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
      const bool ok = (tid + i * C::NT) / C::OCT < left;
      okbits |= (ok ? 1u : 0u) << i;
      stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)((ok ? voffc[i] : 0x80000000u) | kill), soff, 0);
    }
  };
  auto store_chunk = [&](int c) {
    if (a.in_scale != nullptr) {
      const int ch = c * CK + oct * 8;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(a.in_scale + ch), s1 = *reinterpret_cast<const f32x4*>(a.in_scale + ch + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(a.in_shift + ch), h1 = *reinterpret_cast<const f32x4*>(a.in_shift + ch + 4);
#pragma unroll
      for (int i = 0; i < C::NI; ++i) {
        u32x4 d = stage[i];
        const bool ok = (okbits >> i) & 1u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sc0 = e < 2 ? s0[2 * e] : s1[2 * e - 4], sc1 = e < 2 ? s0[2 * e + 1] : s1[2 * e - 3];
          const float sh0 = e < 2 ? h0[2 * e] : h1[2 * e - 4], sh1 = e < 2 ? h0[2 * e + 1] : h1[2 * e - 3];
          float t0 = __builtin_fmaf(bf_lo(d[e]), sc0, sh0), t1 = __builtin_fmaf(bf_hi(d[e]), sc1, sh1);
          if (a.in_act == UDASEG_ACT_LEAKY && a.in_slope == 0.f) {      // uniform: ReLU is one v_max
            t0 = t0 > 0.f ? t0 : 0.f;
            t1 = t1 > 0.f ? t1 : 0.f;
          } else {
            t0 = act_apply(t0, a.in_act, a.in_slope);
            t1 = act_apply(t1, a.in_act, a.in_slope);
          }
          d[e] = ok ? pack_bf16x2(t0, t1) : 0u;
        }
        stage[i] = d;
      }
    }
#pragma unroll
    for (int i = 0; i < C::NI; ++i)
      *reinterpret_cast<u32x4*>(smem + ((tid + i * C::NT) / C::OCT) * C::STRIDE + oct * 16) = stage[i];
  };
  auto load_w = [&](int c) {
    const int soff = c * C::FPC * 1024;
#pragma unroll
    for (int i = 0; i < C::NWI; ++i)
      wstage[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wvoff[i] < 0 ? 0x80000000u : wlane16), wvoff[i] < 0 ? 0 : wvoff[i] + soff, 0);
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < C::NWI; ++i) {
      const int q = wave + C::NW * i;
      if (i < C::NWI - 1 || q < WN * C::FPC) *reinterpret_cast<u32x4*>(wlds + q * 1024 + lane * 16) = wstage[i];
    }
  };

  const int cbase = nb * 32;
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, 0x00020000);
  const bool want_stats = a.stats != nullptr && a.bnb_y == nullptr;
  const bool want_bnb = a.bnb_y != nullptr;
  __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(want_bnb ? a.bnb_y : a.x), 0,
                                                                  (int)(want_bnb ? a.bnb_bytes : 0u), 0x00020000);
  if ((want_stats || want_bnb) && tid < C::NW * 2 * 32) sred[tid] = 0.f;     // visible after the first barrier below
  const bool w_resident = nchunk == 1;             // single-chunk layers: the weights are staged once and stay in LDS

  if (slot < ntiles) {
    load_w(0);
    load_chunk(slot, 0);
  }
  bool first = true;
  for (int tile = slot; tile < ntiles; tile += P) {
    // opaque copies: keep the epilogue's addresses from being hoisted out of the tile loop (and held in registers across it)
    int lh_t = lh, cbase_t = cbase;
    asm volatile("" : "+v"(lh_t), "+s"(cbase_t));
    f32x16 acc[C::RPW];
#pragma unroll
    for (int r = 0; r < C::RPW; ++r)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[r][v] = 0.f;
    for (int c = 0; c < nchunk; ++c) {
      store_chunk(c);
      if (!w_resident || first) store_w();
      __syncthreads();
      if (c + 1 < nchunk) {
        load_w(c + 1);
        load_chunk(tile, c + 1);
      } else if (tile + P < ntiles) {          // the block's next tile: requested before this tile's MFMAs and epilogue
        if (!w_resident) load_w(0);
        load_chunk(tile + P, 0);
      }
#pragma unroll
      for (int k = 0; k < C::FPC; ++k) {
        const u32x4 bf = *reinterpret_cast<const u32x4*>(wlds + wrd + k * 1024);
#pragma unroll
        for (int r = 0; r < C::RPW; ++r) {
          const u32x4 pf = *reinterpret_cast<const u32x4*>(smem + pbase + r * 32 * C::STRIDE + k * 32);
          acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf), __builtin_bit_cast(bf16x8, pf), acc[r], 0, 0, 0);
        }
      }
      __syncthreads();
      first = false;
    }

    // ---- epilogue (layout as in conv_halo_bf16_kernel; the pixel index is flat)
    float sA[16], sB[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) sA[v] = sB[v] = 0.f;
#pragma unroll
    for (int r = 0; r < C::RPW; ++r) {
      const int pix = tile * C::TP + (wm * C::RPW + r) * 32 + lp;
      const bool pv = wave_live && pix < M;
      const unsigned pixoff = (unsigned)pix;
      unsigned dw[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = cbase_t + 8 * g + 4 * lh_t;
        const bool cv = pv && c0 < a.co;
        float val[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = acc[r][4 * g + e];
        if (a.bias != nullptr) {
          const f32x4 bq = *reinterpret_cast<const f32x4*>(a.bias + (c0 < a.co ? c0 : 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] += bq[e];
        }
        if (want_stats) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float q = cv ? val[e] : 0.f;
            sA[4 * g + e] += q;
            sB[4 * g + e] = __builtin_fmaf(q, q, sB[4 * g + e]);
          }
        }
        if (a.act != UDASEG_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] = act_apply(val[e], a.act, a.slope);
        }
        if (a.accumulate) {
          const unsigned ooff = cv ? (pixoff * (unsigned)a.co + (unsigned)c0) * 2u : 0x80000000u;
          const u32x2 old = __builtin_amdgcn_raw_buffer_load_b64(rs_y, (int)ooff, 0, 0);
          val[0] += bf_lo(old[0]); val[1] += bf_hi(old[0]); val[2] += bf_lo(old[1]); val[3] += bf_hi(old[1]);
        }
        dw[2 * g] = pack_bf16x2(val[0], val[1]);
        dw[2 * g + 1] = pack_bf16x2(val[2], val[3]);
        if (want_bnb) {
          const unsigned poff = cv ? (pixoff * (unsigned)a.co + (unsigned)c0) * 2u : 0x80000000u;
          const u32x2 yv = __builtin_amdgcn_raw_buffer_load_b64(rs_p, (int)poff, 0, 0);
          const f32x4 mu = *reinterpret_cast<const f32x4*>(a.bnb_mean + (c0 < a.co ? c0 : 0));
          const f32x4 rsd = *reinterpret_cast<const f32x4*>(a.bnb_rstd + (c0 < a.co ? c0 : 0));
          const f32x4 gm = *reinterpret_cast<const f32x4*>(a.bnb_gamma + (c0 < a.co ? c0 : 0));
          const f32x4 bt = *reinterpret_cast<const f32x4*>(a.bnb_beta + (c0 < a.co ? c0 : 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float yy = (e & 1) ? bf_hi(yv[e >> 1]) : bf_lo(yv[e >> 1]);
            const float gr = (e & 1) ? bf_hi(dw[2 * g + (e >> 1)]) : bf_lo(dw[2 * g + (e >> 1)]);
            const float sc = gm[e] * rsd[e], sh = bt[e] - mu[e] * sc;
            const float gg = cv ? gr * act_grad(__builtin_fmaf(yy, sc, sh), a.bnb_act, a.bnb_slope) : 0.f;
            sA[4 * g + e] += gg;
            sB[4 * g + e] = __builtin_fmaf(gg, (yy - mu[e]) * rsd[e], sB[4 * g + e]);
          }
        }
      }
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(dw[4 * gp], dw[4 * gp + 2], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(dw[4 * gp + 1], dw[4 * gp + 3], false, false);
        const u32x4 d = {s0[0], s1[0], s0[1], s1[1]};
        const int c8 = cbase_t + 8 * (2 * gp + lh_t);
        const unsigned off = (pv && c8 < a.co) ? (pixoff * (unsigned)a.co + (unsigned)c8) * 2u : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(d, rs_y, (int)off, 0, 0);
      }
    }
    if (want_stats || want_bnb) {
      asm volatile("s_nop 1");
      halfwave_sum_n(sA);
      halfwave_sum_n(sB);
      asm volatile("s_nop 1");
      // each wave adds into ITS OWN slot, tile after tile: a fixed order (an LDS atomic shared by the waves would make the
      // forward statistics depend on which wave came first; the forward is bitwise reproducible)
      if (lp == 31 && wave_live) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int cl = (v & 3) + 8 * (v >> 2) + 4 * lh_t;
          sred[(wave * 2 + 0) * 32 + cl] += sA[v];
          sred[(wave * 2 + 1) * 32 + cl] += sB[v];
        }
      }
    }
  }

  if (want_stats || want_bnb) {
    __syncthreads();
    if (tid < 32 * WN) {
      const int wc = tid >> 5, cl = tid & 31;
      const int c = (cb * WN + wc) * 32 + cl;
      if (c < a.co) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int m = 0; m < WM; ++m) {        // the waves that share this channel block, in wave order
          t1 += sred[((m * WN + wc) * 2 + 0) * 32 + cl];
          t2 += sred[((m * WN + wc) * 2 + 1) * 32 + cl];
        }
        double* rep = a.stats + (size_t)(blockIdx.x % HALO_STATS_REPLICAS) * 2 * a.co;
        atomicAdd(rep + c, (double)t1);
        atomicAdd(rep + a.co + c, (double)t2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ small-GEMM 1x1 (round 4)
// r50's bottleneck projections from 96^2 down (M = n*h*w <= 73728 pixels, K / N = 128 .. 2048 channels; reference
// smp.Unet("resnet50"), src/test_system.py:90-95, driven src/models/train.py:341,343) are ~10-GFLOP GEMMs on which the streaming
// kernel above is latency-bound: per 64-channel K step it issues one tile of loads, multiplies for 0.25 us and then waits a
// memory round trip (41-61 us a layer; round 3 handed these shapes to a vendor library, 19-27 us).  This kernel is a GEMM proper:
//   * 256 pixels x 128 channels per 8-wave block (a wave: 64 x 64), K in 64-channel stages through a THREE-stage LDS ring;
//   * both operands arrive by LDS-DMA (buffer_load ... lds, 16 bytes per lane): no staging registers, no ds_write, two stages in
//     flight across every barrier (counted vmcnt, one barrier per stage).  Pixel rows are unpadded 128-byte rows whose eight
//     16-byte pieces are XOR-swizzled with (row >> 1) & 7 -- applied to the SOURCE address, the DMA writes linearly -- so the
//     16-byte fragment reads of 32 consecutive rows are conflict-free; the weights are already in MFMA-fragment order
//     (udaseg_pack_frag_batched_bf16) and land fragment by fragment;
//   * the epilogue is the streaming kernel's: 16-byte bf16 stores from registers, BatchNorm statistics / BatchNorm-backward sums /
//     accumulation onto the destination -- what the library route had to give up.
template <int PBW_>
struct GemmCfg {
  static constexpr int NT = 512, WM = 4, WN = 2;                     // 8 waves: 4 over pixels x 2 over channels, 64 x 64 per wave ...
  static constexpr int PBW = PBW_;                                   // ... (PBW 2: 256 x 128 tile) or 32 x 64 (PBW 1: 128 x 128 tile)
  static constexpr int BM = 32 * PBW * WM, BN = 64 * WN, BK = 64, NS = 3;
  static constexpr int A_STAGE = BM * BK * 2, W_STAGE = BN * BK * 2, STAGE = A_STAGE + W_STAGE;
  static constexpr int RED = 2 * 8 * 2 * 32 * 4;                     // statistics fold: [2][8 waves][2 channel blocks][32] floats
  static constexpr int LDS = NS * STAGE + RED;
  static constexpr int A_DMA = A_STAGE / 1024 / 8, W_DMA = W_STAGE / 1024 / 8;      // 1 KB pieces per wave per stage
  static constexpr int DMA = A_DMA + W_DMA;
  static constexpr int NBB = BN / 32;                                // 32-channel blocks per tile
};
typedef __attribute__((address_space(3))) void* lds_vptr;
// (a plain function on purpose: with this builtin inside a kernel TEMPLATE the host pass drops the instantiation without a
// diagnostic -- the kernel handle stays an undefined symbol -- hipcc of ROCm 7.2)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, unsigned voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)lds_wave_base, 16, (int)voffset, soffset, 0, 0);
}

// PERSISTENT: a block walks the tiles bid, bid + grid, ...; the ring runs on across tile boundaries (the first two stages of
// the next tile are in flight during the epilogue of the current one), so a layer with four K stages per tile -- 256 -> 1024 at
// 48^2 -- does not pay a memory round trip plus a store tail per tile with nothing beside them.
template <int PBW>
__global__ __launch_bounds__(512, 2) void conv1x1_gemm_bf16_kernel(const HaloArgs a) {
  using C = GemmCfg<PBW>;
  extern __shared__ __attribute__((aligned(1024))) char gsmem[];
  char* const smem = gsmem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lp = lane & 31, lh = lane >> 5;
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int M = a.n * a.h * a.w;
  const int nblocks32 = (a.co + 31) >> 5;
  const int nk = a.ci / C::BK;
  const int ntiles = a.ntx;                          // pixel tiles x channel blocks (host)
  const int G = (int)gridDim.x;

  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wf), 0, (int)a.w_bytes, 0x00020000);
  // this wave's DMA pieces of a stage of tile t: pixel rows (8 rows x 128 bytes per piece, source pre-swizzled) and weight fragments
  auto offsets = [&](int t, unsigned (&ao)[C::A_DMA], unsigned (&wo)[C::W_DMA]) {
    const int cb = t % a.ncb, m0 = (t / a.ncb) * C::BM;
#pragma unroll
    for (int p = 0; p < C::A_DMA; ++p) {
      const int S = (wave * C::A_DMA + p) * 64 + lane, row = S >> 3, j = S & 7, c = j ^ ((row >> 1) & 7);
      ao[p] = (t < ntiles && (m0 + row) < M) ? (unsigned)(((m0 + row) * a.ci + c * 8) * 2) : 0x80000000u;
    }
#pragma unroll
    for (int p = 0; p < C::W_DMA; ++p) {
      const int q = wave * C::W_DMA + p;                 // fragment (32-channel block q >> 2, k16 step q & 3) of the stage
      const int nbq = cb * C::NBB + (q >> 2);
      wo[p] = (t < ntiles && nbq < nblocks32) ? (unsigned)(((nbq * a.nk16 + (q & 3)) * 64 + lane) * 16) : 0x80000000u;
    }
  };
  auto issue = [&](int kt, int slot, const unsigned (&ao)[C::A_DMA], const unsigned (&wo)[C::W_DMA]) {
    char* st = smem + slot * C::STAGE;
#pragma unroll
    for (int p = 0; p < C::A_DMA; ++p)
      lds_dma16(rs_x, st + (wave * C::A_DMA + p) * 1024, ao[p], kt * C::BK * 2);
#pragma unroll
    for (int p = 0; p < C::W_DMA; ++p)
      lds_dma16(rs_w, st + C::A_STAGE + (wave * C::W_DMA + p) * 1024, wo[p], kt * 4 * 1024);
  };

  // fragment read addresses inside a stage
  int prd[C::PBW][4];
#pragma unroll
  for (int pb = 0; pb < C::PBW; ++pb) {
    const int row = (wm * C::PBW + pb) * 32 + lp;
#pragma unroll
    for (int s = 0; s < 4; ++s) prd[pb][s] = row * 128 + (((2 * s + lh) ^ ((row >> 1) & 7)) * 16);
  }
  const int wrd = C::A_STAGE + (wn * 2 * 4) * 1024 + lane * 16;
  float* red = reinterpret_cast<float*>(smem + C::NS * C::STAGE);       // [2 statistics][8 waves][2 channel blocks][32]
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, 0x00020000);
  const bool want_stats = a.stats != nullptr && a.bnb_y == nullptr;
  const bool want_bnb = a.bnb_y != nullptr;
  __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(want_bnb ? a.bnb_y : a.x), 0,
                                                                  (int)(want_bnb ? a.bnb_bytes : 0u), 0x00020000);

  // the stage sequence of this block: tiles t0, t0 + G, ... x nk stages each; stage g sits in ring slot g % NS, two stages ahead
  unsigned ao_c[C::A_DMA], wo_c[C::W_DMA], ao_n[C::A_DMA], wo_n[C::W_DMA];
  int t = (int)blockIdx.x;
  offsets(t, ao_c, wo_c);
  offsets(t + G, ao_n, wo_n);
  // (a tile has at least two stages: the host requires ci >= 128)
  issue(0, 0, ao_c, wo_c);
  issue(1, 1, ao_c, wo_c);
  int g = 0;                                         // global stage index of (t, kt = 0)
  for (; t < ntiles; t += G) {
    f32x16 acc[C::PBW][2];          // [pixel block][channel block]
#pragma unroll
    for (int i = 0; i < C::PBW; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    for (int kt = 0; kt < nk; ++kt, ++g) {
      // stage g has landed (this wave's pieces: everything but the DMA of the one younger stage), every wave's has after the
      // barrier -- which also says that stage g - 1's slot has been read by all and may be refilled.  (Past the block's last
      // tile the younger "stage" is a killed DMA that still counts.)
      if constexpr (C::DMA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      static_assert(C::DMA == 6 || C::DMA == 4, "counted wait");
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < nk) issue(kt + 2, (g + 2) % C::NS, ao_c, wo_c);
      else issue(kt + 2 - nk, (g + 2) % C::NS, ao_n, wo_n);            // the next tile's first stages (killed offsets past the end)
      const char* st = smem + (g % C::NS) * C::STAGE;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        u32x4 wf[2], pf[C::PBW];
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) wf[nn] = *reinterpret_cast<const u32x4*>(st + wrd + (nn * 4 + s) * 1024);
#pragma unroll
        for (int pb = 0; pb < C::PBW; ++pb) pf[pb] = *reinterpret_cast<const u32x4*>(st + prd[pb][s]);
#pragma unroll
        for (int pb = 0; pb < C::PBW; ++pb)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
            acc[pb][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[nn]), __builtin_bit_cast(bf16x8, pf[pb]),
                                                                  acc[pb][nn], 0, 0, 0);
      }
    }
    const int cb = t % a.ncb, m0 = (t / a.ncb) * C::BM;
    // the next tile becomes the current one (its first two stages are already in flight)
#pragma unroll
    for (int p = 0; p < C::A_DMA; ++p) ao_c[p] = ao_n[p];
#pragma unroll
    for (int p = 0; p < C::W_DMA; ++p) wo_c[p] = wo_n[p];
    offsets(t + 2 * G, ao_n, wo_n);

    // ---- epilogue: the streaming kernel's, per (pixel block, channel block) of this wave
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      const int nb = cb * C::NBB + wn * 2 + nn;
      const bool live = nb < nblocks32;
      const int cbase = nb * 32;
      float sA[16], sB[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) sA[v] = sB[v] = 0.f;
#pragma unroll
      for (int pb = 0; pb < C::PBW; ++pb) {
        const int pix = m0 + (wm * C::PBW + pb) * 32 + lp;
        const bool pv = live && pix < M;
        const unsigned pixoff = (unsigned)pix;
        unsigned dw[8];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int c0 = cbase + 8 * gq + 4 * lh;
          const bool cv = pv && c0 < a.co;
          float val[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] = acc[pb][nn][4 * gq + e];
          if (a.bias != nullptr) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(a.bias + (c0 < a.co ? c0 : 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] += bq[e];
          }
          if (want_stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float q = cv ? val[e] : 0.f;
              sA[4 * gq + e] += q;
              sB[4 * gq + e] = __builtin_fmaf(q, q, sB[4 * gq + e]);
            }
          }
          if (a.act != UDASEG_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = act_apply(val[e], a.act, a.slope);
          }
          if (a.accumulate) {
            const unsigned ooff = cv ? (pixoff * (unsigned)a.co + (unsigned)c0) * 2u : 0x80000000u;
            const u32x2 old = __builtin_amdgcn_raw_buffer_load_b64(rs_y, (int)ooff, 0, 0);
            val[0] += bf_lo(old[0]); val[1] += bf_hi(old[0]); val[2] += bf_lo(old[1]); val[3] += bf_hi(old[1]);
          }
          dw[2 * gq] = pack_bf16x2(val[0], val[1]);
          dw[2 * gq + 1] = pack_bf16x2(val[2], val[3]);
          if (want_bnb) {
            const unsigned poff = cv ? (pixoff * (unsigned)a.co + (unsigned)c0) * 2u : 0x80000000u;
            const u32x2 yv = __builtin_amdgcn_raw_buffer_load_b64(rs_p, (int)poff, 0, 0);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(a.bnb_mean + (c0 < a.co ? c0 : 0));
            const f32x4 rsd = *reinterpret_cast<const f32x4*>(a.bnb_rstd + (c0 < a.co ? c0 : 0));
            const f32x4 gm = *reinterpret_cast<const f32x4*>(a.bnb_gamma + (c0 < a.co ? c0 : 0));
            const f32x4 bt = *reinterpret_cast<const f32x4*>(a.bnb_beta + (c0 < a.co ? c0 : 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float yy = (e & 1) ? bf_hi(yv[e >> 1]) : bf_lo(yv[e >> 1]);
              const float gr = (e & 1) ? bf_hi(dw[2 * gq + (e >> 1)]) : bf_lo(dw[2 * gq + (e >> 1)]);
              const float sc = gm[e] * rsd[e], sh = bt[e] - mu[e] * sc;
              const float gg = cv ? gr * act_grad(__builtin_fmaf(yy, sc, sh), a.bnb_act, a.bnb_slope) : 0.f;
              sA[4 * gq + e] += gg;
              sB[4 * gq + e] = __builtin_fmaf(gg, (yy - mu[e]) * rsd[e], sB[4 * gq + e]);
            }
          }
        }
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const u32x2 s0 = __builtin_amdgcn_permlane32_swap(dw[4 * gp], dw[4 * gp + 2], false, false);
          const u32x2 s1 = __builtin_amdgcn_permlane32_swap(dw[4 * gp + 1], dw[4 * gp + 3], false, false);
          const u32x4 d = {s0[0], s1[0], s0[1], s1[1]};
          const int c8 = cbase + 8 * (2 * gp + lh);
          const unsigned off = (pv && c8 < a.co) ? (pixoff * (unsigned)a.co + (unsigned)c8) * 2u : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b128(d, rs_y, (int)off, 0, 0);
        }
      }
      if (want_stats || want_bnb) {
        asm volatile("s_nop 1");
        halfwave_sum_n(sA);
        halfwave_sum_n(sB);
        asm volatile("s_nop 1");
        if (lp == 31) {
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int cl = (v & 3) + 8 * (v >> 2) + 4 * lh;
            red[(wave * 2 + nn) * 32 + cl] = live ? sA[v] : 0.f;
            red[512 + (wave * 2 + nn) * 32 + cl] = live ? sB[v] : 0.f;
          }
        }
      }
    }
    if (want_stats || want_bnb) {
      // (the epilogue's stores and loads must not hold the next tile's counted DMA waits hostage: they are older than the two
      // stages in flight, so vmcnt(DMA) at the next stage waits for them too -- which is what orders `red` as well)
      __syncthreads();
      if (tid < C::BN) {           // one thread per channel of the block: the WM pixel-row waves that share it, in wave order
        const int cl = tid & 31, q = tid >> 5;                 // q: 32-channel block of the tile = (wn, nn)
        const int c = cb * C::BN + q * 32 + cl;
        if (c < a.co) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int m = 0; m < C::WM; ++m) {
            const int w = m * C::WN + (q >> 1);
            t1 += red[(w * 2 + (q & 1)) * 32 + cl];
            t2 += red[512 + (w * 2 + (q & 1)) * 32 + cl];
          }
          double* rep = a.sscr != nullptr ? a.sscr + (size_t)(blockIdx.x % HALO_SCR_REPLICAS) * 2 * a.co
                                          : a.stats + (size_t)(blockIdx.x % HALO_STATS_REPLICAS) * 2 * a.co;
          atomicAdd(rep + c, (double)t1);
          atomicAdd(rep + a.co + c, (double)t2);
        }
      }
      __syncthreads();             // `red` may be rewritten by the next tile
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the killed DMAs of the stages past the end
}

// ------------------------------------------------------------------------------------------------ fragment packing
// Weights in MFMA-fragment order.  For a convolution with N produced and K gathered channels and a KS x KS window:
//   packed[nb][dx][k16][dy][lane][j]  (nb < ceil(N/32), k16 < ceil(K/16), lane < 64, j < 8; bf16)
//     = Wsrc[n = 32 nb + (lane & 31)][tap][k = 16 k16 + 8 (lane >> 5) + j]      (zero where n >= N or k >= K)
// i.e. exactly the 16 bytes lane `lane` feeds v_mfma_f32_32x32x16_bf16 as its A operand.  Wsrc is [N][KS*KS][K] bf16 in both
// directions: the forward reads the OHWI weights (N = co, K = ci, tap = dy * KS + dx), the data gradient reads the dgrad
// packing [ci][taps][co] (N = ci, K = co) with the window flipped (tap = KS*KS - 1 - (dy * KS + dx)).
// table row (int32 x 6): {mode (0 forward / 1 data gradient), src element offset, dst element offset, N, K, KS}
// 4 x 4 / stride 2 convolutions (KS = 4 in the row; packed as the 2 x 2 window the kernel runs, see the head of this file):
//   mode 2      forward: N = co, K = ci; packed K' = 4 K virtual channels, phase-major: W'[n][(a, b)][(2p + q) K + c] = W[n][2a + p][2b + q][c]
//   mode 3 + e  data gradient of input parity class e = 2 ey + ex, from the dgrad packing wt[ci][16][co] (N = ci, K = co):
//               W'[n][(a, b)][k] = wt[n][ky][kx][k],  ky = (ey ? 2 : 3) - 2a,  kx = (ex ? 2 : 3) - 2b
__global__ void pack_frag_batched_bf16_kernel(const __bf16* __restrict__ w16, const __bf16* __restrict__ wt16,
                                              __bf16* __restrict__ packed, const int* __restrict__ table) {
  const int* e = table + 6 * blockIdx.y;
  const int mode = e[0], N = e[3];
  if (mode >= 2) {
    const int Kr = e[4];                               // real K of the source
    const __bf16* src = (mode == 2 ? w16 : wt16) + e[1];
    __bf16* dst = packed + e[2];
    const int Kv = mode == 2 ? 4 * Kr : Kr;            // K of the packed 2 x 2 window
    const int nb = (N + 31) >> 5, nk16 = (Kv + 15) >> 4;
    const int ey = (mode - 3) >> 1, ex = (mode - 3) & 1;
    const long long total = (long long)nb * 2 * nk16 * 2 * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
      const int lane = (int)(i & 63);
      long long f = i >> 6;
      const int ta = (int)(f % 2);                     // window row (dy)
      f /= 2;
      const int kk = (int)(f % nk16);
      f /= nk16;
      const int tb = (int)(f % 2);                     // window column (dx)
      const int b = (int)(f / 2);
      const int n = b * 32 + (lane & 31), k0 = kk * 16 + 8 * (lane >> 5);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (n < N && k0 < Kv) {
        int ky, kx, c0;
        if (mode == 2) {
          const int ph = k0 / Kr;
          c0 = k0 - ph * Kr;
          ky = 2 * ta + (ph >> 1);
          kx = 2 * tb + (ph & 1);
        } else {
          c0 = k0;
          ky = (ey ? 2 : 3) - 2 * ta;
          kx = (ex ? 2 : 3) - 2 * tb;
        }
        v = *reinterpret_cast<const u32x4*>(src + ((size_t)n * 16 + ky * 4 + kx) * Kr + c0);
      }
      *reinterpret_cast<u32x4*>(dst + i * 8) = v;
    }
    return;
  }
  const int K = e[4], KS = e[5];
  const __bf16* src = (mode ? wt16 : w16) + e[1];
  __bf16* dst = packed + e[2];
  const int T = KS * KS, nb = (N + 31) >> 5, nk16 = (K + 15) >> 4;
  const long long total = (long long)nb * KS * nk16 * KS * 64;     // one 16-byte item per (fragment, lane)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    long long f = i >> 6;
    const int dy = (int)(f % KS);
    f /= KS;
    const int kk = (int)(f % nk16);
    f /= nk16;
    const int dx = (int)(f % KS);
    const int b = (int)(f / KS);
    const int n = b * 32 + (lane & 31), k0 = kk * 16 + 8 * (lane >> 5);
    const int tap = mode ? T - 1 - (dy * KS + dx) : dy * KS + dx;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (n < N && k0 < K) v = *reinterpret_cast<const u32x4*>(src + ((size_t)n * T + tap) * K + k0);
    *reinterpret_cast<u32x4*>(dst + i * 8) = v;
  }
}

// partial sums of a many-block launch -> the f64 accumulators (replica 0); leaves the scratch zeroed for the next launch.
// 16 accumulators x 16 replica groups per block: every thread has its 16 loads in flight at once and the groups fold through LDS
// in a fixed order.  (One thread walking all 256 replicas of its accumulator took 18 us per launch, 0.27 ms of an r50 step.)
constexpr int FOLD_W = 16, FOLD_G = 16;
static_assert(HALO_SCR_REPLICAS % FOLD_G == 0, "replica groups");
__global__ void __launch_bounds__(FOLD_W * FOLD_G) halo_stats_fold_kernel(double* __restrict__ sscr, int co, double* __restrict__ stats) {
  __shared__ double part[FOLD_G][FOLD_W];
  const int il = threadIdx.x % FOLD_W, g = threadIdx.x / FOLD_W;
  const int i = blockIdx.x * FOLD_W + il;      // index into [2][co]
  double v[HALO_SCR_REPLICAS / FOLD_G];
  if (i < 2 * co) {
#pragma unroll
    for (int k = 0; k < HALO_SCR_REPLICAS / FOLD_G; ++k) v[k] = sscr[(size_t)(g + FOLD_G * k) * 2 * co + i];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < HALO_SCR_REPLICAS / FOLD_G; ++k) {
      s += v[k];
      sscr[(size_t)(g + FOLD_G * k) * 2 * co + i] = 0.0;
    }
    part[g][il] = s;
  }
  __syncthreads();
  if (g == 0 && i < 2 * co) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < FOLD_G; ++k) s += part[k][il];
    atomicAdd(stats + i, s);
  }
}

// one caller-owned, caller-zeroed fp32 scratch per device for those partial sums (udaseg_set_stats_scratch)
static double* g_sscr[16] = {};
static size_t g_sscr_bytes[16] = {};

static hipStream_t g_sscr_owner[16] = {};
static bool g_sscr_owned[16] = {};
double* halo_stats_scratch(int co, hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || g_sscr[dev] == nullptr ||
      (size_t)HALO_SCR_REPLICAS * 2 * co * sizeof(double) > g_sscr_bytes[dev])
    return nullptr;
  if (!g_sscr_owned[dev]) {
    g_sscr_owned[dev] = true;
    g_sscr_owner[dev] = s;
  }
  return g_sscr_owner[dev] == s ? g_sscr[dev] : nullptr;
}

void launch_halo_stats_fold(double* sscr, int co, double* stats, hipStream_t s) {
  hipLaunchKernelGGL(halo_stats_fold_kernel, dim3(cdiv(2 * co, FOLD_W)), dim3(FOLD_W * FOLD_G), 0, s, sscr, co, stats);
}

// ------------------------------------------------------------------------------------------------- host side
template <int KS, int CK, int WM, int WN, int RPW, int TW>
static int launch_halo_t(HaloArgs a, hipStream_t s, double flops) {
  using C = HaloCfg<KS, CK, WM, WN, RPW, TW>;
  auto kern = conv_halo_bf16_kernel<KS, CK, WM, WN, RPW, TW>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done && C::LDS > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv_halo_bf16)");
    attr_done = true;
  }
  a.ntx = cdiv(a.w, TW);
  a.nty = cdiv(a.h, C::TH);
  a.ncb = cdiv(a.co, 32 * WN);
  a.nk16 = (a.ci + 15) / 16;
  const long long blocks = (long long)a.n * a.nty * a.ntx * a.ncb * (KS == 2 && a.ncls > 1 ? a.ncls : 1);
  if (blocks <= 0) return UDASEG_OK;
  a.timeline = (g_timeline && blocks <= g_timeline_blocks) ? g_timeline : nullptr;
  a.sscr = nullptr;
  if (a.stats != nullptr && blocks > 1024) a.sscr = halo_stats_scratch(a.co, s);
  static std::atomic<int> kid{-1};
  if (kid < 0) {
    char nm[96];
    snprintf(nm, sizeof(nm), "conv_halo_bf16_kernel<%d, %d, %d, %d, %d, %d>", KS, CK, WM, WN, RPW, TW);
    kid = kprof_id(nm);
  }
  hipEvent_t ev = kprof_begin(s);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::NT), C::LDS, s, a);
  kprof_end(kid, ev, s, flops);
  UDASEG_LAUNCH_CHECK("conv_halo_bf16 launch");
  if (a.sscr != nullptr) {
    launch_halo_stats_fold(a.sscr, a.co, a.stats, s);
    UDASEG_LAUNCH_CHECK("halo_stats_fold launch");
  }
  return UDASEG_OK;
}

template <int CK, int WM, int WN>
static int launch_stream_t(HaloArgs a, hipStream_t s, double flops) {
  using C = StreamCfg<CK, WM, WN>;
  auto kern = conv1x1_stream_bf16_kernel<CK, WM, WN>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done && C::LDS > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv1x1_stream_bf16)");
    attr_done = true;
  }
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hip_fail(hipGetLastError(), "hipGetDeviceProperties");
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  a.ncb = cdiv(a.co, 32 * WN);
  a.nk16 = (a.ci + 15) / 16;
  const long long tiles = cdiv64((long long)a.n * a.h * a.w, C::TP);
  if (tiles <= 0) return UDASEG_OK;
  long long P = cus / a.ncb;                 // one 8-wave block per CU
  if (P < 1) P = 1;
  if (P > tiles) P = tiles;
  static std::atomic<int> kid{-1};
  if (kid < 0) {
    char nm[96];
    snprintf(nm, sizeof(nm), "conv1x1_stream_bf16_kernel<%d, %d, %d>", CK, WM, WN);
    kid = kprof_id(nm);
  }
  hipEvent_t ev = kprof_begin(s);
  hipLaunchKernelGGL(kern, dim3((unsigned)(P * a.ncb)), dim3(C::NT), C::LDS, s, a);
  kprof_end(kid, ev, s, flops);
  UDASEG_LAUNCH_CHECK("conv1x1_stream_bf16 launch");
  return UDASEG_OK;
}

template <int PBW>
static int launch_gemm_t(HaloArgs a, hipStream_t s, double flops) {
  using C = GemmCfg<PBW>;
  auto kern = conv1x1_gemm_bf16_kernel<PBW>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv1x1_gemm_bf16)");
    attr_done = true;
  }
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hip_fail(hipGetLastError(), "hipGetDeviceProperties");
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  a.ncb = cdiv(a.co, C::BN);
  a.nk16 = (a.ci + 15) / 16;
  const long long tiles = cdiv64((long long)a.n * a.h * a.w, C::BM) * a.ncb;
  if (tiles <= 0) return UDASEG_OK;
  a.ntx = (int)tiles;
  // persistent blocks, one per CU; an even share where the tile count allows it (288 tiles: 144 blocks x 2 finish with 256 x 1 + 32 x 2)
  long long grid = tiles < cus ? tiles : cus;
  const long long per = cdiv64(tiles, grid);
  grid = cdiv64(tiles, per);
  a.sscr = nullptr;
  static std::atomic<int> kid{-1};
  if (kid < 0) {
    char nm[64];
    snprintf(nm, sizeof(nm), "conv1x1_gemm_bf16_kernel<%d>", PBW);
    kid = kprof_id(nm);
  }
  hipEvent_t ev = kprof_begin(s);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NT), C::LDS, s, a);
  kprof_end(kid, ev, s, flops);
  UDASEG_LAUNCH_CHECK("conv1x1_gemm_bf16 launch");
  return UDASEG_OK;
}

static int launch_gemm(HaloArgs a, hipStream_t s, double flops) {
  // 256 x 128 tiles while they occupy at least half the CUs; 128 x 128 below that (r50's 24^2 stage, 2048 -> 512: 72 tiles).
  // Stand-alone, us (tools/gemm1x1_probe.py, profiles/r04_gemm_1x1.txt), 256- / 128-pixel tiles: M 18432 K 1024 N 256 22.0 / 27.2;
  // M 4608 K 2048 N 512 30.7 / 23.6; M 4608 K 512 N 2048 26.8 / 27.7; M 73728 K 128 N 512 45.4 / 53.0
  const int force = opt_get(UDASEG_OPT_GEMM_1X1_TILE);      // 128 | 256 (A/B)
  const long long t256 = cdiv64((long long)a.n * a.h * a.w, 256) * cdiv(a.co, 128);
  const bool small = force == 128 || (force != 256 && t256 < 128);
  return small ? launch_gemm_t<1>(a, s, flops) : launch_gemm_t<2>(a, s, flops);
}

// The GEMM kernel takes the 1x1 launches that are GEMMs proper: whole 64-channel K stages, at least 128 produced channels, no
// transform of the gathered tensor (LDS-DMA moves bytes), and -- UDASEG_GEMM_1X1_MAXM, default 73728 = r50's 96^2 stage at batch 8 --
// few enough pixels that the streaming kernel's one-tile-per-CU schedule is latency-bound.  UDASEG_GEMM_1X1=0: never (A/B).
static bool gemm_applicable(const HaloArgs& a) {
  const int on = opt_get(UDASEG_OPT_GEMM_1X1);
  const long long maxm = opt_get(UDASEG_OPT_GEMM_1X1_MAXM);
  // expanding layers below 512 gathered channels are bound by their output stream, which the streaming kernel writes as well:
  // 128 -> 512 at 96^2 45.1 us here against 38.6, 256 -> 1024 at 48^2 33.7 against 32.4 (profiles/r04_gemm_1x1.txt)
  if (a.co > a.ci && a.ci < 512) return false;
  return on && a.in_scale == nullptr && a.ci % 64 == 0 && a.ci >= 128 && a.co % 64 == 0 && a.co >= 128 &&
         (long long)a.n * a.h * a.w <= maxm;
}

// The streamer takes the plain 1x1 launches: bf16 output, no fused decoder input, no split.
static bool stream_applicable(const HaloArgs& a, int ks) {
  const int off = opt_get(UDASEG_OPT_NO_STREAM);   // 1: 1x1 layers stay on the tile kernel (A/B)
  return !off && ks == 1 && !a.out_f32 && a.up_ca == 0 && a.split_n == 0 && a.ci % 16 == 0;
}

static int launch_stream(HaloArgs a, hipStream_t s, double flops) {
  // 128 output channels per block at most: the 256-channel form (128 accumulator registers + the prefetched tile + the
  // epilogue's sums) spills; a 256-channel layer reads its input twice instead, through L2
  if (a.ci % 64 == 0) {
    if (a.co > 64) return launch_stream_t<64, 2, 4>(a, s, flops);
    return launch_stream_t<64, 4, 2>(a, s, flops);
  }
  if (a.co > 64) return launch_stream_t<16, 2, 4>(a, s, flops);
  return launch_stream_t<16, 4, 2>(a, s, flops);
}

static int halo_cfg_override() {
  return opt_get(UDASEG_OPT_HALO_CFG);   // tuning aid: 1..n forces one configuration (0: heuristic)
}

// Can this convolution take the halo kernel?  gathered / produced: channel counts of the launch (for a data gradient: co / ci).
bool halo_applicable(const udaseg_conv_desc* d, int gathered, int produced, int up_ca) {
  if (opt_get(UDASEG_OPT_NO_HALO)) return false;   // 1: keep every layer on the shared implicit-GEMM source (A/B, cross-check)
  if (d->kh == 4 && d->kw == 4 && d->stride == 2 && d->pad == 1) {
    // the discriminator's convolutions, as 2 x 2 windows over parity phases (forward) / parity classes (data gradient)
    const int s2off = opt_get(UDASEG_OPT_NO_HALO_S2);   // 1: they stay on the shared implicit-GEMM source (A/B)
    if (s2off || up_ca != 0 || d->hi % 2 != 0 || d->wi % 2 != 0 || d->ho * 2 != d->hi || d->wo * 2 != d->wi) return false;
    if (d->ci % 8 != 0 || d->co % 8 != 0 || (d->ci & (d->ci - 1)) != 0) return false;      // real input channels: a power of two
    const long long pin = (long long)d->n * d->hi * d->wi, pout = (long long)d->n * d->ho * d->wo;
    return pin * d->ci * 2 < (1LL << 31) && pout * d->co * 4 < (1LL << 31);
  }
  if (d->kh != d->kw || (d->kh != 3 && d->kh != 1) || d->stride != 1 || d->pad != d->kh / 2) return false;
  if (gathered % 8 != 0 || produced % 8 != 0) return false;
  if (up_ca > 0 && (up_ca % 16 != 0 || (gathered - up_ca) % 16 != 0 || d->hi % 2 != 0 || d->wi % 2 != 0)) return false;
  const long long px = (long long)d->n * d->hi * d->wi;
  if (px * gathered * 2 >= (1LL << 31) || px * produced * 4 >= (1LL << 31)) return false;     // buffer descriptors: 2 GiB
  return true;
}

// Which tile configuration, or none: the launcher's choice and the "is this kernel the faster one" heuristic are one decision.
//   0: leave the layer to the shared implicit-GEMM source      1: 32 output channels per block (<= 32 produced channels)
//   2: 64 output channels per block (4 waves)                  3: 128 output channels per block (8 waves)
// Measured on MI355X (bench.py --layer-table, r18 8x512^2 and r50 8x768^2 steps, this kernel against conv_igemm_kernel<64, 64,
// ..., bf16>; profiles/r03_halo_layer_table.txt):
//   * 3x3, <= 64 produced channels at >= 128^2: always faster (16 -> 24 at 512^2 195 -> 75 us, 128 -> 32 at 256^2 117 -> 72,
//     192 -> 64 at 128^2 66 -> 49, 64 -> 64 at 128^2 40 -> 34);
//   * 3x3, >= 128 produced channels: faster while the launch still has a block per CU (384 -> 64 at 192^2 217 -> 147 us, the
//     3072-channel decoder input at 48^2 437 -> 367, its data gradient 553 -> 343), slower on the deep low-resolution layers
//     (512 -> 512 at 16^2: 32 blocks of 256 pixels x 128 channels against 256 tiles of the 64 x 64 kernel: 48 -> 78 us);
//   * 1x1 (r50): HBM-bound either way; faster only where the gradient PRODUCES more channels than it gathers (256 <- 64 at
//     192^2: 246 -> 126 us; the old kernel's one-K-tile launches), slower or equal in the forward direction.
static int halo_choice(int ks, int h, int w, int n, int gathered, int produced, bool dgrad) {
  const int ov = halo_cfg_override();
  if (ks == 2 || ks == 4) {
    // 4 x 4 / stride 2 (2 x 2 window; h, w: the convolution's INPUT extent): where the launch has a block per CU.  Per call, us,
    // this kernel / the shared source (bench.py --layer-table, cfg 3, profiles/r04_halo_s2.txt): forward 64 -> 128 at 256^2 67 / 79,
    // 128 -> 256 at 128^2 55 / 73, 256 -> 512 at 64^2 79 / 72 (128 blocks); data gradient (four parity classes in one launch)
    // 73 / 88, 58 / 75, 54 / 71
    const long long blocks = (long long)n * cdiv(h / 2, 8) * cdiv(w / 2, 32) * cdiv(produced, produced <= 64 ? 64 : 128) * (dgrad ? 4 : 1);
    if (ov == 0 && blocks < 200) return 0;
    return produced <= 64 ? 2 : 3;
  }
  if (ov > 0) return ov > 6 ? 3 : ov;
  const long long tiles = (long long)n * cdiv(h, 8) * cdiv(w, 32);
  if (ks == 1) {
    const int off = opt_get(UDASEG_OPT_NO_STREAM);       // 1 (A/B): without the streaming kernel only the expanding gradients pay
    if (off && (!dgrad || produced <= gathered)) return 0;
    return produced >= 128 ? 3 : 2;      // every 1x1 / stride 1 layer: conv1x1_stream_bf16_kernel (launch_halo routes it)
  }
  if (produced <= 32) return 1;
  if (produced <= 64) return 2;
  {
    // widths that 16 divides and 32 does not (r50 at 768^2: 48 x 48): 32-pixel-wide tiles compute a quarter of their columns
    // outside the image; 8 x 16-pixel tiles x 64 channels instead (1.0 LDS reads per MFMA against 0.75, but no dead columns and
    // twice the blocks on a launch that had ~1 per CU): r50 step 18.44 -> 18.13 ms; 16 x 16 tiles 18.24, x 128 channels 18.37
    // (profiles/r03_halo_variants.txt).  UDASEG_HALO_W16 = 0 | 4 | 5 | 6 overrides.
    const int w16 = opt_get(UDASEG_OPT_HALO_W16);
    if (w16 >= 4 && w16 <= 6 && gathered % 32 == 0 && w % 32 != 0 && w % 16 == 0 &&
        (long long)n * cdiv(h, 8) * cdiv(w, 16) * cdiv(produced, 64) >= 256)
      return w16;
  }
  if (tiles * cdiv(produced, 128) >= 192) return 3;
  if (tiles * cdiv(produced, 64) >= 256) return 2;
  {
    // deep low-resolution layers, where 32-pixel tiles leave under a block per CU: the 8 x 16-pixel x 64-channel tile again, when it
    // reaches 256 blocks and at most a quarter of its columns are outside the image.  Forward, us per launch, this tile / the shared
    // source (tools/halo_deep_probe.py, profiles/r04_halo_deep.txt): 256 -> 256 at 8 x 32^2 19.5 / 24.0, 768 -> 256 at 32^2
    // 44.7 / 54.3, 512 -> 512 at 24^2 43.6 / 48.1; 512 -> 512 at 16^2 (128 blocks) 29.7 / 28.7 stays on the shared source
    const int deep = opt_get(UDASEG_OPT_HALO_DEEP);      // 0 (A/B)
    const int wt = cdiv(w, 16);
    if (deep && gathered % 32 == 0 && 4 * w >= 3 * 16 * wt && (long long)n * cdiv(h, 8) * wt * cdiv(produced, 64) >= 256) return 6;
  }
  return 0;
}

int launch_halo(const udaseg_conv_desc* d, HaloArgs a, hipStream_t s, bool dgrad) {
  if (d->kh == 4) {
    // one 2 x 2-window launch of a 4 x 4 / stride 2 convolution (the forward, or one parity class of the data gradient); a.ci / a.co
    // are the launch's (virtual) channel counts; the FLOPs of the whole convolution are booked by the caller
    const double fl = 2.0 * (double)a.n * a.h * a.w * (double)a.co * (double)a.ci * 4.0 * (a.ncls > 1 ? a.ncls : 1);
    if (a.ci % 32 != 0) { set_error("conv_halo (4x4 / stride 2): gathered channels must be a multiple of 32"); return UDASEG_E_UNSUPPORTED; }
    const int ck64 = opt_get(UDASEG_OPT_HALO_S2_CK) == 32 ? 0 : 1;       // 32 | 64 (A/B): channels per staged chunk
    if (ck64 && a.ci % 64 == 0) {
      if (a.co <= 64) return launch_halo_t<2, 64, 2, 2, 4, 32>(a, s, fl);
      return launch_halo_t<2, 64, 2, 4, 4, 32>(a, s, fl);
    }
    if (a.co <= 64) return launch_halo_t<2, 32, 2, 2, 4, 32>(a, s, fl);
    return launch_halo_t<2, 32, 2, 4, 4, 32>(a, s, fl);
  }
  const double flops = 2.0 * (double)a.n * a.h * a.w * (double)a.co * (double)a.ci * d->kh * d->kw;
  // chunk: 32 channels (64 for the 1x1 kernels) when that divides the gathered channels and both sources of a fused input
  int ck = 32;
  if (a.ci % 32 != 0 || (a.up_ca > 0 && (a.up_ca % 32 != 0 || (a.ci - a.up_ca) % 32 != 0))) ck = 16;
  if (stream_applicable(a, d->kh)) return gemm_applicable(a) ? launch_gemm(a, s, flops) : launch_stream(a, s, flops);
  int choice = halo_choice(d->kh, a.h, a.w, a.n, a.ci, a.co, dgrad);
  if (choice == 0) choice = a.co <= 32 ? 1 : (a.co <= 64 ? 2 : 3);     // called although not preferred (tests, UDASEG_FRAG=2)
  if (choice > 3 && (d->kh != 3 || ck != 32)) choice = a.co <= 64 ? 2 : 3;
  if (d->kh == 3) {
    if (ck == 16) {
      if (choice == 1) return launch_halo_t<3, 16, 4, 1, 2, 32>(a, s, flops);
      return launch_halo_t<3, 16, 2, 2, 4, 32>(a, s, flops);
    }
    if (choice == 1) return launch_halo_t<3, 32, 4, 1, 2, 32>(a, s, flops);
    if (choice == 3) return launch_halo_t<3, 32, 2, 4, 4, 32>(a, s, flops);
    if (choice == 4) return launch_halo_t<3, 32, 2, 2, 4, 16>(a, s, flops);     // 16 x 16 pixels x 64 channels
    if (choice == 5) return launch_halo_t<3, 32, 2, 4, 4, 16>(a, s, flops);     // 16 x 16 pixels x 128 channels
    if (choice == 6) return launch_halo_t<3, 32, 2, 2, 2, 16>(a, s, flops);     //  8 x 16 pixels x 64 channels
    return launch_halo_t<3, 32, 2, 2, 4, 32>(a, s, flops);
  }
  // 1x1: two MFMAs per 16 channels -- long chunks, or the loop is all barriers
  if (ck == 32 && a.ci % 64 == 0) {
    if (choice == 3) return launch_halo_t<1, 64, 2, 4, 4, 32>(a, s, flops);
    return launch_halo_t<1, 64, 2, 2, 4, 32>(a, s, flops);
  }
  if (ck == 16) return launch_halo_t<1, 16, 2, 2, 4, 32>(a, s, flops);
  if (choice == 3) return launch_halo_t<1, 32, 2, 4, 4, 32>(a, s, flops);
  return launch_halo_t<1, 32, 2, 2, 4, 32>(a, s, flops);
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_set_stats_scratch(void* ptr, size_t bytes) {
  int dev = 0;
  UDASEG_CHECK_ARG(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, "set_stats_scratch: no current HIP device");
  UDASEG_CHECK_ARG(ptr == nullptr || (reinterpret_cast<uintptr_t>(ptr) & 15) == 0, "set_stats_scratch: 16-byte alignment");
  g_sscr[dev] = static_cast<double*>(ptr);
  g_sscr_owned[dev] = false;            // the next stream that needs it owns it
  g_sscr_bytes[dev] = ptr ? bytes : 0;
  return UDASEG_OK;
}

extern "C" int64_t udaseg_frag_elems(int n_out, int k_in, int ks) {
  if (n_out <= 0 || k_in <= 0 || ks <= 0) return 0;
  return (int64_t)((n_out + 31) / 32) * ks * ((k_in + 15) / 16) * ks * 512;
}

extern "C" int udaseg_pack_frag_batched_bf16(const void* w16, const void* wt16, void* packed, const int* table, int entries,
                                             void* stream) {
  UDASEG_CHECK_ARG(packed && table && entries > 0 && (w16 || wt16), "pack_frag_batched_bf16: NULL pointer / no entries");
  hipLaunchKernelGGL(pack_frag_batched_bf16_kernel, dim3(64, (unsigned)entries), dim3(256), 0, as_stream(stream),
                     static_cast<const __bf16*>(w16), static_cast<const __bf16*>(wt16), static_cast<__bf16*>(packed), table);
  UDASEG_LAUNCH_CHECK("pack_frag_batched_bf16 launch");
  return UDASEG_OK;
}

extern "C" int udaseg_conv_frag_ok(const udaseg_conv_desc* d, int dgrad, int up_ca) {
  if (!d) return 0;
  return halo_applicable(d, dgrad ? d->co : d->ci, dgrad ? d->ci : d->co, up_ca) ? 1 : 0;
}

extern "C" int udaseg_conv_frag_preferred(const udaseg_conv_desc* d, int dgrad, int up_ca) {
  if (!udaseg_conv_frag_ok(d, dgrad, up_ca)) return 0;
  return halo_choice(d->kh, d->hi, d->wi, d->n, dgrad ? d->co : d->ci, dgrad ? d->ci : d->co, dgrad != 0) != 0 ? 1 : 0;
}

static int frag_common(const udaseg_conv_desc* d, HaloArgs& a, const char* who) {
  UDASEG_CHECK_ARG(d != nullptr, "%s: conv desc is NULL", who);
  const bool s2 = d->kh == 4 && d->stride == 2;
  UDASEG_CHECK_ARG(d->n > 0 && d->hi > 0 && d->wi > 0 && d->ci > 0 && d->co > 0 &&
                       (s2 ? (d->ho * 2 == d->hi && d->wo * 2 == d->wi) : (d->ho == d->hi && d->wo == d->wi)),
                   "%s: stride-1 'same' convolutions or 4x4 / stride 2 / pad 1 (hi=%d wi=%d ho=%d wo=%d)", who, d->hi, d->wi, d->ho, d->wo);
  a.n = d->n; a.h = d->ho; a.w = d->wo;
  a.pady = a.padx = 1;
  a.out_h = d->ho; a.out_w = d->wo; a.out_sy = a.out_sx = 1; a.out_oy = a.out_ox = 0;
  return UDASEG_OK;
}

extern "C" int udaseg_conv2d_fwd_frag_bf16(const udaseg_conv_desc* d, const void* x, const void* skip, int up_ca, const void* wfrag,
                                           const float* bias, const float* in_scale, const float* in_shift, int in_act,
                                           float in_slope, void* y, int out_f32, int act, float slope, double* stats, void* stream) {
  HaloArgs a = {};
  int rc = frag_common(d, a, "conv2d_fwd_frag_bf16");
  if (rc) return rc;
  UDASEG_CHECK_ARG(x && wfrag && y, "conv2d_fwd_frag_bf16: NULL pointer");
  UDASEG_CHECK_ARG(up_ca >= 0 && up_ca <= d->ci && (up_ca == 0 ? skip == nullptr : (up_ca == d->ci) == (skip == nullptr)),
                   "conv2d_fwd_frag_bf16: up_ca=%d of ci=%d channels, skip %s", up_ca, d->ci, skip ? "given" : "NULL");
  UDASEG_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "conv2d_fwd_frag_bf16: in_scale and in_shift come together");
  UDASEG_CHECK_ARG(!(in_scale && (up_ca > 0 || d->ci % 16 != 0)),
                   "conv2d_fwd_frag_bf16: the input transform needs a plain input with a multiple of 16 channels");
  if (!halo_applicable(d, d->ci, d->co, up_ca)) {
    set_error("conv2d_fwd_frag_bf16: geometry not supported (ask udaseg_conv_frag_ok first)");
    return UDASEG_E_UNSUPPORTED;
  }
  const long long px = (long long)d->n * d->ho * d->wo;
  a.x = x; a.x2 = skip; a.wf = wfrag; a.bias = bias; a.y = y;
  a.ci = d->ci; a.co = d->co; a.up_ca = up_ca;
  a.out_f32 = out_f32; a.act = act; a.slope = slope; a.stats = stats;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act; a.in_slope = in_slope;
  if (d->kh == 4) {
    UDASEG_CHECK_ARG(in_scale == nullptr, "conv2d_fwd_frag_bf16: no input transform in front of a 4x4 / stride 2 convolution");
    a.s2 = 1; a.cr = d->ci; a.cr_log2 = __builtin_ctz((unsigned)d->ci); a.ci = 4 * d->ci;
    a.x_bytes = (unsigned)((long long)d->n * d->hi * d->wi * d->ci * 2);
    a.w_bytes = (unsigned)(udaseg_frag_elems(d->co, 4 * d->ci, 2) * 2);
    a.y_bytes = (unsigned)(px * d->co * (out_f32 ? 4 : 2));
    hipStream_t st = as_stream(stream);
    prof_begin(0, st);
    rc = launch_halo(d, a, st, false);
    prof_end(0, st, udaseg_conv_flops(d), 0, d);
    return rc;
  }
  a.x_bytes = (unsigned)(up_ca > 0 ? (long long)d->n * (d->hi / 2) * (d->wi / 2) * up_ca * 2 : px * d->ci * 2);
  a.x2_bytes = (unsigned)(up_ca > 0 ? px * (d->ci - up_ca) * 2 : 0);
  a.w_bytes = (unsigned)(udaseg_frag_elems(d->co, d->ci, d->kh) * 2);
  a.y_bytes = (unsigned)(px * d->co * (out_f32 ? 4 : 2));
  hipStream_t st = as_stream(stream);
  prof_begin(0, st);
  rc = launch_halo(d, a, st, false);
  prof_end(0, st, udaseg_conv_flops(d), 0, d);
  return rc;
}

extern "C" int udaseg_conv2d_dgrad_frag_bf16(const udaseg_conv_desc* d, const void* dy, const void* wfrag_t, void* dx, void* dx2,
                                             int split, const void* prev_y, const float* save_mean, const float* save_rstd,
                                             const float* gamma, const float* beta, int bn_act, float bn_slope, double* bsums,
                                             int accumulate, void* stream) {
  HaloArgs a = {};
  int rc = frag_common(d, a, "conv2d_dgrad_frag_bf16");
  if (rc) return rc;
  UDASEG_CHECK_ARG(dy && wfrag_t && dx, "conv2d_dgrad_frag_bf16: NULL pointer");
  UDASEG_CHECK_ARG(split == 0 ? dx2 == nullptr : (dx2 != nullptr && split > 0 && split < d->ci && split % 32 == 0),
                   "conv2d_dgrad_frag_bf16: split=%d needs dx2 and a multiple of 32 inside (0, ci=%d)", split, d->ci);
  UDASEG_CHECK_ARG(prev_y == nullptr || (save_mean && save_rstd && gamma && beta && bsums && split == 0 && !accumulate),
                   "conv2d_dgrad_frag_bf16: the BatchNorm-backward reductions need mean / rstd / gamma / beta / bsums, no split, "
                   "no accumulation");
  UDASEG_CHECK_ARG(!(accumulate && split > 0), "conv2d_dgrad_frag_bf16: accumulation onto a split gradient is not supported");
  if (!halo_applicable(d, d->co, d->ci, 0)) {
    set_error("conv2d_dgrad_frag_bf16: geometry not supported (ask udaseg_conv_frag_ok first)");
    return UDASEG_E_UNSUPPORTED;
  }
  const long long px = (long long)d->n * d->hi * d->wi;
  a.x = dy; a.wf = wfrag_t; a.y = dx; a.y2 = dx2; a.split_n = split;
  a.ci = d->co; a.co = d->ci;        // the launch gathers dy (co channels) and produces dx (ci channels)
  a.act = UDASEG_ACT_NONE;
  a.accumulate = accumulate ? 1 : 0;
  if (d->kh == 4) {
    // four parity classes of dx, each a 2 x 2 window over dy written with stride 2; their fragment packings follow one another
    UDASEG_CHECK_ARG(split == 0, "conv2d_dgrad_frag_bf16: no split destination behind a 4x4 / stride 2 convolution");
    const long long pout = (long long)d->n * d->ho * d->wo;
    a.x_bytes = (unsigned)(pout * d->co * 2);
    a.y_bytes = (unsigned)(px * d->ci * 2);
    const long long fe = udaseg_frag_elems(d->ci, d->co, 2);
    a.w_bytes = (unsigned)(4 * fe * 2);
    a.out_h = d->hi; a.out_w = d->wi; a.out_sy = a.out_sx = 2;
    a.ncls = 4; a.cls_wbytes = (unsigned)(fe * 2);
    if (prev_y) {
      a.bnb_y = prev_y; a.bnb_mean = save_mean; a.bnb_rstd = save_rstd; a.bnb_gamma = gamma; a.bnb_beta = beta;
      a.bnb_act = bn_act; a.bnb_slope = bn_slope; a.stats = bsums;
      a.bnb_bytes = (unsigned)(px * d->ci * 2);
    }
    hipStream_t st = as_stream(stream);
    prof_begin(0, st);
    rc = launch_halo(d, a, st, true);      // ONE launch, class = block index % 4 (four launches of a quarter each left the deep layers
                                           // with 64 blocks at a time: 256 -> 512 at 64^2 170 us against 71 for the shared source)
    prof_end(0, st, udaseg_conv_flops(d), 1, d);
    return rc;
  }
  a.x_bytes = (unsigned)(px * d->co * 2);
  a.w_bytes = (unsigned)(udaseg_frag_elems(d->ci, d->co, d->kh) * 2);
  a.y_bytes = (unsigned)(px * (split > 0 ? split : d->ci) * 2);
  a.y2_bytes = (unsigned)(split > 0 ? px * (d->ci - split) * 2 : 0);
  if (prev_y) {
    a.bnb_y = prev_y; a.bnb_mean = save_mean; a.bnb_rstd = save_rstd; a.bnb_gamma = gamma; a.bnb_beta = beta;
    a.bnb_act = bn_act; a.bnb_slope = bn_slope; a.stats = bsums;
    a.bnb_bytes = (unsigned)(px * d->ci * 2);
  }
  hipStream_t st = as_stream(stream);
  prof_begin(0, st);
  rc = launch_halo(d, a, st, true);
  prof_end(0, st, udaseg_conv_flops(d), 1, d);
  return rc;
}
