// fp32 convolutions on the bf16 matrix pipe (round 3): stride-1 3x3 / pad 1 convolutions, forward and data gradient, NHWC fp32
// storage in and out, every operand split EXACTLY into three bf16 terms, six v_mfma_f32_32x32x16_bf16 products per operand pair,
// fp32 accumulation (gfx950).
//
// Replaces, for the fp32 configurations (BASELINE cfg 1 / 2 / 4: the headline), conv_igemm.hip's v_mfma_f32_32x32x2_f32 path on
// the layers torch.nn.functional.conv2d / its autograd reach from smp.Unet.forward (reference src/models/train.py:341,343).
//
// Why: CDNA4's fp32 MFMA peak is 157 TFLOP/s, its dense bf16 peak 2.5 PFLOP/s -- a 32x32x16 block of products costs 512 cycles
// of a SIMD's matrix pipe as fp32 and 32 as bf16.  An fp32 number has a 24-bit significand, a bf16 number 8 bits with the same
// exponent range, so  x = x0 + x1 + x2  with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)  holds EXACTLY (both
// subtractions are exact in fp32; outside overflow / the subnormal range).  Then
//   a * b = sum_{i,j} a_i b_j ,   |a_i b_j| <= 2^(-8 (i + j)) |a b| ,
// and the six products with i + j <= 2 leave out a1 b2 + a2 b1 + a2 b2 <= 2^-23 |a b| (1 + 2^-9): ONE fp32 unit in the last
// place of the product, below what the fp32 accumulation rounds away at every step anyway (tests/test_gpu_f32x3.py measures
// this kernel and the fp32-MFMA kernel against an f64 convolution: same error).  Six bf16 MFMAs = 192 cycles against 512: the
// matrix pipe does the same fp32-grade arithmetic 2.7 times faster, and the 3x3 layers stop being bound by it alone.
//
// Structure (the halo scheme of conv_halo_bf16.hip, re-balanced for 6x the MFMA work per staged byte):
//   * a block owns TH x 32 output pixels x 32 WN channels; the (TH+2) x 34 input halo of a 16-channel chunk is loaded ONCE as
//     fp32, split in registers and written to three bf16 planes in LDS (32-byte pixel rows, the two 16-byte K halves swapped
//     on every other group of 8 pixels: conflict-free ds_read_b128 without padding);
//   * the weights arrive pre-split and pre-packed in MFMA-fragment order (udaseg_pack_frag_batched_f32x3: three planes of the
//     bf16 kernel's packing) and pass through a double-buffered LDS stage one (dx) group ahead of their use;
//   * operands swapped as in the bf16 kernel (weights = MFMA A, pixels = MFMA B): an accumulator lane holds 4 consecutive
//     channels of one pixel and stores them as one 16-byte fp32 piece straight from registers;
//   * the fused decoder input, the split data gradient, BatchNorm statistics of the output, BatchNorm-backward sums of the
//     producer (data gradient) and accumulation into the destination are the bf16 kernel's options, on fp32 tensors.
// Two kernels share this scheme and one epilogue (conv_halo_f32x3_epilogue.inc): conv3x3_f32x3_kernel, where every wave loads,
// splits, stages and multiplies in turn (<= 32 produced channels: short K loops, three blocks per CU), and
// conv3x3_f32x3_ws_kernel, where four waves only issue MFMAs while four others prepare the next chunk (>= 64 produced channels).
// (Measured and not kept, tools/micro/conv_halo_f32x3_chunk_staged.hip.txt: all 27 weight fragments of a chunk staged at once, one
// chunk ahead like the halo, two barriers per chunk instead of four, next halo row read before the current row's MFMAs: 842 against
// 856 images/s -- the larger LDS block costs the 32-channel configuration its third resident block.)
// inf / values within 2^-8 of FLT_MAX: x0 rounds to inf and the remainders become NaN -- such activations are already lost.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "halo_common.h"

namespace udaseg {

struct F3Args {
  const float* x;       // gathered tensor (forward: input; data gradient: dy), or the half-resolution `a` of a fused decoder input
  const float* x2;      // fused decoder input: the skip tensor (channels [up_ca, ci)), or nullptr when the input is nearest_x2(a)
  const void* wf;       // [3][frag_elems] bf16: the three split planes, each packed[nb][dx][k16][dy][lane][8]
  const float* bias;
  float* y;             // produced tensor; split_n > 0: channels [0, split_n) here, the rest in y2
  float* y2;
  int n, h, w, ci, co;  // ci = gathered channels, co = produced channels of THIS launch
  int up_ca, split_n, accumulate, act;
  float slope;
  double* stats;        // [R][2][co] f64: BatchNorm statistics of the output, or the bnb_* sums
  double* sscr;         // launches of > 1024 blocks: f64 partial sums (see HaloArgs::sscr)
  const float* bnb_y;   // data gradient: conv output of the producing conv+BN+activation layer -> its BatchNorm-backward sums
  const float* bnb_mean;
  const float* bnb_rstd;
  const float* bnb_gamma;
  const float* bnb_beta;
  int bnb_act;
  float bnb_slope;
  int ntx, nty, ncb, nk16;
  int q1, q3;           // (chunk, dx) groups [q1, q3) run on negated weights and a negated accumulator (halo_common.h f3_negated_groups)
  // x is the raw output of a conv + training-mode BatchNorm + activation layer whose activation was never written (round 4,
  // engine.LazyAct on fp32): the staging applies act(fma(x, in_scale[c], in_shift[c])) -- bn_apply's own fused multiply-add and
  // coefficients, so the value is bit for bit what the stand-alone pass would have stored -- before the split; padding stays zero.
  // Both kernels; a single source (plain, or the half-resolution tensor of an up-sampled input: up_ca == ci).  null: x as it is.
  const float* in_scale;
  const float* in_shift;
  int in_act;
  float in_slope;
  // ... and WRITE it out on the way (wave-specialised kernel only): the loader waves of the blocks of channel block 0 store the
  // transformed pixels of their tile's interior to z_out ([n][h][w][ci], the layout of x) -- bit for bit what udaseg_bn_apply would
  // have stored.  The consumer's weight gradient then reads z_out like any activation; what is saved is bn_apply's launch and its
  // read of x.  null: nothing is written.
  float* z_out;
  unsigned z_bytes;
  unsigned x_bytes, x2_bytes, w_plane_bytes, y_bytes, y2_bytes, bnb_bytes;
  unsigned long long* timeline;   // diagnosis (udaseg_debug_set_timeline; stamped twin of the wave-specialised kernel only)
};
extern unsigned long long* g_timeline;      // conv_igemm.hip
extern int g_timeline_blocks;

template <int WM, int WN, int RPW>
struct F3Cfg {
  static constexpr int NT = 64 * WM * WN;
  static constexpr int TH = WM * RPW, TW = 32;
  static constexpr int HR = TH + 2, HWD = TW + 2;
  static constexpr int PLANE = HR * HWD * 32;           // one bf16 plane of a 16-channel halo chunk
  static constexpr int LDS_HALO = 3 * PLANE;
  static constexpr int NPIECE = HR * HWD * 2;           // (halo pixel, 8-channel octet)
  static constexpr int NI = (NPIECE + NT - 1) / NT;
  static constexpr int S = RPW + 2;                     // halo rows a wave reads per (dx) group
  static constexpr int NW = WM * WN;
  static constexpr int NFG = WN * 9;                    // weight fragments of a (chunk, dx) group: [wn][dy][plane]
  static constexpr int NWI = (NFG + NW - 1) / NW;
  static constexpr int LDS_WBUF = NFG * 1024;
  static constexpr int LDS = LDS_HALO + 2 * LDS_WBUF;
  static_assert(LDS_HALO >= 2 * NW * 32 * 4, "reduction scratch fits the halo region");
};

template <int WM, int WN, int RPW, bool XF = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv3x3_f32x3_kernel(const F3Args a) {
  using C = F3Cfg<WM, WN, RPW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lp = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  // ---- block -> (image, tile, channel block); XCD-aware as in conv_halo_bf16_kernel
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int cb = bid % a.ncb;
  int t = bid / a.ncb;
  const int tx = t % a.ntx;
  t /= a.ntx;
  const int ty = t % a.nty;
  const int img = t / a.nty;
  const int y0 = ty * C::TH, x0 = tx * C::TW;
  const int H = a.h, W = a.w;

  // ---- staging slots: piece = tid + i * NT -> (halo pixel, octet of the 16-channel chunk)
  const int oct = tid & 1;
  unsigned voff[C::NI], voff2[C::NI], soffl[C::NI];
  const bool UPC = a.up_ca > 0;
  const int cx = UPC ? a.up_ca : a.ci, cx2 = a.ci - a.up_ca;
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    const int piece = tid + i * C::NT;
    const int pix = piece >> 1;
    const int hy = pix / C::HWD, hx = pix - hy * C::HWD;
    const int iy = y0 + hy - 1, ix = x0 + hx - 1;
    const bool ok = piece < C::NPIECE && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    if (UPC) {
      voff[i] = ok ? (unsigned)((((img * (H >> 1) + (iy >> 1)) * (W >> 1) + (ix >> 1)) * cx + oct * 8) * 4) : 0x80000000u;
      voff2[i] = ok ? (unsigned)((((img * H + iy) * W + ix) * cx2 + oct * 8) * 4) : 0x80000000u;
    } else {
      voff[i] = ok ? (unsigned)((((img * H + iy) * W + ix) * cx + oct * 8) * 4) : 0x80000000u;
      voff2[i] = 0x80000000u;
    }
    // LDS slot of the piece: 32-byte pixel rows, the K halves swapped where bit 3 of the halo column is set
    soffl[i] = (unsigned)(pix * 32 + ((oct ^ ((hx >> 3) & 1)) * 16));
  }
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(UPC && a.x2 ? a.x2 : a.x), 0,
                                                                   (int)(UPC && a.x2 ? a.x2_bytes : 0u), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wf), 0, (int)(3u * a.w_plane_bytes), 0x00020000);

  // ---- this wave's channel block
  const int nblocks32 = (a.co + 31) >> 5;
  int nb = cb * WN + wn;
  const bool wave_live = nb < nblocks32;
  if (!wave_live) nb = 0;

  // ---- weight staging: the WN x 3 (dy) x 3 (plane) fragments of a (chunk, dx) group, 1 KB each, loaded by the waves round-robin
  const int frag_per_nb = 3 * a.nk16 * 3;          // fragments of one 32-channel block in a plane
  int wbase[C::NWI];
  const unsigned wlane16 = (unsigned)lane * 16u;
#pragma unroll
  for (int i = 0; i < C::NWI; ++i) {
    const int q = wave + C::NW * i;                // slot of the group: ((wq * 3 + dy) * 3 + plane)
    const int wq = q / 9, rem = q - wq * 9;
    const int dy = rem / 3, pl = rem - dy * 3;
    const int nbq = cb * WN + wq;
    const bool live = q < C::NFG && nbq < nblocks32;
    wbase[i] = live ? (int)(pl * a.w_plane_bytes) + (nbq * frag_per_nb + dy) * 1024 : -1;
  }
  char* wlds = smem + C::LDS_HALO;
  const int wrd = (wn * 9) * 1024 + lane * 16;

  // pixel fragment address per dx: lane pixel lp of a row, K half lh (swapped where bit 3 of the halo column is set)
  int poff[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int hx = lp + dx;
    poff[dx] = (wm * RPW * C::HWD + hx) * 32 + ((lh ^ ((hx >> 3) & 1)) * 16);
  }

  f32x16 acc[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[r][v] = 0.f;

  const int nchunk = (a.ci + 15) >> 4;
  u32x4 stage[C::NI][2];
  auto load_chunk = [&](int c) {
    const int cbeg = c * 16;
    const bool second = UPC && cbeg >= a.up_ca;
    const int soff = (second ? cbeg - a.up_ca : cbeg) * 4;
    const unsigned kill = (cbeg + oct * 8 < a.ci) ? 0u : 0x80000000u;     // channel tail: zeros, not the next pixel
    if (second) {
#pragma unroll
      for (int i = 0; i < C::NI; ++i) {
        stage[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x2, (int)(voff2[i] | kill), soff, 0);
        stage[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_x2, (int)(voff2[i] | kill), soff + 16, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < C::NI; ++i) {
        stage[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(voff[i] | kill), soff, 0);
        stage[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(voff[i] | kill), soff + 16, 0);
      }
    }
  };
  // XF: the gathered tensor is an unwritten BatchNorm activation (F3Args::in_scale) -- an instantiation of its own, the plain
  // launches carry none of it
  const bool xf_relu = a.in_act == UDASEG_ACT_LEAKY && a.in_slope == 0.f;
  auto store_chunk = [&](int c) {
    f32x4 sc0 = {0.f, 0.f, 0.f, 0.f}, sc1 = sc0, sh0 = sc0, sh1 = sc0;
    if constexpr (XF) {                                      // this thread's 8 channels of chunk c (its octet is fixed)
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
          const bool inside = voff[i] != 0x80000000u;        // zero padding is padding of the ACTIVATION: stays zero
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
  u32x4 wstage[C::NWI];
  auto load_w = [&](int c, int dx) {
    const int soff = (dx * a.nk16 + c) * 3 * 1024;
#pragma unroll
    for (int i = 0; i < C::NWI; ++i)
      wstage[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wbase[i] < 0 ? 0x80000000u : wlane16), wbase[i] < 0 ? 0 : wbase[i] + soff, 0);
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < C::NWI; ++i) {
      const int q = wave + C::NW * i;
      if (i < C::NWI - 1 || q < C::NFG) *reinterpret_cast<u32x4*>(wlds + buf * C::LDS_WBUF + q * 1024 + lane * 16) = wstage[i];
    }
  };

  load_w(0, 0);
  load_chunk(0);
  int gbuf = 0;
  for (int c = 0; c < nchunk; ++c) {
    store_chunk(c);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      store_w(gbuf);
      __syncthreads();               // this group's weights (and, at dx == 0, the chunk's halo) are visible
      if (3 * c + dx == a.q1 || 3 * c + dx == a.q3) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[r] = -acc[r];
      }
      if (dx < 2) load_w(c, dx + 1);
      else if (c + 1 < nchunk) load_w(c + 1, 0);
      if (dx == 0 && c + 1 < nchunk) load_chunk(c + 1);
      const char* wb = wlds + gbuf * C::LDS_WBUF + wrd;
      u32x4 bf[3][3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bf[dy][pl] = *reinterpret_cast<const u32x4*>(wb + (dy * 3 + pl) * 1024);
      // the halo row of step s + 1 is requested before the MFMAs of step s
      u32x4 pf[2][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) pf[0][pl] = *reinterpret_cast<const u32x4*>(smem + pl * C::PLANE + poff[dx]);
#pragma unroll
      for (int s = 0; s < C::S; ++s) {
        if (s + 1 < C::S) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            pf[(s + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(smem + pl * C::PLANE + poff[dx] + (s + 1) * C::HWD * 32);
        }
        // smallest terms first (weight piece i x pixel piece j, i + j <= 2); consecutive MFMAs go to DIFFERENT accumulators (the
        // up to three output rows this halo row feeds)
#pragma unroll
        for (int ij = 2; ij >= 0; --ij)
#pragma unroll
          for (int i = 0; i <= ij; ++i)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
              const int r = s - dy;
              if (r >= 0 && r < RPW)
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[dy][i]),
                                                                 __builtin_bit_cast(bf16x8, pf[s & 1][ij - i]), acc[r], 0, 0, 0);
            }
      }
      gbuf ^= 1;
    }
    __syncthreads();                 // every wave is done with the halo before the next chunk overwrites it
  }

  constexpr int NW_EPI = WM * WN;
  constexpr int EPI_TW = 32;
#include "conv_halo_f32x3_epilogue.inc"
}

// ------------------------------------------------------------------------------------------------ wave-specialised form
// Timing-only probes of the kernel above (tools/f3_probe.py, 64 -> 64 at 128^2: 67 us) put its MFMA phase at ~30 us and
// everything else -- fp32 loads, the operand split, LDS stores, weight staging, barriers, epilogue -- at ~37 us, and the two
// hardly overlap: a wave does them in turn and the waves of a block are in step.  Here the roles are split between waves:
//   waves 0-3 (2 x 2 over the tile): LDS fragment reads and MFMAs, nothing else until the epilogue;
//   waves 4-7: request the fp32 halo of chunk c + 2 and the weights of group G + 2, split chunk c + 1 into the OTHER halo buffer
//              and store the weights of group G + 1 into the other weight buffer -- all while the MFMA waves work on group G.
// One barrier per (chunk, dx) group for all eight waves.  LDS: two halo buffers (3 planes each) + two weight buffers.
// (The probes were uniform run-time branches in a diagnostic build, since removed: a branch around every MFMA cost the one-role
// kernel 8 %.  Measured and not kept, tools/micro/conv_halo_f32x3_ws_persistent.hip.txt: blocks that walk several spatial tiles
// with the loader running on across tile boundaries, statistics kept in registers until the block ends, weights requested three
// groups ahead -- 59 / 57 / 60 us on the 64 / 128 / 256-channel layers against 58 / 53 / 59 for this form, whose two co-resident
// blocks per CU already cover each other's prologue and epilogue.)
template <int WM_, int WN_, int RPW, int TW_ = 32>
struct F3WsCfg {
  static constexpr int WM = WM_, WN = WN_;
  static_assert(WM_ * WN_ == 4, "four MFMA waves");
  static_assert(TW_ == 32 || TW_ == 16, "32-pixel rows, or 16-pixel rows (the 32 pixels of an MFMA block are then TWO image rows)");
  static constexpr int NT = 512, NLD = 256;               // threads; loader threads
  static constexpr int TW = TW_, RL = 32 / TW_;           // image rows per 32-pixel MFMA block
  static constexpr int TH = WM * RPW * RL;
  static constexpr int NJ = RL * (RPW - 1) + 3;            // fragment start rows a wave reads per (dx) group (the last one is tap row 2 of its last MFMA block)
  static constexpr bool DEEP = TW_ == 16;                  // loader roles split, weights requested three groups ahead (see the kernel)
  static constexpr int HR = TH + 2, HWD = TW + 2;
  // LDS pitch of a halo row, in pixels (32 bytes each).  16-pixel rows: a fragment spans TWO halo rows, and ds_read_b128 serves
  // lanes {0-3, 12-15, 20-27} in one cycle -- columns 0-3 / 12-15 of row r with columns 4-11 of row r + 1: conflict-free only if the
  // rows are a multiple of 256 bytes apart (24 pixels).  With the natural 18-pixel pitch (576 bytes = +4 slots of 16 bytes) columns
  // 4-7 of row r + 1 landed on the banks of columns 0-1 / 6-7 of row r: PMC SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.24 on the
  // r18 layer4 launches of round 4 (profiles/r04_pmc_sq.txt).
  static constexpr int HWP = TW == 16 ? 24 : HWD;
  static constexpr int PLANE = HR * HWP * 32;
  static constexpr int LDS_HALO = 3 * PLANE;
  static constexpr int NPIECE = HR * HWD * 2;
  static constexpr int NI = (NPIECE + NLD - 1) / NLD;
  static constexpr int S = RPW + 2;
  static constexpr int NFG = WN * 9;
  static constexpr int NWI = (NFG + 3) / 4;               // fragments a loader wave stages per group
  static constexpr int LDS_WBUF = NFG * 1024;
  static constexpr int LDS = 2 * LDS_HALO + 2 * LDS_WBUF;
};

template <int WM, int WN, int RPW, bool TL = false, int TW = 32, bool XF = false>
__global__ __launch_bounds__(512, 1) void conv3x3_f32x3_ws_kernel(const F3Args a) {
  using C = F3WsCfg<WM, WN, RPW, TW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wlds = smem + 2 * C::LDS_HALO;
  // TL: shader-clock stamps of one MFMA wave and one loader wave per block (tools/f3_timeline.py)
  unsigned long long tl_e = 0, tl_first = 0, tl_bar = 0, tl_work = 0, tl_kend = 0, tl_w0 = 0, tl_t = 0;
  if constexpr (TL) {
    tl_w0 = __builtin_amdgcn_s_memrealtime();
    tl_e = __builtin_amdgcn_s_memtime();
  }

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int mw = wave & 3;                         // index inside the role
  const int lp = lane & 31, lh = lane >> 5;
  const int wm = mw / WN, wn = mw % WN;

  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int cb = bid % a.ncb;
  int t = bid / a.ncb;
  const int tx = t % a.ntx;
  t /= a.ntx;
  const int ty = t % a.nty;
  const int img = t / a.nty;
  const int y0 = ty * C::TH, x0 = tx * C::TW;
  const int H = a.h, W = a.w;
  const int nblocks32 = (a.co + 31) >> 5;
  const int nchunk = (a.ci + 15) >> 4, NG = 3 * nchunk;

  int nb = cb * WN + wn;
  const bool wave_live = nb < nblocks32;
  if (!wave_live) nb = 0;
  f32x16 acc[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[r][v] = 0.f;

  if (loader) {
    // DEEP (F3WsCfg::DEEP: the 16-pixel-row tiles, one block per CU, 18 MFMAs per group): the two jobs are split between the loader
    // waves -- waves 4-5 stage weights only and request a group's fragments THREE groups ahead into one of two register sets (a weight
    // fragment comes from L2 / the Infinity Cache: ~1200 cycles, more than one group lasts; with one set the wave sat in s_waitcnt
    // for most of every group: 1093 of 1262 cycles, profiles/r05_f3_deep.txt), waves 6-7 stage the halo.  Every load of the weight
    // waves is issued unconditionally (past the last group: out of range, returns zeros) so that the wait count in front of each
    // LDS store is a compile-time number of younger loads.
    constexpr bool DEEP = C::DEEP;
    constexpr int NLDX = DEEP ? 128 : C::NLD;
    constexpr int NIX = (C::NPIECE + NLDX - 1) / NLDX;
    constexpr int WSTEP = DEEP ? 2 : 4;
    constexpr int NWIX = (C::NFG + WSTEP - 1) / WSTEP;
    const bool h_role = !DEEP || wave >= 6;
    const int lt = tid - (DEEP ? 384 : 256);
    const int wq0 = DEEP ? (wave - 4) : mw;
    const int oct = lt & 1;
    unsigned voff[NIX], voff2[NIX], soffl[NIX];
    unsigned zmask = 0;
    const bool UPC = a.up_ca > 0;
    const int cx = UPC ? a.up_ca : a.ci, cx2 = a.ci - a.up_ca;
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
      const int piece = lt + i * NLDX;
      const int pix = piece >> 1;
      const int hy = pix / C::HWD, hx = pix - hy * C::HWD;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      const bool ok = h_role && piece < C::NPIECE && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      if (UPC) {
        voff[i] = ok ? (unsigned)((((img * (H >> 1) + (iy >> 1)) * (W >> 1) + (ix >> 1)) * cx + oct * 8) * 4) : 0x80000000u;
        voff2[i] = ok ? (unsigned)((((img * H + iy) * W + ix) * cx2 + oct * 8) * 4) : 0x80000000u;
      } else {
        voff[i] = ok ? (unsigned)((((img * H + iy) * W + ix) * cx + oct * 8) * 4) : 0x80000000u;
        voff2[i] = 0x80000000u;
      }
      soffl[i] = (unsigned)((hy * C::HWP + hx) * 32 + ((oct ^ ((hx >> 3) & 1)) * 16));
      if constexpr (XF) {
        if (ok && hy >= 1 && hy <= C::TH && hx >= 1 && hx <= C::TW) zmask |= 1u << i;       // the tile's own pixels (F3Args::z_out)
      }
    }
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(UPC && a.x2 ? a.x2 : a.x), 0,
                                                                     (int)(UPC && a.x2 ? a.x2_bytes : 0u), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wf), 0, (int)(3u * a.w_plane_bytes), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(XF && a.z_out ? a.z_out : const_cast<float*>(a.x), 0,
                                                                    (int)(XF && a.z_out ? a.z_bytes : 0u), 0x00020000);
    const bool zwrite = XF && a.z_out != nullptr && cb == 0 && !UPC;
    const int frag_per_nb = 3 * a.nk16 * 3;
    int wbase[NWIX];
    const unsigned wlane16 = (unsigned)lane * 16u;
#pragma unroll
    for (int i = 0; i < NWIX; ++i) {
      const int q = wq0 + WSTEP * i;               // slot of the group: ((wq * 3 + dy) * 3 + plane)
      const int wq = q / 9, rem = q - wq * 9;
      const int dy = rem / 3, pl = rem - dy * 3;
      const int nbq = cb * WN + wq;
      const bool live = q < C::NFG && nbq < nblocks32;
      wbase[i] = live ? (int)(pl * a.w_plane_bytes) + (nbq * frag_per_nb + dy) * 1024 : -1;
    }
    u32x4 stage[NIX][2], wstage[DEEP ? 2 : 1][NWIX];
    auto load_chunk = [&](int c) {
      const int cbeg = c * 16;
      const bool second = UPC && cbeg >= a.up_ca;
      const int soff = (second ? cbeg - a.up_ca : cbeg) * 4;
      const unsigned kill = (cbeg + oct * 8 < a.ci) ? 0u : 0x80000000u;
      if (second) {
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
          stage[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x2, (int)(voff2[i] | kill), soff, 0);
          stage[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_x2, (int)(voff2[i] | kill), soff + 16, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
          stage[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(voff[i] | kill), soff, 0);
          stage[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(voff[i] | kill), soff + 16, 0);
        }
      }
    };
    const bool xf_relu = a.in_act == UDASEG_ACT_LEAKY && a.in_slope == 0.f;      // XF: see conv3x3_f32x3_kernel
    auto store_chunk = [&](int buf, int c) {
      char* hb = smem + buf * C::LDS_HALO;
      f32x4 sc0 = {0.f, 0.f, 0.f, 0.f}, sc1 = sc0, sh0 = sc0, sh1 = sc0;
      if constexpr (XF) {                                    // this thread's 8 channels of chunk c (see conv3x3_f32x3_kernel)
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
      for (int i = 0; i < NIX; ++i) {
        if (i < NIX - 1 || lt + i * NLDX < C::NPIECE) {
          if constexpr (XF) {
            const bool inside = voff[i] != 0x80000000u;
            f32x4 l = __builtin_bit_cast(f32x4, stage[i][0]), h = __builtin_bit_cast(f32x4, stage[i][1]);
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
            stage[i][0] = __builtin_bit_cast(u32x4, l);
            stage[i][1] = __builtin_bit_cast(u32x4, h);
            if (zwrite && ((zmask >> i) & 1u) && c * 16 + oct * 8 < a.ci) {       // write-through of the activation
              __builtin_amdgcn_raw_buffer_store_b128(stage[i][0], rs_z, (int)voff[i], c * 64, 0);
              __builtin_amdgcn_raw_buffer_store_b128(stage[i][1], rs_z, (int)voff[i], c * 64 + 16, 0);
            }
          }
          u32x4 p0, p1, p2;
          split3(stage[i][0], stage[i][1], p0, p1, p2);
          *reinterpret_cast<u32x4*>(hb + soffl[i]) = p0;
          *reinterpret_cast<u32x4*>(hb + C::PLANE + soffl[i]) = p1;
          *reinterpret_cast<u32x4*>(hb + 2 * C::PLANE + soffl[i]) = p2;
        }
      }
    };
    // SET: the register set (compile-time); G >= NG: out of range, nothing is fetched (the load still counts)
    auto load_w = [&](int G, auto SET) {
      const int c = G / 3, dx = G - 3 * c;
      const int soff = (dx * a.nk16 + c) * 3 * 1024;
      const bool dead = G >= NG;
#pragma unroll
      for (int i = 0; i < NWIX; ++i)
        wstage[SET.value][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)((wbase[i] < 0 || dead) ? 0x80000000u : wlane16),
                                                                     (wbase[i] < 0 || dead) ? 0 : wbase[i] + soff, 0);
    };
    auto store_w = [&](int buf, auto SET) {
#pragma unroll
      for (int i = 0; i < NWIX; ++i) {
        const int q = wq0 + WSTEP * i;
        if (i < NWIX - 1 || q < C::NFG) *reinterpret_cast<u32x4*>(wlds + buf * C::LDS_WBUF + q * 1024 + lane * 16) = wstage[SET.value][i];
      }
    };
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, DEEP ? 1 : 0> S1{};

    if constexpr (DEEP) {
     if (!h_role) {
      // weight waves: group G's fragments live in register set G & 1 and in LDS weight buffer G & 1
      // (scheduling barriers: the issue ORDER of the three requests is what the wait counts at the loop head are derived from)
      load_w(0, S0);
      __builtin_amdgcn_sched_barrier(0);
      load_w(1, S1);
      __builtin_amdgcn_sched_barrier(0);
      store_w(0, S0);
      __builtin_amdgcn_sched_barrier(0);
      load_w(2, S0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                             // group 0 is staged
      if constexpr (TL) tl_first = tl_t = __builtin_amdgcn_s_memtime();
      auto wgroup = [&](int G, auto NEXT) {        // NEXT: set / buffer of group G + 1 (last read by the MFMA waves in group G - 1)
        if (G + 1 < NG) store_w(NEXT.value, NEXT);
        load_w(G + 3, NEXT);
        if constexpr (TL) {
          const unsigned long long t = __builtin_amdgcn_s_memtime();
          tl_work += t - tl_t;
          tl_t = t;
        }
        __syncthreads();
        if constexpr (TL) {
          const unsigned long long t = __builtin_amdgcn_s_memtime();
          tl_bar += t - tl_t;
          tl_t = t;
        }
      };
      // both halves unconditional inside the loop: a path that skips one would make the younger-load count at the loop head
      // unknown, and the compiler then waits for everything (vmcnt(0)) in front of the stores -- the prefetch distance would be gone
      int G = 0;
      for (; G + 1 < NG; G += 2) {
        wgroup(G, S1);
        wgroup(G + 1, S0);
      }
      if (G < NG) wgroup(G, S1);
     } else {
      // halo waves
      load_chunk(0);
      store_chunk(0, 0);
      if (nchunk > 1) load_chunk(1);
      __syncthreads();                             // group 0 is staged
      for (int G = 0; G < NG; ++G) {
        const int c = G / 3, dx = G - 3 * c;
        if (dx == 1 && c + 1 < nchunk) {
          store_chunk((c + 1) & 1, c + 1);         // last read in chunk c - 1
          if (c + 2 < nchunk) load_chunk(c + 2);
        }
        __syncthreads();
      }
     }
    } else {
      load_chunk(0);
      load_w(0, S0);
      store_chunk(0, 0);
      store_w(0, S0);
      if (nchunk > 1) load_chunk(1);
      if (NG > 1) load_w(1, S0);
      __syncthreads();                               // group 0 is staged
      if constexpr (TL) tl_first = tl_t = __builtin_amdgcn_s_memtime();
      for (int G = 0; G < NG; ++G) {
        const int c = G / 3, dx = G - 3 * c;
        if (G + 1 < NG) {
          store_w((G + 1) & 1, S0);                  // last read by the MFMA waves in group G - 1
          if (G + 2 < NG) load_w(G + 2, S0);
        }
        if (dx == 1 && c + 1 < nchunk) {
          store_chunk((c + 1) & 1, c + 1);           // last read in chunk c - 1
          if (c + 2 < nchunk) load_chunk(c + 2);
        }
        if constexpr (TL) {
          const unsigned long long t = __builtin_amdgcn_s_memtime();
          tl_work += t - tl_t;
          tl_t = t;
        }
        __syncthreads();
        if constexpr (TL) {
          const unsigned long long t = __builtin_amdgcn_s_memtime();
          tl_bar += t - tl_t;
          tl_t = t;
        }
      }
    }
    if constexpr (TL) tl_kend = tl_t;
  } else {
    // pixel fragment address per dx: lane pixel lp of a row, K half lh (swapped where bit 3 of the halo column is set)
    // (16-pixel rows: lanes 0-15 / 16-31 of a fragment are two consecutive image rows; a fragment may start at ANY halo row)
    int poff[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = lp % C::TW + dx;
      poff[dx] = ((wm * RPW * C::RL + lp / C::TW) * C::HWP + hx) * 32 + ((lh ^ ((hx >> 3) & 1)) * 16);
    }
    const int wrd = (wn * 9) * 1024 + lane * 16;
    __syncthreads();                               // group 0 is staged
    if constexpr (TL) tl_first = tl_t = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < nchunk; ++c) {
      const char* hb = smem + (c & 1) * C::LDS_HALO;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const char* wb = wlds + ((3 * c + dx) & 1) * C::LDS_WBUF + wrd;
        if (3 * c + dx == a.q1 || 3 * c + dx == a.q3) {
#pragma unroll
          for (int r = 0; r < RPW; ++r) acc[r] = -acc[r];
        }
        u32x4 bf[3][3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) bf[dy][pl] = *reinterpret_cast<const u32x4*>(wb + (dy * 3 + pl) * 1024);
        u32x4 pf[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) pf[0][pl] = *reinterpret_cast<const u32x4*>(hb + pl * C::PLANE + poff[dx]);
#pragma unroll
        for (int s = 0; s < C::NJ; ++s) {
          if (s + 1 < C::NJ) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              pf[(s + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(hb + pl * C::PLANE + poff[dx] + (s + 1) * C::HWP * 32);
          }
#pragma unroll
          for (int ij = 2; ij >= 0; --ij)
#pragma unroll
            for (int i = 0; i <= ij; ++i)
#pragma unroll
              for (int dy = 0; dy < 3; ++dy) {
                const int r = (s - dy) / C::RL;        // the fragment that starts at halo row s is tap row dy of MFMA block r
                if (s - dy >= 0 && (s - dy) % C::RL == 0 && r < RPW)
                  acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[dy][i]),
                                                                   __builtin_bit_cast(bf16x8, pf[s & 1][ij - i]), acc[r], 0, 0, 0);
              }
        }
        if constexpr (TL) {
          const unsigned long long t = __builtin_amdgcn_s_memtime();
          tl_work += t - tl_t;
          tl_t = t;
        }
        __syncthreads();                           // group G + 1 is staged; this group's buffers may be rewritten
        if constexpr (TL) {
          const unsigned long long t = __builtin_amdgcn_s_memtime();
          tl_bar += t - tl_t;
          tl_t = t;
        }
      }
    }
    if constexpr (TL) tl_kend = tl_t;
  }
  if (loader) {                                    // the statistics reduction has one block barrier
    if (a.stats != nullptr) __syncthreads();
    if constexpr (TL) {
      if (a.timeline != nullptr && tid == 256) {
        unsigned long long* t = a.timeline + ((size_t)blockIdx.x * 2 + 1) * 8;
        t[0] = tl_first - tl_e; t[1] = tl_work; t[2] = tl_bar; t[3] = 0;
        t[4] = __builtin_amdgcn_s_memtime() - tl_e; t[5] = __builtin_amdgcn_s_memrealtime() - tl_w0; t[6] = (unsigned long long)NG; t[7] = 1;
      }
    }
    return;
  }
  constexpr int NW_EPI = WM * WN;
  constexpr int EPI_TW = C::TW;
  {
    const int wave = mw;                           // the epilogue's statistics slot: this wave's index among the MFMA waves
#include "conv_halo_f32x3_epilogue.inc"
  }
  if constexpr (TL) {
    if (a.timeline != nullptr && tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the epilogue's stores have left
      unsigned long long* t = a.timeline + (size_t)blockIdx.x * 2 * 8;
      t[0] = tl_first - tl_e; t[1] = tl_work; t[2] = tl_bar; t[3] = __builtin_amdgcn_s_memtime() - tl_kend;
      t[4] = __builtin_amdgcn_s_memtime() - tl_e; t[5] = __builtin_amdgcn_s_memrealtime() - tl_w0; t[6] = (unsigned long long)NG; t[7] = 1;
    }
  }
}

// ------------------------------------------------------------------------------------------------ fragment packing
// Three planes of conv_halo_bf16.hip's packing, from fp32 sources:  plane[p][nb][dx][k16][dy][lane][j] = piece p of
// Wsrc[n = 32 nb + (lane & 31)][tap][k = 16 k16 + 8 (lane >> 5) + j].  Wsrc is [N][9][K] fp32: the OHWI weights (forward) or the
// dgrad packing [ci][taps][co] with the window flipped (data gradient).
// table row (int32 x 6): {mode, src element offset, dst element offset (plane 0), N, K, KS = 3}; plane stride = frag_elems(N, K, 3)
// The fragments of the (k16, dx) groups [q1, q3) carry -W (halo_common.h f3_negated_groups).
__global__ void pack_frag_batched_f32x3_kernel(const float* __restrict__ w32, const float* __restrict__ wt32,
                                               __bf16* __restrict__ packed, const int* __restrict__ table, int signs) {
  const int* e = table + 6 * blockIdx.y;
  const int mode = e[0], N = e[3], K = e[4], KS = e[5];
  const float* src = (mode ? wt32 : w32) + e[1];
  __bf16* dst = packed + e[2];
  const int T = KS * KS, nb = (N + 31) >> 5, nk16 = (K + 15) >> 4;
  const long long total = (long long)nb * KS * nk16 * KS * 64;     // one 16-byte item per (fragment, lane) per plane
  const long long plane = total * 8;
  int q1 = 0, q3 = 0;
  if (signs) f3_negated_groups(nk16, q1, q3);
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
    u32x4 lo = {0u, 0u, 0u, 0u}, hi = lo;
    if (n < N && k0 < K) {
      const float* s = src + ((size_t)n * T + tap) * K + k0;
      lo = *reinterpret_cast<const u32x4*>(s);                      // K is a multiple of 4 (fp32 channel padding)
      if (k0 + 4 < K) hi = *reinterpret_cast<const u32x4*>(s + 4);
    }
    u32x4 p0, p1, p2;
    split3(lo, hi, p0, p1, p2);
    if (KS == 3 && 3 * kk + dx >= q1 && 3 * kk + dx < q3) {       // the negated groups of the kernels' K loop
      p0 ^= 0x80008000u;
      p1 ^= 0x80008000u;
      p2 ^= 0x80008000u;
    }
    *reinterpret_cast<u32x4*>(dst + i * 8) = p0;
    *reinterpret_cast<u32x4*>(dst + plane + i * 8) = p1;
    *reinterpret_cast<u32x4*>(dst + 2 * plane + i * 8) = p2;
  }
}

// ------------------------------------------------------------------------------------------------- host side
template <int WM, int WN, int RPW, int TW = 32>
static int launch_f3_ws_t(F3Args a, hipStream_t s, double flops) {
  using C = F3WsCfg<WM, WN, RPW, TW>;
  auto kern = conv3x3_f32x3_ws_kernel<WM, WN, RPW, false, TW>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3x3_f32x3_ws)");
    attr_done = true;
  }
  a.ntx = cdiv(a.w, C::TW);
  a.nty = cdiv(a.h, C::TH);
  a.ncb = cdiv(a.co, 32 * C::WN);
  a.nk16 = (a.ci + 15) / 16;
  a.q1 = a.q3 = -1;
  if (f3_signs_on()) f3_negated_groups(a.nk16, a.q1, a.q3);
  const long long blocks = (long long)a.n * a.nty * a.ntx * a.ncb;
  if (blocks <= 0) return UDASEG_OK;
  a.sscr = nullptr;
  if (a.stats != nullptr && blocks > 1024) a.sscr = halo_stats_scratch(a.co, s);
  static std::atomic<int> kid{-1};
  if (kid < 0) {
    char nm[96];
    snprintf(nm, sizeof(nm), "conv3x3_f32x3_ws_kernel<%d, %d, %d, false, %d, false>", WM, WN, RPW, TW);      // the rocprofv3 symbol
    kid = kprof_id(nm);
  }
  hipEvent_t ev = kprof_begin(s);
  if (a.in_scale != nullptr) {          // the instantiation that transforms x while it stages (F3Args::in_scale)
    auto kern_xf = conv3x3_f32x3_ws_kernel<WM, WN, RPW, false, TW, true>;
    static std::atomic<bool> xf_attr{false};
    if (!xf_attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern_xf), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
      if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3x3_f32x3_ws, in-staging transform)");
      xf_attr = true;
    }
    static std::atomic<int> kid_xf{-1};             // its own rocprofv3 symbol
    if (kid_xf < 0) {
      char nm[96];
      snprintf(nm, sizeof(nm), "conv3x3_f32x3_ws_kernel<%d, %d, %d, false, %d, true>", WM, WN, RPW, TW);
      kid_xf = kprof_id(nm);
    }
    hipLaunchKernelGGL(kern_xf, dim3((unsigned)blocks), dim3(C::NT), C::LDS, s, a);
    kprof_end(kid_xf, ev, s, flops);
    UDASEG_LAUNCH_CHECK("conv3x3_f32x3_ws (in-staging transform) launch");
    if (a.sscr != nullptr) {
      launch_halo_stats_fold(a.sscr, a.co, a.stats, s);
      UDASEG_LAUNCH_CHECK("halo_stats_fold launch");
    }
    return UDASEG_OK;
  } else if (g_timeline != nullptr && blocks * 16 <= (long long)g_timeline_blocks * 6) {      // stamped twin (diagnosis only)
    auto kern_tl = conv3x3_f32x3_ws_kernel<WM, WN, RPW, true, TW>;
    static std::atomic<bool> tl_attr{false};
    if (!tl_attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern_tl), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
      if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3x3_f32x3_ws timeline twin)");
      tl_attr = true;
    }
    a.timeline = g_timeline;
    hipLaunchKernelGGL(kern_tl, dim3((unsigned)blocks), dim3(C::NT), C::LDS, s, a);
  } else {
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::NT), C::LDS, s, a);
  }
  kprof_end(kid, ev, s, flops);
  UDASEG_LAUNCH_CHECK("conv3x3_f32x3_ws launch");
  if (a.sscr != nullptr) {
    launch_halo_stats_fold(a.sscr, a.co, a.stats, s);
    UDASEG_LAUNCH_CHECK("halo_stats_fold launch");
  }
  return UDASEG_OK;
}

template <int WM, int WN, int RPW>
static int launch_f3_t(F3Args a, hipStream_t s, double flops) {
  using C = F3Cfg<WM, WN, RPW>;
  auto kern = conv3x3_f32x3_kernel<WM, WN, RPW>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done && C::LDS > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3x3_f32x3)");
    attr_done = true;
  }
  a.ntx = cdiv(a.w, C::TW);
  a.nty = cdiv(a.h, C::TH);
  a.ncb = cdiv(a.co, 32 * WN);
  a.nk16 = (a.ci + 15) / 16;
  a.q1 = a.q3 = -1;
  if (f3_signs_on()) f3_negated_groups(a.nk16, a.q1, a.q3);
  const long long blocks = (long long)a.n * a.nty * a.ntx * a.ncb;
  if (blocks <= 0) return UDASEG_OK;
  a.sscr = nullptr;
  if (a.stats != nullptr && blocks > 1024) a.sscr = halo_stats_scratch(a.co, s);
  static std::atomic<int> kid{-1};
  if (kid < 0) {
    char nm[96];
    snprintf(nm, sizeof(nm), "conv3x3_f32x3_kernel<%d, %d, %d, false>", WM, WN, RPW);
    kid = kprof_id(nm);
  }
  hipEvent_t ev = kprof_begin(s);
  if (a.in_scale != nullptr) {
    auto kern_xf = conv3x3_f32x3_kernel<WM, WN, RPW, true>;
    static std::atomic<bool> xf_attr{false};
    if (!xf_attr && C::LDS > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern_xf), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
      if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3x3_f32x3, in-staging transform)");
      xf_attr = true;
    }
    static std::atomic<int> kid_xf{-1};             // its own rocprofv3 symbol
    if (kid_xf < 0) {
      char nm[96];
      snprintf(nm, sizeof(nm), "conv3x3_f32x3_kernel<%d, %d, %d, true>", WM, WN, RPW);
      kid_xf = kprof_id(nm);
    }
    hipLaunchKernelGGL(kern_xf, dim3((unsigned)blocks), dim3(C::NT), C::LDS, s, a);
    kprof_end(kid_xf, ev, s, flops);
  } else {
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::NT), C::LDS, s, a);
    kprof_end(kid, ev, s, flops);
  }
  UDASEG_LAUNCH_CHECK("conv3x3_f32x3 launch");
  if (a.sscr != nullptr) {
    launch_halo_stats_fold(a.sscr, a.co, a.stats, s);
    UDASEG_LAUNCH_CHECK("halo_stats_fold launch");
  }
  return UDASEG_OK;
}

static int f3_enabled() { return f32_halo_enabled() ? 1 : 0; }      // UDASEG_OPT_F32_HALO = 0: every fp32 layer on the fp32-MFMA kernels

// gathered / produced: channel counts of the launch (for a data gradient: co / ci)
static bool f3_applicable(const udaseg_conv_desc* d, int gathered, int produced, int up_ca) {
  if (!f3_enabled()) return false;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1) return false;
  if (gathered % 8 != 0 || produced % 4 != 0) return false;
  if (up_ca > 0 && (up_ca % 16 != 0 || (gathered - up_ca) % 16 != 0 || d->hi % 2 != 0 || d->wi % 2 != 0)) return false;
  const long long px = (long long)d->n * d->hi * d->wi;
  if (px * gathered * 4 >= (1LL << 31) || px * produced * 4 >= (1LL << 31)) return false;     // buffer descriptors: 2 GiB
  return true;
}

// 0: leave the layer to the fp32-MFMA kernels; 1: 8 x 32 pixels x 32 channels per block; 2: 8 x 32 x 64 channels (forced only);
// 3: 4 x 32 pixels x 64 channels
static int f3_choice(int h, int w, int n, int gathered, int produced) {
  const int force = opt_get(UDASEG_OPT_F3_CFG);      // udaseg_f32x3_force_config: one configuration for every launch
  if (force >= 1 && force <= 12) return force;
  if (w <= 16) return produced >= 64 ? 9 : 0;       // 16-pixel-wide images: 4 x 16 pixel tiles, two image rows per MFMA block
  if (w < 32) return 0;
  const long long tiles = (long long)n * cdiv(h, 8) * cdiv(w, 32);
  if (produced <= 32) return 1;
  // 64 and more produced channels: the wave-specialised kernel (one 8-wave block per CU), on 4 x 32 pixel tiles, or 8 x 32 where the
  // K loop is long and the launch still has a block per CU.  Per layer, stand-alone, us (tools/f3_probe.py; configurations
  // 1 / 2 / 3 / 5 / 6): 64 -> 64 at 128^2 70 / 59 / 67 / 64 / 58; 128 -> 128 at 64^2 62 / 66 / 62 / 55 / 53; 256 -> 256 at 32^2
  // 75 / 101 / 74 / 80 / 59; 192 -> 64 at 128^2 179 / 163 / 180 / 158 / 167; 384 -> 128 at 64^2 172 / 179 / 172 / 156 / 160;
  // 768 -> 256 at 32^2 208 / 281 / 207 / 216 / 166 (profiles/r03_f32x3.txt).  Launches too small for that keep 32-channel blocks.
  const int ws = opt_get(UDASEG_OPT_F3_WS);        // 0 (A/B): the one-role kernel everywhere (4 x 32 pixel tiles x 64 channels)
  if (!ws) return tiles * cdiv(produced, 64) >= 256 ? 3 : 1;
  const long long ncb64 = cdiv(produced, 64);
  const long long blocks4 = (long long)n * cdiv(h, 4) * cdiv(w, 32) * ncb64;
  if (blocks4 < 192) return 1;
  return (gathered >= 192 && tiles * ncb64 >= 256) ? 5 : 6;
}

static int launch_f3(F3Args a, hipStream_t s, double flops) {
  int choice = f3_choice(a.h, a.w, a.n, a.ci, a.co);
  if (choice == 0) choice = a.co <= 32 ? 1 : 6;
  if (choice == 1) return launch_f3_t<4, 1, 2>(a, s, flops);
  if (choice == 3) return launch_f3_t<2, 2, 2>(a, s, flops);      // 4 x 32 pixels x 64 channels
  if (choice == 4) return launch_f3_t<4, 2, 4>(a, s, flops);      // 16 x 32 pixels x 64 channels, 8 waves
  if (choice == 5) return launch_f3_ws_t<2, 2, 4>(a, s, flops);   // 8 x 32 pixels x 64 channels, 4 MFMA waves + 4 loader waves
  if (choice == 6) return launch_f3_ws_t<2, 2, 2>(a, s, flops);   // 4 x 32 pixels x 64 channels, same roles
  if (choice == 7) return launch_f3_ws_t<4, 1, 2>(a, s, flops);   // 8 x 32 pixels x 32 channels, same roles
  if (choice == 8) return launch_f3_ws_t<4, 1, 4>(a, s, flops);   // 16 x 32 pixels x 32 channels, same roles
  if (choice == 9) return launch_f3_ws_t<2, 2, 1, 16>(a, s, flops);   // 4 x 16 pixels x 64 channels (16-pixel-wide images), same roles
  if (choice == 10) return launch_f3_ws_t<2, 2, 2, 16>(a, s, flops);  // 8 x 16 x 64
  if (choice == 11) return launch_f3_ws_t<4, 1, 2, 16>(a, s, flops);  // 16 x 16 x 32
  if (choice == 12) return launch_f3_ws_t<4, 1, 1, 16>(a, s, flops);  // 8 x 16 x 32
  return launch_f3_t<2, 2, 4>(a, s, flops);
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_pack_frag_batched_f32x3(const float* w32, const float* wt32, void* packed, const int* table, int entries,
                                              void* stream) {
  UDASEG_CHECK_ARG(packed && table && entries > 0 && (w32 || wt32), "pack_frag_batched_f32x3: NULL pointer / no entries");
  hipLaunchKernelGGL(pack_frag_batched_f32x3_kernel, dim3(256, (unsigned)entries), dim3(256), 0, as_stream(stream), w32, wt32,
                     static_cast<__bf16*>(packed), table, f3_signs_on() ? 1 : 0);
  UDASEG_LAUNCH_CHECK("pack_frag_batched_f32x3 launch");
  return UDASEG_OK;
}

extern "C" int udaseg_f32x3_force_config(int cfg) {
  UDASEG_CHECK_ARG(cfg >= 0 && cfg <= 12, "f32x3_force_config: 0 (heuristic), 1 (8 x 32 px x 32 ch), 2 (8 x 32 x 64), 3 (4 x 32 x 64), 4 (16 x 32 x 64), 5 / 6 (wave-specialised 8 / 4 x 32 x 64), 7 / 8 (wave-specialised 8 / 16 x 32 x 32), 9 (wave-specialised 4 x 16 x 64)");
  return udaseg_set_option(UDASEG_OPT_F3_CFG, cfg);
}

extern "C" int udaseg_conv_f32x3_ok(const udaseg_conv_desc* d, int dgrad, int up_ca) {
  if (!d) return 0;
  return f3_applicable(d, dgrad ? d->co : d->ci, dgrad ? d->ci : d->co, up_ca) ? 1 : 0;
}

extern "C" int udaseg_conv_f32x3_preferred(const udaseg_conv_desc* d, int dgrad, int up_ca) {
  if (!udaseg_conv_f32x3_ok(d, dgrad, up_ca)) return 0;
  return f3_choice(d->hi, d->wi, d->n, dgrad ? d->co : d->ci, dgrad ? d->ci : d->co) != 0 ? 1 : 0;
}

static int f3_common(const udaseg_conv_desc* d, F3Args& a, const char* who) {
  UDASEG_CHECK_ARG(d != nullptr, "%s: conv desc is NULL", who);
  UDASEG_CHECK_ARG(d->n > 0 && d->hi > 0 && d->wi > 0 && d->ho == d->hi && d->wo == d->wi && d->ci > 0 && d->co > 0,
                   "%s: stride-1 'same' convolutions only (hi=%d wi=%d ho=%d wo=%d)", who, d->hi, d->wi, d->ho, d->wo);
  a.n = d->n; a.h = d->hi; a.w = d->wi;
  return UDASEG_OK;
}

extern "C" int udaseg_conv2d_fwd_f32x3(const udaseg_conv_desc* d, const float* x, const float* skip, int up_ca, const void* wfrag3,
                                       const float* bias, float* y, int act, float slope, double* stats, void* stream) {
  F3Args a = {};
  int rc = f3_common(d, a, "conv2d_fwd_f32x3");
  if (rc) return rc;
  UDASEG_CHECK_ARG(x && wfrag3 && y, "conv2d_fwd_f32x3: NULL pointer");
  UDASEG_CHECK_ARG(up_ca >= 0 && up_ca <= d->ci && (up_ca == 0 ? skip == nullptr : (up_ca == d->ci) == (skip == nullptr)),
                   "conv2d_fwd_f32x3: up_ca=%d of ci=%d channels, skip %s", up_ca, d->ci, skip ? "given" : "NULL");
  if (!f3_applicable(d, d->ci, d->co, up_ca)) {
    set_error("conv2d_fwd_f32x3: geometry not supported (ask udaseg_conv_f32x3_ok first)");
    return UDASEG_E_UNSUPPORTED;
  }
  const long long px = (long long)d->n * d->hi * d->wi;
  a.x = x; a.x2 = skip; a.wf = wfrag3; a.bias = bias; a.y = y;
  a.ci = d->ci; a.co = d->co; a.up_ca = up_ca;
  a.act = act; a.slope = slope; a.stats = stats;
  a.x_bytes = (unsigned)(up_ca > 0 ? (long long)d->n * (d->hi / 2) * (d->wi / 2) * up_ca * 4 : px * d->ci * 4);
  a.x2_bytes = (unsigned)(up_ca > 0 ? px * (d->ci - up_ca) * 4 : 0);
  a.w_plane_bytes = (unsigned)(udaseg_frag_elems(d->co, d->ci, 3) * 2);
  a.y_bytes = (unsigned)(px * d->co * 4);
  hipStream_t st = as_stream(stream);
  prof_begin(0, st);
  rc = launch_f3(a, st, udaseg_conv_flops(d));
  prof_end(0, st, udaseg_conv_flops(d), 0, d);
  return rc;
}

extern "C" int udaseg_conv2d_fwd_f32x3_bnin_ok(const udaseg_conv_desc* d, int up) {
  return d && f3_applicable(d, d->ci, d->co, up ? d->ci : 0) ? 1 : 0;
}

// 1: this launch would take the wave-specialised kernel, whose loader waves can write the transformed activation out (z_out)
extern "C" int udaseg_conv2d_fwd_f32x3_bnin_writes(const udaseg_conv_desc* d) {
  if (!d || !f3_applicable(d, d->ci, d->co, 0)) return 0;
  const int ch = f3_choice(d->hi, d->wi, d->n, d->ci, d->co);
  return (ch == 5 || ch == 6 || (ch >= 7 && ch <= 12)) ? 1 : 0;
}

extern "C" int udaseg_conv2d_fwd_f32x3_bnin(const udaseg_conv_desc* d, const float* x, int up, const float* in_scale,
                                            const float* in_shift, int in_act, float in_slope, float* z_out, const void* wfrag3,
                                            const float* bias, float* y, int act, float slope, double* stats, void* stream) {
  F3Args a = {};
  int rc = f3_common(d, a, "conv2d_fwd_f32x3_bnin");
  if (rc) return rc;
  UDASEG_CHECK_ARG(x && in_scale && in_shift && wfrag3 && y, "conv2d_fwd_f32x3_bnin: NULL pointer");
  UDASEG_CHECK_ARG(in_act == UDASEG_ACT_NONE || in_act == UDASEG_ACT_LEAKY, "conv2d_fwd_f32x3_bnin: unknown activation %d", in_act);
  if (!udaseg_conv2d_fwd_f32x3_bnin_ok(d, up)) {
    set_error("conv2d_fwd_f32x3_bnin: geometry not supported (ask udaseg_conv2d_fwd_f32x3_bnin_ok)");
    return UDASEG_E_UNSUPPORTED;
  }
  const long long px = (long long)d->n * d->hi * d->wi;
  a.x = x; a.wf = wfrag3; a.bias = bias; a.y = y;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act; a.in_slope = in_slope;
  a.ci = d->ci; a.co = d->co;
  a.up_ca = up ? d->ci : 0;            // up: x is the half-resolution tensor behind a nearest x2 up-sampling (no skip source)
  if (z_out != nullptr) {
    UDASEG_CHECK_ARG(!up && udaseg_conv2d_fwd_f32x3_bnin_writes(d),
                     "conv2d_fwd_f32x3_bnin: z_out needs a plain source and a launch on the wave-specialised kernel (ask "
                     "udaseg_conv2d_fwd_f32x3_bnin_writes)");
    a.z_out = z_out;
    a.z_bytes = (unsigned)(px * d->ci * 4);
  }
  a.act = act; a.slope = slope; a.stats = stats;
  a.x_bytes = (unsigned)(up ? (long long)d->n * (d->hi / 2) * (d->wi / 2) * d->ci * 4 : px * d->ci * 4);
  a.w_plane_bytes = (unsigned)(udaseg_frag_elems(d->co, d->ci, 3) * 2);
  a.y_bytes = (unsigned)(px * d->co * 4);
  hipStream_t st = as_stream(stream);
  prof_begin(0, st);
  rc = launch_f3(a, st, udaseg_conv_flops(d));
  prof_end(0, st, udaseg_conv_flops(d), 0, d);
  return rc;
}

extern "C" int udaseg_conv2d_dgrad_f32x3(const udaseg_conv_desc* d, const float* dy, const void* wfrag3_t, float* dx, float* dx2,
                                         int split, const float* prev_y, const float* save_mean, const float* save_rstd,
                                         const float* gamma, const float* beta, int bn_act, float bn_slope, double* bsums,
                                         int accumulate, void* stream) {
  F3Args a = {};
  int rc = f3_common(d, a, "conv2d_dgrad_f32x3");
  if (rc) return rc;
  UDASEG_CHECK_ARG(dy && wfrag3_t && dx, "conv2d_dgrad_f32x3: NULL pointer");
  UDASEG_CHECK_ARG(split == 0 ? dx2 == nullptr : (dx2 != nullptr && split > 0 && split < d->ci && split % 32 == 0),
                   "conv2d_dgrad_f32x3: split=%d of ci=%d channels (a multiple of 32 inside the range, with dx2)", split, d->ci);
  const bool bn = prev_y != nullptr;
  UDASEG_CHECK_ARG(!bn || (save_mean && save_rstd && gamma && beta && bsums && split == 0 && !accumulate),
                   "conv2d_dgrad_f32x3: the BatchNorm-backward sums need mean, rstd, gamma, beta, bsums, one destination, no accumulation");
  UDASEG_CHECK_ARG(!(accumulate && split != 0), "conv2d_dgrad_f32x3: accumulation needs a single destination");
  if (!f3_applicable(d, d->co, d->ci, 0)) {
    set_error("conv2d_dgrad_f32x3: geometry not supported (ask udaseg_conv_f32x3_ok first)");
    return UDASEG_E_UNSUPPORTED;
  }
  const long long px = (long long)d->n * d->hi * d->wi;
  a.x = dy; a.wf = wfrag3_t; a.y = dx; a.y2 = dx2;
  a.ci = d->co; a.co = d->ci; a.split_n = split; a.accumulate = accumulate;
  a.act = UDASEG_ACT_NONE;
  if (bn) {
    a.bnb_y = prev_y; a.bnb_mean = save_mean; a.bnb_rstd = save_rstd; a.bnb_gamma = gamma; a.bnb_beta = beta;
    a.bnb_act = bn_act; a.bnb_slope = bn_slope; a.stats = bsums;
    a.bnb_bytes = (unsigned)(px * d->ci * 4);
  }
  a.x_bytes = (unsigned)(px * d->co * 4);
  a.w_plane_bytes = (unsigned)(udaseg_frag_elems(d->ci, d->co, 3) * 2);
  a.y_bytes = (unsigned)(px * (split > 0 ? split : d->ci) * 4);
  a.y2_bytes = (unsigned)(split > 0 ? px * (d->ci - split) * 4 : 0);
  hipStream_t st = as_stream(stream);
  prof_begin(0, st);
  rc = launch_f3(a, st, udaseg_conv_flops(d));
  prof_end(0, st, udaseg_conv_flops(d), 1, d);
  return rc;
}
