// The decoder's up-sampled input convolved as what it is (round 5): conv3x3(nearest_x2(a)) without the nine taps.
//
// smp's DecoderBlock (reference: the model created at src/test_system.py:90-95, called src/models/train.py:341, differentiated
// :343; trace fixture: aten::upsample_nearest2d + aten::cat in front of every decoder conv1) runs a 3x3 / pad 1 convolution over
// cat([nearest_x2(a), skip]).  On the up-sampled channels the nine taps of an output pixel read only FOUR distinct pixels of `a`:
// for the output parity phase (py, px) = (oy & 1, ox & 1), with oy = 2 q + py,
//     rows oy - 1, oy, oy + 1 of nearest_x2(a)  =  a-rows  q - 1, q, q   (py = 0)   |   q, q, q + 1   (py = 1)
// so   y[2q+py, 2r+px] = sum_{u,v in {0,1}}  W'_{py,px}[u][v] . a[q + py - 1 + u, r + px - 1 + v]
// with the pre-summed weights  W'_{py,px}[u][v] = sum_{ky in Ky(py,u)} sum_{kx in Kx(px,v)} W[ky][kx],
//     Ky(0,0) = {0}, Ky(0,1) = {1,2}, Ky(1,0) = {0,1}, Ky(1,1) = {2}   (zero padding of nearest_x2(a) = zero padding of a).
// 4 taps instead of 9 on the up-sampled half of every decoder conv1: 53.7 of the 358.8 forward GFLOP of BASELINE cfg 2, and the
// same share of the data gradient, which in this form is produced at a's own resolution (no 2x2 sum-pool pass afterwards):
//     da[i, j] = sum_{py,px} sum_{u,v}  W'_{py,px}[u][v]^T . dy[2 (i - py + 1 - u) + py, 2 (j - px + 1 - v) + px].
// The sums W[ky] + W[ky'] are formed in fp32 by the packer (one rounding per sum, <= 2^-24 relative: tests grade the kernels
// against a float64 convolution of cat([interpolate(a, 2, 'nearest'), skip]) next to the nine-tap kernels), then split into the
// three bf16 planes exactly like every other weight of the fp32-on-the-bf16-pipe kernels (conv_halo_f32x3.hip).
//
// Two kernels, both the wave-specialised scheme of conv3x3_f32x3_ws_kernel (4 MFMA waves + 4 loader waves, double-buffered halo
// and weight stages in LDS, one barrier per group), both working in a's coordinates:
//   * conv_up_fwd_f32x3_kernel: a block owns TH x TW pixels of `a` = 2TH x 2TW output pixels; the (TH+2) x (TW+2) halo of a
//     16-channel chunk of `a` is staged ONCE for all four phases; every MFMA wave holds the accumulators of the four phases of its
//     a-rows (the halo staging is shared by 4x the output pixels: the loader waves keep up although a chunk now carries 4/9 of the
//     MFMA work per output pixel); a chunk runs as four groups (px, ex) in {(0,0), (0,1), (1,1), (1,2)} of halo column offset ex,
//     each with the four (halo row offset ey, py) combinations (0,0), (1,0), (1,1), (2,1); the epilogue writes the phases to their
//     strided output pixels, optionally on top of what the skip half's convolution (the plain nine-tap kernel over `skip`, run
//     first) has left there, and makes the BatchNorm statistics of the sum.
//   * conv_up_dgrad_f32x3_kernel: gathers dy through a space-to-depth view (phase (py, px) of dy is an image of a's size whose
//     pixels are 2 pixels apart in memory: the same 64-byte pieces per (pixel, chunk) as any other gather), one "virtual chunk"
//     per (16 channels of dy, phase) with the 2 x 2 taps that phase has in a's 3 x 3 neighbourhood, produces da directly.
// Weight fragments: udaseg_pack_up_batched_f32x3 (modes 2 / 3), layout plane[p][nb][G][4][lane][8] -- G the kernel's group index
// (forward: 4 chunk + (px, ex) group; data gradient: 4 chunk + phase, one group per virtual chunk), four fragments per group and
// 32-channel block.  Sign pattern + - - + over the groups as in
// halo_common.h (the bf16 MFMA adder truncates).
#include <stdlib.h>

#include "common.h"
#include "halo_common.h"

namespace udaseg {

struct UpArgs {
  const float* x;       // forward: a [n][h][w][ci]; data gradient: dy [n][2h][2w][ci]
  const void* wf;       // [3][frag_elems(co, ci, 4)] bf16
  float* y;             // forward: y [n][2h][2w][co]; data gradient: da [n][h][w][co]
  int n, h, w, ci, co;  // h, w: a's extents; ci gathered / co produced channels of THIS launch
  int accumulate;
  double* stats;        // forward: [R][2][co] f64 BatchNorm statistics of the (accumulated) output, or null
  double* sscr;
  int ntx, nty, ncb, nk16;
  int q1, q3;           // groups [q1, q3) run on negated weights and a negated accumulator
  unsigned x_bytes, w_plane_bytes, y_bytes;
  // data gradient: the conv output of the conv + BatchNorm + activation layer that PRODUCED a (a's only consumer is this convolution,
  // so da is that activation's complete gradient) -> its two BatchNorm-backward sums into stats, as conv_halo_f32x3_epilogue.inc
  const float* bnb_y;
  const float* bnb_mean;
  const float* bnb_rstd;
  const float* bnb_gamma;
  const float* bnb_beta;
  int bnb_act;
  float bnb_slope;
};

__host__ __device__ inline void up_negated_groups(int ng, int& q1, int& q3) {
  q1 = (ng + 2) / 4;
  q3 = ng - q1;
}

template <int WM_, int WN_, int RPW_, int TW_, int J_>
struct UpCfg {
  static constexpr int WM = WM_, WN = WN_, RPW = RPW_;
  static_assert(WM_ * WN_ == 4, "four MFMA waves");
  static_assert(TW_ == 32 || TW_ == 16, "32-pixel rows, or 16-pixel rows (two image rows per MFMA block)");
  static constexpr int NT = 512, NLD = 256;
  static constexpr int TW = TW_, RL = 32 / TW_;
  static constexpr int TH = WM * RPW * RL;
  static constexpr int NJ = RL * RPW + 2;                  // fragment start rows per halo column offset
  static constexpr int HR = TH + 2, HWD = TW + 2;
  static constexpr int HWP = TW == 16 ? 24 : HWD;          // LDS pitch of a halo row in pixels (conv_halo_f32x3.hip F3WsCfg::HWP)
  static constexpr int PLANE = HR * HWP * 32;
  static constexpr int LDS_HALO = 3 * PLANE;
  static constexpr int NPIECE = HR * HWD * 2;
  static constexpr int NI = (NPIECE + NLD - 1) / NLD;
  static constexpr int J = J_;                             // weight fragments per group, 32-channel block and plane
  static constexpr int NFG = WN * J * 3;
  static constexpr int NWI = (NFG + 3) / 4;
  static constexpr int LDS_WBUF = NFG * 1024;
  static constexpr int LDS = 2 * LDS_HALO + 2 * LDS_WBUF;
  static_assert(LDS_HALO >= 2 * 4 * 32 * 4, "statistics scratch fits the halo region");
};

// ---- the loader role, shared by both kernels.  S2D: the gathered tensor is dy seen through the space-to-depth view.
// GPH: groups per staged halo (forward: 4 per chunk; data gradient: 1 per virtual chunk).
template <class C, bool S2D, int GPH>
__device__ __forceinline__ void up_loader_role(const UpArgs& a, char* smem, char* wlds, int lt, int lane, int mw, int cb, int img,
                                               int y0, int x0, int nhalo, int NG) {
  const int oct = lt & 1;
  unsigned voff[C::NI], soffl[C::NI];
  const int H = a.h, W = a.w;
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    const int piece = lt + i * C::NLD;
    const int pix = piece >> 1;
    const int hy = pix / C::HWD, hx = pix - hy * C::HWD;
    const int iy = y0 + hy - 1, ix = x0 + hx - 1;
    const bool ok = piece < C::NPIECE && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    if (S2D) voff[i] = ok ? (unsigned)((((img * 2 * H + 2 * iy) * (2 * W) + 2 * ix) * a.ci + oct * 8) * 4) : 0x80000000u;
    else voff[i] = ok ? (unsigned)((((img * H + iy) * W + ix) * a.ci + oct * 8) * 4) : 0x80000000u;
    soffl[i] = (unsigned)((hy * C::HWP + hx) * 32 + ((oct ^ ((hx >> 3) & 1)) * 16));
  }
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wf), 0, (int)(3u * a.w_plane_bytes), 0x00020000);
  const int nblocks32 = (a.co + 31) >> 5;
  int wbase[C::NWI];
  const unsigned wlane16 = (unsigned)lane * 16u;
#pragma unroll
  for (int i = 0; i < C::NWI; ++i) {
    const int q = mw + 4 * i;                      // slot of the group: ((wq * J + j) * 3 + plane)
    const int wq = q / (3 * C::J), rem = q - wq * 3 * C::J;
    const int j = rem / 3, pl = rem - j * 3;
    const int nbq = cb * C::WN + wq;
    const bool live = q < C::NFG && nbq < nblocks32;
    wbase[i] = live ? (int)(pl * a.w_plane_bytes) + (nbq * NG * C::J + j) * 1024 : -1;
  }
  u32x4 stage[C::NI][2], wstage[C::NWI];
  auto load_halo = [&](int hc) {
    int soff;
    unsigned kill;
    if (S2D) {                                     // virtual chunk hc = (chunk, phase): phase (py, px) starts (py * 2W + px) pixels later
      const int c = hc >> 2, p = hc & 3;
      soff = c * 64 + (((p >> 1) * 2 * W + (p & 1)) * a.ci) * 4;
      kill = (c * 16 + oct * 8 < a.ci) ? 0u : 0x80000000u;
    } else {
      soff = hc * 64;
      kill = (hc * 16 + oct * 8 < a.ci) ? 0u : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
      stage[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(voff[i] | kill), soff, 0);
      stage[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(voff[i] | kill), soff + 16, 0);
    }
  };
  auto store_halo = [&](int buf) {
    char* hb = smem + buf * C::LDS_HALO;
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
      if (i < C::NI - 1 || lt + i * C::NLD < C::NPIECE) {
        u32x4 p0, p1, p2;
        split3(stage[i][0], stage[i][1], p0, p1, p2);
        *reinterpret_cast<u32x4*>(hb + soffl[i]) = p0;
        *reinterpret_cast<u32x4*>(hb + C::PLANE + soffl[i]) = p1;
        *reinterpret_cast<u32x4*>(hb + 2 * C::PLANE + soffl[i]) = p2;
      }
    }
  };
  auto load_w = [&](int G) {
    const int soff = G * C::J * 1024;
#pragma unroll
    for (int i = 0; i < C::NWI; ++i)
      wstage[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wbase[i] < 0 ? 0x80000000u : wlane16), wbase[i] < 0 ? 0 : wbase[i] + soff, 0);
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < C::NWI; ++i) {
      const int q = mw + 4 * i;
      if (i < C::NWI - 1 || q < C::NFG) *reinterpret_cast<u32x4*>(wlds + buf * C::LDS_WBUF + q * 1024 + lane * 16) = wstage[i];
    }
  };

  load_halo(0);
  load_w(0);
  store_halo(0);
  store_w(0);
  if (nhalo > 1) load_halo(1);
  if (NG > 1) load_w(1);
  __syncthreads();                                 // group 0 is staged
  for (int G = 0; G < NG; ++G) {
    const int hc = G / GPH, gi = G - hc * GPH;
    if (G + 1 < NG) {
      store_w((G + 1) & 1);                        // last read by the MFMA waves in group G - 1
      if (G + 2 < NG) load_w(G + 2);
    }
    if (gi == GPH / 2 && hc + 1 < nhalo) {
      store_halo((hc + 1) & 1);                    // last read while halo hc - 1 was current
      if (hc + 2 < nhalo) load_halo(hc + 2);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------ forward
template <int WM, int WN, int RPW, int TW>
__global__ __launch_bounds__(512, 1) void conv_up_fwd_f32x3_kernel(const UpArgs a) {
  using C = UpCfg<WM, WN, RPW, TW, 4>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wlds = smem + 2 * C::LDS_HALO;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int mw = wave & 3;
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
  const int nchunk = (a.ci + 15) >> 4, NG = 4 * nchunk;

  if (loader) {
    up_loader_role<C, false, 4>(a, smem, wlds, tid - 256, lane, mw, cb, img, y0, x0, nchunk, NG);
    if (a.stats != nullptr) __syncthreads();       // the statistics reduction has one block barrier
    return;
  }

  int nb = cb * WN + wn;
  const bool wave_live = nb < nblocks32;
  if (!wave_live) nb = 0;
  f32x16 acc[4][RPW];                              // [py * 2 + px][MFMA block of this wave]
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[p][r][v] = 0.f;

  int poff[3];
#pragma unroll
  for (int ex = 0; ex < 3; ++ex) {
    const int hx = lp % C::TW + ex;
    poff[ex] = ((wm * RPW * C::RL + lp / C::TW) * C::HWP + hx) * 32 + ((lh ^ ((hx >> 3) & 1)) * 16);
  }
  const int wrd = (wn * 4 * 3) * 1024 + lane * 16;
  __syncthreads();                                 // group 0 is staged
  for (int c = 0; c < nchunk; ++c) {
    const char* hb = smem + (c & 1) * C::LDS_HALO;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      constexpr int PXG[4] = {0, 0, 1, 1}, EXG[4] = {0, 1, 1, 2};
      const int px = PXG[g], ex = EXG[g];
      const int G = 4 * c + g;
      const char* wb = wlds + (G & 1) * C::LDS_WBUF + wrd;
      if (G == a.q1 || G == a.q3) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int r = 0; r < RPW; ++r) acc[p][r] = -acc[p][r];
      }
      u32x4 bf[4][3];                              // j = (ey, py) in {(0,0), (1,0), (1,1), (2,1)}
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bf[j][pl] = *reinterpret_cast<const u32x4*>(wb + (j * 3 + pl) * 1024);
      u32x4 pf[2][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) pf[0][pl] = *reinterpret_cast<const u32x4*>(hb + pl * C::PLANE + poff[ex]);
#pragma unroll
      for (int s = 0; s < C::NJ; ++s) {
        if (s + 1 < C::NJ) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            pf[(s + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(hb + pl * C::PLANE + poff[ex] + (s + 1) * C::HWP * 32);
        }
#pragma unroll
        for (int ij = 2; ij >= 0; --ij)
#pragma unroll
          for (int i = 0; i <= ij; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              constexpr int EYJ[4] = {0, 1, 1, 2}, PYJ[4] = {0, 0, 1, 1};
              const int d = s - EYJ[j];            // the fragment that starts at halo row s is row offset ey of MFMA block d / RL
              if (d >= 0 && d % C::RL == 0 && d / C::RL < RPW)
                acc[PYJ[j] * 2 + px][d / C::RL] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, bf[j][i]), __builtin_bit_cast(bf16x8, pf[s & 1][ij - i]), acc[PYJ[j] * 2 + px][d / C::RL], 0, 0, 0);
            }
      }
      __syncthreads();                             // group G + 1 is staged; this group's buffers may be rewritten
    }
  }

  // ---- epilogue: phase (py, px) of a-pixel (ay, ax) is output pixel (2 ay + py, 2 ax + px)
  const int cbase = nb * 32;
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, 0x00020000);
  const bool want_stats = a.stats != nullptr;
  float sA[16], sB[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) sA[v] = sB[v] = 0.f;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int ay = y0 + (wm * RPW + r) * C::RL + lp / C::TW, ax = x0 + lp % C::TW;
    const bool pv = wave_live && ay < H && ax < W;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const unsigned pixoff = (unsigned)((img * 2 * H + 2 * ay + (p >> 1)) * (2 * W) + 2 * ax + (p & 1));
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = cbase + 8 * g + 4 * lh;
        const bool cv = pv && c0 < a.co;
        const unsigned off = cv ? (pixoff * (unsigned)a.co + (unsigned)c0) * 4u : 0x80000000u;
        float val[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = acc[p][r][4 * g + e];
        if (a.accumulate) {
          const f32x4 old = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)off, 0, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] += old[e];
        }
        if (want_stats) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float q = cv ? val[e] : 0.f;
            sA[4 * g + e] += q;
            sB[4 * g + e] = __builtin_fmaf(q, q, sB[4 * g + e]);
          }
        }
        u32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = __builtin_bit_cast(unsigned, val[e]);
        __builtin_amdgcn_raw_buffer_store_b128(d, rs_y, (int)off, 0, 0);
      }
    }
  }
  if (want_stats) {
    asm volatile("s_nop 1");
    halfwave_sum_n(sA);
    halfwave_sum_n(sB);
    asm volatile("s_nop 1");
    float* red = reinterpret_cast<float*>(smem);   // [2][4 waves][32]; the K loop ended with a barrier
    if (lp == 31) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int cl = (v & 3) + 8 * (v >> 2) + 4 * lh;
        red[mw * 32 + cl] = wave_live ? sA[v] : 0.f;
        red[4 * 32 + mw * 32 + cl] = wave_live ? sB[v] : 0.f;
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
          t2 += red[4 * 32 + (m * WN + wc) * 32 + cl];
        }
        double* rep = a.sscr != nullptr ? a.sscr + (size_t)(blockIdx.x % HALO_SCR_REPLICAS) * 2 * a.co
                                        : a.stats + (size_t)(blockIdx.x % HALO_STATS_REPLICAS) * 2 * a.co;
        atomicAdd(rep + c, (double)t1);
        atomicAdd(rep + a.co + c, (double)t2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------ data gradient
template <int WM, int WN, int RPW, int TW>
__global__ __launch_bounds__(512, 1) void conv_up_dgrad_f32x3_kernel(const UpArgs a) {
  using C = UpCfg<WM, WN, RPW, TW, 4>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wlds = smem + 2 * C::LDS_HALO;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int mw = wave & 3;
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
  const int nvc = 4 * ((a.ci + 15) >> 4);          // one group per virtual chunk

  if (loader) {
    up_loader_role<C, true, 1>(a, smem, wlds, tid - 256, lane, mw, cb, img, y0, x0, nvc, nvc);
    if (a.stats != nullptr) __syncthreads();       // the reduction of the BatchNorm-backward sums has one block barrier
    return;
  }

  int nb = cb * WN + wn;
  const bool wave_live = nb < nblocks32;
  if (!wave_live) nb = 0;
  f32x16 acc[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[r][v] = 0.f;

  int poff[3];
#pragma unroll
  for (int ex = 0; ex < 3; ++ex) {
    const int hx = lp % C::TW + ex;
    poff[ex] = ((wm * RPW * C::RL + lp / C::TW) * C::HWP + hx) * 32 + ((lh ^ ((hx >> 3) & 1)) * 16);
  }
  const int wrd = (wn * 4 * 3) * 1024 + lane * 16;
  __syncthreads();                                 // group 0 is staged
  for (int vc = 0; vc < nvc; ++vc) {
    const char* hb = smem + (vc & 1) * C::LDS_HALO;
    const int py = (vc >> 1) & 1, px = vc & 1;     // phase of this virtual chunk (scalar)
    const char* wb = wlds + (vc & 1) * C::LDS_WBUF + wrd;
    if (vc == a.q1 || vc == a.q3) {
#pragma unroll
      for (int r = 0; r < RPW; ++r) acc[r] = -acc[r];
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ex = px ? e : 1 + e;               // px = 0: halo column offsets {1, 2}; px = 1: {0, 1}
      u32x4 bf[2][3];                              // halo row offsets ey = ey0 + {0, 1}, ey0 = 1 (py = 0) / 0 (py = 1)
#pragma unroll
      for (int jy = 0; jy < 2; ++jy)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bf[jy][pl] = *reinterpret_cast<const u32x4*>(wb + ((e * 2 + jy) * 3 + pl) * 1024);
      const int po = ex == 0 ? poff[0] : (ex == 1 ? poff[1] : poff[2]);
      const int ey0 = py ? 0 : 1;
      const char* hrow = hb + po + ey0 * C::HWP * 32;        // the rows this phase reads start at halo row ey0
      u32x4 pf[2][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) pf[0][pl] = *reinterpret_cast<const u32x4*>(hrow + pl * C::PLANE);
      constexpr int NS = C::RL * RPW + 1;          // fragment start rows ey0 .. ey0 + RL * RPW
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            pf[(s + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(hrow + pl * C::PLANE + (s + 1) * C::HWP * 32);
        }
#pragma unroll
        for (int ij = 2; ij >= 0; --ij)
#pragma unroll
          for (int i = 0; i <= ij; ++i)
#pragma unroll
            for (int jy = 0; jy < 2; ++jy) {
              const int d = s - jy;
              if (d >= 0 && d % C::RL == 0 && d / C::RL < RPW)
                acc[d / C::RL] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[jy][i]),
                                                                        __builtin_bit_cast(bf16x8, pf[s & 1][ij - i]), acc[d / C::RL], 0, 0, 0);
            }
      }
    }
    __syncthreads();                               // virtual chunk vc + 1 is staged; this one's buffers may be rewritten
  }

  const int cbase = nb * 32;
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, 0x00020000);
  const bool want_bnb = a.stats != nullptr;
  __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(want_bnb ? a.bnb_y : a.x), 0,
                                                                  (int)(want_bnb ? a.y_bytes : 0u), 0x00020000);
  float sA[16], sB[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) sA[v] = sB[v] = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c0 = cbase + 8 * g + 4 * lh;
    const bool cok = wave_live && c0 < a.co;
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc, mu = sc, rsd = sc;
    if (want_bnb && cok) {
      mu = *reinterpret_cast<const f32x4*>(a.bnb_mean + c0);
      rsd = *reinterpret_cast<const f32x4*>(a.bnb_rstd + c0);
      const f32x4 gm = *reinterpret_cast<const f32x4*>(a.bnb_gamma + c0), bt = *reinterpret_cast<const f32x4*>(a.bnb_beta + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[e] = gm[e] * rsd[e];                     // as bn_apply forms them
        sh[e] = bt[e] - mu[e] * sc[e];
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int oy = y0 + (wm * RPW + r) * C::RL + lp / C::TW, ox = x0 + lp % C::TW;
      const bool cv = cok && oy < H && ox < W;
      const unsigned off = cv ? (((unsigned)((img * H + oy) * W + ox)) * (unsigned)a.co + (unsigned)c0) * 4u : 0x80000000u;
      float val[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) val[e] = acc[r][4 * g + e];
      if (a.accumulate) {
        const f32x4 old = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)off, 0, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] += old[e];
      }
      u32x4 d;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = __builtin_bit_cast(unsigned, val[e]);
      __builtin_amdgcn_raw_buffer_store_b128(d, rs_y, (int)off, 0, 0);
      if (want_bnb) {
        const f32x4 yv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, (int)off, 0, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float yy = yv[e];
          const float gg = cv ? val[e] * act_grad(__builtin_fmaf(yy, sc[e], sh[e]), a.bnb_act, a.bnb_slope) : 0.f;
          sA[4 * g + e] += gg;
          sB[4 * g + e] = __builtin_fmaf(gg, (yy - mu[e]) * rsd[e], sB[4 * g + e]);
        }
      }
    }
  }
  if (want_bnb) {
    asm volatile("s_nop 1");
    halfwave_sum_n(sA);
    halfwave_sum_n(sB);
    asm volatile("s_nop 1");
    float* red = reinterpret_cast<float*>(smem);   // [2][4 waves][32]; the K loop ended with a barrier
    if (lp == 31) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int cl = (v & 3) + 8 * (v >> 2) + 4 * lh;
        red[mw * 32 + cl] = wave_live ? sA[v] : 0.f;
        red[4 * 32 + mw * 32 + cl] = wave_live ? sB[v] : 0.f;
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
          t2 += red[4 * 32 + (m * WN + wc) * 32 + cl];
        }
        double* rep = a.sscr != nullptr ? a.sscr + (size_t)(blockIdx.x % HALO_SCR_REPLICAS) * 2 * a.co
                                        : a.stats + (size_t)(blockIdx.x % HALO_STATS_REPLICAS) * 2 * a.co;
        atomicAdd(rep + c, (double)t1);
        atomicAdd(rep + a.co + c, (double)t2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ fragment packing
// table row (int32 x 8): {mode, src element offset, dst element offset (plane 0), N, K, ldk, 0, 0}
//   mode 0: plain 3x3 forward packing (conv_halo_f32x3.hip's mode 0) of channels [src offset .. +K) of OHWI rows of ldk channels
//           -- the skip half of a decoder conv1 as a convolution of its own;
//   mode 1: plain 3x3 data-gradient packing from wt32 [N][9][K] (ldk = K): rows [up_ca, ci) of the dgrad packing = d skip;
//   mode 2: phase packing for conv_up_fwd_f32x3_kernel from w32 OHWI [N][3][3][ldk], channels [0, K);
//   mode 3: phase packing for conv_up_dgrad_f32x3_kernel from wt32 [ci][9][K = co] rows [0, N).
//   mode 4 / 5: sixteen-wide-tile packings (conv_n16_f32x3.hip), N == 16, plane stride = ceil(K / 16) * 5 * 512 elements.
//   mode 8: the stem's packing (conv_stem_f32x3.hip), N == 64, K == 4, plane stride 28 * 512 elements.
// plane stride = frag_elems(N, K, 3) (modes 0, 1) / frag_elems(N, K, 4) (modes 2, 3).
__device__ __forceinline__ void up_taps(int ph, int uv, int& k0, int& k1) {      // Ky(py, u) / Kx(px, v) as a range [k0, k1]
  if (ph == 0) { k0 = uv ? 1 : 0; k1 = uv ? 2 : 0; }
  else { k0 = uv ? 2 : 0; k1 = uv ? 2 : 1; }
}

__global__ void pack_up_batched_f32x3_kernel(const float* __restrict__ w32, const float* __restrict__ wt32,
                                             __bf16* __restrict__ packed, const int* __restrict__ table, int signs) {
  const int* e = table + 8 * blockIdx.y;
  const int mode = e[0], N = e[3], K = e[4], ldk = e[5];
  const float* src = ((mode & 1) ? wt32 : w32) + e[1];
  __bf16* dst = packed + e[2];
  const int nb = (N + 31) >> 5, nk16 = (K + 15) >> 4;
  if (mode < 2) {                                  // the plain packing, with a row stride
    const long long total = (long long)nb * 3 * nk16 * 3 * 64;
    const long long plane = total * 8;
    int q1 = 0, q3 = 0;
    if (signs) f3_negated_groups(nk16, q1, q3);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
      const int lane = (int)(i & 63);
      long long f = i >> 6;
      const int dy = (int)(f % 3);
      f /= 3;
      const int kk = (int)(f % nk16);
      f /= nk16;
      const int dx = (int)(f % 3);
      const int b = (int)(f / 3);
      const int n = b * 32 + (lane & 31), k0 = kk * 16 + 8 * (lane >> 5);
      const int tap = mode ? 8 - (dy * 3 + dx) : dy * 3 + dx;
      u32x4 lo = {0u, 0u, 0u, 0u}, hi = lo;
      if (n < N && k0 < K) {
        const float* s = src + ((size_t)n * 9 + tap) * ldk + k0;
        lo = *reinterpret_cast<const u32x4*>(s);
        if (k0 + 4 < K) hi = *reinterpret_cast<const u32x4*>(s + 4);
      }
      u32x4 p0, p1, p2;
      split3(lo, hi, p0, p1, p2);
      if (3 * kk + dx >= q1 && 3 * kk + dx < q3) {
        p0 ^= 0x80008000u;
        p1 ^= 0x80008000u;
        p2 ^= 0x80008000u;
      }
      *reinterpret_cast<u32x4*>(dst + i * 8) = p0;
      *reinterpret_cast<u32x4*>(dst + plane + i * 8) = p1;
      *reinterpret_cast<u32x4*>(dst + 2 * plane + i * 8) = p2;
    }
    return;
  }
  if (mode == 8) {
    // the stem (conv_stem_f32x3.hip): plane[p][ky][h][cb][lane][8], lane l: channel n = 32 cb + (l & 31), elements j = 16 h + 8 (l >> 5) .. + 7
    // of kernel row ky, j = 4 kx + c (j >= 28: the four zero columns that pad a row's 28 values to K = 32); source w32 OHWI [64][7][7][4]
    const long long total = 28 * 64;
    const long long plane = total * 8;
    int q1 = 0, q3 = 0;
    if (signs) up_negated_groups(14, q1, q3);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
      const int lane = (int)(i & 63), f = (int)(i >> 6);
      const int cb = f & 1, h = (f >> 1) & 1, ky = f >> 2;
      const int n = cb * 32 + (lane & 31), j0 = 16 * h + 8 * (lane >> 5);
      u32x4 lo = {0u, 0u, 0u, 0u}, hi = lo;
      if (n < N) {
        const float* sp = src + ((size_t)n * 49 + ky * 7) * 4 + j0;
        if (j0 < 28) lo = *reinterpret_cast<const u32x4*>(sp);
        if (j0 + 4 < 28) hi = *reinterpret_cast<const u32x4*>(sp + 4);
      }
      u32x4 p0, p1, p2;
      split3(lo, hi, p0, p1, p2);
      const int G = 2 * ky + h;
      if (G >= q1 && G < q3) {
        p0 ^= 0x80008000u;
        p1 ^= 0x80008000u;
        p2 ^= 0x80008000u;
      }
      *reinterpret_cast<u32x4*>(dst + i * 8) = p0;
      *reinterpret_cast<u32x4*>(dst + plane + i * 8) = p1;
      *reinterpret_cast<u32x4*>(dst + 2 * plane + i * 8) = p2;
    }
    return;
  }
  if (mode == 4 || mode == 5) {
    // sixteen-wide tile (conv_n16_f32x3.hip): plane[p][chunk][pair j][lane][8], lane l: channel n = l & 15 (N == 16), K slice
    // g = l >> 4: tap 2 j + (g >> 1) (the ninth tap's partner: zeros), channels 16 chunk + 8 (g & 1) .. + 7 of the chunk.
    // mode 4: forward, w32 OHWI [16][9][ldk]; mode 5: data gradient, wt32 [16 = ci][9][ldk = co], window flipped
    const long long total = (long long)nk16 * 5 * 64;
    const long long plane = total * 8;
    int q1 = 0, q3 = 0;
    if (signs) up_negated_groups(5 * nk16, q1, q3);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
      const int lane = (int)(i & 63);
      const int G = (int)(i >> 6);
      const int kk = G / 5, j = G - 5 * kk;
      const int n = lane & 15, gsl = lane >> 4;
      const int tapl = 2 * j + (gsl >> 1), k0 = kk * 16 + 8 * (gsl & 1);
      u32x4 lo = {0u, 0u, 0u, 0u}, hi = lo;
      if (n < N && tapl < 9 && k0 < K) {
        const int tap = mode == 5 ? 8 - tapl : tapl;
        const float* sp = src + ((size_t)n * 9 + tap) * ldk + k0;
        lo = *reinterpret_cast<const u32x4*>(sp);
        if (k0 + 4 < K) hi = *reinterpret_cast<const u32x4*>(sp + 4);
      }
      u32x4 p0, p1, p2;
      split3(lo, hi, p0, p1, p2);
      if (G >= q1 && G < q3) {
        p0 ^= 0x80008000u;
        p1 ^= 0x80008000u;
        p2 ^= 0x80008000u;
      }
      *reinterpret_cast<u32x4*>(dst + i * 8) = p0;
      *reinterpret_cast<u32x4*>(dst + plane + i * 8) = p1;
      *reinterpret_cast<u32x4*>(dst + 2 * plane + i * 8) = p2;
    }
    return;
  }
  const int J = 4;
  const int NG = 4 * nk16;
  const long long total = (long long)nb * NG * J * 64;
  const long long plane = total * 8;
  int q1 = 0, q3 = 0;
  if (signs) up_negated_groups(NG, q1, q3);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    long long f = i >> 6;
    const int j = (int)(f % J);
    f /= J;
    const int G = (int)(f % NG);
    const int b = (int)(f / NG);
    int kk, py, px, u, v;
    if (mode == 2) {                               // G = 4 chunk + g, g = (px, ex) in {(0,0), (0,1), (1,1), (1,2)}; j = (ey, py)
      kk = G >> 2;
      const int g = G & 3;
      px = g >> 1;
      v = ((g + 1) >> 1) - px;
      py = j >> 1;
      u = j & 1;
    } else {                                       // G = 4 chunk + phase; j = 2 e + jy
      const int ee = j >> 1, jy = j & 1;
      kk = G >> 2;
      py = (G >> 1) & 1;
      px = G & 1;
      const int ex = px ? ee : 1 + ee, ey = (py ? 0 : 1) + jy;
      u = 2 - py - ey;
      v = 2 - px - ex;
    }
    const int n = b * 32 + (lane & 31), k0 = kk * 16 + 8 * (lane >> 5);
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
    if (n < N && k0 < K) {
      int ya, yb, xa, xb;
      up_taps(py, u, ya, yb);
      up_taps(px, v, xa, xb);
      for (int ky = ya; ky <= yb; ++ky)
        for (int kx = xa; kx <= xb; ++kx) {
          const float* s = src + ((size_t)n * 9 + ky * 3 + kx) * ldk + k0;
          lo += *reinterpret_cast<const f32x4*>(s);
          if (k0 + 4 < K) hi += *reinterpret_cast<const f32x4*>(s + 4);
        }
    }
    u32x4 p0, p1, p2;
    split3(__builtin_bit_cast(u32x4, lo), __builtin_bit_cast(u32x4, hi), p0, p1, p2);
    if (G >= q1 && G < q3) {
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

template <int WM, int WN, int RPW, int TW, bool DGRAD>
static int launch_up_t(UpArgs a, hipStream_t s, double flops) {
  using C = UpCfg<WM, WN, RPW, TW, 4>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    hipError_t e;
    if constexpr (DGRAD)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_up_dgrad_f32x3_kernel<WM, WN, RPW, TW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    else
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_up_fwd_f32x3_kernel<WM, WN, RPW, TW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv_up_f32x3)");
    attr_done = true;
  }
  a.ntx = cdiv(a.w, C::TW);
  a.nty = cdiv(a.h, C::TH);
  a.ncb = cdiv(a.co, 32 * C::WN);
  a.nk16 = (a.ci + 15) / 16;
  a.q1 = a.q3 = -1;
  if (f3_signs_on()) up_negated_groups(4 * a.nk16, a.q1, a.q3);
  const long long blocks = (long long)a.n * a.nty * a.ntx * a.ncb;
  if (blocks <= 0) return UDASEG_OK;
  a.sscr = nullptr;
  if (a.stats != nullptr && blocks > 1024) a.sscr = halo_stats_scratch(a.co, s);
  static std::atomic<int> kid{-1};
  if (kid < 0) {
    char nm[96];
    snprintf(nm, sizeof(nm), "conv_up_%s_f32x3_kernel<%d, %d, %d, %d>", DGRAD ? "dgrad" : "fwd", WM, WN, RPW, TW);
    kid = kprof_id(nm);
  }
  hipEvent_t ev = kprof_begin(s);
  if constexpr (DGRAD)
    hipLaunchKernelGGL((conv_up_dgrad_f32x3_kernel<WM, WN, RPW, TW>), dim3((unsigned)blocks), dim3(C::NT), C::LDS, s, a);
  else
    hipLaunchKernelGGL((conv_up_fwd_f32x3_kernel<WM, WN, RPW, TW>), dim3((unsigned)blocks), dim3(C::NT), C::LDS, s, a);
  kprof_end(kid, ev, s, flops);
  UDASEG_LAUNCH_CHECK("conv_up_f32x3 launch");
  if (a.sscr != nullptr) {
    launch_halo_stats_fold(a.sscr, a.co, a.stats, s);
    UDASEG_LAUNCH_CHECK("halo_stats_fold launch");
  }
  return UDASEG_OK;
}

// configuration of a launch: (a's extents, produced channels).  1: 2 x 32 a-pixels x 64 channels; 2: 4 x 32 x 64; 3: 4 x 32 x 32;
// 4: 8 x 32 x 32; 5: 4 x 16 x 64 (16-pixel-wide a); 6: 8 x 16 x 64; 7: 8 x 16 x 32; 8: 8 x 32 x 64 (data gradient only).
// Measured per decoder block of BASELINE cfg 2 (tools/up_probe.py, profiles/r05_up_probe.txt; us forward / data gradient):
// the FORWARD is fastest on 32-channel blocks everywhere (cfg 3 / 7: 83 / 53 / 51 / 61 / 96 us for blocks 0..4 against 109 / 94 /
// 104 / 113 / 211 for the nine-tap gather of the same half) -- four phases x 12 weight fragments per group are this kernel's LDS
// traffic, and two channel blocks per workgroup double it per staged halo; the DATA GRADIENT wants the 8 x 32 x 64 tile wherever
// that still gives a block per CU (43 / 53 us on blocks 2 / 3), 4 x 32 x 64 below that, 32-channel blocks for <= 32 produced.
static int up_choice(int n, int h, int w, int produced, bool dgrad) {
  const int force = opt_get(UDASEG_OPT_UP_CFG);      // udaseg_up_f32x3_force_config: one configuration for every launch
  if (force > 0 && force <= 8) return force;
  if (w <= 16) return 7;
  if (!dgrad || produced <= 32) return 3;
  const long long ncb = cdiv(produced, 64);
  if ((long long)n * cdiv(h, 8) * cdiv(w, 32) * ncb >= 256) return 8;
  return 2;
}

template <bool DGRAD>
static int launch_up(UpArgs a, hipStream_t s, double flops) {
  const int ch = up_choice(a.n, a.h, a.w, a.co, DGRAD);
  switch (ch) {
    case 2: return launch_up_t<2, 2, 2, 32, DGRAD>(a, s, flops);
    case 3: return launch_up_t<4, 1, 1, 32, DGRAD>(a, s, flops);
    case 4: return launch_up_t<4, 1, 2, 32, DGRAD>(a, s, flops);
    case 5: return launch_up_t<2, 2, 1, 16, DGRAD>(a, s, flops);
    case 6: return launch_up_t<2, 2, 2, 16, DGRAD>(a, s, flops);
    case 7: return launch_up_t<4, 1, 1, 16, DGRAD>(a, s, flops);
    case 8:
      if constexpr (DGRAD) return launch_up_t<2, 2, 4, 32, true>(a, s, flops);
      else return launch_up_t<2, 2, 2, 32, false>(a, s, flops);
    default: return launch_up_t<2, 2, 1, 32, DGRAD>(a, s, flops);
  }
}

static bool up_applicable(const udaseg_conv_desc* d, int up_ca) {
  if (!d || !f32_halo_enabled()) return false;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->ho != d->hi || d->wo != d->wi) return false;
  if (d->hi % 2 != 0 || d->wi % 2 != 0 || d->hi < 2 || d->wi < 2) return false;
  if (up_ca <= 0 || up_ca > d->ci || up_ca % 16 != 0 || d->co % 4 != 0 || d->ci % 4 != 0) return false;
  const long long px = (long long)d->n * d->hi * d->wi;
  if (px * d->co * 4 >= (1LL << 31) || px / 4 * up_ca * 4 >= (1LL << 31)) return false;      // buffer descriptors: 2 GiB
  return true;
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_up_f32x3_force_config(int cfg) {
  UDASEG_CHECK_ARG(cfg >= 0 && cfg <= 8, "up_f32x3_force_config: 0 (heuristic) .. 8");
  return udaseg_set_option(UDASEG_OPT_UP_CFG, cfg);
}

extern "C" int udaseg_conv_up_f32x3_ok(const udaseg_conv_desc* d, int up_ca) { return up_applicable(d, up_ca) ? 1 : 0; }

extern "C" int udaseg_pack_up_batched_f32x3(const float* w32, const float* wt32, void* packed, const int* table, int entries,
                                            void* stream) {
  UDASEG_CHECK_ARG(packed && table && entries > 0 && (w32 || wt32), "pack_up_batched_f32x3: NULL pointer / no entries");
  // 1024 blocks per table row: the ~15 rows of a network are few and large (a 768 -> 256 decoder conv1 is 1.8 M weights x 16 phase taps);
  // with 64 blocks per row the launch ran on a fraction of the chip: 32.8 -> 12.4 us (forward table), 37.5 -> 14.6 us (backward),
  // +0.8 % on the cfg 2 step (both sit on the main stream's chain, in front of the stem and of the first data gradient)
  hipLaunchKernelGGL(pack_up_batched_f32x3_kernel, dim3(1024, (unsigned)entries), dim3(256), 0, as_stream(stream), w32, wt32,
                     static_cast<__bf16*>(packed), table, f3_signs_on() ? 1 : 0);
  UDASEG_LAUNCH_CHECK("pack_up_batched_f32x3 launch");
  return UDASEG_OK;
}

// y (+)= conv3x3(nearest_x2(a)) restricted to the first up_ca input channels of the convolution d describes (d: the whole decoder
// conv1 at the OUTPUT resolution, ci = up_ca + skip channels); stats: BatchNorm statistics of y after the accumulation.
extern "C" int udaseg_conv2d_fwd_up_f32x3(const udaseg_conv_desc* d, const float* a, int up_ca, const void* wfrag_up, float* y,
                                          int accumulate, double* stats, void* stream) {
  UDASEG_CHECK_ARG(d && a && wfrag_up && y, "conv2d_fwd_up_f32x3: NULL pointer");
  if (!up_applicable(d, up_ca)) {
    set_error("conv2d_fwd_up_f32x3: geometry not supported (ask udaseg_conv_up_f32x3_ok first)");
    return UDASEG_E_UNSUPPORTED;
  }
  UpArgs u = {};
  u.x = a; u.wf = wfrag_up; u.y = y;
  u.n = d->n; u.h = d->hi / 2; u.w = d->wi / 2; u.ci = up_ca; u.co = d->co;
  u.accumulate = accumulate; u.stats = stats;
  const long long pa = (long long)d->n * u.h * u.w;
  u.x_bytes = (unsigned)(pa * up_ca * 4);
  u.w_plane_bytes = (unsigned)(udaseg_frag_elems(d->co, up_ca, 4) * 2);
  u.y_bytes = (unsigned)(4 * pa * d->co * 4);
  udaseg_conv_desc dd = *d;                        // profile record / FLOPs: the 2 x 2 window over a, four phases
  dd.ci = up_ca;
  const double flops = 2.0 * 4.0 * (double)pa * 4.0 * up_ca * d->co;
  hipStream_t st = as_stream(stream);
  prof_begin(0, st);
  const int rc = launch_up<false>(u, st, flops);
  prof_end(0, st, flops, 0, &dd);
  return rc;
}

// da (+)= the gradient of `a` through conv3x3(nearest_x2(a)): the whole chain (convolution transpose + the up-sampling's 2 x 2
// sum) in one pass at a's resolution.  wfrag_up_t: mode-3 packing of the dgrad-packed weights.  prev_y != NULL: also the
// BatchNorm-backward sums of the layer that produced a (udaseg_conv2d_dgrad_f32x3's contract; no accumulation then).
extern "C" int udaseg_conv2d_dgrad_up_f32x3(const udaseg_conv_desc* d, const float* dy, int up_ca, const void* wfrag_up_t, float* da,
                                            const float* prev_y, const float* save_mean, const float* save_rstd, const float* gamma,
                                            const float* beta, int bn_act, float bn_slope, double* bsums, int accumulate, void* stream) {
  UDASEG_CHECK_ARG(d && dy && wfrag_up_t && da, "conv2d_dgrad_up_f32x3: NULL pointer");
  const bool bn = prev_y != nullptr;
  UDASEG_CHECK_ARG(!bn || (save_mean && save_rstd && gamma && beta && bsums && !accumulate),
                   "conv2d_dgrad_up_f32x3: the BatchNorm-backward sums need mean, rstd, gamma, beta, bsums and no accumulation");
  UDASEG_CHECK_ARG(bn || bsums == nullptr, "conv2d_dgrad_up_f32x3: bsums without prev_y");
  if (!up_applicable(d, up_ca) || d->co % 8 != 0) {
    set_error("conv2d_dgrad_up_f32x3: geometry not supported (ask udaseg_conv_up_f32x3_ok first; co a multiple of 8)");
    return UDASEG_E_UNSUPPORTED;
  }
  UpArgs u = {};
  u.x = dy; u.wf = wfrag_up_t; u.y = da;
  u.n = d->n; u.h = d->hi / 2; u.w = d->wi / 2; u.ci = d->co; u.co = up_ca;
  u.accumulate = accumulate;
  if (bn) {
    u.bnb_y = prev_y; u.bnb_mean = save_mean; u.bnb_rstd = save_rstd; u.bnb_gamma = gamma; u.bnb_beta = beta;
    u.bnb_act = bn_act; u.bnb_slope = bn_slope; u.stats = bsums;
  }
  const long long pa = (long long)d->n * u.h * u.w;
  u.x_bytes = (unsigned)(4 * pa * d->co * 4);
  u.w_plane_bytes = (unsigned)(udaseg_frag_elems(up_ca, d->co, 4) * 2);
  u.y_bytes = (unsigned)(pa * up_ca * 4);
  udaseg_conv_desc dd = *d;
  dd.ci = up_ca;
  const double flops = 2.0 * 4.0 * (double)pa * 4.0 * up_ca * d->co;
  hipStream_t st = as_stream(stream);
  prof_begin(0, st);
  const int rc = launch_up<true>(u, st, flops);
  prof_end(0, st, flops, 1, &dd);
  return rc;
}
