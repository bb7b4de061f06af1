// Implicit-GEMM 2-D convolution, forward and data-gradient, NHWC fp32 on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Replaces torch.nn.functional.conv2d / its autograd dgrad as reached from smp.Unet.forward and
// DomainDiscriminator.forward (reference src/models/train.py:341,343; src/models/discriminator.py:54).
//
// GEMM view:  C[M x N] = A[M x K] * B[K x N]
//   M = output pixels of one launch (a sub-lattice of the output tensor, see below)
//   N = output channels, K = taps * gathered channels
//   A is gathered on the fly from the NHWC operand (im2col never materialised), B is the OHWI weight.
// One kernel serves forward and dgrad: a launch is described by an output sub-lattice
//   out(y, x) = (cy + sy_o*jy, cx + sx_o*jx)      jy < JY, jx < JX
// and a tap table   in(y, x) = (sy_i*jy + dy[t], sx_i*jx + dx[t]),  weight tap wt[t].
//   forward, stride s, pad p :  lattice = whole output, sy_i = s, dy[t] = r - p
//   dgrad of a stride-s conv :  one launch per output parity class (ph, pw) in [0,s)^2 with the taps
//                               r = (ph+p) mod s, ...;  dy[t] = (ph + p - r)/s, sy_i = 1  -> dense work, no atomics.
//
// Tiling (256 threads = 4 waves): block tile BM x BN x 32, double-buffered LDS, A and B staged through
// registers (global_load_dwordx4 issued before the MFMA phase of the current tile, ds_write_b128 after it),
// one barrier per K-tile.  LDS rows are K-contiguous with a 4-float pad: ds_read_b128 conflict-free.
// Each lane's 16-byte LDS read feeds 4 consecutive MFMAs (the K order inside a tile is permuted identically
// for A and B, which a GEMM does not care about).
#include <stdlib.h>

#include "common.h"
#include "halo_common.h"

namespace udaseg {

constexpr int NCLS = 4;  // classes per launch: dgrad parity classes of a strided conv, or K-slices (tap ranges)

struct IgemmArgs {
  const void* x;      // fp32 or bf16 NHWC (template parameter BF of the kernel)
  const void* w;
  const float* bias;  // always fp32
  void* y;
  int hi, wi, ci;
  int ho, wo, co;
  int sy_o, sx_o, sy_i, sx_i;
  int tfull;
  float inv_ci;
  int accumulate, act;
  float slope;
  int dense_out;
  const void* residual;  // same layout / type as y: added before the activation (eval-mode residual blocks), or null
  int out_f32;            // bf16 kernels only: write fp32 outputs (segmentation logits)
  int bf16;               // host-side selector of the kernel instantiation
  double* stats;    // BN statistics of the output: [R][2][co] f64 accumulators (sum, sum of squares), or null
  int atomic_out;   // K-slices of one output: epilogue adds with global_atomic_add_f32 (output pre-zeroed or accumulated)
  int nclass;
  // per class
  int JY[NCLS], JX[NCLS], M[NCLS], cy[NCLS], cx[NCLS];
  int ntaps[NCLS], K[NCLS], tap_off[NCLS];
  int tile_begin[NCLS + 1];   // prefix sums of the classes' tile counts (blockIdx.x ranges)
  float inv_jx[NCLS], inv_jy[NCLS];
  signed char dy[64];
  signed char dx[64];
  unsigned char wt[64];
  // The taps of a class are a contiguous run [g_t0, g_t0 + ntaps) of a row-major g_ny x g_nx GRID of offsets
  // (dy, dx) = (g_dy0 + g_sy * a, g_dx0 + g_sx * b), unit steps of either sign: the whole k x k window (forward), a parity
  // class of it (strided dgrad), a K-slice of either.  The uniform-tap loop builds each row's tap-validity mask from that
  // description with a handful of ALU operations; reading dy[] / dx[] in a loop costs one vector memory round trip per tap
  // (byte-sized kernarg reads are global loads): ~2 us per row, a fifth of a K = 576 tile's life (round 2, ISA + trace).
  int g_dy0[NCLS], g_dx0[NCLS], g_sy[NCLS], g_sx[NCLS], g_ny[NCLS], g_nx[NCLS], g_t0[NCLS];
  // uniform-tap fast path (ci a multiple of the K-tile: every thread of a block is in the same tap during a K-step)
  int uniform;                 // host-side selector
  unsigned x_bytes, w_bytes;   // operand sizes for the buffer descriptors (range-checked loads: out of range reads 0)
  int tap_xoff[32];            // ((dy*wi + dx) * ci) * element_size
  int tap_woff[32];            // wt * ci * element_size
  // fused nearest-x2-upsample + concat input (decoder blocks of smp.Unet: cat([up(a), skip], 1) is never materialised):
  // the gathered tensor is VIRTUAL, channels [0, up_ca) come from x = a [n][hi/2][wi/2][up_ca] read at (iy >> 1, ix >> 1),
  // channels [up_ca, ci) from x2 = skip [n][hi][wi][ci - up_ca].  Uniform-tap loop only (up_ca, ci multiples of the K-tile).
  const void* x2;
  int up_ca;
  unsigned x2_bytes;
  // split output (data gradient of such a convolution): output channels [0, split_n) go to y with row length split_n,
  // channels [split_n, co) to y2 with row length co - split_n.  split_n is a multiple of every tile width, or 0.
  void* y2;
  int split_n;
  // diagnosis (udaseg_debug_set_timeline): per block {entry, first load, end of K loop, exit} in 100 MHz ticks + HW_ID + XCC_ID
  unsigned long long* timeline;
  // BatchNorm-backward statistics of the layer BEHIND this data gradient (training-mode BN + activation between the
  // previous convolution's output bnb_y and this convolution's input): with g = out * act'(bn(bnb_y)) and
  // xhat = (bnb_y - mean) * rstd the epilogue adds sum(g), sum(g * xhat) per channel into `stats` -- what
  // udaseg_bn_bwd_reduce would compute in a separate pass over (out, bnb_y).  Whole-tile launches only (host-checked).
  // Pixel folding (bf16, <= 32-channel 3x3 layers; see fold_plan): F neighbouring pixels of a row are addressed as ONE pixel
  // with F x the channels, so physical channel n stands for logical channel n % cmod in every per-channel vector (bias,
  // BatchNorm statistics, the bnb_* coefficients).  0 = not folded.
  int cmod;
  const float* bnb_y;
  const float* bnb_mean;
  const float* bnb_rstd;
  const float* bnb_gamma;
  const float* bnb_beta;
  int bnb_act;
  float bnb_slope;
};

__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // what a masked-out gather lane reads

constexpr int STATS_REPLICAS = 16;  // == udaseg_bn_replicas() (norm_act.hip)
constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

// X3 (fp32 storage on the bf16 matrix pipe): a K-tile of 32 fp32 per row is kept as three bf16 planes of 64-byte rows
template <int BM, int BN, bool X3 = false>
constexpr int igemm_lds_bytes() { return (X3 ? 2 * 3 * (BM + BN) * 64 : 2 * (BM + BN) * LDS_LD * 4) + 3 * 64 * 4; }

// BF = false: fp32 storage, v_mfma_f32_32x32x2_f32, 32 K-elements per tile.
// BF = true : bf16 storage, v_mfma_f32_32x32x16_bf16, 64 K-elements per tile, fp32 accumulation / bias / statistics.
// Both stage 128 bytes per row per tile with 16-byte loads, so gather, LDS layout and fragment reads are byte-identical;
// a lane's 16-byte LDS read is 4 fp32 (-> 4 MFMAs) or 8 bf16 (-> exactly one bf16 MFMA's K slice: k = 16s + 8*lh + j).
// UNI = true: the uniform-tap main loop.  The generic loop spends ~54 VALU instructions per wave per K-step on gather
// addresses (tap lookup, bounds tests, 64-bit address arithmetic); fp32 MFMA runs on the same ALUs as the VALU
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0 on this kernel), so those instructions come straight out of the matrix rate.  Here the
// tap of a K-step is a scalar, the per-row validity of every tap is a bit mask built once per tile, addresses are 32-bit
// offsets into range-checked buffer loads (a masked-out lane gets bit 31 set and reads zeros): 3 VALU per A row-load,
// 1 per B row-load.
// UP = true (with UNI): the fused upsample + concat gather described at IgemmArgs::x2.
// X3 = true (with BF = false; round 4): fp32 storage in and out, the products on v_mfma_f32_32x32x16_bf16 -- every staged operand is
// split EXACTLY into three bf16 terms (halo_common.h split3_4) on its way into LDS, six MFMAs per fragment pair (conv_halo_f32x3.hip
// has the arithmetic).  The layers the halo kernels do not take (stride 2, 7x7 stem, 1x1 / stride 2, 4x4 / stride 2 in fp32) leave the
// fp32 matrix pipe this way.  LDS: per buffer three planes of A and of B, 64-byte rows (32 bf16), the four 16-byte units of a row
// XOR-swizzled with bits 2..3 of the row so that the 16 lanes of a ds_read_b128 phase cover all 64 banks without padding.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool BF, bool UNI, bool UP, bool X3 = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const IgemmArgs a) {
  static_assert(UNI || !UP, "the fused gather lives in the uniform-tap loop");
  static_assert(!X3 || !BF, "the three-term split reads fp32 operands");
  constexpr int X3_BUF = 3 * (BM + BN) * 64;     // bytes of one X3 stage
  constexpr int ES = BF ? 2 : 4;        // element bytes
  constexpr int EPV = 16 / ES;          // elements per 16-byte vector
  constexpr int BKE = BK * 4 / ES;      // K elements per tile
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_PASS = BM / 32;
  constexpr int B_PASS = (BN + 31) / 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");
  static_assert(BN % 32 == 0, "every B load pass covers 32 whole rows (no per-thread guard in the K loop)");

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* As = reinterpret_cast<float*>(smem_raw);          // [2][BM][LDS_LD]
  float* Bs = As + 2 * BM * LDS_LD;                         // [2][BN][LDS_LD]
  int* taps = X3 ? reinterpret_cast<int*>(smem_raw + 2 * X3_BUF) : reinterpret_cast<int*>(Bs + 2 * BN * LDS_LD);  // [3][64]: dy, dx, wt

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;
  unsigned long long tl0 = 0, tl1 = 0, tl2 = 0;
  if (a.timeline) tl0 = wall_clock64();

  // ---- block -> (class, tile).  Within a class the tile order is XCD-aware: blocks that share blockIdx%8 (one XCD's
  // L2) take a contiguous run of tiles, N-tiles innermost, so an XCD re-reads its own A rows / halos from its own L2.
  int cls = 0;
#pragma unroll
  for (int c = 1; c < NCLS; ++c)
    if (c < a.nclass && (int)blockIdx.x >= a.tile_begin[c]) cls = c;
  const int ntn = (a.co + BN - 1) / BN;
  const int nblk = a.tile_begin[cls + 1] - a.tile_begin[cls];
  int bid = blockIdx.x - a.tile_begin[cls];
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / ntn) * BM;
  const int n0 = (bid % ntn) * BN;
  const int cM = a.M[cls], cJX = a.JX[cls], cJY = a.JY[cls], cK = a.K[cls], cnt = a.ntaps[cls], toff = a.tap_off[cls];
  const float cinv_jx = a.inv_jx[cls], cinv_jy = a.inv_jy[cls];

  if constexpr (!UNI) {       // the generic loop looks its taps up in LDS; the uniform loop never reads the table
    if (tid < 64) {
      taps[tid] = a.dy[tid];
      taps[64 + tid] = a.dx[tid];
      taps[128 + tid] = a.wt[tid];
    }
  }

  // ---- per-thread load slots: float4 column kq of rows lrow + 32*p
  const int kq = tid & 7, lrow = tid >> 3;
  int a_base[A_PASS], a_iy[A_PASS], a_ix[A_PASS];
#pragma unroll
  for (int p = 0; p < A_PASS; ++p) {
    const int m = m0 + lrow + 32 * p;
    if (m < cM) {
      const int t1 = fast_div(m, cJX, cinv_jx);
      const int jx = m - t1 * cJX;
      const int ni = fast_div(t1, cJY, cinv_jy);
      const int jy = t1 - ni * cJY;
      a_base[p] = ni * a.hi * a.wi;
      a_iy[p] = jy * a.sy_i;
      a_ix[p] = jx * a.sx_i;
    } else {
      a_base[p] = 0;
      a_iy[p] = -(1 << 20);
      a_ix[p] = 0;
    }
  }
  int b_off[B_PASS];
#pragma unroll
  for (int p = 0; p < B_PASS; ++p) {
    const int n = n0 + lrow + 32 * p;
    b_off[p] = n < a.co ? n * a.tfull * a.ci : -1;
  }

  // uniform path: byte offset of (row's pixel, this thread's 16-byte K column) and the row's per-tap "invalid" bits
  unsigned u_aoff[A_PASS], u_ainv[A_PASS], u_boff[B_PASS];
  unsigned u_acur[A_PASS];   // gather offset of each row for the tap (and source) the loop is in: recomputed when the tap changes
  if constexpr (UNI) {
    const int gy0 = a.g_dy0[cls], gsy = a.g_sy[cls], gny = a.g_ny[cls];
    const int gx0 = a.g_dx0[cls], gsx = a.g_sx[cls], gnx = a.g_nx[cls], gt0 = a.g_t0[cls];
#pragma unroll
    for (int p = 0; p < A_PASS; ++p) {
      unsigned inv = 0xffffffffu;
      if (a_iy[p] > -(1 << 19)) {
        // which grid rows / columns fall inside the image for this output row, then their outer product
        unsigned yb = 0u, xb = 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (q < gny) yb |= ((unsigned)(a_iy[p] + gy0 + gsy * q) < (unsigned)a.hi ? 1u : 0u) << q;
          if (q < gnx) xb |= ((unsigned)(a_ix[p] + gx0 + gsx * q) < (unsigned)a.wi ? 1u : 0u) << q;
        }
        unsigned long long valid = 0ull;
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q < gny) valid |= (unsigned long long)((yb >> q) & 1u ? xb : 0u) << (q * gnx);
        inv = ~(unsigned)(valid >> gt0);        // bit tl = tap tl of this class is outside the image (bits >= ntaps unused)
      }
      u_ainv[p] = inv;
      // rows past the end of the class keep offset 0: with every tap marked invalid their address is 2^31 + (a small tap
      // offset), safely outside any buffer of at most 2^30 bytes (a wrapped garbage offset might not be)
      u_aoff[p] = inv == 0xffffffffu
                      ? 0u
                      : ((unsigned)(a_base[p] + a_iy[p] * a.wi + a_ix[p]) * (unsigned)a.ci + (unsigned)(kq * EPV)) * (unsigned)ES;
      u_acur[p] = 0x80000000u;
      if constexpr (UP) {
        if (inv == 0xffffffffu) a_base[p] = a_iy[p] = a_ix[p] = 0;   // keeps the recomputed offsets of dead rows small
      }
    }
#pragma unroll
    for (int p = 0; p < B_PASS; ++p)
      u_boff[p] = b_off[p] >= 0 ? ((unsigned)b_off[p] + (unsigned)(kq * EPV)) * (unsigned)ES : 0x80000000u;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  // X3: the bf16 MFMA's adder TRUNCATES toward minus infinity where the fp32 one rounds to nearest (measured, tools/x3_bias_check.py:
  // mean signed error -3.8e-8 of mean |y| on a 64-channel 3x3 layer against 1e-10 on the fp32 pipe; the l2 error is the same).  A
  // bias is what an ill-conditioned sum downstream amplifies (the stem's weight gradient, tests/test_gpu_suites.py), so odd K-tiles
  // are staged with the sign of A flipped and accumulate into a second set: y = acc - acc2, both sets biased the same way.
  constexpr int NACC2 = X3 ? TM : 0;
  f32x16 acc2[NACC2 > 0 ? NACC2 : 1][TN];
  if constexpr (X3) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc2[i][j][v] = 0.f;
  }

  // Two register stages (S0, S1): the global loads of K-tile kt+2 are issued before the MFMA phase of tile kt and are only
  // waited for one whole iteration later (just before they are written to LDS), so two tiles' loads are in flight per
  // wave.  Loads are UNCONDITIONAL: out-of-range taps / rows / K tails read a 16-byte block of zeros in global memory
  // instead (address select BEFORE the load, nothing to fix up after it).  With exec-masked loads the compiler cannot
  // count outstanding loads and falls back to s_waitcnt vmcnt(0); with a select after the load it waits for the data in
  // the iteration that issued it -- either way the two stages would serialise.
  f32x4 ra0[A_PASS], rb0[B_PASS], ra1[A_PASS], rb1[B_PASS];
  const int nkt = (cK + BKE - 1) / BKE;
  // all three as integers: the zero block's address goes through an empty asm so the compiler cannot tell it is a known
  // global (it would turn  *(ok ? p : zero)  into  ok ? *p : 0  again), and the loads are typed global explicitly
  typedef const __attribute__((address_space(1))) f32x4* gvec_t;
  const uint64_t xb = reinterpret_cast<uint64_t>(a.x);
  const uint64_t wb = reinterpret_cast<uint64_t>(a.w);
  uint64_t zp = reinterpret_cast<uint64_t>(&g_zero16[0]);
  asm volatile("" : "+s"(zp));

  if constexpr (!UNI) __syncthreads();  // tap table visible

  __amdgpu_buffer_rsrc_t rsrc_x, rsrc_w, rsrc_x2;
  if constexpr (UNI) {
    rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)a.w_bytes, 0x00020000);
    if constexpr (UP) rsrc_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x2), 0, (int)a.x2_bytes, 0x00020000);
  }
  int u_tl = 0, u_c0 = 0;                     // uniform path: local tap / byte offset inside the tap of the NEXT tile to load
  int u_ga = 0, u_gb = 0;                     // ... and that tap's (row, column) in the class's tap grid (fused input only)
  if constexpr (UP) {
    u_ga = a.g_t0[cls] / a.g_nx[cls];
    u_gb = a.g_t0[cls] - u_ga * a.g_nx[cls];
  }
  const int u_cend = a.ci * ES;
  const int u_ca = UP ? a.up_ca * ES : 0;     // bytes of a virtual pixel that come from the up-sampled source
  auto load_tile = [&](int kt, f32x4* ra, f32x4* rb) {
    if constexpr (UNI) {
      // tiles are requested in order 0, 1, 2, ...: scalar bookkeeping instead of a division per K-step.  A row's gather
      // offset only changes with the tap (and, fused input, with the source): it is recomputed then (a uniform branch) and
      // the position inside the tap travels in the load's SCALAR offset -- no vector ALU work per K-step at all.
      const int tl = u_tl, c0 = u_c0;
      // (dy, dx) of this tap from the grid description: a byte read of a.dy[] / a.dx[] would be a vector memory round trip
      const int dyt = UP ? a.g_dy0[cls] + a.g_sy[cls] * u_ga : 0, dxt = UP ? a.g_dx0[cls] + a.g_sx[cls] * u_gb : 0;
      u_c0 += BKE * ES;
      if (u_c0 == u_cend) {
        u_c0 = 0;
        ++u_tl;
        if constexpr (UP) {
          if (++u_gb == a.g_nx[cls]) {
            u_gb = 0;
            ++u_ga;
          }
        }
      }
      if constexpr (UP) {
        if (c0 == 0) {                        // new tap, channels [0, up_ca): nearest-x2 source, pixel (iy >> 1, ix >> 1)
          const int w2 = a.wi >> 1;
#pragma unroll
          for (int p = 0; p < A_PASS; ++p) {
            const int pix = (a_base[p] >> 2) + ((a_iy[p] + dyt) >> 1) * w2 + ((a_ix[p] + dxt) >> 1);
            u_acur[p] = ((unsigned)pix * (unsigned)u_ca + (unsigned)(kq * 16)) | ((u_ainv[p] >> tl) << 31);
          }
        } else if (c0 == u_ca) {              // same tap, channels [up_ca, ci): the skip tensor at full resolution
#pragma unroll
          for (int p = 0; p < A_PASS; ++p) {
            const int pix = a_base[p] + (a_iy[p] + dyt) * a.wi + (a_ix[p] + dxt);
            u_acur[p] = ((unsigned)pix * (unsigned)(u_cend - u_ca) + (unsigned)(kq * 16)) | ((u_ainv[p] >> tl) << 31);
          }
        }
      } else {
        if (c0 == 0) {
          const unsigned s_x = (unsigned)a.tap_xoff[toff + tl];
#pragma unroll
          for (int p = 0; p < A_PASS; ++p) u_acur[p] = (u_aoff[p] + s_x) + ((u_ainv[p] >> tl) << 31);
        }
      }
      const bool second = UP && c0 >= u_ca;
      const int s_a = second ? c0 - u_ca : c0;
      __amdgpu_buffer_rsrc_t rs = rsrc_x;
      if constexpr (UP) rs = second ? rsrc_x2 : rsrc_x;      // scalar select of the descriptor
#pragma unroll
      for (int p = 0; p < A_PASS; ++p)
        ra[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)u_acur[p], s_a, 0));
      const int s_w = a.tap_woff[toff + tl] + c0;
#pragma unroll
      for (int p = 0; p < B_PASS; ++p)
        rb[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)u_boff[p], s_w, 0));
      return;
    }
    const int kk = kt * BKE + kq * EPV;
    const bool kvalid = kk < cK;
    int t = (int)(((float)kk + 0.5f) * a.inv_ci);
    const int c = kk - t * a.ci;
    t = t < cnt ? t : cnt - 1;
    t = (t < 0 ? 0 : t) + toff;
    const int dyt = taps[t], dxt = taps[64 + t], wtt = taps[128 + t];
#pragma unroll
    for (int p = 0; p < A_PASS; ++p) {
      const int iy = a_iy[p] + dyt, ix = a_ix[p] + dxt;
      const bool ok = kvalid && (unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi;
      const size_t off = (size_t)(a_base[p] + iy * a.wi + ix) * (size_t)a.ci + (size_t)c;
      ra[p] = *reinterpret_cast<gvec_t>(ok ? xb + off * ES : zp);
    }
#pragma unroll
    for (int p = 0; p < B_PASS; ++p) {
      const bool ok = kvalid && b_off[p] >= 0;
      const size_t off = (size_t)b_off[p] + (size_t)(wtt * a.ci + c);
      rb[p] = *reinterpret_cast<gvec_t>(ok ? wb + off * ES : zp);
    }
  };
  // X3: this thread's 8-byte slot of a row (4 K-values of the 16-byte unit kq >> 1) and a lane's fragment row
  const int x3_woff = lrow * 64 + ((((kq >> 1) ^ ((lrow >> 2) & 3))) << 4) + (kq & 1) * 8;
  const int x3_swz = (lr >> 2) & 3;
  auto store_tile = [&](int buf, const f32x4* ra, const f32x4* rb) {
    if constexpr (X3) {
      char* Ad = smem_raw + buf * X3_BUF + x3_woff;
      char* Bd = Ad + 3 * BM * 64;
#pragma unroll
      for (int p = 0; p < A_PASS; ++p) {
        u32x2 q0, q1, q2;
        split3_4(ra[p], q0, q1, q2);
        if (buf) {             // odd K-tiles live in buffer 1: -A (see acc2)
          q0 ^= 0x80008000u;
          q1 ^= 0x80008000u;
          q2 ^= 0x80008000u;
        }
        *reinterpret_cast<u32x2*>(Ad + p * 32 * 64) = q0;
        *reinterpret_cast<u32x2*>(Ad + BM * 64 + p * 32 * 64) = q1;
        *reinterpret_cast<u32x2*>(Ad + 2 * BM * 64 + p * 32 * 64) = q2;
      }
#pragma unroll
      for (int p = 0; p < B_PASS; ++p) {
        u32x2 q0, q1, q2;
        split3_4(rb[p], q0, q1, q2);
        *reinterpret_cast<u32x2*>(Bd + p * 32 * 64) = q0;
        *reinterpret_cast<u32x2*>(Bd + BN * 64 + p * 32 * 64) = q1;
        *reinterpret_cast<u32x2*>(Bd + 2 * BN * 64 + p * 32 * 64) = q2;
      }
      return;
    }
    float* Ad = As + buf * BM * LDS_LD;
    float* Bd = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int p = 0; p < A_PASS; ++p)
      *reinterpret_cast<f32x4*>(Ad + (lrow + 32 * p) * LDS_LD + kq * 4) = ra[p];
#pragma unroll
    for (int p = 0; p < B_PASS; ++p) *reinterpret_cast<f32x4*>(Bd + (lrow + 32 * p) * LDS_LD + kq * 4) = rb[p];
  };
  auto mfma_tile = [&](int buf) {
    if constexpr (X3) {
      const char* Ac = smem_raw + buf * X3_BUF + (wm + lr) * 64;
      const char* Bc = smem_raw + buf * X3_BUF + 3 * BM * 64 + (wn + lr) * 64;
#pragma unroll
      for (int s = 0; s < 2; ++s) {               // two 16-deep MFMA steps per 32-element K-tile: k = 16 s + 8 lh + j
        const int uo = ((2 * s + lh) ^ x3_swz) << 4;
        u32x4 af[3][TM], bf[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
          for (int i = 0; i < TM; ++i) af[pl][i] = *reinterpret_cast<const u32x4*>(Ac + pl * BM * 64 + i * 32 * 64 + uo);
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const u32x4*>(Bc + pl * BN * 64 + j * 32 * 64 + uo);
        }
        // smallest terms first (piece p of A x piece q of B, p + q <= 2); consecutive MFMAs go to different accumulators
#pragma unroll
        for (int pq = 2; pq >= 0; --pq)
#pragma unroll
          for (int pa = 0; pa <= pq; ++pa)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                if (buf)
                  acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[pa][i]),
                                                                       __builtin_bit_cast(bf16x8, bf[pq - pa][j]), acc2[i][j], 0, 0, 0);
                else
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[pa][i]),
                                                                      __builtin_bit_cast(bf16x8, bf[pq - pa][j]), acc[i][j], 0, 0, 0);
              }
      }
      return;
    }
    const float* Ac = As + buf * BM * LDS_LD + (wm + lr) * LDS_LD + lh * 4;
    const float* Bc = Bs + buf * BN * LDS_LD + (wn + lr) * LDS_LD + lh * 4;
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * LDS_LD + s * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * LDS_LD + s * 8);
      if constexpr (BF) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][k2], bf[j][k2], acc[i][j], 0, 0, 0);
      }
    }
  };

  if (a.timeline) tl1 = wall_clock64();
  // invariant at the top of a pair: tile kt sits in LDS buffer 0, tile kt+1 (if any) is in flight in stage S1
  int kt = 0;
  if (nkt > 0) {
    load_tile(0, ra0, rb0);
    if (nkt > 1) load_tile(1, ra1, rb1);
    store_tile(0, ra0, rb0);
  }
  // enter the steady-state loop with no load outstanding: the compiler's wait-count bookkeeping merges the loop-entry
  // state into the loop header, and loads left pending here (in prologue registers) cost a vmcnt wait at the top of
  // EVERY iteration
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
  __syncthreads();
  for (; kt + 3 < nkt; kt += 2) {
    load_tile(kt + 2, ra0, rb0);
    mfma_tile(0);
    store_tile(1, ra1, rb1);
    __syncthreads();
    load_tile(kt + 3, ra1, rb1);
    mfma_tile(1);
    store_tile(0, ra0, rb0);
    __syncthreads();
  }
  const int rem = nkt - kt;   // 0 (nkt == 0), 1, 2 or 3 tiles left
  if (rem == 3) {
    load_tile(kt + 2, ra0, rb0);
    mfma_tile(0);
    store_tile(1, ra1, rb1);
    __syncthreads();
    mfma_tile(1);
    store_tile(0, ra0, rb0);
    __syncthreads();
    mfma_tile(0);
  } else if (rem == 2) {
    mfma_tile(0);
    store_tile(1, ra1, rb1);
    __syncthreads();
    mfma_tile(1);
  } else if (rem == 1) {
    mfma_tile(0);
  }
  __syncthreads();   // every wave is done with the LDS tiles (the statistics epilogue reuses them)
  if (a.timeline) tl2 = wall_clock64();
  if constexpr (X3) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] -= acc2[i][j];
  }

  // ---- epilogue: D[i][j] reg v of lane (lr, lh) = C[row = (v&3) + 8*(v>>2) + 4*lh][col = lr]
  const int ccy = a.cy[cls], ccx = a.cx[cls];
  // split output: this block's channel tile lies entirely on one side of split_n (block-uniform)
  void* yb = a.y;
  int yld = a.co, nsub = 0;
  if (a.split_n > 0) {
    if (n0 >= a.split_n) {
      yb = a.y2;
      yld = a.co - a.split_n;
      nsub = a.split_n;
    } else {
      yld = a.split_n;
    }
  }
  float ssum[TN], ssq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) ssum[j] = ssq[j] = 0.f;
  // Fast path (fp32, dense output, tile entirely inside the output, plain store): one 32-bit offset per 32x32 block, the
  // 16 row displacements go into the store's scalar offset -- no per-element address arithmetic or bounds tests.  The
  // epilogue's VALU work matters: a K = 576 layer spends as many VALU cycles here as in a fifth of its MFMAs.
  bool fast_epi = false;
  if constexpr (!BF) {
    fast_epi = a.dense_out && !a.atomic_out && !a.accumulate && a.residual == nullptr && m0 + BM <= cM && n0 + BN <= a.co &&
               (long long)cM * a.co * 4 < (1LL << 31);
  }
  bool lds_epi = false, stats_done = false;
  if constexpr (BF) {
    lds_epi = !a.atomic_out && !a.accumulate && a.residual == nullptr && !a.out_f32 && (a.co & 7) == 0 && (yld & 7) == 0 &&
              (nsub & 7) == 0 && (reinterpret_cast<uintptr_t>(yb) & 15) == 0;
  }
  if (fast_epi) {
    __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(yb, 0, cM * yld * 4, 0x00020000);
    const int row_bytes = yld * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn + j * 32 + lr;
        const int voff = ((m0 + wm + i * 32 + 4 * lh) * yld + (n - nsub)) * 4;
        // uniform variants so that the common launches carry no dead per-element work: dgrad feeding a BatchNorm backward
        // (store + that layer's two reductions), dgrad / plain conv (store only), conv feeding BatchNorm (statistics), and
        // the general one (bias, activation)
        if (a.bnb_y != nullptr) {
          __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bnb_y), 0, cM * a.co * 4, 0x00020000);
          float yv[16];
#pragma unroll
          for (int v = 0; v < 16; ++v)
            yv[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_p, voff, ((v & 3) + 8 * (v >> 2)) * row_bytes, 0));
          const float mean = a.bnb_mean[n], rstd = a.bnb_rstd[n];
          const float sc = a.bnb_gamma[n] * rstd, sh = a.bnb_beta[n] - mean * sc;      // bn_apply's own coefficients
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const float val = acc[i][j][v];
            const float g = val * act_grad(__builtin_fmaf(yv[v], sc, sh), a.bnb_act, a.bnb_slope);
            ssum[j] += g;
            ssq[j] = __builtin_fmaf(g, (yv[v] - mean) * rstd, ssq[j]);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rsrc_y, voff,
                                                  ((v & 3) + 8 * (v >> 2)) * row_bytes, 0);
          }
        } else if (a.bias == nullptr && a.act == UDASEG_ACT_NONE && a.stats == nullptr) {
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const float val = acc[i][j][v];   // (bit_cast straight from the vector element stored element 0 sixteen times)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rsrc_y, voff,
                                                  ((v & 3) + 8 * (v >> 2)) * row_bytes, 0);
          }
        } else if (a.bias == nullptr && a.act == UDASEG_ACT_NONE) {
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const float val = acc[i][j][v];
            ssum[j] += val;
            ssq[j] = __builtin_fmaf(val, val, ssq[j]);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rsrc_y, voff,
                                                  ((v & 3) + 8 * (v >> 2)) * row_bytes, 0);
          }
        } else {
          const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            float val = acc[i][j][v] + bv;
            ssum[j] += val;
            ssq[j] += val * val;
            val = act_apply(val, a.act, a.slope);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rsrc_y, voff,
                                                  ((v & 3) + 8 * (v >> 2)) * row_bytes, 0);
          }
        }
      }
  } else if (lds_epi) {
    // bf16 output through LDS: the accumulator layout gives a lane ONE column, so direct stores are 2-byte scalars with
    // per-element index arithmetic -- at bf16 MFMA rates that epilogue was 85 % of a K = 576 layer (35 of 41 us, r02 probe).
    // Here: bias / statistics / activation on the fp32 accumulators, neighbouring lanes swap one value (DPP) so that every
    // lane owns a (column pair, row) dword, the tile is staged row-major in the free K-loop LDS and leaves as 16-byte
    // row-contiguous stores; the output pixel of a row (strided data-gradient classes) is computed once per row.
    if constexpr (BF) {
      constexpr int TS = BN / 2 + 4;                 // dwords per staged row: rows r, r+1, r+4, r+5 of one store spread over the banks
      static_assert(BM * TS <= 2 * (BM + BN) * LDS_LD, "staged output tile fits the K-loop buffers");
      unsigned* T = reinterpret_cast<unsigned*>(As);
      const bool inside = m0 + BM <= cM;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn + j * 32 + lr;
          const float bv = (a.bias != nullptr && n < a.co) ? a.bias[a.cmod ? n % a.cmod : n] : 0.f;
          const int tr0 = wm + i * 32 + 4 * lh;
          const int cd = (wn + j * 32 + lr) >> 1;
#pragma unroll
          for (int v = 0; v < 16; v += 2) {
            const int tr = tr0 + (v & 3) + 8 * (v >> 2);
            float w0 = acc[i][j][v] + bv, w1 = acc[i][j][v + 1] + bv;
            if (a.stats != nullptr && a.bnb_y == nullptr) {   // from the fp32 accumulator, before rounding; rows past the class are no outputs
              const float q0 = (inside || m0 + tr < cM) ? w0 : 0.f, q1 = (inside || m0 + tr + 1 < cM) ? w1 : 0.f;
              ssum[j] += q0 + q1;
              ssq[j] = __builtin_fmaf(q0, q0, __builtin_fmaf(q1, q1, ssq[j]));
            }
            w0 = act_apply(w0, a.act, a.slope);
            w1 = act_apply(w1, a.act, a.slope);
            const float send = (lr & 1) ? w0 : w1;
            const float recv = __shfl_xor(send, 1, 64);
            const float lo = (lr & 1) ? recv : w0, hi = (lr & 1) ? w1 : recv;
            const unsigned d = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) |
                               ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
            T[(tr + (lr & 1)) * TS + cd] = d;
          }
        }
      __syncthreads();
      constexpr int CH = BN / 8, RPP = 256 / CH;     // 16-byte chunks per row, rows per pass
      const int ch = tid % CH;
      const int n = n0 + ch * 8;
      // BatchNorm-backward sums of the layer behind (IgemmArgs::bnb_*): formed here, where a thread holds eight consecutive
      // channels of a row -- g from the bf16-ROUNDED gradient, exactly what the stand-alone bn_bwd_reduce_bf16 would read back
      const bool bnb = a.bnb_y != nullptr;
      float b_sc[8], b_sh[8], b_mu[8], b_rs[8], s1[8], s2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[e] = s2[e] = b_sc[e] = b_sh[e] = b_mu[e] = b_rs[e] = 0.f;
      if (bnb && n < a.co) {
        const int nl = a.cmod ? n % a.cmod : n;      // a chunk of 8 never straddles a fold (cmod is a multiple of 8)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          b_mu[e] = a.bnb_mean[nl + e];
          b_rs[e] = a.bnb_rstd[nl + e];
          b_sc[e] = a.bnb_gamma[nl + e] * b_rs[e];
          b_sh[e] = a.bnb_beta[nl + e] - b_mu[e] * b_sc[e];
        }
      }
      if (n < a.co) {
#pragma unroll
        for (int r = tid / CH; r < BM; r += RPP) {
          const int m = m0 + r;
          if (m < cM) {
            size_t pix = (size_t)m;
            if (!a.dense_out) {
              const int t1 = fast_div(m, cJX, cinv_jx);
              const int jx = m - t1 * cJX;
              const int ni = fast_div(t1, cJY, cinv_jy);
              const int jy = t1 - ni * cJY;
              pix = ((size_t)ni * a.ho + (size_t)(ccy + a.sy_o * jy)) * a.wo + (size_t)(ccx + a.sx_o * jx);
            }
            const f32x4 d = *reinterpret_cast<const f32x4*>(T + r * TS + ch * 4);
            *reinterpret_cast<f32x4*>(static_cast<__bf16*>(yb) + pix * (size_t)yld + (size_t)(n - nsub)) = d;
            if (bnb) {     // host-checked: dense output, no split -- the producer's y has this tensor's geometry
              const f32x4 yv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const __bf16*>(a.bnb_y) + (size_t)m * a.co + n);
              const bf16x8 dh = __builtin_bit_cast(bf16x8, d), yh = __builtin_bit_cast(bf16x8, yv);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float yy = (float)yh[e];
                const float g = (float)dh[e] * act_grad(yy * b_sc[e] + b_sh[e], a.bnb_act, a.bnb_slope);
                s1[e] += g;
                s2[e] = __builtin_fmaf(g, (yy - b_mu[e]) * b_rs[e], s2[e]);
              }
            }
          }
        }
      }
      if (bnb) {
        // fold the RPP row-threads of a channel chunk through LDS (behind the staged tile), one f64 atomic per (channel, sum)
        float* red = reinterpret_cast<float*>(T + BM * TS);       // [2][RPP][BN]
        static_assert(BM * TS + 2 * RPP * BN <= 2 * (BM + BN) * LDS_LD, "reduction scratch fits behind the staged tile");
        const int rr = tid / CH;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          red[rr * BN + ch * 8 + e] = s1[e];
          red[RPP * BN + rr * BN + ch * 8 + e] = s2[e];
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.co) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll 8
          for (int q = 0; q < RPP; ++q) {
            t1 += red[q * BN + tid];
            t2 += red[RPP * BN + q * BN + tid];
          }
          const int cl = a.cmod ? a.cmod : a.co, nl = (n0 + tid) % cl;
          double* rep = a.stats + (size_t)(blockIdx.x % STATS_REPLICAS) * 2 * cl;
          atomicAdd(rep + nl, (double)t1);
          atomicAdd(rep + cl + nl, (double)t2);
        }
        stats_done = true;
      }
    }
  } else
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int m = m0 + wm + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
      if (m >= cM) continue;
      size_t pix;
      if (a.dense_out) {
        pix = (size_t)m;
      } else {
        const int t1 = fast_div(m, cJX, cinv_jx);
        const int jx = m - t1 * cJX;
        const int ni = fast_div(t1, cJY, cinv_jy);
        const int jy = t1 - ni * cJY;
        pix = ((size_t)ni * a.ho + (size_t)(ccy + a.sy_o * jy)) * a.wo + (size_t)(ccx + a.sx_o * jx);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn + j * 32 + lr;
        if (n < a.co) {
          const size_t o = pix * (size_t)yld + (size_t)(n - nsub);
          if constexpr (!BF) {
            float* dst = static_cast<float*>(yb) + o;
            if (a.atomic_out) {
              atomicAdd(dst, acc[i][j][v]);
            } else {
              float val = acc[i][j][v];
              if (a.bias) val += a.bias[n];
              ssum[j] += val;          // BN statistics see the conv output (bias included), before any activation
              ssq[j] += val * val;
              if (a.residual) val += static_cast<const float*>(a.residual)[o];
              val = act_apply(val, a.act, a.slope);
              if (a.accumulate) val += *dst;
              *dst = val;
            }
          } else {
            float val = acc[i][j][v];
            if (a.bias) val += a.bias[a.cmod ? n % a.cmod : n];
            ssum[j] += val;            // statistics from the fp32 accumulator, before rounding to bf16
            ssq[j] += val * val;
            if (a.residual) val += (float)static_cast<const __bf16*>(a.residual)[o];
            val = act_apply(val, a.act, a.slope);
            if (a.out_f32) {
              float* dst = static_cast<float*>(yb) + o;
              if (a.accumulate) val += *dst;
              *dst = val;
            } else {
              __bf16* dst = static_cast<__bf16*>(yb) + o;
              if (a.accumulate) val += (float)*dst;
              *dst = (__bf16)val;
            }
          }
        }
      }
    }
  }
  // (bnb_y set but neither bnb epilogue ran -- fp32 off the fast path -- must not pass plain sum(out) / sum(out^2) off as
  // the BatchNorm-backward sums: the host refuses such launches, this keeps a future host bug from being silent)
  if (a.stats && !stats_done && (a.bnb_y == nullptr || fast_epi)) {
    // fold the two half-waves (same column), then the WAVES_M waves that share a column through LDS (the K loop is
    // over: its tiles are free), then one f64 atomic per (column, statistic) per block into replica blockIdx % R.
    float* red = As;  // [2][4 waves][TN][32]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float s1 = ssum[j] + __shfl_xor(ssum[j], 32, 64), s2 = ssq[j] + __shfl_xor(ssq[j], 32, 64);
      if (lh == 0) {
        red[(wave * TN + j) * 32 + lr] = s1;
        red[4 * TN * 32 + (wave * TN + j) * 32 + lr] = s2;
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.co) {
      const int wc = tid / WN, jj = (tid % WN) / 32, l = tid % 32;  // wave column, tile, lane of this output column
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int wr = 0; wr < WAVES_M; ++wr) {
        const int w = wr * WAVES_N + wc;
        s1 += red[(w * TN + jj) * 32 + l];
        s2 += red[4 * TN * 32 + (w * TN + jj) * 32 + l];
      }
      const int cl = a.cmod ? a.cmod : a.co, nl = (n0 + tid) % cl;
      double* rep = a.stats + (size_t)(blockIdx.x % STATS_REPLICAS) * 2 * cl;
      atomicAdd(rep + nl, (double)s1);
      atomicAdd(rep + cl + nl, (double)s2);
    }
  }
  if (a.timeline && tid == 0) {
    unsigned long long* t = a.timeline + (size_t)blockIdx.x * 6;
    t[0] = tl0; t[1] = tl1; t[2] = tl2; t[3] = wall_clock64();
    t[4] = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    t[5] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
  }
}

// ------------------------------------------------------------------------------------------------- host side

unsigned long long* g_timeline = nullptr;     // also read by conv_halo_bf16.hip
int g_timeline_blocks = 0;

template <int BM, int BN, int WAVES_M, int WAVES_N, bool BF, bool UNI, bool UP, bool X3 = false>
static int launch_cfg_t(const IgemmArgs& a, hipStream_t s) {
  static std::atomic<bool> attr_done{false};
  constexpr int lds = igemm_lds_bytes<BM, BN, X3>();
  auto kern = conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, BF, UNI, UP, X3>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv_igemm)");
    attr_done = true;
  }
  IgemmArgs b = a;
  const int nt = cdiv(a.co, BN);
  double flops = 0.0;
  b.tile_begin[0] = 0;
  for (int c = 0; c < a.nclass; ++c) {
    b.tile_begin[c + 1] = b.tile_begin[c] + cdiv(a.M[c], BM) * nt;
    flops += 2.0 * (double)a.M[c] * a.co * a.K[c];
  }
  for (int c = a.nclass; c < NCLS; ++c) b.tile_begin[c + 1] = b.tile_begin[a.nclass];
  if (b.tile_begin[a.nclass] == 0) return UDASEG_OK;
  if (a.split_n > 0 && a.split_n % BN != 0) {
    set_error("conv_igemm: split output at channel %d is not a multiple of the %d-wide tile", a.split_n, BN);
    return UDASEG_E_UNSUPPORTED;
  }
  b.timeline = (g_timeline && b.tile_begin[a.nclass] <= g_timeline_blocks) ? g_timeline : nullptr;
  dim3 grid((unsigned)b.tile_begin[a.nclass]), block(256);
  constexpr int tile_id = (BM == 128 && BN == 128) ? 0 : (BM == 128 && BN == 64) ? 1 : (BM == 64) ? 2 : 3;
  int kid = UP ? 20 + tile_id : (UNI ? 14 + tile_id : tile_id);
  if constexpr (BF || X3) {      // bf16 / X3 instantiations report under their own rocprofv3 symbol (round 2 lumped them into one id)
    static std::atomic<int> bf_kid{-1};
    if (bf_kid < 0) {
      char nm[96];
      snprintf(nm, sizeof(nm), "conv_igemm_kernel<%d, %d, %d, %d, %s, %s, %s%s>", BM, BN, WAVES_M, WAVES_N, BF ? "true" : "false",
               UNI ? "true" : "false", UP ? "true" : "false", X3 ? ", true" : "");
      bf_kid = kprof_id(nm);
    }
    kid = bf_kid;
  }
  hipEvent_t ev = kprof_begin(s);
  hipLaunchKernelGGL(kern, grid, block, lds, s, b);
  kprof_end(kid, ev, s, flops);
  UDASEG_LAUNCH_CHECK("conv_igemm launch");
  return UDASEG_OK;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_cfg(const IgemmArgs& a, hipStream_t s) {
  if (a.up_ca > 0) {
    if (!a.uniform) {
      set_error("conv_igemm: the fused upsample+concat input needs channel counts that are multiples of the K-tile");
      return UDASEG_E_UNSUPPORTED;
    }
    if (a.bf16) return launch_cfg_t<BM, BN, WAVES_M, WAVES_N, true, true, true>(a, s);
    return launch_cfg_t<BM, BN, WAVES_M, WAVES_N, false, true, true>(a, s);
  }
  if (a.uniform) {
    if (a.bf16) return launch_cfg_t<BM, BN, WAVES_M, WAVES_N, true, true, false>(a, s);
    return launch_cfg_t<BM, BN, WAVES_M, WAVES_N, false, true, false>(a, s);
  }
  if (a.bf16) return launch_cfg_t<BM, BN, WAVES_M, WAVES_N, true, false, false>(a, s);
  return launch_cfg_t<BM, BN, WAVES_M, WAVES_N, false, false, false>(a, s);
}

static int uniform_off() { return opt_get(UDASEG_OPT_GENERIC_GATHER); }      // 1: every layer on the generic gather loop

// Decide whether a fully described launch can take the uniform-tap loop and fill its tables.
static void finish_args(IgemmArgs& a, long long x_elems, long long w_elems) {
  const int es = a.bf16 ? 2 : 4, bke = BK * 4 / es;
  int total = 0;
  for (int c = 0; c < a.nclass; ++c) total = a.tap_off[c] + a.ntaps[c] > total ? a.tap_off[c] + a.ntaps[c] : total;
  a.uniform = 0;
  a.x_bytes = a.w_bytes = 0;
  if ((uniform_off() && a.up_ca == 0) || a.ci % bke != 0 || a.up_ca % bke != 0 || total > 32 || x_elems * es > (1LL << 30) ||
      w_elems * es > (1LL << 30))
    return;
  for (int c = 0; c < a.nclass; ++c) {
    if (a.ntaps[c] > 32 || a.K[c] != a.ntaps[c] * a.ci) return;
    // the class's taps must be the grid run its description claims (they are, for every launch the library builds)
    if (a.g_nx[c] < 1 || a.g_ny[c] < 1 || a.g_nx[c] > 8 || a.g_ny[c] > 8 || a.g_t0[c] < 0 || a.g_t0[c] + a.ntaps[c] > a.g_nx[c] * a.g_ny[c])
      return;
    for (int tl = 0; tl < a.ntaps[c]; ++tl) {
      const int gi = a.g_t0[c] + tl, ga = gi / a.g_nx[c], gb = gi % a.g_nx[c], t = a.tap_off[c] + tl;
      if (a.dy[t] != a.g_dy0[c] + a.g_sy[c] * ga || a.dx[t] != a.g_dx0[c] + a.g_sx[c] * gb) return;
    }
  }
  for (int t = 0; t < total; ++t) {
    a.tap_xoff[t] = (a.dy[t] * a.wi + a.dx[t]) * a.ci * es;
    a.tap_woff[t] = a.wt[t] * a.ci * es;
  }
  a.x_bytes = (unsigned)(x_elems * es);
  a.w_bytes = (unsigned)(w_elems * es);
  a.uniform = 1;
}

static int tile_override() {
  // tuning aid: UDASEG_IGEMM_TILE = 1 (128x128) | 2 (128x64) | 3 (64x64) | 4 (128x32); unset/0 = heuristic
  return opt_get(UDASEG_OPT_IGEMM_TILE);
}

// fp32 launches on the bf16 matrix pipe (conv_igemm_kernel X3): > 32 produced channels.  UDASEG_IGEMM_X3 = 0 (off) | 1 (64 x 64 tile)
// | 2 (128 x 64) | 3 (128 x 128); UDASEG_F32_SPLIT=0 switches every three-term kernel off.  Per call, us, fp32 pipe / X3 64 x 64 /
// X3 128 x 64 (bench.py --layer-table, cfg 2, profiles/r04_igemm_x3.txt): 7x7 / stride 2 stem 164 / 130 / 126; 3x3 / stride 2 forward
// 58 / 53 / 52, 58 / 50 / 50, 62 / 54 / 55; their data gradients (four parity classes, short K loops) 82 / 77 / 90, 70 / 65 / 73,
// 77 / 69 / 81; 1x1 / stride 2 unchanged (24-38 us, latency).  The split is VALU work every block repeats for its A rows AND the weights
// (88 vector instructions per thread and K-tile beside 12 MFMAs per wave): these layers gain 10-25 %, not the 2.7x of the matrix rate.
static int x3_tile() {
  int v = opt_get(UDASEG_OPT_IGEMM_X3);
  if (v < 0 || v > 3) v = 1;
  return f32_split_enabled() ? v : 0;
}
template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_cfg_x3(const IgemmArgs& a, hipStream_t s) {
  if (a.up_ca > 0) {
    if (!a.uniform) {
      set_error("conv_igemm: the fused upsample+concat input needs channel counts that are multiples of the K-tile");
      return UDASEG_E_UNSUPPORTED;
    }
    return launch_cfg_t<BM, BN, WAVES_M, WAVES_N, false, true, true, true>(a, s);
  }
  if (a.uniform) return launch_cfg_t<BM, BN, WAVES_M, WAVES_N, false, true, false, true>(a, s);
  return launch_cfg_t<BM, BN, WAVES_M, WAVES_N, false, false, false, true>(a, s);
}

static int launch_igemm(const IgemmArgs& a, hipStream_t s) {
  long long rows = 0;
  for (int c = 0; c < a.nclass; ++c) rows += a.M[c];
  if (rows <= 0) return UDASEG_OK;
  if (!a.bf16 && a.co > 32 && x3_tile() != 0) {
    const int t = x3_tile();
    if (t == 1) return launch_cfg_x3<64, 64, 2, 2>(a, s);
    if (t == 3) return launch_cfg_x3<128, 128, 2, 2>(a, s);
    return launch_cfg_x3<128, 64, 2, 2>(a, s);
  }
  const long long tiles128 = (rows + 127) / 128;
  switch (tile_override()) {
    case 1: return launch_cfg<128, 128, 2, 2>(a, s);
    case 2: return launch_cfg<128, 64, 2, 2>(a, s);
    case 3: return launch_cfg<64, 64, 2, 2>(a, s);
    case 4: return launch_cfg<128, 32, 4, 1>(a, s);
    default: break;
  }
  // tile choice, re-measured per shape after the uniform-tap loop went in (profiles/r01_tile_ab.txt): with the gather
  // address arithmetic out of the K loop the 64x64 tile (4 blocks per CU) wins or ties on every layer with more than 32
  // output channels, 128x32 on the rest; the 128-row tiles stay selectable through UDASEG_IGEMM_TILE.
  (void)tiles128;
  if (a.co > 32) return launch_cfg<64, 64, 2, 2>(a, s);
  return launch_cfg<128, 32, 4, 1>(a, s);
}

// Common part of the launch description.
static void base_args(IgemmArgs& a, const void* x, const void* w, const float* bias, void* y, int hi, int wi, int ci,
                      int ho, int wo, int co, int tfull, int accumulate, int act, float slope) {
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.hi = hi; a.wi = wi; a.ci = ci; a.ho = ho; a.wo = wo; a.co = co;
  a.tfull = tfull; a.inv_ci = 1.0f / ci;
  a.accumulate = accumulate; a.act = act; a.slope = slope;
  a.atomic_out = 0; a.nclass = 0;
}

// Deep low-resolution layers have too few output tiles to fill 256 CUs (layer4 of r18 at 512^2: 256 tiles of 64x64):
// split the taps of a single-class launch into K-slices that add into the output atomically.  Only for plain conv
// outputs (no bias / activation) -- those are applied once, which a sliced sum cannot do.
static int k_slices(int M, int co, int ntaps, bool plain) {
  if (!plain || ntaps < 2) return 1;
  const long long tiles = (long long)cdiv(M, 64) * cdiv(co, 64);
  if (tiles >= 384) return 1;
  int want = (int)((511 + tiles) / tiles);  // aim at >= 512 blocks
  if (want > 2) want = 2;  // two adders onto a zeroed output commute exactly: the forward stays bitwise reproducible
  if (want > ntaps) want = ntaps;
  return want < 1 ? 1 : want;
}

static int check_desc(const udaseg_conv_desc* d) {
  UDASEG_CHECK_ARG(d != nullptr, "conv desc is NULL");
  UDASEG_CHECK_ARG(d->n > 0 && d->hi > 0 && d->wi > 0 && d->ho > 0 && d->wo > 0, "conv desc: non-positive extent");
  UDASEG_CHECK_ARG(d->ci > 0 && d->co > 0 && d->ci % 4 == 0 && d->co % 4 == 0,
                   "conv desc: channel counts must be positive multiples of 4 (ci=%d co=%d)", d->ci, d->co);
  UDASEG_CHECK_ARG(d->kh > 0 && d->kw > 0 && d->kh * d->kw <= 64, "conv desc: kernel %dx%d unsupported", d->kh, d->kw);
  UDASEG_CHECK_ARG(d->stride >= 1 && d->stride <= 4 && d->pad >= 0 && d->pad < 64, "conv desc: stride/pad unsupported");
  UDASEG_CHECK_ARG(d->ho == (d->hi + 2 * d->pad - d->kh) / d->stride + 1 && d->wo == (d->wi + 2 * d->pad - d->kw) / d->stride + 1,
                   "conv desc: output extent %dx%d inconsistent with input %dx%d k%d s%d p%d", d->ho, d->wo, d->hi, d->wi,
                   d->kh, d->stride, d->pad);
  UDASEG_CHECK_ARG((long long)d->n * d->hi * d->wi * d->ci < (1LL << 31) && (long long)d->n * d->ho * d->wo * d->co < (1LL << 31),
                   "conv desc: tensor exceeds 2^31 elements");
  return UDASEG_OK;
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_debug_set_timeline(void* buffer, int blocks) {
  udaseg::g_timeline = static_cast<unsigned long long*>(buffer);
  udaseg::g_timeline_blocks = buffer ? blocks : 0;
  return UDASEG_OK;
}

extern "C" double udaseg_conv_flops(const udaseg_conv_desc* d) {
  if (!d) return 0.0;
  return 2.0 * (double)d->n * d->ho * d->wo * (double)d->co * (double)d->ci * d->kh * d->kw;
}

// skip / up_ca: the fused nearest-x2-upsample + concat input (IgemmArgs::x2): x is then the HALF-resolution tensor
// [n][hi/2][wi/2][up_ca], skip the full-resolution [n][hi][wi][ci - up_ca] (NULL when ci == up_ca), d describes the conv on
// the virtual concatenation.
// ------------------------------------------------------------------------------------------------ pixel folding (bf16)
// The <= 32-channel 3x3 / stride 1 / pad 1 layers at full resolution (decoder tail, head) do not fit the uniform-tap loop in
// bf16: a K-tile is 64 elements = 128 bytes, a tap of 16 or 32 channels is 32 or 64.  On the generic loop they ran
// block-overhead-bound (16 384 blocks of 1.2 MFLOP; 150-190 us per 9.66-19 GFLOP layer).  Folding: F = 64 / ci neighbouring
// pixels of an image row ARE one pixel of F * ci = 64 channels in memory ([n][h][w][ci] == [n][h][w / F][F * ci], zero copy),
// and the convolution is again a 3x3 / pad 1 convolution over those units with folded weights
//     W'[(fo, o)][r][qu][(fi, c)] = W[o][r][dx][c],  dx = F * (qu - 1) + fi - fo + 1   if 0 <= dx <= 2, else 0
// (output pixel F * xo + fo reads input pixel F * (xo + qu - 1) + fi).  F x the multiplications, all of them on the uniform
// loop at bf16 MFMA rates, for a layer whose time is its operand traffic.  The folded weights (<= 96 x 9 x 64 bf16) are
// rebuilt per call by a one-block-per-row kernel into the tail of the device's bound scratch; per-channel vectors are
// addressed modulo the logical channel count (IgemmArgs::cmod).  The data gradient is the same thing on the flipped,
// transposed weights (it IS a forward convolution of dy with V[c][r][q][o] = W[o][2 - r][2 - q][c]).
constexpr size_t FOLD_SCRATCH_BYTES = 256 << 10;

static int fold_factor(int kh, int kw, int stride, int pad, int gathered_c, int produced_c, int width, int bf16) {
  if (!bf16 || kh != 3 || kw != 3 || stride != 1 || pad != 1) return 1;
  if (gathered_c != 16 && gathered_c != 32) return 1;
  if (opt_get(UDASEG_OPT_NO_FOLD)) return 1;      // tuning aid / A-B
  const int F = 64 / gathered_c;
  if (width % F != 0 || produced_c % 8 != 0) return 1;
  // F x the multiplications and F x the output columns: measured per layer (r18 8x512^2 bf16, API-level events, folded vs generic
  // loop): 32 -> 16 forward 129 vs 173 us, 16 -> 16 forward 125 vs 141 and data gradient 131 vs 192, 32 -> 32 at 256^2 58 vs 60 /
  // 70 vs 73; but 16 -> 24 forward 215 vs 192 and the gradient that gathers 16 and produces 32 channels 157 vs 124: a fold of
  // four only pays while the folded output still fits one 64-wide column tile
  if (F == 4 && produced_c > 16) return 1;
  if ((size_t)produced_c * F * 9 * 64 * 2 > FOLD_SCRATCH_BYTES) return 1;
  size_t ws = 0;
  if (workspace_ptr(&ws) == nullptr || ws < FOLD_SCRATCH_BYTES) return 1;
  return F;
}

// src: forward weights [co][9][ci] (transposed == 0) or dgrad-packed weights [c_g][9][c_p] read as the flipped forward
// convolution (transposed == 1: gathered channels are the packing's LAST axis).  dst: [F * cp][9][F * cg].
__global__ void fold_weights_bf16_kernel(const __bf16* __restrict__ src, __bf16* __restrict__ dst, int cp, int cg, int F,
                                         int transposed) {
  const int row = blockIdx.x;                 // (fo, o)
  const int fo = row / cp, o = row % cp;
  const int KP = 9 * F * cg;
  for (int i = threadIdx.x; i < KP; i += blockDim.x) {
    const int c = i % cg, fi = (i / cg) % F, t = i / (cg * F);
    const int r = t / 3, qu = t % 3;
    const int dx = F * (qu - 1) + fi - fo + 1;
    __bf16 v = (__bf16)0.f;
    if (dx >= 0 && dx <= 2) {
      if (!transposed) v = src[((size_t)o * 9 + r * 3 + dx) * cg + c];
      else v = src[((size_t)o * 9 + (2 - r) * 3 + (2 - dx)) * cg + c];
    }
    dst[(size_t)row * KP + i] = v;
  }
}

struct BnReduceArgs {   // IgemmArgs::bnb_*
  const float* y;
  const float* mean;
  const float* rstd;
  const float* gamma;
  const float* beta;
  int act;
  float slope;
  double* bsums;
};

// Run the folded problem: gathered tensor g [n][h][w][cg], produced tensor p [n][h][w][cp], weights as fold_weights expects.
static int conv2d_folded(const udaseg_conv_desc* d, int F, const void* g, int cg, const void* wsrc, int transposed, const float* bias,
                         void* p, int cp, int act, float slope, int accumulate, double* stats, const void* residual,
                         hipStream_t st, int out_f32, const BnReduceArgs* bnb);

static int conv2d_fwd_impl(const udaseg_conv_desc* d, const void* x, const void* w, const float* bias, void* y,
                           int act, float slope, int accumulate, double* stats, const void* residual, void* stream,
                           int bf16 = 0, int out_f32 = 0, const void* skip = nullptr, int up_ca = 0, int cmod = 0,
                           const BnReduceArgs* bnb = nullptr) {
  int rc = check_desc(d);
  if (rc) return rc;
  UDASEG_CHECK_ARG(x && w && y, "conv2d_fwd: NULL pointer");
  if (cmod == 0 && up_ca == 0) {
    const int F = fold_factor(d->kh, d->kw, d->stride, d->pad, d->ci, d->co, d->wi, bf16);
    if (F > 1)
      return conv2d_folded(d, F, x, d->ci, w, 0, bias, y, d->co, act, slope, accumulate, stats, residual, as_stream(stream), out_f32,
                           nullptr);
  }
  UDASEG_CHECK_ARG(!bf16 || (d->ci % 8 == 0 && d->co % 8 == 0), "conv2d_fwd(bf16): channels must be multiples of 8 (ci=%d co=%d)",
                   d->ci, d->co);
  if (up_ca > 0) {
    UDASEG_CHECK_ARG(d->stride == 1 && d->hi % 2 == 0 && d->wi % 2 == 0, "conv2d_fwd_upcat: stride 1 and even extents only");
    UDASEG_CHECK_ARG(up_ca <= d->ci && (up_ca == d->ci) == (skip == nullptr),
                     "conv2d_fwd_upcat: up_ca=%d of ci=%d channels, skip %s", up_ca, d->ci, skip ? "given" : "NULL");
    UDASEG_CHECK_ARG(!accumulate && residual == nullptr, "conv2d_fwd_upcat: accumulate / residual are not supported");
  }
  hipStream_t st = as_stream(stream);
  if (!bf16 && d->kh == d->kw && small_conv_applicable(d->kh, d->stride, d->pad, d->ci, d->co)) {
    if (up_ca > 0 && skip != nullptr) {
      set_error("conv2d_fwd_upcat: the small-channel kernel has no skip input (ci=%d)", d->ci);
      return UDASEG_E_UNSUPPORTED;
    }
    prof_begin(0, st);
    rc = launch_small_conv(static_cast<const float*>(x), static_cast<const float*>(w), bias, static_cast<float*>(y), d->n,
                           d->hi, d->wi, d->ci, d->co, 0, accumulate, act, slope, stats, static_cast<const float*>(residual), st,
                           up_ca > 0 ? 1 : 0);
    prof_end(0, st, udaseg_conv_flops(d), 0, d);
    return rc;
  }
  IgemmArgs a = {};
  const int ntaps = d->kh * d->kw;
  base_args(a, x, w, bias, y, d->hi, d->wi, d->ci, d->ho, d->wo, d->co, ntaps, accumulate, act, slope);
  a.sy_o = 1; a.sx_o = 1; a.sy_i = d->stride; a.sx_i = d->stride; a.dense_out = 1;
  for (int r = 0; r < d->kh; ++r)
    for (int q = 0; q < d->kw; ++q) {
      const int t = r * d->kw + q;
      a.dy[t] = (signed char)(r - d->pad);
      a.dx[t] = (signed char)(q - d->pad);
      a.wt[t] = (unsigned char)t;
    }
  const int M = d->n * d->ho * d->wo;
  a.residual = residual;
  a.bf16 = bf16;
  a.out_f32 = out_f32;
  a.cmod = cmod;
  if (bnb) {
    a.bnb_y = bnb->y; a.bnb_mean = bnb->mean; a.bnb_rstd = bnb->rstd; a.bnb_gamma = bnb->gamma; a.bnb_beta = bnb->beta;
    a.bnb_act = bnb->act; a.bnb_slope = bnb->slope;
    stats = bnb->bsums;
  }
  a.x2 = skip;
  a.up_ca = up_ca;
  a.x2_bytes = (unsigned)((long long)d->n * d->hi * d->wi * (d->ci - up_ca) * (bf16 ? 2 : 4));
  const int ns = k_slices(M, d->co, ntaps, !bf16 && bias == nullptr && act == UDASEG_ACT_NONE && residual == nullptr);
  a.stats = (ns == 1) ? stats : nullptr;  // squares of partial sums do not add up: sliced launches take the separate pass
  a.nclass = ns;
  for (int c = 0; c < ns; ++c) {
    const int t0 = (int)((long long)ntaps * c / ns), t1 = (int)((long long)ntaps * (c + 1) / ns);
    a.JY[c] = d->ho; a.JX[c] = d->wo; a.M[c] = M; a.cy[c] = 0; a.cx[c] = 0;
    a.ntaps[c] = t1 - t0; a.K[c] = (t1 - t0) * d->ci; a.tap_off[c] = t0;
    a.inv_jx[c] = 1.0f / d->wo; a.inv_jy[c] = 1.0f / d->ho;
    a.g_dy0[c] = -d->pad; a.g_dx0[c] = -d->pad; a.g_sy[c] = 1; a.g_sx[c] = 1; a.g_ny[c] = d->kh; a.g_nx[c] = d->kw; a.g_t0[c] = t0;
  }
  prof_begin(0, st);
  if (ns > 1) {
    a.atomic_out = 1;
    if (!accumulate) {
      hipError_t e = hipMemsetAsync(y, 0, (size_t)M * d->co * sizeof(float), st);
      if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(conv out)");
    }
  }
  if (up_ca > 0) {
    UDASEG_CHECK_ARG((long long)d->n * d->hi * d->wi * (d->ci - up_ca) * (bf16 ? 2 : 4) <= (1LL << 30),
                     "conv2d_fwd_upcat: skip tensor exceeds 2^30 bytes");
    finish_args(a, (long long)d->n * (d->hi / 2) * (d->wi / 2) * up_ca, (long long)d->co * ntaps * d->ci);
  } else {
    finish_args(a, (long long)d->n * d->hi * d->wi * d->ci, (long long)d->co * ntaps * d->ci);
  }
  rc = launch_igemm(a, st);
  prof_end(0, st, udaseg_conv_flops(d), 0, d);
  if (rc == UDASEG_OK && stats && ns > 1) rc = udaseg_bn_stats(static_cast<const float*>(y), (int64_t)M, d->co, stats, stream);
  return rc;
}

extern "C" int udaseg_conv2d_fwd_bf16(const udaseg_conv_desc* d, const void* x, const void* w, const float* bias,
                                      const void* residual, void* y, int out_f32, int act, float slope, double* stats,
                                      void* stream) {
  UDASEG_CHECK_ARG(!(out_f32 && residual), "conv2d_fwd_bf16: residual with fp32 output is not supported");
  return conv2d_fwd_impl(d, x, w, bias, y, act, slope, 0, stats, residual, stream, 1, out_f32);
}

static int conv2d_folded(const udaseg_conv_desc* d, int F, const void* g, int cg, const void* wsrc, int flip, const float* bias,
                         void* p, int cp, int act, float slope, int accumulate, double* stats, const void* residual,
                         hipStream_t st, int out_f32, const BnReduceArgs* bnb) {
  size_t ws = 0;
  char* base = static_cast<char*>(workspace_ptr(&ws));
  if (base == nullptr || ws < FOLD_SCRATCH_BYTES) {
    set_error("conv (bf16, folded pixels): no scratch bound to the current device");
    return UDASEG_E_BADARG;
  }
  __bf16* wf = reinterpret_cast<__bf16*>(base + ws - FOLD_SCRATCH_BYTES);
  hipLaunchKernelGGL(fold_weights_bf16_kernel, dim3(cp * F), dim3(256), 0, st, static_cast<const __bf16*>(wsrc), wf, cp, cg, F, flip);
  UDASEG_LAUNCH_CHECK("fold_weights_bf16 launch");
  udaseg_conv_desc f = *d;        // the same 3x3 / stride 1 / pad 1 convolution over units of F pixels
  f.wi = d->wi / F; f.wo = d->wo / F; f.ci = cg * F; f.co = cp * F;
  prof_begin(0, st);
  prof_suspend(1);                // FLOPs of the record: the LOGICAL convolution's (the folded launch does F x the multiplies)
  const int rc = conv2d_fwd_impl(&f, g, wf, bias, p, act, slope, accumulate, stats, residual, st, 1, out_f32, nullptr, 0, cp, bnb);
  prof_suspend(0);
  prof_end(0, st, udaseg_conv_flops(d), flip ? 1 : 0, d);
  return rc;
}

extern "C" int udaseg_conv2d_fwd(const udaseg_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                                 int act, float slope, int accumulate, void* stream) {
  return conv2d_fwd_impl(d, x, w, bias, y, act, slope, accumulate, nullptr, nullptr, stream);
}

extern "C" int udaseg_conv2d_fwd_fused(const udaseg_conv_desc* d, const float* x, const float* w, const float* bias,
                                       const float* residual, float* y, int act, float slope, void* stream) {
  return conv2d_fwd_impl(d, x, w, bias, y, act, slope, 0, nullptr, residual, stream);
}

extern "C" int udaseg_conv2d_fwd_bnstats(const udaseg_conv_desc* d, const float* x, const float* w, const float* bias,
                                         float* y, double* stats, void* stream) {
  UDASEG_CHECK_ARG(stats != nullptr, "conv2d_fwd_bnstats: stats is NULL");
  return conv2d_fwd_impl(d, x, w, bias, y, UDASEG_ACT_NONE, 0.f, 0, stats, nullptr, stream);
}

extern "C" int udaseg_conv2d_fwd_upcat(const udaseg_conv_desc* d, const float* a, const float* skip, int ca, const float* w,
                                       const float* bias, float* y, int act, float slope, double* stats, void* stream) {
  UDASEG_CHECK_ARG(ca > 0, "conv2d_fwd_upcat: ca must be positive");
  return conv2d_fwd_impl(d, a, w, bias, y, act, slope, 0, stats, nullptr, stream, 0, 0, skip, ca);
}

extern "C" int udaseg_conv2d_fwd_upcat_bf16(const udaseg_conv_desc* d, const void* a, const void* skip, int ca, const void* w,
                                            const float* bias, void* y, int act, float slope, double* stats, void* stream) {
  UDASEG_CHECK_ARG(ca > 0 && ca % 8 == 0, "conv2d_fwd_upcat_bf16: ca must be a positive multiple of 8");
  return conv2d_fwd_impl(d, a, w, bias, y, act, slope, 0, stats, nullptr, stream, 1, 0, skip, ca);
}


// Can the data gradient of this convolution carry the BatchNorm-backward reductions of the layer behind it?  Only when every
// tile of the launch is whole and plainly stored: stride 1, implicit-GEMM kernel (not the small-channel one), no K-slices,
// pixel and channel counts that are multiples of the tile the launcher picks.
static bool dgrad_bnreduce_ok(const udaseg_conv_desc* d) {
  if (check_desc(d) != UDASEG_OK || d->stride != 1 || tile_override() != 0) return false;
  if (d->kh == d->kw && small_conv_applicable(d->kh, d->stride, d->pad, d->co, d->ci)) return false;
  const long long M = (long long)d->n * d->hi * d->wi;
  // the sums are formed in the kernel's fast epilogue only, which also needs the output to fit a 2 GiB buffer descriptor
  // (fast_epi in conv_igemm_kernel); beyond that the launch would take the generic epilogue, which knows nothing of bnb_*
  if (M * d->ci * 4 >= (1LL << 31)) return false;
  if (k_slices((int)M, d->ci, d->kh * d->kw, true) != 1) return false;
  if (d->ci > 32) return M % 64 == 0 && d->ci % 64 == 0;
  return M % 128 == 0 && d->ci % 32 == 0;
}

// dx2 / split: the data gradient of a convolution over a virtual concatenation lands in two tensors: input channels
// [0, split) in dx [n][hi][wi][split], channels [split, ci) in dx2 [n][hi][wi][ci - split] (IgemmArgs::y2).
static int conv2d_dgrad_impl(const udaseg_conv_desc* d, const void* dy, const void* w_t, void* dx, int accumulate,
                             void* stream, int bf16, void* dx2 = nullptr, int split = 0, const BnReduceArgs* bnb = nullptr) {
  int rc = check_desc(d);
  if (rc) return rc;
  UDASEG_CHECK_ARG(dy && w_t && dx, "conv2d_dgrad: NULL pointer");
  UDASEG_CHECK_ARG(split == 0 || (dx2 != nullptr && split > 0 && split < d->ci && d->stride == 1 && !accumulate),
                   "conv2d_dgrad_split: needs dx2, 0 < split < ci, stride 1, no accumulation");
  UDASEG_CHECK_ARG(!bf16 || (d->ci % 8 == 0 && d->co % 8 == 0), "conv2d_dgrad(bf16): channels must be multiples of 8");
  hipStream_t st = as_stream(stream);
  if (split == 0) {
    // pixel folding: the data gradient as the flipped forward convolution of dy (gathered: co channels, produced: ci)
    const int F = fold_factor(d->kh, d->kw, d->stride, d->pad, d->co, d->ci, d->wi, bf16);
    if (F > 1)
      return conv2d_folded(d, F, dy, d->co, w_t, 1, nullptr, dx, d->ci, UDASEG_ACT_NONE, 0.f, accumulate, nullptr, nullptr, st, 0, bnb);
  }
  const int s = d->stride;
  prof_begin(0, st);
  if (!bf16 && d->kh == d->kw && small_conv_applicable(d->kh, d->stride, d->pad, d->co, d->ci)) {
    if (split > 0) {
      set_error("conv2d_dgrad_split: not available on the small-channel kernel (co=%d ci=%d)", d->co, d->ci);
      return UDASEG_E_UNSUPPORTED;
    }
    // dx = correlation of dy with the flipped taps; w_t is already [ci][9][co]
    rc = launch_small_conv(static_cast<const float*>(dy), static_cast<const float*>(w_t), nullptr, static_cast<float*>(dx), d->n,
                           d->hi, d->wi, d->co, d->ci, 1, accumulate, UDASEG_ACT_NONE, 0.f, nullptr, nullptr, st);
    prof_end(0, st, udaseg_conv_flops(d), 1, d);
    return rc;
  }
  UDASEG_CHECK_ARG(s * s <= NCLS, "conv2d_dgrad: stride %d unsupported (at most %d parity classes)", s, NCLS);
  IgemmArgs a = {};
  // gathered operand: dy [n][ho][wo][co]; output: dx [n][hi][wi][ci]
  base_args(a, dy, w_t, nullptr, dx, d->ho, d->wo, d->co, d->hi, d->wi, d->ci, d->kh * d->kw, accumulate, UDASEG_ACT_NONE, 0.f);
  a.sy_o = s; a.sx_o = s; a.sy_i = 1; a.sx_i = 1; a.dense_out = (s == 1) ? 1 : 0;
  int nc = 0, ntot = 0;
  for (int ph = 0; ph < s; ++ph)
    for (int pw = 0; pw < s; ++pw) {
      const int JY = (d->hi - ph + s - 1) / s, JX = (d->wi - pw + s - 1) / s;
      if (JY <= 0 || JX <= 0) continue;
      const int t0 = ntot;
      int gny = 0, gnx = 0;      // the class's taps form a gny x gnx grid, dy and dx falling by one per step
      for (int r = 0; r < d->kh; ++r) {
        if ((ph + d->pad - r) % s != 0) continue;
        ++gny;
        gnx = 0;
        for (int q = 0; q < d->kw; ++q) {
          if ((pw + d->pad - q) % s != 0) continue;
          ++gnx;
          a.dy[ntot] = (signed char)((ph + d->pad - r) / s);
          a.dx[ntot] = (signed char)((pw + d->pad - q) / s);
          a.wt[ntot] = (unsigned char)(r * d->kw + q);
          ++ntot;
        }
      }
      const int nt = ntot - t0;
      if (nt == 0 && accumulate) continue;  // nothing to add for this parity class (with !accumulate it writes zeros)
      a.JY[nc] = JY; a.JX[nc] = JX; a.M[nc] = d->n * JY * JX; a.cy[nc] = ph; a.cx[nc] = pw;
      a.ntaps[nc] = nt; a.K[nc] = nt * d->co; a.tap_off[nc] = t0;
      a.inv_jx[nc] = 1.0f / JX; a.inv_jy[nc] = 1.0f / JY;
      a.g_ny[nc] = nt ? gny : 1; a.g_nx[nc] = nt ? gnx : 1; a.g_t0[nc] = 0;
      a.g_dy0[nc] = nt ? a.dy[t0] : 0; a.g_dx0[nc] = nt ? a.dx[t0] : 0; a.g_sy[nc] = -1; a.g_sx[nc] = -1;
      ++nc;
    }
  a.nclass = nc;
  a.bf16 = bf16;
  a.y2 = dx2;
  a.split_n = split;
  if (bnb) {
    a.bnb_y = bnb->y; a.bnb_mean = bnb->mean; a.bnb_rstd = bnb->rstd; a.bnb_gamma = bnb->gamma; a.bnb_beta = bnb->beta;
    a.bnb_act = bnb->act; a.bnb_slope = bnb->slope;
    a.stats = bnb->bsums;
  }
  if (s == 1 && nc == 1 && !bf16) {
    // single class: K-slices for the deep layers, as in the forward
    const int ntaps = a.ntaps[0];
    const int ns = k_slices(a.M[0], d->ci, ntaps, true);
    if (ns > 1) {
      for (int c = ns - 1; c >= 0; --c) {
        const int t0 = (int)((long long)ntaps * c / ns), t1 = (int)((long long)ntaps * (c + 1) / ns);
        a.JY[c] = a.JY[0]; a.JX[c] = a.JX[0]; a.M[c] = a.M[0]; a.cy[c] = 0; a.cx[c] = 0;
        a.inv_jx[c] = a.inv_jx[0]; a.inv_jy[c] = a.inv_jy[0];
        a.g_ny[c] = a.g_ny[0]; a.g_nx[c] = a.g_nx[0]; a.g_dy0[c] = a.g_dy0[0]; a.g_dx0[c] = a.g_dx0[0];
        a.g_sy[c] = -1; a.g_sx[c] = -1; a.g_t0[c] = t0;
        a.ntaps[c] = t1 - t0; a.K[c] = (t1 - t0) * d->co; a.tap_off[c] = t0;
      }
      a.nclass = ns;
      a.atomic_out = 1;
      if (!accumulate) {
        const size_t pix = (size_t)d->n * d->hi * d->wi;
        hipError_t e = hipMemsetAsync(dx, 0, pix * (split > 0 ? split : d->ci) * sizeof(float), st);  // fp32 only
        if (e == hipSuccess && split > 0) e = hipMemsetAsync(dx2, 0, pix * (d->ci - split) * sizeof(float), st);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(dgrad out)");
      }
    }
  }
  finish_args(a, (long long)d->n * d->ho * d->wo * d->co, (long long)d->ci * d->kh * d->kw * d->co);
  rc = launch_igemm(a, st);
  // dgrad FLOPs equal the forward's (every (pixel, tap, ci, co) product appears once)
  prof_end(0, st, udaseg_conv_flops(d), 1, d);
  return rc;
}

extern "C" int udaseg_conv2d_dgrad(const udaseg_conv_desc* d, const float* dy, const float* w_t, float* dx,
                                   int accumulate, void* stream) {
  return conv2d_dgrad_impl(d, dy, w_t, dx, accumulate, stream, 0);
}

extern "C" int udaseg_conv2d_dgrad_bf16(const udaseg_conv_desc* d, const void* dy, const void* w_t, void* dx, int accumulate,
                                        void* stream) {
  return conv2d_dgrad_impl(d, dy, w_t, dx, accumulate, stream, 1);
}

extern "C" int udaseg_conv2d_dgrad_bnreduce_ok(const udaseg_conv_desc* d) { return d && dgrad_bnreduce_ok(d) ? 1 : 0; }

// bf16: the reductions ride on the LDS-staged epilogue, which guards rows and channel chunks itself -- any stride-1 geometry
static bool dgrad_bnreduce_ok_bf16(const udaseg_conv_desc* d) {
  return check_desc(d) == UDASEG_OK && d->stride == 1 && d->ci % 8 == 0 && d->co % 8 == 0;
}
extern "C" int udaseg_conv2d_dgrad_bnreduce_bf16_ok(const udaseg_conv_desc* d) { return d && dgrad_bnreduce_ok_bf16(d) ? 1 : 0; }

extern "C" int udaseg_conv2d_dgrad_bnreduce_bf16(const udaseg_conv_desc* d, const void* dy, const void* w_t, void* dx,
                                                 const void* prev_y, const float* save_mean, const float* save_rstd,
                                                 const float* gamma, const float* beta, int act, float slope, double* bsums,
                                                 void* stream) {
  UDASEG_CHECK_ARG(d && prev_y && save_mean && save_rstd && gamma && beta && bsums, "conv2d_dgrad_bnreduce_bf16: NULL pointer");
  UDASEG_CHECK_ARG(dx && (reinterpret_cast<uintptr_t>(dx) & 15) == 0 && (reinterpret_cast<uintptr_t>(prev_y) & 15) == 0,
                   "conv2d_dgrad_bnreduce_bf16: dx and prev_y must be 16-byte aligned");
  if (!dgrad_bnreduce_ok_bf16(d)) {
    set_error("conv2d_dgrad_bnreduce_bf16: stride-1 convolutions with channel counts that are multiples of 8 only");
    return UDASEG_E_UNSUPPORTED;
  }
  const BnReduceArgs b = {static_cast<const float*>(prev_y), save_mean, save_rstd, gamma, beta, act, slope, bsums};
  return conv2d_dgrad_impl(d, dy, w_t, dx, 0, stream, 1, nullptr, 0, &b);
}

extern "C" int udaseg_conv2d_dgrad_bnreduce(const udaseg_conv_desc* d, const float* dy, const float* w_t, float* dx,
                                            const float* prev_y, const float* save_mean, const float* save_rstd,
                                            const float* gamma, const float* beta, int act, float slope, double* bsums,
                                            void* stream) {
  UDASEG_CHECK_ARG(d && prev_y && save_mean && save_rstd && gamma && beta && bsums, "conv2d_dgrad_bnreduce: NULL pointer");
  if (!dgrad_bnreduce_ok(d)) {
    set_error("conv2d_dgrad_bnreduce: this geometry cannot carry the reductions (ask udaseg_conv2d_dgrad_bnreduce_ok first)");
    return UDASEG_E_UNSUPPORTED;
  }
  const BnReduceArgs b = {prev_y, save_mean, save_rstd, gamma, beta, act, slope, bsums};
  return conv2d_dgrad_impl(d, dy, w_t, dx, 0, stream, 0, nullptr, 0, &b);
}

extern "C" int udaseg_conv2d_dgrad_split(const udaseg_conv_desc* d, const float* dy, const float* w_t, float* dx_a, float* dx_b,
                                         int ca, void* stream) {
  return conv2d_dgrad_impl(d, dy, w_t, dx_a, 0, stream, 0, dx_b, ca);
}

extern "C" int udaseg_conv2d_dgrad_split_bf16(const udaseg_conv_desc* d, const void* dy, const void* w_t, void* dx_a, void* dx_b,
                                              int ca, void* stream) {
  UDASEG_CHECK_ARG(ca % 8 == 0, "conv2d_dgrad_split_bf16: ca must be a multiple of 8");
  return conv2d_dgrad_impl(d, dy, w_t, dx_a, 0, stream, 1, dx_b, ca);
}
