// Direct 3x3 / stride 1 / pad 1 convolution for SMALL channel counts (<= 32), NHWC fp32, on v_mfma_f32_16x16x4_f32.
//
// Serves the decoder tail and segmentation head of smp.Unet (reference src/models/train.py:341,343): dec.4.conv1
// 32->16, dec.4.conv2 16->16, head 16->23(+1) at full image resolution, their data gradients and weight gradients.
// These 9 launches are 12 % of the step's conv FLOPs; through the generic implicit-GEMM kernel they ran at 23-45
// TFLOP/s (half of every 32-wide MFMA tile was padding at 16 output channels, and a K of 144-288 is 5-9 short K-tiles
// of latency per block).  Here:
//   * one block = one 16x16 pixel tile; its 18x18 input halo is staged ONCE in LDS (no 9x im2col re-gather);
//   * the 16-wide MFMA (16 pixels x 16 channels x 4) has no padding waste at 16 channels;
//   * forward/dgrad keep the whole weight tensor in registers (36-72 VGPRs) -- LDS traffic is A fragments only;
//   * each lane's 16-byte LDS read feeds 4 MFMAs (K order permuted identically for A and B);
//   * blocks are persistent over tiles; wgrad accumulates in registers across all its tiles and writes ONE partial
//     per block (no atomics on the 2-5 K hot addresses), a second tiny kernel folds the partials.
#include "common.h"

namespace udaseg {

constexpr int ST = 16;            // tile edge (pixels)
constexpr int SH = ST + 2;        // halo edge

// one caller-owned scratch buffer PER DEVICE (udaseg_set_workspace binds it to the device that is current at the call)
constexpr int MAX_DEVICES = 16;
static void* g_workspace[MAX_DEVICES] = {};
static size_t g_workspace_bytes[MAX_DEVICES] = {};
static int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return -1;
  return dev;
}

struct SmallArgs {
  const float* x;     // gathered operand [n][h][w][ci]
  const float* w;     // [co][9][ci]
  const float* bias;  // [co] or null
  float* y;           // [n][h][w][co]
  int n, h, wd, ci, co;
  const float* residual;  // added before the activation, same layout as y, or null
  double* stats;      // BN statistics of the output, [R][2][co] f64 accumulators, or null
  int flip;           // 0: tap (r,s) reads (y+r-1, x+s-1) (forward); 1: (y+1-r, x+1-s) (data gradient)
  int up;             // 1: x is [n][h/2][wd/2][ci], read through nearest x2 up-sampling (fused decoder up-sample)
  int accumulate, act;
  float slope;
  int tiles_x, tiles_y, ntiles;
};

// Halo staging in two halves so the global loads of ALL of a thread's pieces are in flight together (a load->write
// loop with a dynamic trip count serialises one global-load latency per iteration) and so the next tile's loads can
// fly during the current tile's MFMA phase.
template <int CIP>
struct HaloRegs {
  static constexpr int Q = CIP / 4;                        // float4 per pixel
  static constexpr int TOTAL = SH * SH * Q;
  static constexpr int ITER = (TOTAL + 255) / 256;
  f32x4 v[ITER];
  unsigned inside = 0;     // bit it: piece it of the tile in the registers lies inside the image (issue() only)

  // up: the tensor behind x is [n][h/2][w/2][ci] and halo pixel (gy, gx) reads its pixel (gy >> 1, gx >> 1)
  __device__ __forceinline__ void issue(const float* __restrict__ x, int ni, int ty0, int tx0, int h, int w, int ci, int up = 0) {
    inside = 0;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = threadIdx.x + 256 * it;
      const int pos = idx / Q, cq = idx - pos * Q;
      const int hy = pos / SH, hx = pos - hy * SH;
      const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (idx < TOTAL && (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w && cq * 4 < ci) {
        const size_t pix = up ? (size_t)(ni * (h >> 1) + (gy >> 1)) * (w >> 1) + (gx >> 1) : (size_t)(ni * h + gy) * w + gx;
        t = *reinterpret_cast<const f32x4*>(x + pix * (size_t)ci + cq * 4);
        inside |= 1u << it;
      }
      v[it] = t;
    }
  }
  // The tile in the registers is the RAW output of a conv + BatchNorm + activation layer whose activation was never written
  // (engine.LazyAct on fp32): apply act(fma(v, scale[c], shift[c])) -- bn_apply's own arithmetic -- to the pieces inside the image
  // (padding is padding of the activation and stays zero).  A thread's channel quad is the same for every piece (256 % Q == 0).
  __device__ __forceinline__ void transform(const float* __restrict__ scale, const float* __restrict__ shift, int ci, int act, float slope) {
    static_assert(256 % Q == 0, "one channel quad per thread");
    const int cq = threadIdx.x % Q;
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
    if (cq * 4 < ci) {
      sc = *reinterpret_cast<const f32x4*>(scale + cq * 4);
      sh = *reinterpret_cast<const f32x4*>(shift + cq * 4);
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const bool in = (inside >> it) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = act_apply(__builtin_fmaf(v[it][e], sc[e], sh[e]), act, slope);
        v[it][e] = in ? t : 0.f;
      }
    }
  }
  // HP = floats per halo pixel in LDS (>= CIP).  The forward kernel pads a pixel to CIP + 4 floats: its 16-byte fragment
  // reads have one lane per PIXEL, and at a stride of 64 or 128 bytes those lanes fall on 4 or 2 bank groups (4- to 8-way
  // conflicts); +16 bytes spreads 16 consecutive pixels over all 64 banks.
  template <int HP = CIP>
  __device__ __forceinline__ void commit(float* __restrict__ halo) const {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = threadIdx.x + 256 * it;
      const int pos = idx / Q, cq = idx - pos * Q;
      if (idx < TOTAL) *reinterpret_cast<f32x4*>(halo + pos * HP + cq * 4) = v[it];
    }
  }

  // Interior tiles (halo entirely inside the image): a piece's byte offset is (tile base) + (a per-thread constant that
  // never changes): one add per piece, range-checked buffer loads.  off[it] holds the constants, 2^30 for the pieces
  // a thread does not own (idx >= TOTAL or padded channel groups) -- those read zeros.  The per-element divisions, bounds
  // tests and 64-bit addresses of issue() are VALU work that competes with the fp32 MFMAs for the same ALUs.
  __device__ __forceinline__ void offsets(unsigned (&off)[ITER], int w, int ci, int up = 0) const {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = threadIdx.x + 256 * it;
      const int pos = idx / Q, cq = idx - pos * Q;
      const int hy = pos / SH, hx = pos - hy * SH;
      // up: tile origins are even, so (ty0 + hy - 1) >> 1 == ty0 / 2 + ((hy - 1) >> 1) with an arithmetic shift
      const int rel = up ? ((hy - 1) >> 1) * (w >> 1) + ((hx - 1) >> 1) : (hy - 1) * w + (hx - 1);
      off[it] = (idx < TOTAL && cq * 4 < ci) ? (unsigned)(rel * ci + cq * 4) * 4u : 0x40000000u;
    }
  }
  __device__ __forceinline__ void issue_interior(__amdgpu_buffer_rsrc_t rsrc, const unsigned (&off)[ITER], unsigned tile_base) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      // real constants are small (|off| < 2^24, the top-left halo row / column being negative); the marker 2^30 plus any
      // tile base is past the end of a tensor of at most 2^30 bytes
      v[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off[it] + tile_base), 0, 0));
    }
  }
};

// 16x16 tile of dy (no halo)
template <int COP>
struct TileRegs {
  static constexpr int Q = COP / 4;
  static constexpr int TOTAL = ST * ST * Q;
  static constexpr int ITER = (TOTAL + 255) / 256;
  f32x4 v[ITER];

  __device__ __forceinline__ void issue(const float* __restrict__ dy, int ni, int ty0, int tx0, int h, int w, int co) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = threadIdx.x + 256 * it;
      const int pos = idx / Q, cq = idx - pos * Q;
      const int py = pos / ST, px = pos - py * ST;
      const int gy = ty0 + py, gx = tx0 + px;
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (idx < TOTAL && gy < h && gx < w && cq * 4 < co)
        t = *reinterpret_cast<const f32x4*>(dy + ((size_t)(ni * h + gy) * w + gx) * (size_t)co + cq * 4);
      v[it] = t;
    }
  }
  __device__ __forceinline__ void commit(float* __restrict__ tile) const {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = threadIdx.x + 256 * it;
      if (idx < TOTAL) *reinterpret_cast<f32x4*>(tile + idx * 4) = v[it];
    }
  }
};

struct TileCoord {
  int ni, ty0, tx0;
};
__device__ __forceinline__ TileCoord tile_coord(int tile, int tiles_x, int tiles_y) {
  const int tx = tile % tiles_x, r1 = tile / tiles_x;
  return TileCoord{r1 / tiles_y, (r1 % tiles_y) * ST, tx * ST};
}

// G = 16-channel K groups of the gathered operand, CO_T = 16-wide output-channel tiles.
template <int G, int CO_T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(G * CO_T == 1 ? 3 : 2))) void conv3x3_small_kernel(const SmallArgs a) {
  constexpr int CIP = 16 * G;
  constexpr int HP = CIP + 4;   // padded pixel stride, see HaloRegs::commit
  __shared__ __attribute__((aligned(16))) float halo[SH * SH * HP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;

  // whole weight tensor in registers: B[k = 16g + 4kq + e][j = co] for tap t
  f32x4 wreg[9][G][CO_T];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int ct = 0; ct < CO_T; ++ct) {
        const int co = 16 * ct + li, c0 = 16 * g + 4 * kq;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (co < a.co && c0 < a.ci) v = *reinterpret_cast<const f32x4*>(a.w + ((size_t)co * 9 + t) * a.ci + c0);
        wreg[t][g][ct] = v;
      }
  float bias[CO_T];
#pragma unroll
  for (int ct = 0; ct < CO_T; ++ct) bias[ct] = (a.bias && 16 * ct + li < a.co) ? a.bias[16 * ct + li] : 0.f;

  float ssum[CO_T], ssq[CO_T];   // BN statistics of this lane's channel, running over all of the block's tiles
#pragma unroll
  for (int ct = 0; ct < CO_T; ++ct) ssum[ct] = ssq[ct] = 0.f;

  HaloRegs<CIP> stage;
  const bool small_tensors = (long long)a.n * a.h * a.wd * (a.ci > a.co ? a.ci : a.co) * 4 <= (1LL << 30);
  __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, small_tensors ? (a.up ? a.n * (a.h >> 1) * (a.wd >> 1) : a.n * a.h * a.wd) * a.ci * 4 : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, small_tensors ? a.n * a.h * a.wd * a.co * 4 : 0,
                                                                    0x00020000);
  unsigned hoff[HaloRegs<CIP>::ITER];
  stage.offsets(hoff, a.wd, a.ci, a.up);
  auto issue_tile = [&](int t) {
    const TileCoord c = tile_coord(t, a.tiles_x, a.tiles_y);
    // interior: the 18x18 halo lies inside the image (uniform per tile)
    if (small_tensors && c.ty0 >= 1 && c.tx0 >= 1 && c.ty0 + ST + 1 <= a.h && c.tx0 + ST + 1 <= a.wd) {
      const int base = a.up ? (c.ni * (a.h >> 1) + (c.ty0 >> 1)) * (a.wd >> 1) + (c.tx0 >> 1) : (c.ni * a.h + c.ty0) * a.wd + c.tx0;
      stage.issue_interior(rsrc_x, hoff, (unsigned)(base * a.ci) * 4u);
    } else {
      stage.issue(a.x, c.ni, c.ty0, c.tx0, a.h, a.wd, a.ci, a.up);
    }
  };
  int tile = blockIdx.x;
  if (tile < a.ntiles) issue_tile(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    const TileCoord tc = tile_coord(tile, a.tiles_x, a.tiles_y);
    const int ni = tc.ni, ty0 = tc.ty0, tx0 = tc.tx0;
    __syncthreads();  // previous tile's readers are done with the halo
    stage.template commit<HP>(halo);
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) issue_tile(tile + gridDim.x);  // next tile's halo flies during this tile's MFMAs

    f32x4 acc[4][CO_T];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ct = 0; ct < CO_T; ++ct) acc[r][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int tr = t / 3, ts = t % 3;
      const int dy = a.flip ? 1 - tr : tr - 1, dx = a.flip ? 1 - ts : ts - 1;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        f32x4 af[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int hy = 4 * wave + r + 1 + dy, hx = li + 1 + dx;
          af[r] = *reinterpret_cast<const f32x4*>(halo + (hy * SH + hx) * HP + 16 * g + 4 * kq);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ct = 0; ct < CO_T; ++ct)
              acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][e], wreg[t][g][ct][e], acc[r][ct], 0, 0, 0);
      }
    }

    // D reg v of lane (li, kq): pixel x = 4*kq + v of the row, channel 16*ct + li
    // Fast path (tile inside the image, all 16*CO_T channels real, plain store): one 32-bit offset per lane and tile, the
    // 16 (row, pixel) displacements as scalar offsets of buffer stores.
    if (small_tensors && ty0 + ST <= a.h && tx0 + ST <= a.wd && a.co == 16 * CO_T && !a.accumulate && a.residual == nullptr) {
      const int lane_off = (((ni * a.h + ty0 + 4 * wave) * a.wd + tx0 + 4 * kq) * a.co + li) * 4;
      const int row_b = a.wd * a.co * 4, pix_b = a.co * 4;
      const bool plain = a.bias == nullptr && a.act == UDASEG_ACT_NONE;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int ct = 0; ct < CO_T; ++ct) {
            float val = acc[r][ct][v];
            if (!plain) val += bias[ct];
            ssum[ct] += val;
            ssq[ct] = __builtin_fmaf(val, val, ssq[ct]);
            if (!plain) val = act_apply(val, a.act, a.slope);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rsrc_y, lane_off + ct * 64,
                                                  r * row_b + v * pix_b, 0);
          }
      continue;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gy = ty0 + 4 * wave + r;
      if (gy >= a.h) continue;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int gx = tx0 + 4 * kq + v;
        if (gx >= a.wd) continue;
        float* dst = a.y + ((size_t)(ni * a.h + gy) * a.wd + gx) * (size_t)a.co;
#pragma unroll
        for (int ct = 0; ct < CO_T; ++ct) {
          const int co = 16 * ct + li;
          if (co < a.co) {
            float val = acc[r][ct][v] + bias[ct];
            ssum[ct] += val;
            ssq[ct] += val * val;
            if (a.residual) val += a.residual[((size_t)(ni * a.h + gy) * a.wd + gx) * (size_t)a.co + co];
            val = act_apply(val, a.act, a.slope);
            if (a.accumulate) val += dst[co];
            dst[co] = val;
          }
        }
      }
    }
  }
  if (a.stats) {
    // lanes li, li+16, li+32, li+48 hold the same channel: fold them, then the 4 waves through LDS, then one f64 atomic
    // per (channel, statistic) per block
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < CO_T; ++ct) {
      float s1 = ssum[ct], s2 = ssq[ct];
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (kq == 0) {
        halo[(wave * CO_T + ct) * 16 + li] = s1;
        halo[4 * CO_T * 16 + (wave * CO_T + ct) * 16 + li] = s2;
      }
    }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < 16 * CO_T && tid < a.co) {
      const int ct = tid / 16, l = tid % 16;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s1 += halo[(w * CO_T + ct) * 16 + l];
        s2 += halo[4 * CO_T * 16 + (w * CO_T + ct) * 16 + l];
      }
      double* rep = a.stats + (size_t)(blockIdx.x % 16) * 2 * a.co;
      atomicAdd(rep + tid, (double)s1);
      atomicAdd(rep + a.co + tid, (double)s2);
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct SmallWgradArgs {
  const float* x;    // [n][h][w][ci]
  const float* dy;   // [n][h][w][co]
  float* partial;    // [gridDim.x][COP*9*CIP]
  int n, h, w, ci, co;
  int up;            // 1: x is [n][h/2][w/2][ci] behind a nearest x2 up-sampling
  int tiles_x, tiles_y, ntiles;
  const float* in_scale;   // x is an unwritten BatchNorm activation (HaloRegs::transform); null: x is used as it is
  const float* in_shift;
  int in_act;
  float in_slope;
};

template <int G, int CO_T>
__global__ __launch_bounds__(256) void conv3x3_small_wgrad_kernel(const SmallWgradArgs a) {
  constexpr int CIP = 16 * G, COP = 16 * CO_T;
  constexpr int OUT = COP * 9 * CIP;
  constexpr int HALO = SH * SH * CIP;
  constexpr int LDSF = (HALO + ST * ST * COP) > OUT ? (HALO + ST * ST * COP) : OUT;
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  float* halo = lds;
  float* dyt = lds + HALO;  // [16*16][COP]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;

  f32x4 acc[9][G][CO_T];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int ct = 0; ct < CO_T; ++ct) acc[t][g][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  HaloRegs<CIP> hx;
  TileRegs<COP> hd;
  int tile = blockIdx.x;
  if (tile < a.ntiles) {
    const TileCoord c = tile_coord(tile, a.tiles_x, a.tiles_y);
    hx.issue(a.x, c.ni, c.ty0, c.tx0, a.h, a.w, a.ci, a.up);
    hd.issue(a.dy, c.ni, c.ty0, c.tx0, a.h, a.w, a.co);
  }
  for (; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();  // previous tile's readers are done
    if (a.in_scale != nullptr) hx.transform(a.in_scale, a.in_shift, a.ci, a.in_act, a.in_slope);      // uniform
    hx.commit(halo);
    hd.commit(dyt);
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) {  // next tile's operands fly during this tile's MFMA phase
      const TileCoord c = tile_coord(tile + gridDim.x, a.tiles_x, a.tiles_y);
      hx.issue(a.x, c.ni, c.ty0, c.tx0, a.h, a.w, a.ci, a.up);
      hd.issue(a.dy, c.ni, c.ty0, c.tx0, a.h, a.w, a.co);
    }

    // GEMM: dW[co][t][ci] += sum_pix dy[pix][co] * x[pix + tap][ci];  M = co (lane li), N = ci (lane li), K = 4 pixels (kq)
#pragma unroll 1   // keep the live range to one row: full unrolling hoists 144+ LDS reads and overflows 256 VGPRs
    for (int r = 0; r < 4; ++r) {
      const int py = 4 * wave + r;
#pragma unroll 2
      for (int q = 0; q < 4; ++q) {
        const int px = 4 * q + kq;
        float af[CO_T];
#pragma unroll
        for (int ct = 0; ct < CO_T; ++ct) af[ct] = dyt[(py * ST + px) * COP + 16 * ct + li];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int hy = py + t / 3, hx = px + t % 3;  // halo coords of (py + r - 1, px + s - 1)
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const float bf = halo[(hy * SH + hx) * CIP + 16 * g + li];
#pragma unroll
            for (int ct = 0; ct < CO_T; ++ct)
              acc[t][g][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ct], bf, acc[t][g][ct], 0, 0, 0);
          }
        }
      }
    }
  }

  // fold the 4 waves through LDS, then one partial per block.  D reg v of lane (li, kq): co = 16ct + 4kq + v, ci = 16g + li
  __syncthreads();
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int ct = 0; ct < CO_T; ++ct)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int idx = ((16 * ct + 4 * kq + v) * 9 + t) * CIP + 16 * g + li;
              lds[idx] = (wv == 0 ? 0.f : lds[idx]) + acc[t][g][ct][v];
            }
    }
    __syncthreads();
  }
  float* dst = a.partial + (size_t)blockIdx.x * OUT;
  for (int idx = threadIdx.x; idx < OUT; idx += 256) dst[idx] = lds[idx];
}

// dw[co][9][ci] += sum over blocks of partial[b][COP][9][CIP].  gridDim.y slices of the block range, 8 loads in flight
// per thread, one atomic per (element, slice): 16 adds per address.
constexpr int FOLD_SLICES = 16;
__global__ void small_wgrad_fold_kernel(const float* __restrict__ partial, int nblocks, int cop, int cip, int co, int ci,
                                        float* __restrict__ dw) {
  const int out = cop * 9 * cip;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= out) return;
  const int c = idx % cip, t = (idx / cip) % 9, o = idx / (cip * 9);
  if (c >= ci || o >= co) return;
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = b0;
  for (; b + 8 <= b1; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += partial[(size_t)(b + u) * out + idx];
  }
  for (; b < b1; ++b) s[0] += partial[(size_t)b * out + idx];
  const float tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  if (b1 > b0) atomicAdd(dw + ((size_t)o * 9 + t) * ci + c, tot);
}

static int small_grid(int ntiles, int blocks_per_cu) {
  int g = 256 * blocks_per_cu;
  return ntiles < g ? ntiles : g;
}

// ---- dispatch helpers used by conv_igemm.hip / conv_wgrad.hip ----------------------------------------------------
bool small_conv_applicable(int k, int stride, int pad, int ci_gather, int co_out) {
  if (k != 3 || stride != 1 || pad != 1) return false;
  const int G = (ci_gather + 15) / 16, T = (co_out + 15) / 16;
  return G * T <= 2;
}

int launch_small_conv(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int ci, int co,
                      int flip, int accumulate, int act, float slope, double* stats, const float* residual, hipStream_t s,
                      int up) {
  SmallArgs a = {};
  a.up = up;
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = stats; a.residual = residual;
  a.n = n; a.h = h; a.wd = wd; a.ci = ci; a.co = co;
  a.flip = flip; a.accumulate = accumulate; a.act = act; a.slope = slope;
  a.tiles_x = cdiv(wd, ST); a.tiles_y = cdiv(h, ST); a.ntiles = n * a.tiles_x * a.tiles_y;
  const int G = cdiv(ci, 16), T = cdiv(co, 16);
  dim3 block(256);
  const double fl = 2.0 * (double)n * h * wd * co * 9.0 * ci;
  hipEvent_t ev = kprof_begin(s);
  if (G == 1 && T == 1) {
    hipLaunchKernelGGL((conv3x3_small_kernel<1, 1>), dim3(small_grid(a.ntiles, 6)), block, 0, s, a);
    kprof_end(4, ev, s, fl);
  } else if (G == 2 && T == 1) {
    hipLaunchKernelGGL((conv3x3_small_kernel<2, 1>), dim3(small_grid(a.ntiles, 3)), block, 0, s, a);
    kprof_end(5, ev, s, fl);
  } else if (G == 1 && T == 2) {
    hipLaunchKernelGGL((conv3x3_small_kernel<1, 2>), dim3(small_grid(a.ntiles, 6)), block, 0, s, a);
    kprof_end(6, ev, s, fl);
  } else {
    set_error("small conv: unsupported channel groups G=%d T=%d", G, T);
    return UDASEG_E_UNSUPPORTED;
  }
  UDASEG_LAUNCH_CHECK("conv3x3_small launch");
  return UDASEG_OK;
}

bool small_wgrad_applicable(int k, int stride, int pad, int ci, int co, int ntiles) {
  if (!small_conv_applicable(k, stride, pad, ci, co)) return false;
  const int G = (ci + 15) / 16, T = (co + 15) / 16;
  const size_t need = (size_t)small_grid(ntiles, 2) * (16 * T) * 9 * (16 * G) * sizeof(float);
  const int dev = current_device();
  return dev >= 0 && g_workspace[dev] != nullptr && g_workspace_bytes[dev] >= need;
}

int launch_small_wgrad(const float* x, const float* dy, float* dw, int n, int h, int wd, int ci, int co, int accumulate,
                       hipStream_t s, int up, const float* in_scale, const float* in_shift, int in_act, float in_slope) {
  SmallWgradArgs a = {};
  a.up = up;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act; a.in_slope = in_slope;
  const int dev = current_device();
  if (dev < 0 || g_workspace[dev] == nullptr) {
    set_error("small wgrad: no workspace bound to the current device");
    return UDASEG_E_WORKSPACE;
  }
  a.x = x; a.dy = dy; a.partial = static_cast<float*>(g_workspace[dev]);
  a.n = n; a.h = h; a.w = wd; a.ci = ci; a.co = co;
  a.tiles_x = cdiv(wd, ST); a.tiles_y = cdiv(h, ST); a.ntiles = n * a.tiles_x * a.tiles_y;
  const int G = cdiv(ci, 16), T = cdiv(co, 16);
  const int grid = small_grid(a.ntiles, 2);
  dim3 block(256);
  const double fl = 2.0 * (double)n * h * wd * co * 9.0 * ci;
  hipEvent_t ev = kprof_begin(s);
  if (G == 1 && T == 1) {
    hipLaunchKernelGGL((conv3x3_small_wgrad_kernel<1, 1>), dim3(grid), block, 0, s, a);
    kprof_end(9, ev, s, fl);
  } else if (G == 2 && T == 1) {
    hipLaunchKernelGGL((conv3x3_small_wgrad_kernel<2, 1>), dim3(grid), block, 0, s, a);
    kprof_end(10, ev, s, fl);
  } else if (G == 1 && T == 2) {
    hipLaunchKernelGGL((conv3x3_small_wgrad_kernel<1, 2>), dim3(grid), block, 0, s, a);
    kprof_end(11, ev, s, fl);
  } else {
    set_error("small wgrad: unsupported channel groups G=%d T=%d", G, T);
    return UDASEG_E_UNSUPPORTED;
  }
  UDASEG_LAUNCH_CHECK("conv3x3_small_wgrad launch");
  const int out = 16 * T * 9 * 16 * G;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(dw, 0, (size_t)co * 9 * ci * sizeof(float), s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(small wgrad)");
  }
  hipLaunchKernelGGL(small_wgrad_fold_kernel, dim3(cdiv(out, 256), FOLD_SLICES), dim3(256), 0, s, a.partial, grid, 16 * T, 16 * G,
                     co, ci, dw);
  UDASEG_LAUNCH_CHECK("small_wgrad_fold launch");
  return UDASEG_OK;
}

}  // namespace udaseg

namespace udaseg {
// the device's bound scratch (udaseg_set_workspace), for the other translation units
void* workspace_ptr(size_t* bytes) {
  const int dev = current_device();
  if (dev < 0 || g_workspace[dev] == nullptr) {
    if (bytes) *bytes = 0;
    return nullptr;
  }
  if (bytes) *bytes = g_workspace_bytes[dev];
  return g_workspace[dev];
}
}  // namespace udaseg

extern "C" size_t udaseg_workspace_bytes(const udaseg_conv_desc* d) {
  // what udaseg_conv2d_wgrad wants to find in the workspace for this convolution (0: it needs none)
  if (!d || d->kh != d->kw || !udaseg::small_conv_applicable(d->kh, d->stride, d->pad, d->ci, d->co)) return 0;
  const int G = (d->ci + 15) / 16, T = (d->co + 15) / 16;
  const int ntiles = d->n * udaseg::cdiv(d->hi, udaseg::ST) * udaseg::cdiv(d->wi, udaseg::ST);
  return (size_t)udaseg::small_grid(ntiles, 2) * (16 * T) * 9 * (16 * G) * sizeof(float);
}

extern "C" int udaseg_set_workspace(void* ptr, size_t bytes) {
  const int dev = udaseg::current_device();
  UDASEG_CHECK_ARG(dev >= 0, "set_workspace: no current HIP device (or more than %d devices)", udaseg::MAX_DEVICES);
  udaseg::g_workspace[dev] = ptr;
  udaseg::g_workspace_bytes[dev] = ptr ? bytes : 0;
  return UDASEG_OK;
}
