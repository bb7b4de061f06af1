// Shared host/device helpers for libudaseg_hip.so (gfx950 only).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/udaseg.h"
#include "options.h"

namespace udaseg {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define UDASEG_CHECK_ARG(cond, ...)                 \
  do {                                              \
    if (!(cond)) {                                  \
      ::udaseg::set_error(__VA_ARGS__);             \
      return UDASEG_E_BADARG;                       \
    }                                               \
  } while (0)

#define UDASEG_LAUNCH_CHECK(what)                          \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return ::udaseg::hip_fail(e__, what); \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// q = n / d for 0 <= n < 2^31, d >= 1, with rcp = 1.0f / d: float estimate + exact fix-up.
__device__ __forceinline__ int fast_div(int n, int d, float rcp) {
  int q = (int)((float)n * rcp);
  int r = n - q * d;
  if (r < 0) { q -= 1; r += d; }
  if (r >= d) { q += 1; }
  return q;
}

__device__ __forceinline__ float act_apply(float x, int act, float slope) {
  return (act == UDASEG_ACT_LEAKY && x < 0.f) ? x * slope : x;
}
// derivative factor from the OUTPUT z (leaky/relu preserve sign; torch's relu'/leaky' at 0 is 0/slope)
__device__ __forceinline__ float act_grad(float z, int act, float slope) {
  return (act == UDASEG_ACT_LEAKY && !(z > 0.f)) ? slope : 1.f;
}

// wave64 sum via DPP-free shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Fold per-thread V-channel partials over the rows of a block, then one atomic per channel per block with CONSECUTIVE LANES ON
// CONSECUTIVE CHANNELS: one atomic instruction then covers 64 adjacent words (4-8 cache lines).  (The first form had the owner
// thread of a channel group issue its V atomics itself -- lanes V words apart, every instruction touching 64 lines: the
// memory-side atomic units see requests per line, and bn_bwd_reduce's time followed its block count, 2.2 ms per r50 step at
// 512 blocks, 4.5 ms at 2048: profiles/r03_bn_reduce_atomics.txt.)  Thread g owns channel group g % cq; bs % cq == 0 when
// cq < bs (stream_shape), so rows r = q0, q0 + cq, ... of the block share group q0.  red: bs * V words of LDS.
template <int V, typename T>
__device__ __forceinline__ void block_fold_atomic(const T (&v)[V], T* __restrict__ dst, int cq, T* __restrict__ red) {
  const int tid = threadIdx.x, bs = blockDim.x;
#pragma unroll
  for (int e = 0; e < V; ++e) red[tid * V + e] = v[e];
  __syncthreads();
  if (cq >= bs) {       // every thread alone on its channels: the block's groups are base .. base + bs - 1 (mod cq)
    const int base = (int)(((int64_t)blockIdx.x * bs) % cq);
    for (int j = tid; j < bs * V; j += bs) {
      int qq = base + j / V;
      if (qq >= cq) qq -= cq;
      atomicAdd(dst + qq * V + (j % V), red[j]);
    }
  } else {
    const int C = cq * V;
    for (int t = tid; t < C; t += bs) {
      T s = 0;
      for (int r = t / V; r < bs; r += cq) s += red[r * V + (t % V)];
      atomicAdd(dst + t, s);
    }
  }
  __syncthreads();
}

// Per-channel f64 accumulators are replicated BN_REPLICAS times ([R][2][C]); block b adds into replica b % R and the
// consumer sums the replicas.  Same-address memory-side atomics serialise (~100 ns each): 2048 blocks on one address
// cost ~130 us per launch (measured, profiles/r01_kernel_stats_first.csv); 512 blocks over 16 replicas = 32 per address.
constexpr int BN_REPLICAS = 16;
constexpr int REDUCE_MAX_BLOCKS = 512;
static inline int reduce_max_blocks() {   // UDASEG_OPT_REDUCE_BLOCKS: measurement override of the cap above
  const int x = opt_get(UDASEG_OPT_REDUCE_BLOCKS);
  return x > 0 ? x : REDUCE_MAX_BLOCKS;
}

// dgrad weight repack w[co][t][ci] -> wt[ci][t][co] of one table row, as 32x32 tile transposes through LDS: 128-byte row
// reads, 128-byte (fp32) / 64-byte (bf16) row writes.  (The element-wise form read 4 bytes per lane at a stride of a whole
// weight row: 0.93 TB/s, 97 us per step for r18's 45 MB; 408 us for r50.)  Blocks of gridDim.x stride over the row's tiles.
template <typename OutT>
__device__ __forceinline__ void pack_dgrad_tiles(const float* __restrict__ w, OutT* __restrict__ wt, int co, int T, int ci) {
  __shared__ float tile[32][33];
  const int ta = (co + 31) >> 5, tb = (ci + 31) >> 5;
  const int ntiles = T * ta * tb;
  const int row = threadIdx.x >> 3, q = threadIdx.x & 7;
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int b = tl % tb, r = tl / tb, a = r % ta, t = r / ta;
    {
      const int o = 32 * a + row, c = 32 * b + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (o < co && c < ci) v = *reinterpret_cast<const f32x4*>(w + ((size_t)o * T + t) * ci + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) tile[row][4 * q + k] = v[k];
    }
    __syncthreads();
    {
      const int c = 32 * b + row, o = 32 * a + 4 * q;
      if (c < ci && o < co) {
        OutT* dst = wt + ((size_t)c * T + t) * co + o;
        if constexpr (sizeof(OutT) == 4) {
          f32x4 v;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = tile[4 * q + k][row];
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
          unsigned short h[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) h[k] = __builtin_bit_cast(unsigned short, (__bf16)tile[4 * q + k][row]);
          uint2 pk;
          pk.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
          pk.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
          *reinterpret_cast<uint2*>(dst) = pk;
        }
      }
    }
    __syncthreads();
  }
}
constexpr int PACK_DGRAD_GRID_X = 256;

void* workspace_ptr(size_t* bytes);   // conv_small.hip: the scratch bound to the current device, or nullptr

// channel_sum (bias gradients): above CHSUM_DIRECT_BLOCKS blocks the per-block partials go into CHSUM_REPLICAS zeroed copies of
// the output in a stream-ordered scratch allocation and a second tiny kernel folds them -- 2048 blocks adding into one
// 64-float vector took 415 us for a 67 MB tensor (r02: 24 % of the bf16 adversarial step), 30x its HBM time.
constexpr int CHSUM_REPLICAS = 16;
constexpr int CHSUM_DIRECT_BLOCKS = 32;
static __global__ void fold_replicas_kernel(const float* __restrict__ rep, int c, float* __restrict__ out, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < CHSUM_REPLICAS; ++r) s += rep[(size_t)r * c + i];
  out[i] = accumulate ? out[i] + s : s;
}

struct StreamShape {
  int bs;      // threads per block (multiple of 64... or of C4 when C4 is not a power of two)
  int grid;
  int c4;
};

// Choose block/grid so grid*bs is a multiple of c4 (see header comment).
// per_thread: vectors a thread streams (4: measured best for every kernel that uses this shape)
static inline StreamShape stream_shape(int64_t n4, int c4, int max_blocks = 2048, int per_thread = 4) {
  StreamShape s;
  s.c4 = c4;
  int unit;  // grid must be a multiple of `unit`
  if (c4 <= 256) {
    s.bs = (256 / c4) * c4;
    unit = 1;
  } else {
    s.bs = 256;
    unit = (c4 + 255) / 256;
    while ((unit * 256) % c4 != 0) ++unit;  // c4 = 512 -> 2
  }
  int64_t want = (n4 + (int64_t)s.bs * per_thread - 1) / ((int64_t)s.bs * per_thread);
  if (want > max_blocks) want = max_blocks;
  if (want < 1) want = 1;
  s.grid = (int)(((want + unit - 1) / unit) * unit);
  // every channel quad needs an owning thread among the first c4 global threads
  while ((int64_t)s.grid * s.bs < c4) s.grid += unit;
  return s;
}


// vectors per thread of the BatchNorm apply kernels (UDASEG_BN_APPLY_PT: tuning aid).  Every block of these kernels opens with a pass
// over the 16 replicated f64 accumulator sets (256 bytes per channel, ~33 MB per launch over all blocks), which suggested fewer, longer
// blocks for the <= 17 MB layers: measured and NOT so -- 4 / 8 / 16 / 32 / 64 vectors per thread leave the small layers at the same
// ~6-7 us device time (tools/bn_bandwidth.py's 14 us floor is the host's launch rate) and the step is best at 4 (977 against 972
// images/s at 16, same box).
static inline int apply_per_thread() {
  const int v = opt_get(UDASEG_OPT_BN_APPLY_PT);
  return (v < 1 || v > 256) ? 4 : v;
}

// small-channel direct 3x3 kernels (conv_small.hip)
bool small_conv_applicable(int k, int stride, int pad, int ci_gather, int co_out);
// up != 0: x is the half-resolution tensor [n][h/2][wd/2][ci], read through nearest x2 up-sampling (pixel (y >> 1, x >> 1))
int launch_small_conv(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int ci, int co,
                      int flip, int accumulate, int act, float slope, double* stats, const float* residual, hipStream_t s,
                      int up = 0);
bool small_wgrad_applicable(int k, int stride, int pad, int ci, int co, int ntiles);
// in_scale != null: x is an unwritten BatchNorm activation, act(fma(x, in_scale[c], in_shift[c])) is applied while staging
int launch_small_wgrad(const float* x, const float* dy, float* dw, int n, int h, int wd, int ci, int co, int accumulate,
                       hipStream_t s, int up = 0, const float* in_scale = nullptr, const float* in_shift = nullptr,
                       int in_act = 0, float in_slope = 0.f);

// live launch timing (bench.py roofline leg): per API call (prof_*) and per kernel launch (kprof_*).  The launchers keep their
// kernel id / "attribute set" flags in function-local std::atomic statics: launches come from two host threads' streams, a racing
// first call registers the same name twice (kprof_id is serialised and returns the same id) and sets the same attribute twice.
constexpr int PROF_NKERNELS = 24;
int kprof_id(const char* rocprof_symbol);   // id of a kernel symbol outside the fixed table (registered on first use)
hipEvent_t kprof_begin(hipStream_t s);
void kprof_end(int kid, hipEvent_t a, hipStream_t s, double flops);
// RAII form for the bandwidth-bound kernels: `work` carries their ALGORITHMIC BYTES (bench.py reports TB/s for symbols that do
// not start with "conv").  Costs one static-int test per call when profiling is off.
struct KTimer {
  std::atomic<int>* kid;
  hipStream_t s;
  hipEvent_t ev;
  double work;
  KTimer(std::atomic<int>* k, const char* rocprof_symbol, hipStream_t st, double w) : kid(k), s(st), work(w) {
    if (*kid < 0) *kid = kprof_id(rocprof_symbol);
    ev = kprof_begin(s);
  }
  ~KTimer() { kprof_end(*kid, ev, s, work); }
};
void prof_begin(int family, hipStream_t s);
void prof_suspend(int on);
void prof_end(int family, hipStream_t s, double flops, int kind, const udaseg_conv_desc* d);

}  // namespace udaseg
