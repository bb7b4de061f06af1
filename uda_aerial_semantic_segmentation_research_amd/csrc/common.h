// Shared host/device helpers for libudaseg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/udaseg.h"

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

// small-channel direct 3x3 kernels (conv_small.hip)
bool small_conv_applicable(int k, int stride, int pad, int ci_gather, int co_out);
int launch_small_conv(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int ci, int co,
                      int flip, int accumulate, int act, float slope, double* stats, const float* residual, hipStream_t s);
bool small_wgrad_applicable(int k, int stride, int pad, int ci, int co, int ntiles);
int launch_small_wgrad(const float* x, const float* dy, float* dw, int n, int h, int wd, int ci, int co, int accumulate,
                       hipStream_t s);

// live launch timing (bench.py roofline leg): per API call (prof_*) and per kernel launch (kprof_*)
constexpr int PROF_NKERNELS = 16;
hipEvent_t kprof_begin(hipStream_t s);
void kprof_end(int kid, hipEvent_t a, hipStream_t s, double flops);
void prof_begin(int family, hipStream_t s);
void prof_end(int family, hipStream_t s, double flops, int kind, const udaseg_conv_desc* d);

}  // namespace udaseg
