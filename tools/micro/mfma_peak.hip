// Pure-MFMA throughput probe (no memory): fp32 32x32x2 and 16x16x4, bf16 32x32x16, with 1/2/4 independent accumulators per
// wave and 1..4 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  f32x4 acc4[NACC];
  for (int i = 0; i < NACC; ++i) {
    for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    for (int v = 0; v < 4; ++v) acc4[i][v] = 0.f;
  }
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  bf16x8 ab, bb;
  for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)a; bb[j] = (__bf16)b; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        if (KIND == 1) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
        if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) {
    for (int v = 0; v < 16; ++v) s += acc[i][v];
    for (int v = 0; v < 4; ++v) s += acc4[i][v];
  }
  if (s == 123.456f) out[0] = s;
}

// same loop with operands that change every 8 MFMAs (random mantissas): data-dependent power draw
template <int NACC>
__global__ __launch_bounds__(256) void probe_random(float* out, int iters, unsigned seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
  unsigned ra = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u), rb = ra * 747796405u + 2891336453u;
  for (int it = 0; it < iters; ++it) {
    ra = ra * 1664525u + 1013904223u;
    rb = rb * 22695477u + 1u;
    const float a = __uint_as_float(0x3f800000u | (ra >> 9)) - 1.5f;      // uniform in [-0.5, 0.5)
    const float b = __uint_as_float(0x3f800000u | (rb >> 9)) - 1.5f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int v = 0; v < 16; ++v) s += acc[i][v];
  if (s == 123.456f) out[0] = s;
}

template <int NACC>
void run_random(int blocks_per_cu) {
  float* out;
  hipMalloc(&out, 4);
  const int iters = 200000 / NACC;
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_random<NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 12345u + rep);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    printf("f32 32x32x2 RANDOM operands acc/wave %d waves/SIMD %d : %8.3f ms  %8.1f TFLOP/s\n", NACC, blocks_per_cu, ms,
           flops / ms / 1e9);
  }
  hipFree(out);
}

template <int NACC, int KIND>
void run(const char* name, int blocks_per_cu, double flop_per_mfma) {
  float* out;
  hipMalloc(&out, 4);
  const int iters = 20000 / NACC;
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NACC, KIND>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 8 * NACC * flop_per_mfma;
  printf("%-14s acc/wave %d  waves/SIMD %d : %8.3f ms  %8.1f TFLOP/s\n", name, NACC, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int b = 1; b <= 4; b *= 2) {
    run<1, 0>("f32 32x32x2", b, 4096.0);
    run<2, 0>("f32 32x32x2", b, 4096.0);
    run<4, 0>("f32 32x32x2", b, 4096.0);
  }
  for (int b = 1; b <= 4; b *= 2) {
    run<1, 1>("f32 16x16x4", b, 2048.0);
    run<4, 1>("f32 16x16x4", b, 2048.0);
  }
  for (int b = 1; b <= 4; b *= 2) {
    run<1, 2>("bf16 32x32x16", b, 32768.0);
    run<4, 2>("bf16 32x32x16", b, 32768.0);
  }
  run_random<1>(1);
  run_random<4>(2);
  run_random<2>(4);
  // sustained: 10 back-to-back long launches of the best fp32 config
  for (int r = 0; r < 3; ++r) run<4, 0>("f32 sustained", 2, 4096.0);
  return 0;
}
