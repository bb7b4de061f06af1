// Where does the implicit-GEMM main loop lose time?  Same staging scheme as csrc/conv_igemm.hip (64x64x32 tile, 4 waves,
// K-contiguous LDS rows padded to 36 floats, global -> registers -> LDS, one barrier per K-step, v_mfma_f32_32x32x2_f32),
// on a plain GEMM  C[M][N] = A[M][K] * B[N][K]^T,  with parts of the loop switched off by MODE:
//   0 full loop   1 no global loads (registers re-stored)   2 no LDS stores / barrier either   3 MFMA only
//   4 full loop, loads for tile kt+2 (two register stages)
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/gemm_probe.hip -o tools/micro/gemm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 64, BN = 64, BK = 32, LD = 36;

template <int MODE>
__global__ __launch_bounds__(256) void gemm(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
                                            int N, int K) {
  __shared__ __attribute__((aligned(16))) float As[2][BM * LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int ntn = N / BN;
  const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
  const int kq = tid & 7, lrow = tid >> 3;
  f32x16 acc;
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  f32x4 ra[2][2], rb[2][2];
  const int nkt = K / BK;
  auto load = [&](int kt, int st) {
    for (int p = 0; p < 2; ++p) {
      ra[st][p] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + lrow + 32 * p) * K + kt * BK + kq * 4);
      rb[st][p] = *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + lrow + 32 * p) * K + kt * BK + kq * 4);
    }
  };
  auto store = [&](int buf, int st) {
    for (int p = 0; p < 2; ++p) {
      *reinterpret_cast<f32x4*>(&As[buf][(lrow + 32 * p) * LD + kq * 4]) = ra[st][p];
      *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + 32 * p) * LD + kq * 4]) = rb[st][p];
    }
  };
  auto mfma = [&](int buf) {
    const float* Ac = &As[buf][(wm + lr) * LD + lh * 4];
    const float* Bc = &Bs[buf][(wn + lr) * LD + lh * 4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 af, bf;
      if (MODE == 3) {
        af = ra[0][0];
        bf = rb[0][0];
      } else {
        af = *reinterpret_cast<const f32x4*>(Ac + s * 8);
        bf = *reinterpret_cast<const f32x4*>(Bc + s * 8);
      }
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k2], bf[k2], acc, 0, 0, 0);
    }
  };
  load(0, 0);
  store(0, 0);
  if (MODE == 4 && nkt > 1) load(1, 1);
  __syncthreads();
  if (MODE == 4) {
    int kt = 0;
    for (; kt + 3 < nkt; kt += 2) {
      load(kt + 2, 0);
      mfma(0);
      store(1, 1);
      __syncthreads();
      load(kt + 3, 1);
      mfma(1);
      store(0, 0);
      __syncthreads();
    }
    // (tail tiles skipped: timing probe)
  } else {
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
      if (MODE == 0 && kt + 1 < nkt) load(kt + 1, 0);
      mfma(cur);
      if (MODE <= 1) {
        store(cur ^ 1, 0);
        __syncthreads();
      }
      if (MODE <= 1) cur ^= 1;
    }
  }
  for (int v = 0; v < 16; ++v) {
    const int m = m0 + wm + (v & 3) + 8 * (v >> 2) + 4 * lh;
    C[(size_t)m * N + n0 + wn + lr] = acc[v];
  }
}

template <int MODE>
void run(const float* A, const float* B, float* C, int M, int N, int K, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = (M / BM) * (N / BN);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm<MODE>, dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K);
  hipEventRecord(e0);
  const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(gemm<MODE>, dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= it;
  printf("  mode %d %-44s %8.1f us  %7.1f TFLOP/s\n", MODE, what, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
}

int main(int argc, char** argv) {
  const int shapes[][3] = {{131072, 64, 576}, {131072, 64, 3136}, {32768, 128, 1152}, {8192, 256, 2304}, {8192, 4096, 4096}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4);
    hipMalloc(&B, (size_t)N * K * 4);
    hipMalloc(&C, (size_t)M * N * 4);
    float* h = (float*)malloc((size_t)M * K * 4);
    for (size_t i = 0; i < (size_t)M * K; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(A, h, (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h, (size_t)N * K * 4 < (size_t)M * K * 4 ? (size_t)N * K * 4 : (size_t)M * K * 4, hipMemcpyHostToDevice);
    printf("M=%d N=%d K=%d\n", M, N, K);
    run<0>(A, B, C, M, N, K, "full loop (load kt+1, mfma, store, barrier)");
    run<4>(A, B, C, M, N, K, "full loop, loads two tiles ahead");
    run<1>(A, B, C, M, N, K, "no global loads");
    run<2>(A, B, C, M, N, K, "no loads, no LDS stores, no barrier");
    run<3>(A, B, C, M, N, K, "MFMA only");
    free(h);
    hipFree(A);
    hipFree(B);
    hipFree(C);
  }
  return 0;
}
