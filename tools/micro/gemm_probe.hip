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
  f32x4 ra[3][2], rb[3][2];
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
  if ((MODE == 4 || MODE == 6) && nkt > 1) load(1, 1);
  __syncthreads();
  if (MODE == 6) {      // three register stages: tiles kt+1, kt+2, kt+3 in flight around the MFMA phase of kt
    // (entered with tile 0 in LDS buffer 0 and tile 1 in flight in stage 1, like MODE 4)
    if (nkt > 2) load(2, 2);
    int kt = 0;
    for (; kt + 5 < nkt; kt += 6) {   // buffers alternate 0/1, stages rotate 1 -> 2 -> 0
      load(kt + 3, 0); mfma(0); store(1, 1); __syncthreads();
      load(kt + 4, 1); mfma(1); store(0, 2); __syncthreads();
      load(kt + 5, 2); mfma(0); store(1, 0); __syncthreads();
      load(kt + 6 < nkt ? kt + 6 : kt, 0); mfma(1); store(0, 1); __syncthreads();
      load(kt + 7 < nkt ? kt + 7 : kt, 1); mfma(0); store(1, 2); __syncthreads();
      load(kt + 8 < nkt ? kt + 8 : kt, 2); mfma(1); store(0, 0); __syncthreads();
    }
  } else if (MODE == 4) {
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


// MODE 5: the same GEMM with direct-to-LDS loads (raw_ptr_buffer_load_lds, 16 B per lane): no staging registers, no
// ds_write; LDS rows are unpadded 128-B rows whose eight 16-B chunks are XOR-swizzled with (row >> 1) & 7 so that the
// 16-byte fragment reads of 32 consecutive rows stay conflict-free.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ __launch_bounds__(256) void gemm_glds(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                 int M, int N, int K) {
  __shared__ __attribute__((aligned(1024))) float As[2][BM * 32];
  __shared__ __attribute__((aligned(1024))) float Bs[2][BN * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int ntn = N / BN;
  const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((long long)M * K * 4 < (1LL << 31) ? (long long)M * K * 4 : 0x7fffffff), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)((long long)N * K * 4 < (1LL << 31) ? (long long)N * K * 4 : 0x7fffffff), 0x00020000);
  // this wave fills LDS slots [(wave*2 + p) * 64, +64) of a tile (16-B slots, 8 per row)
  unsigned offa[2], offb[2];
  for (int p = 0; p < 2; ++p) {
    const int S = (wave * 2 + p) * 64 + lane, row = S >> 3, j = S & 7, c = j ^ ((row >> 1) & 7);
    offa[p] = (unsigned)((m0 + row) * K + c * 4) * 4u;
    offb[p] = (unsigned)((n0 + row) * K + c * 4) * 4u;
  }
  f32x16 acc;
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  const int nkt = K / BK;
  auto load = [&](int kt, int buf) {
    const unsigned kb = (unsigned)kt * BK * 4u;
    for (int p = 0; p < 2; ++p) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(&As[buf][(wave * 2 + p) * 256]), 16, (int)(offa[p] + kb), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(&Bs[buf][(wave * 2 + p) * 256]), 16, (int)(offb[p] + kb), 0, 0, 0);
    }
  };
  const int ra_row = wm + lr, rb_row = wn + lr;
  const int sa = (ra_row >> 1) & 7, sb = (rb_row >> 1) & 7;
  auto mfma = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 2 * s + lh;
      const f32x4 af = *reinterpret_cast<const f32x4*>(&As[buf][ra_row * 32 + ((c ^ sa) << 2)]);
      const f32x4 bf = *reinterpret_cast<const f32x4*>(&Bs[buf][rb_row * 32 + ((c ^ sb) << 2)]);
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k2], bf[k2], acc, 0, 0, 0);
    }
  };
  load(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) load(kt + 1, cur ^ 1);
    mfma(cur);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's LDS-DMA pieces have landed
    __syncthreads();
    cur ^= 1;
  }
  for (int v = 0; v < 16; ++v) {
    const int m = m0 + wm + (v & 3) + 8 * (v >> 2) + 4 * lh;
    C[(size_t)m * N + n0 + wn + lr] = acc[v];
  }
}

void run_glds(const float* A, const float* B, float* C, int M, int N, int K) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = (M / BM) * (N / BN);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_glds, dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K);
  hipEventRecord(e0);
  const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(gemm_glds, dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= it;
  printf("  mode 5 %-44s %8.1f us  %7.1f TFLOP/s\n", "direct-to-LDS loads, swizzled rows", ms * 1e3, 2.0 * M * N * K / ms / 1e9);
}

// checks mode 5 against mode 0 on the current C contents

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
    run<6>(A, B, C, M, N, K, "full loop, loads three tiles ahead (timing)");
    {
      float* C2;
      hipMalloc(&C2, (size_t)M * N * 4);
      run_glds(A, B, C2, M, N, K);
      hipLaunchKernelGGL(gemm<0>, dim3((M / BM) * (N / BN)), dim3(256), 0, 0, A, B, C, M, N, K);
      hipDeviceSynchronize();
      float* h1 = (float*)malloc(4096 * 4); float* h2 = (float*)malloc(4096 * 4);
      hipMemcpy(h1, C, 4096 * 4, hipMemcpyDeviceToHost); hipMemcpy(h2, C2, 4096 * 4, hipMemcpyDeviceToHost);
      double md = 0; for (int i = 0; i < 4096; ++i) { double d = h1[i] - h2[i]; if (d < 0) d = -d; if (d > md) md = d; }
      printf("         (mode 5 vs mode 0: max |diff| over the first 4096 outputs = %g)\n", md);
      free(h1); free(h2); hipFree(C2);
    }
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
