// Which lane / register of v_mfma_f32_16x16x32_bf16 holds D[row][col], with A[row i][k = 0] = i + 1 in lane i (K slice 0, element 0)
// and B[k = 0][col j] = 64 (j + 1) in lane j: D[i][j] = 64 (i + 1)(j + 1).  Prints the decoded (row, col) of every (lane, reg).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int ka_lanegrp, int ka_elem) {
  const int lane = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)0.f; b[e] = (__bf16)0.f; }
  if ((lane >> 4) == ka_lanegrp) {
    a[ka_elem] = (__bf16)(float)((lane & 15) + 1);
    b[ka_elem] = (__bf16)(float)(64 * ((lane & 15) + 1));
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 4);
  for (int grp = 0; grp < 4; grp += 3)
    for (int el = 0; el < 8; el += 7) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, grp, el);
      float h[256];
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("operands in lane group %d element %d:\n", grp, el);
      int ok = 1;
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
          const int v = (int)h[l * 4 + r] / 64;   // (i + 1)(j + 1); expected row = 4 (l >> 4) + r, col = l & 15
          const int er = 4 * (l >> 4) + r, ec = l & 15;
          if (v != (er + 1) * (ec + 1)) { ok = 0; if (l < 20) printf("  lane %d reg %d: value %d, expected (row %d + 1)(col %d + 1) = %d\n", l, r, v, er, ec, (er + 1) * (ec + 1)); }
        }
      printf("  documented C/D map (col = lane & 15, row = 4 (lane >> 4) + reg) %s\n", ok ? "CONFIRMED" : "does NOT hold");
    }
  return 0;
}
