// Fused Adam over a flat fp32 arena (gfx950).  HBM-bound: 16 B read + 12 B written per parameter.
//
// Replaces torch.optim.Adam(...).step() (reference src/models/train.py:344,461;
// src/models/adversarial_trainer.py:56-59,98,114): lr from the caller, betas (0.9, 0.999), eps 1e-8,
// no weight decay, no amsgrad.  Arithmetic order follows torch's single-tensor Adam:
//   m = m + (g - m)*(1-b1);  v = v*b2 + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.h"

namespace udaseg {

__global__ void adam_flat_kernel(f32x4* __restrict__ p, const f32x4* __restrict__ g, f32x4* __restrict__ m,
                                 f32x4* __restrict__ v, int64_t n4, float* __restrict__ ptail, const float* __restrict__ gtail,
                                 float* __restrict__ mtail, float* __restrict__ vtail, int tail, float step_size, float beta1,
                                 float beta2, float eps, float inv_sqrt_bc2) {
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t g0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = g0; i < n4; i += T) {
    const f32x4 gg = g[i];
    f32x4 mm = m[i], vv = v[i], pp = p[i];
    mm = mm + (gg - mm) * (1.f - beta1);
    vv = vv * beta2 + gg * gg * (1.f - beta2);
    f32x4 upd;
#pragma unroll
    for (int e = 0; e < 4; ++e) upd[e] = mm[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps);
    pp = pp - upd * step_size;
    m[i] = mm;
    v[i] = vv;
    p[i] = pp;
  }
  if (g0 < tail) {
    const float gg = gtail[g0];
    float mm = mtail[g0], vv = vtail[g0];
    mm = mm + (gg - mm) * (1.f - beta1);
    vv = vv * beta2 + gg * gg * (1.f - beta2);
    ptail[g0] -= step_size * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
    mtail[g0] = mm;
    vtail[g0] = vv;
  }
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_adam_flat(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1,
                                float beta2, float eps, float bc1, float bc2, void* stream) {
  UDASEG_CHECK_ARG(p && g && m && v && count > 0, "adam_flat: bad arguments");
  UDASEG_CHECK_ARG(bc1 > 0.f && bc2 > 0.f, "adam_flat: bias corrections must be positive");
  UDASEG_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_flat: pointers must be 16-byte aligned");
  const int64_t n4 = count / 4;
  const int tail = (int)(count - n4 * 4);
  int64_t want = (n4 + 511) / 512;
  if (want > 2048) want = 2048;
  if (want < 1) want = 1;
  hipLaunchKernelGGL(adam_flat_kernel, dim3((int)want), dim3(256), 0, as_stream(stream), (f32x4*)p, (const f32x4*)g, (f32x4*)m,
                     (f32x4*)v, n4, p + n4 * 4, g + n4 * 4, m + n4 * 4, v + n4 * 4, tail, lr / bc1, beta1, beta2, eps,
                     1.0f / sqrtf(bc2));
  UDASEG_LAUNCH_CHECK("adam_flat launch");
  return UDASEG_OK;
}
