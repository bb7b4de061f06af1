// Segmentation loss family of the reference beyond plain cross entropy, fp32, HBM-bound:
//   DiceLoss                     reference src/models/losses.py:110-152
//   WeightedSegmentationLoss     :154-215  (focal-weighted cross entropy + Dice)
//   ConsistencyLoss              :53-108   (symmetric temperature KL, 'batchmean')
// Logits are NHWC rows [pixel][ldc] with `classes` valid channels (ldc % 4 == 0, ldc <= 32), as produced by Unet.
// One thread per pixel, softmax row in registers, wave/block reductions, f64 cross-block accumulation.
#include "common.h"

namespace udaseg {

constexpr int SL_BLOCKS = 1024;

// lp[c] = log softmax(z * inv_t)[c] for c < classes (0 in the pad lanes)
template <int LDC4>
__device__ __forceinline__ void load_log_softmax(const f32x4* __restrict__ logits, int64_t p, int classes, float inv_t, float* lp) {
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < LDC4; ++k) {
    const f32x4 v = logits[p * LDC4 + k];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lp[k * 4 + e] = v[e] * inv_t;
      if (k * 4 + e < classes) mx = fmaxf(mx, lp[k * 4 + e]);
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < LDC4 * 4; ++c)
    if (c < classes) sum += expf(lp[c] - mx);
  const float lse = mx + logf(sum);
#pragma unroll
  for (int c = 0; c < LDC4 * 4; ++c) lp[c] = c < classes ? lp[c] - lse : 0.f;
}

template <int LDC4>
__device__ __forceinline__ void load_softmax(const f32x4* __restrict__ logits, int64_t p, int classes, float inv_t, float* pr) {
  load_log_softmax<LDC4>(logits, p, classes, inv_t, pr);
#pragma unroll
  for (int c = 0; c < LDC4 * 4; ++c) pr[c] = c < classes ? expf(pr[c]) : 0.f;
}

// ---------------------------------------------------------------------------------------------------------- Dice
// sums[b][0][c] = sum_pix p_c * onehot_c, [b][1][c] = sum_pix p_c, [b][2][c] = sum_pix onehot_c   (f64, caller-zeroed)
template <int LDC4>
__global__ __launch_bounds__(256) void dice_stats_kernel(const f32x4* __restrict__ logits, const int64_t* __restrict__ target,
                                                         int64_t pix_per_image, int classes, double* __restrict__ sums) {
  constexpr int LDC = LDC4 * 4;
  __shared__ float sh[3][32];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 96; i += 256) (&sh[0][0])[i] = 0.f;
  __syncthreads();
  float psum[LDC];
#pragma unroll
  for (int c = 0; c < LDC; ++c) psum[c] = 0.f;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < pix_per_image; q += (int64_t)gridDim.x * 256) {
    const int64_t p = (int64_t)b * pix_per_image + q;
    float pr[LDC];
    load_softmax<LDC4>(logits, p, classes, 1.f, pr);
    const int t = (int)target[p];
    float pt = 0.f;
#pragma unroll
    for (int c = 0; c < LDC; ++c) {
      psum[c] += pr[c];
      if (c == t) pt = pr[c];
    }
    if ((unsigned)t < (unsigned)classes) {
      atomicAdd(&sh[0][t], pt);
      atomicAdd(&sh[2][t], 1.f);
    }
  }
#pragma unroll
  for (int c = 0; c < LDC; ++c) {
    const float s = wave_sum(psum[c]);
    if ((threadIdx.x & 63) == 0 && c < classes) atomicAdd(&sh[1][c], s);
  }
  __syncthreads();
  if (threadIdx.x < 96) {
    const int k = threadIdx.x / 32, c = threadIdx.x % 32;
    if (c < classes) atomicAdd(sums + ((size_t)b * 3 + k) * classes + c, (double)sh[k][c]);
  }
}

// per-image mode (reference DiceLoss): loss = 1 - mean_{b,c} (2I + s)/(U + s).
// pooled mode (segmentation_models_pytorch DiceLoss(mode='multiclass'), the reference's UDALoss at src/models/uda.py:84):
//   sums pooled over the batch; score_c = (2I_c + s)/max(U_c + s, eps); loss = mean_c (1 - score_c) * [class c present].
// coef[b][0][c] = a, coef[b][1][c] = b with dLoss/dp_c(pixel of image b) = a * onehot_c + b.
__global__ void dice_finish_kernel(const double* __restrict__ sums, int batch, int classes, float smooth, float eps, int pooled,
                                   float* __restrict__ loss, float* __restrict__ coef) {
  double acc = 0.0;
  if (!pooled) {
    const int n = batch * classes;
    for (int i = threadIdx.x; i < n; i += 64) {
      const int b = i / classes, c = i % classes;
      const double I = sums[((size_t)b * 3 + 0) * classes + c];
      const double U = sums[((size_t)b * 3 + 1) * classes + c] + sums[((size_t)b * 3 + 2) * classes + c];
      const double den = U + (double)smooth;
      acc += (2.0 * I + (double)smooth) / den;
      coef[((size_t)b * 2 + 0) * classes + c] = (float)(-(2.0 / den) / n);
      coef[((size_t)b * 2 + 1) * classes + c] = (float)(((2.0 * I + (double)smooth) / (den * den)) / n);
    }
    acc = wave_sum_d(acc);
    if (threadIdx.x == 0) *loss = (float)(1.0 - acc / n);
    return;
  }
  for (int c = threadIdx.x; c < classes; c += 64) {
    double I = 0.0, U = 0.0, T = 0.0;
    for (int b = 0; b < batch; ++b) {
      I += sums[((size_t)b * 3 + 0) * classes + c];
      U += sums[((size_t)b * 3 + 1) * classes + c] + sums[((size_t)b * 3 + 2) * classes + c];
      T += sums[((size_t)b * 3 + 2) * classes + c];
    }
    const double den = U + (double)smooth;
    const bool present = T > 0.0, clamped = den < (double)eps;
    const double score = (2.0 * I + (double)smooth) / (clamped ? (double)eps : den);
    if (present) acc += 1.0 - score;
    // d(1 - score)/dp = -(2 t)/den + (2I + s)/den^2 (the second term vanishes where the denominator is clamped)
    const float a = present ? (float)(-(2.0 / (clamped ? (double)eps : den)) / classes) : 0.f;
    const float bq = present && !clamped ? (float)(((2.0 * I + (double)smooth) / (den * den)) / classes) : 0.f;
    for (int b = 0; b < batch; ++b) {
      coef[((size_t)b * 2 + 0) * classes + c] = a;
      coef[((size_t)b * 2 + 1) * classes + c] = bq;
    }
  }
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) *loss = (float)(acc / classes);
}

// dlogits (+)= scale * p_k * (g_k - sum_c p_c g_c),  g_c = coef_a[b][c] * onehot_c + coef_b[b][c]
template <int LDC4>
__global__ __launch_bounds__(256) void dice_bwd_kernel(const f32x4* __restrict__ logits, const int64_t* __restrict__ target,
                                                       const float* __restrict__ coef, const float* __restrict__ grad_out,
                                                       float weight, int64_t pix_per_image, int batch, int classes,
                                                       f32x4* __restrict__ dlogits, int accumulate) {
  constexpr int LDC = LDC4 * 4;
  const float scale = (grad_out ? *grad_out : 1.f) * weight;
  const int64_t pixels = pix_per_image * batch;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 256) {
    const int b = (int)(p / pix_per_image);
    float pr[LDC], g[LDC];
    load_softmax<LDC4>(logits, p, classes, 1.f, pr);
    const int t = (int)target[p];
    float S = 0.f;
#pragma unroll
    for (int c = 0; c < LDC; ++c) {
      g[c] = 0.f;
      if (c < classes) {
        g[c] = coef[((size_t)b * 2 + 1) * classes + c] + (c == t ? coef[((size_t)b * 2 + 0) * classes + c] : 0.f);
        S += pr[c] * g[c];
      }
    }
#pragma unroll
    for (int k = 0; k < LDC4; ++k) {
      f32x4 d;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = scale * pr[k * 4 + e] * (g[k * 4 + e] - S);
      if (accumulate) d += dlogits[p * LDC4 + k];
      dlogits[p * LDC4 + k] = d;
    }
  }
}

// ------------------------------------------------------------------------------------------- focal-weighted cross entropy
// per pixel: ce = w[t] * (lse - z_t); pt = exp(-ce); f = alpha * (1 - pt)^gamma * ce;  partials[block] = sum f
template <int LDC4>
__global__ __launch_bounds__(256) void focal_fwd_kernel(const f32x4* __restrict__ logits, const int64_t* __restrict__ target,
                                                        const float* __restrict__ class_w, float alpha, float gamma,
                                                        int64_t pixels, int classes, double* __restrict__ partials) {
  constexpr int LDC = LDC4 * 4;
  __shared__ double red[4];
  double local = 0.0;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 256) {
    float lp[LDC];
    load_log_softmax<LDC4>(logits, p, classes, 1.f, lp);
    const int t = (int)target[p];
    float lpt = 0.f;
#pragma unroll
    for (int c = 0; c < LDC; ++c)
      if (c == t) lpt = lp[c];
    const float ce = -(class_w ? class_w[t] : 1.f) * lpt;
    const float pt = expf(-ce);
    local += (double)(alpha * powf(fmaxf(1.f - pt, 0.f), gamma) * ce);
  }
  local = wave_sum_d(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void sum_finish_kernel(const double* __restrict__ partials, int n, double divisor, float* __restrict__ out, int accumulate) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partials[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) {
    const float v = (float)(s / divisor);
    *out = accumulate ? *out + v : v;
  }
}

// df/dce = alpha * [(1-pt)^gamma + ce * gamma * (1-pt)^(gamma-1) * pt];  dce/dz_k = w_t * (p_k - [k == t])
template <int LDC4>
__global__ __launch_bounds__(256) void focal_bwd_kernel(const f32x4* __restrict__ logits, const int64_t* __restrict__ target,
                                                        const float* __restrict__ class_w, float alpha, float gamma,
                                                        const float* __restrict__ grad_out, float weight, int64_t pixels,
                                                        int classes, f32x4* __restrict__ dlogits, int accumulate) {
  constexpr int LDC = LDC4 * 4;
  const float scale = (grad_out ? *grad_out : 1.f) * weight;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 256) {
    float pr[LDC];
    load_log_softmax<LDC4>(logits, p, classes, 1.f, pr);
    const int t = (int)target[p];
    float lpt = 0.f;
#pragma unroll
    for (int c = 0; c < LDC; ++c) {
      if (c == t) lpt = pr[c];
      pr[c] = c < classes ? expf(pr[c]) : 0.f;
    }
    const float w = class_w ? class_w[t] : 1.f;
    const float ce = -w * lpt;
    const float pt = expf(-ce);
    const float om = fmaxf(1.f - pt, 0.f);
    const float dfdce = om > 0.f ? alpha * (powf(om, gamma) + ce * gamma * powf(om, gamma - 1.f) * pt) : (gamma == 0.f ? alpha : 0.f);
    const float k0 = scale * dfdce * w;
#pragma unroll
    for (int k = 0; k < LDC4; ++k) {
      f32x4 d;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = k * 4 + e;
        d[e] = c < classes ? k0 * (pr[c] - (c == t ? 1.f : 0.f)) : 0.f;
      }
      if (accumulate) d += dlogits[p * LDC4 + k];
      dlogits[p * LDC4 + k] = d;
    }
  }
}

// ----------------------------------------------------------------------------------------------- consistency (sym. KL)
// per pixel: kl12 = sum p1 (l1 - l2), kl21 = sum p2 (l2 - l1) with p = softmax(z / T), l = log p; partial = sum (kl12 + kl21)
template <int LDC4>
__global__ __launch_bounds__(256) void consistency_fwd_kernel(const f32x4* __restrict__ z1, const f32x4* __restrict__ z2,
                                                              float inv_t, int64_t pixels, int classes,
                                                              double* __restrict__ partials) {
  constexpr int LDC = LDC4 * 4;
  __shared__ double red[4];
  double local = 0.0;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 256) {
    float l1[LDC], l2[LDC];
    load_log_softmax<LDC4>(z1, p, classes, inv_t, l1);
    load_log_softmax<LDC4>(z2, p, classes, inv_t, l2);
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < LDC; ++c)
      if (c < classes) acc += (expf(l1[c]) - expf(l2[c])) * (l1[c] - l2[c]);
    local += (double)acc;
  }
  local = wave_sum_d(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// dz1 = s/T * [ (p1 - p2) + p1 * ((l1 - l2) - kl12) ],  dz2 symmetric;  s = grad_out * weight / (2 * batch)
template <int LDC4>
__global__ __launch_bounds__(256) void consistency_bwd_kernel(const f32x4* __restrict__ z1, const f32x4* __restrict__ z2,
                                                              float inv_t, const float* __restrict__ grad_out, float weight,
                                                              int batch, int64_t pixels, int classes, f32x4* __restrict__ d1,
                                                              f32x4* __restrict__ d2, int accumulate) {
  constexpr int LDC = LDC4 * 4;
  const float s = (grad_out ? *grad_out : 1.f) * weight * inv_t / (2.f * (float)batch);
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 256) {
    float p1[LDC], p2[LDC], dl[LDC];
    load_log_softmax<LDC4>(z1, p, classes, inv_t, p1);
    load_log_softmax<LDC4>(z2, p, classes, inv_t, p2);
    float kl12 = 0.f, kl21 = 0.f;
#pragma unroll
    for (int c = 0; c < LDC; ++c) {
      dl[c] = p1[c] - p2[c];                                   // log p1 - log p2 (0 in the pad lanes)
      p1[c] = c < classes ? expf(p1[c]) : 0.f;
      p2[c] = c < classes ? expf(p2[c]) : 0.f;
      kl12 += p1[c] * dl[c];
      kl21 -= p2[c] * dl[c];
    }
#pragma unroll
    for (int k = 0; k < LDC4; ++k) {
      f32x4 a, b;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = k * 4 + e;
        a[e] = c < classes ? s * ((p1[c] - p2[c]) + p1[c] * (dl[c] - kl12)) : 0.f;
        b[e] = c < classes ? s * ((p2[c] - p1[c]) + p2[c] * (-dl[c] - kl21)) : 0.f;
      }
      if (d1) {
        if (accumulate) a += d1[p * LDC4 + k];
        d1[p * LDC4 + k] = a;
      }
      if (d2) {
        if (accumulate) b += d2[p * LDC4 + k];
        d2[p * LDC4 + k] = b;
      }
    }
  }
}

static int check_seg(const void* logits, int64_t pixels, int classes, int ldc, const char* who) {
  UDASEG_CHECK_ARG(logits != nullptr && pixels > 0 && classes > 0 && classes <= ldc && ldc % 4 == 0 && ldc <= 32,
                   "%s: need classes <= ldc <= 32, ldc %% 4 == 0 (classes=%d ldc=%d)", who, classes, ldc);
  return UDASEG_OK;
}
static int grid_pix(int64_t pixels) {
  const int64_t g = (pixels + 255) / 256;
  return (int)(g > SL_BLOCKS ? SL_BLOCKS : g);
}

#define SEG_DISPATCH(KERN, GRID, ...)                                                                       \
  switch (ldc / 4) {                                                                                        \
    case 1: hipLaunchKernelGGL(KERN<1>, GRID, dim3(256), 0, st, __VA_ARGS__); break;                        \
    case 2: hipLaunchKernelGGL(KERN<2>, GRID, dim3(256), 0, st, __VA_ARGS__); break;                        \
    case 3: hipLaunchKernelGGL(KERN<3>, GRID, dim3(256), 0, st, __VA_ARGS__); break;                        \
    case 4: hipLaunchKernelGGL(KERN<4>, GRID, dim3(256), 0, st, __VA_ARGS__); break;                        \
    case 5: hipLaunchKernelGGL(KERN<5>, GRID, dim3(256), 0, st, __VA_ARGS__); break;                        \
    case 6: hipLaunchKernelGGL(KERN<6>, GRID, dim3(256), 0, st, __VA_ARGS__); break;                        \
    case 7: hipLaunchKernelGGL(KERN<7>, GRID, dim3(256), 0, st, __VA_ARGS__); break;                        \
    default: hipLaunchKernelGGL(KERN<8>, GRID, dim3(256), 0, st, __VA_ARGS__); break;                       \
  }

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_seg_partials(void) { return SL_BLOCKS; }

extern "C" int udaseg_dice_fwd(const float* logits, const int64_t* target, int batch, int64_t pix_per_image, int classes, int ldc,
                               float smooth, float eps, int pooled, double* sums, float* coef, float* loss, void* stream) {
  int rc = check_seg(logits, pix_per_image, classes, ldc, "dice_fwd");
  if (rc) return rc;
  UDASEG_CHECK_ARG(target && sums && coef && loss && batch > 0, "dice_fwd: NULL pointer");
  hipStream_t st = as_stream(stream);
  const int gx = (int)((pix_per_image + 255) / 256 > 256 ? 256 : (pix_per_image + 255) / 256);
  SEG_DISPATCH(dice_stats_kernel, dim3(gx, batch), (const f32x4*)logits, target, pix_per_image, classes, sums)
  UDASEG_LAUNCH_CHECK("dice_stats launch");
  hipLaunchKernelGGL(dice_finish_kernel, dim3(1), dim3(64), 0, st, sums, batch, classes, smooth, eps, pooled, loss, coef);
  UDASEG_LAUNCH_CHECK("dice_finish launch");
  return UDASEG_OK;
}

extern "C" int udaseg_dice_bwd(const float* logits, const int64_t* target, const float* coef, const float* grad_out, float weight,
                               int batch, int64_t pix_per_image, int classes, int ldc, float* dlogits, int accumulate,
                               void* stream) {
  int rc = check_seg(logits, pix_per_image, classes, ldc, "dice_bwd");
  if (rc) return rc;
  UDASEG_CHECK_ARG(target && coef && dlogits && batch > 0, "dice_bwd: NULL pointer");
  hipStream_t st = as_stream(stream);
  SEG_DISPATCH(dice_bwd_kernel, dim3(grid_pix(pix_per_image * batch)), (const f32x4*)logits, target, coef, grad_out, weight,
               pix_per_image, batch, classes, (f32x4*)dlogits, accumulate)
  UDASEG_LAUNCH_CHECK("dice_bwd launch");
  return UDASEG_OK;
}

extern "C" int udaseg_focal_fwd(const float* logits, const int64_t* target, const float* class_weights, float alpha, float gamma,
                                int64_t pixels, int classes, int ldc, int mean, double* partials, float* loss, int accumulate,
                                void* stream) {
  int rc = check_seg(logits, pixels, classes, ldc, "focal_fwd");
  if (rc) return rc;
  UDASEG_CHECK_ARG(target && partials && loss, "focal_fwd: NULL pointer");
  hipStream_t st = as_stream(stream);
  const int grid = grid_pix(pixels);
  SEG_DISPATCH(focal_fwd_kernel, dim3(grid), (const f32x4*)logits, target, class_weights, alpha, gamma, pixels, classes, partials)
  UDASEG_LAUNCH_CHECK("focal_fwd launch");
  hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(64), 0, st, partials, grid, mean ? (double)pixels : 1.0, loss, accumulate);
  UDASEG_LAUNCH_CHECK("focal_finish launch");
  return UDASEG_OK;
}

extern "C" int udaseg_focal_bwd(const float* logits, const int64_t* target, const float* class_weights, float alpha, float gamma,
                                const float* grad_out, float weight, int64_t pixels, int classes, int ldc, float* dlogits,
                                int accumulate, void* stream) {
  int rc = check_seg(logits, pixels, classes, ldc, "focal_bwd");
  if (rc) return rc;
  UDASEG_CHECK_ARG(target && dlogits, "focal_bwd: NULL pointer");
  hipStream_t st = as_stream(stream);
  SEG_DISPATCH(focal_bwd_kernel, dim3(grid_pix(pixels)), (const f32x4*)logits, target, class_weights, alpha, gamma, grad_out, weight,
               pixels, classes, (f32x4*)dlogits, accumulate)
  UDASEG_LAUNCH_CHECK("focal_bwd launch");
  return UDASEG_OK;
}

extern "C" int udaseg_consistency_fwd(const float* z1, const float* z2, float temperature, int batch, int64_t pixels, int classes,
                                      int ldc, double* partials, float* loss, void* stream) {
  int rc = check_seg(z1, pixels, classes, ldc, "consistency_fwd");
  if (rc) return rc;
  UDASEG_CHECK_ARG(z2 && partials && loss && batch > 0 && temperature > 0.f, "consistency_fwd: bad arguments");
  hipStream_t st = as_stream(stream);
  const int grid = grid_pix(pixels);
  SEG_DISPATCH(consistency_fwd_kernel, dim3(grid), (const f32x4*)z1, (const f32x4*)z2, 1.f / temperature, pixels, classes, partials)
  UDASEG_LAUNCH_CHECK("consistency_fwd launch");
  // (KL(p2||p1) + KL(p1||p2)) / 2, each 'batchmean' = sum / batch
  hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(64), 0, st, partials, grid, 2.0 * (double)batch, loss, 0);
  UDASEG_LAUNCH_CHECK("consistency_finish launch");
  return UDASEG_OK;
}

extern "C" int udaseg_consistency_bwd(const float* z1, const float* z2, float temperature, const float* grad_out, float weight,
                                      int batch, int64_t pixels, int classes, int ldc, float* d1, float* d2, int accumulate,
                                      void* stream) {
  int rc = check_seg(z1, pixels, classes, ldc, "consistency_bwd");
  if (rc) return rc;
  UDASEG_CHECK_ARG(z2 && (d1 || d2) && batch > 0 && temperature > 0.f, "consistency_bwd: bad arguments");
  hipStream_t st = as_stream(stream);
  SEG_DISPATCH(consistency_bwd_kernel, dim3(grid_pix(pixels)), (const f32x4*)z1, (const f32x4*)z2, 1.f / temperature, grad_out,
               weight, batch, pixels, classes, (f32x4*)d1, (f32x4*)d2, accumulate)
  UDASEG_LAUNCH_CHECK("consistency_bwd launch");
  return UDASEG_OK;
}
