// Loss kernels (no MFMA; wavefront reductions), fp32 (gfx950).
//
//  * per-pixel cross entropy, mean reduction, no class weights / ignore_index:
//      nn.CrossEntropyLoss() constructed at reference src/models/train.py:208, applied at :342,:403 and at
//      src/models/adversarial_trainer.py:105,155.
//  * discriminator tail  AdaptiveAvgPool2d(1) -> Flatten -> Linear(512,1) -> Sigmoid
//      (src/models/discriminator.py:37-42) and its autograd.
//  * AdversarialLoss' nn.BCEWithLogitsLoss terms (src/models/losses.py:16,33-36,51).  The reference feeds the
//      discriminator's *probabilities* into the with-logits loss; these kernels are agnostic: they compute
//      mean(softplus(x) - x*label) of whatever x is.
#include "common.h"

namespace udaseg {

constexpr int CE_BLOCKS = 1024;
constexpr int CE_MAXC = 64;

// One thread per pixel, logits row in registers (classes <= 64, ldc % 4 == 0).  HBM-bound: 4*ldc B/pixel read.
template <int LDC4>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const f32x4* __restrict__ logits, const int64_t* __restrict__ target,
                                                     int64_t pixels, int classes, float* __restrict__ lse,
                                                     double* __restrict__ partials) {
  __shared__ double red[4];
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  double local = 0.0;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += T) {
    f32x4 v[LDC4];
#pragma unroll
    for (int k = 0; k < LDC4; ++k) v[k] = logits[p * LDC4 + k];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < LDC4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k * 4 + e < classes) mx = fmaxf(mx, v[k][e]);
    float sum = 0.f;
    const int t = (int)target[p];
    float xt = 0.f;
#pragma unroll
    for (int k = 0; k < LDC4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k * 4 + e < classes) {
          sum += expf(v[k][e] - mx);
          if (k * 4 + e == t) xt = v[k][e];
        }
    const float l = mx + logf(sum);
    lse[p] = l;
    local += (double)(l - xt);
  }
  local = wave_sum_d(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void ce_finish_kernel(const double* __restrict__ partials, int n, int64_t pixels, float* __restrict__ loss) {
  // single wave: deterministic final sum
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partials[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) *loss = (float)(s / (double)pixels);
}

// Generic fallback (ldc > 32): one thread per pixel, strided 16-B stores.
template <int LDC4>
__global__ __launch_bounds__(256) void ce_bwd_simple_kernel(const f32x4* __restrict__ logits, const int64_t* __restrict__ target,
                                                            const float* __restrict__ lse, const float* __restrict__ grad_out,
                                                            int64_t pixels, int classes, f32x4* __restrict__ dlogits) {
  const float scale = (grad_out ? *grad_out : 1.f) / (float)pixels;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += T) {
    const float l = lse[p];
    const int t = (int)target[p];
#pragma unroll
    for (int k = 0; k < LDC4; ++k) {
      const f32x4 v = logits[p * LDC4 + k];
      f32x4 g;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = k * 4 + e;
        g[e] = c < classes ? (expf(v[e] - l) - (c == t ? 1.f : 0.f)) * scale : 0.f;
      }
      dlogits[p * LDC4 + k] = g;
    }
  }
}

// ldc <= 32: blocks walk 256-pixel chunks; gradients go through an LDS tile so the HBM stores are whole contiguous
// lines (per-thread 96-B-strided 16-B stores ran at 1.7 TB/s), and the same tile yields the per-class column sums =
// the segmentation head's bias gradient (saves a separate 200 MB pass).  colpart: [gridDim.x][LDC] or null.
template <int LDC4>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const f32x4* __restrict__ logits, const int64_t* __restrict__ target,
                                                     const float* __restrict__ lse, const float* __restrict__ grad_out,
                                                     int64_t pixels, int classes, f32x4* __restrict__ dlogits,
                                                     float* __restrict__ colpart) {
  constexpr int LDC = LDC4 * 4;
  constexpr int NG = 256 / LDC;  // row groups for the column sums
  __shared__ __attribute__((aligned(16))) float tile[256 * LDC];
  __shared__ float colred[NG * LDC];
  const int tid = threadIdx.x;
  const float scale = (grad_out ? *grad_out : 1.f) / (float)pixels;
  const int64_t nchunks = (pixels + 255) / 256;
  const int cc = tid % LDC, rg = tid / LDC;
  float colacc = 0.f;
  for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int64_t p = chunk * 256 + tid;
    if (p < pixels) {
      const float l = lse[p];
      const int t = (int)target[p];
#pragma unroll
      for (int k = 0; k < LDC4; ++k) {
        const f32x4 v = logits[p * LDC4 + k];
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = k * 4 + e;
          g[e] = c < classes ? (expf(v[e] - l) - (c == t ? 1.f : 0.f)) * scale : 0.f;
        }
        *reinterpret_cast<f32x4*>(tile + tid * LDC + k * 4) = g;
      }
    } else {
#pragma unroll
      for (int k = 0; k < LDC4; ++k) *reinterpret_cast<f32x4*>(tile + tid * LDC + k * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int64_t base4 = chunk * 256 * LDC4, lim4 = pixels * LDC4;
#pragma unroll
    for (int it = 0; it < LDC4; ++it) {
      const int idx = it * 256 + tid;
      if (base4 + idx < lim4) dlogits[base4 + idx] = reinterpret_cast<const f32x4*>(tile)[idx];
    }
    if (colpart && rg < NG) {
      for (int r = rg; r < 256; r += NG) colacc += tile[r * LDC + cc];
    }
    __syncthreads();
  }
  if (colpart) {
    if (rg < NG) colred[rg * LDC + cc] = colacc;
    __syncthreads();
    if (tid < LDC) {
      float s = 0.f;
      for (int g2 = 0; g2 < NG; ++g2) s += colred[g2 * LDC + tid];
      colpart[(size_t)blockIdx.x * LDC + tid] = s;
    }
  }
}

// Forward and backward in ONE pass over the logits (round 5): the loss partials of ce_fwd_kernel (same pixel-to-thread map, same
// block reduction: the loss is bit-identical) and the gradient of ce_bwd_kernel for an upstream gradient of 1 (scale 1 / pixels; same
// expressions: bit-identical), through the same LDS tile with the same column sums.  The training step read the 200 MB of logits
// twice (ce_fwd for the loss, ce_bwd for the gradient); loss.backward() hands this loss a gradient of exactly 1, and
// scale_unless_one_kernel below covers every other caller.
template <int LDC4>
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const f32x4* __restrict__ logits, const int64_t* __restrict__ target,
                                                         int64_t pixels, int classes, double* __restrict__ partials,
                                                         f32x4* __restrict__ dlogits, float* __restrict__ colpart) {
  constexpr int LDC = LDC4 * 4;
  constexpr int NG = 256 / LDC;
  __shared__ __attribute__((aligned(16))) float tile[256 * LDC];
  __shared__ float colred[NG * LDC];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const float scale = 1.f / (float)pixels;
  const int64_t nchunks = (pixels + 255) / 256;
  const int cc = tid % LDC, rg = tid / LDC;
  float colacc = 0.f;
  double local = 0.0;
  for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int64_t p = chunk * 256 + tid;
    if (p < pixels) {
      f32x4 v[LDC4];
#pragma unroll
      for (int k = 0; k < LDC4; ++k) v[k] = logits[p * LDC4 + k];
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < LDC4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k * 4 + e < classes) mx = fmaxf(mx, v[k][e]);
      float sum = 0.f;
      const int t = (int)target[p];
      float xt = 0.f;
#pragma unroll
      for (int k = 0; k < LDC4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k * 4 + e < classes) {
            sum += expf(v[k][e] - mx);
            if (k * 4 + e == t) xt = v[k][e];
          }
      const float l = mx + logf(sum);
      local += (double)(l - xt);
#pragma unroll
      for (int k = 0; k < LDC4; ++k) {
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = k * 4 + e;
          g[e] = c < classes ? (expf(v[k][e] - l) - (c == t ? 1.f : 0.f)) * scale : 0.f;
        }
        *reinterpret_cast<f32x4*>(tile + tid * LDC + k * 4) = g;
      }
    } else {
#pragma unroll
      for (int k = 0; k < LDC4; ++k) *reinterpret_cast<f32x4*>(tile + tid * LDC + k * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int64_t base4 = chunk * 256 * LDC4, lim4 = pixels * LDC4;
#pragma unroll
    for (int it = 0; it < LDC4; ++it) {
      const int idx = it * 256 + tid;
      if (base4 + idx < lim4) dlogits[base4 + idx] = reinterpret_cast<const f32x4*>(tile)[idx];
    }
    if (colpart && rg < NG) {
      for (int r = rg; r < 256; r += NG) colacc += tile[r * LDC + cc];
    }
    __syncthreads();
  }
  local = wave_sum_d(local);
  if ((tid & 63) == 0) red[tid >> 6] = local;
  if (colpart && rg < NG) colred[rg * LDC + cc] = colacc;
  __syncthreads();
  if (tid == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  if (colpart && tid < LDC) {
    float s = 0.f;
    for (int g2 = 0; g2 < NG; ++g2) s += colred[g2 * LDC + tid];
    colpart[(size_t)blockIdx.x * LDC + tid] = s;
  }
}

// x[i] *= *g unless *g == 1 (then the launch returns at once: the gradient made for an upstream gradient of 1 is already right)
__global__ void scale_unless_one_kernel(f32x4* __restrict__ x, int64_t n4, float* __restrict__ x2, int n2, const float* __restrict__ g) {
  const float s = *g;
  if (s == 1.f) return;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = i0; i < n4; i += T) x[i] = x[i] * s;
  if (x2 != nullptr && i0 < n2) x2[i0] *= s;
}

// colsum[c] (+)= sum over blocks of colpart[b][c]: one 256-thread block per column, fixed-order tree => deterministic
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ colpart, int nblocks, int ldc,
                                                            float* __restrict__ colsum, int accumulate) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += colpart[(size_t)b * ldc + c];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (red[0] + red[1]) + (red[2] + red[3]);
    colsum[c] = accumulate ? colsum[c] + t : t;
  }
}

// ---- validation metrics: per-pixel argmax + confusion matrix (SegmentationTrainer.calculate_metrics, reference
// src/models/train.py:225-243; confusion-matrix definition src/analysis/metrics.py:17-29: bincount(C*true + pred)).
// One thread per pixel, row in registers; the block's histogram lives in LDS (classes <= 32 -> 4 KiB of counters), one
// global atomic per non-empty cell per block.  argmax tie rule = torch.argmax: first maximal index.
template <int LDC4>
__global__ __launch_bounds__(256) void argmax_confusion_kernel(const f32x4* __restrict__ logits, const int64_t* __restrict__ target,
                                                               int64_t pixels, int classes, unsigned long long* __restrict__ cm,
                                                               int64_t* __restrict__ pred_out) {
  __shared__ unsigned int hist[32 * 32];
  for (int i = threadIdx.x; i < classes * classes; i += 256) hist[i] = 0;
  __syncthreads();
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += T) {
    float best = -INFINITY;
    int bi = 0;
#pragma unroll
    for (int k = 0; k < LDC4; ++k) {
      const f32x4 v = logits[p * LDC4 + k];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = k * 4 + e;
        if (c < classes && (v[e] > best || (c == 0))) {
          if (c == 0 || v[e] > best) { best = v[e]; bi = c; }
        }
      }
    }
    if (pred_out) pred_out[p] = bi;
    const int t = (int)target[p];
    if ((unsigned)t < (unsigned)classes) atomicAdd(&hist[t * classes + bi], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < classes * classes; i += 256)
    if (hist[i]) atomicAdd(&cm[i], (unsigned long long)hist[i]);
}

// ---- discriminator tail ---------------------------------------------------------------------------------------
constexpr int GAP_SPLITS = 32;

// partial[n][s][c] = sum over the s-th slice of the hw pixels of z[n][p][c]
__global__ void gap_partial_kernel(const f32x4* __restrict__ z, f32x4* __restrict__ partial, int hw, int c4, int splits) {
  const int ni = blockIdx.y, s = blockIdx.x;
  const int per = (hw + splits - 1) / splits;
  const int p0 = s * per, p1 = min(hw, p0 + per);
  for (int q = threadIdx.x; q < c4; q += blockDim.x) {
    f32x4 acc = {0, 0, 0, 0};
    for (int p = p0; p < p1; ++p) acc += z[((int64_t)ni * hw + p) * c4 + q];
    partial[((int64_t)ni * splits + s) * c4 + q] = acc;
  }
}

// one block per image: pooled = sum_s partial / hw ; p = sigmoid(dot(pooled, w) + b)
__global__ __launch_bounds__(256) void gap_linear_sigmoid_kernel(const float* __restrict__ partial, const float* __restrict__ w,
                                                                 const float* __restrict__ b, float* __restrict__ pooled,
                                                                 float* __restrict__ p, int hw, int c, int splits,
                                                                 int sigmoid) {
  __shared__ float red[4];
  const int ni = blockIdx.x;
  float dot = 0.f;
  const float inv = 1.f / (float)hw;
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += partial[((int64_t)ni * splits + k) * c + ch];
    s *= inv;
    pooled[(int64_t)ni * c + ch] = s;
    dot += s * w[ch];
  }
  dot = wave_sum(dot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = red[0] + red[1] + red[2] + red[3] + b[0];
    p[ni] = sigmoid ? 1.f / (1.f + expf(-t)) : t;
  }
}

// dz[n][p][c] = dlogit[n] * w[c] / hw, broadcast over the hw pixels
__global__ void gap_bwd_broadcast_kernel(const float* __restrict__ dp, const float* __restrict__ p, const f32x4* __restrict__ w,
                                         f32x4* __restrict__ dz, int n, int hw, int c4, int sigmoid) {
  const int64_t total = (int64_t)n * hw * c4;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const float inv = 1.f / (float)hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += T) {
    const int q = (int)(i % c4);
    const int ni = (int)(i / ((int64_t)hw * c4));
    const float pv = sigmoid ? p[ni] : 0.f;
    const float dl = dp[ni] * (sigmoid ? pv * (1.f - pv) : 1.f) * inv;
    dz[i] = w[q] * dl;
  }
}

// dw[c] (+)= sum_n dlogit[n]*pooled[n][c] ; db (+)= sum_n dlogit[n]
__global__ void gap_bwd_param_kernel(const float* __restrict__ dp, const float* __restrict__ p, const float* __restrict__ pooled,
                                     float* __restrict__ dw, float* __restrict__ db, int n, int c, int accumulate, int sigmoid) {
  for (int ch = blockIdx.x * blockDim.x + threadIdx.x; ch < c; ch += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int ni = 0; ni < n; ++ni) s += dp[ni] * (sigmoid ? p[ni] * (1.f - p[ni]) : 1.f) * pooled[(int64_t)ni * c + ch];
    dw[ch] = accumulate ? dw[ch] + s : s;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int ni = 0; ni < n; ++ni) s += dp[ni] * (sigmoid ? p[ni] * (1.f - p[ni]) : 1.f);
    db[0] = accumulate ? db[0] + s : s;
  }
}

// ---- BCE with logits on a short vector: single wave -------------------------------------------------------------
__device__ __forceinline__ float softplus_f(float x) {
  // log(1 + exp(x)) = max(x,0) + log1p(exp(-|x|))
  return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}

__global__ void bce_fwd_kernel(const float* __restrict__ x, int n, float label, float weight, float* __restrict__ loss,
                               int accumulate) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) {
    const float v = x[i];
    // torch: (1 - y) * x + softplus(-x)   [= max(-x,0) + log1p(exp(-|x|)) form]
    s += (1.f - label) * v + softplus_f(-v);
  }
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    const float l = weight * (s / (float)n);
    *loss = accumulate ? *loss + l : l;
  }
}

__global__ void bce_bwd_kernel(const float* __restrict__ x, int n, float label, float weight, const float* __restrict__ grad_out,
                               float* __restrict__ dx, int accumulate) {
  const float g = (grad_out ? *grad_out : 1.f) * weight / (float)n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float v = x[i];
    const float sg = 1.f / (1.f + expf(-v));
    const float d = (sg - label) * g;
    dx[i] = accumulate ? dx[i] + d : d;
  }
}

// per-sample targets (nn.BCEWithLogitsLoss()(x, y), reference src/models/uda.py:85,96)
__global__ void bce_target_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, float weight,
                                      float* __restrict__ loss, int accumulate) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += (1.f - y[i]) * x[i] + softplus_f(-x[i]);
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    const float l = weight * (s / (float)n);
    *loss = accumulate ? *loss + l : l;
  }
}

__global__ void bce_target_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, float weight,
                                      const float* __restrict__ grad_out, float* __restrict__ dx, int accumulate) {
  const float g = (grad_out ? *grad_out : 1.f) * weight / (float)n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float d = (1.f / (1.f + expf(-x[i])) - y[i]) * g;
    dx[i] = accumulate ? dx[i] + d : d;
  }
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_ce_partials(void) { return CE_BLOCKS; }

extern "C" int udaseg_ce_fwd(const float* logits, const int64_t* target, int64_t pixels, int classes, int ldc, float* lse,
                             double* partials, float* loss, void* stream) {
  UDASEG_CHECK_ARG(logits && target && lse && partials && loss, "ce_fwd: NULL pointer");
  UDASEG_CHECK_ARG(pixels > 0 && classes > 0 && classes <= ldc && ldc % 4 == 0 && ldc <= CE_MAXC,
                   "ce_fwd: need 0 < classes <= ldc <= %d, ldc %% 4 == 0 (classes=%d ldc=%d)", CE_MAXC, classes, ldc);
  hipStream_t st = as_stream(stream);
  int grid = (int)((pixels + 255) / 256 > CE_BLOCKS ? CE_BLOCKS : (pixels + 255) / 256);
#define CE_FWD_CASE(L)                                                                                                 \
  case L:                                                                                                              \
    hipLaunchKernelGGL(ce_fwd_kernel<L>, dim3(grid), dim3(256), 0, st, (const f32x4*)logits, target, pixels, classes, lse, \
                       partials);                                                                                      \
    break;
  switch (ldc / 4) {
    CE_FWD_CASE(1) CE_FWD_CASE(2) CE_FWD_CASE(3) CE_FWD_CASE(4) CE_FWD_CASE(5) CE_FWD_CASE(6) CE_FWD_CASE(7) CE_FWD_CASE(8)
    CE_FWD_CASE(9) CE_FWD_CASE(10) CE_FWD_CASE(11) CE_FWD_CASE(12) CE_FWD_CASE(13) CE_FWD_CASE(14) CE_FWD_CASE(15) CE_FWD_CASE(16)
    default:
      set_error("ce_fwd: unsupported ldc %d", ldc);
      return UDASEG_E_UNSUPPORTED;
  }
#undef CE_FWD_CASE
  UDASEG_LAUNCH_CHECK("ce_fwd launch");
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(64), 0, st, partials, grid, pixels, loss);
  UDASEG_LAUNCH_CHECK("ce_finish launch");
  return UDASEG_OK;
}

extern "C" int udaseg_ce_fwd_bwd(const float* logits, const int64_t* target, int64_t pixels, int classes, int ldc, double* partials,
                                 float* loss, float* dlogits, float* colsum_partials, float* colsum, void* stream) {
  UDASEG_CHECK_ARG(logits && target && partials && loss && dlogits, "ce_fwd_bwd: NULL pointer");
  UDASEG_CHECK_ARG(pixels > 0 && classes > 0 && classes <= ldc && ldc % 4 == 0 && ldc <= 32,
                   "ce_fwd_bwd: need 0 < classes <= ldc <= 32, ldc %% 4 == 0 (classes=%d ldc=%d)", classes, ldc);
  UDASEG_CHECK_ARG((colsum == nullptr) == (colsum_partials == nullptr), "ce_fwd_bwd: colsum and colsum_partials come together");
  hipStream_t st = as_stream(stream);
  const int64_t nchunks = (pixels + 255) / 256;
  const int grid = (int)(nchunks > CE_BLOCKS ? CE_BLOCKS : nchunks);
#define CE_FB_CASE(L)                                                                                                    \
  case L:                                                                                                                \
    hipLaunchKernelGGL(ce_fwd_bwd_kernel<L>, dim3(grid), dim3(256), 0, st, (const f32x4*)logits, target, pixels, classes,  \
                       partials, (f32x4*)dlogits, colsum_partials);                                                      \
    break;
  switch (ldc / 4) {
    CE_FB_CASE(1) CE_FB_CASE(2) CE_FB_CASE(3) CE_FB_CASE(4) CE_FB_CASE(5) CE_FB_CASE(6) CE_FB_CASE(7) CE_FB_CASE(8)
    default:
      set_error("ce_fwd_bwd: unsupported ldc %d", ldc);
      return UDASEG_E_UNSUPPORTED;
  }
#undef CE_FB_CASE
  UDASEG_LAUNCH_CHECK("ce_fwd_bwd launch");
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(64), 0, st, partials, grid, pixels, loss);
  UDASEG_LAUNCH_CHECK("ce_finish launch");
  if (colsum) {
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(ldc), dim3(256), 0, st, colsum_partials, grid, ldc, colsum, 0);
    UDASEG_LAUNCH_CHECK("colsum_finish launch");
  }
  return UDASEG_OK;
}

extern "C" int udaseg_scale_unless_one(float* x, int64_t count, float* x2, int count2, const float* g, void* stream) {
  UDASEG_CHECK_ARG(x && g && count > 0 && count % 4 == 0 && count2 >= 0 && count2 <= 256 && (x2 != nullptr || count2 == 0),
                   "scale_unless_one: bad arguments");
  const int64_t n4 = count / 4;
  int64_t grid = (n4 + 1023) / 1024;
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(scale_unless_one_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), (f32x4*)x, n4, x2, count2, g);
  UDASEG_LAUNCH_CHECK("scale_unless_one launch");
  return UDASEG_OK;
}

extern "C" int udaseg_ce_bwd(const float* logits, const int64_t* target, const float* lse, const float* grad_out,
                             int64_t pixels, int classes, int ldc, float* dlogits, float* colsum_partials, float* colsum,
                             void* stream) {
  UDASEG_CHECK_ARG(logits && target && lse && dlogits, "ce_bwd: NULL pointer");
  UDASEG_CHECK_ARG(pixels > 0 && classes > 0 && classes <= ldc && ldc % 4 == 0 && ldc <= CE_MAXC, "ce_bwd: bad shape");
  UDASEG_CHECK_ARG((colsum == nullptr) == (colsum_partials == nullptr), "ce_bwd: colsum and colsum_partials come together");
  UDASEG_CHECK_ARG(colsum == nullptr || ldc <= 32, "ce_bwd: fused column sums need ldc <= 32");
  hipStream_t st = as_stream(stream);
  if (ldc <= 32) {
    const int64_t nchunks = (pixels + 255) / 256;
    const int grid = (int)(nchunks > CE_BLOCKS ? CE_BLOCKS : nchunks);
#define CE_BWD_CASE(L)                                                                                                   \
  case L:                                                                                                                \
    hipLaunchKernelGGL(ce_bwd_kernel<L>, dim3(grid), dim3(256), 0, st, (const f32x4*)logits, target, lse, grad_out, pixels, \
                       classes, (f32x4*)dlogits, colsum_partials);                                                       \
    break;
    switch (ldc / 4) {
      CE_BWD_CASE(1) CE_BWD_CASE(2) CE_BWD_CASE(3) CE_BWD_CASE(4) CE_BWD_CASE(5) CE_BWD_CASE(6) CE_BWD_CASE(7) CE_BWD_CASE(8)
      default:
        set_error("ce_bwd: unsupported ldc %d", ldc);
        return UDASEG_E_UNSUPPORTED;
    }
#undef CE_BWD_CASE
    UDASEG_LAUNCH_CHECK("ce_bwd launch");
    if (colsum) {
      hipLaunchKernelGGL(colsum_finish_kernel, dim3(ldc), dim3(256), 0, st, colsum_partials, grid, ldc, colsum, 0);
      UDASEG_LAUNCH_CHECK("colsum_finish launch");
    }
    return UDASEG_OK;
  }
  int grid = (int)((pixels + 255) / 256 > 4096 ? 4096 : (pixels + 255) / 256);
#define CE_BWD_CASE(L)                                                                                                   \
  case L:                                                                                                                \
    hipLaunchKernelGGL(ce_bwd_simple_kernel<L>, dim3(grid), dim3(256), 0, st, (const f32x4*)logits, target, lse, grad_out, \
                       pixels, classes, (f32x4*)dlogits);                                                                \
    break;
  switch (ldc / 4) {
    CE_BWD_CASE(9) CE_BWD_CASE(10) CE_BWD_CASE(11) CE_BWD_CASE(12) CE_BWD_CASE(13) CE_BWD_CASE(14) CE_BWD_CASE(15) CE_BWD_CASE(16)
    default:
      set_error("ce_bwd: unsupported ldc %d", ldc);
      return UDASEG_E_UNSUPPORTED;
  }
#undef CE_BWD_CASE
  UDASEG_LAUNCH_CHECK("ce_bwd launch");
  return UDASEG_OK;
}

extern "C" int udaseg_argmax_confusion(const float* logits, const int64_t* target, int64_t pixels, int classes, int ldc,
                                       int64_t* confusion, int64_t* pred, void* stream) {
  UDASEG_CHECK_ARG(logits && target && confusion, "argmax_confusion: NULL pointer");
  UDASEG_CHECK_ARG(pixels > 0 && classes > 0 && classes <= 32 && classes <= ldc && ldc % 4 == 0 && ldc <= 32,
                   "argmax_confusion: need classes <= 32, ldc %% 4 == 0 (classes=%d ldc=%d)", classes, ldc);
  hipStream_t st = as_stream(stream);
  int grid = (int)((pixels + 255) / 256 > 1024 ? 1024 : (pixels + 255) / 256);
#define AMX_CASE(L)                                                                                                      \
  case L:                                                                                                                \
    hipLaunchKernelGGL(argmax_confusion_kernel<L>, dim3(grid), dim3(256), 0, st, (const f32x4*)logits, target, pixels,   \
                       classes, (unsigned long long*)confusion, pred);                                                   \
    break;
  switch (ldc / 4) {
    AMX_CASE(1) AMX_CASE(2) AMX_CASE(3) AMX_CASE(4) AMX_CASE(5) AMX_CASE(6) AMX_CASE(7) AMX_CASE(8)
    default:
      set_error("argmax_confusion: unsupported ldc %d", ldc);
      return UDASEG_E_UNSUPPORTED;
  }
#undef AMX_CASE
  UDASEG_LAUNCH_CHECK("argmax_confusion launch");
  return UDASEG_OK;
}

extern "C" int udaseg_gap_splits(int hw) {
  int s = GAP_SPLITS;
  if (s > hw) s = hw;
  return s < 1 ? 1 : s;
}

extern "C" int udaseg_gap_linear_sigmoid_fwd(const float* z, const float* w, const float* b, float* partial, float* pooled,
                                             float* p, int n, int hw, int c, void* stream) {
  UDASEG_CHECK_ARG(z && w && b && partial && pooled && p, "gap_linear_sigmoid_fwd: NULL pointer");
  UDASEG_CHECK_ARG(n > 0 && hw > 0 && c > 0 && c % 4 == 0, "gap_linear_sigmoid_fwd: bad shape");
  hipStream_t st = as_stream(stream);
  const int splits = udaseg_gap_splits(hw);
  const int bs = (c / 4) < 256 ? (((c / 4) + 63) / 64) * 64 : 256;
  hipLaunchKernelGGL(gap_partial_kernel, dim3(splits, n), dim3(bs), 0, st, (const f32x4*)z, (f32x4*)partial, hw, c / 4, splits);
  UDASEG_LAUNCH_CHECK("gap_partial launch");
  hipLaunchKernelGGL(gap_linear_sigmoid_kernel, dim3(n), dim3(256), 0, st, partial, w, b, pooled, p, hw, c, splits, 1);
  UDASEG_LAUNCH_CHECK("gap_linear_sigmoid launch");
  return UDASEG_OK;
}

extern "C" int udaseg_gap_linear_sigmoid_bwd(const float* dp, const float* p, const float* pooled, const float* w, float* dz,
                                             float* dw, float* db, int n, int hw, int c, int accumulate_param, void* stream) {
  UDASEG_CHECK_ARG(dp && p && pooled && w && dz && dw && db, "gap_linear_sigmoid_bwd: NULL pointer");
  UDASEG_CHECK_ARG(n > 0 && hw > 0 && c > 0 && c % 4 == 0, "gap_linear_sigmoid_bwd: bad shape");
  hipStream_t st = as_stream(stream);
  const int64_t total = (int64_t)n * hw * (c / 4);
  int grid = (int)((total + 511) / 512 > 2048 ? 2048 : (total + 511) / 512);
  hipLaunchKernelGGL(gap_bwd_broadcast_kernel, dim3(grid), dim3(256), 0, st, dp, p, (const f32x4*)w, (f32x4*)dz, n, hw, c / 4, 1);
  UDASEG_LAUNCH_CHECK("gap_bwd_broadcast launch");
  hipLaunchKernelGGL(gap_bwd_param_kernel, dim3((c + 255) / 256), dim3(256), 0, st, dp, p, pooled, dw, db, n, c, accumulate_param, 1);
  UDASEG_LAUNCH_CHECK("gap_bwd_param launch");
  return UDASEG_OK;
}

// dtype-independent stages of the tail, for the bf16 path (its pooling partials / broadcast live in elem_bf16.hip)
extern "C" int udaseg_gap_finish(const float* partial, const float* w, const float* b, float* pooled, float* p, int n, int hw,
                                 int c, void* stream) {
  UDASEG_CHECK_ARG(partial && w && b && pooled && p && n > 0 && hw > 0 && c > 0, "gap_finish: bad arguments");
  hipLaunchKernelGGL(gap_linear_sigmoid_kernel, dim3(n), dim3(256), 0, as_stream(stream), partial, w, b, pooled, p, hw, c,
                     udaseg_gap_splits(hw), 1);
  UDASEG_LAUNCH_CHECK("gap_finish launch");
  return UDASEG_OK;
}

extern "C" int udaseg_gap_bwd_param(const float* dp, const float* p, const float* pooled, float* dw, float* db, int n, int c,
                                    int accumulate_param, void* stream) {
  UDASEG_CHECK_ARG(dp && p && pooled && dw && db && n > 0 && c > 0, "gap_bwd_param: bad arguments");
  hipLaunchKernelGGL(gap_bwd_param_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), dp, p, pooled, dw, db, n, c,
                     accumulate_param, 1);
  UDASEG_LAUNCH_CHECK("gap_bwd_param launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bce_logits_fwd(const float* x, int n, float label, float weight, float* loss, int accumulate,
                                     void* stream) {
  UDASEG_CHECK_ARG(x && loss && n > 0, "bce_logits_fwd: bad arguments");
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(1), dim3(64), 0, as_stream(stream), x, n, label, weight, loss, accumulate);
  UDASEG_LAUNCH_CHECK("bce_fwd launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bce_logits_bwd(const float* x, int n, float label, float weight, const float* grad_out, float* dx,
                                     int accumulate, void* stream) {
  UDASEG_CHECK_ARG(x && dx && n > 0, "bce_logits_bwd: bad arguments");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), x, n, label, weight, grad_out, dx,
                     accumulate);
  UDASEG_LAUNCH_CHECK("bce_bwd launch");
  return UDASEG_OK;
}

// ---- feature-level discriminator tail: Conv2d(c, 1, 1) -> AdaptiveAvgPool2d(1) (reference src/models/uda.py:22-23).
// The two are linear, so mean_hw(w.x + b) = w.mean_hw(x) + b: same kernels as the image-level tail, no sigmoid.
extern "C" int udaseg_gap_linear_fwd(const float* z, const float* w, const float* b, float* partial, float* pooled, float* logit,
                                     int n, int hw, int c, void* stream) {
  UDASEG_CHECK_ARG(z && w && b && partial && pooled && logit, "gap_linear_fwd: NULL pointer");
  UDASEG_CHECK_ARG(n > 0 && hw > 0 && c > 0 && c % 4 == 0, "gap_linear_fwd: bad shape");
  hipStream_t st = as_stream(stream);
  const int splits = udaseg_gap_splits(hw);
  const int bs = (c / 4) < 256 ? (((c / 4) + 63) / 64) * 64 : 256;
  hipLaunchKernelGGL(gap_partial_kernel, dim3(splits, n), dim3(bs), 0, st, (const f32x4*)z, (f32x4*)partial, hw, c / 4, splits);
  UDASEG_LAUNCH_CHECK("gap_partial launch");
  hipLaunchKernelGGL(gap_linear_sigmoid_kernel, dim3(n), dim3(256), 0, st, partial, w, b, pooled, logit, hw, c, splits, 0);
  UDASEG_LAUNCH_CHECK("gap_linear launch");
  return UDASEG_OK;
}

extern "C" int udaseg_gap_linear_bwd(const float* dlogit, const float* pooled, const float* w, float* dz, float* dw, float* db,
                                     int n, int hw, int c, int accumulate_param, void* stream) {
  UDASEG_CHECK_ARG(dlogit && pooled && w && dz && dw && db, "gap_linear_bwd: NULL pointer");
  UDASEG_CHECK_ARG(n > 0 && hw > 0 && c > 0 && c % 4 == 0, "gap_linear_bwd: bad shape");
  hipStream_t st = as_stream(stream);
  const int64_t total = (int64_t)n * hw * (c / 4);
  int grid = (int)((total + 511) / 512 > 2048 ? 2048 : (total + 511) / 512);
  hipLaunchKernelGGL(gap_bwd_broadcast_kernel, dim3(grid), dim3(256), 0, st, dlogit, (const float*)nullptr, (const f32x4*)w,
                     (f32x4*)dz, n, hw, c / 4, 0);
  UDASEG_LAUNCH_CHECK("gap_bwd_broadcast launch");
  hipLaunchKernelGGL(gap_bwd_param_kernel, dim3((c + 255) / 256), dim3(256), 0, st, dlogit, (const float*)nullptr, pooled, dw, db,
                     n, c, accumulate_param, 0);
  UDASEG_LAUNCH_CHECK("gap_bwd_param launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bce_logits_target_fwd(const float* x, const float* target, int n, float weight, float* loss, int accumulate,
                                            void* stream) {
  UDASEG_CHECK_ARG(x && target && loss && n > 0, "bce_logits_target_fwd: bad arguments");
  hipLaunchKernelGGL(bce_target_fwd_kernel, dim3(1), dim3(64), 0, as_stream(stream), x, target, n, weight, loss, accumulate);
  UDASEG_LAUNCH_CHECK("bce_target_fwd launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bce_logits_target_bwd(const float* x, const float* target, int n, float weight, const float* grad_out,
                                            float* dx, int accumulate, void* stream) {
  UDASEG_CHECK_ARG(x && target && dx && n > 0, "bce_logits_target_bwd: bad arguments");
  hipLaunchKernelGGL(bce_target_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), x, target, n, weight, grad_out,
                     dx, accumulate);
  UDASEG_LAUNCH_CHECK("bce_target_bwd launch");
  return UDASEG_OK;
}
