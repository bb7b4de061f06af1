// Training-mode batch norm (+ReLU / LeakyReLU / residual add) forward and backward, NHWC fp32 (gfx950).
//
// Replaces torch.nn.BatchNorm2d / ReLU / LeakyReLU / `out += identity` and their autograd inside smp.Unet
// (reference src/models/train.py:341,343) and DomainDiscriminator (src/models/discriminator.py:21-33).
//
// All kernels here are HBM-bound streams over a [pixels][C] tensor read as float4: the launch is shaped so
// that (total threads) % (C/4) == 0, hence every thread meets the same 4 channels on every grid-stride step
// and keeps its per-channel partials in registers; one LDS pass folds the rows of a block, one f64 atomic
// per channel per block folds the blocks (f64: the cross-block order no longer shows after rounding to f32).
#include "common.h"

namespace udaseg {

// Fold a per-thread 4-channel f64 partial over the rows of the block and add it to dst[q*4 .. q*4+3] (f64 atomics).
// Threads with equal (tid % c4) own the same channels when bs % c4 == 0; otherwise (c4 >= bs) each is unique.
struct d4 {
  double v[4];
};
__device__ __forceinline__ void block_fold_add(const d4& v, double* dst, int c4, int q, d4* red) {
  (void)q;
  block_fold_atomic<4, double>(v.v, dst, c4, reinterpret_cast<double*>(red));
}

__global__ void bn_stats_kernel(const f32x4* __restrict__ y, int64_t n4, int c4, double* __restrict__ sums) {
  // f64 accumulation: E[x^2] - mean^2 then loses nothing to the fp32 partials (the kernel is HBM-bound anyway)
  __shared__ d4 red[256];
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(g % c4);
  d4 s = {{0, 0, 0, 0}}, ss = {{0, 0, 0, 0}};
  for (int64_t i = g; i < n4; i += T) {
    const f32x4 v = y[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double d = (double)v[e];
      s.v[e] += d;
      ss.v[e] += d * d;
    }
  }
  double* rep = sums + (size_t)(blockIdx.x % BN_REPLICAS) * 2 * c4 * 4;
  block_fold_add(s, rep, c4, q, red);
  block_fold_add(ss, rep + (size_t)c4 * 4, c4, q, red);
}

constexpr int MAX_C4 = 1024;  // channels <= 4096

// Per-channel-quad coefficients are computed ONCE per block (threads q < c4, strided) into LDS; every thread then picks
// the quad it streams.  (Summing the 16 replicas in every thread cost more than the streaming itself on small tensors.)
__global__ void bn_apply_kernel(const f32x4* __restrict__ y, const double* __restrict__ sums,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const f32x4* __restrict__ residual, f32x4* __restrict__ z, int64_t n4, int c4,
                                int64_t pixels, float eps, float momentum, float* running_mean, float* running_var,
                                float* save_mean, float* save_rstd, int act, float slope) {
  extern __shared__ __attribute__((aligned(16))) float4 coef[];  // [2][c4]: scale, shift
  const double inv = 1.0 / (double)pixels;
  // one CHANNEL per thread (not one quad: four times the threads share the replica sums, 32 independent loads each)
  {
    float* cf = reinterpret_cast<float*>(coef);
    const int C = c4 * 4;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int r = 0; r < BN_REPLICAS; ++r) {
        s1 += sums[(size_t)r * 2 * C + c];
        s2 += sums[(size_t)r * 2 * C + C + c];
      }
      const double m = s1 * inv;
      double var = s2 * inv - m * m;
      var = var > 0.0 ? var : 0.0;
      const float rstd = (float)(1.0 / sqrt(var + (double)eps));
      const float sc = gamma[c] * rstd;
      cf[c] = sc;
      cf[C + c] = beta[c] - (float)m * sc;
      if (blockIdx.x == 0) {  // one owner per channel
        if (save_mean) save_mean[c] = (float)m;
        if (save_rstd) save_rstd[c] = rstd;
        if (running_mean) {
          const float v = (float)var;
          const float unb = pixels > 1 ? v * ((float)pixels / (float)(pixels - 1)) : v;
          running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
          running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
        }
      }
    }
  }
  __syncthreads();
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(g % c4);
  const float4 sc4 = coef[q], sh4 = coef[c4 + q];
  const f32x4 scale = {sc4.x, sc4.y, sc4.z, sc4.w}, shift = {sh4.x, sh4.y, sh4.z, sh4.w};
  // four independent 16-byte streams per thread in flight (a single dependent load per iteration left HBM at ~3.9 TB/s)
  int64_t i = g;
  for (; i + 3 * T < n4; i += 4 * T) {
    f32x4 yy[4], rr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) yy[u] = y[i + u * T];
    if (residual) {
#pragma unroll
      for (int u = 0; u < 4; ++u) rr[u] = residual[i + u * T];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(yy[u][e], scale[e], shift[e]);   // the backward re-evaluates exactly this
      if (residual) v += rr[u];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], act, slope);
      z[i + u * T] = v;
    }
  }
  for (; i < n4; i += T) {
    const f32x4 yy = y[i];
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(yy[e], scale[e], shift[e]);
    if (residual) v += residual[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], act, slope);
    z[i] = v;
  }
}

__global__ void bn_apply_eval_kernel(const f32x4* __restrict__ y, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, const float* __restrict__ rm,
                                     const float* __restrict__ rv, const f32x4* __restrict__ residual,
                                     f32x4* __restrict__ z, int64_t n4, int c4, float eps, int act, float slope) {
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(g % c4);
  f32x4 scale, shift;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = q * 4 + e;
    const float rstd = 1.0f / sqrtf(rv[c] + eps);
    scale[e] = gamma[c] * rstd;
    shift[e] = beta[c] - rm[c] * scale[e];
  }
  for (int64_t i = g; i < n4; i += T) {
    f32x4 v = y[i] * scale + shift;
    if (residual) v += residual[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], act, slope);
    z[i] = v;
  }
}

// z == nullptr with an activation: the layer had no residual input, so the activation's argument is bn(y) itself and is
// re-evaluated from y (same fused multiply-add, same coefficients as bn_apply_kernel) instead of reading z: one fewer
// pass over the activation in each of the two backward kernels.
__global__ void bn_bwd_reduce_kernel(const f32x4* __restrict__ dz, const f32x4* __restrict__ z,
                                     const f32x4* __restrict__ y, const float* __restrict__ save_mean,
                                     const float* __restrict__ save_rstd, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, int64_t n4, int c4,
                                     double* __restrict__ bsums, int act, float slope) {
  __shared__ d4 red[256];
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(g % c4);
  f32x4 mean, rstd, sc = {0, 0, 0, 0}, sh = {0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mean[e] = save_mean[q * 4 + e];
    rstd[e] = save_rstd[q * 4 + e];
    if (!z && act != UDASEG_ACT_NONE) {
      sc[e] = gamma[q * 4 + e] * rstd[e];
      sh[e] = beta[q * 4 + e] - mean[e] * sc[e];
    }
  }
  d4 sg = {{0, 0, 0, 0}}, sgx = {{0, 0, 0, 0}};
  auto body = [&](f32x4 gz, const f32x4 yy, const f32x4 zin) {
    if (act != UDASEG_ACT_NONE) {
      f32x4 zz = zin;
      if (!z) {
#pragma unroll
        for (int e = 0; e < 4; ++e) zz[e] = __builtin_fmaf(yy[e], sc[e], sh[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) gz[e] *= act_grad(zz[e], act, slope);
    }
    const f32x4 xh = (yy - mean) * rstd;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sg.v[e] += (double)gz[e];
      sgx.v[e] += (double)gz[e] * (double)xh[e];
    }
  };
  // four independent iterations of loads in flight per thread (same summation order as the plain loop)
  const bool need_z = z != nullptr && act != UDASEG_ACT_NONE;
  int64_t i = g;
  for (; i + 3 * T < n4; i += 4 * T) {
    f32x4 a[4], b[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = dz[i + u * T];
      b[u] = y[i + u * T];
    }
    if (need_z) {
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = z[i + u * T];
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = f32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) body(a[u], b[u], c[u]);
  }
  for (; i < n4; i += T) body(dz[i], y[i], need_z ? z[i] : f32x4{0, 0, 0, 0});
  double* rep = bsums + (size_t)(blockIdx.x % BN_REPLICAS) * 2 * c4 * 4;
  block_fold_add(sg, rep, c4, q, red);
  block_fold_add(sgx, rep + (size_t)c4 * 4, c4, q, red);
}

__global__ void bn_bwd_apply_kernel(const f32x4* __restrict__ dz, const f32x4* __restrict__ z,
                                    const f32x4* __restrict__ y, const float* __restrict__ save_mean,
                                    const float* __restrict__ save_rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const double* __restrict__ bsums,
                                    f32x4* __restrict__ dy, f32x4* __restrict__ dres,
                                    float* dgamma, float* dbeta, int64_t n4, int c4, int64_t pixels, int act,
                                    float slope, int acc_dy, int acc_dres, int acc_param) {
  extern __shared__ __attribute__((aligned(16))) float4 coef[];  // [6][c4]: mean, rstd, scale, mean(g), mean(g*xhat), shift
  const double inv = 1.0 / (double)pixels;
  const bool recompute = !z && act != UDASEG_ACT_NONE;
  {
    float* cf = reinterpret_cast<float*>(coef);
    const int C = c4 * 4;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {     // one channel per thread, see bn_apply_kernel
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int r = 0; r < BN_REPLICAS; ++r) {
        s1 += bsums[(size_t)r * 2 * C + c];
        s2 += bsums[(size_t)r * 2 * C + C + c];
      }
      const float rs = save_rstd[c], mu = save_mean[c], gs = gamma[c] * rs;
      cf[c] = mu;
      cf[C + c] = rs;
      cf[2 * C + c] = gs;
      cf[3 * C + c] = (float)(s1 * inv);
      cf[4 * C + c] = (float)(s2 * inv);
      cf[5 * C + c] = recompute ? beta[c] - mu * gs : 0.f;
      if (blockIdx.x == 0) {
        const float db = (float)s1, dg = (float)s2;
        if (dbeta) dbeta[c] = acc_param ? dbeta[c] + db : db;
        if (dgamma) dgamma[c] = acc_param ? dgamma[c] + dg : dg;
      }
    }
  }
  __syncthreads();
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(g % c4);
  f32x4 mean, rstd, scale, mg, mgx, shift;
  {
    const float4 a0 = coef[q], a1 = coef[c4 + q], a2 = coef[2 * c4 + q], a3 = coef[3 * c4 + q], a4 = coef[4 * c4 + q];
    const float4 a5 = coef[5 * c4 + q];
    shift = f32x4{a5.x, a5.y, a5.z, a5.w};
    mean = f32x4{a0.x, a0.y, a0.z, a0.w};
    rstd = f32x4{a1.x, a1.y, a1.z, a1.w};
    scale = f32x4{a2.x, a2.y, a2.z, a2.w};
    mg = f32x4{a3.x, a3.y, a3.z, a3.w};
    mgx = f32x4{a4.x, a4.y, a4.z, a4.w};
  }
  auto body = [&](int64_t i, f32x4 gz, const f32x4 yy, const f32x4 zin, const f32x4 old_dy, const f32x4 old_res) {
    if (act != UDASEG_ACT_NONE) {
      f32x4 zz = zin;
      if (!z) {
#pragma unroll
        for (int e = 0; e < 4; ++e) zz[e] = __builtin_fmaf(yy[e], scale[e], shift[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) gz[e] *= act_grad(zz[e], act, slope);
    }
    const f32x4 xh = (yy - mean) * rstd;
    f32x4 out = scale * (gz - mg - xh * mgx);
    if (acc_dy) out += old_dy;
    dy[i] = out;
    if (dres) {
      f32x4 r = gz;
      if (acc_dres) r += old_res;
      dres[i] = r;
    }
  };
  // (a four-deep unroll like bn_apply's was measured here and LOST: 82.7 -> 123 us on the 134 MB layers -- five arrays of
  // four vectors cost the occupancy the streams need; the plain loop already has two or three loads in flight per thread)
  const f32x4 zero = {0, 0, 0, 0};
  const bool need_z = z != nullptr && act != UDASEG_ACT_NONE, need_res = dres != nullptr && acc_dres;
  for (int64_t i = g; i < n4; i += T) body(i, dz[i], y[i], need_z ? z[i] : zero, acc_dy ? dy[i] : zero, need_res ? dres[i] : zero);
}

__global__ void act_bwd_kernel(const f32x4* __restrict__ dz, const f32x4* __restrict__ z, f32x4* __restrict__ dy,
                               int64_t n4, int act, float slope) {
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += T) {
    f32x4 gz = dz[i];
    const f32x4 zz = z[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) gz[e] *= act_grad(zz[e], act, slope);
    dy[i] = gz;
  }
}

__global__ void channel_sum_kernel(const f32x4* __restrict__ x, int64_t n4, int c4, float* __restrict__ out, int replicas) {
  out += (size_t)(blockIdx.x % replicas) * c4 * 4;
  __shared__ float4 red[256];
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 s = {0, 0, 0, 0};
  for (int64_t i = g; i < n4; i += T) s += x[i];
  const float sv[4] = {s[0], s[1], s[2], s[3]};
  block_fold_atomic<4, float>(sv, out, c4, reinterpret_cast<float*>(red));
}

// Eval mode: fold BatchNorm into the preceding conv.  w'[co][j] = w[co][j] * gamma/sqrt(var+eps); b'[co] = beta +
// (bias - mean) * gamma/sqrt(var+eps).  Row-wise over the OHWI weight (j = taps*ci).
__global__ void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ rm, const float* __restrict__ rv,
                               float eps, int co, int J, float* __restrict__ wf, float* __restrict__ bf) {
  const int o = blockIdx.x;
  if (o >= co) return;
  const float s = gamma[o] / sqrtf(rv[o] + eps);
  for (int j = threadIdx.x; j < J; j += blockDim.x) wf[(size_t)o * J + j] = w[(size_t)o * J + j] * s;
  if (threadIdx.x == 0) bf[o] = beta[o] + ((bias ? bias[o] : 0.f) - rm[o]) * s;
}

static int check_pc(int64_t pixels, int c, const char* who) {
  UDASEG_CHECK_ARG(pixels > 0 && c > 0 && c % 4 == 0, "%s: need pixels > 0 and channels a positive multiple of 4 (got %lld, %d)",
                   who, (long long)pixels, c);
  UDASEG_CHECK_ARG(c <= 4096, "%s: channels > 4096 unsupported", who);
  return UDASEG_OK;
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_bn_replicas(void) { return BN_REPLICAS; }

extern "C" int udaseg_bn_stats(const float* y, int64_t pixels, int c, double* sums, void* stream) {
  int rc = check_pc(pixels, c, "bn_stats");
  if (rc) return rc;
  UDASEG_CHECK_ARG(y && sums, "bn_stats: NULL pointer");
  const int64_t n4 = pixels * (c / 4);
  const StreamShape s = stream_shape(n4, c / 4, reduce_max_blocks());
  hipLaunchKernelGGL(bn_stats_kernel, dim3(s.grid), dim3(s.bs), 0, as_stream(stream), (const f32x4*)y, n4, s.c4, sums);
  UDASEG_LAUNCH_CHECK("bn_stats launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bn_apply(const float* y, const double* sums, const float* gamma, const float* beta,
                               const float* residual, float* z, int64_t pixels, int c, float eps, float momentum,
                               float* running_mean, float* running_var, float* save_mean, float* save_rstd, int act,
                               float slope, void* stream) {
  int rc = check_pc(pixels, c, "bn_apply");
  if (rc) return rc;
  UDASEG_CHECK_ARG(y && sums && gamma && beta && z, "bn_apply: NULL pointer");
  UDASEG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_apply: running_mean/var must come together");
  const int64_t n4 = pixels * (c / 4);
  const StreamShape s = stream_shape(n4, c / 4, 2048, apply_per_thread());
  static std::atomic<int> kid_ba{-1};
  KTimer kt_ba(&kid_ba, "bn_apply_kernel", as_stream(stream), (double)pixels * c * 4.0 * (residual ? 3.0 : 2.0));
  hipLaunchKernelGGL(bn_apply_kernel, dim3(s.grid), dim3(s.bs), (size_t)2 * s.c4 * sizeof(float4), as_stream(stream), (const f32x4*)y, sums, gamma, beta,
                     (const f32x4*)residual, (f32x4*)z, n4, s.c4, pixels, eps, momentum, running_mean, running_var,
                     save_mean, save_rstd, act, slope);
  UDASEG_LAUNCH_CHECK("bn_apply launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bn_apply_eval(const float* y, const float* gamma, const float* beta, const float* running_mean,
                                    const float* running_var, const float* residual, float* z, int64_t pixels, int c,
                                    float eps, int act, float slope, void* stream) {
  int rc = check_pc(pixels, c, "bn_apply_eval");
  if (rc) return rc;
  UDASEG_CHECK_ARG(y && gamma && beta && running_mean && running_var && z, "bn_apply_eval: NULL pointer");
  const int64_t n4 = pixels * (c / 4);
  const StreamShape s = stream_shape(n4, c / 4);
  hipLaunchKernelGGL(bn_apply_eval_kernel, dim3(s.grid), dim3(s.bs), 0, as_stream(stream), (const f32x4*)y, gamma, beta,
                     running_mean, running_var, (const f32x4*)residual, (f32x4*)z, n4, s.c4, eps, act, slope);
  UDASEG_LAUNCH_CHECK("bn_apply_eval launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bn_bwd_reduce(const float* dz, const float* z, const float* y, const float* save_mean,
                                    const float* save_rstd, const float* gamma, const float* beta, int64_t pixels, int c,
                                    double* bsums, int act, float slope, void* stream) {
  int rc = check_pc(pixels, c, "bn_bwd_reduce");
  if (rc) return rc;
  UDASEG_CHECK_ARG(dz && y && save_mean && save_rstd && bsums, "bn_bwd_reduce: NULL pointer");
  UDASEG_CHECK_ARG(act == UDASEG_ACT_NONE || z || (gamma && beta),
                   "bn_bwd_reduce: an activation follows the norm: pass z, or gamma and beta to re-evaluate its argument");
  const int64_t n4 = pixels * (c / 4);
  const StreamShape s = stream_shape(n4, c / 4, reduce_max_blocks());
  static std::atomic<int> kid_br{-1};
  KTimer kt_br(&kid_br, "bn_bwd_reduce_kernel", as_stream(stream), (double)pixels * c * 4.0 * (z ? 3.0 : 2.0));
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(s.grid), dim3(s.bs), 0, as_stream(stream), (const f32x4*)dz, (const f32x4*)z,
                     (const f32x4*)y, save_mean, save_rstd, gamma, beta, n4, s.c4, bsums, act, slope);
  UDASEG_LAUNCH_CHECK("bn_bwd_reduce launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bn_bwd_apply(const float* dz, const float* z, const float* y, const float* save_mean,
                                   const float* save_rstd, const float* gamma, const float* beta, const double* bsums,
                                   float* dy, float* dres, float* dgamma, float* dbeta, int64_t pixels, int c, int act,
                                   float slope, int accumulate_dy, int accumulate_dres, int accumulate_param,
                                   void* stream) {
  int rc = check_pc(pixels, c, "bn_bwd_apply");
  if (rc) return rc;
  UDASEG_CHECK_ARG(dz && y && save_mean && save_rstd && gamma && bsums && dy, "bn_bwd_apply: NULL pointer");
  UDASEG_CHECK_ARG(act == UDASEG_ACT_NONE || z || beta,
                   "bn_bwd_apply: an activation follows the norm: pass z, or beta to re-evaluate its argument");
  const int64_t n4 = pixels * (c / 4);
  const StreamShape s = stream_shape(n4, c / 4, 2048, apply_per_thread());
  static std::atomic<int> kid_bw{-1};
  KTimer kt_bw(&kid_bw, "bn_bwd_apply_kernel", as_stream(stream), (double)pixels * c * 4.0 * ((z ? 4.0 : 3.0) + (dres ? 1.0 : 0.0)));
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(s.grid), dim3(s.bs), (size_t)6 * s.c4 * sizeof(float4), as_stream(stream), (const f32x4*)dz, (const f32x4*)z,
                     (const f32x4*)y, save_mean, save_rstd, gamma, beta, bsums, (f32x4*)dy, (f32x4*)dres, dgamma, dbeta, n4, s.c4,
                     pixels, act, slope, accumulate_dy, accumulate_dres, accumulate_param);
  UDASEG_LAUNCH_CHECK("bn_bwd_apply launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bn_fold(const float* w, const float* bias, const float* gamma, const float* beta,
                              const float* running_mean, const float* running_var, float eps, int co, int row_len, float* w_folded,
                              float* bias_folded, void* stream) {
  UDASEG_CHECK_ARG(w && gamma && beta && running_mean && running_var && w_folded && bias_folded && co > 0 && row_len > 0,
                   "bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(co), dim3(256), 0, as_stream(stream), w, bias, gamma, beta, running_mean, running_var,
                     eps, co, row_len, w_folded, bias_folded);
  UDASEG_LAUNCH_CHECK("bn_fold launch");
  return UDASEG_OK;
}

extern "C" int udaseg_act_bwd(const float* dz, const float* z, float* dy, int64_t count, int act, float slope,
                              void* stream) {
  UDASEG_CHECK_ARG(dz && z && dy && count > 0 && count % 4 == 0, "act_bwd: bad arguments");
  const int64_t n4 = count / 4;
  int grid = (int)((n4 + 1023) / 1024 > 2048 ? 2048 : (n4 + 1023) / 1024);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), (const f32x4*)dz, (const f32x4*)z,
                     (f32x4*)dy, n4, act, slope);
  UDASEG_LAUNCH_CHECK("act_bwd launch");
  return UDASEG_OK;
}

static int channel_sum_impl(const float* x, int64_t pixels, int c, float* out, int accumulate, float* scratch, size_t scratch_bytes,
                            void* stream) {
  int rc = check_pc(pixels, c, "channel_sum");
  if (rc) return rc;
  UDASEG_CHECK_ARG(x && out, "channel_sum: NULL pointer");
  hipStream_t st = as_stream(stream);
  const int64_t n4 = pixels * (c / 4);
  StreamShape s = stream_shape(n4, c / 4, reduce_max_blocks());
  if (s.grid <= CHSUM_DIRECT_BLOCKS) {
    if (!accumulate) {
      hipError_t e = hipMemsetAsync(out, 0, (size_t)c * sizeof(float), st);
      if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(channel_sum)");
    }
    hipLaunchKernelGGL(channel_sum_kernel, dim3(s.grid), dim3(s.bs), 0, st, (const f32x4*)x, n4, s.c4, out, 1);
    UDASEG_LAUNCH_CHECK("channel_sum launch");
    return UDASEG_OK;
  }
  // replicas of the output in the caller's scratch (stream-ordered ownership is the caller's), or in a stream-ordered allocation
  const size_t rep_bytes = (size_t)CHSUM_REPLICAS * c * sizeof(float);
  float* rep = scratch;
  if (rep != nullptr) {
    UDASEG_CHECK_ARG(scratch_bytes >= rep_bytes, "channel_sum: scratch of %zu bytes, need %zu", scratch_bytes, rep_bytes);
  } else {
    hipError_t e = hipMallocAsync(reinterpret_cast<void**>(&rep), rep_bytes, st);
    if (e != hipSuccess) return hip_fail(e, "hipMallocAsync(channel_sum)");
  }
  hipError_t e = hipMemsetAsync(rep, 0, rep_bytes, st);
  hipError_t e1 = hipSuccess, e2 = hipSuccess;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(channel_sum_kernel, dim3(s.grid), dim3(s.bs), 0, st, (const f32x4*)x, n4, s.c4, rep, CHSUM_REPLICAS);
    e1 = hipGetLastError();
    hipLaunchKernelGGL(fold_replicas_kernel, dim3((c + 255) / 256), dim3(256), 0, st, rep, c, out, accumulate);
    e2 = hipGetLastError();
  }
  hipError_t e3 = scratch == nullptr ? hipFreeAsync(rep, st) : hipSuccess;
  if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(channel_sum)");
  if (e1 != hipSuccess) return hip_fail(e1, "channel_sum launch");
  if (e2 != hipSuccess) return hip_fail(e2, "channel_sum fold launch");
  if (e3 != hipSuccess) return hip_fail(e3, "hipFreeAsync(channel_sum)");
  return UDASEG_OK;
}

extern "C" size_t udaseg_channel_sum_scratch_bytes(int c) { return c > 0 ? (size_t)CHSUM_REPLICAS * c * sizeof(float) : 0; }

extern "C" int udaseg_channel_sum(const float* x, int64_t pixels, int c, float* out, int accumulate, void* stream) {
  return channel_sum_impl(x, pixels, c, out, accumulate, nullptr, 0, stream);
}

extern "C" int udaseg_channel_sum_ws(const float* x, int64_t pixels, int c, float* out, int accumulate, float* scratch,
                                size_t scratch_bytes, void* stream) {
  UDASEG_CHECK_ARG(scratch != nullptr, "channel_sum_ws: NULL scratch");
  return channel_sum_impl(x, pixels, c, out, accumulate, scratch, scratch_bytes, stream);
}
