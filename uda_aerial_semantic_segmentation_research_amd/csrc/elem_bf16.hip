// HBM-bound kernels of the bf16 storage path (BASELINE configs 3 / 5): the same operations as norm_act.hip /
// pool_resize.hip / losses.hip's discriminator tail, on bf16 NHWC tensors (channel count a multiple of 8, one 16-byte
// vector = 8 channels per thread per step), arithmetic and all statistics in fp32 / f64.
//
// Kept apart from the fp32 kernels on purpose: those are the measured headline path and stay byte-identical.
#include "common.h"

namespace udaseg {

typedef __bf16 bf16x8e __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void unpack8(const f32x4& raw, float* f) {
  const bf16x8e v = __builtin_bit_cast(bf16x8e, raw);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
}
__device__ __forceinline__ f32x4 pack8(const float* f) {
  bf16x8e v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (__bf16)f[e];
  return __builtin_bit_cast(f32x4, v);
}

struct d8 {
  double v[8];
};
// fold per-thread 8-channel f64 partials over the block's rows, then f64 atomics (same scheme as norm_act.hip)
__device__ __forceinline__ void block_fold_add8(const d8& v, double* dst, int c8, int q, d8* red) {
  (void)q;
  block_fold_atomic<8, double>(v.v, dst, c8, reinterpret_cast<double*>(red));
}

// ------------------------------------------------------------------------------------------------ batch norm, bf16
__global__ void bn_apply_bf16_kernel(const f32x4* __restrict__ y, const double* __restrict__ sums,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const f32x4* __restrict__ residual, f32x4* __restrict__ z, int64_t n8, int c8,
                                     int64_t pixels, float eps, float momentum, float* running_mean, float* running_var,
                                     float* save_mean, float* save_rstd, int act, float slope) {
  extern __shared__ __attribute__((aligned(16))) float coef[];  // [2][C]: scale, shift
  const int C = c8 * 8;
  const double inv = 1.0 / (double)pixels;
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
    coef[c] = sc;
    coef[C + c] = beta[c] - (float)m * sc;
    if (blockIdx.x == 0) {
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
  __syncthreads();
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(g % c8);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = coef[q * 8 + e];
    sh[e] = coef[C + q * 8 + e];
  }
  auto body = [&](int64_t i, const f32x4 yv, const f32x4 rv) {
    float v[8], r[8];
    unpack8(yv, v);
    if (residual) unpack8(rv, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = v[e] * sc[e] + sh[e];
      if (residual) t += r[e];
      v[e] = act_apply(t, act, slope);
    }
    z[i] = pack8(v);
  };
  const f32x4 zero = {0, 0, 0, 0};
  int64_t i = g;
  for (; i + 3 * T < n8; i += 4 * T) {     // four independent 16-byte loads per array in flight per thread
    f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = y[i + u * T];
      b[u] = zero;
    }
    if (residual) {
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = residual[i + u * T];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) body(i + u * T, a[u], b[u]);
  }
  for (; i < n8; i += T) body(i, y[i], residual ? residual[i] : zero);
}

__global__ void bn_bwd_reduce_bf16_kernel(const f32x4* __restrict__ dz, const f32x4* __restrict__ z,
                                          const f32x4* __restrict__ y, const float* __restrict__ save_mean,
                                          const float* __restrict__ save_rstd, int64_t n8, int c8,
                                          double* __restrict__ bsums, int act, float slope) {
  __shared__ d8 red[256];
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(g % c8);
  float mean[8], rstd[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mean[e] = save_mean[q * 8 + e];
    rstd[e] = save_rstd[q * 8 + e];
  }
  d8 sg = {{0, 0, 0, 0, 0, 0, 0, 0}}, sgx = {{0, 0, 0, 0, 0, 0, 0, 0}};
  auto body = [&](const f32x4 gv, const f32x4 yv, const f32x4 zv) {
    float gz[8], zz[8], yy[8];
    unpack8(gv, gz);
    unpack8(yv, yy);
    if (act != UDASEG_ACT_NONE) unpack8(zv, zz);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float gg = gz[e];
      if (act != UDASEG_ACT_NONE) gg *= act_grad(zz[e], act, slope);
      const float xh = (yy[e] - mean[e]) * rstd[e];
      sg.v[e] += (double)gg;
      sgx.v[e] += (double)gg * (double)xh;
    }
  };
  const f32x4 zero = {0, 0, 0, 0};
  const bool need_z = act != UDASEG_ACT_NONE;
  int64_t i = g;
  for (; i + 3 * T < n8; i += 4 * T) {     // four independent iterations of loads in flight per thread, same summation order
    f32x4 a[4], b[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = dz[i + u * T];
      b[u] = y[i + u * T];
      c[u] = zero;
    }
    if (need_z) {
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = z[i + u * T];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) body(a[u], b[u], c[u]);
  }
  for (; i < n8; i += T) body(dz[i], y[i], need_z ? z[i] : zero);
  double* rep = bsums + (size_t)(blockIdx.x % BN_REPLICAS) * 2 * c8 * 8;
  block_fold_add8(sg, rep, c8, q, red);
  block_fold_add8(sgx, rep + (size_t)c8 * 8, c8, q, red);
}

// Per-channel sum and sum of squares of a bf16 tensor into the f64 replicas (what a convolution's fused statistics epilogue adds):
// behind the library GEMMs of csrc/gemm_lt.hip, which have no such epilogue.
__global__ void bn_stats_bf16_kernel(const f32x4* __restrict__ y, int64_t n8, int c8, double* __restrict__ sums) {
  __shared__ d8 red[256];
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(g % c8);
  d8 s1 = {{0, 0, 0, 0, 0, 0, 0, 0}}, s2 = {{0, 0, 0, 0, 0, 0, 0, 0}};
  for (int64_t i = g; i < n8; i += T) {
    float v[8];
    unpack8(y[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1.v[e] += (double)v[e];
      s2.v[e] += (double)v[e] * (double)v[e];
    }
  }
  double* rep = sums + (size_t)(blockIdx.x % BN_REPLICAS) * 2 * c8 * 8;
  block_fold_add8(s1, rep, c8, q, red);
  block_fold_add8(s2, rep + (size_t)c8 * 8, c8, q, red);
}

__global__ void bn_bwd_apply_bf16_kernel(const f32x4* __restrict__ dz, const f32x4* __restrict__ z,
                                         const f32x4* __restrict__ y, const float* __restrict__ save_mean,
                                         const float* __restrict__ save_rstd, const float* __restrict__ gamma,
                                         const double* __restrict__ bsums, f32x4* __restrict__ dy, f32x4* __restrict__ dres,
                                         float* dgamma, float* dbeta, int64_t n8, int c8, int64_t pixels, int act,
                                         float slope, int acc_dy, int acc_dres, int acc_param,
                                         const float* __restrict__ fwd_scale, const float* __restrict__ fwd_shift) {
  // fwd_scale / fwd_shift != null: the activation was never written (its consumer applied BatchNorm + activation while staging,
  // udaseg_conv2d_fwd_frag_bf16's in_scale / in_shift): its argument is re-evaluated here from y with the SAME fused
  // multiply-add and the same two vectors, so the mask is bit-identical to the forward's; z is not read
  extern __shared__ __attribute__((aligned(16))) float coef[];  // [5 (+2)][C]: mean, rstd, scale, mean(g), mean(g*xhat) (, fwd scale, shift)
  const int C = c8 * 8;
  const double inv = 1.0 / (double)pixels;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < BN_REPLICAS; ++r) {
      s1 += bsums[(size_t)r * 2 * C + c];
      s2 += bsums[(size_t)r * 2 * C + C + c];
    }
    const float rs = save_rstd[c];
    coef[c] = save_mean[c];
    coef[C + c] = rs;
    coef[2 * C + c] = gamma[c] * rs;
    coef[3 * C + c] = (float)(s1 * inv);
    coef[4 * C + c] = (float)(s2 * inv);
    if (fwd_scale) {
      coef[5 * C + c] = fwd_scale[c];
      coef[6 * C + c] = fwd_shift[c];
    }
    if (blockIdx.x == 0) {
      const float db = (float)s1, dg = (float)s2;
      if (dbeta) dbeta[c] = acc_param ? dbeta[c] + db : db;
      if (dgamma) dgamma[c] = acc_param ? dgamma[c] + dg : dg;
    }
  }
  __syncthreads();
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(g % c8);
  float mean[8], rstd[8], scale[8], mg[8], mgx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = q * 8 + e;
    mean[e] = coef[c]; rstd[e] = coef[C + c]; scale[e] = coef[2 * C + c]; mg[e] = coef[3 * C + c]; mgx[e] = coef[4 * C + c];
  }
  const bool recompute = fwd_scale != nullptr;
  auto body = [&](int64_t i, const f32x4 gv, const f32x4 yv, const f32x4 zv, const f32x4 ov, const f32x4 rv) {
    float gz[8], zz[8], yy[8], out[8], old[8];
    unpack8(gv, gz);
    unpack8(yv, yy);
    if (act != UDASEG_ACT_NONE && !recompute) unpack8(zv, zz);
    if (acc_dy) unpack8(ov, old);
    if (recompute) {
#pragma unroll
      for (int e = 0; e < 8; ++e) zz[e] = __builtin_fmaf(yy[e], coef[5 * C + q * 8 + e], coef[6 * C + q * 8 + e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (act != UDASEG_ACT_NONE) gz[e] *= act_grad(zz[e], act, slope);
      const float xh = (yy[e] - mean[e]) * rstd[e];
      out[e] = scale[e] * (gz[e] - mg[e] - xh * mgx[e]);
      if (acc_dy) out[e] += old[e];
    }
    dy[i] = pack8(out);
    if (dres) {
      if (acc_dres) {
        unpack8(rv, old);
#pragma unroll
        for (int e = 0; e < 8; ++e) gz[e] += old[e];
      }
      dres[i] = pack8(gz);
    }
  };
  const f32x4 zero = {0, 0, 0, 0};       // plain loop: see bn_bwd_apply_kernel (the four-deep unroll lost here)
  const bool need_z = act != UDASEG_ACT_NONE && !recompute, need_res = dres != nullptr && acc_dres;
  for (int64_t i = g; i < n8; i += T) body(i, dz[i], y[i], need_z ? z[i] : zero, acc_dy ? dy[i] : zero, need_res ? dres[i] : zero);
}

// Training-mode BatchNorm statistics -> everything the layer's consumers need, WITHOUT touching the activation: mean / rstd
// (saved for the backward), the running statistics, and the per-channel scale = gamma * rstd, shift = beta - mean * scale that
// a consumer applies while it stages its input (udaseg_conv2d_fwd_frag_bf16 in_scale / in_shift).  Same arithmetic as the
// prologue of bn_apply_bf16_kernel, so a layer gives the same numbers whichever way it is applied.
__global__ void bn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   int C, int64_t pixels, float eps, float momentum, float* running_mean, float* running_var,
                                   float* save_mean, float* save_rstd, float* scale, float* shift) {
  const double inv = 1.0 / (double)pixels;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
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
    scale[c] = sc;
    shift[c] = beta[c] - (float)m * sc;
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

__global__ void act_bwd_bf16_kernel(const f32x4* __restrict__ dz, const f32x4* __restrict__ z, f32x4* __restrict__ dy,
                                    int64_t n8, int act, float slope) {
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += T) {
    float g[8], zz[8];
    unpack8(dz[i], g);
    unpack8(z[i], zz);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] *= act_grad(zz[e], act, slope);
    dy[i] = pack8(g);
  }
}

__global__ void channel_sum_bf16_kernel(const f32x4* __restrict__ x, int64_t n8, int c8, float* __restrict__ out, int replicas) {
  out += (size_t)(blockIdx.x % replicas) * c8 * 8;
  __shared__ float red[256][8];
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = g; i < n8; i += T) {
    float v[8];
    unpack8(x[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += v[e];
  }
  block_fold_atomic<8, float>(s, out, c8, &red[0][0]);
}

// --------------------------------------------------------------------------------------------- layout / pool / resize
__global__ void nchw_to_nhwc_bf16_kernel(const float* __restrict__ x, f32x4* __restrict__ y, int c, int64_t hw, int c8,
                                         int64_t total_pix) {
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total_pix; p += T) {
    const int64_t n = p / hw, s = p - n * hw;
    const float* src = x + n * c * hw + s;
    for (int k = 0; k < c8; ++k) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (k * 8 + e) < c ? src[(int64_t)(k * 8 + e) * hw] : 0.f;
      y[p * c8 + k] = pack8(v);
    }
  }
}

__global__ void cast_f32_bf16_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int64_t n8) {
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += T) {
    const f32x4 a = x[2 * i], b = x[2 * i + 1];
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    y[i] = pack8(v);
  }
}

__global__ void maxpool_fwd_bf16_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, uint2* __restrict__ idx, int n,
                                        int h, int w, int c8, int ho, int wo) {
  const int64_t total = (int64_t)n * ho * wo * c8;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += T) {
    const int q = (int)(i % c8);
    int64_t r = i / c8;
    const int ox = (int)(r % wo);
    r /= wo;
    const int oy = (int)(r % ho);
    const int ni = (int)(r / ho);
    float best[8];
    uint32_t bi[8];
    bool first = true;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)w) continue;
        float v[8];
        unpack8(x[((int64_t)(ni * h + iy) * w + ix) * c8 + q], v);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (first || v[e] > best[e] || v[e] != v[e]) {
            best[e] = v[e];
            bi[e] = (uint32_t)(ky * 3 + kx);
          }
        first = false;
      }
    }
    y[i] = pack8(best);
    idx[i] = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24), bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24));
  }
}

__global__ void maxpool_bwd_bf16_kernel(const f32x4* __restrict__ dy, const uint2* __restrict__ idx, f32x4* __restrict__ dx,
                                        int n, int h, int w, int c8, int ho, int wo, int accumulate) {
  const int64_t total = (int64_t)n * h * w * c8;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += T) {
    const int q = (int)(i % c8);
    int64_t r = i / c8;
    const int ix = (int)(r % w);
    r /= w;
    const int iy = (int)(r % h);
    const int ni = (int)(r / h);
    float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = iy + 1 - ky;
      if (ty < 0 || (ty & 1)) continue;
      const int oy = ty >> 1;
      if (oy >= ho) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = ix + 1 - kx;
        if (tx < 0 || (tx & 1)) continue;
        const int ox = tx >> 1;
        if (ox >= wo) continue;
        const int64_t o = ((int64_t)(ni * ho + oy) * wo + ox) * c8 + q;
        const uint2 id = idx[o];
        float d[8];
        unpack8(dy[o], d);
        const uint32_t me = (uint32_t)(ky * 3 + kx);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t word = e < 4 ? id.x : id.y;
          if (((word >> (8 * (e & 3))) & 0xffu) == me) g[e] += d[e];
        }
      }
    }
    if (accumulate) {
      float old[8];
      unpack8(dx[i], old);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] += old[e];
    }
    dx[i] = pack8(g);
  }
}

__global__ void upcat_bwd_a_bf16_kernel(const f32x4* __restrict__ dout, f32x4* __restrict__ da, int n, int h, int w, int ca8,
                                        int ct8, int accumulate) {
  const int W = 2 * w;
  const int64_t total = (int64_t)n * h * w * ca8;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += T) {
    const int q = (int)(i % ca8);
    int64_t r = i / ca8;
    const int x = (int)(r % w);
    r /= w;
    const int y = (int)(r % h);
    const int ni = (int)(r / h);
    const int64_t p00 = ((int64_t)(ni * 2 * h + 2 * y) * W + 2 * x);
    float g[8], t[8];
    unpack8(dout[p00 * ct8 + q], g);
    unpack8(dout[(p00 + 1) * ct8 + q], t);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] += t[e];
    unpack8(dout[(p00 + W) * ct8 + q], t);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] += t[e];
    unpack8(dout[(p00 + W + 1) * ct8 + q], t);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] += t[e];
    if (accumulate) {
      unpack8(da[i], t);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] += t[e];
    }
    da[i] = pack8(g);
  }
}

__global__ void upcat_bwd_skip_bf16_kernel(const f32x4* __restrict__ dout, f32x4* __restrict__ dskip, int64_t pixels, int ca8,
                                           int cb8, int accumulate) {
  const int ct8 = ca8 + cb8;
  const int64_t total = pixels * cb8;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += T) {
    const int q = (int)(i % cb8);
    const int64_t p = i / cb8;
    f32x4 raw = dout[p * ct8 + ca8 + q];
    if (accumulate) {
      float g[8], t[8];
      unpack8(raw, g);
      unpack8(dskip[i], t);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] += t[e];
      raw = pack8(g);
    }
    dskip[i] = raw;
  }
}

// ---------------------------------------------------------------------------------------- discriminator tail, bf16
__global__ void gap_partial_bf16_kernel(const f32x4* __restrict__ z, float* __restrict__ partial, int hw, int c8, int splits) {
  const int ni = blockIdx.y, s = blockIdx.x;
  const int per = (hw + splits - 1) / splits;
  const int p0 = s * per, p1 = min(hw, p0 + per);
  for (int q = threadIdx.x; q < c8; q += blockDim.x) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = p0; p < p1; ++p) {
      float v[8];
      unpack8(z[((int64_t)ni * hw + p) * c8 + q], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) partial[((int64_t)ni * splits + s) * c8 * 8 + q * 8 + e] = acc[e];
  }
}

__global__ void gap_bwd_broadcast_bf16_kernel(const float* __restrict__ dp, const float* __restrict__ p, const float* __restrict__ w,
                                              f32x4* __restrict__ dz, int n, int hw, int c8) {
  const int64_t total = (int64_t)n * hw * c8;
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  const float inv = 1.f / (float)hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += T) {
    const int q = (int)(i % c8);
    const int ni = (int)(i / ((int64_t)hw * c8));
    const float pv = p[ni];
    const float dl = dp[ni] * pv * (1.f - pv) * inv;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w[q * 8 + e] * dl;
    dz[i] = pack8(v);
  }
}

// weights: fp32 arena -> bf16 copy (same layout) happens through cast_f32_bf16; the dgrad packing [ci][taps][co] in bf16:
__global__ void pack_dgrad_batched_bf16_kernel(const float* __restrict__ arena, __bf16* __restrict__ packed,
                                               const int* __restrict__ table) {
  const int* e = table + 5 * blockIdx.y;
  const float* w = arena + e[0];
  __bf16* wt = packed + e[1];
  if ((e[2] | e[4]) & 3) {
    const int co = e[2], T = e[3], ci = e[4];
    const int total = co * T * ci;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
      const int o = i % co;
      const int r = i / co;
      const int t = r % T;
      const int c = r / T;
      wt[i] = (__bf16)w[(o * T + t) * ci + c];
    }
    return;
  }
  pack_dgrad_tiles<__bf16>(w, wt, e[2], e[3], e[4]);
}

static inline int grid_for8(int64_t items, int per_thread = 2) {
  int64_t g = (items + 256LL * per_thread - 1) / (256LL * per_thread);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

static int check_pc8(int64_t pixels, int c, const char* who) {
  UDASEG_CHECK_ARG(pixels > 0 && c > 0 && c % 8 == 0 && c <= 4096, "%s: need pixels > 0 and channels a positive multiple of 8, <= 4096 (got %lld, %d)",
                   who, (long long)pixels, c);
  return UDASEG_OK;
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_bn_apply_bf16(const void* y, const double* sums, const float* gamma, const float* beta,
                                    const void* residual, void* z, int64_t pixels, int c, float eps, float momentum,
                                    float* running_mean, float* running_var, float* save_mean, float* save_rstd, int act,
                                    float slope, void* stream) {
  int rc = check_pc8(pixels, c, "bn_apply_bf16");
  if (rc) return rc;
  UDASEG_CHECK_ARG(y && sums && gamma && beta && z, "bn_apply_bf16: NULL pointer");
  const int64_t n8 = pixels * (c / 8);
  const StreamShape s = stream_shape(n8, c / 8, 2048, apply_per_thread());
  static std::atomic<int> kid_bn_apply_bf16_kernel{-1};
  KTimer kt_bn_apply_bf16_kernel(&kid_bn_apply_bf16_kernel, "bn_apply_bf16_kernel", as_stream(stream), (double)pixels * c * 2.0 * (residual ? 3.0 : 2.0));
  hipLaunchKernelGGL(bn_apply_bf16_kernel, dim3(s.grid), dim3(s.bs), (size_t)2 * c * sizeof(float), as_stream(stream),
                     (const f32x4*)y, sums, gamma, beta, (const f32x4*)residual, (f32x4*)z, n8, s.c4, pixels, eps, momentum,
                     running_mean, running_var, save_mean, save_rstd, act, slope);
  UDASEG_LAUNCH_CHECK("bn_apply_bf16 launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bn_stats_bf16(const void* y, int64_t pixels, int c, double* sums, void* stream) {
  int rc = check_pc8(pixels, c, "bn_stats_bf16");
  if (rc) return rc;
  UDASEG_CHECK_ARG(y && sums, "bn_stats_bf16: NULL pointer");
  const int64_t n8 = pixels * (c / 8);
  const StreamShape s = stream_shape(n8, c / 8, reduce_max_blocks());
  static std::atomic<int> kid_bn_stats_bf16_kernel{-1};
  KTimer kt_bn_stats_bf16_kernel(&kid_bn_stats_bf16_kernel, "bn_stats_bf16_kernel", as_stream(stream), (double)pixels * c * 2.0);
  hipLaunchKernelGGL(bn_stats_bf16_kernel, dim3(s.grid), dim3(s.bs), 0, as_stream(stream), (const f32x4*)y, n8, s.c4, sums);
  UDASEG_LAUNCH_CHECK("bn_stats_bf16 launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bn_bwd_reduce_bf16(const void* dz, const void* z, const void* y, const float* save_mean,
                                         const float* save_rstd, int64_t pixels, int c, double* bsums, int act, float slope,
                                         void* stream) {
  int rc = check_pc8(pixels, c, "bn_bwd_reduce_bf16");
  if (rc) return rc;
  UDASEG_CHECK_ARG(dz && y && save_mean && save_rstd && bsums && (act == UDASEG_ACT_NONE || z), "bn_bwd_reduce_bf16: NULL pointer");
  const int64_t n8 = pixels * (c / 8);
  const StreamShape s = stream_shape(n8, c / 8, reduce_max_blocks());
  static std::atomic<int> kid_bn_bwd_reduce_bf16_kernel{-1};
  KTimer kt_bn_bwd_reduce_bf16_kernel(&kid_bn_bwd_reduce_bf16_kernel, "bn_bwd_reduce_bf16_kernel", as_stream(stream), (double)pixels * c * 2.0 * (act != UDASEG_ACT_NONE ? 3.0 : 2.0));
  hipLaunchKernelGGL(bn_bwd_reduce_bf16_kernel, dim3(s.grid), dim3(s.bs), 0, as_stream(stream), (const f32x4*)dz,
                     (const f32x4*)z, (const f32x4*)y, save_mean, save_rstd, n8, s.c4, bsums, act, slope);
  UDASEG_LAUNCH_CHECK("bn_bwd_reduce_bf16 launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bn_bwd_apply_bf16(const void* dz, const void* z, const void* y, const float* save_mean,
                                        const float* save_rstd, const float* gamma, const double* bsums, void* dy, void* dres,
                                        float* dgamma, float* dbeta, int64_t pixels, int c, int act, float slope,
                                        int accumulate_dy, int accumulate_dres, int accumulate_param, void* stream) {
  int rc = check_pc8(pixels, c, "bn_bwd_apply_bf16");
  if (rc) return rc;
  UDASEG_CHECK_ARG(dz && y && save_mean && save_rstd && gamma && bsums && dy && (act == UDASEG_ACT_NONE || z),
                   "bn_bwd_apply_bf16: NULL pointer");
  UDASEG_CHECK_ARG((size_t)5 * c * sizeof(float) <= 65536, "bn_bwd_apply_bf16: too many channels");
  const int64_t n8 = pixels * (c / 8);
  const StreamShape s = stream_shape(n8, c / 8, 2048, apply_per_thread());
  static std::atomic<int> kid_bwa{-1};
  KTimer kt_bwa(&kid_bwa, "bn_bwd_apply_bf16_kernel", as_stream(stream), (double)pixels * c * 2.0 * ((act != UDASEG_ACT_NONE ? 4.0 : 3.0) + (dres ? 1.0 : 0.0)));
  hipLaunchKernelGGL(bn_bwd_apply_bf16_kernel, dim3(s.grid), dim3(s.bs), (size_t)5 * c * sizeof(float), as_stream(stream),
                     (const f32x4*)dz, (const f32x4*)z, (const f32x4*)y, save_mean, save_rstd, gamma, bsums, (f32x4*)dy,
                     (f32x4*)dres, dgamma, dbeta, n8, s.c4, pixels, act, slope, accumulate_dy, accumulate_dres, accumulate_param,
                     (const float*)nullptr, (const float*)nullptr);
  UDASEG_LAUNCH_CHECK("bn_bwd_apply_bf16 launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bn_bwd_apply_recompute_bf16(const void* dz, const void* y, const float* fwd_scale, const float* fwd_shift,
                                                  const float* save_mean, const float* save_rstd, const float* gamma,
                                                  const double* bsums, void* dy, float* dgamma, float* dbeta, int64_t pixels, int c,
                                                  int act, float slope, void* stream) {
  int rc = check_pc8(pixels, c, "bn_bwd_apply_recompute_bf16");
  if (rc) return rc;
  UDASEG_CHECK_ARG(dz && y && fwd_scale && fwd_shift && save_mean && save_rstd && gamma && bsums && dy,
                   "bn_bwd_apply_recompute_bf16: NULL pointer");
  UDASEG_CHECK_ARG((size_t)7 * c * sizeof(float) <= 65536, "bn_bwd_apply_recompute_bf16: too many channels");
  const int64_t n8 = pixels * (c / 8);
  const StreamShape s = stream_shape(n8, c / 8, 2048, apply_per_thread());
  static std::atomic<int> kid_bwr{-1};
  KTimer kt_bwr(&kid_bwr, "bn_bwd_apply_bf16_kernel", as_stream(stream), (double)pixels * c * 2.0 * 3.0);
  hipLaunchKernelGGL(bn_bwd_apply_bf16_kernel, dim3(s.grid), dim3(s.bs), (size_t)7 * c * sizeof(float), as_stream(stream),
                     (const f32x4*)dz, (const f32x4*)nullptr, (const f32x4*)y, save_mean, save_rstd, gamma, bsums, (f32x4*)dy,
                     (f32x4*)nullptr, dgamma, dbeta, n8, s.c4, pixels, act, slope, 0, 0, 0, fwd_scale, fwd_shift);
  UDASEG_LAUNCH_CHECK("bn_bwd_apply_recompute_bf16 launch");
  return UDASEG_OK;
}

extern "C" int udaseg_bn_finalize(const double* sums, const float* gamma, const float* beta, int64_t pixels, int c, float eps,
                                  float momentum, float* running_mean, float* running_var, float* save_mean, float* save_rstd,
                                  float* scale, float* shift, void* stream) {
  UDASEG_CHECK_ARG(sums && gamma && beta && scale && shift && pixels > 0 && c > 0, "bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(c, 256)), dim3(256), 0, as_stream(stream), sums, gamma, beta, c, pixels, eps,
                     momentum, running_mean, running_var, save_mean, save_rstd, scale, shift);
  UDASEG_LAUNCH_CHECK("bn_finalize launch");
  return UDASEG_OK;
}

extern "C" int udaseg_act_bwd_bf16(const void* dz, const void* z, void* dy, int64_t count, int act, float slope, void* stream) {
  UDASEG_CHECK_ARG(dz && z && dy && count > 0 && count % 8 == 0, "act_bwd_bf16: bad arguments");
  const int64_t n8 = count / 8;
  hipLaunchKernelGGL(act_bwd_bf16_kernel, dim3(grid_for8(n8)), dim3(256), 0, as_stream(stream), (const f32x4*)dz, (const f32x4*)z,
                     (f32x4*)dy, n8, act, slope);
  UDASEG_LAUNCH_CHECK("act_bwd_bf16 launch");
  return UDASEG_OK;
}

static int channel_sum_bf16_impl(const void* x, int64_t pixels, int c, float* out, int accumulate, float* scratch, size_t scratch_bytes,
                            void* stream) {
  int rc = check_pc8(pixels, c, "channel_sum_bf16");
  if (rc) return rc;
  UDASEG_CHECK_ARG(x && out, "channel_sum_bf16: NULL pointer");
  hipStream_t st = as_stream(stream);
  const int64_t n8 = pixels * (c / 8);
  StreamShape s = stream_shape(n8, c / 8, reduce_max_blocks());
  if (s.grid <= CHSUM_DIRECT_BLOCKS) {
    if (!accumulate) {
      hipError_t e = hipMemsetAsync(out, 0, (size_t)c * sizeof(float), st);
      if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(channel_sum_bf16)");
    }
    hipLaunchKernelGGL(channel_sum_bf16_kernel, dim3(s.grid), dim3(s.bs), 0, st, (const f32x4*)x, n8, s.c4, out, 1);
    UDASEG_LAUNCH_CHECK("channel_sum_bf16 launch");
    return UDASEG_OK;
  }
  // replicas of the output in the caller's scratch (stream-ordered ownership is the caller's), or in a stream-ordered allocation
  const size_t rep_bytes = (size_t)CHSUM_REPLICAS * c * sizeof(float);
  float* rep = scratch;
  if (rep != nullptr) {
    UDASEG_CHECK_ARG(scratch_bytes >= rep_bytes, "channel_sum_bf16: scratch of %zu bytes, need %zu", scratch_bytes, rep_bytes);
  } else {
    hipError_t e = hipMallocAsync(reinterpret_cast<void**>(&rep), rep_bytes, st);
    if (e != hipSuccess) return hip_fail(e, "hipMallocAsync(channel_sum_bf16)");
  }
  hipError_t e = hipMemsetAsync(rep, 0, rep_bytes, st);
  hipError_t e1 = hipSuccess, e2 = hipSuccess;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(channel_sum_bf16_kernel, dim3(s.grid), dim3(s.bs), 0, st, (const f32x4*)x, n8, s.c4, rep, CHSUM_REPLICAS);
    e1 = hipGetLastError();
    hipLaunchKernelGGL(fold_replicas_kernel, dim3((c + 255) / 256), dim3(256), 0, st, rep, c, out, accumulate);
    e2 = hipGetLastError();
  }
  hipError_t e3 = scratch == nullptr ? hipFreeAsync(rep, st) : hipSuccess;
  if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(channel_sum_bf16)");
  if (e1 != hipSuccess) return hip_fail(e1, "channel_sum_bf16 launch");
  if (e2 != hipSuccess) return hip_fail(e2, "channel_sum_bf16 fold launch");
  if (e3 != hipSuccess) return hip_fail(e3, "hipFreeAsync(channel_sum_bf16)");
  return UDASEG_OK;
}

extern "C" int udaseg_channel_sum_bf16(const void* x, int64_t pixels, int c, float* out, int accumulate, void* stream) {
  return channel_sum_bf16_impl(x, pixels, c, out, accumulate, nullptr, 0, stream);
}

extern "C" int udaseg_channel_sum_bf16_ws(const void* x, int64_t pixels, int c, float* out, int accumulate, float* scratch,
                                size_t scratch_bytes, void* stream) {
  UDASEG_CHECK_ARG(scratch != nullptr, "channel_sum_bf16_ws: NULL scratch");
  return channel_sum_bf16_impl(x, pixels, c, out, accumulate, scratch, scratch_bytes, stream);
}

extern "C" int udaseg_nchw_to_nhwc_bf16(const float* x, void* y, int n, int c, int h, int w, int cpad, void* stream) {
  UDASEG_CHECK_ARG(x && y && n > 0 && c > 0 && h > 0 && w > 0 && cpad >= c && cpad % 8 == 0, "nchw_to_nhwc_bf16: bad arguments");
  const int64_t hw = (int64_t)h * w, total = hw * n;
  hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, dim3(grid_for8(total, 1)), dim3(256), 0, as_stream(stream), x, (f32x4*)y, c, hw,
                     cpad / 8, total);
  UDASEG_LAUNCH_CHECK("nchw_to_nhwc_bf16 launch");
  return UDASEG_OK;
}

extern "C" int udaseg_cast_f32_to_bf16(const float* x, void* y, int64_t count, void* stream) {
  UDASEG_CHECK_ARG(x && y && count > 0 && count % 8 == 0, "cast_f32_to_bf16: count must be a positive multiple of 8");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for8(count / 8, 4)), dim3(256), 0, as_stream(stream), (const f32x4*)x,
                     (f32x4*)y, count / 8);
  UDASEG_LAUNCH_CHECK("cast_f32_to_bf16 launch");
  return UDASEG_OK;
}

extern "C" int udaseg_maxpool3x3s2_fwd_bf16(const void* x, void* y, uint8_t* idx, int n, int h, int w, int c, void* stream) {
  UDASEG_CHECK_ARG(x && y && idx && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "maxpool_fwd_bf16: bad arguments");
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)n * ho * wo * (c / 8);
  hipLaunchKernelGGL(maxpool_fwd_bf16_kernel, dim3(grid_for8(total, 1)), dim3(256), 0, as_stream(stream), (const f32x4*)x,
                     (f32x4*)y, (uint2*)idx, n, h, w, c / 8, ho, wo);
  UDASEG_LAUNCH_CHECK("maxpool_fwd_bf16 launch");
  return UDASEG_OK;
}

extern "C" int udaseg_maxpool3x3s2_bwd_bf16(const void* dy, const uint8_t* idx, void* dx, int n, int h, int w, int c,
                                            int accumulate, void* stream) {
  UDASEG_CHECK_ARG(dy && idx && dx && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "maxpool_bwd_bf16: bad arguments");
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)n * h * w * (c / 8);
  hipLaunchKernelGGL(maxpool_bwd_bf16_kernel, dim3(grid_for8(total, 1)), dim3(256), 0, as_stream(stream), (const f32x4*)dy,
                     (const uint2*)idx, (f32x4*)dx, n, h, w, c / 8, ho, wo, accumulate);
  UDASEG_LAUNCH_CHECK("maxpool_bwd_bf16 launch");
  return UDASEG_OK;
}

extern "C" int udaseg_upsample2x_concat_bwd_bf16(const void* dout, void* da, void* dskip, int n, int h, int w, int ca, int cb,
                                                 int accumulate_da, int accumulate_dskip, void* stream) {
  UDASEG_CHECK_ARG(dout && n > 0 && h > 0 && w > 0 && ca > 0 && ca % 8 == 0 && cb >= 0 && cb % 8 == 0,
                   "upsample2x_concat_bwd_bf16: bad arguments");
  hipStream_t st = as_stream(stream);
  if (da) {
    const int64_t total = (int64_t)n * h * w * (ca / 8);
    hipLaunchKernelGGL(upcat_bwd_a_bf16_kernel, dim3(grid_for8(total, 1)), dim3(256), 0, st, (const f32x4*)dout, (f32x4*)da, n, h,
                       w, ca / 8, (ca + cb) / 8, accumulate_da);
    UDASEG_LAUNCH_CHECK("upsample2x_concat_bwd_bf16(a) launch");
  }
  if (dskip && cb > 0) {
    const int64_t pixels = (int64_t)n * 4 * h * w;
    hipLaunchKernelGGL(upcat_bwd_skip_bf16_kernel, dim3(grid_for8(pixels * (cb / 8))), dim3(256), 0, st, (const f32x4*)dout,
                       (f32x4*)dskip, pixels, ca / 8, cb / 8, accumulate_dskip);
    UDASEG_LAUNCH_CHECK("upsample2x_concat_bwd_bf16(skip) launch");
  }
  return UDASEG_OK;
}

extern "C" int udaseg_gap_partial_bf16(const void* z, float* partial, int n, int hw, int c, void* stream) {
  UDASEG_CHECK_ARG(z && partial && n > 0 && hw > 0 && c > 0 && c % 8 == 0, "gap_partial_bf16: bad arguments");
  const int splits = udaseg_gap_splits(hw);
  const int bs = (c / 8) < 256 ? (((c / 8) + 63) / 64) * 64 : 256;
  hipLaunchKernelGGL(gap_partial_bf16_kernel, dim3(splits, n), dim3(bs), 0, as_stream(stream), (const f32x4*)z, partial, hw,
                     c / 8, splits);
  UDASEG_LAUNCH_CHECK("gap_partial_bf16 launch");
  return UDASEG_OK;
}

extern "C" int udaseg_gap_bwd_broadcast_bf16(const float* dp, const float* p, const float* w, void* dz, int n, int hw, int c,
                                             void* stream) {
  UDASEG_CHECK_ARG(dp && p && w && dz && n > 0 && hw > 0 && c > 0 && c % 8 == 0, "gap_bwd_broadcast_bf16: bad arguments");
  const int64_t total = (int64_t)n * hw * (c / 8);
  hipLaunchKernelGGL(gap_bwd_broadcast_bf16_kernel, dim3(grid_for8(total)), dim3(256), 0, as_stream(stream), dp, p, w,
                     (f32x4*)dz, n, hw, c / 8);
  UDASEG_LAUNCH_CHECK("gap_bwd_broadcast_bf16 launch");
  return UDASEG_OK;
}

extern "C" int udaseg_pack_dgrad_batched_bf16(const float* arena, void* packed, const int* table, int entries, void* stream) {
  UDASEG_CHECK_ARG(arena && packed && table && entries > 0, "pack_dgrad_batched_bf16: bad arguments");
  hipLaunchKernelGGL(pack_dgrad_batched_bf16_kernel, dim3(PACK_DGRAD_GRID_X, entries), dim3(256), 0, as_stream(stream), arena, (__bf16*)packed,
                     table);
  UDASEG_LAUNCH_CHECK("pack_dgrad_batched_bf16 launch");
  return UDASEG_OK;
}
