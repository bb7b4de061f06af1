// Plain GEMMs behind 1x1 / stride-1 convolutions, bf16 storage, through hipBLASLt (gfx950).
//
// A 1x1 convolution over NHWC tensors IS a GEMM over M = n*h*w contiguous pixel rows: forward  y[M][co] = x[M][ci] . w[co][ci]^T,
// data gradient dx[M][ci] (+)= dy[M][co] . w[co][ci], weight gradient dW[co][ci] += dy[M][co]^T . x[M][ci] -- what
// torch.nn.functional.conv2d / its autograd do for the bottleneck projections of smp.Unet("resnet50") (reference
// src/models/train.py:341,343; BASELINE cfg 5).  For r50's deeper stages (96^2 and below: M <= 73728, K and N 128..2048) these are
// small square-ish GEMMs on which the hand-written streaming kernel (conv1x1_stream_bf16_kernel: built for the HBM-bound 192^2
// layers) and the split-K weight gradient are latency-bound: 41-61 us per launch against 19-27 us for the vendor library on the
// same shapes (profiles/r03_gemm_1x1.txt).  These are plain library GEMMs: they go to hipBLASLt (workspace: caller-owned); the fused epilogues of the
// hand-written kernels (BatchNorm statistics, BatchNorm-backward sums) become the stand-alone passes on those layers, which is
// still a net gain below 96^2 (the tensors are small).  One handle, descriptors and the heuristic's algorithm cached per shape.
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

#include "common.h"

namespace udaseg {

struct LtPlan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  bool ok = false;
};

static hipblasLtHandle_t g_lt = nullptr;
// caller-owned workspaces (udaseg_gemm_set_workspace), one pair per device: forward / data gradient run on the compute stream,
// weight gradients on the side stream and may overlap them.  The library allocates nothing; unbound = workspace-free algorithms.
static void* g_lt_ws[16][2] = {};
static size_t g_lt_ws_bytes[16] = {};
static std::map<std::tuple<int, int64_t, int, int, int>, LtPlan> g_plans;      // (mode, M, ci, co, device)
static std::mutex g_lt_mu;
constexpr size_t LT_WS_MAX = 32u << 20;

static int lt_fail(hipblasStatus_t st, const char* what) {
  set_error("%s: hipBLASLt status %d", what, (int)st);
  return UDASEG_E_HIP;
}

// mode 0 forward, 1 data gradient, 2 weight gradient.  Column-major formulations (hipBLASLt's convention):
//   0: y^T (co x M)   = w^T-as-stored (ci x co)^T . x^T-as-stored (ci x M)
//   1: dx^T (ci x M)  = w-as-stored (ci x co)      . dy^T-as-stored (co x M)
//   2: dW^T (ci x co) = x^T-as-stored (ci x M)     . dy^T-as-stored (co x M)^T
static int lt_plan(int mode, int64_t M, int ci, int co, int dev, LtPlan** out) {
  const auto key = std::make_tuple(mode, M, ci, co, dev);
  auto it = g_plans.find(key);
  if (it != g_plans.end()) {
    *out = &it->second;
    return it->second.ok ? UDASEG_OK : UDASEG_E_UNSUPPORTED;
  }
  LtPlan p;
  hipblasStatus_t st;
  if (g_lt == nullptr && (st = hipblasLtCreate(&g_lt)) != HIPBLAS_STATUS_SUCCESS) return lt_fail(st, "hipblasLtCreate");
  if ((st = hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F)) != HIPBLAS_STATUS_SUCCESS) return lt_fail(st, "hipblasLtMatmulDescCreate");
  const int32_t ta = mode == 0 ? HIPBLAS_OP_T : HIPBLAS_OP_N, tb = mode == 2 ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta));
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb));
  // stored shapes (rows x cols, leading dimension = rows: every operand is a dense row-major [outer][inner] tensor)
  if (mode == 0) {
    st = hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16BF, ci, co, ci);            // w [co][ci]
    if (st == HIPBLAS_STATUS_SUCCESS) st = hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_16BF, ci, M, ci);     // x [M][ci]
    if (st == HIPBLAS_STATUS_SUCCESS) st = hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_16BF, co, M, co);     // y [M][co]
  } else if (mode == 1) {
    st = hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16BF, ci, co, ci);            // w [co][ci]
    if (st == HIPBLAS_STATUS_SUCCESS) st = hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_16BF, co, M, co);     // dy [M][co]
    if (st == HIPBLAS_STATUS_SUCCESS) st = hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_16BF, ci, M, ci);     // dx [M][ci]
  } else {
    st = hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16BF, ci, M, ci);             // x [M][ci]
    if (st == HIPBLAS_STATUS_SUCCESS) st = hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_16BF, co, M, co);     // dy [M][co]
    if (st == HIPBLAS_STATUS_SUCCESS) st = hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_32F, ci, co, ci);     // dW [co][ci] fp32
  }
  if (st != HIPBLAS_STATUS_SUCCESS) return lt_fail(st, "hipblasLtMatrixLayoutCreate");
  hipblasLtMatmulPreference_t pref;
  if ((st = hipblasLtMatmulPreferenceCreate(&pref)) != HIPBLAS_STATUS_SUCCESS) return lt_fail(st, "hipblasLtMatmulPreferenceCreate");
  uint64_t ws_max = (dev >= 0 && dev < 16) ? g_lt_ws_bytes[dev] : 0;
  if (ws_max > LT_WS_MAX) ws_max = LT_WS_MAX;
  hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_max, sizeof(ws_max));
  hipblasLtMatmulHeuristicResult_t res[1];
  int found = 0;
  st = hipblasLtMatmulAlgoGetHeuristic(g_lt, p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, res, &found);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st == HIPBLAS_STATUS_SUCCESS && found > 0) {
    p.algo = res[0].algo;
    p.ws = res[0].workspaceSize;
    p.ok = true;
  }
  auto ins = g_plans.emplace(key, p);
  *out = &ins.first->second;
  if (!p.ok) {
    set_error("gemm_1x1_bf16: hipBLASLt has no algorithm for mode %d M=%lld ci=%d co=%d (status %d)", mode, (long long)M, ci, co, (int)st);
    return UDASEG_E_UNSUPPORTED;
  }
  return UDASEG_OK;
}

}  // namespace udaseg

using namespace udaseg;

// Is the library GEMM the faster path for this 1x1 / stride-1 convolution?  Measured on MI355X (r50 at 768^2, profiles/
// r03_gemm_1x1.txt): yes from 96^2 down; at 192^2 the hand-written streaming kernel with its fused statistics wins.
extern "C" int udaseg_gemm_1x1_preferred(const udaseg_conv_desc* d) {
  static int on = -1;          // UDASEG_GEMM_1X1=0: never (A/B)
  if (on < 0) {
    const char* e = getenv("UDASEG_GEMM_1X1");
    on = (e && atoi(e) == 0) ? 0 : 1;
  }
  if (!on || !d) return 0;
  if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0) return 0;
  const long long M = (long long)d->n * d->hi * d->wi;
  if (d->ci % 8 != 0 || d->co % 8 != 0 || d->ci < 64 || d->co < 64) return 0;
  return M <= 73728 ? 1 : 0;
}

// Two caller-owned scratch buffers of `bytes` each for the CURRENT device (compute-stream GEMMs / side-stream weight gradients).
// Bind before the first GEMM of a shape is planned: the plan keeps the algorithm chosen under the workspace available then.
extern "C" int udaseg_gemm_set_workspace(void* ws_main, void* ws_side, size_t bytes) {
  int dev = 0;
  UDASEG_CHECK_ARG(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, "gemm_set_workspace: no current HIP device");
  UDASEG_CHECK_ARG((bytes == 0) == (ws_main == nullptr) && (bytes == 0) == (ws_side == nullptr) &&
                       (reinterpret_cast<uintptr_t>(ws_main) & 255) == 0 && (reinterpret_cast<uintptr_t>(ws_side) & 255) == 0,
                   "gemm_set_workspace: two 256-byte-aligned buffers of `bytes` each, or NULL / NULL / 0");
  std::lock_guard<std::mutex> lock(g_lt_mu);
  g_lt_ws[dev][0] = ws_main;
  g_lt_ws[dev][1] = ws_side;
  g_lt_ws_bytes[dev] = bytes;
  return UDASEG_OK;
}

extern "C" int udaseg_gemm_1x1_bf16(int mode, int64_t M, int ci, int co, const void* a, const void* b, void* c, float beta, void* stream) {
  UDASEG_CHECK_ARG(mode >= 0 && mode <= 2 && M > 0 && ci > 0 && co > 0 && ci % 8 == 0 && co % 8 == 0,
                   "gemm_1x1_bf16: mode %d M=%lld ci=%d co=%d (channels multiples of 8)", mode, (long long)M, ci, co);
  UDASEG_CHECK_ARG(a && b && c, "gemm_1x1_bf16: NULL pointer");
  UDASEG_CHECK_ARG(beta == 0.f || beta == 1.f, "gemm_1x1_bf16: beta is 0 or 1");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hip_fail(hipGetLastError(), "hipGetDevice");
  std::lock_guard<std::mutex> lock(g_lt_mu);
  LtPlan* p = nullptr;
  int rc = lt_plan(mode, M, ci, co, dev, &p);
  if (rc) return rc;
  void* ws = nullptr;
  if (p->ws > 0) {
    UDASEG_CHECK_ARG(dev >= 0 && dev < 16 && p->ws <= g_lt_ws_bytes[dev] && g_lt_ws[dev][mode == 2 ? 1 : 0] != nullptr,
                     "gemm_1x1_bf16: the planned algorithm wants %zu bytes of workspace; bind one with udaseg_gemm_set_workspace", p->ws);
    ws = g_lt_ws[dev][mode == 2 ? 1 : 0];
  }
  const float alpha = 1.f;
  // mode 0: A = w (b), B = x (a); mode 1: A = w (b), B = dy (a); mode 2: A = x (a), B = dy (b)
  const void* A = mode == 2 ? a : b;
  const void* B = mode == 2 ? b : a;
  hipStream_t st = as_stream(stream);
  static int kid[3] = {-1, -1, -1};
  static const char* const names[3] = {"conv_gemm_lt_fwd_bf16", "conv_gemm_lt_dgrad_bf16", "conv_gemm_lt_wgrad_bf16"};
  if (kid[mode] < 0) kid[mode] = kprof_id(names[mode]);
  hipEvent_t ev = kprof_begin(st);
  hipblasStatus_t hs = hipblasLtMatmul(g_lt, p->desc, &alpha, A, p->la, B, p->lb, &beta, c, p->lc, c, p->lc, &p->algo, ws, p->ws, st);
  kprof_end(kid[mode], ev, st, 2.0 * (double)M * ci * co);
  if (hs != HIPBLAS_STATUS_SUCCESS) return lt_fail(hs, "hipblasLtMatmul");
  return UDASEG_OK;
}
