// Library-level entry points: version, thread-local error text, device probe, live launch timing.
#include <string.h>
#include <mutex>
#include <vector>

#include "common.h"
#include "options.h"

namespace udaseg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return UDASEG_E_HIP;
}

// ---- live timing of the conv kernel families (bench.py roofline leg) ------------------------------------------
struct ProfRec {
  hipEvent_t a, b;
  double flops;
  int kind;  // 0 fwd, 1 dgrad, 2 wgrad
  udaseg_conv_desc d;
};
static bool g_prof_on = false;
struct KRec {
  hipEvent_t a, b;
  double flops;
};
// exactly the symbols rocprofv3 prints (minus "void udaseg::" and the parameter list), so HIP-event and rocprof averages
// can be compared per kernel
static const char* const g_knames_fixed[PROF_NKERNELS] = {
    "conv_igemm_kernel<128, 128, 2, 2, false, false, false>", "conv_igemm_kernel<128, 64, 2, 2, false, false, false>",
    "conv_igemm_kernel<64, 64, 2, 2, false, false, false>",   "conv_igemm_kernel<128, 32, 4, 1, false, false, false>",
    "conv3x3_small_kernel<1, 1>",                             "conv3x3_small_kernel<2, 1>",
    "conv3x3_small_kernel<1, 2>",                             "conv_wgrad_kernel<64, 64, 2, 2, false>",
    "conv_wgrad_kernel<32, 128, 1, 4, false>",                "conv3x3_small_wgrad_kernel<1, 1>",
    "conv3x3_small_wgrad_kernel<2, 1>",                       "conv3x3_small_wgrad_kernel<1, 2>",
    "conv_igemm_kernel<*, bf16>",                             "conv_wgrad_bf16_kernel",
    "conv_igemm_kernel<128, 128, 2, 2, false, true, false>",  "conv_igemm_kernel<128, 64, 2, 2, false, true, false>",
    "conv_igemm_kernel<64, 64, 2, 2, false, true, false>",    "conv_igemm_kernel<128, 32, 4, 1, false, true, false>",
    "conv_wgrad_kernel<64, 64, 2, 2, true>",                  "conv_wgrad_kernel<32, 128, 1, 4, true>",
    "conv_igemm_kernel<128, 128, 2, 2, false, true, true>",   "conv_igemm_kernel<128, 64, 2, 2, false, true, true>",
    "conv_igemm_kernel<64, 64, 2, 2, false, true, true>",     "conv_igemm_kernel<128, 32, 4, 1, false, true, true>"};
// ids 0 .. PROF_NKERNELS-1 are the fixed table above; kernels added later register their rocprofv3 symbol on first launch
// (kprof_id) and get the next id, up to PROF_MAX_KERNELS
constexpr int PROF_MAX_KERNELS = 160;
static std::vector<KRec> g_krecs[PROF_MAX_KERNELS];
static char g_knames_dyn[PROF_MAX_KERNELS][160];
static int g_nkernels = PROF_NKERNELS;

// Launches come from the compute and the weight-gradient streams, possibly from different host threads: registration is
// serialised (callers cache the id in a function-local static; a racing first call of two threads registers the name once and
// both get the same id).  A full table is reported, not silently aliased: the bench's per-kernel figures must not mix symbols.
static std::mutex g_kprof_mutex;
int kprof_id(const char* name) {
  std::lock_guard<std::mutex> lock(g_kprof_mutex);
  for (int k = PROF_NKERNELS; k < g_nkernels; ++k)
    if (strcmp(g_knames_dyn[k], name) == 0) return k;
  if (g_nkernels >= PROF_MAX_KERNELS) {
    static bool warned = false;
    if (!warned) {
      fprintf(stderr, "libudaseg_hip: kernel-timing table full (%d symbols); '%s' and later symbols are booked under '%s'\n",
              PROF_MAX_KERNELS, name, g_knames_dyn[PROF_NKERNELS]);
      warned = true;
    }
    return PROF_NKERNELS;
  }
  strncpy(g_knames_dyn[g_nkernels], name, sizeof(g_knames_dyn[0]) - 1);
  return g_nkernels++;
}
static const char* kname(int kid) {
  if (kid < 0 || kid >= g_nkernels) return "";
  return kid < PROF_NKERNELS ? g_knames_fixed[kid] : g_knames_dyn[kid];
}
static std::vector<ProfRec> g_recs[2];
static std::vector<hipEvent_t> g_pool;
static hipEvent_t g_open[2];

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

hipEvent_t kprof_begin(hipStream_t s) {
  if (!g_prof_on) return nullptr;
  hipEvent_t a = get_event();
  if (a) (void)hipEventRecord(a, s);
  return a;
}

void kprof_end(int kid, hipEvent_t a, hipStream_t s, double flops) {
  if (!a) return;
  hipEvent_t b = get_event();
  if (!b || kid < 0 || kid >= PROF_MAX_KERNELS) return;
  (void)hipEventRecord(b, s);
  g_krecs[kid].push_back({a, b, flops});
}

static int g_prof_suspended = 0;
void prof_suspend(int on) { g_prof_suspended = on; }   // a launcher that wraps another launcher keeps ONE record (its own)

void prof_begin(int family, hipStream_t s) {
  if (!g_prof_on || g_prof_suspended) return;
  g_open[family] = get_event();
  if (g_open[family]) (void)hipEventRecord(g_open[family], s);
}

void prof_end(int family, hipStream_t s, double flops, int kind, const udaseg_conv_desc* d) {
  if (!g_prof_on || g_prof_suspended || !g_open[family]) return;
  hipEvent_t b = get_event();
  if (!b) return;
  (void)hipEventRecord(b, s);
  ProfRec r = {g_open[family], b, flops, kind, {}};
  if (d) r.d = *d;
  g_recs[family].push_back(r);
  g_open[family] = nullptr;
}

}  // namespace udaseg

using namespace udaseg;

extern "C" int udaseg_version(void) { return 100; }
extern "C" const char* udaseg_last_error(void) { return g_err; }

extern "C" int udaseg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---- host-side helpers for the Python plan (round 4: BASELINE cfg 3 is bound by the HOST's launch rate, profiles/r04_host_bound.txt;
// a torch fill costs ~30 us of host time, a torch.cuda.Event pair ~8 us, these 3 us each)
extern "C" int udaseg_memset_async(void* ptr, int value, size_t bytes, void* stream) {
  UDASEG_CHECK_ARG(ptr != nullptr || bytes == 0, "memset_async: NULL pointer");
  if (bytes == 0) return UDASEG_OK;
  hipError_t e = hipMemsetAsync(ptr, value, bytes, as_stream(stream));
  if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync");
  return UDASEG_OK;
}

// `waiter` does not run past this point before everything enqueued on `signal` so far has finished (an event from a per-device ring:
// hipStreamWaitEvent captures the record it waits for at call time, so a slot may be recorded again while older waits are pending)
// Both streams must belong to the CURRENT device (the ring is per device; the engine's main and side streams are): an event
// recorded on another device's stream fails with hipErrorInvalidHandle, reported, not hidden.  Events are created one slot at a
// time on first use (a failed creation leaves the slot empty and is retried by the next call: nothing is leaked or overwritten).
extern "C" int udaseg_stream_wait(void* waiter, void* signal) {
  constexpr int RING = 256;
  static hipEvent_t ring[16][RING] = {};
  static unsigned next[16] = {};
  static std::mutex mu;
  int dev = 0;
  UDASEG_CHECK_ARG(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, "stream_wait: no current HIP device");
  std::lock_guard<std::mutex> lk(mu);
  const unsigned slot = next[dev] % RING;
  if (ring[dev][slot] == nullptr) {
    hipError_t e = hipEventCreateWithFlags(&ring[dev][slot], hipEventDisableTiming);
    if (e != hipSuccess) {
      ring[dev][slot] = nullptr;
      return hip_fail(e, "hipEventCreateWithFlags(stream_wait ring)");
    }
  }
  hipEvent_t ev = ring[dev][slot];
  hipError_t e = hipEventRecord(ev, as_stream(signal));
  if (e != hipSuccess) return hip_fail(e, "hipEventRecord(stream_wait: is `signal` a stream of the current device?)");
  e = hipStreamWaitEvent(as_stream(waiter), ev, 0);
  if (e != hipSuccess) return hip_fail(e, "hipStreamWaitEvent(stream_wait)");
  ++next[dev];
  return UDASEG_OK;
}

extern "C" int udaseg_prof_enable(int on) {
  g_prof_on = on != 0;
  return UDASEG_OK;
}

extern "C" int udaseg_prof_reset(void) {
  for (int f = 0; f < 2; ++f) {
    for (auto& r : g_recs[f]) {
      g_pool.push_back(r.a);
      g_pool.push_back(r.b);
    }
    g_recs[f].clear();
  }
  for (int k = 0; k < g_nkernels; ++k) {
    for (auto& r : g_krecs[k]) {
      g_pool.push_back(r.a);
      g_pool.push_back(r.b);
    }
    g_krecs[k].clear();
  }
  return UDASEG_OK;
}

extern "C" int udaseg_prof_read(int family, double* total_ms, double* total_flops, int64_t* launches) {
  UDASEG_CHECK_ARG(family >= 0 && family < 2 && total_ms && total_flops && launches, "prof_read: bad arguments");
  double ms = 0.0, fl = 0.0;
  for (auto& r : g_recs[family]) {
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize(prof)");
    float t = 0.f;
    e = hipEventElapsedTime(&t, r.a, r.b);
    if (e != hipSuccess) return hip_fail(e, "hipEventElapsedTime(prof)");
    ms += t;
    fl += r.flops;
  }
  *total_ms = ms;
  *total_flops = fl;
  *launches = (int64_t)g_recs[family].size();
  return UDASEG_OK;
}

extern "C" int udaseg_prof_records(int family, int max_records, double* ms, double* flops, int* kind, int* desc11) {
  UDASEG_CHECK_ARG(family >= 0 && family < 2 && ms && flops && kind && desc11 && max_records >= 0, "prof_records: bad arguments");
  int n = 0;
  for (auto& r : g_recs[family]) {
    if (n >= max_records) break;
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize(prof)");
    float t = 0.f;
    e = hipEventElapsedTime(&t, r.a, r.b);
    if (e != hipSuccess) return hip_fail(e, "hipEventElapsedTime(prof)");
    ms[n] = t;
    flops[n] = r.flops;
    kind[n] = r.kind;
    const int v[11] = {r.d.n, r.d.hi, r.d.wi, r.d.ci, r.d.ho, r.d.wo, r.d.co, r.d.kh, r.d.kw, r.d.stride, r.d.pad};
    for (int i = 0; i < 11; ++i) desc11[n * 11 + i] = v[i];
    ++n;
  }
  return n;
}

// ---- the switchboard (csrc/options.h): the ONLY place of the library that reads the environment
namespace udaseg {
enum { K_INT, K_FLAG, K_OFF0 };      // the variable's integer value | 1 when the variable exists | 0 when it is set to 0, else 1
struct OptRow {
  const char* env;
  int kind, dflt;
};
static const OptRow g_opt_rows[UDASEG_OPT_COUNT] = {
    /*  0 GENERIC_GATHER */ {"UDASEG_IGEMM_GENERIC", K_INT, 0},
    /*  1 F32_SPLIT */ {"UDASEG_F32_SPLIT", K_OFF0, 1},
    /*  2 WGRAD_GENERIC */ {"UDASEG_WGRAD_GENERIC", K_INT, 0},
    /*  3 F32_HALO */ {"UDASEG_F32_SPLIT", K_OFF0, 1},
    /*  4 F3_CFG */ {"UDASEG_F3_CFG", K_INT, 0},
    /*  5 F3_WS */ {"UDASEG_F3_WS", K_OFF0, 1},
    /*  6 F3_SIGNS */ {"UDASEG_F3_SIGNS", K_OFF0, 1},
    /*  7 IGEMM_TILE */ {"UDASEG_IGEMM_TILE", K_INT, 0},
    /*  8 IGEMM_X3 */ {"UDASEG_IGEMM_X3", K_INT, 1},
    /*  9 NO_FOLD */ {"UDASEG_NO_FOLD", K_FLAG, 0},
    /* 10 WGRAD_X3_BLOCKS */ {"UDASEG_WGRAD_X3_BLOCKS", K_INT, 1024},
    /* 11 WGRAD_BLOCKS */ {"UDASEG_WGRAD_BLOCKS", K_INT, 0},
    /* 12 WGRAD_NO_XCD */ {"UDASEG_WGRAD_NO_XCD", K_FLAG, 0},
    /* 13 WGRAD_X3 */ {"UDASEG_WGRAD_X3", K_OFF0, 1},
    /* 14 NO_WGRAD_HALO */ {"UDASEG_NO_WGRAD_HALO", K_FLAG, 0},
    /* 15 WGRAD_F3_BLOCKS */ {"UDASEG_WGRAD_F3_BLOCKS", K_INT, 120},
    /* 16 WGRAD_HALO_BLOCKS */ {"UDASEG_WGRAD_HALO_BLOCKS", K_INT, 96},
    /* 17 WGRAD_DEEP_BLOCKS */ {"UDASEG_WGRAD_DEEP_BLOCKS", K_INT, 0},
    /* 18 WGRAD_DB */ {"UDASEG_WGRAD_DB", K_INT, 1},
    /* 19 REDUCE_BLOCKS */ {"UDASEG_REDUCE_BLOCKS", K_INT, 0},
    /* 20 BN_APPLY_PT */ {"UDASEG_BN_APPLY_PT", K_INT, 4},
    /* 21 GEMM_1X1_TILE */ {"UDASEG_GEMM_1X1_TILE", K_INT, 0},
    /* 22 GEMM_1X1 */ {"UDASEG_GEMM_1X1", K_OFF0, 1},
    /* 23 GEMM_1X1_MAXM */ {"UDASEG_GEMM_1X1_MAXM", K_INT, 73728},
    /* 24 NO_STREAM */ {"UDASEG_NO_STREAM", K_FLAG, 0},
    /* 25 HALO_CFG */ {"UDASEG_HALO_CFG", K_INT, 0},
    /* 26 NO_HALO */ {"UDASEG_NO_HALO", K_FLAG, 0},
    /* 27 NO_HALO_S2 */ {"UDASEG_NO_HALO_S2", K_FLAG, 0},
    /* 28 HALO_W16 */ {"UDASEG_HALO_W16", K_INT, 6},
    /* 29 HALO_DEEP */ {"UDASEG_HALO_DEEP", K_OFF0, 1},
    /* 30 HALO_S2_CK */ {"UDASEG_HALO_S2_CK", K_INT, 64},
    /* 31 UP_CFG */ {"UDASEG_UP_CFG", K_INT, 0},
    /* 32 WGRAD_UP_BLOCKS */ {"UDASEG_WGRAD_UP_BLOCKS", K_INT, 0}};
int g_opt_val[UDASEG_OPT_COUNT];
static int g_opt_default[UDASEG_OPT_COUNT];
std::atomic<bool> g_opt_ready{false};
static std::atomic<int> g_opt_epoch{0};
static std::mutex g_opt_mutex;
void opt_init() {
  std::lock_guard<std::mutex> lk(g_opt_mutex);
  if (g_opt_ready.load(std::memory_order_relaxed)) return;
  for (int k = 0; k < UDASEG_OPT_COUNT; ++k) {
    const OptRow& r = g_opt_rows[k];
    const char* e = getenv(r.env);
    int v = r.dflt;
    if (r.kind == K_INT) v = e ? atoi(e) : r.dflt;
    else if (r.kind == K_FLAG) v = e != nullptr ? 1 : 0;
    else v = (e && atoi(e) == 0) ? 0 : 1;
    g_opt_default[k] = g_opt_val[k] = v;
  }
  g_opt_ready.store(true, std::memory_order_release);
}
}  // namespace udaseg

extern "C" int udaseg_set_option(int key, int value) {
  UDASEG_CHECK_ARG(key >= 0 && key < UDASEG_OPT_COUNT, "set_option: unknown key %d (0 .. %d)", key, UDASEG_OPT_COUNT - 1);
  UDASEG_CHECK_ARG(value >= -1, "set_option: value must be >= 0, or -1 for the default");
  udaseg::opt_get(key);           // the defaults are in place
  std::lock_guard<std::mutex> lk(udaseg::g_opt_mutex);
  udaseg::g_opt_val[key] = value < 0 ? udaseg::g_opt_default[key] : value;
  if (key == UDASEG_OPT_GENERIC_GATHER)      // one key for both gather loops, as before the table existed
    udaseg::g_opt_val[UDASEG_OPT_WGRAD_GENERIC] = value < 0 ? udaseg::g_opt_default[UDASEG_OPT_WGRAD_GENERIC] : value;
  udaseg::g_opt_epoch.fetch_add(1);
  return UDASEG_OK;
}

extern "C" int udaseg_get_option(int key) {
  if (key < 0 || key >= UDASEG_OPT_COUNT) return -1;
  return udaseg::opt_get(key);
}

extern "C" int udaseg_option_count(void) { return UDASEG_OPT_COUNT; }
extern "C" const char* udaseg_option_name(int key) { return key >= 0 && key < UDASEG_OPT_COUNT ? udaseg::g_opt_rows[key].env : ""; }
extern "C" int udaseg_option_epoch(void) { return udaseg::g_opt_epoch.load(); }

extern "C" int udaseg_prof_kernel_count(void) { return g_nkernels; }
extern "C" const char* udaseg_prof_kernel_name(int kid) { return kname(kid); }

extern "C" int udaseg_prof_kernel_read(int kid, double* total_ms, double* total_flops, int64_t* launches) {
  UDASEG_CHECK_ARG(kid >= 0 && kid < g_nkernels && total_ms && total_flops && launches, "prof_kernel_read: bad arguments");
  double ms = 0.0, fl = 0.0;
  for (auto& r : g_krecs[kid]) {
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize(kprof)");
    float t = 0.f;
    e = hipEventElapsedTime(&t, r.a, r.b);
    if (e != hipSuccess) return hip_fail(e, "hipEventElapsedTime(kprof)");
    ms += t;
    fl += r.flops;
  }
  *total_ms = ms;
  *total_flops = fl;
  *launches = (int64_t)g_krecs[kid].size();
  return UDASEG_OK;
}
