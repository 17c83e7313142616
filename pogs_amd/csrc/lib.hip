// Library-level entry points of libpogs_rast.so: version, error text.
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "common.h"

namespace pr {
static thread_local char g_hip_error[512] = "";
void set_hip_error(hipError_t e, const char *where) {
  snprintf(g_hip_error, sizeof(g_hip_error), "%s: %s", where, hipGetErrorString(e));
}
void set_error_text(const char *text) { snprintf(g_hip_error, sizeof(g_hip_error), "%s", text); }

static Tuning g_tuning;
static std::once_flag g_tuning_once;
static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
static void read_tuning() {
  g_tuning.fwd_ppl = env_int("POGS_RAST_FWD_PPL", 0);
  g_tuning.fwd_chunk = env_int("POGS_RAST_FWD_CHUNK", 0);
  g_tuning.fwd_mfma = env_int("POGS_RAST_FWD_MFMA", 0);
  g_tuning.fwd_blocks = env_int("POGS_RAST_FWD_BLOCKS", -1);
  g_tuning.bwd_ppl = env_int("POGS_RAST_BWD_PPL", 0);
  g_tuning.bwd_mfma = env_int("POGS_RAST_BWD_MFMA", -1);
  g_tuning.bwd_window = env_int("POGS_RAST_BWD_WINDOW", 0);
  g_tuning.mlp_variant = env_int("POGS_MLP_VARIANT", -1);
  g_tuning.depth_sort = env_int("POGS_RAST_DEPTH_SORT", 0);
}
const Tuning &tuning() {
  std::call_once(g_tuning_once, read_tuning);
  return g_tuning;
}
}  // namespace pr

extern "C" void pr_reload_tuning(void) {
  (void)pr::tuning();
  pr::read_tuning();
}

extern "C" int pr_version(void) { return (1 << 16) | 0; }

extern "C" const char *pr_last_hip_error(void) { return pr::g_hip_error; }

extern "C" const char *pr_error_string(int code) {
  switch (code) {
    case PR_OK: return "ok";
    case PR_EINVAL: return "invalid argument";
    case PR_EUNSUPPORTED: return "unsupported configuration";
    case PR_ESCRATCH: return "scratch buffer too small";
    case PR_EHIP: return "HIP runtime error";
    default: return "unknown error code";
  }
}
