// Library-level entry points of libpogs_rast.so: version, error text.
#include <stdio.h>

#include "common.h"

namespace pr {
static thread_local char g_hip_error[512] = "";
void set_hip_error(hipError_t e, const char *where) {
  snprintf(g_hip_error, sizeof(g_hip_error), "%s: %s", where, hipGetErrorString(e));
}
}  // namespace pr

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
