// Error/version plumbing of the C ABI (include/sam6d_hip.h).
#include "s6d_common.h"

#include <string.h>

namespace s6d {
static thread_local char g_hip_err[256] = {0};
void set_hip_error(hipError_t e) {
  const char *s = hipGetErrorString(e);
  strncpy(g_hip_err, s ? s : "unknown", sizeof(g_hip_err) - 1);
}
}  // namespace s6d

extern "C" int s6d_version(void) { return S6D_ABI_VERSION; }

extern "C" const char *s6d_last_hip_error(void) { return s6d::g_hip_err; }

extern "C" const char *s6d_strerror(int code) {
  switch (code) {
    case S6D_OK: return "ok";
    case S6D_EINVAL: return "invalid argument (size, null pointer or shape)";
    case S6D_ELAUNCH: return "HIP kernel launch failed";
    case S6D_EUNSUPPORTED: return "shape not supported by this build";
    default: return "unknown sam6d_hip error";
  }
}
