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

// Upper bound on the workgroups of the PERSISTENT kernels (attn_window16p_kernel walks its (window, head) items with one workgroup
// per CU): 0 = the device's 256.  A process-wide setting made through the ABI (it replaced a getenv() inside the launch path in
// round 5); the tests use it to make few workgroups walk many items on small problems.
namespace s6d {
int g_s6d_persistent_grid_limit = 0;
}
extern "C" int s6d_set_persistent_grid_limit(int max_workgroups) {
  if (max_workgroups < 0) return S6D_EINVAL;
  s6d::g_s6d_persistent_grid_limit = max_workgroups;
  return S6D_OK;
}

// Form of the bf16 / f16 GEMM kernel (include/sam6d_hip.h: s6d_set_gemm_wave_tile): 0 = per shape, 64 / 128 = forced.
namespace s6d {
int g_s6d_gemm_wave_tile = 0;
}
extern "C" int s6d_set_gemm_wave_tile(int columns) {
  if (columns != 0 && columns != 64 && columns != 128) return S6D_EINVAL;
  s6d::g_s6d_gemm_wave_tile = columns;
  return S6D_OK;
}

// 256 x 128 tiles for the plain / GELU GEMM launches that would leave most CUs idle (include/sam6d_hip.h: s6d_set_gemm_small_tile).
namespace s6d {
int g_s6d_gemm_small_tile = 1;
}
extern "C" int s6d_set_gemm_small_tile(int enable) {
  if (enable < 0 || enable > 2) return S6D_EINVAL;
  s6d::g_s6d_gemm_small_tile = enable;
  return S6D_OK;
}

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
