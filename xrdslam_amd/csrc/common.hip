// Error plumbing shared by all entry points.
#include "common.h"

namespace xrd {
thread_local const char* g_last_error = "";

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_error = hipGetErrorString(e);
    (void)what;
    return XRD_ERR_LAUNCH;
  }
  return XRD_OK;
}
}  // namespace xrd

extern "C" {
int xrd_abi_version(void) { return 1; }
const char* xrd_last_error(void) { return xrd::g_last_error; }
}
