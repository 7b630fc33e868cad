// common.hip — error plumbing and ABI version of libhqq_hip.so
#include <stdarg.h>
#include <string.h>

#include "hqq_common.h"

namespace hqq {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
    return static_cast<int>(e);
  }
  return 0;
}
}  // namespace hqq

extern "C" {
int hqq_hip_abi_version(void) { return HQQ_HIP_ABI_VERSION; }
const char* hqq_hip_last_error(void) { return hqq::g_err; }
}
