#include "common.h"

#include <stdarg.h>

namespace collie {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", static_cast<int>(e), hipGetErrorString(e), what);
  return -2;
}

}  // namespace collie

extern "C" const char* collie_last_error(void) { return collie::g_err; }
extern "C" int collie_version(void) { return 100; }
