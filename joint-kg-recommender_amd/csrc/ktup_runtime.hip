// Error reporting for libktup_hip.so: thread-local message, no global mutable state.
#include <cstdarg>
#include <cstdio>

#include "ktup_common.h"

namespace ktup {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(KTUP_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return KTUP_OK;
}

}  // namespace ktup

extern "C" int ktup_version(void) { return 1; }
extern "C" const char* ktup_last_error(void) { return ktup::g_err; }
