// Library plumbing: version, last-error text, launch check.
#include <stdarg.h>
#include <stdio.h>

#include "xvit_common.h"

namespace xvit {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return XVIT_OK;
}
}  // namespace xvit

extern "C" int xvit_version(void) { return XVIT_VERSION; }
extern "C" const char* xvit_last_error_string(void) { return xvit::g_err; }
