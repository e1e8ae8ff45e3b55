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

// process-wide: the device address every dropout-carrying launch passes on to its kernel (nullptr: seeds are used as given)
static const uint64_t* g_drop_epoch = nullptr;
namespace xvit { const uint64_t* drop_epoch_ptr() { return g_drop_epoch; } }
extern "C" int xvit_set_dropout_epoch(const uint64_t* device_counter) {
  XVIT_REQUIRE(((uintptr_t)device_counter & 7) == 0, "xvit_set_dropout_epoch: the counter must be 8-byte aligned");
  g_drop_epoch = device_counter;
  return XVIT_OK;
}
extern "C" const char* xvit_last_error_string(void) { return xvit::g_err; }
