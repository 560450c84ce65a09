// Library-level entry points of libdss2_hip.so: error string, version, topology hash.
#include <stdarg.h>

#include "dss2_common.hpp"

namespace dss2 {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace dss2

extern "C" const char* dss2_last_error(void) { return dss2::g_err; }

extern "C" int dss2_version(void) { return 1; }

