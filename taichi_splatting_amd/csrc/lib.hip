// lib.hip — library-level entry points: version and last-error string.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace ms {
static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace ms

extern "C" int ms_version(void) { return MS_VERSION; }
extern "C" const char* ms_last_error_string(void) { return ms::g_error; }
