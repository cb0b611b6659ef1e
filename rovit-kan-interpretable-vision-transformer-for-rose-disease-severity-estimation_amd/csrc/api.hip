// Library-level entry points: version and the thread-local error string.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void rovit_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int rovit_version(void) { return 100; }
extern "C" const char* rovit_last_error_string(void) { return g_err; }
