// Library-level entry points: version and thread-local error message.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void dgtta_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int dgtta_version(void) { return 10000; /* 1.0.0 */ }
extern "C" const char *dgtta_last_error(void) { return g_err; }
