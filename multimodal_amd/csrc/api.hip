// libmmhip: version + thread-local error string of the C ABI (include/mmhip.h).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void mm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int mm_version(void) { return 100; }
extern "C" const char* mm_last_error(void) { return g_err; }
