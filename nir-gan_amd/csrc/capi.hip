// C-ABI housekeeping: version, thread-local last-error string.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void nirgan_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int nirgan_version(void) { return 100; }
extern "C" const char* nirgan_last_error(void) { return g_err; }
