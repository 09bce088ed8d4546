// Error reporting + version for libds2hip.so.
#include <stdarg.h>

#include "ds2_host.h"

static thread_local char g_err[512] = "";

void ds2_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ds2_last_error(void) { return g_err; }
// = DS2_ABI_VERSION of include/ds2hip.h (the header is C documentation of the ABI and is not included by the sources; the
// CPU test suite compares the two numbers)
extern "C" int ds2_version(void) { return 401; }
