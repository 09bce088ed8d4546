// Error reporting + version for libds2hip.so.
#include <stdarg.h>

#include "ds2_host.h"
#if __has_include("build_id.h")
#include "build_id.h"       // csrc/build/build_id.h, written by build.py: a digest of the sources this library is built from
#endif
#ifndef DS2_BUILD_ID
#define DS2_BUILD_ID "unstamped"
#endif

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
extern "C" int ds2_version(void) { return 402; }
// the digest of the sources this binary was built from (csrc/build.py: source_id()); ds2hip/lib.py compares it with the tree
extern "C" const char* ds2_build_id(void) { return DS2_BUILD_ID; }
