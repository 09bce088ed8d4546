// Host-only part of the shared helpers: status macros + the thread-local error string.  No HIP header: the two host-side
// translation units (api.hip, decode_host.hip) also compile with plain g++ -- that is how the AddressSanitizer /
// UndefinedBehaviorSanitizer build of the host C++ is made (csrc/build.py::build_host_sanitized, tests/test_host_asan_cpu.py).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "ds2hip.h"

void ds2_set_error(const char* fmt, ...);

#define DS2_CHECK_ARG(cond)                                                        \
    do {                                                                           \
        if (!(cond)) {                                                             \
            ds2_set_error("%s: bad argument: %s", __func__, #cond);                \
            return DS2_ERR_ARG;                                                    \
        }                                                                          \
    } while (0)
