// Shared host-side helpers for libmustache_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/mustache_hip.h"

namespace mst {

char *error_buffer();                       // thread-local, 512 bytes
int fail(int code, const char *fmt, ...);   // formats into error_buffer(), returns code

#define MST_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return ::mst::fail(MST_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),   \
                               __FILE__, __LINE__);                                                \
    } while (0)

#define MST_LAUNCH_CHECK() MST_HIP(hipGetLastError())

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace mst
