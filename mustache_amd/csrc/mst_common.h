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

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: raise it once per (call site, device),
// so a process that drives several GPUs gets it on each of them.  `done` = the call site's static bit mask of devices.
inline hipError_t allow_dynamic_lds(const void *kernel, int bytes, unsigned long long *done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(done, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) __atomic_fetch_or(done, bit, __ATOMIC_RELEASE);
    return e;
}

// Host -> device upload of a small table (level table, block starts, tile lists) whose source may be reused or freed as
// soon as the call returns (a stack object, a thread_local vector, a ctypes array the Python caller drops): the bytes are
// copied into a PINNED staging slot first and the asynchronous copy reads from there, so its correctness does not rest on
// the runtime staging pageable sources before hipMemcpyAsync returns.  A ring of slots per host thread; a slot is reused
// only after the event recorded behind its copy has completed (normally long ago).
hipError_t upload_small(void *dst, const void *src, size_t bytes, hipStream_t s);

// A host list that is uploaded again and again unchanged (the work list of a launch, kept in a per-thread cache of recent
// launches): ONE page-locked copy made when the list is built (assign), every launch copies straight out of it (upload: no host
// memcpy, no staging slot), and the memory is released with its owner -- after the last copy out of it has completed.
struct PinnedList {
    void *p = nullptr;
    size_t cap = 0, bytes = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};                 // behind the last two copies (callers alternate two streams)
    int dev = -1, turn = 0;
    bool pending[2] = {false, false};
    PinnedList() = default;
    PinnedList(const PinnedList &) = delete;
    PinnedList &operator=(const PinnedList &) = delete;
    ~PinnedList();
    void release();
    hipError_t wait();
    hipError_t assign(const void *src, size_t n);          // waits for copies still reading the old contents
    hipError_t upload(void *dst, hipStream_t s);
};

// PROFILE builds mark the stage every entry point belongs to (read / normalise / launch / finish / tail) as a roctx range, so
// that `rocprofv3 --marker-trace` shows the stages of a run next to its kernels (profiles/r06_marker_trace.md); the product
// library carries no marker and does not link the roctx library.
#ifdef MST_PROFILE
struct Range {
    explicit Range(const char *name);
    ~Range();
};
#define MST_RANGE(name_) mst::Range mst_range_scope_(name_)
#else
#define MST_RANGE(name_)
#endif

// diagnostics (MUSTACHE_GRAPH_DEBUG set): a short in-memory ring of notes about recent launches, printed when a call fails
void note(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void dump_notes();

}  // namespace mst
