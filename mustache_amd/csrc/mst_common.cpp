#include "mst_common.h"
#include <cstring>

namespace mst {

char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

namespace {
struct StageSlot {
    void *p = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    int dev = -1;
    bool pending = false;
};
constexpr int kStageSlots = 16;
struct StageRing {
    StageSlot slot[kStageSlots];
    int next = 0;
    // no destructor: a host thread's ring lives as long as the thread; at process exit the HIP runtime may already be gone
};
}  // namespace

hipError_t upload_small(void *dst, const void *src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    static thread_local StageRing ring;
    StageSlot &q = ring.slot[ring.next];
    ring.next = (ring.next + 1) % kStageSlots;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (q.pending) {
        e = hipEventSynchronize(q.ev);
        if (e != hipSuccess) return e;
        q.pending = false;
    }
    if (q.ev && q.dev != dev) {
        (void)hipEventDestroy(q.ev);
        q.ev = nullptr;
    }
    if (!q.ev) {
        e = hipEventCreateWithFlags(&q.ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
        q.dev = dev;
    }
    if (q.cap < bytes) {
        if (q.p) (void)hipHostFree(q.p);
        q.p = nullptr;
        q.cap = 0;
        const size_t want = (bytes + 65535) / 65536 * 65536;
        e = hipHostMalloc(&q.p, want, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        q.cap = want;
    }
    memcpy(q.p, src, bytes);
    e = hipMemcpyAsync(dst, q.p, bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    e = hipEventRecord(q.ev, s);
    if (e != hipSuccess) return e;
    q.pending = true;
    return hipSuccess;
}

}  // namespace mst

extern "C" int mst_abi_version(void) { return MST_ABI_VERSION; }
extern "C" const char *mst_last_error(void) { return mst::error_buffer(); }
