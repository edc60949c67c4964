#include "mst_common.h"
#include <cstdarg>
#include <cstdlib>
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
// Two rings per host thread: 16 slots for the small tables (level table, block origins: at most kSmallBytes each, so the ring
// never holds more than 1 MB of page-locked memory) and 4 slots for the occasional larger list (a difference-kernel tile list
// of a whole-genome launch: grow-only to the largest seen).  The slots are released when the thread ends -- the main thread's
// thread_local objects are destroyed at the start of exit(), before the HIP runtime's own teardown.
constexpr int kSmallSlots = 16, kLargeSlots = 4;
constexpr size_t kSmallBytes = 64 * 1024;
template <int N>
struct StageRing {
    StageSlot slot[N];
    int next = 0;
    ~StageRing() {
        for (StageSlot &q : slot) {
            if (q.pending && q.ev) (void)hipEventSynchronize(q.ev);
            if (q.ev) (void)hipEventDestroy(q.ev);
            if (q.p) (void)hipHostFree(q.p);
            q = StageSlot();
        }
    }
};

hipError_t stage(StageSlot &q, size_t granule, void *dst, const void *src, size_t bytes, hipStream_t s) {
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
        const size_t want = (bytes + granule - 1) / granule * granule;
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
}  // namespace

hipError_t upload_small(void *dst, const void *src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if (bytes <= kSmallBytes) {
        static thread_local StageRing<kSmallSlots> ring;
        StageSlot &q = ring.slot[ring.next];
        ring.next = (ring.next + 1) % kSmallSlots;
        return stage(q, kSmallBytes, dst, src, bytes, s);
    }
    static thread_local StageRing<kLargeSlots> big;
    StageSlot &q = big.slot[big.next];
    big.next = (big.next + 1) % kLargeSlots;
    return stage(q, 1 << 20, dst, src, bytes, s);
}

PinnedList::~PinnedList() { release(); }

void PinnedList::release() {
    (void)wait();
    for (int i = 0; i < 2; ++i) {
        if (ev[i]) (void)hipEventDestroy(ev[i]);
        ev[i] = nullptr;
    }
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    dev = -1;
}

hipError_t PinnedList::wait() {
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2; ++i) {
        if (pending[i] && ev[i]) {
            const hipError_t r = hipEventSynchronize(ev[i]);
            if (r != hipSuccess) e = r;
        }
        pending[i] = false;
    }
    return e;
}

hipError_t PinnedList::assign(const void *src, size_t n) {
    hipError_t e = wait();                                 // copies out of the old contents may still be in flight
    if (e != hipSuccess) return e;
    if (cap < n) {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = (n + 65535) / 65536 * 65536;
        e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        cap = want;
    }
    if (n) memcpy(p, src, n);
    bytes = n;
    return hipSuccess;
}

hipError_t PinnedList::upload(void *dst, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return e;
    if (dev != d) {                                        // events belong to a device
        (void)wait();
        for (int i = 0; i < 2; ++i) {
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            ev[i] = nullptr;
        }
        dev = d;
    }
    turn ^= 1;
    if (pending[turn] && ev[turn]) {                       // the copy before last: long done; keeps the bookkeeping exact
        e = hipEventSynchronize(ev[turn]);
        if (e != hipSuccess) return e;
        pending[turn] = false;
    }
    if (!ev[turn]) {
        e = hipEventCreateWithFlags(&ev[turn], hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    e = hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    e = hipEventRecord(ev[turn], s);
    if (e != hipSuccess) return e;
    pending[turn] = true;
    return hipSuccess;
}

namespace {
char g_notes[64][240];
unsigned g_note_at = 0;
}  // namespace

void note(const char *fmt, ...) {
    static const char *mode = getenv("MUSTACHE_GRAPH_DEBUG");
    if (!mode) return;
    char *line = g_notes[g_note_at++ % 64];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(line, sizeof(g_notes[0]), fmt, ap);
    va_end(ap);
    if (mode[0] == 'p') fprintf(stderr, "[note] %s\n", line);          // "print": every note at once (changes the timing)
}

void dump_notes() {
    const unsigned n = g_note_at < 64 ? g_note_at : 64;
    for (unsigned i = g_note_at - n; i != g_note_at; ++i) fprintf(stderr, "[note %u] %s\n", i, g_notes[i % 64]);
}

}  // namespace mst

extern "C" int mst_abi_version(void) { return MST_ABI_VERSION; }
extern "C" const char *mst_last_error(void) { return mst::error_buffer(); }

#ifdef MST_PROFILE
#include <rocprofiler-sdk-roctx/roctx.h>
namespace mst {
Range::Range(const char *name) { (void)roctxRangePushA(name); }
Range::~Range() { (void)roctxRangePop(); }
}  // namespace mst
#endif
