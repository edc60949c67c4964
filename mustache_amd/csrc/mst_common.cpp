#include "mst_common.h"

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

}  // namespace mst

extern "C" int mst_abi_version(void) { return MST_ABI_VERSION; }
extern "C" const char *mst_last_error(void) { return mst::error_buffer(); }
