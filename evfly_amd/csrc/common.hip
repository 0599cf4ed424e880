// Error text, ABI version and the per-device scratch arena.
#include "common.h"

#include <map>
#include <mutex>
#include <tuple>

namespace evfly {

std::string &last_error_ref() {
    static thread_local std::string s;
    return s;
}

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

namespace {
struct Scratch {
    void *ptr = nullptr;
    size_t cap = 0;
};
std::map<std::tuple<int, hipStream_t, int>, Scratch> g_scratch;   // (device, stream, slot)
std::mutex g_scratch_mu;
}  // namespace

int scratch_get(size_t bytes, void **out, hipStream_t stream, int slot) {
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    Scratch &s = g_scratch[std::make_tuple(dev, stream, slot)];
    if (s.cap < bytes) {
        if (s.ptr) EVFLY_HIP(hipFree(s.ptr));
        s.ptr = nullptr;
        s.cap = 0;
        size_t want = bytes + (bytes >> 2) + (1 << 20);
        EVFLY_HIP(hipMalloc(&s.ptr, want));
        s.cap = want;
    }
    *out = s.ptr;
    return 0;
}

}  // namespace evfly

extern "C" int evfly_abi_version(void) { return EVFLY_ABI_VERSION; }
extern "C" const char *evfly_last_error(void) { return evfly::last_error_ref().c_str(); }
