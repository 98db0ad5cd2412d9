// Shared host-side helpers for the gfx950 SSD-VGG library.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstdint>
#include <cstring>
#include <string>
#include <stdexcept>

namespace ssd {

// Thread-local last-error text returned by ssd_last_error().
void set_error(const char* fmt, ...);
const char* get_error();

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline void fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(buf);
}

#define HIP_OK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) ::ssd::fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define SSD_REQUIRE(cond, ...)                         \
    do {                                               \
        if (!(cond)) ::ssd::fail(__VA_ARGS__);         \
    } while (0)

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// TF SAME padding: pad_total = max((ceil(in/s)-1)*s + k_eff - in, 0); before = total/2.
inline void tf_same(int in, int k, int s, int d, int* before, int* out) {
    int keff = (k - 1) * d + 1;
    int o = (in + s - 1) / s;
    int tot = (o - 1) * s + keff - in;
    if (tot < 0) tot = 0;
    *before = tot / 2;
    *out = o;
}

}  // namespace ssd
