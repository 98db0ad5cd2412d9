// Shared host-side helpers for the gfx950 SSD-VGG library.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdarg>
#include <cstdint>
#include <cstring>
#include <string>
#include <stdexcept>
#include <vector>

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

// RAII: make `device` the calling thread's current HIP device for the scope and put the caller's
// device back afterwards.  Every handle-based or device-numbered C entry point holds one, so a rank of
// a multi-GPU job can mix calls on its own GPU with free functions without its kernels migrating.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device) {
        HIP_OK(hipGetDevice(&prev));
        if (prev != device) {
            HIP_OK(hipSetDevice(device));
            switched = true;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// Per-launch timing with HIP events on the launching stream (bench.py's roofline block).
// Off by default; a Net turns it on for its steps.  One record per kernel launch.
struct Profiler {
    struct Rec {
        std::string kernel;
        double flops, bytes;
        hipEvent_t e0, e1;
    };
    bool on = false;
    bool detailed = false;          // label records "kernel:layer"
    const char* layer = "";         // set by the executor around each op
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    hipEvent_t get();
    void reset() { recs.clear(); used = 0; }
    ~Profiler();
};
extern thread_local Profiler* g_prof;

// A launch can carry the event its consumers on OTHER streams wait for (hipExtLaunchKernelGGL's stop event: the kernel packet's
// own completion signal).  An event recorded BEHIND the kernel is a packet of its own that the stream's next kernel queues
// behind: on this chip a chain of dependent kernels pays 1.5 us per link bare, 6.8 us with "record, the other stream waits and
// runs a kernel", 3.3 us when the kernel carries the event (tools/probes/event_gap.hip, profiles/r05_s_event_gap.txt).  The
// executor arms g_stop_event right before the one call whose (single, last) launch should carry it; the gather launchers take it.
extern thread_local hipEvent_t g_stop_event;
inline hipEvent_t take_stop_event() {
    hipEvent_t e = g_stop_event;
    g_stop_event = nullptr;
    return e;
}
#define SSD_LAUNCH_STOP(kern, grid, block, lds, stream, ...) \
    hipExtLaunchKernelGGL(kern, grid, block, lds, stream, nullptr, ssd::take_stop_event(), 0, __VA_ARGS__)

// RAII: records an event before and after the launches issued inside the scope.
struct ProfScope {
    hipStream_t s;
    bool active;
    size_t idx;
    ProfScope(const char* kernel, double flops, double bytes, hipStream_t stream);
    ~ProfScope();
};

// Measurement aid (net.hip): groups of launches dropped from the step, set through ssd_debug_set_ablate only
void set_ablate(const char* tokens);
bool ablated(const char* token);

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
