// What does an event between two dependent kernels cost the stream that records it?  (round 5)
// A chain of N dependent kernels of ~T us on stream A; variants:
//   plain        nothing between them
//   record       hipEventRecord(ev, A) after every kernel
//   record+wait  ... and stream B waits for it and runs a short kernel (the weight-gradient pattern of the step executor)
//   ext          hipExtLaunchKernelGGL with a stop event (the kernel packet's own completion signal), B waits for that
//   wait-in      A waits for an event recorded on B before every kernel (B runs a short kernel first)
// build: hipcc --offload-arch=gfx950 -O2 -o event_gap.bin event_gap.hip ; run: ./event_gap.bin [spin_us]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define OK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

__global__ void spin(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1);
}

int main(int argc, char** argv) {
    const double us = argc > 1 ? atof(argv[1]) : 10.0;
    const int wgs = argc > 2 ? atoi(argv[2]) : 64;
    const long long cyc = (long long)(us * 100.0);      // wall_clock64: 100 MHz
    const int N = 200;
    hipStream_t A, B;
    OK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    OK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    int* sink;
    OK(hipMalloc(&sink, 4));
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t one;
    OK(hipEventCreateWithFlags(&one, hipEventDisableTiming));
    auto run = [&](const char* name, auto body) {
        for (int rep = 0; rep < 3; ++rep) {
            OK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            body();
            auto t1 = std::chrono::steady_clock::now();
            OK(hipDeviceSynchronize());
            auto t2 = std::chrono::steady_clock::now();
            if (rep == 2)
                printf("%-28s %7.2f us per link (host issue %5.2f us per link)\n", name,
                       std::chrono::duration<double, std::micro>(t2 - t0).count() / N, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        }
    };
    printf("chain of %d kernels, %d workgroups x 256 threads, %.1f us each\n", N, wgs, us);
    run("plain", [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, A, cyc, sink); });
    run("record (one event, reused)", [&] {
        for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, A, cyc, sink); OK(hipEventRecord(one, A)); } });
    run("record (event per link)", [&] {
        for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, A, cyc, sink); OK(hipEventRecord(ev[i], A)); } });
    run("record + B waits, B kernel", [&] {
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, A, cyc, sink);
            OK(hipEventRecord(one, A));
            OK(hipStreamWaitEvent(B, one, 0));
            hipLaunchKernelGGL(spin, dim3(4), dim3(256), 0, B, cyc / 4, sink);
        } });
    run("ext stop event, B waits", [&] {
        for (int i = 0; i < N; ++i) {
            hipExtLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, A, nullptr, ev[i], 0, cyc, sink);
            OK(hipStreamWaitEvent(B, ev[i], 0));
            hipLaunchKernelGGL(spin, dim3(4), dim3(256), 0, B, cyc / 4, sink);
        } });
    run("every 2nd link: record + B", [&] {
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, A, cyc, sink);
            if (i & 1) {
                OK(hipEventRecord(one, A));
                OK(hipStreamWaitEvent(B, one, 0));
                hipLaunchKernelGGL(spin, dim3(4), dim3(256), 0, B, cyc / 4, sink);
                hipLaunchKernelGGL(spin, dim3(4), dim3(256), 0, B, cyc / 4, sink);
            }
        } });
    run("A waits for B before each", [&] {
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(spin, dim3(4), dim3(256), 0, B, cyc / 4, sink);
            OK(hipEventRecord(one, B));
            OK(hipStreamWaitEvent(A, one, 0));
            hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, A, cyc, sink);
        } });
    run("ping-pong A <-> B", [&] {
        for (int i = 0; i < N; ++i) {
            hipStream_t s = (i & 1) ? B : A, o = (i & 1) ? A : B;
            hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, s, cyc, sink);
            OK(hipEventRecord(one, s));
            OK(hipStreamWaitEvent(o, one, 0));
        } });
    return 0;
}
