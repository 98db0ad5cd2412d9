// What does HBM take when a kernel only WRITES?  conv1_1's forward (368 MB of bf16 at batch 32, nothing to read but the image)
// and the un-pooling data gradients (4 x their neighbours' store bytes) sit at 2.8-3.0 TB/s whatever their occupancy
// (profiles/r06_w_*), while read+write passes (momentum, l2-norm) reach 6 TB/s.  This probe streams 16 bytes per lane:
//   fill   : stores only            (default policy / nontemporal)
//   read   : loads only
//   copy   : one load, one store    (default / nontemporal store)
//   rw2to1 : two loads per store (the momentum pattern's ratio)
// over a span well beyond the memory-side cache (1.5 GB) and over conv1_1's own 368 MB.
//   hipcc --offload-arch=gfx950 -O3 store_rate.hip -o store_rate.bin && ./store_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void fill_kernel(u32x4* dst, size_t n) {
    const u32x4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}
// a workgroup writes whole 4-KB pieces (the conv1_1 pattern: 32 pixels x 128 B per wave)
template <bool NT>
__global__ __launch_bounds__(256) void fill_chunks_kernel(u32x4* dst, size_t n) {
    const u32x4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
    const size_t per = 1024;      // 16 KB per workgroup and step
    for (size_t c = blockIdx.x; c * per < n; c += gridDim.x)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = c * per + j * 256 + threadIdx.x;
            if (i < n) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
        }
}
__global__ __launch_bounds__(256) void read_kernel(const u32x4* src, size_t n, unsigned* sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= src[i];
    if (acc[0] == 0x12345678u) sink[0] = acc[1] ^ acc[2] ^ acc[3];
}
template <bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* src, u32x4* dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u32x4 v = src[i];
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}
__global__ __launch_bounds__(256) void rw2to1_kernel(const u32x4* a, const u32x4* b, u32x4* dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = a[i] ^ b[i];
}

template <typename F>
static double time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main() {
    const size_t BIG = (size_t)1536 << 20;
    u32x4 *a, *b, *c; unsigned* sink;
    CK(hipMalloc(&a, BIG)); CK(hipMalloc(&b, BIG)); CK(hipMalloc(&c, BIG)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, BIG)); CK(hipMemset(b, 2, BIG)); CK(hipMemset(c, 3, BIG));
    for (size_t bytes : {(size_t)368640000, BIG}) {
        const size_t n = bytes / 16;
        for (int grid : {2048, 8192}) {
            const double mb = bytes / 1e6;
            double t;
            t = time_us([&] { hipLaunchKernelGGL(fill_kernel<false>, dim3(grid), dim3(256), 0, 0, c, n); }, 20);
            printf("%7.0f MB grid %5d  fill          %8.1f us  %6.2f TB/s (stored)\n", mb, grid, t, bytes / t / 1e6);
            t = time_us([&] { hipLaunchKernelGGL(fill_kernel<true>, dim3(grid), dim3(256), 0, 0, c, n); }, 20);
            printf("%7.0f MB grid %5d  fill nt       %8.1f us  %6.2f TB/s (stored)\n", mb, grid, t, bytes / t / 1e6);
            t = time_us([&] { hipLaunchKernelGGL(fill_chunks_kernel<false>, dim3(grid), dim3(256), 0, 0, c, n); }, 20);
            printf("%7.0f MB grid %5d  fill chunks   %8.1f us  %6.2f TB/s (stored)\n", mb, grid, t, bytes / t / 1e6);
            t = time_us([&] { hipLaunchKernelGGL(fill_chunks_kernel<true>, dim3(grid), dim3(256), 0, 0, c, n); }, 20);
            printf("%7.0f MB grid %5d  fill chunks nt%8.1f us  %6.2f TB/s (stored)\n", mb, grid, t, bytes / t / 1e6);
            t = time_us([&] { hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, n, sink); }, 20);
            printf("%7.0f MB grid %5d  read          %8.1f us  %6.2f TB/s (loaded)\n", mb, grid, t, bytes / t / 1e6);
            t = time_us([&] { hipLaunchKernelGGL(copy_kernel<false>, dim3(grid), dim3(256), 0, 0, a, c, n); }, 20);
            printf("%7.0f MB grid %5d  copy          %8.1f us  %6.2f TB/s (loaded + stored)\n", mb, grid, t, 2.0 * bytes / t / 1e6);
            t = time_us([&] { hipLaunchKernelGGL(copy_kernel<true>, dim3(grid), dim3(256), 0, 0, a, c, n); }, 20);
            printf("%7.0f MB grid %5d  copy nt       %8.1f us  %6.2f TB/s (loaded + stored)\n", mb, grid, t, 2.0 * bytes / t / 1e6);
            t = time_us([&] { hipLaunchKernelGGL(rw2to1_kernel, dim3(grid), dim3(256), 0, 0, a, b, c, n); }, 20);
            printf("%7.0f MB grid %5d  2 loads : 1 st%8.1f us  %6.2f TB/s (loaded + stored)\n", mb, grid, t, 3.0 * bytes / t / 1e6);
        }
    }
    return 0;
}
