// How fast can ONE workgroup (one CU) stream a filter-sized span that (a) comes from beyond its XCD's L2 (first pass after a
// 256-MB sweep by another kernel) and (b) sits in that L2 (later passes inside the same launch)?  And does it matter that 32 or
// 256 workgroups read the SAME span at the same time (the tail chain: every image's workgroup streams the same filters)?
//   hipcc --offload-arch=gfx950 -O3 cu_stream.hip -o cu_stream.bin && ./cu_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int INFLIGHT>
__global__ __launch_bounds__(1024) void stream_kernel(const char* src, unsigned span, int passes, unsigned long long* ticks, int* sink) {
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, span, 0x00020000);
    i32x4 fold = {0, 0, 0, 0};
    const unsigned stride = blockDim.x * 16u;
    for (int p = 0; p < passes; ++p) {
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned base = 0; base < span; base += stride * INFLIGHT) {
            i32x4 v[INFLIGHT];
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) {
                const unsigned o = base + j * stride + tid * 16u;
                v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(o < span ? o : 0xFFFFFFF0u), 0, 0);
            }
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) fold ^= v[j];
        }
        __syncthreads();
        if (tid == 0) ticks[blockIdx.x * passes + p] = __builtin_amdgcn_s_memrealtime() - t0;
    }
    if (fold[0] == 0x12345678) sink[0] = fold[1] ^ fold[2] ^ fold[3];
}

__global__ void sweep_kernel(const int* src, size_t n, int* sink) {
    int acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i];
    if (acc == 0x12345678) sink[0] = acc;
}

int main() {
    char* src; int* big; int* sink; unsigned long long* ticks;
    const size_t BIG = 512u << 20;
    CK(hipMalloc(&src, 8 << 20)); CK(hipMalloc(&big, BIG)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&ticks, 256 * 8 * 8));
    CK(hipMemset(src, 1, 8 << 20)); CK(hipMemset(big, 2, BIG));
    const int passes = 4;
    std::vector<unsigned long long> h(256 * passes);
    for (unsigned span : {590u * 1024u, 3700u * 1024u}) {
        for (int grid : {1, 32, 256}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(sweep_kernel, dim3(2048), dim3(256), 0, 0, big, BIG / 4, sink);      // evict the span from L2 and the memory-side cache
                hipLaunchKernelGGL(stream_kernel<8>, dim3(grid), dim3(1024), 0, 0, src, span, passes, ticks, sink);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h.data(), ticks, grid * passes * 8, hipMemcpyDeviceToHost));
                printf("span %5u KB  grid %3d  rep %d:", span / 1024, grid, rep);
                for (int p = 0; p < passes; ++p) {
                    unsigned long long mx = 0, mn = ~0ull;
                    for (int g = 0; g < grid; ++g) { mx = h[g * passes + p] > mx ? h[g * passes + p] : mx; mn = h[g * passes + p] < mn ? h[g * passes + p] : mn; }
                    printf("  pass %d: %.1f-%.1f us (%.0f GB/s/CU)", p, mn / 100.0, mx / 100.0, span / (mx / 100.0) / 1e3);
                }
                printf("\n");
            }
        }
    }
    // does an XCD's L2 keep the span across a kernel boundary?  one pass per launch, three launches back to back after one sweep
    for (unsigned span : {590u * 1024u, 3700u * 1024u}) {
        for (int grid : {32, 256}) {
            hipLaunchKernelGGL(sweep_kernel, dim3(2048), dim3(256), 0, 0, big, BIG / 4, sink);
            printf("span %5u KB  grid %3d  one pass per LAUNCH:", span / 1024, grid);
            for (int l = 0; l < 3; ++l) {
                hipLaunchKernelGGL(stream_kernel<8>, dim3(grid), dim3(1024), 0, 0, src, span, 1, ticks, sink);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h.data(), ticks, grid * 8, hipMemcpyDeviceToHost));
                unsigned long long mx = 0;
                for (int g = 0; g < grid; ++g) mx = h[g] > mx ? h[g] : mx;
                printf("  launch %d: %.1f us (%.0f GB/s/CU)", l, mx / 100.0, span / (mx / 100.0) / 1e3);
            }
            printf("\n");
        }
    }
    // ... and when the FIRST launch is a 256-workgroup toucher (one load per 128-byte line) and the second the 32 streaming workgroups?
    return 0;
}
