// How fast can one CU pull L2-resident bytes into LDS (or registers), and is the cap per CU or shared?
// Each workgroup streams 1-KiB pieces (64 lanes x 16 B) from a small, L2-resident span, P pieces per wave and stage,
// keeping `depth` stages in flight (counted vmcnt waits, no barriers), and reports its shader-clock span.
//   hipcc --offload-arch=gfx950 -O3 dma_rate.hip -o dma_rate.bin && ./dma_rate.bin
// Output: bytes / clk / CU for  path {lds-dma, vgpr} x grid {32, 64, 128, 256, 512} x waves {4, 8} x depth {1, 2, 3}.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int P>
__device__ __forceinline__ void wait_keep(int keep_stages) {      // leave keep_stages * P pieces in flight
    if (keep_stages <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (keep_stages == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
    else if (keep_stages == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * P) : "memory");
}

// MODE 0: buffer_load_dwordx4 ... lds     MODE 1: buffer_load_dwordx4 to VGPRs (xor-folded so they are not dead)
// SEG: contiguous bytes per tile row inside a piece (1024 = fully contiguous piece; 128 / 64 = conv tile rows at a 1-KB pitch)
template <int MODE, int P, int SEG = 1024>
__global__ void stream_kernel(const char* src, unsigned span, int iters, int depth, unsigned long long* clk, int* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, span, 0x00020000);
    // every wave walks its own 1-KiB pieces through the span; workgroups start at different places
    constexpr unsigned LPR = SEG / 16;                      // lanes per row
    const unsigned lane_off = SEG == 1024 ? lane * 16u : (lane / LPR) * 1024u + (lane % LPR) * 16u;      // row pitch 1 KB
    unsigned off = (unsigned)(((blockIdx.x * nw + wave) * P * 1024u * 7u) % span) + lane_off;
    const unsigned step = (unsigned)(nw * P * 1024u * 13u) % span;
    i32x4 fold = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        unsigned char* stage = smem + (it % (depth + 1)) * (nw * P * 1024) + wave * P * 1024;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            unsigned o = off + j * (SEG == 1024 ? 1024u : SEG);          // strided mode: the next piece takes the next SEG bytes of the same rows
            o -= o >= span ? span : 0u;
            if (MODE == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(stage + j * 1024), 16, (int)o, 0, 0, 0);
            else {
                i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0);
                fold ^= v;
            }
        }
        off += step;
        off -= off >= span ? span : 0u;
        if (MODE == 0) {
            if (depth == 0) wait_keep<P>(0);
            else if (depth == 1) wait_keep<P>(1);
            else if (depth == 2) wait_keep<P>(2);
            else wait_keep<P>(3);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
    if (MODE == 1 && (fold[0] ^ fold[1] ^ fold[2] ^ fold[3]) == 0x12345678) sink[0] = 1;
}

template <int MODE, int P, int SEG = 1024>
static void run(const char* src, unsigned span, int grid, int waves, int depth, unsigned long long* dclk, int* sink) {
    const int iters = 400;
    const size_t lds = (size_t)(depth + 1) * waves * P * 1024;
    if (lds > 160 * 1024) return;
    auto k = stream_kernel<MODE, P, SEG>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // pad the LDS request so that exactly one workgroup fits a CU: `grid` workgroups = `grid` CUs (up to 256), 512 = two rounds
    const size_t lds_req = std::max(lds, (size_t)96 * 1024);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(waves * 64), lds_req, 0, src, span, iters, depth, dclk, sink);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), dclk, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double bytes = (double)iters * waves * P * 1024;
    // s_memtime / readcyclecounter tick = 100 MHz constant clock on gfx9: convert with the event time instead
    const int cus = std::min(grid, 256), rounds = (grid + 255) / 256;
    const double gbs_cu = bytes * rounds / (ms * 1e-3) / 1e9;                 // per CU
    printf("seg=%4d %-5s P=%d grid=%3d waves=%d depth=%d  %8.3f ms  %7.1f GB/s/CU  %6.1f B/clk/CU@2.4GHz  chip %6.2f TB/s\n", SEG, MODE ? "vgpr" : "lds", P, grid,
           waves, depth, ms, gbs_cu, gbs_cu / 2.4, gbs_cu * cus / 1e3);
}

int main() {
    const unsigned span = 2u << 20;            // 2 MiB: resident in every XCD's 4-MiB L2
    char* src;
    unsigned long long* dclk;
    int* sink;
    CK(hipMalloc(&src, span)); CK(hipMemset(src, 1, span));
    CK(hipMalloc(&dclk, 1024 * sizeof(unsigned long long))); CK(hipMalloc(&sink, 4));
    for (int grid : {32, 128, 256, 512})
        for (int waves : {4, 8})
            for (int depth : {1, 3}) {
                run<0, 4>(src, span, grid, waves, depth, dclk, sink);
            }
    for (int grid : {32, 256})
        for (int waves : {4, 8}) run<0, 8>(src, span, grid, waves, 1, dclk, sink);
    for (int grid : {32, 256})
        for (int waves : {4, 8, 16}) run<1, 4>(src, span, grid, waves, 0, dclk, sink);
    for (int grid : {32, 256})
        for (int waves : {4, 8}) run<1, 8>(src, span, grid, waves, 0, dclk, sink);
    // conv-tile shaped pieces: 8 rows x 128 B and 16 rows x 64 B per 1-KB piece, rows 1 KB apart
    for (int waves : {4, 8}) {
        run<0, 4, 128>(src, span, 256, waves, 1, dclk, sink);
        run<0, 4, 64>(src, span, 256, waves, 1, dclk, sink);
        run<0, 8, 128>(src, span, 256, waves, 1, dclk, sink);
        run<0, 8, 64>(src, span, 256, waves, 1, dclk, sink);
    }
    return 0;
}
