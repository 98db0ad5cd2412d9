// fp32 counterpart of mfma_loop.hip: 64 v_mfma_f32_32x32x2_f32 per wave and iteration on an LDS-resident 32-pixel stage
// (the weight-gradient form: x tile [32 px][128 ch], dy tile [32 px][128 ch], operands by ds_read_b32), with switches:
//   bit 0: LDS operand reads     bit 1: workgroup barrier per iteration     bit 2: 40 integer VALU ops per iteration
//   hipcc --offload-arch=gfx950 -O3 mfma_loop_f32.hip -o mfma_loop_f32.bin && ./mfma_loop_f32.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FLAGS>
__global__ __launch_bounds__(256) void loop_kernel(const float* __restrict__ src, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // 32 KB: x [32][128], dy [32][128]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) smem[i] = src[i];
    __syncthreads();
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const float* Xs = smem;
    const float* Ys = smem + 4096;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float fa[3][2], fb[3][2];
    auto frags = [&](int st) {
        const int r = st * 2 + lh;
        for (int mi = 0; mi < 2; ++mi) fa[st % 3][mi] = Xs[r * 128 + wm * 64 + mi * 32 + li];
        for (int ni = 0; ni < 2; ++ni) fb[st % 3][ni] = Ys[r * 128 + wn * 64 + ni * 32 + li];
    };
    frags(0); frags(1); frags(2);
    unsigned v0 = tid, v1 = tid * 3, v2 = tid * 5, v3 = tid * 7;
    for (int it = 0; it < iters; ++it) {
        if (FLAGS & 2) __builtin_amdgcn_s_barrier();
        if (FLAGS & 4) {
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                v0 = v0 * 3u + v1;
                v1 = (v1 >> 1) ^ v2;
                v2 = v2 + (v3 & 0xFFu);
                v3 = v3 ^ (v0 << 2);
            }
        }
        if (FLAGS & 1) { frags(0); frags(1); }
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            if ((FLAGS & 1) && st + 2 < 16) frags(st + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st % 3][mi], fb[st % 3][ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    if (s == 12345.678f || (v0 ^ v1 ^ v2 ^ v3) == 0x7FFFFFF1u) out[0] = s;
}

template <int FLAGS>
static void run(const float* src, float* out, int wgs_per_cu, const char* data) {
    const int iters = 500, grid = 256 * wgs_per_cu;
    auto k = loop_kernel<FLAGS>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = wgs_per_cu == 1 ? 96 * 1024 : (wgs_per_cu == 2 ? 64 * 1024 : 40 * 1024);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, src, out, iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double flops = (double)grid * 4 * iters * 64 * 4096.0;
    printf("%-6s wgs/cu=%d  lds_reads=%d barrier=%d valu40=%d   %7.3f ms  %6.1f TF  (%4.1f %% of 157.3)\n", data, wgs_per_cu, FLAGS & 1, (FLAGS >> 1) & 1,
           (FLAGS >> 2) & 1, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573);
}

int main() {
    float *src, *out;
    CK(hipMalloc(&src, 32768)); CK(hipMalloc(&out, 4));
    std::vector<float> h(8192);
    for (int pass = 0; pass < 2; ++pass) {
        for (auto& v : h) v = pass == 0 ? 0.f : (float)(rand() % 2001 - 1000) / 500.f;
        CK(hipMemcpy(src, h.data(), 32768, hipMemcpyHostToDevice));
        const char* d = pass == 0 ? "zeros" : "random";
        for (int w : {1, 2, 3}) {
            run<0>(src, out, w, d);
            run<1>(src, out, w, d);
            run<3>(src, out, w, d);
            run<7>(src, out, w, d);
        }
    }
    return 0;
}
