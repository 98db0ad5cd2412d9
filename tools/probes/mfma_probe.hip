// Microbenchmark: what a wave's {LDS read -> v_mfma_f32_32x32x2_f32} loop can sustain on gfx950.
// variants: 0 = MFMA only (4 accumulators); 1 = + independent ds_read_b128 stream;
//           2 = operands re-read from LDS before every group of 4 MFMAs (dependent, like the conv loop)
//           3 = like 2 but 16 MFMAs per read group
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int lane = threadIdx.x & 63;
    float a0 = 1.f + lane, b0 = 0.5f;
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[k & 3], 0, 0, 0);
        } else if (V == 1) {
            f32x4 x = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + it * 64) & 8188));
            s += x[0];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[k & 3], 0, 0, 0);
        } else if (V == 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 x = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + (it * 4 + g) * 256) & 8188));
                f32x4 y = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + (it * 4 + g) * 256 + 1024) & 8188));
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[k], y[k], acc[k], 0, 0, 0);
            }
        } else {
            f32x4 x = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + it * 256) & 8188));
            f32x4 y = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + it * 256 + 1024) & 8188));
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[k & 3], y[(k >> 2) & 3], acc[k & 3], 0, 0, 0);
        }
    }
    float t = s;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) t += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int V>
void run(const char* name, int blocks_per_cu) {
    float* out;
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<V><<<blocks, 256>>>(out, 100);
    hipEventRecord(e0);
    probe<V><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
    printf("%-44s blocks/CU %d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, ms, fl / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int bpc : {1, 2, 3, 4}) {
        run<0>("mfma only, 4 acc", bpc);
        run<1>("mfma + independent ds_read_b128", bpc);
        run<2>("2x ds_read_b128 -> 4 mfma (dependent)", bpc);
        run<3>("2x ds_read_b128 -> 16 mfma (dependent)", bpc);
    }
    return 0;
}
