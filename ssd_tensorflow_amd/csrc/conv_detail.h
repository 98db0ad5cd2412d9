// Pieces shared by the fp32 (conv_igemm.hip) and bf16 (conv_bf16.hip) convolution kernels.
#pragma once
#include "conv.h"
#include <cstdlib>

namespace ssd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// XCD-aware bijective remap: workgroup b runs on XCD b % 8; give every XCD a
// contiguous run of tiles so neighbouring tiles share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

inline int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

template <typename K>
inline void set_lds(K kern, size_t bytes) {
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

// algorithmic work of one conv pass (forward, dgrad or wgrad alike): 2*M*N*K
inline double conv_flops(const ConvDesc& d) { return 2.0 * d.B * d.Ho * d.Wo * (double)d.Co * d.Ci * d.KH * d.KW; }
// the elements a pass must move at least once (activations in + out, the filter)
inline double conv_elems(const ConvDesc& d) {
    return (double)d.B * d.Hi * d.Wi * d.Ci + (double)d.B * d.Ho * d.Wo * d.Co + (double)d.KH * d.KW * d.Ci * d.Co;
}

// Fixed-order reduce of the split-M weight-gradient slabs (conv_igemm.hip): dw = sum_s slab_s + wd*w, db = sum_s bias_s.
// While a ReduceBatch is installed (conv.h) the call only queues its descriptor: the step executor flushes a whole
// backward stage's reduces as ONE grouped launch.
void wgrad_reduce(const float* ws, int nsplit, size_t wcount, int Co, float* dw, float* db, const float* w, float wd,
                  hipStream_t s);

}  // namespace ssd
