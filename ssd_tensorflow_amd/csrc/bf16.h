// bfloat16 storage type and device-side conversions (gfx950: v_cvt_pk_bf16_f32, round to nearest even).
// The bf16 configuration (BASELINE.json configs[2]) keeps fp32 master weights, fp32 accumulation on the
// matrix cores and fp32 loss math; activations, their gradients and the filter mirrors are bf16.
#pragma once
#include <hip/hip_runtime.h>

namespace ssd {

struct bf16_t {
    unsigned short v;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
// two floats -> one dword (lo = a, hi = b): ONE v_cvt_pk_bf16_f32 (the scalar form compiles to two of them plus and / shift / or)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ float lo2f(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float hi2f(unsigned w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }

// 4 consecutive channels as floats, from fp32 (16 bytes) or bf16 (8 bytes) storage
__device__ __forceinline__ f32x4 ld4t(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4t(const bf16_t* p) {
    const u32x2 w = *reinterpret_cast<const u32x2*>(p);
    return f32x4{lo2f(w[0]), hi2f(w[0]), lo2f(w[1]), hi2f(w[1])};
}
__device__ __forceinline__ void st4t(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4t(bf16_t* p, f32x4 v) {
    *reinterpret_cast<u32x2*>(p) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
}

}  // namespace ssd
