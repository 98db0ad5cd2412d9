// Ground-truth probe for three gfx950 features the bf16 path relies on (run on the GPU box):
//   1. v_mfma_f32_32x32x16_bf16 operand layout (A: row = lane&31, k = 8*(lane>>5)+j; B likewise by column)
//   2. ds_read_b64_tr_b16: which LDS element lands in (lane, j) for per-lane addresses
//   3. buffer_load_dwordx4 ... lds: where each lane's 16 bytes land, and what an out-of-range lane writes
// hipcc --offload-arch=gfx950 -O2 bf16_probe.hip -o bf16_probe && ./bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return u >> 16; }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void mfma_probe(const unsigned short* A, const unsigned short* B, float* D) {
    // A [32][16] row-major (i, k), B [16][32] row-major (k, j); D [32][32]
    const int l = threadIdx.x, li = l & 31, lh = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = __builtin_bit_cast(__bf16, A[li * 16 + lh * 8 + j]);
        b[j] = __builtin_bit_cast(__bf16, B[(lh * 8 + j) * 32 + li]);
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = c[r];
}

__global__ void tr_probe(short* out, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[2048];
    const int l = threadIdx.x;
    for (int i = l; i < 2048; i += 64) lds[i] = (short)i;
    __syncthreads();
    int idx;
    if (mode == 0) idx = l * 4;                                  // lane-linear 8-byte pieces
    else {                                                       // [row = q>>2][quad = q&3] inside each 16-lane group, row pitch 64 elements
        const int g = l >> 4, q = l & 15;
        idx = g * 512 + (q >> 2) * 64 + (q & 3) * 4;
    }
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + idx));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

__global__ void glds_probe(const int* g, int* out, int nbytes) {
    __shared__ __attribute__((aligned(16))) int lds[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = -7;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(g), 0, nbytes, 0x00020000);
    // lane l fetches the 16 bytes at element offset 4*((l*5)%64); lanes 3 and 40 are steered out of range
    unsigned off = (unsigned)(((l * 5) % 64) * 16);
    if (l == 3 || l == 40) off = 0xFFFFFFF0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, off, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = l; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
    // ---- 1. MFMA layout
    std::vector<unsigned short> A(32 * 16), B(16 * 32);
    std::vector<float> Af(32 * 16), Bf(16 * 32), D(32 * 32), R(32 * 32, 0.f);
    srand(1);
    for (int i = 0; i < 512; ++i) { A[i] = f2bf((rand() % 2001 - 1000) / 500.f); Af[i] = bf2f(A[i]); B[i] = f2bf((rand() % 2001 - 1000) / 500.f); Bf[i] = bf2f(B[i]); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += (double)Af[i * 16 + k] * Bf[k * 32 + j]; R[i * 32 + j] = (float)s; }
    unsigned short *dA, *dB; float* dD;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 4096));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    mfma_probe<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    double me = 0; for (int i = 0; i < 1024; ++i) me = fmax(me, fabs(D[i] - R[i]));
    printf("MFMA 32x32x16 bf16 natural-layout max abs err %.3e (%s)\n", me, me < 1e-4 ? "LAYOUT OK" : "LAYOUT MISMATCH");
    // ---- 2. transpose read
    short* dO; CK(hipMalloc(&dO, 512)); std::vector<short> O(256);
    for (int mode = 0; mode < 2; ++mode) {
        tr_probe<<<1, 64>>>(dO, mode);
        CK(hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost));
        printf("ds_read_b64_tr_b16 mode %d (lane: 4 element indices)\n", mode);
        for (int l = 0; l < 64; ++l) printf("%2d:[%4d %4d %4d %4d]%s", l, O[l * 4], O[l * 4 + 1], O[l * 4 + 2], O[l * 4 + 3], (l & 3) == 3 ? "\n" : "  ");
    }
    // ---- 3. buffer load to LDS
    std::vector<int> G(256), L(1024);
    for (int i = 0; i < 256; ++i) G[i] = 1000 + i;
    int *dG, *dL; CK(hipMalloc(&dG, 1024)); CK(hipMalloc(&dL, 4096));
    CK(hipMemcpy(dG, G.data(), 1024, hipMemcpyHostToDevice));
    glds_probe<<<1, 64>>>(dG, dL, 1024);
    CK(hipMemcpy(L.data(), dL, 4096, hipMemcpyDeviceToHost));
    printf("buffer_load_dwordx4 lds: lds[256 + 4*l .. +3] per lane (expect 1000 + 4*((5l)%%64) + 0..3; lanes 3, 40 out of range)\n");
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        const int* p = &L[256 + 4 * l];
        const int e = 1000 + 4 * ((l * 5) % 64);
        const bool oob = l == 3 || l == 40;
        if (oob ? !(p[0] == 0 && p[3] == 0) : !(p[0] == e && p[3] == e + 3)) ok = 0;
        if (l < 8 || oob) printf("  lane %2d: %d %d %d %d\n", l, p[0], p[1], p[2], p[3]);
    }
    int untouched = 1;
    for (int i = 0; i < 256; ++i) if (L[i] != -7) untouched = 0;
    for (int i = 512; i < 1024; ++i) if (L[i] != -7) untouched = 0;
    printf("glds lane-linear image + zero fill: %s; rest untouched: %s\n", ok ? "OK" : "MISMATCH", untouched ? "yes" : "NO");
    return 0;
}
