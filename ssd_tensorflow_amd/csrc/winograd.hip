// Winograd F(4x4, 3x3) for the fp32 3x3 / stride 1 / SAME layers of the trunk (conv2_x ... conv5_x: ssdvgg.py:195-207 builds them
// through tf.nn.conv2d, whose cuDNN back end picks the same algorithm family on the reference's hardware).
//
// The fp32 matrix pipe is the bound of the fp32 step (157 TFLOP/s dense; the direct kernels of conv_igemm.hip sit at 118-131).  For a
// 3x3 filter the minimal-filtering form computes a 4x4 block of outputs from a 6x6 block of inputs with 36 multiplies per
// (input channel, output channel) instead of 144: 4x fewer MFMA flops (3.6x at 38x38 and 19x19, whose edge tiles are partly empty), paid
// for with three HBM-bound passes (input transform, output transform, and the transformed tensors the GEMMs stream):
//
//   forward        V = B^T d B  per 6x6 input tile            [36][T][Ci]     (wino_in_kernel; kept for the weight gradient)
//                  M_p = V_p . U_p,  U = G g G^T               36 GEMMs [T x Ci] . [Ci x Co]  (wino_gemm_nn_kernel, one launch)
//                  y = A^T M A + bias, relu                    (wino_out_kernel)
//   data gradient  the same three steps on dy with U' = G rot180(g)^T G^T ([36][Co][Ci]), masked by the producer's relu
//   weight grad.   dU_p = V_p^T . (A dy A^T)_p                 36 GEMMs [Ci x T] . [T x Co], split over T into slabs (wino_gemm_tn_kernel)
//                  dg = G^T (sum of slabs, fixed order) G + weight_decay g;  dbias = column sums of dy = a weighted sum of the
//                  column sums of (A dy A^T) at 16 positions (bias_axis_weight)
//                  (wino_wgrad_reduce_kernel)
//
// The same form serves the dilated layer (mod_conv6: every residue class of the dilation is an image of its own, wino_in_kernel) and the
// multibox heads of the big maps (output channels in multiples of 4: the data gradient's k is padded to 32 with zeros).  The relu mask a
// data gradient applies to dx is the layer's own input > 0: the forward's input transform has it in registers and leaves it as bits.
//
// Tiles t = (image, tile row, tile column) in raster order; T = B * ceil(H/4) * ceil(W/4).  Every transformed tensor is position-major
// [36][T][C] so that each of the 36 GEMMs reads plain row-major matrices and the transforms write whole 128-byte lines per (tile, position).
// The transforms are exact-arithmetic identities; in fp32 their rounding error is ~1e-6 of the output scale (tests: 1e-3 relative
// bound of BASELINE.json north_star, measured 6e-7 ... 6e-6), summation orders are fixed, so a step stays bit-reproducible run to run.
#include "conv.h"
#include "conv_detail.h"
#include "bf16.h"
#include <algorithm>
#include <cmath>

namespace ssd {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned WOOB = 0xFFFFFFF0u;      // offset no buffer covers: the load returns / the DMA writes zeros

// ---- the three transforms on 4-channel vectors --------------------------------------------------------------------------------
// Interpolation points 0, +-3/4, +-3/2, inf: Lavin & Gray's 0, +-1, +-2 scaled by 3/4 -- the same +- pairing, B^T and A^T still exact in
// fp32, 2.3x less rounding error (tools/probes/wino_points.py: 8x the direct fp32 sum's error instead of 19x).
// B^T (6x6)
__device__ __forceinline__ void bt6(f32x4& d0, f32x4& d1, f32x4& d2, f32x4& d3, f32x4& d4, f32x4& d5) {
    const f32x4 t0 = 1.265625f * d0 - 2.8125f * d2 + d4;
    const f32x4 e = d4 - 2.25f * d2, f = 1.6875f * d1 - 0.75f * d3;
    const f32x4 g = d4 - 0.5625f * d2, h = 0.84375f * d1 - 1.5f * d3;
    const f32x4 t5 = 1.265625f * d1 - 2.8125f * d3 + d5;
    d0 = t0; d1 = e - f; d2 = e + f; d3 = g - h; d4 = g + h; d5 = t5;
}
// A (6x4) applied to a 4-vector: the weight gradient's transform of a dy tile
__device__ __forceinline__ void a6(const f32x4 e0, const f32x4 e1, const f32x4 e2, const f32x4 e3, f32x4* u) {
    const f32x4 p = e0 + 0.5625f * e2, q = 0.75f * e1 + 0.421875f * e3;
    const f32x4 r = e0 + 2.25f * e2, t = 1.5f * e1 + 3.375f * e3;
    u[0] = e0;
    u[1] = p + q;
    u[2] = p - q;
    u[3] = r + t;
    u[4] = r - t;
    u[5] = e3;
}
// A^T (4x6) applied to a 6-vector: the output transform
__device__ __forceinline__ void at4(const f32x4 m0, const f32x4 m1, const f32x4 m2, const f32x4 m3, const f32x4 m4, const f32x4 m5,
                                    f32x4* y) {
    const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    y[0] = m0 + s12 + s34;
    y[1] = 0.75f * d12 + 1.5f * d34;
    y[2] = 0.5625f * s12 + 2.25f * s34;
    y[3] = 0.421875f * d12 + 3.375f * d34 + m5;
}

// The bias gradient is the sum of dy over the pixels = sum over tiles of 1^T e 1 = c^T (A e A^T) c with c^T A = 1^T: for these points
// c = (-7/9, 14/9, 2/9, 0, 0, 7/16), i.e. a weighted sum of the column sums of Ya at the 16 positions (k, l), k, l in {0, 1, 2, 5}.
__host__ __device__ inline int bias_axis_slot(int k) { return k == 0 ? 0 : k == 1 ? 1 : k == 2 ? 2 : k == 5 ? 3 : -1; }
__device__ __forceinline__ float bias_axis_weight(int slot) {
    return slot == 0 ? -7.f / 9.f : slot == 1 ? 14.f / 9.f : slot == 2 ? 2.f / 9.f : 7.f / 16.f;
}

// ---- input transforms: one thread = 4 channels of one tile ---------------------------------------------------------------------
// BT: V = B^T d B of the 6x6 patch at (4i - 1, 4j - 1), zero outside the image        (forward input; data gradient's dy)
// AT: Va = A e A^T of the patch's inner 4x4 (rows 4i ... 4i + 3), zero outside         (weight gradient's dy)
template <bool BT, bool AT>
__global__ __launch_bounds__(256) void wino_in_kernel(const float* __restrict__ x, float* __restrict__ V, float* __restrict__ Va,
                                                      int H, int W, int C, int Cp, int th, int tw, int T, unsigned x_bytes,
                                                      size_t v_ps, size_t va_ps, int D, unsigned long long* __restrict__ bits) {
    // Cp >= C: row length of V / Va (a GEMM's k must be a multiple of 32: the multibox heads' 104 / 152 channels are padded with zeros)
    const int c4n = Cp >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int q = idx % c4n, t = idx / c4n;
    if (t >= T) return;
    const int j = t % tw, t2 = t / tw;
    const int i = t2 % th, b = t2 / th;
    // dilation D: the pixels of one residue class (h mod D, w mod D) form an image of their own that the filter walks with unit steps;
    // tile row i = (tile of the class) * D + class.  D = 1: hb = 4 i, one class.
    const int hb = (i % D) + D * 4 * (i / D), wb = (j % D) + D * 4 * (j / D);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    constexpr int LO = BT ? 0 : 1, HI = BT ? 6 : 5;
    f32x4 d[6][6];
#pragma unroll
    for (int r = LO; r < HI; ++r) {
        const int h = hb + D * (r - 1);
        const bool hv = (unsigned)h < (unsigned)H;
#pragma unroll
        for (int s = LO; s < HI; ++s) {
            const int w = wb + D * (s - 1);
            const bool ok = hv && (unsigned)w < (unsigned)W && q * 4 < C;
            const unsigned off = ok ? (unsigned)(((b * H + h) * W + w) * C + q * 4) * 4u : WOOB;
            d[r][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
    }
    const size_t row = (size_t)t * Cp + q * 4;
    if constexpr (BT) {
        // forward only: which of the tile's 16 pixels x 4 channels are positive -- the relu mask the data gradient's output transform
        // of THIS layer applies to dx (conv.h wino_dgrad mask_bits): 8 bytes instead of a second read of the fp32 tensor
        if (bits) {
            unsigned long long m = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int e = 0; e < 4; ++e) m |= (unsigned long long)(d[r + 1][s + 1][e] > 0.f) << ((r * 4 + s) * 4 + e);
            bits[(size_t)t * c4n + q] = m;
        }
    }
    if constexpr (AT) {
        f32x4 c[6][4];      // A e: columns first
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 u[6];
            a6(d[1][s + 1], d[2][s + 1], d[3][s + 1], d[4][s + 1], u);
#pragma unroll
            for (int k = 0; k < 6; ++k) c[k][s] = u[k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            f32x4 u[6];
            a6(c[k][0], c[k][1], c[k][2], c[k][3], u);
#pragma unroll
            for (int l = 0; l < 6; ++l) *reinterpret_cast<f32x4*>(Va + (size_t)(k * 6 + l) * va_ps + row) = u[l];
        }
    }
    if constexpr (BT) {
#pragma unroll
        for (int s = 0; s < 6; ++s) bt6(d[0][s], d[1][s], d[2][s], d[3][s], d[4][s], d[5][s]);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            bt6(d[r][0], d[r][1], d[r][2], d[r][3], d[r][4], d[r][5]);
#pragma unroll
            for (int s = 0; s < 6; ++s) *reinterpret_cast<f32x4*>(V + (size_t)(r * 6 + s) * v_ps + row) = d[r][s];
        }
    }
}

// ---- output transform: y tile = A^T M A ------------------------------------------------------------------------------------------
// MODE 0 forward:        y = relu?(. + bias)
// MODE 1 data gradient:  dx = (. + old dx if accum), zero where mask <= 0 (mask: the producer's output, dx's shape)
// MODE 2 forward with the 2x2 / stride-2 pool behind it (conv.h conv_fwd_pool): the tile's four complete windows -> pooled tensor + record
// MODE 3 data gradient scattered through a pool's record into the pool's input gradient (conv.h conv_dgrad_unpool)
struct WinoOutArgs {
    const float* M;
    size_t m_ps;
    float* y;
    const float* bias;
    const float* mask;
    const unsigned long long* mask_bits;      // MODE 1: the mask as the forward's bits (one word per tile and 4 channels) instead of `mask`
    int relu, accum;
    int H, W, N, th, tw, T;
    int D;                             // dilation (MODE 0 / 1; the pool forms are D = 1)
    unsigned short* pool_rec;          // MODE 2: may be nullptr
    int PH, PW;                        // MODE 2: pooled size
    const unsigned short* unpool_rec;  // MODE 3
    int UH, UW;                        // MODE 3: un-pooled size
};

template <int MODE>
__global__ __launch_bounds__(256) void wino_out_kernel(WinoOutArgs p) {
    const int c4n = p.N >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int q = idx % c4n, t = idx / c4n;
    if (t >= p.T) return;
    const int j = t % p.tw, t2 = t / p.tw;
    const int i = t2 % p.th, b = t2 / p.th;
    const int D = (MODE == 0 || MODE == 1) ? p.D : 1;
    const int hb = (i % D) + D * 4 * (i / D), wb = (j % D) + D * 4 * (j / D);      // (wino_in_kernel)
    const float* src = p.M + (size_t)t * p.N + q * 4;
    f32x4 c[4][6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        f32x4 m[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) m[r] = *reinterpret_cast<const f32x4*>(src + (size_t)(r * 6 + s) * p.m_ps);
        f32x4 y[4];
        at4(m[0], m[1], m[2], m[3], m[4], m[5], y);
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r][s] = y[r];
    }
    f32x4 o[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) at4(c[r][0], c[r][1], c[r][2], c[r][3], c[r][4], c[r][5], o[r]);

    if constexpr (MODE == 0 || MODE == 2) {
        const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x4 v = o[r][s] + bv;
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                o[r][s] = v;
            }
    }
    if constexpr (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = hb + D * r;
            if (h >= p.H) continue;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int w = wb + D * s;
                if (w < p.W) *reinterpret_cast<f32x4*>(p.y + ((size_t)(b * p.H + h) * p.W + w) * p.N + q * 4) = o[r][s];
            }
        }
    } else if constexpr (MODE == 1) {
        const unsigned long long mbits = p.mask_bits ? p.mask_bits[(size_t)t * c4n + q] : 0ull;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = hb + D * r;
            if (h >= p.H) continue;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int w = wb + D * s;
                if (w >= p.W) continue;
                const size_t e0 = ((size_t)(b * p.H + h) * p.W + w) * p.N + q * 4;
                f32x4 v = o[r][s];
                if (p.accum) v += *reinterpret_cast<const f32x4*>(p.y + e0);
                if (p.mask_bits) {
                    const unsigned mb = (unsigned)(mbits >> ((r * 4 + s) * 4)) & 15u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (mb >> e) & 1u ? v[e] : 0.f;
                } else if (p.mask) {
                    const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + e0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
                }
                *reinterpret_cast<f32x4*>(p.y + e0) = v;
            }
        }
    } else if constexpr (MODE == 2) {
        // the pool's windows start at even rows / columns and the tile at a multiple of four: four complete windows per tile.
        // First maximum in scan order wins, cells outside the image never (ops.hip maxpool_fwd_rec); record = 3 bits per channel
        // (cell | positive << 2), four channels per 16-bit word.
#pragma unroll
        for (int wr = 0; wr < 2; ++wr) {
            const int ph = 2 * i + wr;
            if (ph >= p.PH) continue;
            const bool okh = 4 * i + 2 * wr + 1 < p.H;
#pragma unroll
            for (int wc = 0; wc < 2; ++wc) {
                const int pw = 2 * j + wc;
                if (pw >= p.PW) continue;
                const bool okw = 4 * j + 2 * wc + 1 < p.W;
                const bool ok[4] = {true, okw, okh, okw && okh};
                f32x4 mm = o[2 * wr][2 * wc];
                unsigned rec = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned a = 0;
                    float best = mm[e];
#pragma unroll
                    for (int cq = 1; cq < 4; ++cq) {
                        const float v = o[2 * wr + (cq >> 1)][2 * wc + (cq & 1)][e];
                        if (ok[cq] && v > best) { best = v; a = cq; }
                    }
                    mm[e] = best;
                    rec |= (a | (best > 0.f ? 4u : 0u)) << (3 * e);
                }
                const size_t pix = (size_t)(b * p.PH + ph) * p.PW + pw;
                *reinterpret_cast<f32x4*>(p.y + pix * p.N + q * 4) = mm;
                if (p.pool_rec) p.pool_rec[pix * c4n + q] = (unsigned short)rec;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = 4 * i + r;
            if (h >= p.H) continue;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int w = 4 * j + s;
                if (w >= p.W) continue;
                const unsigned rec = p.unpool_rec[((size_t)(b * p.H + h) * p.W + w) * c4n + q];
#pragma unroll
                for (int cq = 0; cq < 4; ++cq) {
                    const int uh = 2 * h + (cq >> 1), uw = 2 * w + (cq & 1);
                    if (uh >= p.UH || uw >= p.UW) continue;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned re = (rec >> (3 * e)) & 7u;
                        v[e] = ((re & 3u) == (unsigned)cq && (re & 4u)) ? o[r][s][e] : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(p.y + ((size_t)(b * p.UH + uh) * p.UW + uw) * p.N + q * 4) = v;
                }
            }
        }
    }
}

// ---- filter transforms ------------------------------------------------------------------------------------------------------------
// G (6x3) applied to a 3-vector
__device__ __forceinline__ void g6(const float g0, const float g1, const float g2, float* u) {
    u[0] = (64.f / 81.f) * g0;
    const float a = (-128.f / 243.f) * g0 - (8.f / 27.f) * g2, b = (32.f / 81.f) * g1;
    u[1] = a - b;
    u[2] = a + b;
    const float c = (32.f / 243.f) * g0 + (8.f / 27.f) * g2, d = (16.f / 81.f) * g1;
    u[3] = c + d;
    u[4] = c - d;
    u[5] = g2;
}
// U[p][ci][co] = (G g G^T)_p of g = w[.][.][ci][co]; FLIP: U'[p][co][ci] of the filter rotated by 180 degrees (the data gradient's).
// One launch transforms every layer of a plan (conv.h WinoFilterPlan): a block's layer is found by its first-block table.
template <bool FLIP>
__global__ __launch_bounds__(256) void wino_filter_kernel(WinoFilterPlan plan) {
    int li = 0;
    for (int k = 1; k < plan.n; ++k)
        if ((int)blockIdx.x >= plan.it[k].blk0) li = k;
    const float* __restrict__ w = plan.it[li].w;
    float* __restrict__ U = FLIP ? plan.it[li].Uf : plan.it[li].U;
    if (!U) return;      // (this layer does not take the pass that reads this transform)
    const int Ci = plan.it[li].Ci, Co = plan.it[li].Co;
    const int idx = ((int)blockIdx.x - plan.it[li].blk0) * 256 + threadIdx.x;
    if (idx >= Ci * Co) return;
    // forward: lanes along co (w's and U's rows); flipped: lanes along ci (U's rows; w is read across its rows, 9 values per thread)
    const int ci = FLIP ? idx % Ci : idx / Co, co = FLIP ? idx / Ci : idx % Co;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = w[((size_t)((FLIP ? 2 - a : a) * 3 + (FLIP ? 2 - b : b)) * Ci + ci) * Co + co];
    float c[6][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        float u[6];
        g6(g[0][b], g[1][b], g[2][b], u);
#pragma unroll
        for (int k = 0; k < 6; ++k) c[k][b] = u[k];
    }
    const size_t ps = FLIP ? (size_t)plan.it[li].Cop * Ci : (size_t)Ci * Co;      // (the flipped form has Cop >= Co rows; the pad rows stay zero)
    const size_t e = FLIP ? (size_t)co * Ci + ci : (size_t)ci * Co + co;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float u[6];
        g6(c[k][0], c[k][1], c[k][2], u);
#pragma unroll
        for (int l = 0; l < 6; ++l) U[(size_t)(k * 6 + l) * ps + e] = u[l];
    }
}

// ---- the 36 GEMMs of forward / data gradient: C_p[M x N] = A_p[M x K] . B_p[K x N] ---------------------------------------------------
// One launch; a 256-thread workgroup owns a (32 TM WM) x (32 TN WN) tile of one position.  Staging as in conv_gather_dma_kernel
// (conv_igemm.hip): both tiles go global -> LDS by DMA, 32 k per stage, two stages, one barrier per stage; A rows are 128-byte lines
// with their 16-byte chunks swizzled by (row >> 1) & 7 (conflict-free ds_read_b128 of 4 consecutive k), B is k-major [32][BN].
// Rows past M read as zeros through the buffer descriptor (no masks in the loop); K is a multiple of 32.
struct WinoGemmArgs {
    const float* A;
    const float* Bm;
    float* C;
    int M, N, K;
    size_t a_ps, b_ps, c_ps;      // elements between positions
    int MT, NT;
};

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void wino_gemm_nn_kernel(WinoGemmArgs p) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, BK = 32;
    constexpr int A_N = BM / 32;
    constexpr int B_CPR = BN / 4, B_RPP = 256 / B_CPR, B_N = BK / B_RPP;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BK % B_RPP == 0, "filter tile vs staging pass");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* const lds = reinterpret_cast<unsigned char*>(smem);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = wg % p.NT;
    wg /= p.NT;
    const int mt = wg % p.MT, pos = wg / p.MT;
    const int m0 = mt * BM, n0 = nt * BN;

    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A + (size_t)pos * p.a_ps), 0,
                                                                           (unsigned)((size_t)p.M * p.K * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Bm + (size_t)pos * p.b_ps), 0,
                                                                           (unsigned)((size_t)p.K * p.N * 4u), 0x00020000);
    const int a_c4 = ((tid & 7) ^ ((tid >> 4) & 7)) * 4;
    unsigned a_off[A_N], b_off[B_N];
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
        const int m = m0 + (tid >> 3) + 32 * i;
        a_off[i] = m < p.M ? (unsigned)(m * p.K + a_c4) * 4u : WOOB;
    }
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
        const int kr = tid / B_CPR + B_RPP * i, col = (tid % B_CPR) * 4;
        b_off[i] = n0 + col < p.N ? (unsigned)(kr * p.N + n0 + col) * 4u : WOOB;
    }
    const int nk = p.K / BK;
    auto issue = [&](int k, int stage) {
        unsigned char* As = lds + stage * STAGE + wave * 1024;
        unsigned char* Bs = lds + stage * STAGE + A_BYTES + wave * 1024;
        const int a_so = k * BK * 4, b_so = k * BK * p.N * 4;
#pragma unroll
        for (int i = 0; i < A_N; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, LDS_PTR(As + i * 4096), 16, (int)a_off[i], a_so, 0, 0);
#pragma unroll
        for (int i = 0; i < B_N; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, LDS_PTR(Bs + i * 4096), 16, (int)b_off[i], b_so, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, lh = lane >> 5;
    const int q0 = (lh ^ ((li >> 1) & 7)) * 16;
    const int a_row = (wm * 32 * TM + li) * 128 + q0;
    const int b_col = A_BYTES + (wn * 32 * TN + li) * 4;
    auto compute = [&](int stage) {
        const unsigned char* S = lds + stage * STAGE;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(S + ((a_row + mi * 4096) ^ (g * 32)));
            const int kb = g * 8 + lh * 4;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int t = 0; t < 4; ++t) b[ni][t] = *reinterpret_cast<const float*>(S + b_col + ((kb + t) * BN + ni * 32) * 4);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t], b[ni][t], acc[mi][ni], 0, 0, 0);
        }
    };
    if (nk > 0) issue(0, 0);
    for (int k = 0; k < nk; ++k) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (k + 1 < nk) issue(k + 1, (k + 1) & 1);
        compute(k & 1);
    }
    float* const Cp = p.C + (size_t)pos * p.c_ps;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n0 + wn * 32 * TN + ni * 32 + li;
            if (n >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M) Cp[(size_t)m * p.N + n] = acc[mi][ni][r];
            }
        }
}

// ---- the 36 GEMMs of the weight gradient: dU_p[Ci x Co] = sum_t V_p[t][Ci] . Ya_p[t][Co], split over t into slabs ---------------------
// Both operands keep their t-major global rows in LDS (conv_wgrad_dma_kernel's layout: ds_read_b32 fragments, a tile is a plain DMA
// copy).  Slab s = [36][Ci][Co] partial sums + [16][Co] column sums of Ya at the positions the bias gradient is made of.
struct WinoTnArgs {
    const float* X;      // V   [36][.][Ci]
    const float* Y;      // Ya  [36][.][Co]
    float* ws;           // [nsplit][36 * Ci * Co + Co]
    int T, Ci, Co;
    int x_ld, y_ld;      // row lengths of X / Y (>= Ci / Co)
    size_t x_ps, y_ps;
    int CT, NT, tchunk, nsplit;
};

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void wino_gemm_tn_kernel(WinoTnArgs p) {
    constexpr int BKT = 32 * TM * WM, BNT = 32 * TN * WN, BP = 32;
    constexpr int XC = BKT / 4, YC = BNT / 4;
    constexpr int XRPP = 256 / XC, YRPP = 256 / YC;
    constexpr int X_N = BP / XRPP, Y_N = BP / YRPP;
    constexpr int X_BYTES = BP * BKT * 4, STAGE = BP * (BKT + BNT) * 4;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* const lds = reinterpret_cast<unsigned char*>(smem);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int wg = xcd_remap(blockIdx.x, gridDim.x);
    // split fastest: the slabs of one (position, tile) are finished together and a position's V / Ya rows are walked by
    // neighbouring workgroups of one XCD at the same time
    const int nt = wg % p.NT;
    wg /= p.NT;
    const int ct = wg % p.CT;
    wg /= p.CT;
    const int split = wg % p.nsplit, pos = wg / p.nsplit;
    const int c0 = ct * BKT, n0 = nt * BNT;
    const int tbeg = split * p.tchunk, tend = min(p.T, tbeg + p.tchunk);
    const int niter = (tend - tbeg + BP - 1) / BP;
    const int bk = bias_axis_slot(pos / 6), bl = bias_axis_slot(pos % 6);
    const int bslot = (bk >= 0 && bl >= 0) ? 4 * bk + bl : -1;      // this position's column sums are a term of the bias gradient
    const bool do_bias = bslot >= 0 && ct == 0;

    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X + (size_t)pos * p.x_ps), 0,
                                                                           (unsigned)((size_t)tend * p.x_ld * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Y + (size_t)pos * p.y_ps), 0,
                                                                           (unsigned)((size_t)tend * p.y_ld * 4u), 0x00020000);
    const int xr = tid / XC, xc = tid % XC, yr = tid / YC, yc = tid % YC;
    const bool xcv = c0 + xc * 4 < p.Ci, ycv = n0 + yc * 4 < p.Co;
    unsigned xoff[X_N], yoff[Y_N];
    // rows at or past tend fall outside the descriptor (its size ends at row tend): zeros, no per-iteration test.  The offsets
    // advance in the VECTOR offset, which is what the descriptor's range check sees.
#pragma unroll
    for (int j = 0; j < X_N; ++j) xoff[j] = xcv ? (unsigned)(((tbeg + xr + j * XRPP) * p.x_ld + c0 + xc * 4) * 4) : WOOB;
#pragma unroll
    for (int j = 0; j < Y_N; ++j) yoff[j] = ycv ? (unsigned)(((tbeg + yr + j * YRPP) * p.y_ld + n0 + yc * 4) * 4) : WOOB;
    const unsigned xadv = xcv ? (unsigned)(BP * p.x_ld * 4) : 0u, yadv = ycv ? (unsigned)(BP * p.y_ld * 4) : 0u;
    auto issue = [&](int it, int stage) {
        unsigned char* Xs = lds + stage * STAGE + wave * 1024;
        unsigned char* Ys = lds + stage * STAGE + X_BYTES + wave * 1024;
#pragma unroll
        for (int j = 0; j < X_N; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, LDS_PTR(Xs + j * 4096), 16, (int)xoff[j], 0, 0, 0);
            xoff[j] += xadv;
        }
#pragma unroll
        for (int j = 0; j < Y_N; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rs, LDS_PTR(Ys + j * 4096), 16, (int)yoff[j], 0, 0, 0);
            yoff[j] += yadv;
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum = 0.f;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, lh = lane >> 5;
    auto compute = [&](int stage) {
        const float* Xs = reinterpret_cast<const float*>(lds + stage * STAGE);
        const float* Ys = reinterpret_cast<const float*>(lds + stage * STAGE + X_BYTES);
#pragma unroll 4
        for (int st = 0; st < BP / 2; ++st) {
            const int r = st * 2 + lh;
            float a[TM], b[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) a[mi] = Xs[r * BKT + wm * 32 * TM + mi * 32 + li];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) b[ni] = Ys[r * BNT + wn * 32 * TN + ni * 32 + li];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (do_bias && tid < BNT) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < BP; ++r) s += Ys[r * BNT + tid];
            bsum += s;
        }
    };
    if (niter > 0) issue(0, 0);
    for (int it = 0; it < niter; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + 1 < niter) issue(it + 1, (it + 1) & 1);
        compute(it & 1);
    }
    const size_t ucount = (size_t)36 * p.Ci * p.Co;
    float* slab = p.ws + (size_t)split * (ucount + 16 * (size_t)p.Co);
    float* up = slab + (size_t)pos * p.Ci * p.Co;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n0 + wn * 32 * TN + ni * 32 + li;
            if (n >= p.Co) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = c0 + wm * 32 * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (kl < p.Ci) up[(size_t)kl * p.Co + n] = acc[mi][ni][r];
            }
        }
    if (do_bias && tid < BNT && n0 + tid < p.Co) slab[ucount + (size_t)bslot * p.Co + n0 + tid] = bsum;
}

// G^T (3x6) applied to a 6-vector
__device__ __forceinline__ void gt3(const f32x4 s0, const f32x4 s1, const f32x4 s2, const f32x4 s3, const f32x4 s4, const f32x4 s5,
                                    f32x4* g) {
    const f32x4 a = s1 + s2, b = s2 - s1, c = s3 + s4, d = s3 - s4;
    g[0] = (64.f / 81.f) * s0 - (128.f / 243.f) * a + (32.f / 243.f) * c;
    g[1] = (32.f / 81.f) * b + (16.f / 81.f) * d;
    g[2] = (8.f / 27.f) * (c - a) + s5;
}
// dw[tap][ci][co] = (G^T (sum_s slab_s) G)_tap + wd * w ; dbias = sum_s slab_s' bias part.  One thread = (ci, 4 co); slabs added in order.
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ ws, int nsplit, int Ci, int Co,
                                                                float* __restrict__ dw, float* __restrict__ db,
                                                                const float* __restrict__ w, float wd) {
    const int c4n = Co >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const size_t cc = (size_t)Ci * Co, ucount = 36 * cc, stride = ucount + 16 * (size_t)Co;
    if (idx >= Ci * c4n) {
        const int q = idx - Ci * c4n;
        if (q < c4n && db) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int slot = 0; slot < 16; ++slot) {      // fixed order: position, then slab
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < nsplit; ++k) t += *reinterpret_cast<const f32x4*>(ws + (size_t)k * stride + ucount + (size_t)slot * Co + q * 4);
                s += (bias_axis_weight(slot >> 2) * bias_axis_weight(slot & 3)) * t;
            }
            *reinterpret_cast<f32x4*>(db + q * 4) = s;
        }
        return;
    }
    const int ci = idx / c4n, q = idx % c4n;
    const size_t e = (size_t)ci * Co + q * 4;
    f32x4 c[3][6];
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        // a column's six positions x four slabs = 24 loads in flight, added in slab order
        f32x4 s[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) s[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int sp0 = 0; sp0 < nsplit; sp0 += 4) {
            f32x4 v[6][4];
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    v[k][u] = *reinterpret_cast<const f32x4*>(ws + (size_t)min(sp0 + u, nsplit - 1) * stride + (size_t)(k * 6 + l) * cc + e);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (sp0 + u < nsplit) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) s[k] += v[k][u];
                }
        }
        f32x4 g[3];
        gt3(s[0], s[1], s[2], s[3], s[4], s[5], g);
#pragma unroll
        for (int a = 0; a < 3; ++a) c[a][l] = g[a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        f32x4 g[3];
        gt3(c[a][0], c[a][1], c[a][2], c[a][3], c[a][4], c[a][5], g);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const size_t o = (size_t)(a * 3 + b) * cc + e;
            f32x4 v = g[b];
            if (wd != 0.f) v += wd * *reinterpret_cast<const f32x4*>(w + o);
            *reinterpret_cast<f32x4*>(dw + o) = v;
        }
    }
}

// =====================================================================================================================================
// host side
// =====================================================================================================================================
// tiles along an axis of n pixels: each of the `dil` residue classes has ceil(n / dil) pixels = ceil(that / 4) tiles
static int tiles_1d(int n, int dil) { return dil * cdiv(cdiv(n, dil), 4); }
bool wino_applicable(const ConvDesc& d) {
    return d.KH == 3 && d.KW == 3 && d.stride == 1 && d.dil >= 1 && d.pad_h == d.dil && d.pad_w == d.dil && d.Ho == d.Hi && d.Wo == d.Wi &&
           d.Ci % 32 == 0 && d.Co % 4 == 0 && wino_tiles(d) >= 1 &&
           (long long)wino_tiles(d) * std::max(d.Ci, wino_kpad(d.Co)) < (1LL << 30) - 4 &&        // 32-bit byte offsets inside one position
           (long long)d.B * d.Hi * d.Wi * std::max(d.Ci, d.Co) < (1LL << 30) - 4;
}
int wino_kpad(int c) { return (c + 31) / 32 * 32; }
int wino_tiles(const ConvDesc& d) { return d.B * tiles_1d(d.Ho, d.dil) * tiles_1d(d.Wo, d.dil); }

static void require(const ConvDesc& d) { SSD_REQUIRE(wino_applicable(d), "winograd: 3x3 / stride 1 / SAME layers (any dilation), Ci in multiples of 32, Co of 4"); }

void WinoFilterPlan::add(const float* w, float* U, float* Uf, int Ci, int Co) {
    SSD_REQUIRE(n < MAX, "winograd: more than %d layers in one filter plan", MAX);
    it[n] = Item{w, U, Uf, Ci, Co, wino_kpad(Co), blocks};
    blocks += cdiv((long long)Ci * Co, 256);
    elems += (double)Ci * Co;
    ++n;
}
void wino_filter_plan(const WinoFilterPlan& plan, bool forward, bool flipped, hipStream_t s) {
    if (plan.n == 0) return;
    const double by = 4.0 * plan.elems * (9 + 36);
    if (forward) {
        ProfScope prof("wino_filter", 0, by, s);
        hipLaunchKernelGGL(wino_filter_kernel<false>, dim3(plan.blocks), dim3(256), 0, s, plan);
    }
    if (flipped) {
        ProfScope prof("wino_filter_flip", 0, by, s);
        hipLaunchKernelGGL(wino_filter_kernel<true>, dim3(plan.blocks), dim3(256), 0, s, plan);
    }
    HIP_OK(hipGetLastError());
}
void wino_filter(const ConvDesc& d, const float* w, float* U, float* Uflip, hipStream_t s) {
    require(d);
    WinoFilterPlan plan;
    plan.add(w, U, Uflip, d.Ci, d.Co);
    wino_filter_plan(plan, U != nullptr, Uflip != nullptr, s);
}

// x [B][H][W][C] -> V (B^T d B) and / or Va (A e A^T), both [36][.][C] with v_ps / va_ps elements between positions
static void launch_in(const float* x, float* V, float* Va, int B, int H, int W, int C, int Cp, int D, size_t v_ps, size_t va_ps, hipStream_t s,
                      unsigned long long* bits = nullptr) {
    const int th = tiles_1d(H, D), tw = tiles_1d(W, D), T = B * th * tw;
    const unsigned xb = (unsigned)((size_t)B * H * W * C * 4u);
    const int grid = cdiv((long long)T * (Cp / 4), 256);
    const double by = 4.0 * ((double)B * H * W * C + 36.0 * T * Cp * ((V ? 1 : 0) + (Va ? 1 : 0)));
    if (V && Va) {      // (one kernel for both was never measured faster than the data gradient's stream running its own: two launches)
        launch_in(x, V, nullptr, B, H, W, C, Cp, D, v_ps, 0, s, bits);
        launch_in(x, nullptr, Va, B, H, W, C, Cp, D, 0, va_ps, s);
        return;
    } else if (V) {
        ProfScope prof("wino_in", 0, by, s);
        hipLaunchKernelGGL((wino_in_kernel<true, false>), dim3(grid), dim3(256), 0, s, x, V, Va, H, W, C, Cp, th, tw, T, xb, v_ps, va_ps, D, bits);
    } else {
        ProfScope prof("wino_in_wgrad", 0, by, s);
        hipLaunchKernelGGL((wino_in_kernel<false, true>), dim3(grid), dim3(256), 0, s, x, V, Va, H, W, C, Cp, th, tw, T, xb, v_ps, va_ps, D, bits);
    }
    HIP_OK(hipGetLastError());
}

template <int WM, int WN, int TM, int TN>
static void launch_nn(WinoGemmArgs& a, const char* label, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr size_t lds = 2 * (size_t)(BM + BN) * 128;
    auto kern = wino_gemm_nn_kernel<WM, WN, TM, TN>;
    static bool once = (set_lds(kern, lds), true);
    (void)once;
    a.MT = cdiv(a.M, BM);
    a.NT = cdiv(a.N, BN);
    ProfScope prof(label, 2.0 * 36 * a.M * (double)a.N * a.K, 4.0 * 36 * ((double)a.M * a.K + (double)a.K * a.N + (double)a.M * a.N), s);
    hipLaunchKernelGGL(kern, dim3(36 * a.MT * a.NT), dim3(256), lds, s, a);      // (the pass's LAST launch carries its event: launch_out)
    HIP_OK(hipGetLastError());
}

static void gemm_nn(const float* A, size_t a_ps, const float* Bm, float* C, size_t c_ps, int M, int N, int K, hipStream_t s) {
    WinoGemmArgs a{};
    a.A = A; a.Bm = Bm; a.C = C; a.M = M; a.N = N; a.K = K;
    a.a_ps = a_ps; a.b_ps = (size_t)K * N; a.c_ps = c_ps;
    static const int forced = env_int("SSD_WINO_TILE", -1);
    // 128 x 128 unless its last row of tiles would be mostly empty (19 x 19 maps: 800 tiles at batch 32)
    int cfg = 0;
    if (forced >= 0) cfg = forced;
    else if (cdiv(N, 64) * 64 < cdiv(N, 128) * 128) cfg = 1;      // (a 128-wide n tile would multiply 64 columns of zeros: N = 64, the 152-channel heads)
    else if ((double)cdiv(M, 128) * 128 > 1.06 * M && (double)cdiv(M, 64) * 64 < (double)cdiv(M, 128) * 128) cfg = 2;
    switch (cfg) {
    case 0: launch_nn<2, 2, 2, 2>(a, "wino_gemm_128x128", s); break;
    case 1: launch_nn<4, 1, 1, 2>(a, "wino_gemm_128x64", s); break;
    case 2: launch_nn<2, 2, 1, 2>(a, "wino_gemm_64x128", s); break;
    default: launch_nn<2, 2, 1, 1>(a, "wino_gemm_64x64", s); break;
    }
}

template <int MODE>
static void launch_out(WinoOutArgs& a, const char* label, double bytes, hipStream_t s) {
    ProfScope prof(label, 0, bytes, s);
    SSD_LAUNCH_STOP(wino_out_kernel<MODE>, dim3(cdiv((long long)a.T * (a.N / 4), 256)), dim3(256), 0, s, a);
    HIP_OK(hipGetLastError());
}

size_t wino_fwd_ws_floats(const ConvDesc& d) { return (size_t)36 * wino_tiles(d) * d.Co; }

void wino_fwd(const ConvDesc& d, const float* x, const float* U, const float* bias, float* y, bool relu, float* V, size_t v_ps,
              float* Mws, float* y_pool, void* pool_rec, hipStream_t s, void* relu_bits) {
    require(d);
    const int T = wino_tiles(d);
    launch_in(x, V, nullptr, d.B, d.Hi, d.Wi, d.Ci, d.Ci, d.dil, v_ps, 0, s, static_cast<unsigned long long*>(relu_bits));
    gemm_nn(V, v_ps, U, Mws, (size_t)T * d.Co, T, d.Co, d.Ci, s);
    WinoOutArgs a{};
    a.M = Mws; a.m_ps = (size_t)T * d.Co; a.bias = bias; a.relu = relu; a.H = d.Ho; a.W = d.Wo; a.N = d.Co;
    a.th = tiles_1d(d.Ho, d.dil); a.tw = tiles_1d(d.Wo, d.dil); a.T = T; a.D = d.dil;
    const double mb = 4.0 * 36 * T * d.Co;
    if (y_pool) {
        SSD_REQUIRE(d.dil == 1, "winograd: the fused pool is for undilated layers");
        a.y = y_pool; a.pool_rec = static_cast<unsigned short*>(pool_rec); a.PH = (d.Ho + 1) / 2; a.PW = (d.Wo + 1) / 2;
        launch_out<2>(a, "wino_out_pool", mb + 4.0 * d.B * a.PH * a.PW * d.Co, s);
    } else {
        a.y = y;
        launch_out<0>(a, "wino_out", mb + 4.0 * d.B * d.Ho * d.Wo * d.Co, s);
    }
}

size_t wino_dgrad_ws_floats(const ConvDesc& d) { return (size_t)36 * wino_tiles(d) * d.Ci; }

void wino_bwd_transform(const ConvDesc& d, const float* dy, float* Yt, float* Ya, hipStream_t s) {
    require(d);
    const size_t ps = (size_t)wino_tiles(d) * wino_kpad(d.Co);
    launch_in(dy, Yt, Ya, d.B, d.Ho, d.Wo, d.Co, wino_kpad(d.Co), d.dil, ps, ps, s);
}

void wino_dgrad(const ConvDesc& d, const float* Yt, const float* Uflip, float* dx, const float* mask, bool accumulate, float* Xws,
                const void* unpool_rec, int UH, int UW, hipStream_t s, const void* mask_bits) {
    require(d);
    const int T = wino_tiles(d);
    gemm_nn(Yt, (size_t)T * wino_kpad(d.Co), Uflip, Xws, (size_t)T * d.Ci, T, d.Ci, wino_kpad(d.Co), s);
    WinoOutArgs a{};
    a.M = Xws; a.m_ps = (size_t)T * d.Ci; a.y = dx; a.mask = mask; a.accum = accumulate; a.H = d.Hi; a.W = d.Wi; a.N = d.Ci;
    a.mask_bits = mask ? static_cast<const unsigned long long*>(mask_bits) : nullptr;
    a.th = tiles_1d(d.Hi, d.dil); a.tw = tiles_1d(d.Wi, d.dil); a.T = T; a.D = d.dil;
    const double mb = 4.0 * 36 * T * d.Ci;
    if (unpool_rec) {
        SSD_REQUIRE(d.dil == 1, "winograd: the un-pooling form is for undilated layers");
        a.unpool_rec = static_cast<const unsigned short*>(unpool_rec); a.UH = UH; a.UW = UW;
        launch_out<3>(a, "wino_out_unpool", mb + 4.0 * d.B * UH * UW * d.Ci, s);
    } else {
        launch_out<1>(a, "wino_out_dgrad", mb + 4.0 * d.B * d.Hi * d.Wi * d.Ci * (mask && !a.mask_bits ? 2 : 1), s);
    }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------------------
// Splits of T: the launch should be a whole number of rounds of the chip's 512 workgroup slots (two 64 KB workgroups per CU) and every
// slab costs a write + a read of 36 Ci Co floats; modelled in microseconds, the cheapest of 1 ... 16 splits wins.
// tile of the weight-gradient GEMMs: 128 x 128 unless a channel count is 64 or less
static void tn_tile(const ConvDesc& d, int* bk, int* bn) {
    *bk = d.Ci <= 64 ? 64 : 128;
    *bn = d.Co <= 64 ? 64 : 128;
}
static int tn_splits(const ConvDesc& d) {
    const int T = wino_tiles(d);
    int bk, bn;
    tn_tile(d, &bk, &bn);
    const int base = 36 * cdiv(d.Ci, bk) * cdiv(d.Co, bn);
    const double slots = 256.0 * std::min(8, (int)(160 * 1024 / (2 * 32 * (bk + bn) * 4)));      // co-resident workgroups of the chip
    const double it_us = 3.4 * (bk * bn) / (128.0 * 128.0) * slots / 512.0;                     // one 32-row step of every resident workgroup
    int best = 1;
    double bc = 1e300;
    for (int ns = 1; ns <= 16; ++ns) {
        const int chunk = cdiv(cdiv(T, ns), 32) * 32;
        if (ns > 1 && chunk < 128) break;
        const double rounds = std::ceil((double)base * ns / slots);
        const double cost = rounds * (chunk / 32) * it_us + (ns > 1 ? 2.0 * ns * 36.0 * d.Ci * d.Co * 4.0 / 4.0e6 : 0.0);
        if (cost < bc) { bc = cost; best = ns; }
    }
    return best;
}
size_t wino_wgrad_ws_floats(const ConvDesc& d) { return (size_t)tn_splits(d) * ((size_t)36 * d.Ci * d.Co + 16 * (size_t)d.Co); }

template <int WM, int WN, int TM, int TN>
static void launch_tn(WinoTnArgs& a, const char* label, double flops, double bytes, hipStream_t s) {
    constexpr int BKT = 32 * TM * WM, BNT = 32 * TN * WN;
    constexpr size_t lds = 2 * (size_t)32 * (BKT + BNT) * 4;
    auto kern = wino_gemm_tn_kernel<WM, WN, TM, TN>;
    static bool once = (set_lds(kern, lds), true);
    (void)once;
    a.CT = cdiv(a.Ci, BKT); a.NT = cdiv(a.Co, BNT);
    ProfScope prof(label, flops, bytes, s);
    hipLaunchKernelGGL(kern, dim3(36 * a.CT * a.NT * a.nsplit), dim3(256), lds, s, a);
    HIP_OK(hipGetLastError());
}

void wino_wgrad(const ConvDesc& d, const float* V, size_t v_ps, const float* Ya, float* dw, float* dbias, const float* w,
                float weight_decay, float* ws, hipStream_t s) {
    require(d);
    const int T = wino_tiles(d);
    WinoTnArgs a{};
    a.X = V; a.Y = Ya; a.ws = ws; a.T = T; a.Ci = d.Ci; a.Co = d.Co; a.x_ps = v_ps; a.y_ps = (size_t)T * wino_kpad(d.Co);
    a.x_ld = d.Ci; a.y_ld = wino_kpad(d.Co);
    a.nsplit = tn_splits(d);
    a.tchunk = cdiv(cdiv(T, a.nsplit), 32) * 32;
    a.nsplit = cdiv(T, a.tchunk);      // (rounding the chunk up to whole iterations may empty the last split)
    SSD_REQUIRE((size_t)a.nsplit <= (size_t)tn_splits(d), "winograd: split plan");
    const double fl = 2.0 * 36 * T * (double)d.Ci * d.Co, by = 4.0 * 36 * ((double)T * (d.Ci + d.Co) + (double)a.nsplit * d.Ci * d.Co);
    int bk, bn;
    tn_tile(d, &bk, &bn);
    if (bk == 128 && bn == 128) launch_tn<2, 2, 2, 2>(a, "wino_gemm_tn_128x128", fl, by, s);
    else if (bk == 64 && bn == 128) launch_tn<2, 2, 1, 2>(a, "wino_gemm_tn_64x128", fl, by, s);
    else if (bk == 128 && bn == 64) launch_tn<4, 1, 1, 2>(a, "wino_gemm_tn_128x64", fl, by, s);
    else launch_tn<2, 2, 1, 1>(a, "wino_gemm_tn_64x64", fl, by, s);
    {
        const int n = d.Ci * (d.Co / 4) + d.Co / 4;
        ProfScope prof("wino_wgrad_reduce", 0, 4.0 * ((double)a.nsplit * 36 + 18) * d.Ci * d.Co, s);
        hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, ws, a.nsplit, d.Ci, d.Co, dw, dbias, w,
                           weight_decay);
        HIP_OK(hipGetLastError());
    }
}

}  // namespace ssd
