// First layer, fp32 (conv1_1: 3 input channels, 3x3): the whole reduction k = tap*Ci + c fits one 32-deep
// block, so the layer is bound by writing its 64-channel output.  Every lane gathers the image values its
// MFMA operands need straight from the fp32 image (zero padding = out-of-range buffer offsets), the filter
// sits in registers for the whole kernel (exact fp32 products on v_mfma_f32_32x32x2_f32), and outputs leave
// through a per-wave LDS tile as full rows, 16 bytes per lane.  (The generic gather kernel's packed small-C
// path stored 4 bytes per lane: 0.40 ms; this kernel is bound by the 737 MB write.)
#include "conv.h"
#include "conv_detail.h"
#include "bf16.h"

namespace ssd {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct FirstF32Args {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    int M, Hi, Wi, Ci, Ho, Wo, Co;
    int ntaps, stride, relu, ntiles;
    int tap_dh[9], tap_dw[9];
};

template <int NT>
__global__ __launch_bounds__(256) void conv_first_fwd_f32_kernel(FirstF32Args p) {
    constexpr int ROWF = NT * 32 + 4;                       // floats per LDS row (+4: conflict-free float4 writes)
    __shared__ __attribute__((aligned(16))) float smem[4 * 32 * ROWF];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    float* T = smem + wave * 32 * ROWF;
    const int K = p.ntaps * p.Ci;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    // MFMA k-step j multiplies k = 2j + lh: filter operand rows = output channels, image operand columns = pixels
    float wa[NT][16];
    int koff[16];
    unsigned kbit[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = 2 * j + lh;
        const bool kv = k < K;
        const int tp = kv ? k / p.Ci : 0, c = kv ? k - tp * p.Ci : 0;
        koff[j] = ((p.tap_dh[tp] * p.Wi + p.tap_dw[tp]) * p.Ci + c) * 4;
        kbit[j] = kv ? (1u << tp) : 0u;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wa[nt][j] = kv ? p.w[(size_t)k * p.Co + nt * 32 + li] : 0.f;
    }
    float bv[NT][4][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[nt][g][e] = p.bias ? p.bias[nt * 32 + 8 * g + 4 * lh + e] : 0.f;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((size_t)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.Ci * 4u), 0x00020000);

    const int nwaves = gridDim.x * 4;
    for (int tile = blockIdx.x * 4 + wave; tile < p.ntiles; tile += nwaves) {
        const int m = tile * 32 + li;
        const int mm = m < p.M ? m : 0;
        const int ow = mm % p.Wo;
        const int t2 = mm / p.Wo;
        const int oh = t2 % p.Ho;
        const int b = t2 / p.Ho;
        const int h0 = oh * p.stride, w0 = ow * p.stride;
        unsigned mk = 0;
        for (int t = 0; t < p.ntaps; ++t) {
            const int sh = h0 + p.tap_dh[t], sw = w0 + p.tap_dw[t];
            if ((unsigned)sh < (unsigned)p.Hi && (unsigned)sw < (unsigned)p.Wi) mk |= 1u << t;
        }
        if (m >= p.M) mk = 0;
        const unsigned base = (unsigned)(((b * p.Hi + h0) * p.Wi + w0) * p.Ci) * 4u;
        float xb[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned ok = 0u - (unsigned)((mk & kbit[j]) != 0u);
            xb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (int)(((base + (unsigned)koff[j]) & ok) | (OOB & ~ok)), 0, 0));
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[nt][j], xb[j], acc, 0, 0, 0);
            // D rows = channels (r&3) + 8*(r>>2) + 4*lh, D col = pixel li
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[4 * g + e] + bv[nt][g][e];
                    if (p.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                *reinterpret_cast<f32x4*>(T + li * ROWF + nt * 32 + 8 * g + 4 * lh) = v;
            }
        }
        constexpr int CPR = NT * 8;                  // 16-byte chunks per pixel row
        constexpr int PPI = 64 / CPR;                // pixels per store instruction
#pragma unroll
        for (int i = 0; i < 32 / PPI; ++i) {
            const int px = i * PPI + lane / CPR, ch = lane % CPR;
            const int mo = tile * 32 + px;
            const f32x4 v = *reinterpret_cast<const f32x4*>(T + px * ROWF + ch * 4);
            if (mo < p.M) *reinterpret_cast<f32x4*>(p.y + (size_t)mo * p.Co + ch * 4) = v;
        }
    }
}

bool conv_first_fwd_f32_applicable(const ConvDesc& d) { return d.Ci * d.KH * d.KW <= 32 && d.Co == 64 && d.Ci % 4 != 0; }

void conv_first_fwd_f32(const ConvDesc& d, const float* x, const float* w, const float* bias, float* y, bool relu, hipStream_t s) {
    SSD_REQUIRE(conv_first_fwd_f32_applicable(d), "first-layer kernel: Ci*taps <= 32 and Co == 64");
    SSD_REQUIRE((long long)d.B * d.Ho * d.Wo * d.Co < (1LL << 30) - 4 && (long long)d.B * d.Hi * d.Wi * d.Ci < (1LL << 30) - 4,
                "first-layer kernel: tensor exceeds 4 GiB");
    FirstF32Args a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.relu = relu;
    a.M = d.B * d.Ho * d.Wo; a.Hi = d.Hi; a.Wi = d.Wi; a.Ci = d.Ci; a.Ho = d.Ho; a.Wo = d.Wo; a.Co = d.Co;
    a.ntaps = d.KH * d.KW; a.stride = d.stride;
    for (int kh = 0; kh < d.KH; ++kh)
        for (int kw = 0; kw < d.KW; ++kw) {
            a.tap_dh[kh * d.KW + kw] = kh * d.dil - d.pad_h;
            a.tap_dw[kh * d.KW + kw] = kw * d.dil - d.pad_w;
        }
    a.ntiles = cdiv(a.M, 32);
    int blocks = cdiv(a.ntiles, 4);
    // the workgroups that are resident at once (three per CU at 168 registers): the prologue -- 32 filter and 32 bias loads per lane -- is
    // paid once per slot: 2048 workgroups 0.221 ms, 1024 0.266, 768 0.202, 512 0.223 (profiles/r06_ak_*; conv_first_bf16.hip has the story)
    static const int cap = env_int("SSD_FIRST_GRID_F32", 256 * 3);
    if (blocks > cap) blocks = cap;
    ProfScope prof("conv_first_fwd", conv_flops(d), 4.0 * ((double)d.B * d.Hi * d.Wi * d.Ci + (double)a.M * d.Co), s);
    hipLaunchKernelGGL(conv_first_fwd_f32_kernel<2>, dim3(blocks), dim3(256), 0, s, a);
    HIP_OK(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// weight gradient of the first layer, fp32:  dw[k][n] = sum_pixels xcol[pix][k] * dy[pix][n],  k = tap*Ci + c < 32.
// The layer is bound by reading dy once (64 channels x 4 B per pixel: 737 MB at batch 32).  No LDS staging: on
// v_mfma_f32_32x32x2_f32 the reduction index is the PIXEL (two per instruction), so every lane feeds the matrix
// core straight from two global loads per pixel pair -- one gathered image value (its fixed k, zero padding by an
// out-of-range buffer offset) and one 8-byte piece of dy (channels 2*li, 2*li+1: a wave reads 512 contiguous
// bytes; the two components feed two accumulators = the even / odd output channels) -- and sums those two dy values
// for the bias gradient.  Loads of the next two groups of pixel pairs are in flight while the current group multiplies.  Four waves of a workgroup add their accumulators through LDS
// and leave one slab; the fixed-order slab reduce (wgrad_reduce) finishes.  Exact fp32 products, fixed order.
// (The packed small-C path of the generic weight gradient ran this layer at 0.44 ms = 1.7 TB/s.)
// ---------------------------------------------------------------------------------
struct FirstWgradF32Args {
    const float* x;
    const float* dy;
    float* ws;
    int M, Hi, Wi, Ci, Ho, Wo, Co;
    int ntaps, stride, chunk;       // chunk: pixels per workgroup (multiple of 8)
    int tap_dh[9], tap_dw[9];
};

template <int U>
__global__ __launch_bounds__(256) void conv_first_wgrad_f32_kernel(FirstWgradF32Args p) {
    __shared__ float red[4][32][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int K = p.ntaps * p.Ci;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const bool kvalid = li < K;
    const int tp = kvalid ? li / p.Ci : 0, c = kvalid ? li - tp * p.Ci : 0;
    const int dh = p.tap_dh[tp], dw = p.tap_dw[tp];
    const unsigned koff = (unsigned)(((dh * p.Wi + dw) * p.Ci + c) * 4);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((size_t)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.Ci * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (unsigned)((size_t)p.M * p.Co * 4u), 0x00020000);

    const int per_wave = p.chunk >> 2;                              // even
    const int p0 = blockIdx.x * p.chunk + wave * per_wave;
    const int wg_end = min(p.M, (int)(blockIdx.x + 1) * p.chunk);
    const int p1 = min(wg_end, p0 + per_wave);

    // this lane's pixel walks p0 + lh, +2, +2, ...: (b, oh, ow) kept incrementally
    int pix = p0 + lh;
    int ow, oh, b;
    {
        const int mm = pix < p.M ? pix : 0;
        ow = mm % p.Wo;
        const int t2 = mm / p.Wo;
        oh = t2 % p.Ho;
        b = t2 / p.Ho;
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    float bs0 = 0.f, bs1 = 0.f;

    // ring of three groups of U pixel pairs: two groups of loads in flight while the third multiplies (the layer is
    // bound by memory-level parallelism: 16 resident waves per CU x 2 groups x 8 x 512 B of dy in flight)
    float av[3][U];
    f32x2 bv[3][U];
    auto load_group = [&](int buf) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool valid = pix < p1;
            const int sh = oh * p.stride + dh, sw = ow * p.stride + dw;
            const bool ok = valid && kvalid && (unsigned)sh < (unsigned)p.Hi && (unsigned)sw < (unsigned)p.Wi;
            const unsigned xo = (unsigned)((((b * p.Hi + oh * p.stride) * p.Wi + ow * p.stride) * p.Ci) * 4) + koff;
            // (nothing may CONSUME a loaded value in here: a select on it would wait for every load in turn)
            av[buf][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (int)(ok ? xo : OOB), 0, 0));
            const unsigned yo = (unsigned)((pix * p.Co + 2 * li) * 4);
            bv[buf][u] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(y_rsrc, (int)(valid ? yo : OOB), 0, 0));
            pix += 2;
            ow += 2;
#pragma unroll
            for (int w = 0; w < 2; ++w)
                if (ow >= p.Wo) {
                    ow -= p.Wo;
                    if (++oh >= p.Ho) { oh = 0; ++b; }
                }
        }
    };
    auto multiply = [&](int buf) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u], bv[buf][u][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u], bv[buf][u][1], acc1, 0, 0, 0);
            bs0 += bv[buf][u][0];          // bias gradient: this lane's two channels over its pixels (zeros past the end)
            bs1 += bv[buf][u][1];
        }
    };
    const int ngroups = (p1 - p0 + 2 * U - 1) / (2 * U);
    if (ngroups > 0) load_group(0);
    if (ngroups > 1) load_group(1);
    // the ring index is compile-time inside each of the three unrolled bodies (dynamic indexing would spill the ring)
    for (int g = 0; g < ngroups; g += 3) {
        if (g + 2 < ngroups) load_group(2);
        multiply(0);
        if (g + 1 < ngroups) {
            if (g + 3 < ngroups) load_group(0);
            multiply(1);
        }
        if (g + 2 < ngroups) {
            if (g + 4 < ngroups) load_group(1);
            multiply(2);
        }
    }
    // D rows = k index (r&3) + 8*(r>>2) + 4*lh, D column li = channel 2*li (acc0) / 2*li + 1 (acc1)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2) + 4 * lh;
        *reinterpret_cast<f32x2*>(&red[wave][k][2 * li]) = f32x2{acc0[r], acc1[r]};
    }
    __syncthreads();
    // rows K.. of the tile are unused by the filter: rows 30 / 31 carry the bias partial sums of the two pixel parities
    *reinterpret_cast<f32x2*>(&red[wave][30 + lh][2 * li]) = f32x2{bs0, bs1};
    __syncthreads();
    const size_t wcount = (size_t)K * p.Co;
    float* slab = p.ws + (size_t)blockIdx.x * (wcount + p.Co);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + 256 * q, k = e >> 6, n = e & 63;
        const float sum = (red[0][k][n] + red[1][k][n]) + (red[2][k][n] + red[3][k][n]);
        if (k < K) slab[(size_t)k * p.Co + n] = sum;
    }
    if (tid < 64) {
        float bsum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) bsum += red[w][30][tid] + red[w][31][tid];
        slab[wcount + tid] = bsum;
    }
}

bool conv_first_wgrad_f32_applicable(const ConvDesc& d) {
    return d.Ci * d.KH * d.KW <= 30 && d.Co == 64 && d.Ci % 4 != 0 && d.dil == 1 && d.Wo >= 1;
}

static int first_wgrad_f32_plan(const ConvDesc& d, int* chunk) {
    const int M = d.B * d.Ho * d.Wo;
    int nwg = cdiv(M, 1024);                   // >= 256 pixels per wave
    if (nwg > 1024) nwg = 1024;                // 4 workgroups per CU are resident: one round of the chip
    if (nwg < 1) nwg = 1;
    *chunk = cdiv(cdiv(M, nwg), 8) * 8;
    return cdiv(M, *chunk);
}

size_t conv_first_wgrad_f32_ws_floats(const ConvDesc& d) {
    int chunk;
    const int nwg = first_wgrad_f32_plan(d, &chunk);
    return (size_t)nwg * ((size_t)d.KH * d.KW * d.Ci * d.Co + d.Co);
}

void conv_first_wgrad_f32(const ConvDesc& d, const float* x, const float* dy, float* dw, float* dbias, const float* w,
                          float weight_decay, float* ws, hipStream_t s) {
    SSD_REQUIRE(conv_first_wgrad_f32_applicable(d), "first-layer weight gradient: Ci*taps <= 30, Co == 64, dilation 1");
    SSD_REQUIRE((long long)d.B * d.Ho * d.Wo * d.Co < (1LL << 30) - 4 && (long long)d.B * d.Hi * d.Wi * d.Ci < (1LL << 30) - 4,
                "first-layer kernel: tensor exceeds 4 GiB");
    FirstWgradF32Args a{};
    a.x = x; a.dy = dy; a.ws = ws;
    a.M = d.B * d.Ho * d.Wo; a.Hi = d.Hi; a.Wi = d.Wi; a.Ci = d.Ci; a.Ho = d.Ho; a.Wo = d.Wo; a.Co = d.Co;
    a.ntaps = d.KH * d.KW; a.stride = d.stride;
    for (int kh = 0; kh < d.KH; ++kh)
        for (int kw = 0; kw < d.KW; ++kw) {
            a.tap_dh[kh * d.KW + kw] = kh * d.dil - d.pad_h;
            a.tap_dw[kh * d.KW + kw] = kw * d.dil - d.pad_w;
        }
    const int nwg = first_wgrad_f32_plan(d, &a.chunk);
    {
        ProfScope prof("conv_first_wgrad", conv_flops(d), 4.0 * ((double)d.B * d.Hi * d.Wi * d.Ci + (double)a.M * d.Co), s);
        hipLaunchKernelGGL(conv_first_wgrad_f32_kernel<8>, dim3(nwg), dim3(256), 0, s, a);
        HIP_OK(hipGetLastError());
    }
    wgrad_reduce(ws, nwg, (size_t)a.ntaps * d.Ci * d.Co, d.Co, dw, dbias, w, weight_decay, s);
}

}  // namespace ssd
