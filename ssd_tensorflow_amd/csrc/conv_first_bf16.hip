// First layer of the bf16 configuration (conv1_1: 3 input channels, 3x3): the whole reduction
// k = tap*Ci + c fits one 32-deep block (27 of 32 used), so there is nothing to tile or pipeline --
// the layer is bound by writing (forward) / reading (weight gradient) the 64-channel tensor.
//   forward : every lane gathers the 16 image values of its pixel that its two MFMA operands need
//             straight from the fp32 image (buffer loads, zero padding by out-of-range offsets), the
//             filter sits in registers for the whole kernel, outputs leave through a per-wave LDS tile as
//             full 128-byte rows (16 bytes per lane).
//   wgrad   : dy tiles stream in by LDS-DMA and are consumed through ds_read_b64_tr_b16 exactly as in
//             conv_bf16.hip; the im2col image tile [64 pixels][32 k] is built in registers and written
//             next to it, so both MFMA operands come out of LDS pixel-major.
// Image and filter values are rounded to bf16 on the way in (0..255 pixel values are exact).
#include "conv.h"
#include "conv_detail.h"
#include "bf16.h"

namespace ssd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned OOBF = 0xFFFFFFF0u;

struct FirstArgs {
    const float* x;         // [B][Hi][Wi][Ci] fp32
    const float* w;         // [taps][Ci][Co] fp32 master
    const float* bias;
    bf16_t* y;              // forward: [M][Co] out;  wgrad: dy in
    float* ws;              // wgrad slabs
    int M, Hi, Wi, Ci, Ho, Wo, Co;
    int ntaps, stride, relu;
    int ntiles;             // forward: 32-pixel tiles
    int mchunk, nsplit;     // wgrad
    int tap_dh[9], tap_dw[9];
};

__device__ __forceinline__ u32x4 pack8(const float* v) {
    return u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
}

// ---------------------------------------------------------------------------------
// forward: one wave = 32 pixels x NT*32 channels per step, grid-stride over pixel tiles
// ---------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void conv_first_fwd_body(const FirstArgs& p, unsigned char* smem) {
    constexpr int ROWB = NT * 64 + 16;                      // LDS row: NT*32 channels bf16 + pad
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    unsigned char* T = smem + wave * 32 * ROWB;
    const int K = p.ntaps * p.Ci;

    // filter operand (MFMA rows = output channels): lane (co = nt*32 + li, k = ks*16 + 8*lh + j)
    bf16x8 wa[NT][2];
    float bv[NT][4][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // (all 8 loads first, from a clamped row: `k < K ? w[..] : 0` made each of the 32 loads of this prologue wait for
            // the one before -- 25 us per workgroup before its first tile)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = ks * 16 + 8 * lh + j;
                v[j] = p.w[(size_t)(k < K ? k : K - 1) * p.Co + nt * 32 + li];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (ks * 16 + 8 * lh + j >= K) v[j] = 0.f;
            wa[nt][ks] = __builtin_bit_cast(bf16x8, pack8(v));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[nt][g][e] = p.bias ? p.bias[nt * 32 + 8 * g + 4 * lh + e] : 0.f;
    }
    // image operand (MFMA columns = pixels): this lane's 16 k's are fixed -> byte offset from the pixel and tap bit
    int koff[2][8];
    unsigned kbit[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + 8 * lh + j;
            const bool kv = k < K;
            const int tp = kv ? k / p.Ci : 0, c = kv ? k - tp * p.Ci : 0;
            koff[ks][j] = ((p.tap_dh[tp] * p.Wi + p.tap_dw[tp]) * p.Ci + c) * 4;
            kbit[ks][j] = kv ? (1u << tp) : 0u;
        }
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
        bf16x8 xb[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned ok = 0u - (unsigned)((mk & kbit[ks][j]) != 0u);
                const unsigned off = ((base + (unsigned)koff[ks][j]) & ok) | (OOBF & ~ok);
                v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off, 0, 0));
            }
            xb[ks] = __builtin_bit_cast(bf16x8, pack8(v));
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[nt][0], xb[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[nt][1], xb[1], acc, 0, 0, 0);
            // D rows = channels (r&3) + 8*(r>>2) + 4*lh, D col = pixel li: 4 consecutive channels per register quad
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[4 * g + e] + bv[nt][g][e];
                    if (p.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                *reinterpret_cast<u32x2*>(T + li * ROWB + (nt * 32 + 8 * g + 4 * lh) * 2) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
            }
        }
        // wave-private tile -> global, 16 bytes per lane, whole rows (NT*64 bytes per pixel)
        constexpr int CPR = NT * 4;                  // 16-byte chunks per pixel row
        constexpr int PPI = 64 / CPR;                // pixels per store instruction
#pragma unroll
        for (int i = 0; i < 32 / PPI; ++i) {
            const int px = i * PPI + lane / CPR, ch = lane % CPR;
            const int mo = tile * 32 + px;
            const u32x4 v = *reinterpret_cast<const u32x4*>(T + px * ROWB + ch * 16);
            if (mo < p.M) *reinterpret_cast<u32x4*>(p.y + (size_t)mo * p.Co + ch * 8) = v;
        }
    }
}
// (A wave handles one 32-pixel tile at a time -- 16 gathers -> 4 MFMAs -> LDS -> 4 KB of stores, a dependent chain of ~4.6 us --
// so the rate follows the resident waves: 160 registers = 3 per SIMD.  Capped at 128 registers (4 per SIMD, 9 spilled)
// the kernel measured 0.135 instead of 0.132 ms: not kept, gpurun r03_u.)
template <int NT>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(FirstArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 32 * (NT * 64 + 16)];
    conv_first_fwd_body<NT>(p, smem);
}

// ---------------------------------------------------------------------------------
// forward, row-aligned tiles (round 6).  The kernel above turned out ISSUE-bound, not store-bound: ~400 VALU instructions per
// 32-pixel tile (two integer divisions for the pixel's row and image, nine tap tests, sixteen gathers each with two selects, 32
// accumulator reads + bias + relu + packing), ~64 us of pure issue per SIMD at batch 32, and more resident waves do not help
// (profiles/r06_w_*: 176 -> 87 registers, 0.134 -> 0.140 ms) while a kernel that only writes the same 368 MB takes 62-75 us
// (profiles/r06_x_store_rate_probe.txt).  Here a tile is 32 consecutive pixels of ONE image row:
//   * the tile's image, row and column segment are wave-uniform: decoded with two scalar multiply-highs (host-made magic
//     numbers), they go into the gathers' SCALAR offset; a lane's sixteen offsets (pixel li x stride, tap, channel) are constants;
//   * a tile whose 32 pixels have all taps inside the image (all but the first / last row and column segment) gathers with no
//     vector instruction at all; the others select the out-of-range offset per gather from a tap-validity word (LDS table);
//   * the bias is the MFMA's C operand (registers that live for the whole kernel): no zero fill, no add -- the sum starts from
//     the bias instead of ending with it, a different fp32 rounding order than the kernel above (which the tests keep as the
//     reference form of this layer, SSD_FIRST_ROWS_BF16=0);
//   * relu on the packed bf16 pairs (v_pk_max_i16 against zero: a negative bf16 is a negative int16).
// ---------------------------------------------------------------------------------
struct FirstRowsArgs {
    FirstArgs a;
    int segs;                   // 32-pixel segments per output row
    unsigned magic_segs, magic_ho;      // ceil(2^32 / segs), ceil(2^32 / Ho)
    int nrowtiles;              // B * Ho * segs
};
#ifndef FIRST_AHEAD
#define FIRST_AHEAD 1
#endif
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int cdiv_i(int a, int b) { return (a + b - 1) / b; }

template <int NT>
__global__ __launch_bounds__(256) void conv_first_fwd_rows_kernel(FirstRowsArgs pp) {
    constexpr int ROWB = NT * 64 + 16;                      // LDS row: NT*32 channels bf16 + pad
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 32 * ROWB + 512 * 4];
    const FirstArgs& p = pp.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    unsigned char* T = smem + wave * 32 * ROWB;
    unsigned* lut = reinterpret_cast<unsigned*>(smem + 4 * 32 * ROWB);      // [1 << ntaps]: tap mask of a pixel -> mask of its valid k's
    const int K = p.ntaps * p.Ci;
    for (int e = tid; e < (1 << p.ntaps); e += 256) {
        unsigned v = 0;
        for (int k = 0; k < K; ++k) v |= (((unsigned)e >> (k / p.Ci)) & 1u) << k;
        lut[e] = v;
    }
    __syncthreads();

    // filter operand (MFMA rows = output channels): lane (co = nt*32 + li, k = ks*16 + 8*lh + j); bias = the accumulator's start
    bf16x8 wa[NT][2];
    f32x16 bias16[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = ks * 16 + 8 * lh + j;
                v[j] = p.w[(size_t)(k < K ? k : K - 1) * p.Co + nt * 32 + li];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (ks * 16 + 8 * lh + j >= K) v[j] = 0.f;
            wa[nt][ks] = __builtin_bit_cast(bf16x8, pack8(v));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) bias16[nt][r] = p.bias ? p.bias[nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] : 0.f;
    }
    // this lane's sixteen gather offsets relative to the tile's first pixel (constants), and its tap / k bookkeeping
    int tap_min = 0;
    for (int t = 0; t < p.ntaps; ++t) tap_min = min(tap_min, (p.tap_dh[t] * p.Wi + p.tap_dw[t]) * p.Ci);
    const unsigned tap_bias = (unsigned)(-tap_min) * 4u;
    unsigned koff[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + 8 * lh + j;
            const bool kv = k < K;
            const int tp = kv ? k / p.Ci : 0, c = kv ? k - tp * p.Ci : 0;
            koff[ks][j] = kv ? (unsigned)(((p.tap_dh[tp] * p.Wi + p.tap_dw[tp] + li * p.stride) * p.Ci + c) * 4 + (int)tap_bias) : OOBF;
        }
    const size_t x_bytes = (size_t)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.Ci * 4u;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(const_cast<float*>(p.x)) - tap_bias, 0, (unsigned)(x_bytes + tap_bias), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (unsigned)((size_t)p.M * p.Co * 2u), 0x00020000);
    // column range of the taps (for the interior test)
    int dw_min = 0, dw_max = 0, dh_min = 0, dh_max = 0;
    for (int t = 0; t < p.ntaps; ++t) {
        dw_min = min(dw_min, p.tap_dw[t]); dw_max = max(dw_max, p.tap_dw[t]);
        dh_min = min(dh_min, p.tap_dh[t]); dh_max = max(dh_max, p.tap_dh[t]);
    }

    // (b, oh, seg) of a tile: wave-uniform (a divisor of 1 has no 32-bit magic number: its quotient is the dividend)
    struct TileGeom { int row, ow0, npx; unsigned base; bool interior; int h0, w00; };
    auto geom = [&](int tile) {
        TileGeom g;
        g.row = pp.segs == 1 ? tile : (int)__builtin_amdgcn_readfirstlane((int)__umulhi((unsigned)tile, pp.magic_segs));      // = b * Ho + oh
        const int seg = tile - g.row * pp.segs;
        const int b = p.Ho == 1 ? g.row : (int)__builtin_amdgcn_readfirstlane((int)__umulhi((unsigned)g.row, pp.magic_ho));
        const int oh = g.row - b * p.Ho;
        g.ow0 = seg * 32;
        g.h0 = oh * p.stride; g.w00 = g.ow0 * p.stride;
        g.npx = min(32, p.Wo - g.ow0);                                // pixels of this tile inside the row
        g.base = (unsigned)(((b * p.Hi + g.h0) * p.Wi + g.w00) * p.Ci) * 4u;      // scalar offset of the tile's first pixel
        g.interior = g.npx == 32 && g.h0 + dh_min >= 0 && g.h0 + dh_max < p.Hi && g.w00 + dw_min >= 0 && g.w00 + 31 * p.stride + dw_max < p.Wi;
        return g;
    };
    // the sixteen image values of this lane's two MFMA operands: issued, not yet looked at
    auto gather = [&](const TileGeom& g, float (&v)[2][8]) {
        if (g.interior) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[ks][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, koff[ks][j], g.base, 0));
        } else {
            const int w0 = g.w00 + li * p.stride;
            unsigned mk = 0;
            for (int t = 0; t < p.ntaps; ++t) {
                const int sh = g.h0 + p.tap_dh[t], sw = w0 + p.tap_dw[t];
                if ((unsigned)sh < (unsigned)p.Hi && (unsigned)sw < (unsigned)p.Wi) mk |= 1u << t;
            }
            if (li >= g.npx) mk = 0;
            const unsigned kv = lut[mk] >> (8 * lh);            // bit ks*16 + j: this lane's k = ks*16 + 8*lh + j is inside the image
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned ok = 0u - ((kv >> (ks * 16 + j)) & 1u);
                    v[ks][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (koff[ks][j] & ok) | (OOBF & ~ok), g.base, 0));
                }
        }
    };

    // One tile of look-ahead: the next tile's gathers are issued before this tile's results are stored (stores count in vmcnt like
    // loads and retire in order with them: issued BEHIND the previous tile's stores, every gather waited for four store
    // acknowledgements first).  What a tile then still costs a wave -- 2.8 us at two waves per SIMD, 0.122 ms for the layer against the
    // 0.065-0.075 ms a pure store stream needs -- is the gathers' own return time under 368 MB of write traffic.  Looking further ahead
    // (FIRST_AHEAD 2: 0.121 ms, 3: 0.122, 4: 0.123, profiles/r06_af_*) and handing the stores to waves of their own (four gather + four
    // store waves behind one raw barrier per tile: 0.131 ms, profiles/r06_ae_*) changed nothing, and the deeper rings failed the
    // 300 x 300 case of tests/test_gpu_bf16.py: only the depth-1 form is built.
    constexpr int AHEAD = FIRST_AHEAD;
    const int nwaves = gridDim.x * 4;
    const int tile0 = blockIdx.x * 4 + wave;
    // (every wave runs whole rounds of AHEAD tiles and every tile -- also the null tiles past the end, whose gathers and stores all
    // carry the out-of-range offset -- issues exactly 16 loads and 4 stores: the waits below are COUNTED ones the compiler can prove)
    const int nround = tile0 < pp.nrowtiles ? cdiv_i(cdiv_i(pp.nrowtiles - tile0, nwaves), AHEAD) : 0;
    auto geom_or_null = [&](int tile) {
        if (tile < pp.nrowtiles) return geom(tile);
        TileGeom g{};
        g.npx = 0; g.interior = false;
        return g;
    };
    float raw[AHEAD][2][8];
    TileGeom gq[AHEAD];
    if (nround > 0) {
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) { gq[a] = geom_or_null(tile0 + a * nwaves); gather(gq[a], raw[a]); }
    }
    for (int r = 0; r < nround; ++r) {
#pragma unroll
      for (int a = 0; a < AHEAD; ++a) {
        const int tile = tile0 + (r * AHEAD + a) * nwaves;
        bf16x8 xb[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xb[ks] = __builtin_bit_cast(bf16x8, pack8(raw[a][ks]));
        const TileGeom gc = gq[a];
        gq[a] = geom_or_null(tile + AHEAD * nwaves);
        gather(gq[a], raw[a]);
        const int row = gc.row, ow0 = gc.ow0, npx = gc.npx;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[nt][0], xb[0], bias16[nt], 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[nt][1], xb[1], acc, 0, 0, 0);
            // D rows = channels (r&3) + 8*(r>>2) + 4*lh, D col = pixel li: 4 consecutive channels per register quad
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned q0 = pack2(acc[4 * g], acc[4 * g + 1]), q1 = pack2(acc[4 * g + 2], acc[4 * g + 3]);
                if (p.relu) {
                    q0 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, q0), s16x2{0, 0}));
                    q1 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, q1), s16x2{0, 0}));
                }
                *reinterpret_cast<u32x2*>(T + li * ROWB + (nt * 32 + 8 * g + 4 * lh) * 2) = u32x2{q0, q1};
            }
        }
        // wave-private tile -> global, 16 bytes per lane, whole rows (NT*64 bytes per pixel)
        constexpr int CPR = NT * 4;                  // 16-byte chunks per pixel row
        constexpr int PPI = 64 / CPR;                // pixels per store instruction
        // (buffer stores, an out-of-range offset for the pixels past the row's end: ALWAYS four store instructions per tile, so that the
        // wait for the next tile's gathers can be a counted one -- vmcnt(4) -- instead of vmcnt(0))
        const unsigned ybase = (unsigned)(((size_t)row * p.Wo + ow0) * p.Co * 2u);
#pragma unroll
        for (int i = 0; i < 32 / PPI; ++i) {
            const int px = i * PPI + lane / CPR, ch = lane % CPR;
            const u32x4 v = *reinterpret_cast<const u32x4*>(T + px * ROWB + ch * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, y_rsrc, px < npx ? (unsigned)((px * p.Co + ch * 8) * 2) : OOBF, ybase, 0);
        }
      }
    }
}

// ---------------------------------------------------------------------------------
// weight gradient: dW[k][n] = sum_m xcol[m][k] * dy[m][n], k = tap*Ci + c < 32
// workgroup = one pixel split; waves (kh, nh): pixel half kh of every 64-pixel block, channel tile nh
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(FirstArgs p) {
    constexpr int BP = 64, XROWB = 64, YROWB = 128;         // LDS rows: 32 k bf16; 64 channels bf16
    constexpr int X_LDS = BP * XROWB, STAGE = BP * (XROWB + YROWB);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int split = blockIdx.x;
    const int mbeg = split * p.mchunk;
    const int mend = min(p.M, mbeg + p.mchunk);
    const int niter = (mend - mbeg + BP - 1) / BP;
    const int K = p.ntaps * p.Ci;

    // im2col staging: thread -> pixel row tid>>2, k = (tid&3)*8 .. +7
    const int xr = tid >> 2, xq = tid & 3;
    int koff[8];
    unsigned kbit[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = xq * 8 + j;
        const bool kv = k < K;
        const int tp = kv ? k / p.Ci : 0, c = kv ? k - tp * p.Ci : 0;
        koff[j] = ((p.tap_dh[tp] * p.Wi + p.tap_dw[tp]) * p.Ci + c) * 4;
        kbit[j] = kv ? (1u << tp) : 0u;
    }
    // dy staging (LDS-DMA): 128-byte rows, slot p of row r holds chunk p ^ 4*((r>>1)&1)   (conv_bf16.hip)
    const int yr = tid >> 3, ys = tid & 7;
    const int ychunk = ys ^ (((yr >> 1) & 1) * 4);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((size_t)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.Ci * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (unsigned)((size_t)p.M * p.Co * 2u), 0x00020000);

    auto stage_tile = [&](int it, int stage) {
        unsigned char* S = smem + stage * STAGE;
        const int mb = mbeg + it * BP;
        {   // x: 8 gathered image values -> one 16-byte LDS store
            const int m = mb + xr;
            const int mm = m < mend ? m : 0;
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
            if (m >= mend) mk = 0;
            const unsigned base = (unsigned)(((b * p.Hi + h0) * p.Wi + w0) * p.Ci) * 4u;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned ok = 0u - (unsigned)((mk & kbit[j]) != 0u);
                v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, ((base + (unsigned)koff[j]) & ok) | (OOBF & ~ok), 0, 0));
            }
            *reinterpret_cast<u32x4*>(S + xr * XROWB + xq * 16) = pack8(v);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {      // dy: 32 pixel rows per pass
            const int m = mb + yr + j * 32;
            const unsigned mk = 0u - (unsigned)(m < mend);
            const unsigned off = (unsigned)((m * p.Co + ychunk * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, LDS_PTR(S + X_LDS + wave * 1024 + j * 4096), 16, (off & mk) | (OOBF & ~mk), 0, 0, 0);
        }
    };

    f32x16 acc, accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = accb[r] = 0.f;
    s16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (short)0x3F80;       // bf16 1.0
    const int kh = wave >> 1, nh = wave & 1;
    const int lh = lane >> 5, q = lane & 15, cb = (lane >> 4) & 1;
    const int prow = kh * 32 + lh * 8 + (q >> 2);
    // x rows are 64 bytes: four consecutive pixel rows tile one 256-byte bank row, no swizzle needed
    const int xa = prow * XROWB + (cb * 2 + ((q >> 1) & 1)) * 16 + (q & 1) * 8;
    const int ych = nh * 4 + cb * 2 + ((q >> 1) & 1);
    const int ya = X_LDS + prow * YROWB + ((ych ^ (((prow >> 1) & 1) * 4)) * 16) + (q & 1) * 8;

    auto tr8 = [&](const unsigned char* S, int off, int rowb) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)LDS_PTR(S + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)LDS_PTR(S + off + 4 * rowb));
        return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    if (niter > 0) stage_tile(0, 0);
    for (int it = 0; it < niter; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (it + 1 < niter) stage_tile(it + 1, (it + 1) & 1);
        const unsigned char* S = smem + (it & 1) * STAGE;
#pragma unroll
        for (int st = 0; st < 2; ++st) {       // this wave's 32 pixels = two 16-pixel MFMA steps
            const s16x8 a = tr8(S, xa + st * 16 * XROWB, XROWB);
            const s16x8 bq = tr8(S, ya + st * 16 * YROWB, YROWB);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq), acc, 0, 0, 0);
            // bias gradient on the matrix cores: every row of ones^T dy is the column sum over this wave's 32 pixels (a
            // scalar LDS sweep of the tile by wave 0 made that wave every tile's straggler)
            accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, bq), accb, 0, 0, 0);
        }
    }
    // two partial slabs per split (pixel halves): slab index = 2*split + kh; the fixed-order reduce adds them
    const size_t wcount = (size_t)K * p.Co;
    float* slab = p.ws + (size_t)(2 * split + kh) * (wcount + p.Co);
    const int li = lane & 31;
    const int n = nh * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (k < K) slab[(size_t)k * p.Co + n] = acc[r];
    }
    if (lh == 0) slab[wcount + n] = accb[0];      // row 0 of ones^T dy: this wave's pixel half, its 32 channels
}

// ---------------------------------------------------------------------------------
static void fill_args(FirstArgs& a, const ConvDesc& d) {
    SSD_REQUIRE(d.Ci * d.KH * d.KW <= 32, "first-layer kernel: Ci*taps must be <= 32 (got %d)", d.Ci * d.KH * d.KW);
    SSD_REQUIRE(d.Co == 64, "first-layer kernel: Co must be 64 (got %d)", d.Co);
    SSD_REQUIRE((long long)d.B * d.Ho * d.Wo * d.Co < (1LL << 30), "first-layer kernel: tensor too large");
    a.M = d.B * d.Ho * d.Wo; a.Hi = d.Hi; a.Wi = d.Wi; a.Ci = d.Ci; a.Ho = d.Ho; a.Wo = d.Wo; a.Co = d.Co;
    a.ntaps = d.KH * d.KW; a.stride = d.stride;
    for (int kh = 0; kh < d.KH; ++kh)
        for (int kw = 0; kw < d.KW; ++kw) {
            a.tap_dh[kh * d.KW + kw] = kh * d.dil - d.pad_h;
            a.tap_dw[kh * d.KW + kw] = kw * d.dil - d.pad_w;
        }
}

void conv_first_fwd_bf16(const ConvDesc& d, const float* x, const float* w, const float* bias, bf16_t* y, bool relu, hipStream_t s) {
    FirstArgs a{};
    fill_args(a, d);
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.relu = relu;
    static const int rows = env_int("SSD_FIRST_ROWS_BF16", 1);      // 0: the pixel-raster kernel (bias added behind the sum: the tests' reference form)
    const int segs = cdiv(d.Wo, 32);
    const long long nrowtiles = (long long)d.B * d.Ho * segs;
    if (rows && nrowtiles * segs < (1LL << 32) && (long long)d.B * d.Ho * d.Ho < (1LL << 32)) {
        FirstRowsArgs r{};
        r.a = a; r.segs = segs; r.nrowtiles = (int)nrowtiles;
        r.magic_segs = (unsigned)(((1ULL << 32) + segs - 1) / segs);
        r.magic_ho = (unsigned)(((1ULL << 32) + d.Ho - 1) / d.Ho);
        // exactly the workgroups that are resident at once (two per CU at 179 registers): a workgroup's prologue -- the tap table, 32
        // filter and 32 bias loads per lane -- is paid once per slot; 2048 workgroups: 0.150 ms, 1024: 0.135, 512: 0.122, 256: 0.155
        // (profiles/r06_ad_*)
        int blocks = cdiv((int)nrowtiles, 4);
        static const int cap = env_int("SSD_FIRST_GRID", 256 * 2);
        if (blocks > cap) blocks = cap;
        ProfScope prof("conv_first_fwd_rows_bf16", conv_flops(d), 4.0 * d.B * d.Hi * d.Wi * d.Ci + 2.0 * a.M * d.Co, s);
        hipLaunchKernelGGL(conv_first_fwd_rows_kernel<2>, dim3(blocks), dim3(256), 0, s, r);
        HIP_OK(hipGetLastError());
        return;
    }
    a.ntiles = cdiv(a.M, 32);
    int blocks = cdiv(a.ntiles, 4);
    if (blocks > 256 * 8) blocks = 256 * 8;
    ProfScope prof("conv_first_fwd_bf16", conv_flops(d), 4.0 * d.B * d.Hi * d.Wi * d.Ci + 2.0 * a.M * d.Co, s);
    hipLaunchKernelGGL(conv_first_fwd_kernel<2>, dim3(blocks), dim3(256), 0, s, a);
    HIP_OK(hipGetLastError());
}

static int first_wgrad_splits(const ConvDesc& d, int* mchunk) {
    const int M = d.B * d.Ho * d.Wo;
    int ns = cdiv(M, 64 * 24);                 // >= 24 iterations per workgroup
    // Bytes in flight set this kernel's rate (one 8-KB dy tile per workgroup: 1024 x 8 KB / ~3 us of loaded round trip =
    // 2.8 TB/s, measured 2.5): 88 registers let five workgroups share a CU -> 1280 (1024 = round 2).
    // Round 3, per-layer events of the step: 0.159 ms -> 0.136 with the bias gradient on the matrix cores -> 0.129 with
    // 1280 workgroups; 1536 (a second, partial round) gains nothing.
    constexpr int cap = 1280;
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
    *mchunk = cdiv(cdiv(M, ns), 64) * 64;
    return cdiv(M, *mchunk);
}

size_t conv_first_wgrad_bf16_ws_floats(const ConvDesc& d) {
    int mchunk;
    const int ns = first_wgrad_splits(d, &mchunk);
    return (size_t)2 * ns * ((size_t)d.KH * d.KW * d.Ci * d.Co + d.Co);
}

void conv_first_wgrad_bf16(const ConvDesc& d, const float* x, const bf16_t* dy, float* dw, float* dbias, const float* w,
                           float weight_decay, float* ws, hipStream_t s) {
    FirstArgs a{};
    fill_args(a, d);
    a.x = x; a.y = const_cast<bf16_t*>(dy); a.ws = ws;
    a.nsplit = first_wgrad_splits(d, &a.mchunk);
    {
        ProfScope prof("conv_first_wgrad_bf16", conv_flops(d), 4.0 * d.B * d.Hi * d.Wi * d.Ci + 2.0 * a.M * d.Co, s);
        hipLaunchKernelGGL(conv_first_wgrad_kernel, dim3(a.nsplit), dim3(256), 0, s, a);
        HIP_OK(hipGetLastError());
    }
    wgrad_reduce(ws, 2 * a.nsplit, (size_t)a.ntaps * d.Ci * d.Co, d.Co, dw, dbias, w, weight_decay, s);
}

}  // namespace ssd
