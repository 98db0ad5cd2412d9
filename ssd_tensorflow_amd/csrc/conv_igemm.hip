// im2col-free direct convolution for gfx950 (MI355X), fp32 in / fp32 accumulate on
// the matrix cores (v_mfma_f32_32x32x2_f32, exact f32 == an fmaf chain).
//
// Forward and data-gradient are ONE gather-GEMM kernel:
//     dst[m][n] = sum_{tap, c} src[pix(m, tap)][c] * Wt(tap, c, n)
// with m = (b, oh, ow) the NHWC pixel index of dst, and the source pixel
// pix = ((oh*mul + dh[tap]) / div, (ow*mul + dw[tap]) / div), zero outside the image.
//   forward : src = x,  mul = stride, div = 1,      dh = kh*dil - pad, Wt = W[tap][c][n]
//   dgrad   : src = dy, mul = 1,      div = stride, dh = pad - kh*dil, Wt = W[tap][n][c]
// The weight gradient is a second kernel (reduction over pixels, split-M slabs).
//
// Tiling (per 256-thread workgroup = 4 wave64): block tile (32*TM*WM) x (32*TN*WN),
// BK = 32 channels of one tap per k-iteration, iteration order channel-chunk outer /
// tap inner so the 9 taps re-read the same pixels from L1/L2.  LDS tiles:
//   A  [BM][36]   (k contiguous, +4 pad): ds_read_b128 gives each lane 4 consecutive
//                 k of its row, conflict-free (36*i mod 64 distinct over 16 rows);
//   B  fwd  [32][BN]  read as ds_read_b32 (lanes consecutive in n),
//      dgrad[BN][36]  read as ds_read_b128 like A (the HWIO filter is consumed
//                     transposed straight from its canonical layout, no repack).
// The MFMA k-index is permuted (lane-half h owns k = 4h..4h+3 of each group of 8) so
// one b128 read feeds 4 MFMAs; a permutation of the reduction index is harmless as
// long as A and B use the same one.
// Global->LDS goes through registers (the gather needs zero fill), issued one
// k-iteration ahead of the MFMAs that consume it (two LDS buffers, one barrier / iter).
#include "conv.h"
#include "conv_detail.h"
#include "bf16.h"

namespace ssd {

constexpr int BK = 32;
constexpr int LDA = 36;
enum { MODE_FWD = 0, MODE_DGRAD = 1 };

struct GatherArgs {
    const float* src;
    const float* wgt;
    const float* bias;
    const float* mask;
    float* dst;
    int M, DH, DW, DN;
    int SH, SW, SC;
    int ntaps, mul, div;
    int wci, wco;
    int relu, accum;
    int NT;                 // number of n tiles
    int tap_dh[9], tap_dw[9];
    // parity classes of a strided data gradient (LDS-DMA kernel only): the gathered taps are a subset of the filter's
    // (tap_w = filter tap of gathered tap k, wtaps = taps of the filter), and the M "virtual" pixels (b, a, c) of the
    // class are the real pixels (b, out_mul a + out_ph, out_mul c + out_pw) of an ODH x ODW image.  out_mul = 1: off.
    int tap_w[9], wtaps;
    int out_mul, out_ph, out_pw, ODH, ODW;
    // all classes in ONE launch (the classes are latency-bound on their own: a few workgroups each): class c owns the
    // workgroups [cls_wg0[c], cls_wg0[c + 1]) and replaces M / DH / DW / ntaps / taps / out_ph / out_pw by its own
    int nclass;
    int cls_wg0[5], cls_M[4], cls_DH[4], cls_DW[4], cls_ntaps[4], cls_ph[4], cls_pw[4];
    int cls_dh[4][9], cls_dw[4][9], cls_w[4][9];
    // fused 2x2 pool (LDS-DMA kernel only, see there): forward -> pooled tensor [PB][PH][PW][DN] (+ record); data gradient ->
    // through the record into the pool's input gradient [.][UH][UW][DN]
    float* pool_dst;
    unsigned short* pool_rec;
    int PB, PH, PW;
    const unsigned short* unpool_rec;
    int UH, UW;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// OBF: the output tensor is bf16 (conv1_1 of the bf16 configuration: fp32 image and filter in, bf16 out)
template <int MODE, int WM, int WN, int TM, int TN, bool SMALLC, bool STRIDED, bool OBF = false>
__global__ __launch_bounds__(256) void conv_gather_kernel(GatherArgs p) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int A_ROWS = BM / 32;                   // rows staged per thread
    constexpr int B_CPR = BN / 4;                     // fwd: float4 per k-row
    constexpr int B_RPP = 256 / B_CPR;                // fwd: k-rows per pass
    constexpr int B_FWD_N = (BK + B_RPP - 1) / B_RPP; // fwd: passes
    constexpr int B_DG_N = BN / 32;                   // dgrad: rows per thread
    constexpr int B_N = MODE == MODE_FWD ? B_FWD_N : B_DG_N;
    constexpr int A_LDS = BM * LDA;
    constexpr int B_LDS = MODE == MODE_FWD ? BK * BN : BN * LDA;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(B_RPP <= BK, "tile too narrow");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = wg / p.NT, nt = wg - mt * p.NT;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-thread staging rows of the A (pixel gather) tile ------------------
    const int a_c4 = (tid & 7) * 4;
    int rb[A_ROWS], rh[A_ROWS], rw[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int m = m0 + (tid >> 3) + 32 * i;
        const int mm = m < p.M ? m : 0;
        const int ow = mm % p.DW;
        const int t2 = mm / p.DW;
        const int oh = t2 % p.DH;
        const int b = t2 / p.DH;
        rb[i] = b * p.SH * p.SW;
        rh[i] = m < p.M ? oh * p.mul : -(1 << 20);
        rw[i] = ow * p.mul;
    }

    const int nchunks = SMALLC ? 1 : (p.SC + BK - 1) / BK;
    const int nk = SMALLC ? 1 : nchunks * p.ntaps;

    f32x4 areg[A_ROWS];
    f32x4 breg[B_N];

    // Fast path (every layer but conv1_1 and the strided data-gradients): the gather is a
    // branch-free buffer load.  Per row: the byte offset of (pixel, channel a_c4) and a 9-bit
    // mask of the taps that fall inside the image, both computed once; per iteration one add of
    // the (wave-uniform) tap/chunk offset and one select to an out-of-range offset, for which the
    // buffer unit returns zeros -- the zero padding costs no branch and no VALU select of data.
    constexpr bool FAST = !SMALLC && !STRIDED;
    const __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.src), 0, (FAST || SMALLC) ? (unsigned)((size_t)(p.M / (p.DH * p.DW)) * p.SH * p.SW * p.SC * 4u) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wgt), 0, (unsigned)((size_t)p.ntaps * p.wci * p.wco * 4u), 0x00020000);
    unsigned a_off[A_ROWS], a_msk[A_ROWS];
    // conv1_1 (Ci = 3): k = tap*Ci + c packed into ONE k-iteration.  This thread's four k's are the
    // same for all of its rows: their element offset from the row's pixel, tap index and validity
    // are constants; the per-row part is the same (offset, tap mask) pair as on the fast path.
    int sk_off[4];
    unsigned sk_bit[4];
    if constexpr (SMALLC) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = a_c4 + e;
            const bool kv = k < p.ntaps * p.SC;
            const int tp = kv ? k / p.SC : 0, c = kv ? k - tp * p.SC : 0;
            sk_off[e] = ((p.tap_dh[tp] * p.SW + p.tap_dw[tp]) * p.SC + c) * 4;
            sk_bit[e] = kv ? (1u << tp) : 0u;
        }
    }
    if constexpr (FAST || SMALLC) {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            a_off[i] = (unsigned)((rb[i] + rh[i] * p.SW + rw[i]) * p.SC + (SMALLC ? 0 : a_c4)) * 4u;
            unsigned mk = 0;
            for (int t = 0; t < p.ntaps; ++t) {
                const int sh = rh[i] + p.tap_dh[t], sw = rw[i] + p.tap_dw[t];
                if ((unsigned)sh < (unsigned)p.SH && (unsigned)sw < (unsigned)p.SW) mk |= 1u << t;
            }
            a_msk[i] = mk;
        }
    }
    constexpr unsigned OOB = 0xFFFFFFF0u;

    auto load_tiles = [&](int kiter) {
        const int cc = kiter / p.ntaps;
        const int tap = kiter - cc * p.ntaps;
        if constexpr (FAST) {
            const unsigned toff = (unsigned)(((p.tap_dh[tap] * p.SW + p.tap_dw[tap]) * p.SC + cc * BK) * 4);   // wave-uniform
            // validity as all-ones / all-zeros words, combined with bit ops: hipcc turns a
            // `cond ? offset : OOB` feeding a load into divergent branches around two loads
            const unsigned cmask = 0u - (unsigned)(cc * BK + a_c4 < p.SC);
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                const unsigned m = (0u - ((a_msk[i] >> tap) & 1u)) & cmask;
                const unsigned off = ((a_off[i] + toff) & m) | (OOB & ~m);
                areg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src_rsrc, off, 0, 0));
            }
            if constexpr (MODE == MODE_FWD) {
                const int col = (tid % B_CPR) * 4;
                const unsigned nmask = 0u - (unsigned)(n0 + col < p.DN);
#pragma unroll
                for (int i = 0; i < B_FWD_N; ++i) {
                    const int kr = tid / B_CPR + B_RPP * i;
                    const int c = cc * BK + kr;
                    const unsigned m = nmask & (0u - (unsigned)(c < p.SC && kr < BK));
                    const unsigned off = (((unsigned)((tap * p.wci + c) * p.wco + n0 + col) * 4u) & m) | (OOB & ~m);
                    breg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wgt_rsrc, off, 0, 0));
                }
            } else {
                const int c = cc * BK + a_c4;
                const unsigned cmk = 0u - (unsigned)(c < p.SC);
#pragma unroll
                for (int i = 0; i < B_DG_N; ++i) {
                    const int n = n0 + (tid >> 3) + 32 * i;
                    const unsigned m = cmk & (0u - (unsigned)(n < p.DN));
                    const unsigned off = (((unsigned)((tap * p.wci + n) * p.wco + c) * 4u) & m) | (OOB & ~m);
                    breg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wgt_rsrc, off, 0, 0));
                }
            }
            return;
        }
        if constexpr (SMALLC) {
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned m = 0u - (unsigned)((a_msk[i] & sk_bit[e]) != 0u);
                    const unsigned off = ((a_off[i] + (unsigned)sk_off[e]) & m) | (OOB & ~m);
                    v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src_rsrc, off, 0, 0));
                }
                areg[i] = v;
            }
        } else {
            const int dh = p.tap_dh[tap], dw = p.tap_dw[tap];
            const int c = cc * BK + a_c4;
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                int sh = rh[i] + dh, sw = rw[i] + dw;
                bool ok = c < p.SC;
                if constexpr (STRIDED) {
                    ok = ok && (sh % p.div == 0) && (sw % p.div == 0);
                    sh /= p.div;
                    sw /= p.div;
                }
                ok = ok && (unsigned)sh < (unsigned)p.SH && (unsigned)sw < (unsigned)p.SW;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = ld4(p.src + (size_t)(rb[i] + sh * p.SW + sw) * p.SC + c);
                areg[i] = v;
            }
        }
        if constexpr (MODE == MODE_FWD) {
            const int col = (tid % B_CPR) * 4;
#pragma unroll
            for (int i = 0; i < B_FWD_N; ++i) {
                const int kr = tid / B_CPR + B_RPP * i;
                const int c = cc * BK + kr;            // SMALLC: cc == 0, c is the packed k
                const bool ok = (SMALLC ? c < p.ntaps * p.SC : c < p.SC) && (n0 + col) < p.DN && kr < BK;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = ld4(p.wgt + (size_t)((SMALLC ? 0 : tap * p.wci) + c) * p.wco + n0 + col);
                breg[i] = v;
            }
        } else {
            const int c = cc * BK + a_c4;              // k index = dy channel (filter's Co axis)
#pragma unroll
            for (int i = 0; i < B_DG_N; ++i) {
                const int n = n0 + (tid >> 3) + 32 * i;   // n index = dx channel (filter's Ci axis)
                const bool ok = n < p.DN && c < p.SC;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = ld4(p.wgt + (size_t)(tap * p.wci + n) * p.wco + c);
                breg[i] = v;
            }
        }
    };

    auto store_tiles = [&](int buf) {
        float* As = smem + buf * (A_LDS + B_LDS);
        float* Bs = As + A_LDS;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i)
            *reinterpret_cast<f32x4*>(As + ((tid >> 3) + 32 * i) * LDA + a_c4) = areg[i];
        if constexpr (MODE == MODE_FWD) {
            const int col = (tid % B_CPR) * 4;
#pragma unroll
            for (int i = 0; i < B_FWD_N; ++i) {
                const int kr = tid / B_CPR + B_RPP * i;
                if (kr < BK) *reinterpret_cast<f32x4*>(Bs + kr * BN + col) = breg[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_DG_N; ++i)
                *reinterpret_cast<f32x4*>(Bs + ((tid >> 3) + 32 * i) * LDA + a_c4) = breg[i];
        }
    };

    // ---- accumulators ---------------------------------------------------------
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, lh = lane >> 5;

    auto compute = [&](int buf) {
        const float* As = smem + buf * (A_LDS + B_LDS);
        const float* Bs = As + A_LDS;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            const int kb = g * 8 + lh * 4;
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                a[mi] = *reinterpret_cast<const f32x4*>(As + (wm * 32 * TM + mi * 32 + li) * LDA + kb);
            if constexpr (MODE == MODE_FWD) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int t = 0; t < 4; ++t) b[ni][t] = Bs[(kb + t) * BN + wn * 32 * TN + ni * 32 + li];
            } else {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    b[ni] = *reinterpret_cast<const f32x4*>(Bs + (wn * 32 * TN + ni * 32 + li) * LDA + kb);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t], b[ni][t], acc[mi][ni], 0, 0, 0);
        }
    };

    // ---- main loop --------------------------------------------------------------
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int k = 0; k < nk; ++k) {
        const bool more = k + 1 < nk;
        if (more) load_tiles(k + 1);
        compute(k & 1);
        if (more) store_tiles((k + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n0 + wn * 32 * TN + ni * 32 + li;
            if (n >= p.DN) continue;
            float bv = 0.f;
            if constexpr (MODE == MODE_FWD) bv = p.bias ? p.bias[n] : 0.f;
            // data gradient: the 16 mask values (and old values) of this block are fetched before the first one is used,
            // from clamped rows -- the plain per-element form made every load wait for the previous element's store
            // (32 - 64 dependent round trips per tile: as long as the whole main loop of a 64-channel layer)
            unsigned o[16];                          // element offsets: every tensor is below 2^30 elements (check_desc)
            bool ok[16];
            float mk[16], old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                ok[r] = m < p.M;
                o[r] = (unsigned)(ok[r] ? m : p.M - 1) * (unsigned)p.DN + (unsigned)n;
            }
            if constexpr (MODE != MODE_FWD) {
                if (p.mask) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mk[r] = p.mask[o[r]];
                }
                if (p.accum) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[r] = p.dst[o[r]];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mi][ni][r];
                if constexpr (MODE == MODE_FWD) {
                    v += bv;
                    if (p.relu) v = v > 0.f ? v : 0.f;
                } else {
                    if (p.accum) v += old[r];
                    if (p.mask) v = mk[r] > 0.f ? v : 0.f;
                }
                if (!ok[r]) continue;
                if constexpr (OBF) reinterpret_cast<unsigned short*>(p.dst)[o[r]] = f2bf(v);
                else p.dst[o[r]] = v;
            }
            __builtin_amdgcn_sched_barrier(0);      // one block's addresses and values live at a time
        }
    }
}

// =================================================================================
// The same gather-GEMM with LDS-DMA staging (`buffer_load_dwordx4 ... lds`, as conv_bf16.hip): no staging
// registers, no ds_write, the zero padding still a per-lane out-of-range offset.  Tiles are unpadded
// 128-byte rows (32 k); the +4 pad's job is done by an XOR swizzle on the SOURCE chunk: LDS slot p of row
// r holds 16-byte chunk p ^ ((r>>1)&7), which keeps every 16-lane ds_read_b128 group on 16 distinct slots.
// The forward filter tile stays [32 k][BN] (natural HWIO rows, ds_read_b32 by n).  Fast path only
// (channel counts multiples of 4, no strided data gradient); the register-staged kernel above keeps the rest.
// =================================================================================
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// STRIDED: data gradient of a stride-2^k convolution (conv8_2, conv9_2): source pixel = (oh + dh) / stride when that
// division is exact; the validity of a (row, tap) pair is recomputed per iteration with shifts and masks.
// Two pipeline stages: tile k + 1 streams in while tile k is multiplied -- the loop is bound by the matrix pipe.  (Round 4 also
// carried a six-stage ring and a k split over four wave groups for the latency-bound small layers; in fp32 their serial k loop is
// matrix-pipe time -- 1152 dependent 32x32x2 MFMAs = 31 us per wave however the tiles are staged -- and both measured slower in
// the step: profiles/r04_q_ab_small_tile_f32.txt, r04_ac_ab_ksplit_f32.txt.  Removed in round 5.)
//
// Round 5, the 2x2 pools fused into their neighbours (conv.h):
//  * forward POOL (p.pool_dst): the m index enumerates WINDOWS, not pixels -- row m of the GEMM is cell q = (m>>4 & 1, m & 1)
//    of window (m >> 5) * 8 + ((m & 15) >> 1) in pooled raster order -- so that the 16 accumulator rows of a lane (MFMA C
//    layout: (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) hold four COMPLETE windows of its channel: the maxima, the first-maximum
//    cell and its sign are taken in registers, the pooled value leaves as one 128-byte line per 32 lanes, the 12-bit record
//    of 4 channels is assembled over 4 lanes.  The per-tap staging computes every row's pixel on its own, so the permuted
//    order costs nothing in the loop; tile rows past the last window or outside an odd image are zero rows.
//  * data gradient UNPOOL (p.unpool_rec): every pooled pixel's dx goes through the record to the four cells of the pool's
//    input gradient (recorded cell and positive maximum ? dx : 0).
template <int MODE, int WM, int WN, int TM, int TN, bool STRIDED = false, bool PARITY = false, bool POOL = false>
__global__ __launch_bounds__(256) void conv_gather_dma_kernel(GatherArgs pp) {
    static_assert(!POOL || (MODE == MODE_FWD && !STRIDED && !PARITY), "fused pool: plain forward only");
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    // (no local copy of the argument block: dynamically indexed arrays of a copy would live in scratch)
    const GatherArgs& p = pp;
    int wg_first = 0, wg_count = gridDim.x;
    int P_M = pp.M, P_DH = pp.DH, P_DW = pp.DW, P_ntaps = pp.ntaps, P_out_ph = pp.out_ph, P_out_pw = pp.out_pw;
    const int* P_tap_dh = pp.tap_dh;
    const int* P_tap_dw = pp.tap_dw;
    const int* P_tap_w = pp.tap_w;
    if (PARITY) {                                      // parity classes of a strided data gradient, one launch
        int c = 0;
        for (int k = 1; k < pp.nclass; ++k)
            if ((int)blockIdx.x >= pp.cls_wg0[k]) c = k;
        wg_first = pp.cls_wg0[c]; wg_count = pp.cls_wg0[c + 1] - wg_first;
        P_M = pp.cls_M[c]; P_DH = pp.cls_DH[c]; P_DW = pp.cls_DW[c]; P_ntaps = pp.cls_ntaps[c];
        P_out_ph = pp.cls_ph[c]; P_out_pw = pp.cls_pw[c];
        P_tap_dh = pp.cls_dh[c]; P_tap_dw = pp.cls_dw[c]; P_tap_w = pp.cls_w[c];
    }
    constexpr int A_N = BM / 32;                      // DMA instructions per thread: A tile (8 rows x 128 B per wave-instruction)
    constexpr int B_CPR = BN / 4;                     // forward: 16-byte chunks per k-row of the filter tile
    constexpr int B_RPP = 256 / B_CPR;                // forward: k-rows per pass
    constexpr int B_N = MODE == MODE_FWD ? BK / B_RPP : BN / 32;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BK % B_RPP == 0, "filter tile vs staging pass");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tid = threadIdx.x;
    unsigned char* const lds = reinterpret_cast<unsigned char*>(smem);
    const int wg = xcd_remap(blockIdx.x - wg_first, wg_count);
    const int mt = wg / p.NT, nt = wg - mt * p.NT;
    const int m0 = mt * BM, n0 = nt * BN;

    const int a_c4 = ((tid & 7) ^ ((tid >> 4) & 7)) * 4;      // this lane's (swizzled) k offset inside the 32-wide block
    unsigned a_off[A_N], a_msk[A_N];
    int s_b[A_N], s_h[A_N], s_w[A_N];                          // STRIDED: image base, row and column of the output pixel
    const int dshift = 31 - __builtin_clz(p.div);              // div is a power of two (checked by the launcher)
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
        const int m = m0 + (tid >> 3) + 32 * i;
        int ow, oh, b;
        bool mv = m < P_M;
        if constexpr (POOL) {      // row m = a cell of a pooling window (see above)
            const int g = (m >> 5) * 8 + ((m & 15) >> 1);
            const int pw_ = g % p.PW, t2 = g / p.PW;
            const int ph_ = t2 % p.PH;
            b = t2 / p.PH;
            oh = 2 * ph_ + ((m >> 4) & 1);
            ow = 2 * pw_ + (m & 1);
            mv = b < p.PB && oh < P_DH && ow < P_DW;
            if (!mv) { b = 0; oh = 0; ow = 0; }
        } else {
            const int mm = mv ? m : 0;
            ow = mm % P_DW;
            const int t2 = mm / P_DW;
            oh = t2 % P_DH;
            b = t2 / P_DH;
        }
        const int rh = mv ? oh * p.mul : -(1 << 20), rw = ow * p.mul;
        a_off[i] = (unsigned)((b * p.SH * p.SW + rh * p.SW + rw) * p.SC + a_c4) * 4u;
        s_b[i] = b * p.SH * p.SW; s_h[i] = rh; s_w[i] = rw;
        unsigned mk = 0;
        if constexpr (!STRIDED) {
            for (int t = 0; t < P_ntaps; ++t) {
                const int sh = rh + P_tap_dh[t], sw = rw + P_tap_dw[t];
                if ((unsigned)sh < (unsigned)p.SH && (unsigned)sw < (unsigned)p.SW) mk |= 1u << t;
            }
        }
        a_msk[i] = mk;
    }
    // The tap / channel-chunk part of every offset is uniform: it goes into the DMA instruction's SCALAR offset and the
    // per-lane part (a_off, b_off) stays a constant VGPR that is only swapped for the out-of-range value where the tap
    // falls into the padding.  The scalar part must not be negative, so the activation descriptor starts `tap_bias`
    // bytes before the tensor (the most negative tap offset); nothing is fetched from there.
    int tap_min = 0;
    for (int t = 0; t < P_ntaps; ++t) tap_min = min(tap_min, (P_tap_dh[t] * p.SW + P_tap_dw[t]) * p.SC);
    const unsigned tap_bias = (unsigned)(-tap_min) * 4u;
    const size_t src_bytes = (size_t)(POOL ? p.PB : P_M / (P_DH * P_DW)) * p.SH * p.SW * p.SC * 4u;
    const __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(const_cast<float*>(p.src)) - (STRIDED ? 0u : tap_bias), 0, (unsigned)(src_bytes + (STRIDED ? 0u : tap_bias)), 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wgt), 0, (unsigned)((size_t)p.wtaps * p.wci * p.wco * 4u), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    // filter tile addressing that does not change over the k loop
    unsigned b_off[B_N], b_ok[B_N];
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
        if constexpr (MODE == MODE_FWD) {           // row = k (channel), 16-byte chunk = 4 output channels
            const int kr = tid / B_CPR + B_RPP * i, col = (tid % B_CPR) * 4;
            b_ok[i] = 0u - (unsigned)(n0 + col < p.DN);
            b_off[i] = (unsigned)(kr * p.wco + n0 + col) * 4u;
        } else {                                     // row = n (dx channel), chunk = 4 dy channels (swizzled like A)
            const int n = n0 + (tid >> 3) + 32 * i;
            b_ok[i] = 0u - (unsigned)(n < p.DN);
            b_off[i] = (unsigned)((n < p.DN ? n : 0) * p.wco + a_c4) * 4u;
        }
    }

    const int nchunks = (p.SC + BK - 1) / BK;
    const int nk = nchunks * P_ntaps;

    auto issue = [&](int kiter, int stage) {
        const int cc = kiter / P_ntaps;
        const int tap = kiter - cc * P_ntaps;
        unsigned char* As = lds + stage * STAGE + wave * 1024;
        unsigned char* Bs = lds + stage * STAGE + A_BYTES + wave * 1024;
        const unsigned toff = (unsigned)(((P_tap_dh[tap] * p.SW + P_tap_dw[tap]) * p.SC + cc * BK) * 4);
        const unsigned cmask = 0u - (unsigned)(cc * BK + a_c4 < p.SC);
        if constexpr (STRIDED) {
            const int dh = P_tap_dh[tap], dw = P_tap_dw[tap], lowbits = p.div - 1;
#pragma unroll
            for (int i = 0; i < A_N; ++i) {
                const int sh = s_h[i] + dh, sw = s_w[i] + dw;
                const int qh = sh >> dshift, qw = sw >> dshift;                 // arithmetic shifts: a negative stays negative
                const bool ok = ((sh | sw) & lowbits) == 0 && (unsigned)qh < (unsigned)p.SH && (unsigned)qw < (unsigned)p.SW;
                const unsigned m = (0u - (unsigned)ok) & cmask;
                const unsigned off = (unsigned)((s_b[i] + qh * p.SW + qw) * p.SC + cc * BK + a_c4) * 4u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, LDS_PTR(As + i * 4096), 16, (int)((off & m) | (OOB & ~m)), 0, 0, 0);
            }
        } else {
            const int a_so = (int)(toff + tap_bias);
#pragma unroll
            for (int i = 0; i < A_N; ++i) {
                const bool ok = ((a_msk[i] >> tap) & 1u) != 0u && cmask != 0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, LDS_PTR(As + i * 4096), 16, (int)(ok ? a_off[i] : OOB), a_so, 0, 0);
            }
        }
        if constexpr (MODE == MODE_FWD) {
            const unsigned woff = (unsigned)((P_tap_w[tap] * p.wci + cc * BK) * p.wco) * 4u;
#pragma unroll
            for (int i = 0; i < B_N; ++i) {
                const int kr = tid / B_CPR + B_RPP * i;
                const bool ok = b_ok[i] != 0u && cc * BK + kr < p.SC;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wgt_rsrc, LDS_PTR(Bs + i * 4096), 16, (int)(ok ? b_off[i] : OOB), (int)woff, 0, 0);
            }
        } else {
            const unsigned woff = (unsigned)(P_tap_w[tap] * p.wci * p.wco + cc * BK) * 4u;
#pragma unroll
            for (int i = 0; i < B_N; ++i) {
                const bool ok = b_ok[i] != 0u && cmask != 0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wgt_rsrc, LDS_PTR(Bs + i * 4096), 16, (int)(ok ? b_off[i] : OOB), (int)woff, 0, 0);
            }
        }
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
    // fragment byte address inside a stage: row*128 + ((2g + lh) ^ swz(row))*16 = (row*128 + q0) ^ (g*32)
    const int q0 = (lh ^ ((li >> 1) & 7)) * 16;
    const int a_row = (wm * 32 * TM + li) * 128 + q0;
    const int b_row = A_BYTES + (wn * 32 * TN + li) * 128 + q0;                // data gradient
    const int b_col = A_BYTES + (wn * 32 * TN + li) * 4;                        // forward: [k][BN] floats

    auto compute = [&](int stage) {
        const unsigned char* S = lds + stage * STAGE;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(S + ((a_row + mi * 4096) ^ (g * 32)));
            if constexpr (MODE == MODE_FWD) {
                const int kb = g * 8 + lh * 4;
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int t = 0; t < 4; ++t) b[ni][t] = *reinterpret_cast<const float*>(S + b_col + ((kb + t) * BN + ni * 32) * 4);
            } else {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(S + ((b_row + ni * 4096) ^ (g * 32)));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t], b[ni][t], acc[mi][ni], 0, 0, 0);
        }
    };

    // two stages: tile k+1 streams in while tile k is multiplied; one barrier per iteration
    if (nk > 0) issue(0, 0);
    for (int k = 0; k < nk; ++k) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (k + 1 < nk) issue(k + 1, (k + 1) & 1);
        compute(k & 1);
    }

    // ---- forward with the fused 2x2 pool: maxima, first-maximum cell and sign in registers (see the kernel's header)
    if constexpr (POOL) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            // the four windows of this lane in the 32-row block: window wq = cells r = 2 (wq & 1) + 4 (wq >> 1) + {0, 1, 8, 9}
            unsigned po[4];                               // pooled pixel index, or ~0u
            bool okw[4], okh[4];
#pragma unroll
            for (int wq = 0; wq < 4; ++wq) {
                const int mb = m0 + wm * 32 * TM + mi * 32;
                const int g = (mb >> 5) * 8 + (wq & 1) + 4 * (wq >> 1) + 2 * lh;      // window: cc0 = 2 (wq&1) + 8 (wq>>1) + 4 lh, g = cc0 / 2
                const int pw_ = g % p.PW, t2 = g / p.PW;
                const int ph_ = t2 % p.PH, b = t2 / p.PH;
                po[wq] = b < p.PB ? (unsigned)g : ~0u;
                okw[wq] = 2 * pw_ + 1 < P_DW;
                okh[wq] = 2 * ph_ + 1 < P_DH;
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + wn * 32 * TN + ni * 32 + li;
                const bool nv = n < p.DN;
                const float bv = (p.bias && nv) ? p.bias[n] : 0.f;
#pragma unroll
                for (int wq = 0; wq < 4; ++wq) {
                    const int r0 = 2 * (wq & 1) + 4 * (wq >> 1);
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float t = acc[mi][ni][r0 + (q & 1) + 8 * (q >> 1)] + bv;
                        v[q] = (p.relu && !(t > 0.f)) ? 0.f : t;
                    }
                    const bool ok[4] = {true, okw[wq], okh[wq], okw[wq] && okh[wq]};
                    float mm = v[0];
                    unsigned a = 0;
#pragma unroll
                    for (int q = 1; q < 4; ++q)
                        if (ok[q] && v[q] > mm) { mm = v[q]; a = q; }      // strict: the first maximum wins (ops.hip)
                    unsigned rb = (a | (mm > 0.f ? 4u : 0u)) << (3 * (li & 3));
                    rb |= (unsigned)__shfl_xor((int)rb, 1, 64);
                    rb |= (unsigned)__shfl_xor((int)rb, 2, 64);
                    if (nv && po[wq] != ~0u) {
                        p.pool_dst[(size_t)po[wq] * p.DN + n] = mm;
                        if (p.pool_rec && (li & 3) == 0) p.pool_rec[(size_t)po[wq] * (p.DN >> 2) + (n >> 2)] = (unsigned short)rb;
                    }
                }
            }
        }
        return;
    }
    // ---- data gradient through a pool's record into the pool's input gradient
    if (MODE == MODE_DGRAD && !PARITY && !STRIDED && p.unpool_rec) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {              // four rows of the block at a time (registers)
                unsigned base[4], mrow[4];
                bool okm[4], okw[4], okh[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = m0 + wm * 32 * TM + mi * 32 + e + 8 * rr + 4 * lh;      // r = e + 4 rr
                    okm[e] = m < P_M;
                    const int mm = okm[e] ? m : 0;
                    const int ow = mm % P_DW, t2 = mm / P_DW;
                    const int oh = t2 % P_DH, b = t2 / P_DH;
                    mrow[e] = (unsigned)mm;
                    base[e] = (unsigned)((b * p.UH + 2 * oh) * p.UW + 2 * ow);
                    okw[e] = 2 * ow + 1 < p.UW;
                    okh[e] = 2 * oh + 1 < p.UH;
                }
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const int n = n0 + wn * 32 * TN + ni * 32 + li;
                    if (n >= p.DN) continue;
                    unsigned rc[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) rc[e] = p.unpool_rec[(size_t)mrow[e] * (p.DN >> 2) + (n >> 2)];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!okm[e]) continue;
                        const unsigned re = (rc[e] >> (3 * (n & 3))) & 7u;
                        const float v = acc[mi][ni][e + 4 * rr];
                        const bool ok[4] = {true, okw[e], okh[e], okw[e] && okh[e]};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (!ok[q]) continue;
                            const unsigned pix = base[e] + (unsigned)((q >> 1) * p.UW + (q & 1));
                            p.dst[(size_t)pix * p.DN + n] = ((re & 3u) == (unsigned)q && (re & 4u)) ? v : 0.f;
                        }
                    }
                }
            }
        }
        return;
    }

#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n0 + wn * 32 * TN + ni * 32 + li;
            if (n >= p.DN) continue;
            float bv = 0.f;
            if constexpr (MODE == MODE_FWD) bv = p.bias ? p.bias[n] : 0.f;
            // (data gradient: mask / old values of the block fetched up front from clamped rows, see conv_gather_kernel)
            unsigned o[16];                          // element offsets: every tensor is below 2^30 elements (check_desc)
            bool ok[16];
            float mk[16], old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wm * 32 * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                ok[r] = m < P_M;
                if (!ok[r]) m = P_M - 1;
                unsigned pix = (unsigned)m;
                if (PARITY) {                           // parity class: virtual pixel (b, a, c) -> real pixel
                    const int c = m % P_DW, t2 = m / P_DW;
                    const int a = t2 % P_DH, b = t2 / P_DH;
                    pix = (unsigned)((b * p.ODH + a * p.out_mul + P_out_ph) * p.ODW + c * p.out_mul + P_out_pw);
                }
                o[r] = pix * (unsigned)p.DN + (unsigned)n;
            }
            if constexpr (MODE != MODE_FWD) {
                if (p.mask) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mk[r] = p.mask[o[r]];
                }
                if (p.accum) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[r] = p.dst[o[r]];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mi][ni][r];
                if constexpr (MODE == MODE_FWD) {
                    v += bv;
                    if (p.relu) v = v > 0.f ? v : 0.f;
                } else {
                    if (p.accum) v += old[r];
                    if (p.mask) v = mk[r] > 0.f ? v : 0.f;
                }
                if (ok[r]) p.dst[o[r]] = v;
            }
            __builtin_amdgcn_sched_barrier(0);      // one block's addresses and values live at a time
        }
    }
}

// =================================================================================
// weight gradient:  dW[(tap, c)][n] = sum_m x[pix(m, tap)][c] * dy[m][n]
// One workgroup owns (tap, channel tile, n tile, pixel split); 32 pixels per iteration.
// MFMA roles: A[i = channel][k = pixel], B[k = pixel][j = n]; both tiles are stored
// pixel-major in LDS and read with conflict-free ds_read_b32.
// =================================================================================
struct WgradArgs {
    const float* x;
    const float* dy;
    float* ws;              // [nsplit][ntaps*Ci*Co + Co]
    int M, Hi, Wi, Ci, Ho, Wo, Co;
    int ntaps, stride;
    int CT, NT;             // channel tiles, n tiles
    int ntaps_grid;         // taps enumerated by the grid (1 for the packed small-C case)
    int mchunk, nsplit;
    int tap_dh[9], tap_dw[9];
};

// YBF: dy is bf16 (conv1_1 of the bf16 configuration; x stays the fp32 image)
template <int WM, int WN, int TM, int TN, bool SMALLC, bool YBF = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
    constexpr int BKT = 32 * TM * WM, BNT = 32 * TN * WN, BP = 32;
    // staging map: thread -> ONE pixel row (tid >> 3) and the 16-byte chunks (tid & 7) + 8*j of it,
    // so the per-pixel index math runs once per thread and every load instruction still covers
    // whole 128-byte lines (8 lanes per row)
    constexpr int X_N = BKT / 32, Y_N = BNT / 32;
    constexpr int X_LDS = BP * BKT, Y_LDS = BP * BNT;
    static_assert(WM * WN == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    // 1-D grid: consecutive ids (after the XCD remap) = all (tap, channel, n) tiles of ONE pixel split,
    // so the workgroups that re-read the same x / dy rows run together on one XCD and share its L2
    const int wgid = xcd_remap(blockIdx.x, gridDim.x);
    const int ntiles = p.ntaps_grid * p.CT * p.NT;
    const int split = wgid / ntiles;
    int tile = wgid - split * ntiles;
    const int nt = tile % p.NT;
    tile /= p.NT;
    const int ct = tile % p.CT;
    const int tap = tile / p.CT;
    const int c0 = ct * BKT, n0 = nt * BNT;
    const int mbeg = split * p.mchunk;
    const int mend = min(p.M, mbeg + p.mchunk);
    const int niter = (mend - mbeg + BP - 1) / BP;
    const int dh = p.tap_dh[tap], dw = p.tap_dw[tap];
    const bool do_bias = (tap == 0 && ct == 0);

    f32x4 xreg[X_N], yreg[Y_N];

    // this thread's pixel row, advanced by BP pixels per iteration
    const int row = tid >> 3, cb = (tid & 7) * 4;
    int pw, ph, pb;
    {
        const int m = mbeg + row;
        pw = m % p.Wo;
        const int t2 = m / p.Wo;
        ph = t2 % p.Ho;
        pb = t2 / p.Ho;
    }

    // branch-free buffer loads: out-of-image pixels, rows past the split and channels past the
    // tensor are steered to an out-of-range offset, for which the buffer unit returns zeros
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((size_t)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.Ci * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (unsigned)((size_t)p.M * p.Co * (YBF ? 2u : 4u)), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned xcmask[X_N], ycmask[Y_N];
#pragma unroll
    for (int j = 0; j < X_N; ++j) xcmask[j] = 0u - (unsigned)(c0 + cb + 32 * j < p.Ci);
#pragma unroll
    for (int j = 0; j < Y_N; ++j) ycmask[j] = 0u - (unsigned)(n0 + cb + 32 * j < p.Co);

    int wk_dh[4], wk_dw[4], wk_c[4];
    unsigned wk_valid[4];
    if constexpr (SMALLC) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = cb + e;
            const bool kv = k < p.ntaps * p.Ci;
            const int tp = kv ? k / p.Ci : 0;
            wk_dh[e] = p.tap_dh[tp]; wk_dw[e] = p.tap_dw[tp]; wk_c[e] = kv ? k - tp * p.Ci : 0;
            wk_valid[e] = 0u - (unsigned)kv;
        }
    }

    auto load_tiles = [&](int it) {
        const int m = mbeg + it * BP + row;
        const int ow = pw, oh = ph, b = pb;
        if (p.Wo >= BP) {
            pw += BP;
            if (pw >= p.Wo) {
                pw -= p.Wo;
                if (++ph == p.Ho) { ph = 0; ++pb; }
            }
        } else {
            const int m2 = m + BP;
            pw = m2 % p.Wo;
            const int t2 = m2 / p.Wo;
            ph = t2 % p.Ho;
            pb = t2 / p.Ho;
        }
        const unsigned rowok = 0u - (unsigned)(m < mend);
        if constexpr (SMALLC) {
            // packed k = tap*Ci + c < 32: only chunk 0 holds data; (tap, channel) of this thread's four
            // k's are constants (wk_*), only the pixel moves
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int sh = oh * p.stride + wk_dh[e], sw = ow * p.stride + wk_dw[e];
                const unsigned m = rowok & wk_valid[e] &
                                   (0u - ((unsigned)((unsigned)sh < (unsigned)p.Hi) & (unsigned)((unsigned)sw < (unsigned)p.Wi)));
                const unsigned off = (unsigned)(((b * p.Hi + sh) * p.Wi + sw) * p.Ci + wk_c[e]) * 4u;
                v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (off & m) | (OOB & ~m), 0, 0));
            }
            xreg[0] = v;
#pragma unroll
            for (int j = 1; j < X_N; ++j) xreg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            const int sh = oh * p.stride + dh, sw = ow * p.stride + dw;
            const unsigned inb = 0u - ((unsigned)((unsigned)sh < (unsigned)p.Hi) & (unsigned)((unsigned)sw < (unsigned)p.Wi));
            const unsigned base = (unsigned)(((b * p.Hi + sh) * p.Wi + sw) * p.Ci + c0 + cb) * 4u;
#pragma unroll
            for (int j = 0; j < X_N; ++j) {
                const unsigned mk = rowok & inb & xcmask[j];
                xreg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ((base + 128u * j) & mk) | (OOB & ~mk), 0, 0));
            }
        }
        const unsigned ybase = (unsigned)(m * p.Co + n0 + cb) * (YBF ? 2u : 4u);
#pragma unroll
        for (int j = 0; j < Y_N; ++j) {
            const unsigned mk = rowok & ycmask[j];
            if constexpr (YBF) {
                const u32x2 w = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(y_rsrc, ((ybase + 64u * j) & mk) | (OOB & ~mk), 0, 0));
                yreg[j] = f32x4{lo2f(w[0]), hi2f(w[0]), lo2f(w[1]), hi2f(w[1])};
            } else {
                yreg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, ((ybase + 128u * j) & mk) | (OOB & ~mk), 0, 0));
            }
        }
    };
    auto store_tiles = [&](int buf) {
        float* Xs = smem + buf * (X_LDS + Y_LDS);
        float* Ys = Xs + X_LDS;
#pragma unroll
        for (int j = 0; j < X_N; ++j) *reinterpret_cast<f32x4*>(Xs + row * BKT + cb + 32 * j) = xreg[j];
#pragma unroll
        for (int j = 0; j < Y_N; ++j) *reinterpret_cast<f32x4*>(Ys + row * BNT + cb + 32 * j) = yreg[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum = 0.f;

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, lh = lane >> 5;

    auto compute = [&](int buf) {
        const float* Xs = smem + buf * (X_LDS + Y_LDS);
        const float* Ys = Xs + X_LDS;
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
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (do_bias && tid < BNT) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < BP; ++r) s += Ys[r * BNT + tid];
            bsum += s;
        }
    };

    if (niter > 0) {
        load_tiles(0);
        store_tiles(0);
    }
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const bool more = it + 1 < niter;
        if (more) load_tiles(it + 1);
        compute(it & 1);
        if (more) store_tiles((it + 1) & 1);
        __syncthreads();
    }

    const size_t wcount = (size_t)p.ntaps * p.Ci * p.Co;
    float* slab = p.ws + (size_t)split * (wcount + p.Co);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n0 + wn * 32 * TN + ni * 32 + li;
            if (n >= p.Co) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = wm * 32 * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                size_t row;
                if constexpr (SMALLC) {
                    if (kl >= p.ntaps * p.Ci) continue;
                    row = kl;
                } else {
                    if (c0 + kl >= p.Ci) continue;
                    row = (size_t)tap * p.Ci + c0 + kl;
                }
                slab[row * p.Co + n] = acc[mi][ni][r];
            }
        }
    }
    if (do_bias && tid < BNT && n0 + tid < p.Co) slab[wcount + n0 + tid] = bsum;
}

// The weight gradient with LDS-DMA staging: x and dy tiles keep their pixel-major global rows in LDS (what the
// ds_read_b32 fragment reads want anyway), so a tile is a plain DMA copy -- no staging registers, no ds_write.
// Each staged row walks (b, oh, ow) by a mixed-radix add of 32 pixels per iteration; its byte offset moves with it
// by uniform constants and the tap's zero padding is an interval test (no multiplies or divisions in the loop).
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_wgrad_dma_kernel(WgradArgs p) {
    constexpr int BKT = 32 * TM * WM, BNT = 32 * TN * WN, BP = 32;
    constexpr int XC = BKT / 4, YC = BNT / 4;                   // 16-byte chunks per pixel row
    constexpr int XRPP = 256 / XC, YRPP = 256 / YC;             // pixel rows per DMA pass of the workgroup
    constexpr int X_N = BP / XRPP, Y_N = BP / YRPP;
    constexpr int X_BYTES = BP * BKT * 4, STAGE = BP * (BKT + BNT) * 4;
    static_assert(WM * WN == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* const lds = reinterpret_cast<unsigned char*>(smem);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wgid = xcd_remap(blockIdx.x, gridDim.x);
    const int ntiles = p.ntaps_grid * p.CT * p.NT;
    const int split = wgid / ntiles;
    int tile = wgid - split * ntiles;
    const int nt = tile % p.NT;
    tile /= p.NT;
    const int ct = tile % p.CT;
    const int tap = tile / p.CT;
    const int c0 = ct * BKT, n0 = nt * BNT;
    const int mbeg = split * p.mchunk;
    const int mend = min(p.M, mbeg + p.mchunk);
    const int niter = (mend - mbeg + BP - 1) / BP;
    const int dh = p.tap_dh[tap], dw = p.tap_dw[tap];
    const bool do_bias = (tap == 0 && ct == 0);

    const int xr = tid / XC, xc = tid % XC, yr = tid / YC, yc = tid % YC;
    const unsigned xcm = 0u - (unsigned)(c0 + xc * 4 < p.Ci), ycm = 0u - (unsigned)(n0 + yc * 4 < p.Co);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((size_t)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.Ci * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (unsigned)((size_t)p.M * p.Co * 4u), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;

    const int adv_w = BP % p.Wo, adv_t = BP / p.Wo;
    const int adv_h = adv_t % p.Ho, adv_b = adv_t / p.Ho;
    const int C4 = p.Ci * 4;
    const unsigned xadv = (unsigned)(((adv_b * p.Hi + adv_h * p.stride) * p.Wi + adv_w * p.stride) * C4);
    const unsigned xadv_cw = (unsigned)((p.stride * p.Wi - p.Wo * p.stride) * C4);     // ow wrapped: next image row
    const unsigned xadv_ch = (unsigned)((p.Hi - p.Ho * p.stride) * p.Wi * C4);         // oh wrapped: next image
    // oh*stride + dh in [0, Hi)  <=>  oh in [h_lo, h_lo + h_span]   (likewise ow)
    const int h_lo = dh < 0 ? (-dh + p.stride - 1) / p.stride : 0, w_lo = dw < 0 ? (-dw + p.stride - 1) / p.stride : 0;
    const int h_hi = (p.Hi - 1 - dh) >= 0 ? (p.Hi - 1 - dh) / p.stride : -1, w_hi = (p.Wi - 1 - dw) >= 0 ? (p.Wi - 1 - dw) / p.stride : -1;
    const int h_span_i = (h_hi < p.Ho - 1 ? h_hi : p.Ho - 1) - h_lo, w_span_i = (w_hi < p.Wo - 1 ? w_hi : p.Wo - 1) - w_lo;
    const unsigned xcm_t = (h_span_i < 0 || w_span_i < 0) ? 0u : xcm;                  // the tap never touches the image
    const unsigned h_span = (unsigned)h_span_i, w_span = (unsigned)w_span_i;
    int pw[X_N], ph[X_N];
    unsigned xoff[X_N], yoff[Y_N];
#pragma unroll
    for (int j = 0; j < X_N; ++j) {
        const int m = mbeg + xr + j * XRPP;
        pw[j] = m % p.Wo;
        const int t2 = m / p.Wo;
        ph[j] = t2 % p.Ho;
        const int pb = t2 / p.Ho;
        xoff[j] = (unsigned)(((pb * p.Hi + ph[j] * p.stride + dh) * p.Wi + pw[j] * p.stride + dw) * C4 + (c0 + xc * 4) * 4);
    }
#pragma unroll
    for (int j = 0; j < Y_N; ++j) yoff[j] = (unsigned)(((mbeg + yr + j * YRPP) * p.Co + n0 + yc * 4) * 4);

    auto issue = [&](int it, int stage) {
        unsigned char* Xs = lds + stage * STAGE + wave * 1024;
        unsigned char* Ys = lds + stage * STAGE + X_BYTES + wave * 1024;
        const int left = mend - (mbeg + it * BP);
#pragma unroll
        for (int j = 0; j < X_N; ++j) {
            const bool ok = (xr + j * XRPP < left) && (unsigned)(ph[j] - h_lo) <= h_span && (unsigned)(pw[j] - w_lo) <= w_span;
            const unsigned mk = xcm_t & (0u - (unsigned)ok);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, LDS_PTR(Xs + j * 4096), 16, (int)((xoff[j] & mk) | (OOB & ~mk)), 0, 0, 0);
            pw[j] += adv_w;
            const bool cw = pw[j] >= p.Wo;
            pw[j] -= cw ? p.Wo : 0;
            ph[j] += adv_h + (cw ? 1 : 0);
            const bool ch = ph[j] >= p.Ho;
            ph[j] -= ch ? p.Ho : 0;
            xoff[j] += xadv + (cw ? xadv_cw : 0u) + (ch ? xadv_ch : 0u);
        }
#pragma unroll
        for (int j = 0; j < Y_N; ++j) {
            const unsigned mk = ycm & (0u - (unsigned)(yr + j * YRPP < left));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, LDS_PTR(Ys + j * 4096), 16, (int)((yoff[j] & mk) | (OOB & ~mk)), 0, 0, 0);
            yoff[j] += (unsigned)(BP * p.Co * 4);
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
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
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

    const size_t wcount = (size_t)p.ntaps * p.Ci * p.Co;
    float* slab = p.ws + (size_t)split * (wcount + p.Co);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n0 + wn * 32 * TN + ni * 32 + li;
            if (n >= p.Co) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = wm * 32 * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (c0 + kl >= p.Ci) continue;
                slab[((size_t)tap * p.Ci + c0 + kl) * p.Co + n] = acc[mi][ni][r];
            }
        }
    }
    if (do_bias && tid < BNT && n0 + tid < p.Co) slab[wcount + n0 + tid] = bsum;
}

// Fixed-order reduce of the split slabs: dw = sum_s slab_s + wd*w ; db = sum_s bias_s.
// sum_{k = k0, k0 + step, ... < n} ws[k * stride + i .. i + 3], added in that order; the loads of 8 slabs are issued before
// their first add (the plain loop waited for every slab's load in turn: s_waitcnt vmcnt(0) per iteration)
__device__ __forceinline__ f32x4 slab_sum(const float* __restrict__ ws, size_t stride, size_t i, int k0, int step, int n) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int k = k0;
    for (; k + 7 * step < n; k += 8 * step) {
        f32x4 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = ld4(ws + (size_t)(k + u * step) * stride + i);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u];
    }
    if (k + 3 * step < n) {
        f32x4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = ld4(ws + (size_t)(k + u * step) * stride + i);
#pragma unroll
        for (int u = 0; u < 4; ++u) s += a[u];
        k += 4 * step;
    }
    for (; k < n; k += step) s += ld4(ws + (size_t)k * stride + i);
    return s;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int nsplit, size_t wcount,
                                                           int Co, float* __restrict__ dw, float* __restrict__ db,
                                                           const float* __restrict__ w, float wd) {
    const size_t total = wcount + Co;
    const size_t stride = total;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (size_t)gridDim.x * blockDim.x * 4) {
        // wcount and Co are multiples of 4: a float4 never straddles the weight/bias boundary
        f32x4 s = slab_sum(ws, stride, i, 0, 1, nsplit);
        if (i < wcount) {
            if (wd != 0.f) s += wd * ld4(w + i);
            *reinterpret_cast<f32x4*>(dw + i) = s;
        } else if (db) {
            *reinterpret_cast<f32x4*>(db + (i - wcount)) = s;
        }
    }
}

// Many slabs, few elements (early layers: 9 tiles x ~170 pixel splits): 16 threads share one float4
// element and each sums every 16th slab; the 16 partials are added in a fixed order through LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ ws, int nsplit, size_t wcount,
                                                                int Co, float* __restrict__ dw, float* __restrict__ db,
                                                                const float* __restrict__ w, float wd) {
    __shared__ f32x4 part[16][16];
    const size_t total = wcount + Co;
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const size_t i = ((size_t)blockIdx.x * 16 + e) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < total) s = slab_sum(ws, total, i, sl, 16, nsplit);
    part[sl][e] = s;
    __syncthreads();
    if (sl == 0 && i < total) {
        f32x4 t = part[0][e];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += part[q][e];
        if (i < wcount) {
            if (wd != 0.f) t += wd * ld4(w + i);
            *reinterpret_cast<f32x4*>(dw + i) = t;
        } else if (db) {
            *reinterpret_cast<f32x4*>(db + (i - wcount)) = t;
        }
    }
}

// =================================================================================
// host launchers
// =================================================================================
template <int MODE, int WM, int WN, int TM, int TN, bool SMALLC, bool STRIDED, bool OBF = false>
static void launch_gather(GatherArgs& a, const char* label, double flops, double bytes, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr size_t lds = 2 * (size_t)(BM * LDA + (MODE == MODE_FWD ? BK * BN : BN * LDA)) * sizeof(float);
    auto kern = conv_gather_kernel<MODE, WM, WN, TM, TN, SMALLC, STRIDED, OBF>;
    static bool once = (set_lds(kern, lds), true);
    (void)once;
    const int MT = cdiv(a.M, BM);
    a.NT = cdiv(a.DN, BN);
    ProfScope prof(label, flops, bytes, s);
    SSD_LAUNCH_STOP(kern, dim3(MT * a.NT), dim3(256), lds, s, a);
    HIP_OK(hipGetLastError());
}

static double conv_bytes(const ConvDesc& d) { return 4.0 * conv_elems(d); }

static void check_desc(const ConvDesc& d) {
    SSD_REQUIRE(d.KH * d.KW <= 9 && d.KH * d.KW >= 1, "conv: at most 9 taps (got %dx%d)", d.KH, d.KW);
    SSD_REQUIRE(d.Co % 4 == 0, "conv: Co must be a multiple of 4 (got %d)", d.Co);
    SSD_REQUIRE(d.Ci % 4 == 0 || d.Ci * d.KH * d.KW <= 32, "conv: Ci must be a multiple of 4 or Ci*taps <= 32 (got %d)", d.Ci);
    // buffer descriptors and gather offsets are 32-bit BYTE quantities (0xFFFFFFF0 is the out-of-range sentinel)
    SSD_REQUIRE((long long)d.B * d.Hi * d.Wi * d.Ci < (1LL << 30) - 4 && (long long)d.B * d.Ho * d.Wo * d.Co < (1LL << 30) - 4,
                "conv: a tensor of this layer exceeds 4 GiB (32-bit byte offsets): lower the batch");
}

template <int MODE, int WM, int WN, int TM, int TN, bool STRIDED = false, bool PARITY = false, bool POOL = false>
static void launch_gather_dma(GatherArgs& a, const char* label, double flops, double bytes, hipStream_t s, int grid = 0) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr size_t lds = 2 * (size_t)(BM + BN) * 128;
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = conv_gather_dma_kernel<MODE, WM, WN, TM, TN, STRIDED, PARITY, POOL>;
    static bool once = (set_lds(kern, lds), true);
    (void)once;
    const int MT = cdiv(a.M, BM);
    a.NT = cdiv(a.DN, BN);
    ProfScope prof(label, flops, bytes, s);
    SSD_LAUNCH_STOP(kern, dim3(grid > 0 ? grid : MT * a.NT), dim3(256), lds, s, a);
    HIP_OK(hipGetLastError());
}

// Tile choice: all co-resident workgroups of a CU share its matrix pipes, so a launch costs
// about ceil(workgroups / 256 CUs) * BM * BN (tile work incl. padded rows/columns) divided by a
// per-tile efficiency (bigger tiles re-use LDS fragments better).  0:128x128 1:128x64 2:64x128 3:64x64
static int pick_tile(long long M, int N, int mode) {
    static const int forced = env_int("SSD_TILE", -1);      // tuning override
    if (forced >= 0 && forced < 4) return forced;
    static const int bm[4] = {128, 128, 64, 64}, bn[4] = {128, 64, 128, 64};
    // relative per-tile efficiency measured on vgg300 layers at batch 32 (tools/bench_conv.py):
    // forward runs 116-125 TF/s on every tile; the data-gradient is fastest on 64x64
    // (LDS-DMA kernels: forward is best on 128x128 (conv2_2 126, conv3_2 131, conv4_2 125 TF/s), the data gradient on 64x128
    // (123-128), on 64x64 when N = 64 (conv1_2 110))
    static const double dma_fwd[4] = {1.03, 1.0, 1.0, 0.96}, dma_dg[4] = {0.96, 0.97, 1.0, 0.94};
    // N = 64 (conv1_2, the un-pooling conv2_1).  Round 6 re-measured the data gradient on the LDS-DMA kernels: alone, 128 x 64 runs 1.821 /
    // 0.956 ms against 64 x 64's 1.894 / 0.968 (256 x 64, two workgroups of 80 KB per CU: 1.938 / 1.007; profiles/r06_ai_*) -- in the
    // overlapped step the order is the other way round, 50.197 against 50.133 ms (profiles/r06_aj_*): the 64 x 64 tile stays.
    if (mode == MODE_DGRAD && N <= 64) return 3;
    const double* eff = mode == MODE_FWD ? dma_fwd : dma_dg;
    int best = 0;
    double bc = 1e300;
    for (int c = 0; c < 4; ++c) {
        const long long wgs = (long long)cdiv(M, bm[c]) * cdiv(N, bn[c]);
        const double cost = (double)((wgs + 255) / 256) * bm[c] * bn[c] / eff[c];
        if (cost < bc * 0.999) { bc = cost; best = c; }
    }
    return best;
}

static void conv_fwd_any(const ConvDesc& d, const float* x, const float* w, const float* bias, void* y, bool y_bf16, bool relu,
                         hipStream_t s, float* y_pool = nullptr, void* rec = nullptr) {
    check_desc(d);
    const bool pool = y_pool != nullptr;
    GatherArgs a{};
    a.src = x; a.wgt = w; a.bias = bias; a.mask = nullptr; a.dst = static_cast<float*>(y);
    a.M = d.B * d.Ho * d.Wo; a.DH = d.Ho; a.DW = d.Wo; a.DN = d.Co;
    if (pool) {      // GEMM rows enumerate the cells of the pooling windows, 8 windows per 32-row block (conv_gather_dma_kernel)
        a.pool_dst = y_pool; a.pool_rec = static_cast<unsigned short*>(rec);
        a.PB = d.B; a.PH = (d.Ho + 1) / 2; a.PW = (d.Wo + 1) / 2;
        a.M = cdiv((long long)d.B * a.PH * a.PW, 8) * 32;
    }
    a.SH = d.Hi; a.SW = d.Wi; a.SC = d.Ci;
    a.ntaps = d.KH * d.KW; a.mul = d.stride; a.div = 1;
    a.wci = d.Ci; a.wco = d.Co; a.relu = relu; a.accum = 0;
    a.wtaps = a.ntaps; a.out_mul = 1; a.out_ph = a.out_pw = 0; a.ODH = a.DH; a.ODW = a.DW;
    for (int kh = 0; kh < d.KH; ++kh)
        for (int kw = 0; kw < d.KW; ++kw) {
            a.tap_dh[kh * d.KW + kw] = kh * d.dil - d.pad_h;
            a.tap_dw[kh * d.KW + kw] = kw * d.dil - d.pad_w;
            a.tap_w[kh * d.KW + kw] = kh * d.KW + kw;
        }
    const bool smallc = d.Ci % 4 != 0;
    const double fl = conv_flops(d), by = conv_bytes(d);
    SSD_REQUIRE(!pool || !smallc, "fused pool: not for the packed small-C layer");
    SSD_REQUIRE(!y_bf16 || smallc, "bf16 output from fp32 input: only the packed small-C layer (conv1_1)");
    static const int first_kernel = env_int("SSD_FIRST_F32", 1);        // A/B switch
    if (smallc && !y_bf16 && first_kernel && conv_first_fwd_f32_applicable(d)) {
        conv_first_fwd_f32(d, x, w, bias, static_cast<float*>(y), relu, s);
        return;
    }
    if (smallc) {
        if (y_bf16) launch_gather<MODE_FWD, 4, 1, 1, 2, true, false, true>(a, "conv_fwd_smallc_128x64_bf16out", fl, by, s);
        else launch_gather<MODE_FWD, 4, 1, 1, 2, true, false>(a, "conv_fwd_smallc_128x64", fl, by, s);
        return;
    }
    // (the cost model counts rounds of workgroups on the chip: the launches that run side by side -- the executor's forward
    // lanes, g_conv_lanes -- share it)
    // (the fused pool's GEMM has up to 3 % more rows -- the cells an odd image lacks: the tile is picked for the convolution's own
    // pixel count, or conv3_3 lands one modelled round above conv3_2 and takes the 128 x 64 tile: 1741 us against 1624,
    // profiles/r05_b_per_layer_f32.txt)
    const int cfg = pick_tile((long long)d.B * d.Ho * d.Wo * g_conv_lanes, a.DN, MODE_FWD);
    if (pool) {
        switch (cfg) {
        case 0: launch_gather_dma<MODE_FWD, 2, 2, 2, 2, false, false, true>(a, "conv_fwd_pool_128x128", fl, by, s); break;
        case 1: launch_gather_dma<MODE_FWD, 4, 1, 1, 2, false, false, true>(a, "conv_fwd_pool_128x64", fl, by, s); break;
        case 2: launch_gather_dma<MODE_FWD, 2, 2, 1, 2, false, false, true>(a, "conv_fwd_pool_64x128", fl, by, s); break;
        default: launch_gather_dma<MODE_FWD, 2, 2, 1, 1, false, false, true>(a, "conv_fwd_pool_64x64", fl, by, s); break;
        }
        return;
    }
    switch (cfg) {
    case 0: launch_gather_dma<MODE_FWD, 2, 2, 2, 2>(a, "conv_fwd_128x128", fl, by, s); break;
    case 1: launch_gather_dma<MODE_FWD, 4, 1, 1, 2>(a, "conv_fwd_128x64", fl, by, s); break;
    case 2: launch_gather_dma<MODE_FWD, 2, 2, 1, 2>(a, "conv_fwd_64x128", fl, by, s); break;
    default: launch_gather_dma<MODE_FWD, 2, 2, 1, 1>(a, "conv_fwd_64x64", fl, by, s); break;
    }
}

void conv_fwd(const ConvDesc& d, const float* x, const float* w, const float* bias, float* y, bool relu, hipStream_t s) {
    conv_fwd_any(d, x, w, bias, y, false, relu, s);
}

// ---- fused 2x2 pool (conv.h) -------------------------------------------------------------------------------------
static bool pool_shape(const ConvDesc& d) {      // 3x3 (or any <= 9 taps) stride-1 same-size convolution
    return d.stride == 1 && d.Hi == d.Ho && d.Wi == d.Wo && d.Ci % 4 == 0 && d.Co % 4 == 0 && d.KH * d.KW <= 9;
}
bool conv_fwd_pool_supported(const ConvDesc& d) { return pool_shape(d); }
void conv_fwd_pool(const ConvDesc& d, const float* x, const float* w, const float* bias, float* y_pool, void* rec, hipStream_t s) {
    SSD_REQUIRE(conv_fwd_pool_supported(d), "conv_fwd_pool: unsupported shape");
    SSD_REQUIRE(y_pool != nullptr, "conv_fwd_pool: null output");
    conv_fwd_any(d, x, w, bias, nullptr, false, true, s, y_pool, rec);
}
bool conv_dgrad_unpool_supported(const ConvDesc& d) { return d.stride == 1 && d.Ci % 4 == 0 && d.Co % 4 == 0; }

void conv_fwd_smallc_bf16out(const ConvDesc& d, const float* x, const float* w, const float* bias, bf16_t* y, bool relu,
                             hipStream_t s) {
    conv_fwd_any(d, x, w, bias, y, true, relu, s);
}

static void conv_dgrad_any(const ConvDesc& d, const float* dy, const float* w, float* dx, const float* mask, bool accumulate,
                           hipStream_t s, const void* unpool_rec = nullptr, int UH = 0, int UW = 0) {
    check_desc(d);
    SSD_REQUIRE(d.Ci % 4 == 0, "conv_dgrad: Ci must be a multiple of 4");
    SSD_REQUIRE(unpool_rec == nullptr || d.stride == 1, "conv_dgrad: the fused un-pool is for stride-1 convolutions");
    GatherArgs a{};
    a.src = dy; a.wgt = w; a.bias = nullptr; a.mask = mask; a.dst = dx;
    a.M = d.B * d.Hi * d.Wi; a.DH = d.Hi; a.DW = d.Wi; a.DN = d.Ci;
    a.SH = d.Ho; a.SW = d.Wo; a.SC = d.Co;
    a.ntaps = d.KH * d.KW; a.mul = 1; a.div = d.stride;
    a.wci = d.Ci; a.wco = d.Co; a.relu = 0; a.accum = accumulate;
    a.wtaps = a.ntaps; a.out_mul = 1; a.out_ph = a.out_pw = 0; a.ODH = a.DH; a.ODW = a.DW;
    a.unpool_rec = static_cast<const unsigned short*>(unpool_rec); a.UH = UH; a.UW = UW;
    const bool unpool = unpool_rec != nullptr;
    for (int kh = 0; kh < d.KH; ++kh)
        for (int kw = 0; kw < d.KW; ++kw) {
            a.tap_dh[kh * d.KW + kw] = d.pad_h - kh * d.dil;
            a.tap_dw[kh * d.KW + kw] = d.pad_w - kw * d.dil;
            a.tap_w[kh * d.KW + kw] = kh * d.KW + kw;
        }
    const double fl = conv_flops(d), by = conv_bytes(d) + (mask ? 4.0 * d.B * d.Hi * d.Wi * d.Ci : 0.0);
    const int cfg = pick_tile(a.M, a.DN, MODE_DGRAD);
    static const int parity = env_int("SSD_DGRAD_PARITY", 1);      // A/B switch
    if (d.stride > 1 && parity && d.Co % 4 == 0) {
        // Strided data gradient by parity classes (conv8_2, conv9_2, vgg512 conv10_2).  An input pixel (ih, iw) only
        // meets the taps with (ih + pad - kh dil) divisible by the stride: gathering all taps for every pixel wastes
        // (stride^2 - 1) / stride^2 of the MFMAs on zero rows.  Per class (ih mod s, iw mod s) the pixels form a dense
        // (Hi/s x Wi/s) grid on which the remaining taps are a plain stride-1 gather of dy: one launch per class of the
        // unstrided kernel, with the filter tap and the real output pixel looked up through tap_w / out_*.
        const int sdiv = d.stride;
        SSD_REQUIRE(sdiv == 2, "conv_dgrad: parity classes are laid out for stride 2");
        GatherArgs c = a;
        c.div = 1; c.mul = 1; c.out_mul = sdiv; c.ODH = d.Hi; c.ODW = d.Wi;
        const int tile128 = pick_tile((long long)a.M / 4, a.DN, MODE_DGRAD) <= 1;
        const int bm = tile128 ? 128 : 64, NT = cdiv(a.DN, 128);
        c.nclass = 0;
        c.cls_wg0[0] = 0;
        for (int ph = 0; ph < sdiv; ++ph)
            for (int pw = 0; pw < sdiv; ++pw) {
                const int k = c.nclass;
                c.cls_DH[k] = (d.Hi - ph + sdiv - 1) / sdiv; c.cls_DW[k] = (d.Wi - pw + sdiv - 1) / sdiv;
                if (c.cls_DH[k] <= 0 || c.cls_DW[k] <= 0) continue;
                c.cls_M[k] = d.B * c.cls_DH[k] * c.cls_DW[k];
                c.cls_ph[k] = ph; c.cls_pw[k] = pw;
                int nt = 0;
                for (int kh = 0; kh < d.KH; ++kh)
                    for (int kw = 0; kw < d.KW; ++kw) {
                        const int nh = ph + d.pad_h - kh * d.dil, nw = pw + d.pad_w - kw * d.dil;
                        if (nh % sdiv != 0 || nw % sdiv != 0) continue;       // (C++ % keeps the sign: 0 stays 0)
                        c.cls_dh[k][nt] = nh / sdiv; c.cls_dw[k][nt] = nw / sdiv; c.cls_w[k][nt] = kh * d.KW + kw;
                        ++nt;
                    }
                c.cls_ntaps[k] = nt;
                c.cls_wg0[k + 1] = c.cls_wg0[k] + cdiv(c.cls_M[k], bm) * NT;
                ++c.nclass;
            }
        // single-class fields = class 0 (the kernel patches them per workgroup when nclass > 1)
        c.M = c.cls_M[0]; c.DH = c.cls_DH[0]; c.DW = c.cls_DW[0]; c.ntaps = c.cls_ntaps[0]; c.out_ph = c.cls_ph[0]; c.out_pw = c.cls_pw[0];
        for (int t = 0; t < 9; ++t) { c.tap_dh[t] = c.cls_dh[0][t]; c.tap_dw[t] = c.cls_dw[0][t]; c.tap_w[t] = c.cls_w[0][t]; }
        if (tile128) launch_gather_dma<MODE_DGRAD, 2, 2, 2, 2, false, true>(c, "conv_dgrad_parity_128x128", fl, by, s, c.cls_wg0[c.nclass]);
        else launch_gather_dma<MODE_DGRAD, 2, 2, 1, 2, false, true>(c, "conv_dgrad_parity_64x128", fl, by, s, c.cls_wg0[c.nclass]);
    } else if (d.stride > 1 && (d.stride & (d.stride - 1)) == 0) {      // the all-taps strided kernel (SSD_DGRAD_PARITY=0)
        if (cfg == 0 || cfg == 1) launch_gather_dma<MODE_DGRAD, 2, 2, 2, 2, true>(a, "conv_dgrad_strided_128x128", fl, by, s);
        else launch_gather_dma<MODE_DGRAD, 2, 2, 1, 2, true>(a, "conv_dgrad_strided_64x128", fl, by, s);
    } else if (d.stride > 1) {      // other strides: the register-staged kernel
        if (cfg == 0 || cfg == 1) launch_gather<MODE_DGRAD, 2, 2, 2, 2, false, true>(a, "conv_dgrad_strided_128x128", fl, by, s);
        else launch_gather<MODE_DGRAD, 2, 2, 1, 1, false, true>(a, "conv_dgrad_strided_64x64", fl, by, s);
    } else {
        switch (cfg) {
        case 0: launch_gather_dma<MODE_DGRAD, 2, 2, 2, 2>(a, unpool ? "conv_dgrad_unpool_128x128" : "conv_dgrad_128x128", fl, by, s); break;
        case 1: launch_gather_dma<MODE_DGRAD, 4, 1, 1, 2>(a, unpool ? "conv_dgrad_unpool_128x64" : "conv_dgrad_128x64", fl, by, s); break;
        case 2: launch_gather_dma<MODE_DGRAD, 2, 2, 1, 2>(a, unpool ? "conv_dgrad_unpool_64x128" : "conv_dgrad_64x128", fl, by, s); break;
        default: launch_gather_dma<MODE_DGRAD, 2, 2, 1, 1>(a, unpool ? "conv_dgrad_unpool_64x64" : "conv_dgrad_64x64", fl, by, s); break;
        }
    }
}

void conv_dgrad(const ConvDesc& d, const float* dy, const float* w, float* dx, const float* mask, bool accumulate,
                hipStream_t s) {
    conv_dgrad_any(d, dy, w, dx, mask, accumulate, s);
}

void conv_dgrad_unpool(const ConvDesc& d, const float* dy, const float* w, float* dx_unpooled, const void* rec, int UH, int UW,
                       hipStream_t s) {
    SSD_REQUIRE(conv_dgrad_unpool_supported(d), "conv_dgrad_unpool: unsupported shape");
    SSD_REQUIRE(rec != nullptr && (UH + 1) / 2 == d.Hi && (UW + 1) / 2 == d.Wi, "conv_dgrad_unpool: record / pooled size mismatch");
    conv_dgrad_any(d, dy, w, dx_unpooled, nullptr, false, s, rec, UH, UW);
}

// ---- wgrad planning ---------------------------------------------------------------
struct WgradPlan {
    int cfg;        // (channels x n) 0: 128x128, 1: 64x64, 2: 64x128, 3: 128x64
    int bkt, bnt, CT, NT, tiles, nsplit, mchunk;
    bool smallc;
};

static WgradPlan plan_wgrad(const ConvDesc& d) {
    WgradPlan p{};
    const int M = d.B * d.Ho * d.Wo;
    p.smallc = d.Ci % 4 != 0;
    const int waste128 = cdiv(d.Co, 128) * 128 - d.Co, waste64 = cdiv(d.Co, 64) * 64 - d.Co;
    if (p.smallc || (d.Ci <= 64 && d.Co <= 64)) p.cfg = 1;
    else if (d.Ci <= 64) p.cfg = 2;
    else if (waste64 < waste128) p.cfg = 3;      // fused heads: Co = 100 / 152
    else if (M < 100000) {
        // conv4 and deeper: 3 workgroups / CU.  Round 1 took 128x64 tiles (116 vs 107 TFLOP/s on 128x128); re-measured on the
        // round-2 kernels (tools/bench_conv.py sweep, gpurun r02): 64 input channels x 128 output channels is 1-5 % faster on
        // every one of these layers (conv4_1 0.923 -> 0.907, conv4_2 1.800 -> 1.782, conv5_2 0.498 -> 0.475, mod_conv6
        // 0.954 -> 0.907 ms).
        p.cfg = 2;
    }
    else p.cfg = 0;
    static const int forced = env_int("SSD_WGRAD_CFG", -1);      // tuning override
    if (forced >= 0 && forced < 4 && !p.smallc) p.cfg = forced;
    p.bkt = (p.cfg == 0 || p.cfg == 3) ? 128 : 64;
    p.bnt = (p.cfg == 1 || p.cfg == 3) ? 64 : 128;
    p.CT = p.smallc ? 1 : cdiv(d.Ci, p.bkt);
    p.NT = cdiv(d.Co, p.bnt);
    const int taps = p.smallc ? 1 : d.KH * d.KW;
    p.tiles = taps * p.CT * p.NT;
    int want = cdiv(1536, p.tiles);
    if (want > 256) want = 256;          // the reduce pass reads every slab: keep it short
    int maxs = cdiv(M, 256);
    want = want < 1 ? 1 : (want > maxs ? maxs : want);
    // Round quantisation: the grid runs in rounds of `slots` resident workgroups (LDS-limited: 2 per CU with
    // 128x128 tiles, 3 with 128x64 / 64x128, 4 with 64x64).  Among the split counts near `want`, take the one
    // whose last round is fullest (conv4_x at batch 32: 6 splits = 2.25 rounds -> 8 splits = 3.0 rounds).
    if (!p.smallc && p.tiles * want > 256) {
        const int slots = 256 * (p.cfg == 0 ? 2 : (p.cfg == 1 ? 4 : 3));
        int best = want;
        double best_fill = 0.0;
        const int lo = std::max(1, want - want / 3), hi = std::min(maxs, want + (want + 1) / 2);
        for (int ns = lo; ns <= hi; ++ns) {
            const int wgs = p.tiles * ns;
            const double fill = (double)wgs / ((double)cdiv(wgs, slots) * slots);
            if (fill > best_fill + 0.02) { best_fill = fill; best = ns; }
        }
        want = best;
    }
    p.nsplit = want;
    p.mchunk = cdiv(cdiv(M, p.nsplit), 32) * 32;
    p.nsplit = cdiv(M, p.mchunk);
    return p;
}

static bool use_first_wgrad(const ConvDesc& d) {
    static const int on = env_int("SSD_FIRST_WGRAD_F32", 1);        // A/B switch
    return on && conv_first_wgrad_f32_applicable(d);
}

size_t conv_wgrad_ws_floats(const ConvDesc& d) {
    WgradPlan p = plan_wgrad(d);
    size_t n = (size_t)p.nsplit * ((size_t)d.KH * d.KW * d.Ci * d.Co + d.Co);
    if (conv_first_wgrad_f32_applicable(d)) n = std::max(n, conv_first_wgrad_f32_ws_floats(d));
    return n;
}

template <int WM, int WN, int TM, int TN, bool SMALLC, bool YBF = false>
static void launch_wgrad(WgradArgs& a, const WgradPlan& pl, const char* label, double flops, double bytes, hipStream_t s) {
    constexpr int BKT = 32 * TM * WM, BNT = 32 * TN * WN;
    constexpr size_t lds = 2 * (size_t)(32 * BKT + 32 * BNT) * sizeof(float);
    auto kern = conv_wgrad_kernel<WM, WN, TM, TN, SMALLC, YBF>;
    static bool once = (set_lds(kern, lds), true);
    (void)once;
    ProfScope prof(label, flops, bytes, s);
    hipLaunchKernelGGL(kern, dim3(pl.nsplit * pl.tiles), dim3(256), lds, s, a);
    HIP_OK(hipGetLastError());
}

template <int WM, int WN, int TM, int TN>
static void launch_wgrad_dma(WgradArgs& a, const WgradPlan& pl, const char* label, double flops, double bytes, hipStream_t s) {
    constexpr int BKT = 32 * TM * WM, BNT = 32 * TN * WN;
    constexpr size_t lds = 2 * (size_t)(32 * BKT + 32 * BNT) * sizeof(float);
    auto kern = conv_wgrad_dma_kernel<WM, WN, TM, TN>;
    static bool once = (set_lds(kern, lds), true);
    (void)once;
    ProfScope prof(label, flops, bytes, s);
    hipLaunchKernelGGL(kern, dim3(pl.nsplit * pl.tiles), dim3(256), lds, s, a);
    HIP_OK(hipGetLastError());
}

// ---- the slabs of ONE layer when there are hundreds of them (conv1_1): `lanes` threads share a float4 element --------------
struct ManySlabsArgs {
    const float* ws;
    const float* w;
    float* dw;
    float* db;
    unsigned long long wcount;
    int nsplit, Co;
    float wd;
    int lanes;          // threads sharing one float4 element
};

// A workgroup owns 256 / lanes consecutive float4 elements; `lanes` threads share an element, each summing every
// lanes-th slab in ascending order; the partials are added in lane order through LDS (fixed order).
__global__ __launch_bounds__(256) void wgrad_reduce_many_kernel(ManySlabsArgs it) {
    __shared__ f32x4 part[256];
    const int lanes = it.lanes, per = 256 / lanes;
    const int e = threadIdx.x % per, sl = threadIdx.x / per;
    const size_t total = it.wcount + it.Co;
    const size_t idx = ((size_t)blockIdx.x * per + e) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (idx < total) s = slab_sum(it.ws, total, idx, sl, lanes, it.nsplit);
    part[sl * per + e] = s;
    __syncthreads();
    if (sl == 0 && idx < total) {
        f32x4 t = part[e];
        for (int q = 1; q < lanes; ++q) t += part[q * per + e];
        if (idx < it.wcount) {
            if (it.wd != 0.f) t += it.wd * ld4(it.w + idx);
            *reinterpret_cast<f32x4*>(it.dw + idx) = t;
        } else if (it.db) {
            *reinterpret_cast<f32x4*>(it.db + (idx - it.wcount)) = t;
        }
    }
}

void wgrad_reduce(const float* ws, int nsplit, size_t wcount, int Co, float* dw, float* db, const float* w, float wd,
                  hipStream_t s) {
    if (ablated("reduce")) return;      // measurement aid (tools/step_time.py, ssd_debug_set_ablate)
    const size_t total = wcount + Co;
    if (nsplit >= 256) {        // hundreds of slabs of a small filter (conv1_1): 64 threads share an element
        const ManySlabsArgs it{ws, w, dw, db, (unsigned long long)wcount, nsplit, Co, wd, 64};
        ProfScope prof("wgrad_reduce_many", 0.0, 4.0 * (double)total * (nsplit + 2), s);
        hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3(cdiv((long long)(total / 4), 256 / it.lanes)), dim3(256), 0, s, it);
        HIP_OK(hipGetLastError());
        return;
    }
    int blocks = cdiv((long long)total, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    ProfScope prof("wgrad_reduce", 0.0, 4.0 * (double)total * (nsplit + 2), s);
    if (nsplit >= 32)
        hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3(cdiv((long long)total, 64)), dim3(256), 0, s, ws, nsplit, wcount, Co,
                           dw, db, w, wd);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, nsplit, wcount, Co, dw, db, w, wd);
    HIP_OK(hipGetLastError());
}

static void conv_wgrad_any(const ConvDesc& d, const float* x, const void* dy, bool dy_bf16, float* dw, float* dbias,
                           const float* w, float weight_decay, float* ws, hipStream_t s) {
    check_desc(d);
    WgradPlan pl = plan_wgrad(d);
    SSD_REQUIRE(!dy_bf16 || pl.smallc, "bf16 dy with fp32 x: only the packed small-C layer (conv1_1)");
    WgradArgs a{};
    a.x = x; a.dy = static_cast<const float*>(dy); a.ws = ws;
    a.M = d.B * d.Ho * d.Wo; a.Hi = d.Hi; a.Wi = d.Wi; a.Ci = d.Ci; a.Ho = d.Ho; a.Wo = d.Wo; a.Co = d.Co;
    a.ntaps = d.KH * d.KW; a.stride = d.stride; a.CT = pl.CT; a.NT = pl.NT;
    a.mchunk = pl.mchunk; a.nsplit = pl.nsplit;
    a.ntaps_grid = pl.smallc ? 1 : a.ntaps;
    for (int kh = 0; kh < d.KH; ++kh)
        for (int kw = 0; kw < d.KW; ++kw) {
            a.tap_dh[kh * d.KW + kw] = kh * d.dil - d.pad_h;
            a.tap_dw[kh * d.KW + kw] = kw * d.dil - d.pad_w;
        }
    const double fl = conv_flops(d), by = conv_bytes(d);
    if (pl.smallc && dy_bf16) launch_wgrad<2, 2, 1, 1, true, true>(a, pl, "conv_wgrad_smallc_64x64_bf16dy", fl, by, s);
    else if (pl.smallc) launch_wgrad<2, 2, 1, 1, true>(a, pl, "conv_wgrad_smallc_64x64", fl, by, s);
    // LDS-DMA staging on the 64-channel tiles (conv1_2 86 -> 95 TF/s: their x rows are re-read by 9 taps and the staging
    // registers were the occupancy limit); the 128-wide tiles measured 0..-3 % with it and keep the register-staged kernel
    else if (pl.cfg == 1) launch_wgrad_dma<2, 2, 1, 1>(a, pl, "conv_wgrad_64x64", fl, by, s);
    else if (pl.cfg == 2) launch_wgrad_dma<2, 2, 1, 2>(a, pl, "conv_wgrad_64x128", fl, by, s);
    else if (pl.cfg == 3) launch_wgrad<2, 2, 2, 1, false>(a, pl, "conv_wgrad_128x64", fl, by, s);
    else launch_wgrad<2, 2, 2, 2, false>(a, pl, "conv_wgrad_128x128", fl, by, s);

    wgrad_reduce(ws, pl.nsplit, (size_t)a.ntaps * d.Ci * d.Co, d.Co, dw, dbias, w, weight_decay, s);
}

void conv_wgrad(const ConvDesc& d, const float* x, const float* dy, float* dw, float* dbias, const float* w,
                float weight_decay, float* ws, hipStream_t s) {
    if (use_first_wgrad(d)) {
        check_desc(d);
        conv_first_wgrad_f32(d, x, dy, dw, dbias, w, weight_decay, ws, s);
        return;
    }
    conv_wgrad_any(d, x, dy, false, dw, dbias, w, weight_decay, ws, s);
}

void conv_wgrad_smallc_bf16dy(const ConvDesc& d, const float* x, const bf16_t* dy, float* dw, float* dbias, const float* w,
                              float weight_decay, float* ws, hipStream_t s) {
    conv_wgrad_any(d, x, dy, true, dw, dbias, w, weight_decay, ws, s);
}

}  // namespace ssd
