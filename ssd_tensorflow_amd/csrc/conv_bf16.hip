// im2col-free direct convolution for gfx950 (MI355X), bf16 in / fp32 accumulate on the matrix
// cores (v_mfma_f32_32x32x16_bf16) -- BASELINE.json configs[2] ("bf16 MFMA").  Same gather-GEMM
// formulation as conv_igemm.hip (forward and data gradient are one kernel, the weight gradient a
// second one); what changes is how tiles reach the LDS: a bf16 MFMA retires 8x the k of the fp32
// one per cycle, so staging through registers (address VALU + ds_write per 16 bytes) would bound the
// kernel.  Every tile is therefore filled by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging
// registers, no ds_write, zero fill of the padding by steering a lane's offset out of the buffer's
// range (tools/probes/bf16_probe.hip pins all three behaviours on the hardware).
//
// LDS-DMA writes lane-linear (base + lane*16), so a tile row is exactly its bytes, unpadded; bank
// conflicts are removed by choosing WHICH 16-byte chunk of the global row a lane fetches:
//   gather tiles  [rows][64 k] (128-byte rows): LDS slot p of row r holds chunk p ^ ((r>>1)&7);
//                 a ds_read_b128 lane group (16 rows, MI355X_MICROARCH.md) then covers 16 distinct slots;
//   wgrad tiles   [64 pixels][channels] (natural NHWC rows): slot p of pixel row r holds chunk
//                 p ^ 4*(r&3) (256-byte rows) or p ^ 4*((r>>1)&1) (128-byte rows), read by
//                 ds_read_b64_tr_b16, which hands every lane 4 consecutive PIXELS of its channel:
//                 the k-major MFMA operand straight from the pixel-major image, no transpose pass.
//
// Filters: fp32 masters live in the parameter arena; cast_filters() mirrors them per step as bf16
// in both orientations a k-contiguous B operand needs: [tap][Co][Ci] (forward) and [tap][Ci][Co]
// (data gradient; the arena's own HWIO order).
#include "conv.h"
#include "conv_detail.h"
#ifndef GATHER_PF
#define GATHER_PF 2
#endif
#include "bf16.h"
#include <algorithm>

namespace ssd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int HBK = 64;                 // k elements per iteration: one 128-byte line per tile row
constexpr unsigned OOBH = 0xFFFFFFF0u;  // offset no buffer covers: the load returns / the DMA writes zeros
enum { MODE_FWD = 0, MODE_DGRAD = 1 };

struct GatherArgsH {
    const bf16_t* src;
    const bf16_t* wgt;      // [tap][DN rows][SC k] bf16, k contiguous
    const float* bias;
    const bf16_t* mask;
    void* dst;              // bf16 [M][DN], or fp32 when out_f32
    int M, DH, DW, DN;
    int SH, SW, SC;
    int ntaps, mul, div;
    int relu, accum, out_f32;
    int NT;
    int tap_dh[9], tap_dw[9];
    // parity classes of a strided data gradient (per-tap kernel, PARITY instantiation; see conv_igemm.hip): class c owns
    // the workgroups [cls_wg0[c], cls_wg0[c + 1]), gathers cls_ntaps[c] of the filter's `wtaps` taps (cls_w = filter tap)
    // on its own dense pixel grid cls_DH x cls_DW and writes the real pixels (b, 2 a + ph, 2 c + pw) of an ODH x ODW image
    int wtaps, nclass, ODH, ODW;
    int cls_wg0[5], cls_M[4], cls_DH[4], cls_DW[4], cls_ntaps[4], cls_ph[4], cls_pw[4];
    int cls_dh[4][9], cls_dw[4][9], cls_w[4][9];
    // Round 5, the 2x2 pools fused into their neighbours (conv.h).  Data gradient: unpool_rec != nullptr routes every pooled
    // pixel's dx through the pool's record into the four cells of the pool's input gradient [.][UH][UW][DN] (gather_epilogue).
    // Forward (conv_fwd_pool_bf16_*_kernel): 2-D tiles of R image rows x TW columns (+ a one-column halo either side) of ONE
    // image, NSEG column segments x NBAND row bands per image; pooled tensor [PB][PH][PW][DN] + record.
    const unsigned short* unpool_rec;
    int UH, UW;
    bf16_t* pool_dst;
    unsigned short* pool_rec;
    int PB, PH, PW, TW, NSEG, NBAND;
};
struct OutMap { int M, DH, DW, ph, pw, ODH, ODW; };      // parity class: virtual pixel (b, a, c) -> real pixel (b, 2 a + ph, 2 c + pw)

// Pipeline step: wait until all but the `ahead` most recent tiles of this lane's LDS-DMA have landed
// (L DMA instructions per tile; vmcnt retires loads in order), then the workgroup barrier publishes them.
// A raw s_barrier, not __syncthreads(): the latter's fence drains vmcnt to 0 and with it the tiles that
// are meant to stay in flight across the barrier.  Every ds_read of the previous tile has already been
// consumed by an MFMA (lgkmcnt) when a wave arrives here, so the stage it frees can be refilled at once.
template <int L, int MAXA = 2>
__device__ __forceinline__ void wait_tiles_and_sync(int ahead) {
    static_assert(2 * L <= 63 && MAXA * L <= 63, "vmcnt field");
    // (deep rings of the small-layer configurations: up to MAXA tiles stay in flight; the branches are wave-uniform)
    if constexpr (MAXA >= 4) {
        if (ahead >= 4) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * L) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            return;
        }
    }
    if constexpr (MAXA >= 3) {
        if (ahead == 3) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * L) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            return;
        }
    }
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Epilogue of both gather kernels, through an fp32 LDS tile [BM][BN + 4]: full-line loads / stores, fused bias + relu
// (forward) or accumulate + relu mask (data gradient), ONE rounding to bf16.  The caller has passed a barrier after its
// last LDS read.
// HALVES = 2: the C tile holds half of every wave's rows at a time (a 256-row tile through 66 KB of LDS);
// rows_valid: tile rows that belong to this workgroup (the kernel-row 256-pixel tile owns 256 - 2 dil of its rows).
// ACC_IN_LDS: the C tile already holds the sums (the k-split instantiation of the per-tap kernel adds its wave groups'
// accumulators there): no accumulator write, no barrier -- only the calling threads take part.
template <int MODE, int WM, int WN, int TM, int TN, int HALVES = 1, bool PARITY = false, bool ACC_IN_LDS = false>
__device__ __forceinline__ void gather_epilogue(const GatherArgsH& p, unsigned char* smem, const f32x16 (&acc)[TM][TN], int tid, int wm,
                                                int wn, int li, int lh, int m0, int n0, int rows_valid = 32 * TM * WM,
                                                const OutMap* om = nullptr) {
    constexpr int NTHR = 64 * WM * WN, TMH = TM / HALVES, BMH = 32 * TMH * WM, BN = 32 * TN * WN, LDC = BN + 4;
    static_assert(TM % HALVES == 0, "whole accumulator tiles per pass");
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int TPR = BN / 8;               // threads per row, 8 channels each
    constexpr int RPP = NTHR / TPR;           // rows per pass
    const int cg = tid % TPR, r0 = tid / TPR;
    const int n = n0 + cg * 8;
    const bool ncol = n < p.DN;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (MODE == MODE_FWD && p.bias && ncol) ? p.bias[n + e] : 0.f;
    constexpr int NPS = BMH / RPP;            // rows of this thread per pass over the C tile
    // pixel of C-tile row ml in pass h (-1: nothing to write): wave row block ml / (32 TMH), pass h inside the wave's 32 TM rows
    auto row_offset = [&](int h, int ml, bool& ok) -> size_t {
        const int tr = HALVES == 1 ? ml : (ml / (32 * TMH)) * (32 * TM) + h * (32 * TMH) + ml % (32 * TMH);
        const int m = m0 + tr;
        ok = ncol && tr < rows_valid && m < (PARITY ? om->M : p.M);
        if (!ok) return 0;
        size_t pix = (size_t)m;
        if constexpr (PARITY) {
            const int c = m % om->DW, t2 = m / om->DW;
            const int a = t2 % om->DH, b = t2 / om->DH;
            pix = ((size_t)b * om->ODH + 2 * a + om->ph) * om->ODW + 2 * c + om->pw;
        }
        return pix * p.DN + n;
    };
#pragma unroll
  for (int h = 0; h < HALVES; ++h) {
    // data gradient: the relu mask / old values of this thread's rows are requested BEFORE the accumulators take their trip
    // through LDS (two barriers), all of them at once: fetched row by row inside the write-out loop, every row waited for
    // its own round trip (8 per tile: as long as the main loop of a 64- or 128-channel layer)
    u32x4 pre_mask[NPS], pre_old[NPS];
    unsigned pre_rec[NPS];
    if constexpr (MODE != MODE_FWD && !PARITY) {
        if (p.unpool_rec) {      // the pool's record of this thread's 8 channels: two 12-bit entries
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                bool ok;
                const size_t o = row_offset(h, r0 + ps * RPP, ok);      // = pooled pixel * DN + n (0 when there is nothing to write)
                pre_rec[ps] = *reinterpret_cast<const unsigned*>(p.unpool_rec + (o >> 2));
            }
        }
    }
    if constexpr (MODE != MODE_FWD) {
        if (p.mask) {
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                bool ok;
                const size_t o = row_offset(h, r0 + ps * RPP, ok);      // (row 0 of the tensor when there is nothing to write)
                pre_mask[ps] = *reinterpret_cast<const u32x4*>(p.mask + o);
            }
        }
        if (p.accum) {
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                bool ok;
                const size_t o = row_offset(h, r0 + ps * RPP, ok);
                pre_old[ps] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.dst) + o);
            }
        }
    }
    if constexpr (!ACC_IN_LDS) {
    if (h) __syncthreads();                   // the previous pass has been read out
#pragma unroll
    for (int mi = 0; mi < TMH; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ml = wm * 32 * TMH + mi * 32 + li;
                const int nl = wn * 32 * TN + ni * 32 + 8 * g + 4 * lh;
                const f32x16& c = acc[h * TMH + mi][ni];
                *reinterpret_cast<f32x4*>(Cs + ml * LDC + nl) = f32x4{c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
            }
    __syncthreads();
    }
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
        const int ml = r0 + ps * RPP;
        bool ok;
        const size_t o = row_offset(h, ml, ok);
        if (!ok) continue;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(Cs + ml * LDC + cg * 8);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(Cs + ml * LDC + cg * 8 + 4);
        float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        if constexpr (MODE == MODE_FWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] += bv[e];
                if (p.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
        } else {
            if (p.accum) {
                const u32x4 old = pre_old[ps];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += lo2f(old[e]);
                    v[2 * e + 1] += hi2f(old[e]);
                }
            }
            if (p.mask) {
                const u32x4 y = pre_mask[ps];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = lo2f(y[e]) > 0.f ? v[2 * e] : 0.f;
                    v[2 * e + 1] = hi2f(y[e]) > 0.f ? v[2 * e + 1] : 0.f;
                }
            }
        }
        if constexpr (MODE != MODE_FWD && !PARITY) {
            if (p.unpool_rec) {
                // pooled pixel m = (b, oh, ow) -> cells (2 oh + (q >> 1), 2 ow + (q & 1)) of the [UH][UW] image: the recorded first
                // maximum gets dx if that maximum was positive (relu of the pooled tensor's producer), every other cell zero
                const int m = (int)(o / (size_t)p.DN);
                const int ow = m % p.DW, t2 = m / p.DW;
                const int oh = t2 % p.DH, b = t2 / p.DH;
                const bool okq[4] = {true, 2 * ow + 1 < p.UW, 2 * oh + 1 < p.UH, 2 * ow + 1 < p.UW && 2 * oh + 1 < p.UH};
                const unsigned rw = pre_rec[ps];
                bf16_t* d0 = reinterpret_cast<bf16_t*>(p.dst) + ((size_t)(b * p.UH + 2 * oh) * p.UW + 2 * ow) * p.DN + n;
                // Rounded ONCE, then routed as packed pairs: an entry re = sign << 2 | cell becomes a one-hot nibble over the four
                // cells (0 when the maximum was not positive) through a 32-bit table, a pair's two nibbles sit 16 bits apart, and
                // cell q's store mask of the pair is (bit q, bit 16 + q) x 0xFFFF -- four VALU operations per pair and cell where
                // eight selects on floats + a conversion per cell cost ~13 (the 256 x 64 tile of conv2_1 spent more issue slots
                // in this epilogue than in its 18 k steps: profiles/r06_b_tail_chain_v1_per_layer_bf16.txt, 572 TFLOP/s).
                unsigned pk[4], oh2[4];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    pk[e2] = pack2(v[2 * e2], v[2 * e2 + 1]);
                    const int s0 = 16 * (e2 / 2) + 6 * (e2 & 1), s1 = s0 + 3;      // bit offsets of the pair's two entries
                    const unsigned i0 = s0 >= 2 ? (rw >> (s0 - 2)) & 0x1Cu : (rw << 2) & 0x1Cu;      // 4 x entry
                    const unsigned i1 = (rw >> (s1 - 2)) & 0x1Cu;
                    oh2[e2] = ((0x84210000u >> i0) & 0xFu) | ((0x84210000u >> i1) << 16);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!okq[q]) continue;
                    u32x4 o4;
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) o4[e2] = pk[e2] & (((oh2[e2] >> q) & 0x00010001u) * 0xFFFFu);
                    *reinterpret_cast<u32x4*>(d0 + ((size_t)(q >> 1) * p.UW + (q & 1)) * p.DN) = o4;
                }
                continue;
            }
        }
        if (p.out_f32) {
            float* d = reinterpret_cast<float*>(p.dst) + o;
            *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.dst) + o) =
                u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        }
    }
  }
}

// KSPLIT > 1 (round 4, the latency-bound small layers: conv8_2 ... conv11_2, the small maps' heads -- a handful of tiles, each
// a serial loop of 18 ... 72 k iterations that is one DMA round trip after the other): KSPLIT groups of WM x WN waves share the
// tile; group g multiplies the iterations g, g + KSPLIT, ... from its own ring of stages, so the loop is KSPLIT times shorter
// with KSPLIT times the bytes in flight; the groups then add their accumulators in the fp32 C tile IN GROUP ORDER (a fixed
// summation order: results do not depend on timing) and the first group writes the tile out.
template <int MODE, int WM, int WN, int TM, int TN, bool STRIDED, int NS, bool PARITY = false, int KSPLIT = 1>
__global__ __launch_bounds__(64 * WM * WN * KSPLIT) void conv_gather_bf16_kernel(GatherArgsH pp) {
    // (no local copy of the argument block: dynamically indexed arrays of a copy would live in scratch)
    const GatherArgsH& p = pp;
    int wg_first = 0, wg_count = gridDim.x;
    int P_M = pp.M, P_DH = pp.DH, P_DW = pp.DW, P_ntaps = pp.ntaps;
    const int* P_tap_dh = pp.tap_dh;
    const int* P_tap_dw = pp.tap_dw;
    const int* P_tap_w = nullptr;                       // filter tap of gathered tap k (identity unless PARITY)
    OutMap om{};
    if constexpr (PARITY) {                             // parity classes of a strided data gradient, one launch
        int c = 0;
        for (int k = 1; k < pp.nclass; ++k)
            if ((int)blockIdx.x >= pp.cls_wg0[k]) c = k;
        wg_first = pp.cls_wg0[c]; wg_count = pp.cls_wg0[c + 1] - wg_first;
        P_M = pp.cls_M[c]; P_DH = pp.cls_DH[c]; P_DW = pp.cls_DW[c]; P_ntaps = pp.cls_ntaps[c];
        P_tap_dh = pp.cls_dh[c]; P_tap_dw = pp.cls_dw[c]; P_tap_w = pp.cls_w[c];
        om = OutMap{P_M, P_DH, P_DW, pp.cls_ph[c], pp.cls_pw[c], pp.ODH, pp.ODW};
    }
    constexpr int NTHR = 64 * WM * WN;                // 4 or 8 waves (per k-split group)
    constexpr int RPP_S = NTHR / 8;                   // tile rows one staging pass of the group covers
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int A_N = BM / RPP_S, B_N = BN / RPP_S; // DMA instructions per thread and tile
    constexpr int STAGE = (BM + BN) * 128;            // bytes per pipeline stage
    constexpr int LDC = BN + 4;                       // fp32 epilogue tile pitch
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
    static_assert(BM % RPP_S == 0 && BN % RPP_S == 0, "tile vs staging pass");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int wave_all = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);      // scalar: LDS-DMA bases stay in SGPRs
    const int grp = KSPLIT > 1 ? wave_all / (WM * WN) : 0;                              // k-split group of this wave
    const int wave = KSPLIT > 1 ? wave_all - grp * (WM * WN) : wave_all, lane = threadIdx.x & 63;
    const int tid = KSPLIT > 1 ? wave * 64 + lane : (int)threadIdx.x;                   // thread index inside the group
    unsigned char* const gsm = smem + grp * (NS * STAGE);                               // the group's ring of stages
    const int wg = xcd_remap(blockIdx.x - wg_first, wg_count);
    const int mt = wg / p.NT, nt = wg - mt * p.NT;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- staging: thread -> rows (tid>>3) + RPP_S i, LDS slot tid&7, global chunk slot ^ swizzle(row)
    const int a_ck = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;      // k element offset inside the 64-wide block
    int rb[A_N], rh[A_N], rw[A_N];
    unsigned a_off[A_N], a_msk[A_N];
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
        const int m = m0 + (tid >> 3) + RPP_S * i;
        const int mm = m < P_M ? m : 0;
        const int ow = mm % P_DW;
        const int t2 = mm / P_DW;
        const int oh = t2 % P_DH;
        const int b = t2 / P_DH;
        rb[i] = b * p.SH * p.SW;
        rh[i] = m < P_M ? oh * p.mul : -(1 << 20);
        rw[i] = ow * p.mul;
        a_off[i] = (unsigned)((rb[i] + rh[i] * p.SW + rw[i]) * p.SC + a_ck) * 2u;
        unsigned mk = 0;
        if constexpr (!STRIDED) {
            for (int t = 0; t < P_ntaps; ++t) {
                const int sh = rh[i] + P_tap_dh[t], sw = rw[i] + P_tap_dw[t];
                if ((unsigned)sh < (unsigned)p.SH && (unsigned)sw < (unsigned)p.SW) mk |= 1u << t;
            }
        }
        a_msk[i] = mk;
    }
    unsigned b_off[B_N], b_ok[B_N];      // byte offset of (row n, chunk) inside one tap's filter image; row inside the filter
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
        const int n = n0 + (tid >> 3) + RPP_S * i;
        b_ok[i] = 0u - (unsigned)(n < p.DN);
        b_off[i] = (unsigned)((n < p.DN ? n : 0) * p.SC + a_ck) * 2u;
    }
    const __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.src), 0, (unsigned)((size_t)(P_M / (P_DH * P_DW)) * p.SH * p.SW * p.SC * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.wgt), 0, (unsigned)((size_t)(PARITY ? pp.wtaps : P_ntaps) * p.DN * p.SC * 2u), 0x00020000);

    const int nchunks = (p.SC + HBK - 1) / HBK;
    const int nk = nchunks * P_ntaps;

    auto issue = [&](int kiter, int stage) {
        const int cc = kiter / P_ntaps;
        const int tap = kiter - cc * P_ntaps;
        unsigned char* As = gsm + stage * STAGE + wave * 1024;        // wave-uniform: 8 rows x 128 B per DMA
        unsigned char* Bs = As + BM * 128;
        const unsigned cmask = 0u - (unsigned)(cc * HBK + a_ck < p.SC);
        if constexpr (!STRIDED) {
            const unsigned toff = (unsigned)(((P_tap_dh[tap] * p.SW + P_tap_dw[tap]) * p.SC + cc * HBK) * 2);
#pragma unroll
            for (int i = 0; i < A_N; ++i) {
                const unsigned m = (0u - ((a_msk[i] >> tap) & 1u)) & cmask;
                const unsigned off = ((a_off[i] + toff) & m) | (OOBH & ~m);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, LDS_PTR(As + i * (RPP_S * 128)), 16, off, 0, 0, 0);
            }
        } else {
            const int dh = P_tap_dh[tap], dw = P_tap_dw[tap];
#pragma unroll
            for (int i = 0; i < A_N; ++i) {
                int sh = rh[i] + dh, sw = rw[i] + dw;
                bool ok = (sh % p.div == 0) && (sw % p.div == 0);
                sh /= p.div;
                sw /= p.div;
                ok = ok && (unsigned)sh < (unsigned)p.SH && (unsigned)sw < (unsigned)p.SW;
                const unsigned m = (0u - (unsigned)ok) & cmask;
                const unsigned off = ((unsigned)(((rb[i] + sh * p.SW + sw) * p.SC + cc * HBK + a_ck) * 2) & m) | (OOBH & ~m);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, LDS_PTR(As + i * (RPP_S * 128)), 16, off, 0, 0, 0);
            }
        }
        const unsigned woff = (unsigned)(((PARITY ? P_tap_w[tap] : tap) * p.DN * p.SC + cc * HBK) * 2);
#pragma unroll
        for (int i = 0; i < B_N; ++i) {
            const unsigned m = cmask & b_ok[i];
            const unsigned off = ((b_off[i] + woff) & m) | (OOBH & ~m);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wgt_rsrc, LDS_PTR(Bs + i * (RPP_S * 128)), 16, off, 0, 0, 0);
        }
    };

    // ---- accumulators: D rows = output channels (filter operand first), D cols = pixels, so a lane
    // ends up with 4 consecutive channels of one pixel per register quad
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, lh = lane >> 5;
    // fragment address inside a stage: row*128 + ((2s + lh) ^ swz(row))*16 = (row*128 + q) ^ (s*32)
    const int q0 = (lh ^ ((li >> 1) & 7)) * 16;
    const int a_row = (wm * 32 * TM + li) * 128 + q0;
    const int b_row = BM * 128 + (wn * 32 * TN + li) * 128 + q0;

    auto compute = [&](int stage) {
        const unsigned char* S = gsm + stage * STAGE;
        // fragments of k-steps st+1 .. st+PF-1 are read while the MFMAs of step st run (PF register sets)
        constexpr int PF = GATHER_PF, KS = HBK / 16;
        bf16x8 a[PF][TM], b[PF][TN];
        auto frags = [&](int st) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                a[st % PF][mi] = *reinterpret_cast<const bf16x8*>(S + ((a_row + mi * 4096) ^ (st * 32)));
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                b[st % PF][ni] = *reinterpret_cast<const bf16x8*>(S + ((b_row + ni * 4096) ^ (st * 32)));
        };
#pragma unroll
        for (int st = 0; st < PF - 1 && st < KS; ++st) frags(st);
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            if (st + PF - 1 < KS) frags(st + PF - 1);
            // keep the order written here: left alone, the scheduler folds the register sets into one and every
            // k-step then starts parked on its own LDS reads (SQ_WAIT_ANY was 34 % of the wave cycles)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[st % PF][ni], a[st % PF][mi], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- main loop: NS stages; tiles k+1 .. k+NS-1 stream in while tile k is multiplied --------
    if constexpr (KSPLIT == 1) {
#pragma unroll
        for (int t = 0; t < NS - 1; ++t)
            if (t < nk) issue(t, t);
        int st_c = 0, st_i = NS - 1;
        for (int k = 0; k < nk; ++k) {
            const int later = nk - 1 - k;
            wait_tiles_and_sync<A_N + B_N, (NS - 2 > 4 ? 4 : (NS - 2 < 2 ? 2 : NS - 2))>(later < NS - 2 ? later : NS - 2);      // tile k visible; stage st_i is free
            if (k + NS - 1 < nk) issue(k + NS - 1, st_i);
                compute(st_c);
                st_c = st_c + 1 == NS ? 0 : st_c + 1;
            st_i = st_i + 1 == NS ? 0 : st_i + 1;
        }
        __syncthreads();
        gather_epilogue<MODE, WM, WN, TM, TN, 1, PARITY>(p, smem, acc, tid, wm, wn, li, lh, m0, n0, 32 * TM * WM, &om);
    } else {
        // group g owns the iterations g, g + KSPLIT, ...: local iteration j is global iteration grp + j KSPLIT.  Every group runs
        // the same number of loop trips (the barrier is the workgroup's); a trip past a group's last iteration moves nothing.
        const int nkg = (nk - grp + KSPLIT - 1) / KSPLIT;      // this group's iterations (may be 0)
        const int trips = (nk + KSPLIT - 1) / KSPLIT;
#pragma unroll
        for (int t = 0; t < NS - 1; ++t)
            if (t < nkg) issue(grp + t * KSPLIT, t);
        int st_c = 0, st_i = NS - 1;
        for (int j = 0; j < trips; ++j) {
            const int later = nkg - 1 - j;
            wait_tiles_and_sync<A_N + B_N, (NS - 2 > 4 ? 4 : (NS - 2 < 2 ? 2 : NS - 2))>(later < 0 ? 0 : (later < NS - 2 ? later : NS - 2));
            if (j + NS - 1 < nkg) issue(grp + (j + NS - 1) * KSPLIT, st_i);
                if (j < nkg) compute(st_c);
                st_c = st_c + 1 == NS ? 0 : st_c + 1;
            st_i = st_i + 1 == NS ? 0 : st_i + 1;
        }
        __syncthreads();
        // the groups' accumulators meet in the fp32 C tile, in group order
        float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll 1
        for (int g = 0; g < KSPLIT; ++g) {
            if (grp == g) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int ml = wm * 32 * TM + mi * 32 + li;
                            const int nl = wn * 32 * TN + ni * 32 + 8 * q + 4 * lh;
                            f32x4* dst = reinterpret_cast<f32x4*>(Cs + ml * LDC + nl);
                            f32x4 v = f32x4{acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                            if (g > 0) {
                                const f32x4 o = *dst;
                                v = f32x4{o[0] + v[0], o[1] + v[1], o[2] + v[2], o[3] + v[3]};
                            }
                            *dst = v;
                        }
            }
            __syncthreads();
        }
        if (grp == 0)
            gather_epilogue<MODE, WM, WN, TM, TN, 1, PARITY, true>(p, smem, acc, tid, wm, wn, li, lh, m0, n0, 32 * TM * WM, &om);
    }
}

// =================================================================================
// Kernel-row gather (3x3, stride 1, SAME, any dilation <= 8): one pipeline unit = one KERNEL ROW of one 64-channel
// chunk, i.e. the three taps dw = -dil, 0, +dil.  The per-tap kernel above pays one DMA round trip (~2700 cycles,
// issued -> landed) per 512 MFMA cycles; here a unit stages ONE activation tile with a dil-pixel halo on either side
// (the three shifted tiles of a stride-1 row are the same pixels) plus the three filter tiles, and multiplies for
// 1536 MFMA cycles: 21.7 KB instead of 32 KB staged per tap, a third of the barriers, the round trip amortised 3x.
// No double buffering (the unit fills the LDS budget of a 2-workgroups-per-CU kernel); the second workgroup of the CU
// computes while this one waits.  The halo tile is a run of CONSECUTIVE pixels of the NHWC tensor, loaded without
// looking at image geometry; what is not a neighbour (next image row, padding) is zeroed per lane when the operand is
// read (4 v_cndmask per fragment), from the same 9-bit tap masks the per-tap kernel uses for its DMA.
// =================================================================================
// TM = 4: 256-row tiles.  What a wave pays per DMA piece (60..185 cycles of its own issue time) is the same for every
// tile, so pieces per MFMA is the figure of merit: 17 per 48 MFMAs and wave with 128 rows, 20 per 96 with 256.  The
// activation tile is then exactly 256 rows = 32 KB, i.e. a workgroup owns 256 - 2 dil - 1 output pixels (halo + the empty row), so that two workgroups still fit a CU (2 x 80 KB); the fp32 epilogue tile goes through
// LDS in two halves.  Used where the tiles fill the chip at least twice (conv2_2, conv3_x).
// TN = 1: 128 x 64 tiles (45 KB of LDS: THREE workgroups per CU).  For the 19x19 maps at batch 32 (M = 11,552: conv5_x, mod_conv6,
// the 19x19 head) the 128 x 128 tiling gives 364 workgroups for 512 slots -- 108 CUs carry two, 148 carry one, the launch lasts as
// long as the doubly loaded ones (71 % fill, 800-850 TFLOP/s against 1,200 for conv4_x on the same kernel); 728 workgroups on 768
// slots fill 95 %.  Costs 26 % more staged bytes per MFMA, which the rows kernels are insensitive to.
template <int MODE, int TM, int TN = 2>
__global__ __launch_bounds__(256) void conv_gather_bf16_rows_kernel(GatherArgsH p, int dil) {
    constexpr int WM = 2, WN = 2, BM = 64 * TM, BN = 32 * TN * WN;
    static_assert(TN == 2 || TN == 1, "128- or 64-wide tiles");
    constexpr int AROWS = TM == 2 ? 160 : 256;         // 128 + 2 * dil rows in whole 32-row staging passes | 256 rows = 253 + 2 (dil = 1) + 1 empty
    constexpr int A_N = AROWS / 32, B_N = BN / 32;
    constexpr int ZROW = (AROWS - 1) * 128;            // a tile row that is always zero (past the halo)
    static_assert(TM == 2 || TM == 4, "128- or 256-row tiles");
    constexpr int A_BYTES = AROWS * 128, UNIT = A_BYTES + 3 * BN * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = wg / p.NT, nt = wg - mt * p.NT;
    const int bmv = TM == 2 ? BM : AROWS - 2 * dil - 1;    // output pixels this workgroup owns (the tile's last row stays empty, see ZROW)
    const int m0 = mt * bmv, n0 = nt * BN;
    const int a_ck = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;

    // DMA addressing with (almost) no vector arithmetic in the loop: the per-thread part of every offset is a constant
    // VGPR, what changes from unit to unit (pixel base, tap, channel chunk) goes into the instruction's SCALAR offset.
    // The scalar part must not be negative, so the activation descriptor starts `bias` bytes before the tensor (the
    // first tile's halo reaches that far back); nothing is fetched from there, rows outside the tensor are steered to
    // the out-of-range offset.  SC is a multiple of 64 here (host check): no channel-chunk mask.
    const int total = (p.M / (p.DH * p.DW)) * p.SH * p.SW;
    const int bias_px = dil + dil * p.SW;
    const __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.src) - (size_t)bias_px * p.SC, 0, (unsigned)(((size_t)total + bias_px) * p.SC * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.wgt), 0, (unsigned)((size_t)p.ntaps * p.DN * p.SC * 2u), 0x00020000);
    unsigned a_vo[A_N], b_vo[B_N];
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
        const int r = (tid >> 3) + 32 * i;
        a_vo[i] = r < bmv + 2 * dil ? (unsigned)((r * p.SC + a_ck) * 2) : OOBH;      // rows past the halo: zero-filled, the last one serves as ZROW
    }
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
        const int n = n0 + (tid >> 3) + 32 * i;
        b_vo[i] = n < p.DN ? (unsigned)((n * p.SC + a_ck) * 2) : OOBH;
    }
    const int nchunks = p.SC / HBK;
    const int nunits = nchunks * 3;

    auto issue = [&](int unit) {
        const int cc = unit / 3, kr = unit - cc * 3;
        unsigned char* As = smem + wave * 1024;
        unsigned char* Bs = smem + A_BYTES + wave * 1024;
        // tile row r = source pixel (m0 - dil + r), shifted by the kernel row: consecutive pixels of the NHWC tensor
        const int pix0 = m0 - dil + p.tap_dh[kr * 3] * p.SW;
        const int a_so = (int)(((unsigned)(pix0 + bias_px) * (unsigned)p.SC + (unsigned)(cc * HBK)) * 2u);      // < 4 GB (size guard): unsigned arithmetic
#pragma unroll
        for (int i = 0; i < A_N; ++i) {
            const int px = pix0 + (tid >> 3) + 32 * i;
            const unsigned vo = (unsigned)px < (unsigned)total ? a_vo[i] : OOBH;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, LDS_PTR(As + i * 4096), 16, (int)vo, a_so, 0, 0);
        }
#pragma unroll
        for (int kc = 0; kc < 3; ++kc) {
            const int b_so = (int)(((unsigned)((kr * 3 + kc) * p.DN) * (unsigned)p.SC + (unsigned)(cc * HBK)) * 2u);
#pragma unroll
            for (int i = 0; i < B_N; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wgt_rsrc, LDS_PTR(Bs + kc * (BN * 128) + i * 4096), 16, (int)b_vo[i], b_so, 0, 0);
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
    // per fragment row (output pixel) of this lane: which of the 9 taps fall inside the image
    unsigned fmsk[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + wm * 32 * TM + mi * 32 + li;
        const int mm = m < p.M ? m : 0;
        const int ow = mm % p.DW;
        const int t2 = mm / p.DW;
        const int oh = t2 % p.DH;
        unsigned mk = 0;
        for (int t = 0; t < 9; ++t) {
            const int sh = oh + p.tap_dh[t], sw = ow + p.tap_dw[t];
            if ((unsigned)sh < (unsigned)p.SH && (unsigned)sw < (unsigned)p.SW) mk |= 1u << t;
        }
        fmsk[mi] = (m < p.M && wm * 32 * TM + mi * 32 + li < bmv) ? mk : 0u;
    }
    // fragment addresses: tile row R = (row of the pixel) + tap_dw + dil; slot (2 st + lh) ^ ((R>>1)&7)
    int a_addr[3][TM];
#pragma unroll
    for (int kc = 0; kc < 3; ++kc)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int R = wm * 32 * TM + mi * 32 + li + p.tap_dw[kc] + dil;        // tap_dw of a kernel column is the same in every kernel row
            a_addr[kc][mi] = R * 128 + ((lh ^ ((R >> 1) & 7)) * 16);
        }
    const int q0 = (lh ^ ((li >> 1) & 7)) * 16;
    const int b_row = A_BYTES + (wn * 32 * TN + li) * 128 + q0;

    for (int unit = 0; unit < nunits; ++unit) {
        __syncthreads();                                  // every wave is done with the previous unit's tiles
        issue(unit);
        wait_tiles_and_sync<1>(0);
        const int kr = unit % 3;
        // Zero padding without touching the operands: a lane whose pixel has no neighbour under tap (kr, kc) reads the
        // tile's LAST row instead, which no pixel maps to and the DMA zero-fills with every unit (its offset is out of
        // range).  One address select per fragment and tap replaces 4 v_cndmask per fragment and k-step; the lanes that
        // share the empty row broadcast.  (The k-step XOR moves inside that 128-byte row.)
        int ua[3][TM];
#pragma unroll
        for (int kc = 0; kc < 3; ++kc)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) ua[kc][mi] = ((fmsk[mi] >> (kr * 3 + kc)) & 1u) ? a_addr[kc][mi] : ZROW;
        // 12 k-steps (3 taps x 4): fragments of the next PF-1 steps are in flight while a step multiplies
        constexpr int PF = GATHER_PF, KS = 3 * (HBK / 16);
        bf16x8 a[PF][TM], b[PF][TN];
        auto frags = [&](int ks) {
            const int kc = ks >> 2, st = ks & 3;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) a[ks % PF][mi] = *reinterpret_cast<const bf16x8*>(smem + (ua[kc][mi] ^ (st * 32)));
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                b[ks % PF][ni] = *reinterpret_cast<const bf16x8*>(smem + kc * (BN * 128) + ((b_row + ni * 4096) ^ (st * 32)));
        };
#pragma unroll
        for (int ks = 0; ks < PF - 1; ++ks) frags(ks);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + PF - 1 < KS) frags(ks + PF - 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks % PF][ni], a[ks % PF][mi], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    gather_epilogue<MODE, WM, WN, TM, TN, TM / 2>(p, smem, acc, tid, wm, wn, li, lh, m0, n0, bmv);
}

// =================================================================================
// 64 -> 64 channels, 3x3, stride 1 (conv1_2, forward and data gradient): a PERSISTENT kernel with the whole filter
// resident in LDS.  With K = 64 per tap and N = 64 the per-tap kernel stages 40 KB per 8 MFMAs per wave (it needs
// 80 B/clk of LDS-DMA, the CU delivers ~55) and a workgroup lives for 9 iterations, each a full DMA round trip.
// Here one 8-wave workgroup per CU loads the 9 x 64 x 64 filter once (72 KB), then walks its share of the pixel tiles:
// per kernel row ONE 256-row activation tile (253 pixels + halo + the always-empty row 255, see ZROW above), double
// buffered, feeds three taps: 32 KB per 24 MFMAs per wave = 21 B/clk, 4 pieces per wave and unit, no filter traffic.
// Wave w owns pixel rows 32 w .. 32 w + 31 of the tile x all 64 channels (2 accumulator tiles); the results leave
// straight from the accumulators (4 consecutive channels of one pixel per register quad = one 8-byte store).
// =================================================================================
// FIRSTW (round 5; data gradient of conv1_2 only): the layer below is the FIRST layer (conv1_1: 3 input channels, K = taps x 3
// = 27 <= 32), whose only use for this kernel's result is its weight gradient dW1[k][n] = sum_pixels im2col(image)[pixel][k] *
// dx[pixel][n].  The dx rows of a tile are in LDS anyway (the write-out staging): each wave multiplies its 32 rows against the
// 32 im2col rows of the same pixels (k = 27 carries a one so that row 27 of the product is the bias gradient).  The im2col
// tile is built from three linear RUNS of the image (pixels m0 - 1 + dh W ... + 254, dh = -1, 0, 1: contiguous memory, fetched
// as coalesced dwords by all threads and parked in LDS as bf16 -- the first version gathered its 16 values per thread
// straight from global memory, 4 bytes per lane and ~10 cache lines per instruction: + 2.2 us per tile) -- 4 MFMAs per wave and tile beside the 72 of the
// convolution -- into accumulators that live for the whole kernel; the workgroup's eight partial products are added in wave
// order at the end and leave as ONE slab.  dx itself is never written: 368 MB less to store and 368 MB less to read back at
// batch 32, and conv1_1's own weight-gradient kernel is not launched (0.274 + 0.126 ms -> one kernel).
struct FirstWArgs {
    const float* img;       // [B][DH][DW][3] fp32 (the first layer is 3x3 stride 1 SAME: its image has this kernel's output size)
    float* ws;              // [workgroups][K * 64 + 64] slabs
    int Ci, ntaps;
    int tap_dh[9], tap_dw[9];
};

template <int MODE, bool FIRSTW = false>
__global__ __launch_bounds__(512) void conv_gather_bf16_c64_kernel(GatherArgsH p, int ntiles, int xcd_chunks, FirstWArgs fw) {
    constexpr int AROWS = 256, BMV = 253, A_BYTES = AROWS * 128, B_TAP = 64 * 128, A_BASE = 9 * B_TAP;
    constexpr int ZROW = (AROWS - 1) * 128;
    constexpr int XT_BASE = A_BASE + 2 * A_BYTES, XROWB = 64;      // FIRSTW: im2col tile [256 pixels][32 k] bf16
    constexpr int SR_BASE = XT_BASE + 256 * XROWB, SR_RUN = (BMV + 2) * 3;      // ... and three runs of BMV + 2 image pixels x 3 channels (bf16), one per kernel row
    static_assert(!FIRSTW || MODE == MODE_DGRAD, "the fused first-layer weight gradient belongs to the data gradient");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int a_ck = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;

    const int total = (p.M / (p.DH * p.DW)) * p.SH * p.SW;
    const int bias_px = 1 + p.SW;
    const __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.src) - (size_t)bias_px * 64, 0, (unsigned)(((size_t)total + bias_px) * 64 * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.wgt), 0, (unsigned)(9u * 64u * 64u * 2u), 0x00020000);

    // ---- the filter, once: tap t = rows [64 n][64 k] of 128 bytes, one 1-KB piece per wave and tap
    {
        const unsigned vo = (unsigned)(((tid >> 3) * 64 + a_ck) * 2);
#pragma unroll
        for (int t = 0; t < 9; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wgt_rsrc, LDS_PTR(smem + t * B_TAP + wave * 1024), 16, (int)vo, t * (64 * 64 * 2), 0, 0);
    }
    unsigned a_vo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 3) + 64 * i;
        a_vo[i] = r < AROWS - 1 ? (unsigned)((r * 64 + a_ck) * 2) : OOBH;          // row 255 stays empty
    }
    // unit = (tile, kernel row); its activation tile: row r = pixel m0 - 1 + r of the image row shifted by the kernel row
    auto issue = [&](int tile, int kr, int buf) {
        unsigned char* As = smem + A_BASE + buf * A_BYTES + wave * 1024;
        const int pix0 = tile * BMV - 1 + p.tap_dh[kr * 3] * p.SW;
        const int a_so = (int)((unsigned)(pix0 + bias_px) * 128u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = pix0 + (tid >> 3) + 64 * i;
            const unsigned vo = (unsigned)px < (unsigned)total ? a_vo[i] : OOBH;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, LDS_PTR(As + i * 8192), 16, (int)vo, a_so, 0, 0);
        }
    };

    // fragment addresses: activation row R = 32 wave + li + tap_dw + 1; filter rows n = 32 ni + li
    int a_addr[3], b_addr[2];
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) {
        const int R = wave * 32 + li + p.tap_dw[kc] + 1;
        a_addr[kc] = R * 128 + ((lh ^ ((R >> 1) & 7)) * 16);
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = ni * 32 + li;
        b_addr[ni] = n * 128 + ((lh ^ ((n >> 1) & 7)) * 16);
    }
    float bv[2][4][4];                                   // forward: bias of this lane's channels 32 ni + 8 g + 4 lh + e
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[ni][g][e] = (MODE == MODE_FWD && p.bias) ? p.bias[ni * 32 + 8 * g + 4 * lh + e] : 0.f;

    // FIRSTW: im2col staging map -- thread -> pixel rows (tid >> 2) + 128 pass, k = 8 (tid & 3) .. + 7
    const int xr = tid >> 2, xq = tid & 3;
    int sidx[8];                 // this thread's 8 k: element (kh, (kw) * 3 + c) of the runs, relative to its pixel
    unsigned kbit[8];
    f32x16 accw[2];
    __amdgpu_buffer_rsrc_t img_rsrc = src_rsrc;
    if constexpr (FIRSTW) {
        const int K = fw.ntaps * fw.Ci;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = xq * 8 + j;
            const bool kv = k < K;
            const int tp = kv ? k / fw.Ci : 0, c = kv ? k - tp * fw.Ci : 0;
            sidx[j] = (tp / 3) * SR_RUN + (tp % 3) * 3 + c;      // (3x3 taps in kh-major order, dh = kh - 1, dw = kw - 1: host check)
            kbit[j] = kv ? (1u << tp) : 0u;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[0][r] = accw[1][r] = 0.f;
        img_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fw.img), 0, (unsigned)((size_t)p.M * fw.Ci * 4u), 0x00020000);
    }

    // Which tiles: workgroup w runs on XCD w % 8.  An input pixel row is fetched three times (once per kernel row, by the
    // output tiles W pixels before / at / after it = neighbouring tiles): with xcd_chunks the tile range is cut in 8
    // contiguous chunks, one per XCD, so the three fetches meet in ONE L2 instead of three (plain round robin otherwise).
    int tile = blockIdx.x, tstep = gridDim.x, tend = ntiles, gunit = 0;
    if (xcd_chunks) {
        const int chunk = (ntiles + 7) >> 3, x = blockIdx.x & 7;
        tile = x * chunk + (blockIdx.x >> 3);
        tstep = gridDim.x >> 3;
        tend = min(ntiles, (x + 1) * chunk);
    }
    if (tile < tend) issue(tile, 0, 0);
    for (; tile < tend; tile += tstep) {
        const int m0 = tile * BMV;
        const int rr = wave * 32 + li, m = m0 + rr;
        unsigned fmsk = 0;
        if (rr < BMV && m < p.M) {
            const int ow = m % p.DW, oh = (m / p.DW) % p.DH;
            for (int t = 0; t < 9; ++t) {
                const int sh = oh + p.tap_dh[t], sw = ow + p.tap_dw[t];
                if ((unsigned)sh < (unsigned)p.SH && (unsigned)sw < (unsigned)p.SW) fmsk |= 1u << t;
            }
        }
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;

        // data gradient: the relu mask / the values to accumulate onto are fetched while the last kernel row multiplies
        // (one workgroup per CU: nothing else would hide that round trip)
        u32x2 pre_old[2][4];
        u32x4 row_mask[4];
        float simg[5];
#pragma unroll
        for (int kr = 0; kr < 3; ++kr, ++gunit) {
            wait_tiles_and_sync<1>(0);                    // this unit's tile (and, the first time, the filter) has landed
            if (MODE == MODE_DGRAD && kr == 2 && rr < BMV && m < p.M) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const size_t o = (size_t)m * 64 + ni * 32 + 8 * g + 4 * lh;
                        if (p.accum) pre_old[ni][g] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p.dst) + o);
                    }
            }
            if (MODE == MODE_DGRAD && kr == 2 && p.mask) {      // the mask in the write-out layout: whole 128-byte rows
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r2 = wave * 32 + i * 8 + (lane >> 3), m2 = m0 + r2;
                    const size_t o2 = (size_t)((r2 < BMV && m2 < p.M) ? m2 : 0) * 64 + (lane & 7) * 8;
                    row_mask[i] = *reinterpret_cast<const u32x4*>(p.mask + o2);
                }
            }
            if (kr < 2) issue(tile, kr + 1, (gunit + 1) & 1);
            else if (tile + tstep < tend) issue(tile + tstep, 0, (gunit + 1) & 1);
            if constexpr (FIRSTW) {
                unsigned short* SR = reinterpret_cast<unsigned short*>(smem + SR_BASE);
                if (kr == 0) {      // the three image runs of this tile: 3 x 765 dwords, coalesced, in flight while the first kernel row multiplies
#pragma unroll
                    for (int u = 0; u < 5; ++u) {
                        const int g = tid + 512 * u;
                        const int sr = g / SR_RUN, e = g - sr * SR_RUN;
                        const long long first = (long long)(m0 - 1 + (sr - 1) * p.DW) * 3 + e;      // dword index in the image (negative / past the end: zero)
                        const unsigned off = (g < 3 * SR_RUN && first >= 0 && first < (long long)p.M * 3) ? (unsigned)first * 4u : OOBH;
                        simg[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(img_rsrc, off, 0, 0));
                    }
                } else if (kr == 1) {      // (landed: the unit's wait drains vmcnt) -> LDS as bf16; published by the next unit's barrier
#pragma unroll
                    for (int u = 0; u < 5; ++u) {
                        const int g = tid + 512 * u;
                        if (g < 3 * SR_RUN) SR[g] = f2bf(simg[u]);
                    }
                } else {      // im2col rows of this thread's two pixels x 8 k: 16 two-byte LDS reads, one 16-byte store each
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        const int r2 = xr + 128 * ps, m2 = m0 + r2;
                        const bool pv = r2 < BMV && m2 < p.M;
                        const int mm = pv ? m2 : 0;
                        const int ow = mm % p.DW, t2 = mm / p.DW;
                        const int oh = t2 % p.DH;
                        unsigned mk = 0;
                        for (int t = 0; t < 9; ++t) {
                            const int sh = oh + t / 3 - 1, sw = ow + t % 3 - 1;
                            if ((unsigned)sh < (unsigned)p.DH && (unsigned)sw < (unsigned)p.DW) mk |= 1u << t;
                        }
                        if (!pv) mk = 0;
                        unsigned hv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const unsigned v = SR[sidx[j] + r2 * 3];
                            hv[j] = (mk & kbit[j]) ? v : 0u;
                        }
                        if (xq == 3) hv[3] = pv ? 0x3F80u : 0u;      // k = 27 = K (host check): the ones column -> row 27 of the product is the bias gradient
                        *reinterpret_cast<u32x4*>(smem + XT_BASE + r2 * XROWB + xq * 16) =
                            u32x4{hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16), hv[4] | (hv[5] << 16), hv[6] | (hv[7] << 16)};
                    }
                }
            }
            const unsigned char* A = smem + A_BASE + (gunit & 1) * A_BYTES;
            int ua[3];
#pragma unroll
            for (int kc = 0; kc < 3; ++kc) ua[kc] = ((fmsk >> (kr * 3 + kc)) & 1u) ? a_addr[kc] : ZROW;
            bf16x8 a[2], b[2][2];
            auto frags = [&](int ks) {                   // k-step ks: tap column ks >> 2, 16-channel slice ks & 3
                const int kc = ks >> 2, st = ks & 3;
                a[ks & 1] = *reinterpret_cast<const bf16x8*>(A + (ua[kc] ^ (st * 32)));
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    b[ks & 1][ni] = *reinterpret_cast<const bf16x8*>(smem + (kr * 3 + kc) * B_TAP + (b_addr[ni] ^ (st * 32)));
            };
                frags(0);
#pragma unroll
            for (int ks = 0; ks < 12; ++ks) {
                if (ks + 1 < 12) frags(ks + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks & 1][ni], a[ks & 1], acc[ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        // ---- results: pixel m, channels 32 ni + 8 g + 4 lh + (0..3)
        {
            // Through LDS, so that rows leave (and the mask arrives) as whole 128-byte lines: written straight from the
            // accumulators every 128-byte line is touched by 8 different 16-byte requests (and as many for the mask).  The
            // staging area is this wave's 32 rows of the activation buffer the last unit has just consumed; 16-byte chunk c
            // of row r lives at chunk c ^ sw(r).
            __syncthreads();                              // the neighbours have read their halo rows of that buffer (FIRSTW: and the im2col tile is complete)
            unsigned char* S = smem + A_BASE + ((gunit - 1) & 1) * A_BYTES + wave * (32 * 128);
            const int swl = (li ^ (li >> 3)) & 7;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4] = {acc[ni][4 * g], acc[ni][4 * g + 1], acc[ni][4 * g + 2], acc[ni][4 * g + 3]};
                    if constexpr (MODE == MODE_FWD) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] += bv[ni][g][e];
                            if (p.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
                        }
                    } else if (p.accum && rr < BMV && m < p.M) {
                        const u32x2 old = pre_old[ni][g];
                        v[0] += lo2f(old[0]); v[1] += hi2f(old[0]); v[2] += lo2f(old[1]); v[3] += hi2f(old[1]);
                    }
                    *reinterpret_cast<u32x2*>(S + li * 128 + (((4 * ni + g) ^ swl) * 16) + lh * 8) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
                }
            // (wave-private rows: no barrier between this wave's writes and its reads)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + (lane >> 3), ch = lane & 7;
                const int r2 = wave * 32 + row, m2 = m0 + r2;
                u32x4 v = *reinterpret_cast<const u32x4*>(S + row * 128 + ((ch ^ ((row ^ (row >> 3)) & 7)) * 16));
                if (MODE == MODE_DGRAD && p.mask) {
                    const u32x4 y = row_mask[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned lo = lo2f(y[e]) > 0.f ? 0x0000FFFFu : 0u, hi = hi2f(y[e]) > 0.f ? 0xFFFF0000u : 0u;
                        v[e] &= (lo | hi);
                    }
                }
                if constexpr (FIRSTW) {      // the masked row back into the staging area (zeros for rows that are not pixels)
                    if (!(r2 < BMV && m2 < p.M)) v = u32x4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4*>(S + row * 128 + ((ch ^ ((row ^ (row >> 3)) & 7)) * 16)) = v;
                } else {
                    if (r2 < BMV && m2 < p.M) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.dst) + (size_t)m2 * 64 + ch * 8) = v;
                }
            }
            if constexpr (FIRSTW) {
                // dW1 += im2col^T (k x pixels) * dx (pixels x channels) over this wave's 32 rows: both operands k-major through the
                // transpose read (conv_first_bf16.hip); the im2col tile was published by the barrier above, the dx rows are this
                // wave's own (its LDS writes are ordered before its reads)
                asm volatile("" ::: "memory");
                const int q = lane & 15, cb = (lane >> 4) & 1;
                auto tr8 = [&](const unsigned char* a0, const unsigned char* a1) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)LDS_PTR(a0));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)LDS_PTR(a1));
                    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                };
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const int pr0 = st * 16 + lh * 8 + (q >> 2), pr1 = pr0 + 4;      // rows inside the wave's block
                    const unsigned char* X = smem + XT_BASE + wave * 32 * XROWB + (cb * 2 + ((q >> 1) & 1)) * 16 + (q & 1) * 8;
                    const s16x8 a = tr8(X + pr0 * XROWB, X + pr1 * XROWB);
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) {
                        const int ych = nh * 4 + cb * 2 + ((q >> 1) & 1);
                        const s16x8 bq = tr8(S + pr0 * 128 + ((ych ^ ((pr0 ^ (pr0 >> 3)) & 7)) * 16) + (q & 1) * 8,
                                             S + pr1 * 128 + ((ych ^ ((pr1 ^ (pr1 >> 3)) & 7)) * 16) + (q & 1) * 8);
                        accw[nh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq), accw[nh], 0, 0, 0);
                    }
                }
            }
        }
    }
    if constexpr (FIRSTW) {
        // the workgroup's slab: its eight waves' products added in wave order (fixed order: run-to-run identical), rows 0..26 =
        // dW1[k][n], row 27 = the bias gradient
        __syncthreads();
        float* Rw = reinterpret_cast<float*>(smem + A_BASE);      // [8 waves][32 k][64 n] floats = the two activation buffers
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int r = 0; r < 16; ++r) Rw[wave * 2048 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + nh * 32 + li] = accw[nh][r];
        __syncthreads();
        const int K = fw.ntaps * fw.Ci;
        float* slab = fw.ws + (size_t)blockIdx.x * (K * 64 + 64);
        for (int idx = tid; idx < (K + 1) * 64; idx += 512) {
            float sum = Rw[idx];
#pragma unroll
            for (int w = 1; w < 8; ++w) sum += Rw[w * 2048 + idx];
            slab[idx] = sum;      // (k = K lands at K * 64 + n: the slab's bias row)
        }
    }
}

// =================================================================================
// Round 5: the forward kernels with the 2x2 stride-2 max-pool FUSED into their epilogue (conv.h conv_fwd_pool_bf16).
// pool1-3 are pure HBM traffic between two convolutions: the producer writes its output, the pool reads it back to write a
// quarter of it -- and nothing else ever reads the unpooled tensor, not even backward (the pool's record carries the argmax
// and the relu sign).  Pooling in the producer's epilogue needs whole 2x2 windows in one tile, i.e. tiles of IMAGE ROWS
// instead of the raster runs of the kernels above:
//   the 256-row activation tile of a kernel row is [R image rows][CW slots] (R x CW = 256), slot cc of row j = input pixel
//   (oh0 + j + kh - 1, c0 - 1 + cc): TW = up to CW - 2 output columns plus a one-column halo either side, every slot's zero
//   padding decided by the DMA (a slot outside the image is an out-of-range offset), so there is no tap mask and no empty row.
//   GEMM row m = j * CW + c is output pixel (oh0 + j, c0 + c) and reads tile row m + kw under tap kw -- exactly the
//   fragment addressing of the raster kernels, the compute loops are theirs unchanged.  Rows with c >= TW are dead
//   (R x TW of 256 rows do work: 93.75 % for conv1_2 at 300 / conv2_2 at 150; the executor fuses where >= 88 % do).
// Epilogue: bias + relu + ONE rounding to bf16, then the window's first maximum in scan order over the cells inside the
// image (ops.hip maxpool2x2_fwd_rec_kernel: bit-identical results and records), pooled rows leave as 16-byte pieces of
// whole 128-byte lines.
// =================================================================================
__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }      // the value the unfused path stores and re-reads
__device__ __forceinline__ void lds_barrier() {      // LDS hand-off between waves WITHOUT draining the DMA in flight (vmcnt untouched)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// one pooled pixel x 8 channels: cells v[q][e] (already rounded to bf16 values), validity ok[q] (cell 0 always inside)
__device__ __forceinline__ void pool_window8(const float (&v)[4][8], const bool (&ok)[4], u32x4& out, unsigned& rec) {
    float m[8];
    unsigned r[2] = {0u, 0u};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        unsigned a = 0;
        float mm = v[0][e];
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (ok[q] && v[q][e] > mm) { mm = v[q][e]; a = q; }      // strict: the first maximum wins
        m[e] = mm;
        r[e / 4] |= (a | (mm > 0.f ? 4u : 0u)) << (3 * (e & 3));
    }
    out = u32x4{pack2(m[0], m[1]), pack2(m[2], m[3]), pack2(m[4], m[5]), pack2(m[6], m[7])};
    rec = r[0] | (r[1] << 16);
}

// ---- kernel-row gather, 256-row tiles as [8 image rows][32 slots] (TW <= 30), 128 output channels per workgroup ----------
__global__ __launch_bounds__(256) void conv_fwd_pool_bf16_rows_kernel(GatherArgsH p) {
    constexpr int TM = 4, TN = 2, WN = 2, BN = 128, AROWS = 256, A_N = 8, B_N = 4, CW = 32, R = 8;
    constexpr int A_BYTES = AROWS * 128, LDC = BN + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = wg / p.NT, nt = wg - mt * p.NT;
    const int sg = mt % p.NSEG, t2 = mt / p.NSEG;
    const int rb = t2 % p.NBAND, b = t2 / p.NBAND;
    const int oh0 = rb * R, c0 = sg * p.TW, n0 = nt * BN;
    const int a_ck = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;

    const int total = p.PB * p.SH * p.SW;
    const int bias_px = 1 + p.SW;                        // the first tile's halo reaches one row and one column before the tensor
    const __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.src) - (size_t)bias_px * p.SC, 0, (unsigned)(((size_t)total + bias_px) * p.SC * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.wgt), 0, (unsigned)((size_t)9 * p.DN * p.SC * 2u), 0x00020000);
    // staging: DMA piece i of a thread fills tile row 32 i + (tid >> 3) = (image row i, slot tid >> 3): the slot -- and with it
    // the column's validity -- is a per-thread constant, the row's a per-piece scalar
    const int cc = tid >> 3;
    const bool colok = cc < p.TW + 2 && (unsigned)(c0 - 1 + cc) < (unsigned)p.SW;
    unsigned a_vo[A_N], b_vo[B_N];
#pragma unroll
    for (int i = 0; i < A_N; ++i) a_vo[i] = colok ? (unsigned)(((i * p.SW + cc) * p.SC + a_ck) * 2) : OOBH;
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
        const int n = n0 + (tid >> 3) + 32 * i;
        b_vo[i] = n < p.DN ? (unsigned)((n * p.SC + a_ck) * 2) : OOBH;
    }
    const int nunits = (p.SC / HBK) * 3;

    auto issue = [&](int unit) {
        const int cch = unit / 3, kr = unit - cch * 3;
        unsigned char* As = smem + wave * 1024;
        unsigned char* Bs = smem + A_BYTES + wave * 1024;
        const int row0 = oh0 + kr - 1;                   // image row of tile row block 0 under this kernel row
        const int a_so = (int)(((unsigned)((b * p.SH + row0) * p.SW + c0 - 1 + bias_px) * (unsigned)p.SC + (unsigned)(cch * HBK)) * 2u);
#pragma unroll
        for (int i = 0; i < A_N; ++i) {
            const unsigned vo = (unsigned)(row0 + i) < (unsigned)p.SH ? a_vo[i] : OOBH;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, LDS_PTR(As + i * 4096), 16, (int)vo, a_so, 0, 0);
        }
#pragma unroll
        for (int kc = 0; kc < 3; ++kc) {
            const int b_so = (int)(((unsigned)((kr * 3 + kc) * p.DN) * (unsigned)p.SC + (unsigned)(cch * HBK)) * 2u);
#pragma unroll
            for (int i = 0; i < B_N; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wgt_rsrc, LDS_PTR(Bs + kc * (BN * 128) + i * 4096), 16, (int)b_vo[i], b_so, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int bb = 0; bb < TN; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][bb][r] = 0.f;

    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, lh = lane >> 5;
    int a_addr[3][TM];
#pragma unroll
    for (int kc = 0; kc < 3; ++kc)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int Rr = wm * 32 * TM + mi * 32 + li + kc;      // GEMM row m reads tile row m + kw (dead rows run into the next block / the filter tile: discarded)
            a_addr[kc][mi] = Rr * 128 + ((lh ^ ((Rr >> 1) & 7)) * 16);
        }
    const int q0 = (lh ^ ((li >> 1) & 7)) * 16;
    const int b_row = A_BYTES + (wn * 32 * TN + li) * 128 + q0;

    for (int unit = 0; unit < nunits; ++unit) {
        __syncthreads();                                  // every wave is done with the previous unit's tiles
        issue(unit);
        wait_tiles_and_sync<1>(0);
        constexpr int PF = GATHER_PF, KS = 3 * (HBK / 16);
        bf16x8 a[PF][TM], bfr[PF][TN];
        auto frags = [&](int ks) {
            const int kc = ks >> 2, st = ks & 3;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) a[ks % PF][mi] = *reinterpret_cast<const bf16x8*>(smem + (a_addr[kc][mi] ^ (st * 32)));
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                bfr[ks % PF][ni] = *reinterpret_cast<const bf16x8*>(smem + kc * (BN * 128) + ((b_row + ni * 4096) ^ (st * 32)));
        };
#pragma unroll
        for (int ks = 0; ks < PF - 1; ++ks) frags(ks);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + PF - 1 < KS) frags(ks + PF - 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks % PF][ni], a[ks % PF][mi], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();

    // ---- epilogue through an fp32 C tile [128][132], half of every wave's rows at a time.  Half h of wave row block wm holds
    // tile rows wm 128 + h 64 + x = image rows j = 4 wm + 2 h + (x >> 5), columns x & 31: one complete ROW PAIR per (wm, h).
    float* Cs = reinterpret_cast<float*>(smem);
    const int cg = tid & 15, n = n0 + cg * 8;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (p.bias && n < p.DN) ? p.bias[n + e] : 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ml = wm * 64 + mi * 32 + li;
                    const int nl = wn * 32 * TN + ni * 32 + 8 * g + 4 * lh;
                    const f32x16& c = acc[2 * h + mi][ni];
                    *reinterpret_cast<f32x4*>(Cs + ml * LDC + nl) = f32x4{c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
                }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = tid + 256 * it;              // (wave row block, window, channel group)
            const int wq = (item >> 4) & 15, wmr = item >> 8;
            const int oh = oh0 + 4 * wmr + 2 * h, ow = c0 + 2 * wq;
            if (!(2 * wq < p.TW && ow < p.DW && oh < p.DH && n < p.DN)) continue;
            const bool ok[4] = {true, ow + 1 < p.DW, oh + 1 < p.DH, ow + 1 < p.DW && oh + 1 < p.DH};
            float v[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* cr = Cs + (wmr * 64 + (q >> 1) * 32 + 2 * wq + (q & 1)) * LDC + cg * 8;
                const f32x4 c0v = *reinterpret_cast<const f32x4*>(cr), c1v = *reinterpret_cast<const f32x4*>(cr + 4);
                const float t[8] = {c0v[0], c0v[1], c0v[2], c0v[3], c1v[0], c1v[1], c1v[2], c1v[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = t[e] + bv[e];
                    v[q][e] = rbf(x > 0.f ? x : 0.f);
                }
            }
            u32x4 out;
            unsigned rec;
            pool_window8(v, ok, out, rec);
            const size_t pix = ((size_t)b * p.PH + (oh >> 1)) * p.PW + (ow >> 1);
            *reinterpret_cast<u32x4*>(p.pool_dst + pix * p.DN + n) = out;
            if (p.pool_rec) *reinterpret_cast<unsigned*>(p.pool_rec + pix * (p.DN >> 2) + (n >> 2)) = rec;
        }
    }
}

// ---- 64 -> 64 channels, persistent, filter resident (conv_gather_bf16_c64_kernel): tiles as [4 image rows][64 slots] (TW <= 62) --
__global__ __launch_bounds__(512) void conv_fwd_pool_bf16_c64_kernel(GatherArgsH p, int ntiles, int xcd_chunks) {
    constexpr int AROWS = 256, A_BYTES = AROWS * 128, B_TAP = 64 * 128, A_BASE = 9 * B_TAP, R = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int a_ck = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;

    const int total = p.PB * p.SH * p.SW;
    const int bias_px = 1 + p.SW;
    const __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.src) - (size_t)bias_px * 64, 0, (unsigned)(((size_t)total + bias_px) * 64 * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.wgt), 0, (unsigned)(9u * 64u * 64u * 2u), 0x00020000);
    {
        const unsigned vo = (unsigned)(((tid >> 3) * 64 + a_ck) * 2);
#pragma unroll
        for (int t = 0; t < 9; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wgt_rsrc, LDS_PTR(smem + t * B_TAP + wave * 1024), 16, (int)vo, t * (64 * 64 * 2), 0, 0);
    }
    // staging: DMA piece i of a thread fills tile row 64 i + (tid >> 3) = (image row i, slot tid >> 3)
    const int cc = tid >> 3;
    const bool cc_in = cc < p.TW + 2;
    unsigned a_vo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_vo[i] = (unsigned)(((i * p.SW + cc) * 64 + a_ck) * 2);
    struct Tile { int b, oh0, c0; };
    auto decode = [&](int tile) {
        Tile t;
        const int sg = tile % p.NSEG, t2 = tile / p.NSEG;
        t.c0 = sg * p.TW; t.oh0 = (t2 % p.NBAND) * R; t.b = t2 / p.NBAND;
        return t;
    };
    auto issue = [&](const Tile& t, int kr, int buf) {
        unsigned char* As = smem + A_BASE + buf * A_BYTES + wave * 1024;
        const int row0 = t.oh0 + kr - 1;
        const bool colok = cc_in && (unsigned)(t.c0 - 1 + cc) < (unsigned)p.SW;
        const int a_so = (int)((unsigned)((t.b * p.SH + row0) * p.SW + t.c0 - 1 + bias_px) * 128u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned vo = (colok && (unsigned)(row0 + i) < (unsigned)p.SH) ? a_vo[i] : OOBH;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, LDS_PTR(As + i * 8192), 16, (int)vo, a_so, 0, 0);
        }
    };

    int a_addr[3], b_addr[2];
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) {
        const int Rr = wave * 32 + li + kc;               // GEMM row m reads tile row m + kw
        a_addr[kc] = Rr * 128 + ((lh ^ ((Rr >> 1) & 7)) * 16);
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int nn = ni * 32 + li;
        b_addr[ni] = nn * 128 + ((lh ^ ((nn >> 1) & 7)) * 16);
    }
    float bv[2][4][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[ni][g][e] = p.bias ? p.bias[ni * 32 + 8 * g + 4 * lh + e] : 0.f;

    int tile = blockIdx.x, tstep = gridDim.x, tend = ntiles, gunit = 0;
    if (xcd_chunks) {      // the tile range in 8 contiguous chunks, one per XCD: neighbouring bands (shared halo rows) meet in one L2
        const int chunk = (ntiles + 7) >> 3, x = blockIdx.x & 7;
        tile = x * chunk + (blockIdx.x >> 3);
        tstep = gridDim.x >> 3;
        tend = min(ntiles, (x + 1) * chunk);
    }
    Tile cur = decode(tile < tend ? tile : 0), nxt = cur;
    if (tile < tend) issue(cur, 0, 0);
    for (; tile < tend; tile += tstep, cur = nxt) {
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
#pragma unroll
        for (int kr = 0; kr < 3; ++kr, ++gunit) {
            wait_tiles_and_sync<1>(0);                    // this unit's tile (and, the first time, the filter) has landed
            if (kr < 2) issue(cur, kr + 1, (gunit + 1) & 1);
            else if (tile + tstep < tend) {
                nxt = decode(tile + tstep);
                issue(nxt, 0, (gunit + 1) & 1);
            }
            const unsigned char* A = smem + A_BASE + (gunit & 1) * A_BYTES;
            bf16x8 a[2], bfr[2][2];
            auto frags = [&](int ks) {                   // k-step ks: tap column ks >> 2, 16-channel slice ks & 3
                const int kc = ks >> 2, st = ks & 3;
                a[ks & 1] = *reinterpret_cast<const bf16x8*>(A + (a_addr[kc] ^ (st * 32)));
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    bfr[ks & 1][ni] = *reinterpret_cast<const bf16x8*>(smem + (kr * 3 + kc) * B_TAP + (b_addr[ni] ^ (st * 32)));
            };
                frags(0);
#pragma unroll
            for (int ks = 0; ks < 12; ++ks) {
                if (ks + 1 < 12) frags(ks + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks & 1][ni], a[ks & 1], acc[ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        // ---- results: the whole tile (bias + relu, rounded to bf16) parked in the activation buffer the last unit has just
        // consumed -- row = tile row, 16-byte chunk c at chunk c ^ sw(row & 31) --, then one pooled pixel x 8 channels per thread
        lds_barrier();                                    // every wave has read its fragments of that buffer
        unsigned char* S = smem + A_BASE + ((gunit - 1) & 1) * A_BYTES;
        {
            const int swl = (li ^ (li >> 3)) & 7;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4] = {acc[ni][4 * g], acc[ni][4 * g + 1], acc[ni][4 * g + 2], acc[ni][4 * g + 3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] += bv[ni][g][e];
                        v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    *reinterpret_cast<u32x2*>(S + (wave * 32 + li) * 128 + (((4 * ni + g) ^ swl) * 16) + lh * 8) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
                }
        }
        lds_barrier();
        {
            const int ch = tid & 7, wq = (tid >> 3) & 31, pr = tid >> 8;      // channel chunk, window, pooled row of the tile
            const int oh = cur.oh0 + 2 * pr, ow = cur.c0 + 2 * wq;
            if (2 * wq < p.TW && ow < p.DW && oh < p.DH) {
                const bool ok[4] = {true, ow + 1 < p.DW, oh + 1 < p.DH, ow + 1 < p.DW && oh + 1 < p.DH};
                float v[4][8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = (2 * pr + (q >> 1)) * 64 + 2 * wq + (q & 1), rl = row & 31;
                    const u32x4 w = *reinterpret_cast<const u32x4*>(S + row * 128 + ((ch ^ ((rl ^ (rl >> 3)) & 7)) * 16));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[q][2 * e] = lo2f(w[e]); v[q][2 * e + 1] = hi2f(w[e]); }
                }
                u32x4 out;
                unsigned rec;
                pool_window8(v, ok, out, rec);
                const size_t pix = ((size_t)cur.b * p.PH + (oh >> 1)) * p.PW + (ow >> 1);
                *reinterpret_cast<u32x4*>(p.pool_dst + pix * 64 + ch * 8) = out;
                if (p.pool_rec) *reinterpret_cast<unsigned*>(p.pool_rec + pix * 16 + ch * 2) = rec;
            }
        }
    }
}

// =================================================================================
// weight gradient:  dW[(tap, c)][n] = sum_m x[pix(m, tap)][c] * dy[m][n]      (fp32 slabs, split-M)
// One workgroup owns (tap, channel tile, n tile, pixel split); 64 pixels per iteration.  Both
// tiles keep their global pixel-major rows in LDS; ds_read_b64_tr_b16 transposes on the way out.
// =================================================================================
struct WgradArgsH {
    const bf16_t* x;
    const bf16_t* dy;
    float* ws;              // [nsplit][ntaps*Ci*Co + Co]
    int M, Hi, Wi, Ci, Ho, Wo, Co;
    int ntaps, stride;
    int CT, NT;
    int mchunk, nsplit;
    int tap_dh[9], tap_dw[9];
    // direct epilogue (nsplit == 1, the grouped launch of the tail's layers): dw = sum + wd * w, db = column sums -- no slab, no reduce
    float* dw;
    float* db;
    const float* w;
    float wd;
};

// The body of the per-tap weight gradient for workgroup `wg_index` of `wg_count` of ONE layer: called by the single-layer kernel
// and by the grouped one (conv_wgrad_group_bf16_kernel), whose grid enumerates the workgroups of several layers.
template <int WM, int WN, int TM, int TN, int NS, bool XCD = true>
__device__ __forceinline__ void conv_wgrad_bf16_body(const WgradArgsH& p, const int wg_index, const int wg_count, unsigned char* smem) {
    constexpr int BKT = 32 * TM * WM, BNT = 32 * TN * WN, BP = 64;
    constexpr int XCPR = BKT / 8, YCPR = BNT / 8;               // 16-byte chunks per pixel row
    constexpr int XRPP = 256 / XCPR, YRPP = 256 / YCPR;         // pixel rows per DMA pass of the workgroup
    constexpr int X_N = BP / XRPP, Y_N = BP / YRPP;             // DMA instructions per thread and tile
    constexpr int XROWB = BKT * 2, YROWB = BNT * 2;
    constexpr int X_LDS = BP * XROWB, STAGE = BP * (XROWB + YROWB);
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(XCPR == 8 || XCPR == 16 || XCPR == 32, "channel tile 64, 128 or 256");
    static_assert(YCPR == 8 || YCPR == 16, "n tile 64 or 128");

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;      // scalar: LDS-DMA bases stay in SGPRs
    const int wgid = XCD ? xcd_remap(wg_index, wg_count) : wg_index;
    const int ntiles = p.ntaps * p.CT * p.NT;
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

    // chunk swizzle of a pixel row r: rows of 256 or 512 bytes p ^ 4*(r&3); 128-byte rows p ^ 4*((r>>1)&1)
    auto swz = [](int r, int cpr) { return cpr >= 16 ? (r & 3) * 4 : ((r >> 1) & 1) * 4; };

    // ---- staging map -----------------------------------------------------------------------
    const int xr = tid / XCPR, xs = tid % XCPR;          // pixel row within a pass, LDS slot
    const int yr = tid / YCPR, ys = tid % YCPR;
    const int xchunk = xs ^ swz(xr, XCPR);               // XRPP, YRPP are multiples of 4: the swizzle is per-thread constant
    const int ychunk = ys ^ swz(yr, YCPR);
    const unsigned xcm = 0u - (unsigned)(c0 + xchunk * 8 < p.Ci);
    const unsigned ycm = 0u - (unsigned)(n0 + ychunk * 8 < p.Co);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.x), 0, (unsigned)((size_t)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.Ci * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dy), 0, (unsigned)((size_t)p.M * p.Co * 2u), 0x00020000);

    // This thread's X_N pixel rows advance by BP pixels per iteration (issue() runs for it = 0, 1, 2, ...): a
    // mixed-radix add of BP = (adv_b, adv_h, adv_w) to (b, oh, ow) with at most one carry per digit.  The byte
    // offset of the row's pixel moves with it by uniform constants (no multiplies in the loop), and the tap's
    // zero padding is an interval test on (oh, ow).
    const int adv_w = BP % p.Wo, adv_t = BP / p.Wo;
    const int adv_h = adv_t % p.Ho, adv_b = adv_t / p.Ho;
    const int C2 = p.Ci * 2;
    const unsigned xadv = (unsigned)(((adv_b * p.Hi + adv_h * p.stride) * p.Wi + adv_w * p.stride) * C2);
    const unsigned xadv_cw = (unsigned)((p.stride * p.Wi - p.Wo * p.stride) * C2);     // ow wrapped: next image row
    const unsigned xadv_ch = (unsigned)((p.Hi - p.Ho * p.stride) * p.Wi * C2);         // oh wrapped: next image
    const unsigned xtap = (unsigned)((dh * p.Wi + dw) * C2);
    // oh*stride + dh in [0, Hi)  <=>  oh in [h_lo, h_hi]   (likewise ow)
    const int h_lo = dh < 0 ? (-dh + p.stride - 1) / p.stride : 0, w_lo = dw < 0 ? (-dw + p.stride - 1) / p.stride : 0;
    const int h_hi_ = (p.Hi - 1 - dh) >= 0 ? (p.Hi - 1 - dh) / p.stride : -1, w_hi_ = (p.Wi - 1 - dw) >= 0 ? (p.Wi - 1 - dw) / p.stride : -1;
    const unsigned h_span = (unsigned)((h_hi_ < p.Ho - 1 ? h_hi_ : p.Ho - 1) - h_lo);   // wraps to huge when empty: the compare below then fails...
    const unsigned w_span = (unsigned)((w_hi_ < p.Wo - 1 ? w_hi_ : p.Wo - 1) - w_lo);
    const bool tap_empty = (int)h_span < 0 || (int)w_span < 0;                           // ...so an empty interval is flagged explicitly
    int pw[X_N], ph[X_N];
    unsigned xoff[X_N], yoff[Y_N];
#pragma unroll
    for (int j = 0; j < X_N; ++j) {
        const int m = mbeg + xr + j * XRPP;
        pw[j] = m % p.Wo;
        const int t2 = m / p.Wo;
        ph[j] = t2 % p.Ho;
        const int pb = t2 / p.Ho;
        xoff[j] = (unsigned)(((pb * p.Hi + ph[j] * p.stride) * p.Wi + pw[j] * p.stride) * C2 + (c0 + xchunk * 8) * 2) + xtap;
    }
#pragma unroll
    for (int j = 0; j < Y_N; ++j) yoff[j] = (unsigned)(((mbeg + yr + j * YRPP) * p.Co + n0 + ychunk * 8) * 2);
    const unsigned xcm_t = tap_empty ? 0u : xcm;

    // (the explicit (int) on the DMA offsets matters: hipcc 7.2 silently drops the HOST stub of this kernel when an
    // unsigned expression over a captured array element converts implicitly there -- the library then fails to load)
    auto issue = [&](int it, int stage) {
        unsigned char* Xs = smem + stage * STAGE + wave * 1024;
        unsigned char* Ys = smem + stage * STAGE + X_LDS + wave * 1024;
        const int left = mend - (mbeg + it * BP);          // rows of this tile inside the split
#pragma unroll
        for (int j = 0; j < X_N; ++j) {
            const bool ok = (xr + j * XRPP < left) && (unsigned)(ph[j] - h_lo) <= h_span && (unsigned)(pw[j] - w_lo) <= w_span;
            const unsigned mk = xcm_t & (0u - (unsigned)ok);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, LDS_PTR(Xs + j * 4096), 16, (int)((xoff[j] & mk) | (OOBH & ~mk)), 0, 0, 0);
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
            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, LDS_PTR(Ys + j * 4096), 16, (int)((yoff[j] & mk) | (OOBH & ~mk)), 0, 0, 0);
            yoff[j] += (unsigned)(BP * p.Co * 2);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // bias gradient (workgroups of tap 0, channel tile 0) on the matrix cores: ones^T dy in the waves with wm == 0 -- round 3;
    // the scalar LDS sweep of the dy tile by one wave made those workgroups the grid's stragglers
    f32x16 accb[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[b][r] = 0.f;
    s16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (short)0x3F80;       // bf16 1.0

    const int wm = wave / WN, wn = wave - wm * WN;
    const bool bias_wave = do_bias && wm == 0;
    const int li = lane & 31, lh = lane >> 5;
    // transpose-read addressing (bf16_probe.hip): inside a 16-lane group lane q supplies the 8 bytes at
    // (pixel row q>>2, channel quad q&3); lane l receives, for j = 0..3, pixel row j of channel l&15.
    // Two reads (pixel rows +0..3, +4..7) give the lane k = 8*lh .. 8*lh+7 of channel tile row lane&31.
    const int q = lane & 15, cb = (lane >> 4) & 1;
    const int prow = lh * 8 + (q >> 2);                   // pixel row of this lane's address (first read)
    int xa[TM], ya[TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int ch = (wm * 32 * TM + mi * 32) / 8 + cb * 2 + ((q >> 1) & 1);
        xa[mi] = prow * XROWB + ((ch ^ swz(prow, XCPR)) * 16) + (q & 1) * 8;
    }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int ch = (wn * 32 * TN + ni * 32) / 8 + cb * 2 + ((q >> 1) & 1);
        ya[ni] = X_LDS + prow * YROWB + ((ch ^ swz(prow, YCPR)) * 16) + (q & 1) * 8;
    }

    auto compute = [&](int stage) {
        const unsigned char* S = smem + stage * STAGE;
        // fragments of k-step st+1 are read while the MFMAs of step st run (two register sets).  The two
        // transpose reads of a fragment are 4 pixel rows apart: neither swizzle term (r&3, (r>>1)&1) changes.
        s16x8 a[2][TM], b[2][TN];
        auto tr8 = [&](int off, int rowb) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)LDS_PTR(S + off));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)LDS_PTR(S + off + 4 * rowb));
            return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        auto frags = [&](int st) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) a[st & 1][mi] = tr8(xa[mi] + (st * 16) * XROWB, XROWB);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) b[st & 1][ni] = tr8(ya[ni] + (st * 16) * YROWB, YROWB);
        };
        frags(0);
#pragma unroll
        for (int st = 0; st < BP / 16; ++st) {
            if (st + 1 < BP / 16) frags(st + 1);
            __builtin_amdgcn_sched_barrier(0);      // as in the gather kernel: reads of step st+1 stay ahead of the MFMAs of step st
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[st & 1][mi]),
                                                                         __builtin_bit_cast(bf16x8, b[st & 1][ni]), acc[mi][ni], 0, 0, 0);
            if (bias_wave) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    accb[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, b[st & 1][ni]),
                                                                      accb[ni], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < niter) issue(t, t);
    int st_c = 0, st_i = NS - 1;
    for (int it = 0; it < niter; ++it) {
        const int later = niter - 1 - it;
        wait_tiles_and_sync<X_N + Y_N>(later < NS - 2 ? later : NS - 2);
        if (it + NS - 1 < niter) issue(it + NS - 1, st_i);
        compute(st_c);
        st_c = st_c + 1 == NS ? 0 : st_c + 1;
        st_i = st_i + 1 == NS ? 0 : st_i + 1;
    }

    const size_t wcount = (size_t)p.ntaps * p.Ci * p.Co;
    if (p.dw) {      // direct: this workgroup holds the layer's whole pixel range (nsplit == 1)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + wn * 32 * TN + ni * 32 + li;
                if (n >= p.Co) continue;
                float wv[16];
                size_t o[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = wm * 32 * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    o[r] = ((size_t)tap * p.Ci + (c0 + kl < p.Ci ? c0 + kl : 0)) * p.Co + n;
                    wv[r] = p.wd != 0.f ? p.w[o[r]] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = wm * 32 * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (c0 + kl >= p.Ci) continue;
                    p.dw[o[r]] = p.wd != 0.f ? acc[mi][ni][r] + p.wd * wv[r] : acc[mi][ni][r];      // (the reduce kernels' order: sum, then + wd * w)
                }
            }
        }
        if (bias_wave && lh == 0 && p.db) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + wn * 32 * TN + ni * 32 + li;
                if (n < p.Co) p.db[n] = accb[ni][0];
            }
        }
        return;
    }
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
    if (bias_wave && lh == 0) {      // every row of ones^T dy is the column sum: row 0 lives in r = 0 of lanes 0..31
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n0 + wn * 32 * TN + ni * 32 + li;
            if (n < p.Co) slab[wcount + n] = accb[ni][0];
        }
    }
}

template <int WM, int WN, int TM, int TN, int NS>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(WgradArgsH p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    conv_wgrad_bf16_body<WM, WN, TM, TN, NS>(p, blockIdx.x, gridDim.x, smem);
}

// Several layers' weight gradients in ONE launch (the tail's layers: a handful of tiles each, every one of them a launch of
// 6-25 us plus a reduce of 6 us today): workgroup -> (layer, workgroup of the layer) through a prefix table; every layer has a
// single pixel split and takes the direct epilogue.  64 input x 128 output channels per workgroup, two stages.
constexpr int WGRAD_GROUP_MAX = 12;
struct WgradGroupArgs {
    int n;
    int wg0[WGRAD_GROUP_MAX + 1];
    WgradArgsH L[WGRAD_GROUP_MAX];
};
static_assert(sizeof(WgradGroupArgs) <= 4000, "kernel argument segment");
__global__ __launch_bounds__(256) void conv_wgrad_group_bf16_kernel(WgradGroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int l = 0;
    for (int k = 1; k < g.n; ++k)
        if ((int)blockIdx.x >= g.wg0[k]) l = k;
    conv_wgrad_bf16_body<2, 2, 1, 2, 2, false>(g.L[l], (int)blockIdx.x - g.wg0[l], g.wg0[l + 1] - g.wg0[l], smem);
}

// =================================================================================
// Kernel-row weight gradient (3x3, stride 1, SAME): one workgroup owns a whole KERNEL ROW, the three taps
// dw = -1, 0, +1.  The per-tap kernel above stages 2 tiles per tap; its 3x3 layers are bound by that tile DMA, not by
// the matrix cores.  Here dy is staged once for three taps and the three shifted x tiles collapse into ONE tile with
// a one-slot halo: 136 rows per 64 k-slots instead of 384.
// The k dimension walks the image in PADDED raster order, (W + 2) slots per image row with an empty slot before and
// after the W pixels.  A shift by +-1 from a pixel then lands on its real neighbour or on an empty slot, never in
// another image row, and empty slots are simply not fetched (LDS-DMA zero fill, for x and for dy): the zero padding
// costs no operand masking, and the price is W / (W + 2) of the k-slots doing work.
// =================================================================================
struct WgradRowsArgs {
    const bf16_t* x;
    const bf16_t* dy;
    float* ws;              // [nsplit][9*Ci*Co + Co]
    int B, H, W, Ci, Co;
    int pad_h;              // kernel row kh reads image row y + kh - pad_h
    int CT, NT;
    long long nslots;       // B * H * (W + 2)
    int schunk, nsplit;     // k-slots per split (multiple of 64)
};

template <int L>
__device__ __forceinline__ void wait_pieces(int ahead) {            // leave `ahead` stages of L pieces in flight
    static_assert(2 * L <= 63, "vmcnt field");
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int TM, int TN, int NS = 2>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_rows_kernel(WgradRowsArgs p) {
    constexpr int BKT = 64 * TM, BNT = 64 * TN, BP = 64, XROWS = 72;
    constexpr int XROWB = BKT * 2, YROWB = BNT * 2;
    constexpr int XCPR = BKT / 8, XRPP = 256 / XCPR, X_N = BP / XRPP;      // full staging passes over rows 0..63
    constexpr int XTAIL_WAVES = 8 * XCPR / 64;                             // waves that stage rows 64..71
    constexpr int YCPR = BNT / 8, YRPP = 256 / YCPR, Y_N = BP / YRPP;
    constexpr int X_LDS = XROWS * XROWB, STAGE = X_LDS + BP * YROWB;
    static_assert(TM * TN <= 2, "3 taps x TM x TN accumulator tiles per wave");
    static_assert(NS >= 2 && NS <= 4 && NS * STAGE <= 160 * 1024, "ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wgid = xcd_remap(blockIdx.x, gridDim.x);
    const int per_split = 3 * p.CT * p.NT;
    const int split = wgid / per_split;
    int rem = wgid - split * per_split;
    const int nt = rem % p.NT;
    rem /= p.NT;
    const int ct = rem % p.CT;
    const int kh = rem / p.CT;
    const int c0 = ct * BKT, n0 = nt * BNT;
    const int dh = kh - p.pad_h;
    const long long s0 = (long long)split * p.schunk;
    const long long s1l = s0 + p.schunk < p.nslots ? s0 + p.schunk : p.nslots;
    const int nvalid = (int)(s1l - s0);                     // slots of this split
    const int niter = (nvalid + BP - 1) / BP;
    const bool do_bias = kh == 0 && ct == 0;

    auto swz = [](int r, int cpr) { return cpr >= 16 ? (r & 3) * 4 : ((r >> 1) & 1) * 4; };
    const int WP = p.W + 2;
    // slot walk: +64 slots per iteration = (adv_b, adv_y, adv_x) in the mixed radix (image, row, padded column)
    const int adv_x = BP % WP, adv_t = BP / WP;
    const int adv_y = adv_t % p.H;
    const int qadv = adv_t * p.W + adv_x;                   // pixels the linear (b, y, x) index moves, before the carry fix

    // ---- staging rows of this thread: x rows xr + XRPP j (slot s0 - 1 + row), the tail row 64 + .., dy rows yr + YRPP j
    const int xr = tid / XCPR, xs = tid % XCPR, yr = tid / YCPR, ys = tid % YCPR;
    const int xchunk = xs ^ swz(xr, XCPR), ychunk = ys ^ swz(yr, YCPR);
    const int xtchunk = xs ^ swz(64 + xr, XCPR);
    const unsigned xcm = 0u - (unsigned)(c0 + xchunk * 8 < p.Ci), xtcm = 0u - (unsigned)(c0 + xtchunk * 8 < p.Ci);
    const unsigned ycm = 0u - (unsigned)(n0 + ychunk * 8 < p.Co);
    struct Walk { int xp, y, q; };                          // padded column, image row, linear pixel index (b*H + y)*W + xp - 1
    auto walk_init = [&](long long slot) {
        Walk w;
        if (slot < 0) {                                     // slot -1 (halo of the very first tile): the empty slot that ends "row -1";
            w.xp = WP - 1; w.y = p.H - 1; w.q = -p.W + WP - 2;   // floor division keeps the walk's q linear in the slot
            return w;
        }
        w.xp = (int)(slot % WP);
        const long long t = slot / WP;
        w.y = (int)(t % p.H);
        w.q = (int)(t * p.W) + w.xp - 1;
        return w;
    };
    auto walk_step = [&](Walk& w) {
        w.xp += adv_x;
        const bool cx = w.xp >= WP;
        w.xp -= cx ? WP : 0;
        w.y += adv_y + (cx ? 1 : 0);
        w.y -= w.y >= p.H ? p.H : 0;
        w.q += qadv - (cx ? 2 : 0);
    };
    Walk xw[X_N], xtw, yw[Y_N];
#pragma unroll
    for (int j = 0; j < X_N; ++j) xw[j] = walk_init(s0 - 1 + xr + j * XRPP);
    xtw = walk_init(s0 - 1 + 64 + xr);
#pragma unroll
    for (int j = 0; j < Y_N; ++j) yw[j] = walk_init(s0 + yr + j * YRPP);

    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, (unsigned)((size_t)p.B * p.H * p.W * p.Ci * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dy), 0, (unsigned)((size_t)p.B * p.H * p.W * p.Co * 2u), 0x00020000);
    const int xconst = (dh * p.W) * p.Ci + c0, yconst = n0;

    // x tile row `row` = slot (iteration base) - 1 + row; real iff it is a pixel column, the shifted image row exists and
    // the slot belongs to this split's range (+ the one-slot halo on either side)
    auto issue = [&](int it, int stage) {
        unsigned char* Xs = smem + stage * STAGE + wave * 1024;
        unsigned char* Ys = smem + stage * STAGE + X_LDS + wave * 1024;
        const int left = nvalid - it * BP;                 // slots of this tile inside the split
#pragma unroll
        for (int j = 0; j < X_N; ++j) {
            const int row = xr + j * XRPP;
            const bool ok = (unsigned)(xw[j].xp - 1) < (unsigned)p.W && (unsigned)(xw[j].y + dh) < (unsigned)p.H && row <= left + 1;
            const unsigned mk = xcm & (0u - (unsigned)ok);
            const unsigned off = (unsigned)((xw[j].q * p.Ci + xconst + xchunk * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, LDS_PTR(Xs + j * 4096), 16, (int)((off & mk) | (OOBH & ~mk)), 0, 0, 0);
            walk_step(xw[j]);
        }
        if (wave < XTAIL_WAVES) {
            const int row = 64 + xr;
            const bool ok = (unsigned)(xtw.xp - 1) < (unsigned)p.W && (unsigned)(xtw.y + dh) < (unsigned)p.H && row <= left + 1;
            const unsigned mk = xtcm & (0u - (unsigned)ok);
            const unsigned off = (unsigned)((xtw.q * p.Ci + xconst + xtchunk * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, LDS_PTR(Xs + X_N * 4096), 16, (int)((off & mk) | (OOBH & ~mk)), 0, 0, 0);
        }
        walk_step(xtw);
#pragma unroll
        for (int j = 0; j < Y_N; ++j) {
            const int row = yr + j * YRPP;
            const bool ok = (unsigned)(yw[j].xp - 1) < (unsigned)p.W && row < left;
            const unsigned mk = ycm & (0u - (unsigned)ok);
            const unsigned off = (unsigned)((yw[j].q * p.Co + yconst + ychunk * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, LDS_PTR(Ys + j * 4096), 16, (int)((off & mk) | (OOBH & ~mk)), 0, 0, 0);
            walk_step(yw[j]);
        }
    };

    f32x16 acc[3][TM][TN];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][a][b][r] = 0.f;
    float bsum = 0.f;

    const int wm = wave >> 1, wn = wave & 1;               // 2 x 2 waves: 32*TM input channels x 32*TN output channels each
    const int li = lane & 31, lh = lane >> 5;
    const int q = lane & 15, cb = (lane >> 4) & 1;
    const int prow = lh * 8 + (q >> 2);
    int xa[3][TM], ya[TN];
#pragma unroll
    for (int t = 0; t < 3; ++t)                              // tap t = dw + 1 reads x tile rows j + t
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int r = prow + t;
            const int ch = (wm * 32 * TM + mi * 32) / 8 + cb * 2 + ((q >> 1) & 1);
            xa[t][mi] = r * XROWB + ((ch ^ swz(r, XCPR)) * 16) + (q & 1) * 8;
        }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int ch = (wn * 32 * TN + ni * 32) / 8 + cb * 2 + ((q >> 1) & 1);
        ya[ni] = X_LDS + prow * YROWB + ((ch ^ swz(prow, YCPR)) * 16) + (q & 1) * 8;
    }

    auto compute = [&](int stage) {
        const unsigned char* S = smem + stage * STAGE;
        auto tr8 = [&](int off, int rowb) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)LDS_PTR(S + off));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)LDS_PTR(S + off + 4 * rowb));
            return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
#pragma unroll
        for (int st = 0; st < BP / 16; ++st) {
            s16x8 a[3][TM], b[TN];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) a[t][mi] = tr8(xa[t][mi] + st * 16 * XROWB, XROWB);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) b[ni] = tr8(ya[ni] + st * 16 * YROWB, YROWB);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[t][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t][mi]),
                                                                                __builtin_bit_cast(bf16x8, b[ni]), acc[t][mi][ni], 0, 0, 0);
        }
        if (do_bias && tid < BNT) {
            const unsigned char* Ys = S + X_LDS;
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < BP; ++r) {
                const int slot = (tid >> 3) ^ swz(r, YCPR);
                s += bf2f(*reinterpret_cast<const unsigned short*>(Ys + r * YROWB + slot * 16 + (tid & 7) * 2));
            }
            bsum += s;
        }
    };

    // NS = 2 (default): one tile in flight while one is multiplied -- a stage is 12..24 MFMAs per wave (400..800 cycles)
    // against a DMA round trip of 1300..2500, so a workgroup idles most of the time and the CU's other resident
    // workgroups fill in.  NS = 3 / 4 (a switch until round 5): a ring with two / three tiles in flight per workgroup -- measured
    // SLOWER (profiles/r03_o_rows_wgrad_ring_sweep_bf16.txt: conv1_2 0.374 -> 0.42..0.46 -> 0.52 ms, conv2_1 0.206 -> 0.208 ->
    // 0.283): the deeper ring costs resident workgroups (4 -> 3 -> 2 per CU), and four independent two-stage pipelines hide
    // more than two four-stage ones -- the same answer the gather kernels gave in round 1.
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < niter) issue(t, t);
    int st_c = 0, st_i = NS - 1;
    for (int it = 0; it < niter; ++it) {
        const int later = niter - 1 - it;
        const int ahead = later < NS - 2 ? later : NS - 2;
        if (wave < XTAIL_WAVES) wait_pieces<X_N + 1 + Y_N>(ahead);
        else wait_pieces<X_N + Y_N>(ahead);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + NS - 1 < niter) issue(it + NS - 1, st_i);
        compute(st_c);
        st_c = st_c + 1 == NS ? 0 : st_c + 1;
        st_i = st_i + 1 == NS ? 0 : st_i + 1;
    }

    const size_t wcount = (size_t)9 * p.Ci * p.Co;
    float* slab = p.ws + (size_t)split * (wcount + p.Co);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + wn * 32 * TN + ni * 32 + li;
                if (n >= p.Co) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = c0 + wm * 32 * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (ci >= p.Ci) continue;
                    slab[((size_t)(kh * 3 + t) * p.Ci + ci) * p.Co + n] = acc[t][mi][ni][r];
                }
            }
    if (do_bias && tid < BNT && n0 + tid < p.Co) slab[wcount + n0 + tid] = bsum;
}

// =================================================================================
// Column-walk weight gradient for the <= 64-input-channel layers (conv1_2, conv2_1), round 3.
// The kernel-row variant above gives one workgroup ONE kernel row kh: the three workgroups of a pixel range read the same dy
// tiles and x tiles that are the same pixels one image row apart.  With 1024 workgroups streaming at once an XCD's 4-MB L2
// does not hold a tile for the ~5 iterations until the sibling kernel row needs it: rocprofv3 FETCH_SIZE of conv1_2's weight
// gradient is 1.8 GB per launch against 0.74 GB of tensors (profiles/r03_g_bf16_pmc_FETCH_SIZE.txt) -- the layer runs at the HBM
// roofline of 2.4x its own bytes.  The reduction index of a weight gradient is the pixel, in ANY order, so here a workgroup
// owns a strip of 64 image columns and walks DOWN its rows: per step it stages one x row tile (66 pixels with the halo) into
// a ring of five and one dy row tile (ring of three), and multiplies all NINE taps -- x rows y-1, y, y+1 are the ring's previous fills.
// Every byte of x and dy is fetched once (+ 2 halo columns per 64, + 2 halo rows per unit), 36 MFMAs per wave follow 17 KB of
// LDS-DMA (kernel-row: 12), and the addressing is a constant per thread plus one row stride per step (no slot walk).
// 4 waves as 2 x 2: each 32 input x 32 output channels x 9 taps (nine accumulator tiles); Co > 64 runs as NT column tiles of 64
// (x re-read per tile).  Slabs: one per (image, strip, row chunk) unit, reduced by wgrad_reduce like the other variants'.
// =================================================================================
struct WgradColArgs {
    const bf16_t* x;
    const bf16_t* dy;
    float* ws;              // [nunits][9*Ci*Co + Co]
    int B, H, W, Ci, Co;
    int NT;                 // output-channel tiles of 64
    int nstrips, rpu, RC;   // 64-column strips per image row, image rows per unit, units per (image, strip)
};

// (two waves per SIMD = two workgroups per CU: left alone the scheduler hoists all 36 fragment reads of a step -- 316 registers)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_wgrad_bf16_col_kernel(WgradColArgs p) {
    constexpr int ROWB = 128, XROWS = 72, X_TILE = XROWS * ROWB, Y_TILE = 64 * ROWB, NXR = 5, NYB = 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [NXR x tiles][NYB dy tiles]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = wg % p.NT, unit = wg / p.NT;
    const int rc = unit % p.RC;
    const int us = unit / p.RC;
    const int strip = us % p.nstrips, b = us / p.nstrips;
    const int r0 = rc * p.rpu, r1 = r0 + p.rpu < p.H ? r0 + p.rpu : p.H;
    const int c0x = strip * 64, n0 = nt * 64;

    auto swz = [](int r) { return ((r >> 1) & 1) * 4; };       // 128-byte rows: slot s of row r holds chunk s ^ swz(r)
    // ---- staging: thread -> tile row (tid >> 3) + 32 j, 16-byte slot tid & 7; the COLUMN of a row is fixed for the whole walk
    const int sr = tid >> 3, ss = tid & 7;
    unsigned xoff[2], xmk[2], yoff[2], ymk[2], xtoff, xtmk;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = sr + 32 * j;
        const int xcol = c0x - 1 + row, xchunk = ss ^ swz(row);
        xmk[j] = 0u - (unsigned)((unsigned)xcol < (unsigned)p.W && xchunk * 8 < p.Ci);
        xoff[j] = (unsigned)((xcol * p.Ci + xchunk * 8) * 2);
        const int ycol = c0x + row, ychunk = ss ^ swz(row);
        ymk[j] = 0u - (unsigned)(ycol < p.W && n0 + ychunk * 8 < p.Co);
        yoff[j] = (unsigned)((ycol * p.Co + n0 + ychunk * 8) * 2);
    }
    {
        const int row = 64 + sr;                               // wave 0 stages rows 64..71; the halo needs 64 and 65
        const int xcol = c0x - 1 + row, xchunk = ss ^ swz(row);
        xtmk = 0u - (unsigned)(row <= 65 && (unsigned)xcol < (unsigned)p.W && xchunk * 8 < p.Ci);
        xtoff = (unsigned)((xcol * p.Ci + xchunk * 8) * 2);
    }
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, (unsigned)((size_t)p.B * p.H * p.W * p.Ci * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dy), 0, (unsigned)((size_t)p.B * p.H * p.W * p.Co * 2u), 0x00020000);
    const unsigned x_row_bytes = (unsigned)(p.W * p.Ci * 2), y_row_bytes = (unsigned)(p.W * p.Co * 2);

    // ring slots: x row yy lives in slot (yy - (r0 - 1)) % 5, dy row y in buffer (y - r0) % 3 (two steps are in flight)
    auto xslot = [&](int yy) { return (yy - (r0 - 1)) % NXR; };
    auto yslot = [&](int y) { return (y - r0) % NYB; };
    auto issue_x = [&](int yy) {                               // image row yy (may lie outside the image: zero tile)
        unsigned char* Xs = smem + xslot(yy) * X_TILE + wave * 1024;
        const unsigned ok = 0u - (unsigned)((unsigned)yy < (unsigned)p.H);
        const unsigned base = (unsigned)(b * p.H + yy) * x_row_bytes;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned mk = xmk[j] & ok;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, LDS_PTR(Xs + j * 4096), 16, (int)(((base + xoff[j]) & mk) | (OOBH & ~mk)), 0, 0, 0);
        }
        if (wave == 0) {
            const unsigned mk = xtmk & ok;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, LDS_PTR(Xs + 2 * 4096), 16, (int)(((base + xtoff) & mk) | (OOBH & ~mk)), 0, 0, 0);
        }
    };
    auto issue_y = [&](int y) {                                // rows r0 <= y < r1 only
        unsigned char* Ys = smem + NXR * X_TILE + yslot(y) * Y_TILE + wave * 1024;
        const unsigned base = (unsigned)(b * p.H + y) * y_row_bytes;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, LDS_PTR(Ys + j * 4096), 16, (int)(((base + yoff[j]) & ymk[j]) | (OOBH & ~ymk[j])), 0, 0, 0);
    };

    f32x16 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;
    f32x16 accb;                                               // bias gradient on the matrix cores: ones^T dy (waves with wm == 0)
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    s16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (short)0x3F80;       // bf16 1.0

    const int wm = wave >> 1, wn = wave & 1;               // 2 x 2 waves: 32 input channels x 32 output channels each
    const int li = lane & 31, lh = lane >> 5;
    const int q = lane & 15, cb = (lane >> 4) & 1;
    const int prow = lh * 8 + (q >> 2);
    int xa[3], ya;
#pragma unroll
    for (int t = 0; t < 3; ++t) {                           // tap kw = t reads x tile rows (pixel + t): tile row 0 is column c0x - 1
        const int r = prow + t;
        const int ch = wm * 4 + cb * 2 + ((q >> 1) & 1);
        xa[t] = r * ROWB + ((ch ^ swz(r)) * 16) + (q & 1) * 8;
    }
    {
        const int ch = wn * 4 + cb * 2 + ((q >> 1) & 1);
        ya = prow * ROWB + ((ch ^ swz(prow)) * 16) + (q & 1) * 8;
    }
    auto tr8 = [&](const unsigned char* S, int off) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)LDS_PTR(S + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)LDS_PTR(S + off + 4 * ROWB));
        return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    // A step = 12 blocks (k-step st, kernel row a) of three MFMAs (+ the bias MFMA); the fragments of block k + 1 are read
    // while block k multiplies (two register sets, the order pinned with sched_barrier as in the other kernels: left alone
    // the scheduler either hoists all 40 fragment reads -- 316 registers -- or parks every block on its own reads).  Rows
    // outside the image are ZERO tiles (issue_x masks their loads), so no block is conditional.
    auto compute = [&](int y) {
        const unsigned char* Ys = smem + NXR * X_TILE + yslot(y) * Y_TILE;
        const unsigned char* Xr[3] = {smem + xslot(y - 1) * X_TILE, smem + xslot(y) * X_TILE, smem + xslot(y + 1) * X_TILE};
        s16x8 afr[2][3], bfr[2];
        auto load_block = [&](int k) {
            const int st = k / 3, a = k % 3;
            if (a == 0) bfr[st & 1] = tr8(Ys, ya + st * 16 * ROWB);
#pragma unroll
            for (int t = 0; t < 3; ++t) afr[k & 1][t] = tr8(Xr[a], xa[t] + st * 16 * ROWB);
        };
        load_block(0);
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int st = k / 3, a = k % 3;
            if (k + 1 < 12) load_block(k + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 3; ++t)
                acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[k & 1][t]), __builtin_bit_cast(bf16x8, bfr[st & 1]),
                                                                   acc[a][t], 0, 0, 0);
            if (a == 0 && wm == 0)      // (a scalar LDS sweep of the tile by one wave made that wave every step's straggler)
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, bfr[st & 1]), accb, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- the walk.  Bytes in flight set the rate (a workgroup that keeps ONE 17-KB step in flight: 512 x 17 KB / 3 us of loaded
    // round trip = 2.9 TB/s, measured 0.32 ms): two steps stay in flight.  Step y multiplies x rows y-1, y, y+1 with dy row y
    // while x row y+2 / dy row y+1 are landing and x row y+3 / dy row y+2 are being issued.
    issue_x(r0 - 1); issue_x(r0); issue_x(r0 + 1); issue_y(r0);
    if (r0 + 1 < r1) { issue_x(r0 + 2); issue_y(r0 + 1); }
    for (int y = r0; y < r1; ++y) {
        // everything but the most recent step's group (if there is one) has to have landed
        if (y + 1 < r1) {
            if (wave == 0) wait_pieces<5>(1); else wait_pieces<4>(1);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (y + 2 < r1) { issue_x(y + 3); issue_y(y + 2); }
        compute(y);
    }

    const size_t wcount = (size_t)9 * p.Ci * p.Co;
    float* slab = p.ws + (size_t)unit * (wcount + p.Co);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int n = n0 + wn * 32 + li;
            if (n >= p.Co) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ci >= p.Ci) continue;
                slab[((size_t)(a * 3 + t) * p.Ci + ci) * p.Co + n] = acc[a][t][r];
            }
        }
    if (wm == 0 && lh == 0) {      // every row of ones^T dy is the column sum: row 0 lives in r = 0 of lanes 0..31
        const int n = n0 + wn * 32 + li;
        if (n < p.Co) slab[wcount + n] = accb[0];
    }
}

// =================================================================================
// Kernel-row weight gradient, 8 waves: 128 input channels x 128 output channels x the three taps of a kernel row per
// workgroup (padded-raster k order as above), ONE workgroup per CU, an NS-deep ring of 34-KB stages.
// Why: a CU's LDS-DMA path moves ~55 B/clk at best (tools/probes/dma_rate.hip) and every 1-KB piece costs its issuing
// wave 60..185 cycles.  The 4-wave kernels above need 64 B/clk to keep the matrix cores busy (32 KB per 16 MFMAs per
// wave, two workgroups per CU) and run at ~40 % of peak; this tile needs 22 B/clk (34 KB per 24 MFMAs per wave) and
// 4..5 pieces per wave and stage instead of 8.  Each wave owns 32 input channels x 64 output channels x 3 taps
// (6 accumulator tiles); the bias gradient rides on the matrix cores too (a ones operand against the dy fragments in
// the two waves of the kh = 0, ct = 0 workgroups) instead of a scalar LDS sweep.
// =================================================================================
template <int NS, bool STAGGER = true>
__global__ __launch_bounds__(512) void conv_wgrad_bf16_rows8_kernel(WgradRowsArgs p) {
    constexpr int BKT = 128, BNT = 128, BP = 64, XROWS = 72, ROWB = 256, CPR = 16;
    constexpr int RPP = 512 / CPR;                              // 32 pixel rows per staging pass of the workgroup
    constexpr int X_N = BP / RPP, Y_N = BP / RPP;               // 2 + 2 full passes; rows 64..71 of x: waves 0 and 1
    constexpr int X_LDS = XROWS * ROWB, STAGE = X_LDS + BP * ROWB;
    static_assert(NS >= 2 && NS <= 4 && NS * STAGE <= 160 * 1024, "ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wgid = xcd_remap(blockIdx.x, gridDim.x);
    const int per_split = 3 * p.CT * p.NT;
    const int split = wgid / per_split;
    int rem = wgid - split * per_split;
    const int nt = rem % p.NT;
    rem /= p.NT;
    const int ct = rem % p.CT;
    const int kh = rem / p.CT;
    const int c0 = ct * BKT, n0 = nt * BNT;
    const int dh = kh - p.pad_h;
    const long long s0 = (long long)split * p.schunk;
    const long long s1l = s0 + p.schunk < p.nslots ? s0 + p.schunk : p.nslots;
    const int nvalid = (int)(s1l - s0);
    const int niter = (nvalid + BP - 1) / BP;
    const bool do_bias = kh == 0 && ct == 0;

    const int WP = p.W + 2;
    const int adv_x = BP % WP, adv_t = BP / WP;
    const int adv_y = adv_t % p.H;
    const int qadv = adv_t * p.W + adv_x;

    // staging: thread -> (pixel row sr of a pass, 16-byte slot); both tiles have 256-byte rows, swizzle p ^ 4 (r & 3)
    const int sr = tid >> 4, ss = tid & 15;
    const int chunk = ss ^ ((sr & 3) * 4);                    // RPP and 64 are multiples of 4: one source chunk per thread
    const unsigned xcm = 0u - (unsigned)(c0 + chunk * 8 < p.Ci), ycm = 0u - (unsigned)(n0 + chunk * 8 < p.Co);
    struct Walk { int xp, y, q; };
    auto walk_init = [&](long long slot) {
        Walk w;
        if (slot < 0) {
            w.xp = WP - 1; w.y = p.H - 1; w.q = -p.W + WP - 2;
            return w;
        }
        w.xp = (int)(slot % WP);
        const long long t = slot / WP;
        w.y = (int)(t % p.H);
        w.q = (int)(t * p.W) + w.xp - 1;
        return w;
    };
    auto walk_step = [&](Walk& w) {
        w.xp += adv_x;
        const bool cx = w.xp >= WP;
        w.xp -= cx ? WP : 0;
        w.y += adv_y + (cx ? 1 : 0);
        w.y -= w.y >= p.H ? p.H : 0;
        w.q += qadv - (cx ? 2 : 0);
    };
    Walk xw[X_N], xtw, yw[Y_N];
#pragma unroll
    for (int j = 0; j < X_N; ++j) xw[j] = walk_init(s0 - 1 + sr + j * RPP);
    xtw = walk_init(s0 - 1 + 64 + sr);
#pragma unroll
    for (int j = 0; j < Y_N; ++j) yw[j] = walk_init(s0 + sr + j * RPP);

    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, (unsigned)((size_t)p.B * p.H * p.W * p.Ci * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dy), 0, (unsigned)((size_t)p.B * p.H * p.W * p.Co * 2u), 0x00020000);
    const int xconst = (dh * p.W) * p.Ci + c0 + chunk * 8, yconst = n0 + chunk * 8;

    auto issue = [&](int it, int stage) {
        unsigned char* Xs = smem + stage * STAGE + wave * 1024;
        unsigned char* Ys = Xs + X_LDS;
        const int left = nvalid - it * BP;
#pragma unroll
        for (int j = 0; j < X_N; ++j) {
            const int row = sr + j * RPP;
            const bool ok = (unsigned)(xw[j].xp - 1) < (unsigned)p.W && (unsigned)(xw[j].y + dh) < (unsigned)p.H && row <= left + 1;
            const unsigned mk = xcm & (0u - (unsigned)ok);
            const unsigned off = (unsigned)((xw[j].q * p.Ci + xconst) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, LDS_PTR(Xs + j * (RPP * ROWB)), 16, (int)((off & mk) | (OOBH & ~mk)), 0, 0, 0);
            walk_step(xw[j]);
        }
        if (wave < 2) {                                    // rows 64..71 (the halo needs 64 and 65)
            const int row = 64 + sr;
            const bool ok = (unsigned)(xtw.xp - 1) < (unsigned)p.W && (unsigned)(xtw.y + dh) < (unsigned)p.H && row <= left + 1;
            const unsigned mk = xcm & (0u - (unsigned)ok);
            const unsigned off = (unsigned)((xtw.q * p.Ci + xconst) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, LDS_PTR(Xs + X_N * (RPP * ROWB)), 16, (int)((off & mk) | (OOBH & ~mk)), 0, 0, 0);
            walk_step(xtw);
        }
#pragma unroll
        for (int j = 0; j < Y_N; ++j) {
            const int row = sr + j * RPP;
            const bool ok = (unsigned)(yw[j].xp - 1) < (unsigned)p.W && row < left;
            const unsigned mk = ycm & (0u - (unsigned)ok);
            const unsigned off = (unsigned)((yw[j].q * p.Co + yconst) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, LDS_PTR(Ys + j * (RPP * ROWB)), 16, (int)((off & mk) | (OOBH & ~mk)), 0, 0, 0);
            walk_step(yw[j]);
        }
    };

    f32x16 acc[3][2], accb[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t][0][r] = acc[t][1][r] = 0.f;
        accb[0][r] = accb[1][r] = 0.f;
    }

    const int wm = wave >> 1, wn = wave & 1;               // 4 x 2 waves: 32 input channels x 64 output channels each
    const bool bias_wave = do_bias && wm == 0;
    const int li = lane & 31, lh = lane >> 5;
    const int q = lane & 15, cb = (lane >> 4) & 1;
    const int prow = lh * 8 + (q >> 2);
    int xa[3], ya[2];
#pragma unroll
    for (int t = 0; t < 3; ++t) {                           // tap t = dw + 1 reads x tile rows j + t
        const int r = prow + t;
        const int ch = wm * 4 + cb * 2 + ((q >> 1) & 1);
        xa[t] = r * ROWB + ((ch ^ ((r & 3) * 4)) * 16) + (q & 1) * 8;
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int ch = wn * 8 + ni * 4 + cb * 2 + ((q >> 1) & 1);
        ya[ni] = X_LDS + prow * ROWB + ((ch ^ ((prow & 3) * 4)) * 16) + (q & 1) * 8;
    }
    s16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (short)0x3F80;   // bf16 1.0

    auto compute = [&](int stage) {
        const unsigned char* S = smem + stage * STAGE;
        s16x8 a[2][3], b[2][2];
        auto tr8 = [&](int off) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)LDS_PTR(S + off));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)LDS_PTR(S + off + 4 * ROWB));
            return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        auto frags = [&](int st) {
#pragma unroll
            for (int t = 0; t < 3; ++t) a[st & 1][t] = tr8(xa[t] + st * 16 * ROWB);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[st & 1][ni] = tr8(ya[ni] + st * 16 * ROWB);
        };
        frags(0);
#pragma unroll
        for (int st = 0; st < BP / 16; ++st) {
            if (st + 1 < BP / 16) frags(st + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[t][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[st & 1][t]),
                                                                        __builtin_bit_cast(bf16x8, b[st & 1][ni]), acc[t][ni], 0, 0, 0);
            if (bias_wave) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    accb[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, b[st & 1][ni]),
                                                                      accb[ni], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < niter) issue(t, t);
    int st_c = 0, st_i = NS - 1;
    for (int it = 0; it < niter; ++it) {
        const int later = niter - 1 - it;
        const int ahead = later < NS - 2 ? later : NS - 2;
        if (wave < 2) wait_pieces<X_N + 1 + Y_N>(ahead);
        else wait_pieces<X_N + Y_N>(ahead);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // The issue phase (address walk + 4..5 DMA pieces, ~1200 cycles) is as long as the 24 MFMAs of a stage, and
        // with one workgroup per CU all 8 waves would be in it at the same time.  With a ring of >= 3 stages the target
        // buffer is free for the whole iteration, so waves 4..7 (the SIMD partners of waves 0..3) multiply first and
        // issue afterwards: on every SIMD one wave feeds the matrix pipe while the other one issues.
        const bool late = NS >= 3 && STAGGER && wave >= 4;
        if (!late && it + NS - 1 < niter) issue(it + NS - 1, st_i);
        compute(st_c);
        if (late && it + NS - 1 < niter) issue(it + NS - 1, st_i);
        st_c = st_c + 1 == NS ? 0 : st_c + 1;
        st_i = st_i + 1 == NS ? 0 : st_i + 1;
    }

    const size_t wcount = (size_t)9 * p.Ci * p.Co;
    float* slab = p.ws + (size_t)split * (wcount + p.Co);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + wn * 64 + ni * 32 + li;
            if (n >= p.Co) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = c0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ci >= p.Ci) continue;
                slab[((size_t)(kh * 3 + t) * p.Ci + ci) * p.Co + n] = acc[t][ni][r];
            }
        }
    if (bias_wave && lh == 0) {                              // every row of ones^T dy is the column sum: row 0 lives in r = 0 of lanes 0..31
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + wn * 64 + ni * 32 + li;
            if (n < p.Co) slab[wcount + n] = accb[ni][0];
        }
    }
}

// =================================================================================
// filter mirrors: fp32 [tap][Ci][Co] -> bf16 [tap][Ci][Co] and bf16 [tap][Co][Ci], all layers in one launch
// =================================================================================
struct CastTable {
    int n;
    struct Seg {
        unsigned long long off;     // element offset of the layer's filter in all three arrays
        int taps, ci, co;
        int cit, cot;               // 32x32 tiles per tap
        int blk0;                   // first block of this layer
    } seg[FilterCastPlan::MAX_LAYERS];
};

__global__ __launch_bounds__(256) void cast_filters_kernel(CastTable t, const float* __restrict__ w, bf16_t* __restrict__ io,
                                                           bf16_t* __restrict__ oi) {
    __shared__ float tile[32][33];
    int s = 0;
    while (s + 1 < t.n && (int)blockIdx.x >= t.seg[s + 1].blk0) ++s;
    const CastTable::Seg g = t.seg[s];
    int b = blockIdx.x - g.blk0;
    const int cot = b % g.cot;
    b /= g.cot;
    const int cit = b % g.cit;
    const int tap = b / g.cit;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t base = g.off + (size_t)tap * g.ci * g.co;
    float v[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                      // the four loads first (clamped), then their uses
        const int ci = cit * 32 + ty + 8 * q, co = cot * 32 + tx;
        ok[q] = ci < g.ci && co < g.co;
        v[q] = w[base + (size_t)(ok[q] ? ci : 0) * g.co + (ok[q] ? co : 0)];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ci = cit * 32 + ty + 8 * q, co = cot * 32 + tx;
        if (ok[q]) io[base + (size_t)ci * g.co + co].v = f2bf(v[q]);
        tile[ty + 8 * q][tx] = ok[q] ? v[q] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int co = cot * 32 + r, ci = cit * 32 + tx;
        if (ci < g.ci && co < g.co) oi[base + (size_t)co * g.ci + ci].v = f2bf(tile[tx][r]);
    }
}

void FilterCastPlan::add(size_t off, int taps, int ci, int co) {
    SSD_REQUIRE(n < MAX_LAYERS, "too many conv layers for the filter cast table");
    L[n].off = off; L[n].taps = taps; L[n].ci = ci; L[n].co = co;
    ++n;
}

void cast_filters(const FilterCastPlan& plan, const float* w, bf16_t* io, bf16_t* oi, hipStream_t s) {
    CastTable t{};
    t.n = plan.n;
    int blk = 0;
    double elems = 0;
    for (int i = 0; i < plan.n; ++i) {
        t.seg[i].off = plan.L[i].off; t.seg[i].taps = plan.L[i].taps; t.seg[i].ci = plan.L[i].ci; t.seg[i].co = plan.L[i].co;
        t.seg[i].cit = cdiv(plan.L[i].ci, 32); t.seg[i].cot = cdiv(plan.L[i].co, 32);
        t.seg[i].blk0 = blk;
        blk += plan.L[i].taps * t.seg[i].cit * t.seg[i].cot;
        elems += (double)plan.L[i].taps * plan.L[i].ci * plan.L[i].co;
    }
    if (blk == 0) return;
    ProfScope prof("cast_filters", 0.0, 8.0 * elems, s);
    hipLaunchKernelGGL(cast_filters_kernel, dim3(blk), dim3(256), 0, s, t, w, io, oi);
    HIP_OK(hipGetLastError());
}

// =================================================================================
// host launchers
// =================================================================================
template <int MODE, int WM, int WN, int TM, int TN, bool STRIDED, int NS = 2, bool PARITY = false, int KSPLIT = 1>
static void launch_gather_h(GatherArgsH& a, const char* label, double flops, double bytes, hipStream_t s, int grid = 0) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr size_t stages = KSPLIT * NS * (size_t)(BM + BN) * 128, ctile = (size_t)BM * (BN + 4) * 4;
    constexpr size_t lds = stages > ctile ? stages : ctile;
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = conv_gather_bf16_kernel<MODE, WM, WN, TM, TN, STRIDED, NS, PARITY, KSPLIT>;
    static bool once = (set_lds(kern, lds), true);
    (void)once;
    const int MT = cdiv(a.M, BM);
    a.NT = cdiv(a.DN, BN);
    ProfScope prof(label, flops, bytes, s);
    SSD_LAUNCH_STOP(kern, dim3(grid > 0 ? grid : MT * a.NT), dim3(64 * WM * WN * KSPLIT), lds, s, a);
    HIP_OK(hipGetLastError());
}

static void check_desc_h(const ConvDesc& d) {
    SSD_REQUIRE(d.KH * d.KW <= 9 && d.KH * d.KW >= 1, "conv: at most 9 taps (got %dx%d)", d.KH, d.KW);
    SSD_REQUIRE(d.Co % 8 == 0 && d.Ci % 8 == 0, "bf16 conv: Ci and Co must be multiples of 8 (got %d, %d)", d.Ci, d.Co);
    SSD_REQUIRE((long long)d.B * d.Hi * d.Wi * d.Ci < (1LL << 31) - 8 && (long long)d.B * d.Ho * d.Wo * d.Co < (1LL << 31) - 8,
                "bf16 conv: a tensor of this layer exceeds 4 GiB (32-bit byte offsets): lower the batch");
}

// Tile configurations (pixels x channels, pipeline stages -> workgroups resident per CU):
//   0: 128x128 x2 (2/CU)   1: 128x64 x2 (2/CU)   2: 64x128 x2 (2/CU)
//   3: 256x128 x3 (1/CU)   4: 128x128 x4 (1/CU)  5: 128x64 x3 (2/CU)
//   6: 256x128 x3, 8 waves (1/CU)   7: 256x128 x2, 8 waves (1/CU)   8: 256x64 x2, 8 waves (2/CU)
//   9: 64x64 x6 (1/CU): the latency-bound small layers (see pick_tile_h)   10: 64x64 x2, k split over 4 wave groups (16 waves, 1/CU)
constexpr int NCFG_H = 12;
static int pick_tile_h(long long M, int N, int mode, int nk = 0) {
    static const int forced = env_int("SSD_TILE_BF16", -1);      // tuning override
    if (forced >= 0 && forced < NCFG_H) return forced;
    static const int bm[NCFG_H] = {128, 128, 64, 256, 128, 128, 256, 256, 256, 64, 64, 64}, bn[NCFG_H] = {128, 64, 128, 128, 128, 64, 128, 128, 64, 64, 64, 64};
    static const int per_cu[NCFG_H] = {2, 2, 2, 1, 1, 2, 1, 1, 2, 1, 1, 2};
    // > 0: in the automatic choice.  256x64 (8 waves) serves the 64-channel layers: conv1_2 forward 427 -> 462, data gradient 423 -> 474 TF/s
    static const double eff[NCFG_H] = {1.0, 0.85, 0.85, 0.0, 0.0, 0.0, 0.0, 0.0, 0.93, 0.0, 0.0, 0.0};
    int best = 0;
    double bc = 1e300;
    for (int c = 0; c < NCFG_H; ++c) {
        if (eff[c] <= 0.0) continue;
        const long long wgs = (long long)cdiv(M, bm[c]) * cdiv(N, bn[c]);
        const int slots = 256 * per_cu[c];
        const double cost = (double)((wgs + slots - 1) / slots) * bm[c] * bn[c] * per_cu[c] / eff[c];
        if (cost < bc * 0.999) { bc = cost; best = c; }
    }
    // A launch of at most one workgroup per CU is latency-bound by its serial k loop with ONE tile in flight (the tail
    // conv8_1 ... conv11_2 and the small maps' heads: matrix-pipe duty 0.17); the same tile with 3 / 4 pipeline stages
    // keeps two or three tiles in flight -- LDS is no constraint when a CU holds a single workgroup.  Measured on one
    // box, two interleaved repetitions (profiles/r03_a_ab_small_deep_bf16.txt): step 7.591 / 7.561 -> 7.499 / 7.512 ms
    // (+0.9 %); up to two workgroups per CU: 7.580 / 7.600, no gain.
    {
        const long long wgs = (long long)cdiv(M, bm[best]) * cdiv(N, bn[best]);
        if (wgs <= 256) {
            if (best == 0) best = 4;           // 128x128 x4
            else if (best == 1) best = 5;      // 128x64 x3
            // Round 4: such a launch's time IS its serial k loop (conv9_2: 18 iterations, the 10x10 map's head: 72) at
            // ~0.7 us per iteration -- one DMA round trip for two tiles in flight.  64x64 tiles stage 16 KB per iteration
            // instead of 24 and a ring of SIX keeps four in flight; the launch also spreads over up to four times the
            // CUs.  Taken while it still fits one workgroup per CU.  SSD_SMALL_TILE=0 switches it off (A/B).
            static const int small_tile = env_int("SSD_SMALL_TILE", 1);
            if (small_tile && (long long)cdiv(M, 64) * cdiv(N, 64) <= 256) {
                best = 9;
                // ... and with at least 8 iterations to share, four wave groups split the k loop (conv_gather_bf16_kernel, KSPLIT):
                // SSD_SMALL_KSPLIT=0 keeps the deep ring
                static const int ksplit = env_int("SSD_SMALL_KSPLIT", 1);      // 1: four wave groups (16 waves, 128 KB of LDS: needs an EMPTY CU); 2: two groups (8 waves, 64 KB)
                if (ksplit == 2 && nk >= 8) best = 11;
                else if (ksplit && nk >= 8) best = 10;
            }
        }
    }
    return best;
}

template <int MODE>
static void launch_gather_cfg(int cfg, GatherArgsH& a, double fl, double by, hipStream_t s, bool unpool = false) {
    static const char* const names[2][NCFG_H] = {
        {"conv_fwd_bf16_128x128", "conv_fwd_bf16_128x64", "conv_fwd_bf16_64x128", "conv_fwd_bf16_256x128x3", "conv_fwd_bf16_128x128x4",
         "conv_fwd_bf16_128x64x3", "conv_fwd_bf16_256x128x3_8w", "conv_fwd_bf16_256x128x2_8w", "conv_fwd_bf16_256x64_8w", "conv_fwd_bf16_64x64x6",
         "conv_fwd_bf16_64x64_k4", "conv_fwd_bf16_64x64_k2"},
        {"conv_dgrad_bf16_128x128", "conv_dgrad_bf16_128x64", "conv_dgrad_bf16_64x128", "conv_dgrad_bf16_256x128x3",
         "conv_dgrad_bf16_128x128x4", "conv_dgrad_bf16_128x64x3", "conv_dgrad_bf16_256x128x3_8w", "conv_dgrad_bf16_256x128x2_8w",
         "conv_dgrad_bf16_256x64_8w", "conv_dgrad_bf16_64x64x6", "conv_dgrad_bf16_64x64_k4", "conv_dgrad_bf16_64x64_k2"}};
    static const char* const unpool_names[NCFG_H] = {
        "conv_dgrad_unpool_bf16_128x128", "conv_dgrad_unpool_bf16_128x64", "conv_dgrad_unpool_bf16_64x128", "conv_dgrad_unpool_bf16_256x128x3",
        "conv_dgrad_unpool_bf16_128x128x4", "conv_dgrad_unpool_bf16_128x64x3", "conv_dgrad_unpool_bf16_256x128x3_8w", "conv_dgrad_unpool_bf16_256x128x2_8w",
        "conv_dgrad_unpool_bf16_256x64_8w", "conv_dgrad_unpool_bf16_64x64x6", "conv_dgrad_unpool_bf16_64x64_k4", "conv_dgrad_unpool_bf16_64x64_k2"};
    const char* label = unpool ? unpool_names[cfg] : names[MODE][cfg];
    switch (cfg) {
    case 0: launch_gather_h<MODE, 2, 2, 2, 2, false, 2>(a, label, fl, by, s); break;
    case 1: launch_gather_h<MODE, 4, 1, 1, 2, false, 2>(a, label, fl, by, s); break;
    case 2: launch_gather_h<MODE, 2, 2, 1, 2, false, 2>(a, label, fl, by, s); break;
    case 3: launch_gather_h<MODE, 2, 2, 4, 2, false, 3>(a, label, fl, by, s); break;
    case 4: launch_gather_h<MODE, 2, 2, 2, 2, false, 4>(a, label, fl, by, s); break;
    case 5: launch_gather_h<MODE, 4, 1, 1, 2, false, 3>(a, label, fl, by, s); break;
    case 6: launch_gather_h<MODE, 4, 2, 2, 2, false, 3>(a, label, fl, by, s); break;
    case 7: launch_gather_h<MODE, 4, 2, 2, 2, false, 2>(a, label, fl, by, s); break;
    case 8: launch_gather_h<MODE, 8, 1, 1, 2, false, 2>(a, label, fl, by, s); break;
    case 9: launch_gather_h<MODE, 2, 2, 1, 1, false, 6>(a, label, fl, by, s); break;
    case 11: launch_gather_h<MODE, 2, 2, 1, 1, false, 2, false, 2>(a, label, fl, by, s); break;
    default: launch_gather_h<MODE, 2, 2, 1, 1, false, 2, false, 4>(a, label, fl, by, s); break;
    }
}

// ---- kernel-row gather dispatch ------------------------------------------------------------------------------
// SSD_GATHER_ROWS_BF16: 0 off, 1 (default) forward everywhere + dgrad of the undilated layers (measured +2..9 %; the
// dilated mod_conv6 dgrad measured -2 %), 2 everywhere.
static bool gather_rows_applicable(const ConvDesc& d, bool dgrad) {
    static const int on = env_int("SSD_GATHER_ROWS_BF16", 1);
    if (on == 1 && dgrad && d.dil != 1) return false;
    const int sc = dgrad ? d.Co : d.Ci;            // channels of the gathered tensor: whole 64-channel chunks only
    return on && sc % 64 == 0 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.Hi == d.Ho && d.Wi == d.Wo && d.dil >= 1 && d.dil <= 8 &&
           d.pad_h == d.dil && d.pad_w == d.dil;
}
// conv1_2-shaped layers (64 -> 64, 3x3, stride 1, SAME, bf16 out), large enough to give every CU several tiles
static bool gather_c64_applicable(const ConvDesc& d, bool y_f32) {
    static const int on = env_int("SSD_C64_BF16", 1);      // A/B switch; 2 = also small layers (tests)
    const bool shape = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.dil == 1 && d.pad_h == 1 && d.pad_w == 1 && d.Hi == d.Ho && d.Wi == d.Wo &&
                       d.Ci == 64 && d.Co == 64 && !y_f32;
    return on && shape && (on == 2 || (long long)d.B * d.Ho * d.Wo >= 253LL * 256 * 4);
}
template <int MODE>
static void launch_gather_c64(GatherArgsH& a, const char* label, double flops, double bytes, hipStream_t s) {
    // (Measured and not kept, gpurun r02_s: a 7-wave variant with THREE 28-KB activation buffers, two units in flight.
    // Same time to +-2 % -- the kernel is not waiting for its DMA; like the other bf16 gather kernels its time moves with
    // the DATA: 0.33 ms on random operands, 0.25 ms on the step's post-relu activations, i.e. with power and clock.)
    // (XCD-contiguous tile chunks: forward +2..7 %, step +0.3 %, profiles/r02_r; results leave through LDS as whole 128-byte
    // rows: +0.9 % on the step, profiles/r02_ai -- both were A/B switches until round 5)
    constexpr size_t lds = (size_t)9 * 64 * 128 + 2 * 256 * 128;
    const int ntiles = cdiv(a.M, 253);
    ProfScope prof(label, flops, bytes, s);
    auto kern = conv_gather_bf16_c64_kernel<MODE, false>;
    static bool once = (set_lds(kern, lds), true);
    (void)once;
    SSD_LAUNCH_STOP(kern, dim3(ntiles < 256 ? ntiles : 256), dim3(512), lds, s, a, ntiles, ntiles >= 2048 ? 1 : 0, FirstWArgs{});
    HIP_OK(hipGetLastError());
}

template <int MODE, int TM, int TN = 2>
static void launch_gather_rows(GatherArgsH& a, int dil, const char* label, double flops, double bytes, hipStream_t s) {
    constexpr int BN = 64 * TN;
    constexpr size_t unit = (TM == 2 ? 160 : 256) * 128 + 3 * BN * 128, ctile = (size_t)128 * (BN + 4) * 4;
    constexpr size_t lds = unit > ctile ? unit : ctile;
    auto kern = conv_gather_bf16_rows_kernel<MODE, TM, TN>;
    static bool once = (set_lds(kern, lds), true);
    (void)once;
    a.NT = cdiv(a.DN, BN);
    const int bmv = TM == 2 ? 128 : 256 - 2 * dil - 1;
    ProfScope prof(label, flops, bytes, s);
    SSD_LAUNCH_STOP(kern, dim3(cdiv(a.M, bmv) * a.NT), dim3(256), lds, s, a, dil);
    HIP_OK(hipGetLastError());
}
// 128 x 64 tiles (three workgroups per CU) -- where they cut PADDED COLUMNS: the fused 19x19 head (N = 152: three 64-wide tiles carry
// 79 % useful columns, two 128-wide ones 59 %; 361 -> 520 TFLOP/s sustained, not power-limited: 2.39 GHz either way).
// NOT where they merely fill the chip's workgroup slots better (conv5_x, mod_conv6: 364 workgroups on 512 slots against 728 on
// 768).  A single launch does run faster that way (67 -> 58 us) -- but back to back, as in a training step, the better-filled
// tile draws more power and the chip clocks down: conv5_2 forward 996 TFLOP/s at 2.22 GHz on 128 x 128 tiles against 949 TFLOP/s
// at 1.89 GHz on 128 x 64 (tools/power_probe.py conv5_2, profiles/r04_ae_power_clock_19x19_bf16.txt).  Idle CUs are not waste under
// a power limit; the staged bytes per MFMA are.  SSD_GATHER_ROWS_N64_BF16: 0 off, 1 by the padded-columns rule (default), 2
// everywhere (tests).
thread_local int g_conv_lanes = 1;
static bool gather_rows_n64(int M, int N) {
    static const int on = env_int("SSD_GATHER_ROWS_N64_BF16", 1);
    if (on == 0 || on == 2) return on == 2;
    const double waste128 = (double)N / (cdiv(N, 128) * 128), waste64 = (double)N / (cdiv(N, 64) * 64);      // useful columns
    (void)M;
    return waste64 > waste128 * 1.1;
}
// 256-row tiles where they fill the chip's 512 workgroup slots often enough; dil = 1 only (the tile owns 256 - 2 dil rows).
// Round 4: the count is taken over the launches that run side by side (the forward lanes: g_conv_lanes) and the bar is 700
// instead of 1024 workgroups -- under sustained load (profiles/r04_ag_sustained_tile_sweep_bf16.txt) the 256-row tile is ahead
// wherever it was tried: conv2_2 forward 957 vs 883 TFLOP/s, conv3_2 1122 vs 1077, conv4_2 1161 vs 1135 (732 workgroups); the
// half-batch launches of conv3_x in the forward lanes (712 each) had been on 128-row tiles.  SSD_GATHER_ROWS256_BF16: 0 off,
// 1 this rule, 2 every eligible layer (tests), 3 round 3's rule (1024 workgroups, per launch).
static bool gather_rows256(const ConvDesc& d, int M, int N) {
    static const int on = env_int("SSD_GATHER_ROWS256_BF16", 1);
    if (!on || d.dil != 1) return false;
    if (on == 2) return true;
    if (on == 3) return cdiv(M, 253) * cdiv(N, 128) >= 1024;
    return (long long)cdiv((long long)M * g_conv_lanes, 253) * cdiv(N, 128) >= 700;
}

void conv_fwd_bf16(const ConvDesc& d, const bf16_t* x, const bf16_t* w_oi, const float* bias, void* y, bool y_f32, bool relu,
                   hipStream_t s) {
    check_desc_h(d);
    GatherArgsH a{};
    a.src = x; a.wgt = w_oi; a.bias = bias; a.mask = nullptr; a.dst = y;
    a.M = d.B * d.Ho * d.Wo; a.DH = d.Ho; a.DW = d.Wo; a.DN = d.Co;
    a.SH = d.Hi; a.SW = d.Wi; a.SC = d.Ci;
    a.ntaps = d.KH * d.KW; a.mul = d.stride; a.div = 1;
    a.relu = relu; a.accum = 0; a.out_f32 = y_f32;
    for (int kh = 0; kh < d.KH; ++kh)
        for (int kw = 0; kw < d.KW; ++kw) {
            a.tap_dh[kh * d.KW + kw] = kh * d.dil - d.pad_h;
            a.tap_dw[kh * d.KW + kw] = kw * d.dil - d.pad_w;
        }
    const double fl = conv_flops(d), by = 2.0 * conv_elems(d);
    if (gather_c64_applicable(d, y_f32)) {
        launch_gather_c64<MODE_FWD>(a, "conv_fwd_bf16_c64", fl, by, s);
        return;
    }
    // (multibox heads of the small maps: a handful of workgroups, latency-bound -- the per-tap kernel's shorter units win:
    // head of the 10x10 map 45 -> 36 us, gpurun r02_b heads_tiles_bf16)
    // A/B switch: the fp32-out layers (multibox heads) below this many pixels take the per-tap kernel; 4096 keeps the
    // 19x19 map's head (M = 11552, N = 152: 182 kernel-row workgroups, 281 TFLOP/s) on the kernel-row gather
    constexpr int heads_rows_min_m = 4096;
    // (round 6: the 38x38 head has 104 output channels -- 4 box types x 25, rounded to 8 -- and ran on the per-tap kernel for want of
    // 128; the kernel-row gather zero-fills the filter rows past Co like any tile edge: forward 0.081 -> 0.068 ms, step +-0,
    // profiles/r06_v_*)
    constexpr int rows_min_co = 104;
    if (gather_rows_applicable(d, false) && d.Co >= rows_min_co && !(y_f32 && a.M < heads_rows_min_m)) {
        if (gather_rows256(d, a.M, d.Co)) launch_gather_rows<MODE_FWD, 4>(a, d.dil, "conv_fwd_bf16_rows_256x128", fl, by, s);
        else if (gather_rows_n64(a.M, d.Co)) launch_gather_rows<MODE_FWD, 2, 1>(a, d.dil, "conv_fwd_bf16_rows_128x64", fl, by, s);
        else launch_gather_rows<MODE_FWD, 2>(a, d.dil, "conv_fwd_bf16_rows_128x128", fl, by, s);
        return;
    }
    launch_gather_cfg<MODE_FWD>(pick_tile_h(a.M, a.DN, MODE_FWD, cdiv(a.SC, HBK) * a.ntaps), a, fl, by, s);
}

static void conv_dgrad_bf16_any(const ConvDesc& d, const bf16_t* dy, const bf16_t* w_io, bf16_t* dx, const bf16_t* mask, bool accumulate,
                                hipStream_t s, const void* unpool_rec, int UH, int UW) {
    check_desc_h(d);
    const bool unpool = unpool_rec != nullptr;
    SSD_REQUIRE(!unpool || d.stride == 1, "conv_dgrad_bf16: the fused un-pool is for stride-1 convolutions");
    GatherArgsH a{};
    a.src = dy; a.wgt = w_io; a.bias = nullptr; a.mask = mask; a.dst = dx;
    a.unpool_rec = static_cast<const unsigned short*>(unpool_rec); a.UH = UH; a.UW = UW;
    a.M = d.B * d.Hi * d.Wi; a.DH = d.Hi; a.DW = d.Wi; a.DN = d.Ci;
    a.SH = d.Ho; a.SW = d.Wo; a.SC = d.Co;
    a.ntaps = d.KH * d.KW; a.mul = 1; a.div = d.stride;
    a.relu = 0; a.accum = accumulate; a.out_f32 = 0;
    for (int kh = 0; kh < d.KH; ++kh)
        for (int kw = 0; kw < d.KW; ++kw) {
            a.tap_dh[kh * d.KW + kw] = d.pad_h - kh * d.dil;
            a.tap_dw[kh * d.KW + kw] = d.pad_w - kw * d.dil;
        }
    const double fl = conv_flops(d), by = 2.0 * (conv_elems(d) + (mask ? (double)d.B * d.Hi * d.Wi * d.Ci : 0.0));
    static const int parity = env_int("SSD_DGRAD_PARITY", 1);      // A/B switch (shared with the fp32 path)
    if (d.stride == 2 && parity) {       // conv8_2, conv9_2, vgg512 conv10_2: parity classes, one launch (see conv_igemm.hip)
        GatherArgsH c = a;
        c.div = 1; c.mul = 1; c.wtaps = a.ntaps; c.ODH = d.Hi; c.ODW = d.Wi;
        // a handful of tiles (conv9_2 at batch 32: four classes of 800 pixels): 64 x 64 tiles with the k loop split over four
        // wave groups, like the unstrided small layers (pick_tile_h)
        static const int small_k = env_int("SSD_SMALL_KSPLIT", 1) && env_int("SSD_SMALL_TILE", 1) && env_int("SSD_TILE_BF16", -1) < 0;
        long long tiles64 = 0;
        for (int ph = 0; ph < 2; ++ph)
            for (int pw = 0; pw < 2; ++pw)
                tiles64 += (long long)cdiv((long long)d.B * ((d.Hi - ph + 1) / 2) * ((d.Wi - pw + 1) / 2), 64) * cdiv(a.DN, 64);
        const bool small = small_k && tiles64 <= 256;
        const int bm = small ? 64 : 128;
        const int NT = cdiv(a.DN, bm);
        c.nclass = 0;
        c.cls_wg0[0] = 0;
        for (int ph = 0; ph < 2; ++ph)
            for (int pw = 0; pw < 2; ++pw) {
                const int k = c.nclass;
                c.cls_DH[k] = (d.Hi - ph + 1) / 2; c.cls_DW[k] = (d.Wi - pw + 1) / 2;
                if (c.cls_DH[k] <= 0 || c.cls_DW[k] <= 0) continue;
                c.cls_M[k] = d.B * c.cls_DH[k] * c.cls_DW[k];
                c.cls_ph[k] = ph; c.cls_pw[k] = pw;
                int nt = 0;
                for (int kh = 0; kh < d.KH; ++kh)
                    for (int kw = 0; kw < d.KW; ++kw) {
                        const int nh = ph + d.pad_h - kh * d.dil, nw = pw + d.pad_w - kw * d.dil;
                        if (nh % 2 != 0 || nw % 2 != 0) continue;
                        c.cls_dh[k][nt] = nh / 2; c.cls_dw[k][nt] = nw / 2; c.cls_w[k][nt] = kh * d.KW + kw;
                        ++nt;
                    }
                c.cls_ntaps[k] = nt;
                c.cls_wg0[k + 1] = c.cls_wg0[k] + cdiv(c.cls_M[k], bm) * NT;
                ++c.nclass;
            }
        c.M = c.cls_M[0]; c.DH = c.cls_DH[0]; c.DW = c.cls_DW[0]; c.ntaps = c.cls_ntaps[0];
        static const int small_k2 = env_int("SSD_SMALL_KSPLIT", 1) == 2;
        if (small && small_k2) launch_gather_h<MODE_DGRAD, 2, 2, 1, 1, false, 2, true, 2>(c, "conv_dgrad_bf16_parity_64x64_k2", fl, by, s, c.cls_wg0[c.nclass]);
        else if (small) launch_gather_h<MODE_DGRAD, 2, 2, 1, 1, false, 2, true, 4>(c, "conv_dgrad_bf16_parity_64x64_k4", fl, by, s, c.cls_wg0[c.nclass]);
        else launch_gather_h<MODE_DGRAD, 2, 2, 2, 2, false, 2, true>(c, "conv_dgrad_bf16_parity_128x128", fl, by, s, c.cls_wg0[c.nclass]);
        return;
    }
    if (d.stride > 1) {       // the all-taps strided kernel (other strides, SSD_DGRAD_PARITY=0)
        launch_gather_h<MODE_DGRAD, 2, 2, 2, 2, true, 2>(a, "conv_dgrad_bf16_strided_128x128", fl, by, s);
        return;
    }
    if (!unpool && gather_c64_applicable(d, false)) {      // (the persistent 64 -> 64 kernel has its own write-out: no un-pool there)
        launch_gather_c64<MODE_DGRAD>(a, "conv_dgrad_bf16_c64", fl, by, s);
        return;
    }
    // 64-channel data gradients (conv2_1: 128 -> 64) on the kernel-row gather, 128 x 64 tiles: the per-tap 256 x 64 tile stages
    // 40 KB per 8 MFMAs per wave (52 FLOP per staged byte -- more than a CU's L2 -> LDS path feeds), a kernel row 44 KB per 24
    // (71).  conv2_1 with the fused un-pool 0.180 -> 0.160 ms; 256 x 64 kernel-row tiles 0.185 (two workgroups per CU instead of
    // three).  profiles/r06_y_*.  SSD_DGRAD_ROWS_C64: 0 per-tap, 1 (default) 128 x 64, 2 256 x 64.
    static const int rows_c64 = env_int("SSD_DGRAD_ROWS_C64", 1);
    if (gather_rows_applicable(d, true) && d.Ci == 64 && rows_c64) {
        if (rows_c64 == 2 && d.dil == 1) launch_gather_rows<MODE_DGRAD, 4, 1>(a, d.dil, unpool ? "conv_dgrad_unpool_bf16_rows_256x64" : "conv_dgrad_bf16_rows_256x64", fl, by, s);
        else launch_gather_rows<MODE_DGRAD, 2, 1>(a, d.dil, unpool ? "conv_dgrad_unpool_bf16_rows_128x64" : "conv_dgrad_bf16_rows_128x64", fl, by, s);
        return;
    }
    if (gather_rows_applicable(d, true) && d.Ci >= 128) {
        if (gather_rows256(d, a.M, d.Ci)) launch_gather_rows<MODE_DGRAD, 4>(a, d.dil, unpool ? "conv_dgrad_unpool_bf16_rows_256x128" : "conv_dgrad_bf16_rows_256x128", fl, by, s);
        else if (gather_rows_n64(a.M, d.Ci)) launch_gather_rows<MODE_DGRAD, 2, 1>(a, d.dil, unpool ? "conv_dgrad_unpool_bf16_rows_128x64" : "conv_dgrad_bf16_rows_128x64", fl, by, s);
        else launch_gather_rows<MODE_DGRAD, 2>(a, d.dil, unpool ? "conv_dgrad_unpool_bf16_rows_128x128" : "conv_dgrad_bf16_rows_128x128", fl, by, s);
        return;
    }
    launch_gather_cfg<MODE_DGRAD>(pick_tile_h(a.M, a.DN, MODE_DGRAD, cdiv(a.SC, HBK) * a.ntaps), a, fl, by, s, unpool);
}

void conv_dgrad_bf16(const ConvDesc& d, const bf16_t* dy, const bf16_t* w_io, bf16_t* dx, const bf16_t* mask, bool accumulate,
                     hipStream_t s) {
    conv_dgrad_bf16_any(d, dy, w_io, dx, mask, accumulate, s, nullptr, 0, 0);
}

// ---- conv1_2's data gradient with conv1_1's weight gradient inside (conv_gather_bf16_c64_kernel<MODE_DGRAD, true>) ----------
bool conv_dgrad_first_wgrad_bf16_applicable(const ConvDesc& d, const ConvDesc& d1) {
    return gather_c64_applicable(d, false) && d1.Ci * d1.KH * d1.KW == 27 && d1.Co == 64 && d1.KH == 3 && d1.KW == 3 && d1.stride == 1 &&
           d1.dil == 1 && d1.pad_h == 1 && d1.pad_w == 1 && d1.Hi == d.Hi && d1.Wi == d.Wi && d1.Ho == d.Hi && d1.Wo == d.Wi && d1.B == d.B;
}
size_t conv_dgrad_first_wgrad_bf16_ws_floats(const ConvDesc& d1) { return (size_t)256 * (27 * 64 + 64); }
void conv_dgrad_first_wgrad_bf16(const ConvDesc& d, const bf16_t* dy, const bf16_t* w_io, const bf16_t* mask, const ConvDesc& d1,
                                 const float* image, float* dw1, float* dbias1, const float* w1, float weight_decay, float* ws,
                                 hipStream_t s) {
    check_desc_h(d);
    SSD_REQUIRE(conv_dgrad_first_wgrad_bf16_applicable(d, d1), "conv_dgrad_first_wgrad_bf16: not a 64 -> 64 layer on top of a 3-channel 3x3 first layer");
    GatherArgsH a{};
    a.src = dy; a.wgt = w_io; a.bias = nullptr; a.mask = mask; a.dst = nullptr;
    a.M = d.B * d.Hi * d.Wi; a.DH = d.Hi; a.DW = d.Wi; a.DN = d.Ci;
    a.SH = d.Ho; a.SW = d.Wo; a.SC = d.Co;
    a.ntaps = 9; a.mul = 1; a.div = 1;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
            a.tap_dh[kh * 3 + kw] = d.pad_h - kh;
            a.tap_dw[kh * 3 + kw] = d.pad_w - kw;
        }
    FirstWArgs fw{};
    fw.img = image; fw.ws = ws; fw.Ci = d1.Ci; fw.ntaps = 9;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
            fw.tap_dh[kh * 3 + kw] = kh - d1.pad_h;
            fw.tap_dw[kh * 3 + kw] = kw - d1.pad_w;
        }
    constexpr size_t lds = (size_t)9 * 64 * 128 + 2 * 256 * 128 + 256 * 64 + 3 * 255 * 3 * 2 + 26;      // + the im2col tile + three image runs (bf16)
    static_assert(lds <= 160 * 1024, "LDS");
    const int ntiles = cdiv(a.M, 253);
    const int grid = ntiles < 256 ? ntiles : 256;
    {
        ProfScope prof("conv_dgrad_bf16_c64_first_wgrad", conv_flops(d) + conv_flops(d1),
                       2.0 * ((double)d.B * d.Ho * d.Wo * d.Co + (double)d.B * d.Hi * d.Wi * d.Ci) + 4.0 * d1.B * d1.Hi * d1.Wi * d1.Ci, s);
        auto kern = conv_gather_bf16_c64_kernel<MODE_DGRAD, true>;
        static bool once = (set_lds(kern, lds), true);
        (void)once;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a, ntiles, ntiles >= 2048 ? 1 : 0, fw);
        HIP_OK(hipGetLastError());
    }
    wgrad_reduce(ws, grid, (size_t)27 * 64, 64, dw1, dbias1, w1, weight_decay, s);
}

bool conv_dgrad_unpool_bf16_supported(const ConvDesc& d) { return d.stride == 1 && d.Ci % 8 == 0 && d.Co % 8 == 0; }
void conv_dgrad_unpool_bf16(const ConvDesc& d, const bf16_t* dy, const bf16_t* w_io, bf16_t* dx_unpooled, const void* rec, int UH, int UW,
                            hipStream_t s) {
    SSD_REQUIRE(conv_dgrad_unpool_bf16_supported(d), "conv_dgrad_unpool_bf16: unsupported shape");
    SSD_REQUIRE(rec != nullptr && (UH + 1) / 2 == d.Hi && (UW + 1) / 2 == d.Wi, "conv_dgrad_unpool_bf16: record / pooled size mismatch");
    conv_dgrad_bf16_any(d, dy, w_io, dx_unpooled, nullptr, false, s, rec, UH, UW);
}

// ---- forward with the fused 2x2 pool (conv_fwd_pool_bf16_*_kernel) ------------------------------------------------------
struct PoolTiling { int cw, rows, tw, nseg, nband; double util; };
static PoolTiling pool_tiling(const ConvDesc& d, int cw) {      // cw = 64: [4][64] tiles (64 -> 64 kernel), 32: [8][32] (kernel-row gather)
    PoolTiling t{};
    t.cw = cw; t.rows = 256 / cw;
    const int twmax = cw - 2;
    t.nseg = cdiv(d.Wo, twmax);
    t.tw = (cdiv(d.Wo, t.nseg) + 1) & ~1;                        // even: a window never straddles two segments
    t.nband = cdiv(d.Ho, t.rows);
    t.util = (double)d.Ho * d.Wo / ((double)t.nseg * t.nband * 256.0);
    return t;
}
static bool pool_fwd_shape_bf16(const ConvDesc& d) {
    const bool shape = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.dil == 1 && d.pad_h == 1 && d.pad_w == 1 && d.Hi == d.Ho && d.Wi == d.Wo;
    return shape && d.Ci % 64 == 0 && d.Co % 8 == 0 && ((d.Ci == 64 && d.Co == 64) || d.Co >= 128);
}
// the executor's rule: fuse where at least 88 % of the 2-D tile rows carry pixels (conv1_2 / conv2_2 of both presets; conv3_3's
// 75- / 128-wide rows fill [8][32] tiles to 73 / 80 %: a matrix-bound layer loses more there than the pool costs)
bool conv_fwd_pool_bf16_supported(const ConvDesc& d) {
    return pool_fwd_shape_bf16(d) && pool_tiling(d, (d.Ci == 64 && d.Co == 64) ? 64 : 32).util >= 0.88;
}
void conv_fwd_pool_bf16(const ConvDesc& d, const bf16_t* x, const bf16_t* w_oi, const float* bias, bf16_t* y_pool, void* rec, hipStream_t s) {
    check_desc_h(d);
    SSD_REQUIRE(pool_fwd_shape_bf16(d), "conv_fwd_pool_bf16: 3x3 stride-1 SAME, Ci a multiple of 64, 64 -> 64 or Co >= 128");
    SSD_REQUIRE(y_pool != nullptr, "conv_fwd_pool_bf16: null output");
    const bool c64 = d.Ci == 64 && d.Co == 64;
    const PoolTiling t = pool_tiling(d, c64 ? 64 : 32);
    GatherArgsH a{};
    a.src = x; a.wgt = w_oi; a.bias = bias; a.dst = nullptr;
    a.M = d.B * d.Ho * d.Wo; a.DH = d.Ho; a.DW = d.Wo; a.DN = d.Co;
    a.SH = d.Hi; a.SW = d.Wi; a.SC = d.Ci;
    a.ntaps = 9; a.mul = 1; a.div = 1; a.relu = 1;
    a.pool_dst = y_pool; a.pool_rec = static_cast<unsigned short*>(rec);
    a.PB = d.B; a.PH = (d.Ho + 1) / 2; a.PW = (d.Wo + 1) / 2;
    a.TW = t.tw; a.NSEG = t.nseg; a.NBAND = t.nband;
    const int ntiles = d.B * t.nseg * t.nband;
    const double fl = conv_flops(d), by = 2.0 * ((double)d.B * d.Hi * d.Wi * d.Ci + (double)d.B * a.PH * a.PW * d.Co * 1.25 + 9.0 * d.Ci * d.Co);
    if (c64) {
        constexpr size_t lds = (size_t)9 * 64 * 128 + 2 * 256 * 128 + 512;      // (+ the two rows a dead GEMM row's last tap reads past the tile)
        auto kern = conv_fwd_pool_bf16_c64_kernel;
        static bool once = (set_lds(kern, lds), true);
        (void)once;
        ProfScope prof("conv_fwd_pool_bf16_c64", fl, by, s);
        hipLaunchKernelGGL(kern, dim3(ntiles < 256 ? ntiles : 256), dim3(512), lds, s, a, ntiles, ntiles >= 2048 ? 1 : 0);
        HIP_OK(hipGetLastError());
        return;
    }
    constexpr size_t lds = 256 * 128 + 3 * 128 * 128;      // one unit; the fp32 C tile of the epilogue ([128][132]) fits inside
    auto kern = conv_fwd_pool_bf16_rows_kernel;
    static bool once = (set_lds(kern, lds), true);
    (void)once;
    a.NT = cdiv(d.Co, 128);
    ProfScope prof("conv_fwd_pool_bf16_rows_256x128", fl, by, s);
    hipLaunchKernelGGL(kern, dim3(ntiles * a.NT), dim3(256), lds, s, a);
    HIP_OK(hipGetLastError());
}

// ---- wgrad planning -------------------------------------------------------------------
struct WgradPlanH {
    int cfg;        // (channels x n, stages) 0: 128x128x2, 1: 64x64x2, 2: 64x128x2, 3: 128x64x2, 4: 128x128x4 (1/CU), 5: 256x128x3 (1/CU)
    int bkt, bnt, CT, NT, tiles, nsplit, mchunk;
};

static WgradPlanH plan_wgrad_h(const ConvDesc& d) {
    WgradPlanH p{};
    const int M = d.B * d.Ho * d.Wo;
    const int waste128 = cdiv(d.Co, 128) * 128 - d.Co, waste64 = cdiv(d.Co, 64) * 64 - d.Co;
    if (d.Ci <= 64 && d.Co <= 64) p.cfg = 1;
    else if (d.Ci <= 64) p.cfg = 2;
    else if (waste64 < waste128) p.cfg = 3;      // fused heads: Co = 104 / 152
    else if (M < 20000) p.cfg = 2;               // 19x19 maps and smaller: more, smaller tiles (conv5_2 496 -> 544, mod_conv7 347 -> 411 TF/s)
    else p.cfg = 0;
    static const int forced = env_int("SSD_WGRAD_CFG_BF16", -1);      // tuning override
    if (forced >= 0 && forced < 6 && !(forced == 5 && d.Ci < 256)) p.cfg = forced;
    p.bkt = p.cfg == 5 ? 256 : (p.cfg == 0 || p.cfg == 3 || p.cfg == 4) ? 128 : 64;
    p.bnt = (p.cfg == 1 || p.cfg == 3) ? 64 : 128;
    p.CT = cdiv(d.Ci, p.bkt);
    p.NT = cdiv(d.Co, p.bnt);
    p.tiles = d.KH * d.KW * p.CT * p.NT;
    constexpr int target_wgs = 1536;
    int want = cdiv(target_wgs, p.tiles);
    if (want > 256) want = 256;          // the reduce pass reads every slab: keep it short
    int maxs = cdiv(M, 512);
    p.nsplit = want < 1 ? 1 : (want > maxs ? maxs : want);
    if (p.nsplit < 1) p.nsplit = 1;
    p.mchunk = cdiv(cdiv(M, p.nsplit), 64) * 64;
    p.nsplit = cdiv(M, p.mchunk);
    return p;
}

template <int WM, int WN, int TM, int TN, int NS = 2>
static void launch_wgrad_h(WgradArgsH& a, const WgradPlanH& pl, const char* label, double flops, double bytes, hipStream_t s) {
    constexpr int BKT = 32 * TM * WM, BNT = 32 * TN * WN;
    constexpr size_t lds = NS * (size_t)64 * (BKT + BNT) * 2;
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = conv_wgrad_bf16_kernel<WM, WN, TM, TN, NS>;
    static bool once = (set_lds(kern, lds), true);
    (void)once;
    ProfScope prof(label, flops, bytes, s);
    hipLaunchKernelGGL(kern, dim3(pl.nsplit * pl.tiles), dim3(256), lds, s, a);
    HIP_OK(hipGetLastError());
}

// ---- kernel-row variant (3x3, stride 1, SAME) --------------------------------------------------------------
// 0: off   1: the 64-input-channel layers (conv1_2, conv2_1)   2: every applicable layer
static int rows_mode() {
    static const int v = env_int("SSD_WGRAD_ROWS_BF16", 1);      // tuning / A-B switch
    return v;
}
static bool rows_applicable(const ConvDesc& d) {
    const bool shape = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.dil == 1 && d.pad_h == 1 && d.pad_w == 1 && d.Hi == d.Ho && d.Wi == d.Wo;
    if (!shape || rows_mode() == 0) return false;
    return d.Ci <= 64 || rows_mode() >= 2;
}
struct RowsPlan {
    int tm, tn, CT, NT, schunk, nsplit;
    long long nslots;
};
static RowsPlan plan_rows(const ConvDesc& d) {
    RowsPlan p{};
    p.tm = d.Ci > 64 ? 2 : 1;
    p.tn = (p.tm == 1 && d.Co > 64) ? 2 : 1;
    p.CT = cdiv(d.Ci, 64 * p.tm);
    p.NT = cdiv(d.Co, 64 * p.tn);
    p.nslots = (long long)d.B * d.Ho * (d.Wo + 2);
    // tuning override.  Re-measured on the round-2 kernels incl. the slab reduce (gpurun r02_m): conv1_2 0.515 / 0.424 / 0.374 /
    // 0.383 / 0.399 / 0.381 ms and conv2_1 0.207 / 0.203 / 0.202 / 0.217 / 0.240 / 0.247 ms at 512 / 768 / 1024 / 1536 / 2048 /
    // 3072 workgroups (beyond ~1024 the slabs' write + reduce traffic outgrows the layer's own); step 1024 vs 2048: +0.4 %
    constexpr int target = 1024;
    int want = cdiv(target, 3 * p.CT * p.NT);
    if (want > 1024) want = 1024;
    const int maxs = cdiv(p.nslots, 64 * 16);
    p.nsplit = want > maxs ? maxs : want;
    if (p.nsplit < 1) p.nsplit = 1;
    p.schunk = cdiv(cdiv(p.nslots, p.nsplit), 64) * 64;
    p.nsplit = cdiv(p.nslots, p.schunk);
    return p;
}

// 8-wave variant: SSD_WGRAD_ROWS8_BF16 = 0 off, else the ring depth (2..4) for the layers with >= 128 input channels
static int rows8_mode() {
    static const int v = env_int("SSD_WGRAD_ROWS8_BF16", 4);     // tuning / A-B switch (staggered issue needs >= 3 stages; 4 measured best)
    return v;
}
static bool rows8_applicable(const ConvDesc& d) {
    const bool shape = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.dil == 1 && d.pad_h == 1 && d.pad_w == 1 && d.Hi == d.Ho && d.Wi == d.Wo;
    // (round 5 tried the fused heads -- Co = 104 / 152 -- on the per-tap kernel, because a rows8 launch takes every CU's LDS for
    // 60-80 us while backward's latency-bound chain runs beside it: the step did not move, profiles/r05_m_ab_tail_bf16.txt)
    return shape && rows8_mode() >= 2 && d.Ci >= 128 && rows_mode() != 2;
}
// one workgroup per CU and a single round: the fewest pixel splits (= the least fp32 slab traffic) that fill the chip
static RowsPlan plan_rows8(const ConvDesc& d) {
    RowsPlan p{};
    p.tm = 2; p.tn = 2;
    p.CT = cdiv(d.Ci, 128);
    p.NT = cdiv(d.Co, 128);
    p.nslots = (long long)d.B * d.Ho * (d.Wo + 2);
    const int jobs = 3 * p.CT * p.NT;
    // SSD_WGRAD_ROWS8_WGS: workgroup target (A/B switch).  256 = one per CU.  Fewer = fewer pixel splits = proportionally less
    // fp32 slab traffic (every workgroup leaves 196 KB), at the price of CUs the launch does not use -- which, in the step, the
    // data gradient on the other stream does.
    constexpr int wgs_target = 256;      // (192 ties, 128 loses 9 %: profiles/r04_t_ab_rows8_wgs_bf16.txt)
    int want = jobs >= wgs_target ? 1 : wgs_target / jobs;
    const int maxs = (int)std::max<long long>(1, p.nslots / (64 * 8));
    p.nsplit = want > maxs ? maxs : want;
    p.schunk = cdiv(cdiv(p.nslots, p.nsplit), 64) * 64;
    p.nsplit = cdiv(p.nslots, p.schunk);
    return p;
}
template <int NS, bool STAGGER = true>
static void launch_wgrad_rows8(WgradRowsArgs& a, const char* label, double flops, double bytes, hipStream_t s) {
    constexpr size_t lds = (size_t)NS * (72 * 256 + 64 * 256);
    auto kern = conv_wgrad_bf16_rows8_kernel<NS, STAGGER>;
    static bool once = (set_lds(kern, lds), true);
    (void)once;
    ProfScope prof(label, flops, bytes, s);
    hipLaunchKernelGGL(kern, dim3(a.nsplit * 3 * a.CT * a.NT), dim3(512), lds, s, a);
    HIP_OK(hipGetLastError());
}

template <int TM, int TN, int NS>
static void launch_wgrad_rows_ns(WgradRowsArgs& a, const char* label, double flops, double bytes, hipStream_t s) {
    constexpr size_t lds = NS * (size_t)(72 * 128 * TM + 64 * 128 * TN);
    auto kern = conv_wgrad_bf16_rows_kernel<TM, TN, NS>;
    static bool once = (set_lds(kern, lds), true);
    (void)once;
    ProfScope prof(label, flops, bytes, s);
    hipLaunchKernelGGL(kern, dim3(a.nsplit * 3 * a.CT * a.NT), dim3(256), lds, s, a);
    HIP_OK(hipGetLastError());
}
template <int TM, int TN>
static void launch_wgrad_rows(WgradRowsArgs& a, const char* label, double flops, double bytes, hipStream_t s) {
    launch_wgrad_rows_ns<TM, TN, 2>(a, label, flops, bytes, s);      // (rings of 3 / 4 stages measured slower: see the kernel)
}

// ---- column-walk variant (3x3, stride 1, SAME, <= 64 input channels): SSD_WGRAD_COL_BF16 = 0 off, 1 on (default), 2 also small
// layers (tests).  Measured (profiles/r03_r_col_wgrad_sweep_bf16.txt, r03_s_col_wgrad_ablation_bf16.txt, r03_t_ab_col_wgrad_bf16.txt;
// kernel + slab reduce, batch 32, post-relu operands): conv1_2 0.386 -> 0.252 ms (551 -> 844 TFLOP/s), conv2_1 0.206 -> 0.175;
// 512 workgroups (one round of two per CU) beat 768 / 1024; step 7.53 / 7.55 -> 7.41 / 7.38 ms.  Ablations of conv1_2: multiply
// only 0.212, staging only 0.166, neither 0.049 (launch, first tiles, slabs, reduce): the multiply phase alone runs at
// 1.3 PFLOP/s, the power-limited rate of the big layers -- what is left is the part of the staging the multiply does not hide.
static int col_mode() {
    static const int v = env_int("SSD_WGRAD_COL_BF16", 1);
    return v;
}
static bool col_applicable(const ConvDesc& d) {
    const bool shape = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.dil == 1 && d.pad_h == 1 && d.pad_w == 1 && d.Hi == d.Ho && d.Wi == d.Wo;
    if (!shape || col_mode() == 0 || d.Ci > 64) return false;
    return col_mode() >= 2 || (long long)d.B * d.Ho * d.Wo >= 64LL * 1024;      // enough pixels for a round of units
}
struct ColPlan { int NT, nstrips, rpu, RC, nunits; };
static ColPlan plan_col(const ConvDesc& d) {
    ColPlan p{};
    p.NT = cdiv(d.Co, 64);
    p.nstrips = cdiv(d.Wo, 64);
    // two workgroups per CU (nine accumulator tiles per wave: ~220 registers) = 512 slots; a unit of fewer than 8 rows would
    // spend more than a quarter of its loads on the two halo rows
    constexpr int target = 512;
    const int per_row_chunk = d.B * p.nstrips * p.NT;
    int rc = target / (per_row_chunk > 0 ? per_row_chunk : 1);
    if (rc < 1) rc = 1;
    const int max_rc = cdiv(d.Ho, 8);
    if (rc > max_rc) rc = max_rc;
    p.rpu = cdiv(d.Ho, rc);
    p.RC = cdiv(d.Ho, p.rpu);
    p.nunits = d.B * p.nstrips * p.RC;
    return p;
}

size_t conv_wgrad_bf16_ws_floats(const ConvDesc& d) {
    const size_t per = (size_t)d.KH * d.KW * d.Ci * d.Co + d.Co;
    size_t n = (size_t)plan_wgrad_h(d).nsplit * per;
    if (col_applicable(d)) n = std::max(n, (size_t)plan_col(d).nunits * per);
    if (rows_applicable(d)) n = std::max(n, (size_t)plan_rows(d).nsplit * per);
    if (rows8_applicable(d)) n = std::max(n, (size_t)plan_rows8(d).nsplit * per);
    return n;
}

void conv_wgrad_bf16(const ConvDesc& d, const bf16_t* x, const bf16_t* dy, float* dw, float* dbias, const float* w,
                     float weight_decay, float* ws, hipStream_t s) {
    check_desc_h(d);
    if (col_applicable(d)) {
        const ColPlan cp = plan_col(d);
        WgradColArgs c{};
        c.x = x; c.dy = dy; c.ws = ws;
        c.B = d.B; c.H = d.Ho; c.W = d.Wo; c.Ci = d.Ci; c.Co = d.Co;
        c.NT = cp.NT; c.nstrips = cp.nstrips; c.rpu = cp.rpu; c.RC = cp.RC;
        constexpr size_t lds = 5 * 72 * 128 + 3 * 64 * 128;
        static bool once = (set_lds(conv_wgrad_bf16_col_kernel, lds), true);
        (void)once;
        {
            ProfScope prof("conv_wgrad_bf16_col_64x64", conv_flops(d), 2.0 * conv_elems(d), s);
            hipLaunchKernelGGL(conv_wgrad_bf16_col_kernel, dim3(cp.nunits * cp.NT), dim3(256), lds, s, c);
            HIP_OK(hipGetLastError());
        }
        wgrad_reduce(ws, cp.nunits, (size_t)9 * d.Ci * d.Co, d.Co, dw, dbias, w, weight_decay, s);
        return;
    }
    if (rows8_applicable(d)) {
        const RowsPlan rp = plan_rows8(d);
        WgradRowsArgs r{};
        r.x = x; r.dy = dy; r.ws = ws;
        r.B = d.B; r.H = d.Ho; r.W = d.Wo; r.Ci = d.Ci; r.Co = d.Co; r.pad_h = d.pad_h;
        r.CT = rp.CT; r.NT = rp.NT; r.nslots = rp.nslots; r.schunk = rp.schunk; r.nsplit = rp.nsplit;
        const double fl = conv_flops(d), by = 2.0 * conv_elems(d);
        const char* label = "conv_wgrad_bf16_rows8_128x128";
        if (rows8_mode() == 2) launch_wgrad_rows8<2>(r, label, fl, by, s);
        else if (rows8_mode() == 4) launch_wgrad_rows8<4, true>(r, label, fl, by, s);
        else launch_wgrad_rows8<3, true>(r, label, fl, by, s);
        wgrad_reduce(ws, rp.nsplit, (size_t)9 * d.Ci * d.Co, d.Co, dw, dbias, w, weight_decay, s);
        return;
    }
    if (rows_applicable(d)) {
        const RowsPlan rp = plan_rows(d);
        WgradRowsArgs r{};
        r.x = x; r.dy = dy; r.ws = ws;
        r.B = d.B; r.H = d.Ho; r.W = d.Wo; r.Ci = d.Ci; r.Co = d.Co; r.pad_h = d.pad_h;
        r.CT = rp.CT; r.NT = rp.NT; r.nslots = rp.nslots; r.schunk = rp.schunk; r.nsplit = rp.nsplit;
        const double fl = conv_flops(d), by = 2.0 * conv_elems(d);
        if (rp.tm == 2) launch_wgrad_rows<2, 1>(r, "conv_wgrad_bf16_rows_128x64", fl, by, s);
        else if (rp.tn == 2) launch_wgrad_rows<1, 2>(r, "conv_wgrad_bf16_rows_64x128", fl, by, s);
        else launch_wgrad_rows<1, 1>(r, "conv_wgrad_bf16_rows_64x64", fl, by, s);
        wgrad_reduce(ws, rp.nsplit, (size_t)9 * d.Ci * d.Co, d.Co, dw, dbias, w, weight_decay, s);
        return;
    }
    WgradPlanH pl = plan_wgrad_h(d);
    WgradArgsH a{};
    a.x = x; a.dy = dy; a.ws = ws;
    a.M = d.B * d.Ho * d.Wo; a.Hi = d.Hi; a.Wi = d.Wi; a.Ci = d.Ci; a.Ho = d.Ho; a.Wo = d.Wo; a.Co = d.Co;
    a.ntaps = d.KH * d.KW; a.stride = d.stride; a.CT = pl.CT; a.NT = pl.NT;
    a.mchunk = pl.mchunk; a.nsplit = pl.nsplit;
    for (int kh = 0; kh < d.KH; ++kh)
        for (int kw = 0; kw < d.KW; ++kw) {
            a.tap_dh[kh * d.KW + kw] = kh * d.dil - d.pad_h;
            a.tap_dw[kh * d.KW + kw] = kw * d.dil - d.pad_w;
        }
    const double fl = conv_flops(d), by = 2.0 * conv_elems(d);
    if (pl.cfg == 1) launch_wgrad_h<2, 2, 1, 1, 2>(a, pl, "conv_wgrad_bf16_64x64", fl, by, s);
    else if (pl.cfg == 2) launch_wgrad_h<2, 2, 1, 2, 2>(a, pl, "conv_wgrad_bf16_64x128", fl, by, s);
    else if (pl.cfg == 3) launch_wgrad_h<2, 2, 2, 1, 2>(a, pl, "conv_wgrad_bf16_128x64", fl, by, s);
    else if (pl.cfg == 4) launch_wgrad_h<2, 2, 2, 2, 4>(a, pl, "conv_wgrad_bf16_128x128x4", fl, by, s);
    else if (pl.cfg == 5) launch_wgrad_h<2, 2, 4, 2, 3>(a, pl, "conv_wgrad_bf16_256x128x3", fl, by, s);
    else launch_wgrad_h<2, 2, 2, 2, 2>(a, pl, "conv_wgrad_bf16_128x128", fl, by, s);
    wgrad_reduce(ws, pl.nsplit, (size_t)a.ntaps * d.Ci * d.Co, d.Co, dw, dbias, w, weight_decay, s);
}

// ---- the tail's weight gradients as one launch (conv_wgrad_group_bf16_kernel) --------------------------------------------
int conv_wgrad_group_bf16_max() { return WGRAD_GROUP_MAX; }
void conv_wgrad_group_bf16(const WgradGroupItem* items, int n, float weight_decay, hipStream_t s) {
    SSD_REQUIRE(n >= 1 && n <= WGRAD_GROUP_MAX, "grouped weight gradient: 1..%d layers (got %d)", WGRAD_GROUP_MAX, n);
    WgradGroupArgs g{};
    g.n = n;
    double fl = 0.0, by = 0.0;
    for (int i = 0; i < n; ++i) {
        const ConvDesc& d = items[i].d;
        check_desc_h(d);
        WgradArgsH& a = g.L[i];
        a.x = items[i].x; a.dy = items[i].dy; a.ws = nullptr;
        a.M = d.B * d.Ho * d.Wo; a.Hi = d.Hi; a.Wi = d.Wi; a.Ci = d.Ci; a.Ho = d.Ho; a.Wo = d.Wo; a.Co = d.Co;
        a.ntaps = d.KH * d.KW; a.stride = d.stride; a.CT = cdiv(d.Ci, 64); a.NT = cdiv(d.Co, 128);
        a.mchunk = cdiv(a.M, 64) * 64; a.nsplit = 1;
        for (int kh = 0; kh < d.KH; ++kh)
            for (int kw = 0; kw < d.KW; ++kw) {
                a.tap_dh[kh * d.KW + kw] = kh * d.dil - d.pad_h;
                a.tap_dw[kh * d.KW + kw] = kw * d.dil - d.pad_w;
            }
        a.dw = items[i].dw; a.db = items[i].dbias; a.w = items[i].w; a.wd = weight_decay;
        SSD_REQUIRE(a.dw != nullptr && a.w != nullptr, "grouped weight gradient: null gradient / filter pointer");
        g.wg0[i + 1] = g.wg0[i] + a.ntaps * a.CT * a.NT;
        fl += conv_flops(d);
        by += 2.0 * conv_elems(d);
    }
    constexpr size_t lds = 2 * (size_t)64 * (64 + 128) * 2;
    static bool once = (set_lds(conv_wgrad_group_bf16_kernel, lds), true);
    (void)once;
    ProfScope prof("conv_wgrad_group_bf16_64x128", fl, by, s);
    SSD_LAUNCH_STOP(conv_wgrad_group_bf16_kernel, dim3(g.wg0[n]), dim3(256), lds, s, g);
    HIP_OK(hipGetLastError());
}

}  // namespace ssd
