// The latency-bound tail of the SSD graph as ONE launch per direction (bf16 configuration).
//
// Below the 10x10 map (vgg300: conv9_1 ... conv11_2 and the heads of maps 3-5; vgg512: conv10_1 ... conv12_2 and the heads of
// maps 4-6; ssdvgg.py:300-332, 353-365) a layer is a GEMM of 32 ... 3200 rows for the WHOLE batch: as launches of their own
// they are a chain of 9 (forward) / 8 (data gradients) dependent kernels of 4 ... 100 workgroups, 8-20 us each whatever the
// tile, with an event hand-off between every two of them (profiles/r05_zz_timeline_bf16.txt).  Nothing in those layers couples
// two images, so ONE workgroup walks the whole chain for its image with nothing but workgroup barriers between the layers:
//
//   for every stage (= one convolution, forward or data gradient, of conv.h's gather-GEMM form):
//       dst[m][n] = sum_{tap, c} src[pix(m, tap)][c] * W[tap][n][c]          m = the image's pixels (oh, ow)
//
// What such a workgroup is short of is neither FLOPs nor bytes but REQUESTS IN FLIGHT: a stage streams its filter once (590 KB
// for a 3x3 128 -> 256 layer), every element is used for one or two MFMAs and never again, and one CU pulls ~60 B/clk at best.
// The first version of this kernel (4 waves, [256][64 k] filter tiles through an LDS ring, one barrier per k step) spent 1.3 us per
// k step on DMA issue + barrier and lost to the launches it replaced (258 us against ~110, profiles/r06_b_*).  This one:
//
//   * 16 waves.  A wave owns 16 output channels (one n block) and up to 4 blocks of 16 pixels: v_mfma_f32_16x16x32_bf16, filter
//     operand first, so a lane ends up with 4 consecutive channels of one pixel.  Layers with fewer than 16 n blocks split k over
//     wave groups (partials added in group order through LDS: a fixed summation order).
//   * The FILTER never touches LDS: a lane's operand of a k step is 16 contiguous bytes of its channel's row, loaded straight into
//     registers (a wave-load covers 16 rows x 64 bytes) into a ring of PF register sets -- PF steps in flight per wave, 16 waves:
//     128 KB of requests in flight per CU, no barrier inside a stage.
//   * The ACTIVATIONS never leave LDS between stages: a stage's input feature map (at most 100 pixels) sits in LDS with a padded
//     pixel pitch (conflict-free ds_read_b128), its epilogue writes the output to global memory (backward and the tests read it)
//     AND into the LDS region the next stage gathers from; a tap outside the image reads a zero row.  The host plans the regions
//     by liveness (tail_chain_bf16).
//
// Same arithmetic as the per-layer kernels: bf16 operands, fp32 accumulation, bias + relu (forward) or accumulate + relu mask
// (data gradient) in fp32, ONE rounding to bf16 (heads: fp32 out).  The summation order differs from theirs (k in ascending
// (tap, channel) order per group), so results agree to fp32 rounding, not bit for bit; tests/test_gpu_tail.py checks every stage
// shape of both presets against the oracle.
#include "conv.h"
#include "conv_detail.h"
#include "bf16.h"
#include <algorithm>
#include <vector>

namespace ssd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TAIL_MAX_STAGES = 16;
constexpr int TAIL_WAVES = 16, TAIL_THREADS = 64 * TAIL_WAVES;
constexpr int TAIL_MG = 4;                   // blocks of 16 pixels per task
constexpr int TAIL_PF = 6;                   // filter k steps in flight per wave (8 spills registers: 16 waves share the file, 128 each)
constexpr int TAIL_ZERO_BYTES = 2560;        // the zero row: covers the widest pixel row (1024 channels) plus one chunk
constexpr int TAIL_LDS_MAX = 160 * 1024;
enum { TF_RELU = 1, TF_ACCUM = 2, TF_OUT_F32 = 4, TF_LOAD_IN = 8, TF_LOAD_OUT = 16 };

struct TailStageK {
    const bf16_t* src;      // gathered tensor of THIS launch's first image
    const bf16_t* wgt;      // PACKED filter: fragments [n block][k step][64 lanes][8 bf16] (tail_pack_kernel)
    const float* bias;      // forward (or nullptr)
    const bf16_t* mask;     // data gradient: relu mask, dst's shape (or nullptr)
    void* dst;              // bf16 [.][DH][DW][DN], or fp32 when TF_OUT_F32
    unsigned src_img, dst_img;      // elements per image
    short DH, DW, DN, SH, SW, SC;
    short ntaps, mul, dshift, flags;        // source pixel of (oh, tap) = (oh * mul + dh) >> dshift when the low dshift bits are zero
    short nblk_n, ngrp_m, ksplit, nsteps;   // tasks = nblk_n * ngrp_m * ksplit; nsteps = ntaps * ceil(SC / 32) k steps in all
    int in_off, in_pitch;                   // LDS region of the input feature map (bytes; pitch = SC * 2 + 16)
    int out_off, out_pitch;                 // LDS region of the output (-1: not kept)
    int scratch_off;                        // k-split partials
    int tap_dh[9], tap_dw[9];               // (dwords: indexed by the wave-uniform tap with scalar loads)
};
struct TailArgsK {
    int nstages, nimg;
    unsigned long long* stamps;      // measurement aid (SSD_TAIL_STAMPS=1): 100-MHz clock of image 0's workgroup at each stage boundary
    TailStageK st[TAIL_MAX_STAGES];
};
static_assert(sizeof(TailArgsK) <= 4000, "kernel argument segment");

// The k loop of one task: NMB blocks of 16 pixels x one block of 16 output channels over the k steps [j0, j1) (step j = tap * C32 + c:
// 32 channels c * 32 .. of tap `tap`).  The filter operand of a step is ONE contiguous KB of the packed filter (tail_pack_kernel:
// fragment (n block, step) = [64 lanes][8 bf16], lane (l16, lq) -> row nb * 16 + l16, k = c * 32 + lq * 8 ..; zeros beyond DN / SC), so
// a wave-load is fully coalesced; TAIL_PF of them are in flight per wave.  The pixel operand comes out of the LDS-resident feature
// map (or the zero row where the tap leaves the image).
// Workgroup barrier that publishes LDS writes and nothing else.  __syncthreads() also drains vmcnt to 0: every stage boundary would
// then wait for the epilogue's global stores to be acknowledged AND for the next stage's filter prefetch -- ~2 us per boundary (the
// chain reads nothing from global memory that it wrote itself: feature maps travel through LDS).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
struct KLoopArgs {
    const unsigned char* smem;
    int j0, j1, C32, SC, in_off, in_pitch, SH, SW, DW, mul, dshift, M, mg;
    const int* tap_dh;
    const int* tap_dw;
};
// The filter stream of one task: this lane's 16 bytes of the fragments (nb, j0) .. (nb, j1 - 1), one KB apart
struct BStream {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned poff, pend;
};
__device__ __forceinline__ bf16x8 b_next(BStream& b) {
    const unsigned off = b.poff < b.pend ? b.poff : 0xFFFFFFF0u;      // (past the task's range: zeros, no memory traffic)
    b.poff += 1024u;
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b.rsrc, off, 0, 0));
}
__device__ __forceinline__ void b_prime(BStream& b, bf16x8 (&bq)[TAIL_PF]) {
#pragma unroll
    for (int u = 0; u < TAIL_PF; ++u) bq[u] = b_next(b);
}
// (the ring bq has been primed with the task's first TAIL_PF steps: by the caller, or -- for a wave's first task of a stage -- while
// the previous stage was still finishing: a filter does not depend on the stage before it)
template <int NMB>
__device__ __forceinline__ void tail_kloop(const KLoopArgs& a, BStream& bs, bf16x8 (&bq)[TAIL_PF], f32x4 (&acc)[TAIL_MG], const int l16, const int lq) {
    // this lane's pixel of block i: (oh * mul, ow * mul) packed as two 16-bit halves; 0x7FFF in the row half = no pixel
    unsigned rhw[NMB];
#pragma unroll
    for (int i = 0; i < NMB; ++i) {
        const int m = (a.mg * TAIL_MG + i) * 16 + l16;
        const int mm = m < a.M ? m : 0;
        const int oh = mm / a.DW, ow = mm - oh * a.DW;
        rhw[i] = ((unsigned)(m < a.M ? oh * a.mul : 0x7FFF) << 16) | (unsigned)(ow * a.mul);
    }
    const int lowbits = (1 << a.dshift) - 1;
    int tap = a.j0 / a.C32, c = a.j0 - tap * a.C32;
    int abase[NMB];      // LDS byte address of this lane's source pixel row under the current tap (+ its k group), or the zero row
    // (the offsets of the NEXT tap are requested when a tap starts: the scalar loads then have a tap's worth of steps to land)
    int ndh = a.tap_dh[tap], ndw = a.tap_dw[tap];
    auto set_tap = [&]() {
        const int dh = ndh, dw = ndw;
        const int nt = tap + 1 < 9 ? tap + 1 : 8;
        ndh = a.tap_dh[nt]; ndw = a.tap_dw[nt];
#pragma unroll
        for (int i = 0; i < NMB; ++i) {
            const int sh = (int)(rhw[i] >> 16) + dh, sw = (int)(rhw[i] & 0xFFFFu) + dw;
            const int qh = sh >> a.dshift, qw = sw >> a.dshift;      // arithmetic shifts: a negative stays negative
            const bool ok = ((sh | sw) & lowbits) == 0 && (unsigned)qh < (unsigned)a.SH && (unsigned)qw < (unsigned)a.SW;      // (no pixel: qh >= 0x7FFF - 8 >> dshift, out of range)
            abase[i] = ok ? a.in_off + (qh * a.SW + qw) * a.in_pitch + lq * 16 : lq * 16;
        }
    };
    set_tap();
    for (int jb = a.j0; jb < a.j1; jb += TAIL_PF) {
#pragma unroll
        for (int u = 0; u < TAIL_PF; ++u) {
            // (steps past j1 multiply a zero filter operand; channels past SC likewise -- the packed filter holds zeros there -- but
            // the pixel operand must still be finite: the zero row)
            const bool k_ok = c * 32 + lq * 8 < a.SC;
            const bf16x8 b = bq[u];
            bq[u] = b_next(bs);
            // (two pixel operands at a time: the register file is shared by 16 waves)
#pragma unroll
            for (int i = 0; i < NMB; i += 2) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(a.smem + (k_ok ? abase[i] + c * 64 : lq * 16));
                bf16x8 a1;
                if (i + 1 < NMB) a1 = *reinterpret_cast<const bf16x8*>(a.smem + (k_ok ? abase[i + 1 < NMB ? i + 1 : i] + c * 64 : lq * 16));
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a0, acc[i], 0, 0, 0);
                if (i + 1 < NMB) acc[i + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a1, acc[i + 1], 0, 0, 0);
            }
            ++c;
            if (c == a.C32) {      // wave-uniform
                c = 0;
                tap = tap + 1 < 9 ? tap + 1 : 8;      // (a step past the last tap only ever sees zero filter operands)
                set_tap();
            }
        }
    }
}

// filter mirrors [tap][DN][SC] -> fragments [n block][step][64 lanes][8], several layers per launch
constexpr int TAIL_PACK_MAX = 32;
struct TailPackK {
    int n;
    int frag0[TAIL_PACK_MAX + 1];
    struct L { const bf16_t* w; bf16_t* out; int DN, SC, C32, nsteps; } l[TAIL_PACK_MAX];
};
static_assert(sizeof(TailPackK) <= 4000, "kernel argument segment");
__global__ __launch_bounds__(256) void tail_pack_kernel(TailPackK t) {
    const int gfrag = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (gfrag >= t.frag0[t.n]) return;
    int li = 0;
    for (int k = 1; k < t.n; ++k)
        if (gfrag >= t.frag0[k]) li = k;
    const TailPackK::L& L = t.l[li];
    const int frag = gfrag - t.frag0[li];
    const int nb = frag / L.nsteps, j = frag - nb * L.nsteps;
    const int tap = j / L.C32, c = j - tap * L.C32;
    const int n = nb * 16 + (lane & 15), k = c * 32 + (lane >> 4) * 8;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (n < L.DN && k < L.SC) v = *reinterpret_cast<const u32x4*>(L.w + ((size_t)tap * L.DN + n) * L.SC + k);
    *reinterpret_cast<u32x4*>(L.out + (size_t)frag * 512 + lane * 8) = v;
}

__global__ __launch_bounds__(TAIL_THREADS) void tail_chain_bf16_kernel(TailArgsK pp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l16 = lane & 15, lq = lane >> 4;       // MFMA 16x16x32: operand row / column l16, k group lq (8 consecutive k)
    const int img = blockIdx.x;
    if (img >= pp.nimg) {
        // HELPER workgroups (the grid's tail): nobody's image.  A chain workgroup streams its filters from wherever they are, and
        // right after the pack kernel that is the memory-side cache at best (~60-90 GB/s per CU, profiles/r06_f_cu_stream_probe.txt)
        // against 130-150 GB/s out of its XCD's L2.  The helpers of an XCD (workgroup b runs on XCD b % 8 -- an affinity that only
        // speed depends on) touch every 128-byte line of every stage's packed filter once, in stage order, each its own slice, and
        // leave: by the time the chain reaches its second stage the lines it asks for are L2 hits.
        const int h = img - pp.nimg, nh = (int)gridDim.x - pp.nimg;
        const int xcd = h & 7, slot = h >> 3, nslots = (nh - xcd + 7) >> 3;
        unsigned fold = 0;
        for (int s = 0; s < pp.nstages; ++s) {
            const TailStageK& p = pp.st[s];
            const unsigned lines = (unsigned)(p.nblk_n * p.nsteps) * 8u;      // 128-byte lines of the packed filter
            const unsigned per = (lines + nslots - 1) / nslots;
            const unsigned l0 = slot * per, l1 = min(lines, l0 + per);
            const unsigned* w = reinterpret_cast<const unsigned*>(p.wgt);
            for (unsigned l = l0 + tid; l < l1; l += TAIL_THREADS) fold ^= w[(size_t)l * 32];
        }
        if (fold == 0x9E3779B9u && pp.nstages < 0) *reinterpret_cast<volatile unsigned*>(pp.st[0].dst) = fold;      // (never: keeps the loads)
        return;
    }
    for (int i = tid; i < TAIL_ZERO_BYTES / 16; i += TAIL_THREADS) *reinterpret_cast<u32x4*>(smem + i * 16) = u32x4{0u, 0u, 0u, 0u};

    // task `task` of stage q -> (k group, pixel group, n block) and its filter stream
    struct Task { int ks, t2, mg, nb, j0, j1; };
    auto decode = [&](const TailStageK& q, int task) -> Task {
        Task t;
        const int tiles = q.nblk_n * q.ngrp_m;
        t.ks = task / tiles;      // (the groups of one (pixel group, n block) are `tiles` apart in task order)
        t.t2 = task - t.ks * tiles;
        t.mg = t.t2 / q.nblk_n;
        t.nb = t.t2 - t.mg * q.nblk_n;
        const int per = (q.nsteps + q.ksplit - 1) / q.ksplit;
        t.j0 = t.ks * per;
        t.j1 = min((int)q.nsteps, t.j0 + per);
        return t;
    };
    auto stream_of = [&](const TailStageK& q, const Task& t) -> BStream {
        BStream b;
        b.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(q.wgt), 0, (unsigned)(q.nblk_n * q.nsteps) * 1024u, 0x00020000);
        const unsigned base = (unsigned)(t.nb * q.nsteps) * 1024u + (unsigned)lane * 16u;
        b.poff = base + (unsigned)t.j0 * 1024u;
        b.pend = base + (unsigned)t.j1 * 1024u;
        return b;
    };
    bf16x8 bq[TAIL_PF];
    BStream bs{};
    bool primed = false;      // bq / bs hold the first steps of this wave's first task of the coming stage
    if (wave < pp.st[0].nblk_n * pp.st[0].ngrp_m * pp.st[0].ksplit) {
        bs = stream_of(pp.st[0], decode(pp.st[0], wave));
        b_prime(bs, bq);
        primed = true;
    }

    for (int s = 0; s < pp.nstages; ++s) {
        const TailStageK& p = pp.st[s];
        const int flags = p.flags;
        const int M = p.DH * p.DW, DN = p.DN, SC = p.SC;
        const int C32 = (SC + 31) >> 5;
        const bf16_t* const src0 = p.src + (size_t)img * p.src_img;
        // ---- an input that no earlier stage left in LDS: the image's feature map, row by row, 16 bytes per thread and trip
        if (flags & TF_LOAD_IN) {
            lds_barrier();      // (the region may be one an earlier stage's readers have only just left)
            const int cpr = SC >> 3, total = p.SH * p.SW * cpr;      // 16-byte chunks per pixel row
            for (int i0 = 0; i0 < total; i0 += 4 * TAIL_THREADS) {
                u32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * TAIL_THREADS + tid;
                    v[u] = *reinterpret_cast<const u32x4*>(src0 + (size_t)(i < total ? i : 0) * 8);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * TAIL_THREADS + tid;
                    if (i < total) {
                        const int pix = i / cpr, ch = i - pix * cpr;
                        *reinterpret_cast<u32x4*>(smem + p.in_off + pix * p.in_pitch + ch * 16) = v[u];
                    }
                }
            }
        }
        if (flags & TF_LOAD_OUT) {      // a stage that accumulates into a tensor no earlier stage of the chain wrote: dx as it stands
            if (!(flags & TF_LOAD_IN)) lds_barrier();
            const bf16_t* const old0 = reinterpret_cast<const bf16_t*>(p.dst) + (size_t)img * p.dst_img;
            const int cpr = DN >> 3, total = M * cpr;
            for (int i = tid; i < total; i += TAIL_THREADS) {
                const int pix = i / cpr, ch = i - pix * cpr;
                *reinterpret_cast<u32x4*>(smem + p.out_off + pix * p.out_pitch + ch * 16) = *reinterpret_cast<const u32x4*>(old0 + (size_t)i * 8);
            }
        }
        lds_barrier();      // the input is in LDS (loaded above, or written by an earlier stage's epilogue); zero row written
        if (pp.stamps && img == 0 && tid == 0) pp.stamps[s] = __builtin_amdgcn_s_memrealtime();

        const int ntask = p.nblk_n * p.ngrp_m * p.ksplit;
        const size_t dst0 = (size_t)img * p.dst_img;
        const bool more = s + 1 < pp.nstages;
        const int ntask_next = more ? pp.st[s + 1].nblk_n * pp.st[s + 1].ngrp_m * pp.st[s + 1].ksplit : 0;
        // the first steps of this wave's first task of the NEXT stage, requested while this stage finishes
        auto prime_next = [&]() {
            primed = false;
            if (wave < ntask_next) {
                bs = stream_of(pp.st[s + 1], decode(pp.st[s + 1], wave));
                b_prime(bs, bq);
                primed = true;
            }
        };

        // (with a k split the host plans at most one task per wave: the partial sums then meet behind ONE workgroup barrier)
        f32x4 acc[TAIL_MG];
        Task t{};
        int nmb = 0;
        bool has_task = false;
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned mbits = 0;      // relu mask of the task's block: bit 4 i + e = (forward activation of pixel block i, channel n4 + e) > 0
        // bias and relu mask of the task's block, requested (and the mask folded into 16 bits) before its k loop
        auto pre_epilogue = [&]() {
            const int n4 = t.nb * 16 + 4 * lq;
            if (n4 >= DN || t.ks != 0) return;
            if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n4);
            if (p.mask) {
                u32x2 mk[TAIL_MG];
#pragma unroll
                for (int i = 0; i < TAIL_MG; ++i) {
                    const int m = (t.mg * TAIL_MG + i) * 16 + l16;
                    mk[i] = *reinterpret_cast<const u32x2*>(p.mask + dst0 + (size_t)(m < M ? m : 0) * DN + n4);
                }
                mbits = 0;
#pragma unroll
                for (int i = 0; i < TAIL_MG; ++i)
                    mbits |= ((lo2f(mk[i][0]) > 0.f ? 1u : 0u) | (hi2f(mk[i][0]) > 0.f ? 2u : 0u) | (lo2f(mk[i][1]) > 0.f ? 4u : 0u) | (hi2f(mk[i][1]) > 0.f ? 8u : 0u)) << (4 * i);
            }
        };
        auto epilogue = [&]() {
            // lane (l16, lq) holds pixel l16 of each block and channels nb * 16 + 4 lq .. + 3
            const int n4 = t.nb * 16 + 4 * lq;
            if (n4 >= DN) return;
#pragma unroll
            for (int i = 0; i < TAIL_MG; ++i) {
                const int m = (t.mg * TAIL_MG + i) * 16 + l16;
                if (i >= nmb || m >= M) continue;
                float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                if (p.bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bv[e];
                }
                if (flags & TF_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                if (flags & TF_ACCUM) {      // the earlier consumer's (rounded) contribution, from the LDS copy of dst
                    const u32x2 old = *reinterpret_cast<const u32x2*>(smem + p.out_off + m * p.out_pitch + n4 * 2);
                    v[0] += lo2f(old[0]); v[1] += hi2f(old[0]); v[2] += lo2f(old[1]); v[3] += hi2f(old[1]);
                }
                if (p.mask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ((mbits >> (4 * i + e)) & 1u) ? v[e] : 0.f;
                }
                const size_t o = dst0 + (size_t)m * DN + n4;
                if (flags & TF_OUT_F32) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.dst) + o) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    const u32x2 w = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.dst) + o) = w;
                    if (p.out_off >= 0) *reinterpret_cast<u32x2*>(smem + p.out_off + m * p.out_pitch + n4 * 2) = w;
                }
            }
        };

        if (wave >= ntask) prime_next();      // nothing to do in this stage
        for (int task = wave; task < ntask; task += TAIL_WAVES) {
            has_task = true;
            t = decode(p, task);
            if (!(task == wave && primed)) {
                bs = stream_of(p, t);
                b_prime(bs, bq);
            }
            nmb = min(TAIL_MG, (M - t.mg * TAIL_MG * 16 + 15) >> 4);      // pixel blocks that hold pixels (wave-uniform)
            if (pp.stamps && img == 0 && tid == 0) pp.stamps[2 * TAIL_MAX_STAGES + 1 + 4 * s] = __builtin_amdgcn_s_memrealtime();
            pre_epilogue();
#pragma unroll
            for (int i = 0; i < TAIL_MG; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            const KLoopArgs ka{smem, t.j0, t.j1, C32, SC, p.in_off, p.in_pitch, p.SH, p.SW, p.DW, p.mul, p.dshift, M, t.mg, p.tap_dh, p.tap_dw};
            if (nmb == 1) tail_kloop<1>(ka, bs, bq, acc, l16, lq);
            else if (nmb == 2) tail_kloop<2>(ka, bs, bq, acc, l16, lq);
            else if (nmb == 3) tail_kloop<3>(ka, bs, bq, acc, l16, lq);
            else tail_kloop<4>(ka, bs, bq, acc, l16, lq);
            if (pp.stamps && img == 0 && tid == 0) pp.stamps[2 * TAIL_MAX_STAGES + 2 + 4 * s] = __builtin_amdgcn_s_memrealtime();
            const bool last_task = task + TAIL_WAVES >= ntask;      // this wave's last task of the stage: the next stage's filter can be requested
            if (p.ksplit == 1) {
                epilogue();
                if (last_task) prime_next();      // (behind the epilogue's own loads: a load queued behind the ring would wait for all of it)
            } else if (t.ks > 0 && last_task) {
                // (parks its sums below, then has nothing left to do in this stage)
            }
        }
        if (p.ksplit > 1) {
            // k-split: groups > 0 park their sums, group 0 adds them in group order and writes the block out
            const int tiles = p.nblk_n * p.ngrp_m;
            if (has_task && t.ks > 0) {
                float* sc = reinterpret_cast<float*>(smem + p.scratch_off) + (size_t)((t.ks - 1) * tiles + t.t2) * (TAIL_MG * 256);
#pragma unroll
                for (int i = 0; i < TAIL_MG; ++i)
                    if (i < nmb) *reinterpret_cast<f32x4*>(sc + (i * 64 + lane) * 4) = acc[i];
                prime_next();
            }
            lds_barrier();
            if (pp.stamps && img == 0 && tid == 0) pp.stamps[2 * TAIL_MAX_STAGES + 3 + 4 * s] = __builtin_amdgcn_s_memrealtime();
            if (has_task && t.ks == 0) {
                for (int g = 1; g < p.ksplit; ++g) {
                    const float* sc = reinterpret_cast<const float*>(smem + p.scratch_off) + (size_t)((g - 1) * tiles + t.t2) * (TAIL_MG * 256);
#pragma unroll
                    for (int i = 0; i < TAIL_MG; ++i)
                        if (i < nmb) acc[i] += *reinterpret_cast<const f32x4*>(sc + (i * 64 + lane) * 4);
                }
                epilogue();
                prime_next();
            }
        }
        // (the next stage's first barrier publishes this stage's LDS output)
        if (pp.stamps && img == 0 && tid == 0) pp.stamps[TAIL_MAX_STAGES + s] = __builtin_amdgcn_s_memrealtime();      // thread 0's own end of the stage
    }
    lds_barrier();
    if (pp.stamps && img == 0 && tid == 0) pp.stamps[pp.nstages] = __builtin_amdgcn_s_memrealtime();
}

struct Region {
    const void* key;
    int off, bytes, first, last;
};

}  // namespace

int tail_chain_max_stages() { return TAIL_MAX_STAGES; }

// elements (bf16) of the packed form of a stage's filter: whole 16-channel blocks x whole 32-k steps
static void stage_pack_dims(const ConvDesc& d, bool dgrad, int* DN, int* SC, int* nblk, int* nsteps) {
    *DN = dgrad ? d.Ci : d.Co;
    *SC = dgrad ? d.Co : d.Ci;
    *nblk = cdiv(*DN, 16);
    *nsteps = d.KH * d.KW * cdiv(*SC, 32);
}
size_t tail_chain_packed_elems(const ConvDesc& d, bool dgrad) {
    int DN, SC, nblk, nsteps;
    stage_pack_dims(d, dgrad, &DN, &SC, &nblk, &nsteps);
    return (size_t)nblk * nsteps * 512;
}
void tail_chain_pack_filters(const TailPackItem* items, int n, hipStream_t s) {
    SSD_REQUIRE(n >= 1 && n <= TAIL_PACK_MAX, "tail chain: 1..%d filters per pack launch (got %d)", TAIL_PACK_MAX, n);
    TailPackK t{};
    t.n = n;
    double bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        int DN, SC, nblk, nsteps;
        stage_pack_dims(items[i].d, items[i].dgrad, &DN, &SC, &nblk, &nsteps);
        t.l[i].w = static_cast<const bf16_t*>(items[i].mirror); t.l[i].out = static_cast<bf16_t*>(items[i].packed);
        t.l[i].DN = DN; t.l[i].SC = SC; t.l[i].C32 = cdiv(SC, 32); t.l[i].nsteps = nsteps;
        t.frag0[i + 1] = t.frag0[i] + nblk * nsteps;
        bytes += 2.0 * nblk * nsteps * 1024;
    }
    ProfScope prof("tail_pack_filters", 0.0, bytes, s);
    hipLaunchKernelGGL(tail_pack_kernel, dim3(cdiv(t.frag0[n], 4)), dim3(256), 0, s, t);
    HIP_OK(hipGetLastError());
}

bool tail_chain_stage_supported(const ConvDesc& d) {
    // ONE workgroup walks an image's pixels: its feature maps must fit LDS (the 10x10 map and below)
    return d.KH * d.KW <= 9 && d.Ci % 8 == 0 && d.Co % 8 == 0 && d.stride >= 1 && (d.stride & (d.stride - 1)) == 0 &&
           d.Ci <= 1024 && d.Co <= 1024 && (size_t)d.Hi * d.Wi * (d.Ci * 2 + 16) <= 110 * 1024 && (size_t)d.Ho * d.Wo * (d.Co * 2 + 16) <= 110 * 1024;
}

void tail_chain_bf16(const TailStage* stages, int nstages, int nimg, const char* label, hipStream_t s) {
    SSD_REQUIRE(nstages >= 1 && nstages <= TAIL_MAX_STAGES, "tail chain: 1..%d stages (got %d)", TAIL_MAX_STAGES, nstages);
    SSD_REQUIRE(nimg >= 1, "tail chain: no images");
    TailArgsK a{};
    a.nstages = nstages; a.nimg = nimg;
    double flops = 0.0, bytes = 0.0;
    // ---- per-stage geometry
    for (int i = 0; i < nstages; ++i) {
        const TailStage& t = stages[i];
        ConvDesc d = t.d;
        SSD_REQUIRE(tail_chain_stage_supported(d), "tail chain: stage %d has an unsupported shape", i);
        TailStageK& k = a.st[i];
        SSD_REQUIRE(t.wgt_packed != nullptr, "tail chain: stage %d has no packed filter (tail_chain_pack_filter)", i);
        k.src = static_cast<const bf16_t*>(t.src); k.wgt = static_cast<const bf16_t*>(t.wgt_packed); k.bias = t.bias;
        k.mask = static_cast<const bf16_t*>(t.mask); k.dst = t.dst;
        k.ntaps = (short)(d.KH * d.KW);
        if (!t.dgrad) {
            k.DH = d.Ho; k.DW = d.Wo; k.DN = d.Co; k.SH = d.Hi; k.SW = d.Wi; k.SC = d.Ci;
            k.mul = d.stride; k.dshift = 0;
            SSD_REQUIRE(!t.mask && !t.accum, "tail chain: mask / accumulate belong to a data-gradient stage");
        } else {
            k.DH = d.Hi; k.DW = d.Wi; k.DN = d.Ci; k.SH = d.Ho; k.SW = d.Wo; k.SC = d.Co;
            k.mul = 1; k.dshift = 0;
            while ((1 << k.dshift) < d.stride) ++k.dshift;
            SSD_REQUIRE(!t.bias && !t.relu && !t.out_f32, "tail chain: bias / relu / fp32 output belong to a forward stage");
        }
        for (int kh = 0; kh < d.KH; ++kh)
            for (int kw = 0; kw < d.KW; ++kw) {
                const int dh = t.dgrad ? d.pad_h - kh * d.dil : kh * d.dil - d.pad_h;
                const int dw = t.dgrad ? d.pad_w - kw * d.dil : kw * d.dil - d.pad_w;
                SSD_REQUIRE(dh >= -127 && dh <= 127 && dw >= -127 && dw <= 127, "tail chain: tap offset out of range");
                k.tap_dh[kh * d.KW + kw] = dh;
                k.tap_dw[kh * d.KW + kw] = dw;
            }
        k.src_img = (unsigned)(k.SH * k.SW * k.SC);
        k.dst_img = (unsigned)(k.DH * k.DW * k.DN);
        k.flags = (short)((t.relu ? TF_RELU : 0) | (t.accum ? TF_ACCUM : 0) | (t.out_f32 ? TF_OUT_F32 : 0));
        SSD_REQUIRE(k.DN % 4 == 0, "tail chain: output channels must be a multiple of 4");
        // tasks: n blocks of 16 channels x groups of <= 4 pixel blocks of 16; k split over the waves that would idle
        const int M = k.DH * k.DW;
        k.nblk_n = (short)cdiv(k.DN, 16);
        k.ngrp_m = (short)cdiv(cdiv(M, 16), TAIL_MG);
        k.nsteps = (short)(k.ntaps * cdiv(k.SC, 32));
        int ksplit = TAIL_WAVES / (k.nblk_n * k.ngrp_m);
        if (ksplit < 1) ksplit = 1;
        if (ksplit > 4) ksplit = 4;
        while (ksplit > 1 && k.nsteps / ksplit < 4) --ksplit;      // (a handful of steps per group at least)
        k.ksplit = (short)ksplit;
        SSD_REQUIRE(ksplit == 1 || k.nblk_n * k.ngrp_m * ksplit <= TAIL_WAVES, "tail chain: k split needs one task per wave");
        d.B = nimg;
        flops += conv_flops(d);
        bytes += 2.0 * conv_elems(d);
    }
    // ---- LDS plan.  A tensor (keyed by its pointer) lives in LDS from the stage that loads / produces it to the last stage that
    // gathers from it or accumulates into it; regions are placed first-fit above the zero row, a stage's k-split scratch above
    // everything live during that stage.
    std::vector<Region> regs;
    auto find = [&](const void* key) -> Region* {
        for (Region& r : regs)
            if (r.key == key) return &r;
        return nullptr;
    };
    for (int i = 0; i < nstages; ++i) {
        const TailStage& t = stages[i];
        const TailStageK& k = a.st[i];
        Region* in = find(t.src);
        if (!in) {
            regs.push_back(Region{t.src, -1, k.SH * k.SW * (k.SC * 2 + 16), i, i});
            a.st[i].flags |= TF_LOAD_IN;
        } else {
            in->last = i;
        }
        if (!t.out_f32) {
            Region* out = find(t.dst);
            bool later = false;
            for (int j = i + 1; j < nstages; ++j)
                if (stages[j].src == t.dst || (stages[j].dst == t.dst && stages[j].accum)) later = true;
            if (t.accum && out) {
                out->last = i;
            } else if (t.accum) {      // accumulates into a tensor written before the launch: the stage loads it first
                SSD_REQUIRE(k.DN % 8 == 0, "tail chain: an accumulated tensor needs a multiple of 8 channels");
                regs.push_back(Region{t.dst, -1, k.DH * k.DW * (k.DN * 2 + 16), i, i});
                a.st[i].flags |= TF_LOAD_OUT;
            } else if (later) {
                SSD_REQUIRE(out == nullptr, "tail chain: stage %d overwrites a tensor an earlier stage left in LDS", i);
                regs.push_back(Region{t.dst, -1, k.DH * k.DW * (k.DN * 2 + 16), i, i});
            }
        } else {
            SSD_REQUIRE(!t.accum, "tail chain: fp32 outputs are not accumulated");
        }
    }
    int lds_total = TAIL_ZERO_BYTES;
    std::vector<Region*> order;
    for (Region& r : regs) order.push_back(&r);
    std::sort(order.begin(), order.end(), [](const Region* x, const Region* y) { return x->first < y->first; });
    for (Region* r : order) {
        int off = TAIL_ZERO_BYTES;
        bool moved = true;
        while (moved) {
            moved = false;
            for (const Region* q : order)
                if (q != r && q->off >= 0 && !(q->last < r->first || r->last < q->first) && off < q->off + q->bytes && q->off < off + r->bytes) {
                    off = (q->off + q->bytes + 15) / 16 * 16;
                    moved = true;
                }
        }
        r->off = off;
        lds_total = std::max(lds_total, off + r->bytes);
    }
    for (int i = 0; i < nstages; ++i) {
        TailStageK& k = a.st[i];
        const Region* in = find(stages[i].src);
        k.in_off = in->off; k.in_pitch = k.SC * 2 + 16;
        const Region* out = stages[i].out_f32 ? nullptr : find(stages[i].dst);
        k.out_off = out ? out->off : -1; k.out_pitch = k.DN * 2 + 16;
        int top = TAIL_ZERO_BYTES;
        for (const Region& r : regs)
            if (r.first <= i && i <= r.last) top = std::max(top, r.off + r.bytes);
        k.scratch_off = (top + 15) / 16 * 16;
        const int scratch = (k.ksplit - 1) * k.nblk_n * k.ngrp_m * TAIL_MG * 256 * 4;
        lds_total = std::max(lds_total, k.scratch_off + scratch);
    }
    SSD_REQUIRE(lds_total <= TAIL_LDS_MAX, "tail chain: the stages' feature maps need %d bytes of LDS (limit %d)", lds_total, TAIL_LDS_MAX);
    static bool once = (set_lds(tail_chain_bf16_kernel, TAIL_LDS_MAX), true);
    (void)once;
    ProfScope prof(label, flops, bytes, s);
    static const int stamps_on = env_int("SSD_TAIL_STAMPS", 0);
    static unsigned long long* stamps_dev = nullptr;
    if (stamps_on) {
        if (!stamps_dev) { HIP_OK(hipMalloc((void**)&stamps_dev, 6 * TAIL_MAX_STAGES * 8 + 8)); HIP_OK(hipMemset(stamps_dev, 0, 6 * TAIL_MAX_STAGES * 8 + 8)); }
        a.stamps = stamps_dev;
    }
    static const int helpers = env_int("SSD_TAIL_HELPERS", 64);      // A/B switch: 0 = no L2 warm-up workgroups
    SSD_LAUNCH_STOP(tail_chain_bf16_kernel, dim3(nimg + (helpers > 0 ? helpers : 0)), dim3(TAIL_THREADS), (size_t)lds_total, s, a);
    HIP_OK(hipGetLastError());
    if (stamps_on) {      // measurement aid: serializes the stream
        unsigned long long h[6 * TAIL_MAX_STAGES + 1];
        HIP_OK(hipStreamSynchronize(s));
        HIP_OK(hipMemcpy(h, stamps_dev, sizeof h, hipMemcpyDeviceToHost));
        fprintf(stderr, "[tail stamps] %s:", label);
        for (int i = 0; i < nstages; ++i)
            fprintf(stderr, " s%d(M=%d,N=%d,K=%d,tasks=%dx%dx%d) start+%.1f [setup %.1f kloop %.1f ksplit-barrier %.1f] t0end+%.1f |", i, a.st[i].DH * a.st[i].DW, a.st[i].DN,
                    a.st[i].SC * a.st[i].ntaps, a.st[i].nblk_n, a.st[i].ngrp_m, a.st[i].ksplit, (h[i] - h[0]) / 100.0,
                    (h[2 * TAIL_MAX_STAGES + 1 + 4 * i] - h[i]) / 100.0, (h[2 * TAIL_MAX_STAGES + 2 + 4 * i] - h[2 * TAIL_MAX_STAGES + 1 + 4 * i]) / 100.0,
                    a.st[i].ksplit > 1 ? (h[2 * TAIL_MAX_STAGES + 3 + 4 * i] - h[2 * TAIL_MAX_STAGES + 2 + 4 * i]) / 100.0 : 0.0, (h[TAIL_MAX_STAGES + i] - h[0]) / 100.0);
        fprintf(stderr, " end+%.1f us\n", (h[nstages] - h[0]) / 100.0);
    }
}

}  // namespace ssd
