// Box math of the SSD hot path for gfx950.  Integer / index results are bit-exact with the
// reference's numpy code (ssdutils.py, transforms.py, utils.py geometry) as it runs under
// numpy >= 2: the mixed f32/f64 arithmetic of decode_location / normalize_box is spelled out
// with explicitly rounded operations.  This file is compiled with -ffp-contract=off.
#include "boxes.h"
#include <cmath>
#include <climits>

namespace ssd {

typedef unsigned long long u64;

// =================================================================================
// presets (ssdutils.py:36-62) and box sizes (ssdutils.py:83-99) -- host
// =================================================================================
static Preset make_preset(const char* name, int img, int nmaps, const int* sizes, const double* scales, int last_two,
                          double extra, int A) {
    Preset p{};
    p.name = name; p.image_w = p.image_h = img; p.nmaps = nmaps; p.extra_scale = extra; p.num_anchors = A;
    for (int k = 0; k < nmaps; ++k) {
        p.map_size[k] = sizes[k];
        p.scale[k] = scales[k];
        const bool two = (k == 0) || (k >= nmaps - last_two);
        if (two) {
            p.nratios[k] = 2; p.ratios[k][0] = 2; p.ratios[k][1] = 0.5;
        } else {
            p.nratios[k] = 4; p.ratios[k][0] = 2; p.ratios[k][1] = 3; p.ratios[k][2] = 0.5; p.ratios[k][3] = 1. / 3.;
        }
    }
    int off = 0;
    for (int k = 0; k < nmaps; ++k) {
        const double s = p.scale[k];
        int t = 0;
        double r = std::sqrt(1.0);
        p.bw[k][t] = s * r; p.bh[k][t] = s / r; ++t;
        for (int q = 0; q < p.nratios[k]; ++q) {
            r = std::sqrt(p.ratios[k][q]);
            p.bw[k][t] = s * r; p.bh[k][t] = s / r; ++t;
        }
        const double nxt = k < nmaps - 1 ? p.scale[k + 1] : extra;
        const double sp = std::sqrt(s * nxt);
        p.bw[k][t] = sp; p.bh[k][t] = sp; ++t;
        p.ntypes[k] = t;
        p.off[k] = off;
        off += t * sizes[k] * sizes[k];
    }
    p.off[nmaps] = off;
    if (off != A) fail("preset %s: anchor count %d != %d", name, off, A);
    return p;
}

const Preset& get_preset(const char* name) {
    static const int s300[] = {38, 19, 10, 5, 3, 1};
    static const double c300[] = {0.1, 0.2, 0.375, 0.55, 0.725, 0.9};
    static const int s512[] = {64, 32, 16, 8, 4, 2, 1};
    static const double c512[] = {0.07, 0.15, 0.3, 0.45, 0.6, 0.75, 0.9};
    static const Preset p300 = make_preset("vgg300", 300, 6, s300, c300, 2, 1.075, 8732);
    static const Preset p512 = make_preset("vgg512", 512, 7, s512, c512, 2, 1.05, 24564);
    if (name && !strcmp(name, "vgg300")) return p300;
    if (name && !strcmp(name, "vgg512")) return p512;
    fail("No such preset: %s", name ? name : "(null)");
    return p300;
}

// =================================================================================
// anchors (ssdutils.py:104-130): order map -> type -> row -> col
// =================================================================================
struct AnchorArgs {
    int nmaps, A;
    int fk[PRESET_MAX_MAPS], off[PRESET_MAX_MAPS + 1];
    double bw[PRESET_MAX_MAPS][PRESET_MAX_TYPES], bh[PRESET_MAX_MAPS][PRESET_MAX_TYPES];
};

// utils.py:100-108 in f64: int() truncates toward zero
__device__ __forceinline__ void prop2abs_f64(double cx, double cy, double w, double h, int* o) {
    const double w2 = __ddiv_rn(__dmul_rn(w, 1000.0), 2.0);
    const double h2 = __ddiv_rn(__dmul_rn(h, 1000.0), 2.0);
    const double ax = __dmul_rn(cx, 1000.0), ay = __dmul_rn(cy, 1000.0);
    o[0] = (int)(long long)__dsub_rn(ax, w2);
    o[1] = (int)(long long)__dadd_rn(ax, w2);
    o[2] = (int)(long long)__dsub_rn(ay, h2);
    o[3] = (int)(long long)__dadd_rn(ay, h2);
}

__global__ void anchors_kernel(AnchorArgs p, double* __restrict__ anchors, int* __restrict__ aabs) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.A) return;
    int k = 0;
    for (int q = 1; q < p.nmaps; ++q)
        if (a >= p.off[q]) k = q;
    const int fk = p.fk[k];
    const int al = a - p.off[k];
    const int t = al / (fk * fk);
    const int cell = al - t * fk * fk;
    const int j = cell / fk, i = cell - j * fk;
    const double x = __ddiv_rn((double)i + 0.5, (double)fk);
    const double y = __ddiv_rn((double)j + 0.5, (double)fk);
    const double w = p.bw[k][t], h = p.bh[k][t];
    anchors[a * 4 + 0] = x; anchors[a * 4 + 1] = y; anchors[a * 4 + 2] = w; anchors[a * 4 + 3] = h;
    if (aabs) prop2abs_f64(x, y, w, h, aabs + a * 4);
}

void anchors_device(const Preset& p, double* anchors, int* anchors_abs, hipStream_t s) {
    AnchorArgs a{};
    a.nmaps = p.nmaps; a.A = p.num_anchors;
    for (int k = 0; k < p.nmaps; ++k) {
        a.fk[k] = p.map_size[k];
        a.off[k] = p.off[k];
        for (int t = 0; t < PRESET_MAX_TYPES; ++t) { a.bw[k][t] = p.bw[k][t]; a.bh[k][t] = p.bh[k][t]; }
    }
    a.off[p.nmaps] = p.off[p.nmaps];
    hipLaunchKernelGGL(anchors_kernel, dim3((a.A + 255) / 256), dim3(256), 0, s, a, anchors, anchors_abs);
    HIP_OK(hipGetLastError());
}

// =================================================================================
// label encoding (transforms.py:57-114).  IoU uses the +1 pixel convention on integer
// boxes (ssdutils.py:138-152); ratios are compared exactly by cross-multiplication.
// =================================================================================
__device__ __forceinline__ void iou_terms(const int* g, long long areab, const int* a, long long* inter, long long* uni) {
    const long long areaa = (long long)(a[1] - a[0] + 1) * (a[3] - a[2] + 1);
    const int w = max(0, min(g[1], a[1]) - max(g[0], a[0]) + 1);
    const int h = max(0, min(g[3], a[3]) - max(g[2], a[2]) + 1);
    *inter = (long long)w * h;
    *uni = areab + areaa - *inter;
}

// one workgroup per GT box: its best anchor (np.argmax: first maximum), ssdutils.py:159-165
__global__ __launch_bounds__(256) void gt_best_kernel(int A, const int* __restrict__ aabs, const double* __restrict__ gt,
                                                      int ntot, int* __restrict__ gabs, int* __restrict__ best_a,
                                                      int* __restrict__ best_i, int* __restrict__ best_u) {
    __shared__ long long s_i[256], s_u[256];
    __shared__ int s_a[256];
    __shared__ int gb[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        prop2abs_f64(gt[g * 4 + 0], gt[g * 4 + 1], gt[g * 4 + 2], gt[g * 4 + 3], gb);
        for (int e = 0; e < 4; ++e) gabs[g * 4 + e] = gb[e];
    }
    __syncthreads();
    const int box[4] = {gb[0], gb[1], gb[2], gb[3]};
    const long long areab = (long long)(box[1] - box[0] + 1) * (box[3] - box[2] + 1);
    long long bi = -1, bu = 1;
    int ba = INT_MAX;
    for (int a = tid; a < A; a += 256) {
        long long in, un;
        iou_terms(box, areab, aabs + a * 4, &in, &un);
        if (in * bu > bi * un) { bi = in; bu = un; ba = a; }     // strict: the earlier index keeps a tie
    }
    s_i[tid] = bi; s_u[tid] = bu; s_a[tid] = ba;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            const long long i2 = s_i[tid + st], u2 = s_u[tid + st];
            const int a2 = s_a[tid + st];
            const long long l = i2 * s_u[tid], r = s_i[tid] * u2;
            if (l > r || (l == r && a2 < s_a[tid])) { s_i[tid] = i2; s_u[tid] = u2; s_a[tid] = a2; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const bool ok = 2 * s_i[0] > s_u[0];                      // iou > 0.5
        best_a[g] = ok ? s_a[0] : -1;
        best_i[g] = (int)s_i[0];
        best_u[g] = (int)s_u[0];
    }
}

// one thread per (image, anchor): the winning GT box (pass 1: every 'good' overlap, higher IoU
// wins, earlier box keeps a tie; pass 2: 'best' matches override, same rule among them) and
// the encoded row (ssdutils.py:173-179, transforms.py:47-55).
__global__ __launch_bounds__(256) void label_rows_kernel(int A, int C, int B, const double* __restrict__ anchors,
                                                         const int* __restrict__ aabs, const double* __restrict__ gt,
                                                         const int* __restrict__ cls, const int* __restrict__ offsets,
                                                         const int* __restrict__ gabs, const int* __restrict__ best_a,
                                                         float* __restrict__ vec) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * A) return;
    const int b = idx / A, a = idx - b * A;
    const int an[4] = {aabs[a * 4], aabs[a * 4 + 1], aabs[a * 4 + 2], aabs[a * 4 + 3]};
    int w1 = -1, w2 = -1;
    long long i1 = 0, u1 = 1, i2 = 0, u2 = 1;
    for (int g = offsets[b]; g < offsets[b + 1]; ++g) {
        const int box[4] = {gabs[g * 4], gabs[g * 4 + 1], gabs[g * 4 + 2], gabs[g * 4 + 3]};
        const long long areab = (long long)(box[1] - box[0] + 1) * (box[3] - box[2] + 1);
        long long in, un;
        iou_terms(box, areab, an, &in, &un);
        if (2 * in > un) {
            if (w1 < 0 || in * u1 > i1 * un) { w1 = g; i1 = in; u1 = un; }
        }
        if (best_a[g] == a) {
            if (w2 < 0 || in * u2 > i2 * un) { w2 = g; i2 = in; u2 = un; }
        }
    }
    const int win = w2 >= 0 ? w2 : w1;
    float* row = vec + (size_t)idx * (C + 5);
    for (int c = 0; c < C + 5; ++c) row[c] = 0.f;
    if (win < 0) {
        row[C] = 1.f;
        return;
    }
    row[cls[win]] = 1.f;
    const double acx = anchors[a * 4], acy = anchors[a * 4 + 1], aw = anchors[a * 4 + 2], ah = anchors[a * 4 + 3];
    const double bcx = gt[win * 4], bcy = gt[win * 4 + 1], bw = gt[win * 4 + 2], bh = gt[win * 4 + 3];
    row[C + 1] = (float)__dmul_rn(__ddiv_rn(__dsub_rn(bcx, acx), aw), 10.0);
    row[C + 2] = (float)__dmul_rn(__ddiv_rn(__dsub_rn(bcy, acy), ah), 10.0);
    row[C + 3] = (float)__dmul_rn(log(__ddiv_rn(bw, aw)), 5.0);
    row[C + 4] = (float)__dmul_rn(log(__ddiv_rn(bh, ah)), 5.0);
}

// jaccard_overlap (ssdutils.py:138-152) of one box against n boxes, all (xmin, xmax, ymin, ymax) f64
__global__ void jaccard_kernel(const double* __restrict__ box, const double* __restrict__ arr, int n, double* __restrict__ iou) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double b0 = box[0], b1 = box[1], b2 = box[2], b3 = box[3];
    const double a0 = arr[i * 4], a1 = arr[i * 4 + 1], a2 = arr[i * 4 + 2], a3 = arr[i * 4 + 3];
    const double areaa = __dmul_rn(__dadd_rn(__dsub_rn(a1, a0), 1.0), __dadd_rn(__dsub_rn(a3, a2), 1.0));
    const double areab = __dmul_rn(__dadd_rn(__dsub_rn(b1, b0), 1.0), __dadd_rn(__dsub_rn(b3, b2), 1.0));
    const double w = fmax(0.0, __dadd_rn(__dsub_rn(fmin(b1, a1), fmax(b0, a0)), 1.0));
    const double h = fmax(0.0, __dadd_rn(__dsub_rn(fmin(b3, a3), fmax(b2, a2)), 1.0));
    const double inter = __dmul_rn(w, h);
    iou[i] = __ddiv_rn(inter, __dsub_rn(__dadd_rn(areab, areaa), inter));
}

void jaccard_device(const double* box, const double* arr, int n, double* iou, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(jaccard_kernel, dim3((n + 255) / 256), dim3(256), 0, s, box, arr, n, iou);
    HIP_OK(hipGetLastError());
}

size_t encode_labels_ws_bytes(int ntot) { return (size_t)(ntot > 0 ? ntot : 1) * 7 * sizeof(int) + 64; }

void encode_labels(const Preset& p, int num_classes, const double* anchors, const int* anchors_abs, const double* gt,
                   const int* cls, const int* offsets, int B, int ntot, float* vec, void* ws, hipStream_t s) {
    int* gabs = (int*)ws;
    int* best_a = gabs + (size_t)(ntot > 0 ? ntot : 1) * 4;
    int* best_i = best_a + (ntot > 0 ? ntot : 1);
    int* best_u = best_i + (ntot > 0 ? ntot : 1);
    const int A = p.num_anchors;
    if (ntot > 0)
        hipLaunchKernelGGL(gt_best_kernel, dim3(ntot), dim3(256), 0, s, A, anchors_abs, gt, ntot, gabs, best_a, best_i, best_u);
    hipLaunchKernelGGL(label_rows_kernel, dim3((B * A + 255) / 256), dim3(256), 0, s, A, num_classes, B, anchors, anchors_abs,
                       gt, cls, offsets, gabs, best_a, vec);
    HIP_OK(hipGetLastError());
}

// =================================================================================
// decode (ssdutils.py:192-229) + per-class greedy NMS (ssdutils.py:232-318)
// =================================================================================
// Pass 1, HBM-bound: every anchor row [C+5] f32 is read exactly once through LDS (coalesced
// float4 loads, all issued before the first is consumed; rows are then read at an odd stride:
// conflict-free).  Every anchor whose confidence reaches thr becomes one 64-bit sort key
//   conf bits << 32 | (32767 - anchor) << 8 | 0x80 | class   (descending sort == conf desc, anchor asc)
// A workgroup compacts the keys of its 256 rows in place: ballot prefix per wave, wave totals through LDS,
// the survivors stored from the first row of the segment on, their number in `bcount`.  A workgroup's rows
// belong to at most two images (A >= 256): two segments, the second one starting at the image boundary.
// No atomics (a per-image counter serialises in L2: 137 waves hit each address), no memset, deterministic.
constexpr int SCAN_ROWS = 256;
constexpr int SCAN_MAXV = 32;                       // nv <= 32
constexpr int SCAN_LOADS = SCAN_ROWS * SCAN_MAXV / 4 / 256;   // float4 per thread, worst case

// One scan block = SCAN_ROWS rows, worked by the 256 threads t = 0..255 of a (half) workgroup; `rows` and `s_cnt` are that
// half's LDS.  blk past the last block: nothing to do, but every barrier is still passed.
__device__ __forceinline__ void detect_scan_block(const int blk, const int t, float* __restrict__ rows, int (*s_cnt)[4], int A, int nv,
                                                  int B, const float* __restrict__ pred, float thr, u64* __restrict__ dense,
                                                  int* __restrict__ bcount) {
    const int total_rows = B * A;
    const int r0 = blk * SCAN_ROWS;
    const int nrows = max(0, min(SCAN_ROWS, total_rows - r0));
    const int nfl = nrows * nv;
    const float* src = pred + (size_t)r0 * nv;           // r0*nv*4 bytes: 256*nv*4*block -> 16-byte aligned
    const int n4 = nfl >> 2;
    // all of this thread's loads are issued before the first one is consumed
    float4 v[SCAN_LOADS];
#pragma unroll
    for (int j = 0; j < SCAN_LOADS; ++j) {
        const int i = t + 256 * j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n4) v[j] = *reinterpret_cast<const float4*>(src + (size_t)i * 4);
    }
#pragma unroll
    for (int j = 0; j < SCAN_LOADS; ++j) {
        const int i = t + 256 * j;
        if (i < n4) *reinterpret_cast<float4*>(rows + (size_t)i * 4) = v[j];
    }
    for (int i = (n4 << 2) + t; i < nfl; i += 256) rows[i] = src[i];
    __syncthreads();
    const int lane = t & 63, wv = t >> 6;
    const int img_first = r0 / A;
    const int boundary = (img_first + 1) * A;            // first row of the next image (may lie beyond this workgroup)
    u64 key = 0ull;
    int half = 0;
    if ((int)t < nrows) {
        const float* r = rows + (size_t)t * nv;
        const int nfg = nv - 5;                  // argmax excludes the background class
        int best = 0;
        float conf = r[0];
        for (int c = 1; c < nfg; ++c)
            if (r[c] > conf) { conf = r[c]; best = c; }     // first maximum wins (np.argmax)
        const int row = r0 + t;
        half = row >= boundary ? 1 : 0;
        const int a = row - (img_first + half) * A;
        if (!(conf < thr))                        // the reference breaks at the first conf < thr
            key = ((u64)__float_as_uint(conf) << 32) | ((u64)(32767 - a) << 8) | (u64)best | (1ull << 7);
    }
    const u64 bal0 = __ballot(key != 0ull && half == 0), bal1 = __ballot(key != 0ull && half == 1);
    if (lane == 0) { s_cnt[0][wv] = __popcll(bal0); s_cnt[1][wv] = __popcll(bal1); }
    __syncthreads();
    int base = 0, tot0 = 0, tot1 = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wv) base += s_cnt[half][w];
        tot0 += s_cnt[0][w]; tot1 += s_cnt[1][w];
    }
    if (key != 0ull) {
        const u64 bal = half ? bal1 : bal0;
        __hip_atomic_store(dense + (size_t)(half ? boundary : r0) + base + __popcll(bal & ((1ull << lane) - 1ull)), key, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t == 0 && nrows > 0) {
        __hip_atomic_store(bcount + blk * 2, tot0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(bcount + blk * 2 + 1, tot1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this lane's stores have left before the workgroup's barrier
}

__global__ __launch_bounds__(256) void detect_scan_kernel(int A, int nv, int B, const float* __restrict__ pred, float thr,
                                                          u64* __restrict__ dense, int* __restrict__ bcount) {
    extern __shared__ __attribute__((aligned(16))) float rows[];
    __shared__ int s_cnt[2][4];
    detect_scan_block(blockIdx.x, threadIdx.x, rows, s_cnt, A, nv, B, pred, thr, dense, bcount);
}


// descending bitonic sort of n2 (power of two) keys by one workgroup; keys may live in LDS or global
__device__ void bitonic_desc(u64* keys, int n2) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const u64 a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if (up ? (a < b) : (a > b)) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

// decode_location under numpy>=2 promotion (x, y float32; w, h float64) followed by
// normalize_box's integer box (utils.py:118-135; centre*1000 in f32, half extent cast to f32).
__device__ __forceinline__ void decode_box(const float* loc, const double* an, int* o) {
    float l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) l[e] = loc[e] > 100.f ? 100.f : loc[e];      // ssdutils.py:183
    const double acx = an[0], acy = an[1], aw = an[2], ah = an[3];
    const float x = __fadd_rn(__fmul_rn(__fdiv_rn(l[0], 10.f), (float)aw), (float)acx);
    const float y = __fadd_rn(__fmul_rn(__fdiv_rn(l[1], 10.f), (float)ah), (float)acy);
    const double w = __dmul_rn(exp((double)__fdiv_rn(l[2], 5.f)), aw);
    const double h = __dmul_rn(exp((double)__fdiv_rn(l[3], 5.f)), ah);
    if (!(isfinite(x) && isfinite(y) && isfinite(w) && isfinite(h))) {
        // the reference passes such a box through un-normalised and then fails in NMS's int();
        // here it becomes an empty corner box
        o[0] = o[1] = o[2] = o[3] = 0;
        return;
    }
    const float w2 = (float)__ddiv_rn(__dmul_rn(w, 1000.0), 2.0);
    const float h2 = (float)__ddiv_rn(__dmul_rn(h, 1000.0), 2.0);
    const float cx = __fmul_rn(x, 1000.f), cy = __fmul_rn(y, 1000.f);
    const float fx0 = __fsub_rn(cx, w2), fx1 = __fadd_rn(cx, w2), fy0 = __fsub_rn(cy, h2), fy1 = __fadd_rn(cy, h2);
    const float big = 9.0e18f;
    long long xmin = (long long)fminf(fmaxf(fx0, -big), big), xmax = (long long)fminf(fmaxf(fx1, -big), big);
    long long ymin = (long long)fminf(fmaxf(fy0, -big), big), ymax = (long long)fminf(fmaxf(fy1, -big), big);
    xmin = xmin > 0 ? xmin : 0; xmax = xmax < 999 ? xmax : 999;
    ymin = ymin > 0 ? ymin : 0; ymax = ymax < 999 ? ymax : 999;
    xmin = xmin < xmax ? xmin : xmax;
    ymin = ymin < ymax ? ymin : ymax;
    const long long lo = -(1LL << 30);
    o[0] = (int)(xmin > lo ? xmin : lo); o[1] = (int)(xmax > lo ? xmax : lo);
    o[2] = (int)(ymin > lo ? ymin : lo); o[3] = (int)(ymax > lo ? ymax : lo);
}

// prop2abs(abs2prop(ints)) as non_maximum_suppression recomputes it (not the identity)
__device__ __forceinline__ void nms_roundtrip(const int* b, int* o) {
    const double width = (double)(b[1] - b[0]), height = (double)(b[3] - b[2]);
    const double cx = __ddiv_rn(__dadd_rn((double)b[0], __ddiv_rn(width, 2.0)), 1000.0);
    const double cy = __ddiv_rn(__dadd_rn((double)b[2], __ddiv_rn(height, 2.0)), 1000.0);
    prop2abs_f64(cx, cy, __ddiv_rn(width, 1000.0), __ddiv_rn(height, 1000.0), o);
}

// IoU(+1) > 0.45 without a division: 20*inter > 9*union (exact for the integer areas of the 1000 grid)
__device__ __forceinline__ bool overlaps45(int x0, int x1, int y0, int y1, int area_i, int4 bj) {
    const int w = max(0, min(x1, bj.y) - max(x0, bj.x) + 1);
    const int h = max(0, min(y1, bj.w) - max(y0, bj.z) + 1);
    const int inter = w * h;
    const int uni = area_i + (bj.y - bj.x + 1) * (bj.w - bj.z + 1) - inter;
    return 20 * inter > 9 * uni;
}

// Greedy NMS of one class segment (boxes in descending confidence) by one wave.  Segments of up to 64 boxes
// live in registers: lane j holds box j, the pivot is broadcast with v_readlane, the survivors are a 64-bit
// mask; no memory traffic in the loop.  Longer segments walk the boxes in LDS / global.
// `al` is a plain LDS pointer on purpose: a `volatile` one degrades to a generic pointer whose byte stores compile to
// `flat_store_byte ... sc0 sc1` + `s_waitcnt vmcnt(0)` -- a system-scope store that took ~2 us EACH (three per wave here:
// 6.3 of this kernel's 13.7 us at batch 128, SSD_DETECT_STAMPS=1); only the long-segment walk below, where lanes read flags
// other lanes have just cleared, goes through a volatile alias.
__device__ __forceinline__ void nms_segment(const int4* nbox, unsigned char* al_plain, int s0, int len, int lane) {
    // The segment bounds come out of LDS, i.e. in vector registers: left there, the compiler treats the pivot loop as
    // divergent (exec-mask bookkeeping, 64-bit vector shifts for the survivor mask, a v_readfirstlane per v_readlane) --
    // ~420 cycles per pivot, 7.8 us of the 23 us this kernel took at batch 128 (phase clocks of workgroup 0, a measurement aid of round 3).  They are wave
    // uniform, so say so: the loop counter, the survivor mask and the pivot's coordinates then live in scalar registers.
    s0 = __builtin_amdgcn_readfirstlane(s0);
    len = __builtin_amdgcn_readfirstlane(len);
    if (len <= 64) {
        int4 me = make_int4(0, 0, 0, 0);
        if (lane < len) me = nbox[s0 + lane];
        // Phase A, no dependence between iterations, no branch: lane j collects the set of EARLIER boxes that overlap it
        // (bit i of lo / hi).  The greedy walk "pivot i suppresses what it overlaps, if it is still alive itself" spent
        // ~340 cycles per pivot on its serial chain (survivor mask -> branch -> 4 v_readlane -> compare -> ballot -> mask).
        unsigned lo = 0u, hi = 0u;
        const int n_lo = len < 32 ? len : 32;
        for (int i = 0; i < n_lo; ++i) {
            const int x0 = __builtin_amdgcn_readlane(me.x, i), x1 = __builtin_amdgcn_readlane(me.y, i);
            const int y0 = __builtin_amdgcn_readlane(me.z, i), y1 = __builtin_amdgcn_readlane(me.w, i);
            const bool ov = overlaps45(x0, x1, y0, y1, (x1 - x0 + 1) * (y1 - y0 + 1), me) & (lane > i);
            lo |= ov ? (1u << i) : 0u;
        }
        for (int i = 32; i < len; ++i) {
            const int x0 = __builtin_amdgcn_readlane(me.x, i), x1 = __builtin_amdgcn_readlane(me.y, i);
            const int y0 = __builtin_amdgcn_readlane(me.z, i), y1 = __builtin_amdgcn_readlane(me.w, i);
            const bool ov = overlaps45(x0, x1, y0, y1, (x1 - x0 + 1) * (y1 - y0 + 1), me) & (lane > i);
            hi |= ov ? (1u << (i - 32)) : 0u;
        }
        // Phase B: only boxes that overlap an earlier one can die.  In ascending order (the survivor bits below j are final
        // when j's turn comes): j dies iff one of its earlier overlappers is alive.
        u64 alive = len == 64 ? ~0ull : ((1ull << len) - 1ull);
        u64 cand = __ballot(((lo | hi) != 0u) & (lane < len));
        while (cand) {
            const int j = __builtin_ctzll(cand);
            cand &= cand - 1ull;
            const u64 killers = ((u64)(unsigned)__builtin_amdgcn_readlane((int)hi, j) << 32) | (u64)(unsigned)__builtin_amdgcn_readlane((int)lo, j);
            if (killers & alive) alive &= ~(1ull << j);
        }
        if (lane < len) al_plain[s0 + lane] = (alive >> lane) & 1ull ? 1 : 0;
        return;
    }
    volatile unsigned char* al = al_plain;
    for (int i = 0; i < len; ++i) {
        if (!al[s0 + i]) continue;
        const int4 bi = nbox[s0 + i];
        const int area_i = (bi.y - bi.x + 1) * (bi.w - bi.z + 1);
        for (int j = i + 1 + lane; j < len; j += 64) {
            if (!al[s0 + j]) continue;
            if (overlaps45(bi.x, bi.y, bi.z, bi.w, area_i, nbox[s0 + j])) al[s0 + j] = 0;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

constexpr int DET_THREADS = 512;
constexpr int DET_WAVES = DET_THREADS / 64;
constexpr int DET_FAST = 1024;           // up to this many candidates the whole image is handled in LDS
constexpr int DET_LDS_KEYS = 2048;       // general path: sort in LDS up to this many keys, else in (L2-resident) global
constexpr int DET_MAX_ALIVE = 32768;
constexpr int DET_MAX_SEGS = DET_MAX_ALIVE / SCAN_ROWS + 2;      // scan workgroups touching one image
constexpr int DET_SMEM = 50 * 1024;      // carved per path (general: 16 KB keys + 32 KB flags; fast: 49 KB)
static_assert(DET_SMEM >= DET_FAST * (8 + 8 + 16 + 16 + 1) && DET_SMEM >= DET_LDS_KEYS * 8 + DET_MAX_ALIVE, "LDS carve");

struct DetectArgs {
    int A, A2, nv, B;
    const double* anchors;
    const float* pred;
    int cap, max_out, out_cap, do_nms;
    const int* bcount;  // [scan workgroups][2] candidates per (workgroup, image) segment
    const u64* dense;   // [B*A] the segments' keys, compacted at each segment's first row
    u64* keys1;         // [B][A2] general path: the image's candidates gathered for the sort
    u64* keys2;
    int* box;       // [B][A][4]
    int* nbox;      // [B][A][4]
    DetectOut out;
};
// survivors in order -> the caller's arrays, clipped to [:max_out] / out_cap
template <typename KeyAt, typename BoxAt>
__device__ __forceinline__ void detect_emit(const DetectArgs& p, int b, int m, const unsigned char* alive, KeyAt key_at,
                                            BoxAt box_at, int* s_wtot) {
    const int tid = threadIdx.x;
    int limit = p.out_cap;
    if (p.max_out >= 0 && p.max_out < limit) limit = p.max_out;
    const int lane = tid & 63, wv = tid >> 6;
    int total = 0;
    for (int q0 = 0; q0 < m; q0 += DET_THREADS) {
        const int q = q0 + tid;
        const bool keep = q < m && alive[q];
        const u64 bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wtot[wv] = __popcll(bal);
        __syncthreads();
        int wbase = total, tot = 0;
#pragma unroll
        for (int i = 0; i < DET_WAVES; ++i) {
            if (i < wv) wbase += s_wtot[i];
            tot += s_wtot[i];
        }
        const int o = wbase + before;
        if (keep && o < limit) {
            const u64 key = key_at(q);
            const size_t dst = (size_t)b * p.out_cap + o;
            p.out.conf[dst] = __uint_as_float((unsigned)(key >> 32));
            p.out.cls[dst] = (int)(key & 31ull);
            p.out.idx[dst] = 32767 - (int)((key >> 8) & 0xFFFFull);
            *reinterpret_cast<int4*>(p.out.box + dst * 4) = box_at(q);
        }
        total += tot;
        __syncthreads();
    }
    if (tid == 0) p.out.count[b] = p.max_out >= 0 ? min(total, p.max_out) : total;
}

// The per-image phase: rank, decode, NMS and ordered emit of image b by one workgroup of DET_THREADS threads; `smem` = DET_SMEM
// bytes of the caller's LDS.  Called by detect_image_kernel (one workgroup per image).
__device__ __forceinline__ void detect_image_body(const DetectArgs& p, const int b, unsigned char* smem) {
    __shared__ int firstpos[32], crank[32], ccount[32], segstart[33], order_cls[32];
    __shared__ int s_npresent, s_wtot[DET_WAVES];
    __shared__ u64 cmax[32];
    __shared__ int bstart[32];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    u64* g1 = p.keys1 + (size_t)b * p.A2;
    // ---- the image's candidate segments (one per scan workgroup that touched the image) ----------------
    __shared__ int seg_off[DET_MAX_SEGS + 1], seg_base[DET_MAX_SEGS];
    __shared__ int s_pos[DET_FAST], s_cpos[DET_FAST];
    const int row_lo = b * p.A, row_hi = row_lo + p.A;
    const int first_blk = row_lo / SCAN_ROWS, nseg = (row_hi - 1) / SCAN_ROWS - first_blk + 1;
    if (tid < nseg) {
        const int blk = first_blk + tid;
        const bool tail_of_prev = blk * SCAN_ROWS < row_lo;      // the workgroup started in the previous image
        seg_base[tid] = tail_of_prev ? row_lo : blk * SCAN_ROWS;
        seg_off[tid + 1] = __hip_atomic_load(p.bcount + blk * 2 + (tail_of_prev ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (nseg <= 64) {                       // inclusive scan of the segment counts by one wave
        if (tid < 64) {
            int v = tid < nseg ? seg_off[tid + 1] : 0;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(v, o, 64);
                if (lane >= o) v += t;
            }
            if (tid < nseg) seg_off[tid + 1] = v;
            if (tid == 0) seg_off[0] = 0;
        }
    } else if (tid == 0) {
        int acc = 0;
        seg_off[0] = 0;
        for (int i = 1; i <= nseg; ++i) { acc += seg_off[i]; seg_off[i] = acc; }
    }
    __syncthreads();
    const int n = seg_off[nseg];
    auto candidate = [&](int f) {            // f-th candidate of the image (any order: the keys are unique and get sorted)
        int lo = 0, hi = nseg - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (seg_off[mid] <= f) lo = mid; else hi = mid - 1;
        }
        return __hip_atomic_load(p.dense + (size_t)seg_base[lo] + (f - seg_off[lo]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    if (n <= DET_FAST) {
        // ================= everything in LDS: rank sort, decode, NMS, emit =================
        u64* skey = reinterpret_cast<u64*>(smem);                       // candidates as listed
        u64* okey = skey + DET_FAST;                                    // candidates in output order
        int4* box = reinterpret_cast<int4*>(okey + DET_FAST);
        int4* nbox = box + DET_FAST;
        unsigned char* alive = reinterpret_cast<unsigned char*>(nbox + DET_FAST);      // (not volatile: see nms_segment)
        // every candidate's key, then (addressed by the key's anchor) its offsets and anchor: all global loads of the
        // workgroup are in flight together and land while the ranking sweep below runs.  Indices are clamped instead of
        // predicated: a select on a loaded value would make each load wait in turn.
        constexpr int PER = DET_FAST / DET_THREADS;
        u64 mykey[PER];
        float loc[PER][4];
        double anc[PER][4];
        int pos[PER], cpos[PER];
        if (n > 0) {
#pragma unroll
            for (int r = 0; r < PER; ++r) mykey[r] = candidate(min(tid + DET_THREADS * r, n - 1));
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int a = 32767 - (int)((mykey[r] >> 8) & 0xFFFFull);
                const float* lp = p.pred + ((size_t)b * p.A + a) * p.nv + (p.nv - 4);
                const double* ap = p.anchors + (size_t)a * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) { loc[r][e] = lp[e]; anc[r][e] = ap[e]; }
            }
        }
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int i = tid + DET_THREADS * r;
            if (i < n) skey[i] = mykey[r];
            else mykey[r] = ~0ull;
            s_pos[i] = 0; s_cpos[i] = 0;
        }
        if (tid < 32) { firstpos[tid] = INT_MAX; ccount[tid] = 0; }
        __syncthreads();
        const int m = p.cap >= 0 ? min(n, p.cap) : n;          // detections_cap (ssdutils.py:207-210)
        // The output order is: class groups in first-appearance order of the confidence-sorted list (defaultdict,
        // ssdutils.py:311-314) = classes by their best key, and inside a group by key.  When no cap bites (m == n, the
        // inference setting infer.py:233-234; training's cap of 200 only above 200 candidates) the GLOBAL rank of a
        // candidate is never needed: a counting sort by class and a rank inside the class' bucket (~n / 20 comparisons per
        // candidate instead of n) replace the n x n sweep, which was 9.6 of this kernel's 23 us at batch 128.
        const bool by_class = p.do_nms && m == n;
        if (by_class) {
            int myslot[PER];
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int i = tid + DET_THREADS * r;
                myslot[r] = i < n ? atomicAdd(&ccount[(int)(mykey[r] & 31ull)], 1) : 0;      // any order inside the bucket
                pos[r] = 0;
            }
            __syncthreads();
            if (tid < 64) {         // exclusive prefix of the class counts (raw class ids): where each bucket starts
                const int v = tid < 32 ? ccount[tid] : 0;
                int inc = v;
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) {
                    const int t = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += t;
                }
                if (tid < 32) bstart[tid] = inc - v;
            }
            __syncthreads();
            u64* bkey = reinterpret_cast<u64*>(nbox);       // (nbox is written by the placement phase, two barriers on)
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int i = tid + DET_THREADS * r;
                if (i < n) bkey[bstart[(int)(mykey[r] & 31ull)] + myslot[r]] = mykey[r];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int i = tid + DET_THREADS * r;
                cpos[r] = 0;
                if (i < n) {
                    const u64 key = mykey[r];
                    const int c = (int)(key & 31ull);
                    const u64* bk = bkey + bstart[c];
                    const int cnt = ccount[c];
                    int cp = 0;
                    for (int j = 0; j < cnt; ++j) cp += bk[j] > key ? 1 : 0;
                    cpos[r] = cp;
                    if (cp == 0) cmax[c] = key;      // the class' best key: one writer per class
                }
            }
            __syncthreads();
            if (tid < 64) {         // one lane per class: its rank among the present classes and where its segment starts
                const bool present = tid < 32 && ccount[tid & 31] > 0;
                const u64 mine = present ? cmax[tid & 31] : 0ull;
                int r = 0, start = 0;
                for (int c = 0; c < 32; ++c) {
                    const bool before = ccount[c] > 0 && cmax[c] > mine;
                    r += before ? 1 : 0;
                    start += before ? ccount[c] : 0;
                }
                const int np = __popcll(__ballot(present));
                if (tid < 32) crank[tid] = r;
                if (present) segstart[r] = start;
                if (tid == 0) { s_npresent = np; segstart[np] = m; }
            }
        } else {
        // Rank of every candidate among all (confidence descending, anchor ascending: keys are unique) and among those of
        // its own class: an n x n comparison, spread over ALL thread
        // slots -- with n candidates rounded up to n2 (a power of two) each candidate gets 1024 / n2 slots, each sweeping
        // its share of the list (every read an LDS broadcast); the partial counts meet in LDS.
        {
            const int n2 = next_pow2(n > 1 ? n : 1);
            const int lg = 31 - __builtin_clz(n2);
            const int nsweep = (DET_THREADS * PER) >> lg;
            const int jchunk = (n + nsweep - 1) / nsweep;
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int slot = tid + DET_THREADS * r;
                const int i = slot & (n2 - 1), sw = slot >> lg;
                if (i >= n) continue;
                const u64 ki = skey[i];
                const int j0 = sw * jchunk, j1 = min(n, j0 + jchunk);
                int gt = 0, cgt = 0;
#pragma unroll 4
                for (int j = j0; j < j1; ++j) {
                    const u64 kj = skey[j];
                    const bool g = kj > ki;
                    gt += g ? 1 : 0;
                    cgt += (g && ((kj ^ ki) & 31ull) == 0ull) ? 1 : 0;
                }
                if (gt) atomicAdd(&s_pos[i], gt);
                if (cgt) atomicAdd(&s_cpos[i], cgt);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int i = tid + DET_THREADS * r;
            pos[r] = s_pos[i]; cpos[r] = s_cpos[i];
        }
        // class groups in first-appearance order (defaultdict, ssdutils.py:311-314)
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int i = tid + DET_THREADS * r;
            if (i < n && pos[r] < m) {
                const int c = (int)(mykey[r] & 31ull);
                atomicMin(&firstpos[c], pos[r]);
                atomicAdd(&ccount[c], 1);
            }
        }
        __syncthreads();
        if (tid < 64) {         // one lane per class: its rank among the present classes and where its segment starts
            const int mine = tid < 32 ? firstpos[tid] : INT_MAX;
            int r = 0, start = 0;
            for (int c = 0; c < 32; ++c) {
                const bool before = firstpos[c] < mine;
                r += before ? 1 : 0;
                start += before ? ccount[c] : 0;
            }
            const int np = __popcll(__ballot(mine != INT_MAX));
            if (tid < 32) crank[tid] = r;
            if (p.do_nms && mine != INT_MAX) segstart[r] = start;
            if (tid == 0) {     // decode-only mode: one group in confidence order, no suppression
                s_npresent = p.do_nms ? np : 0;
                segstart[p.do_nms ? np : 0] = m;
            }
        }
        }
        __syncthreads();
        // place, decode
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int i = tid + DET_THREADS * r;
            if (i < n && pos[r] < m) {
                const u64 key = mykey[r];
                const int q = p.do_nms ? segstart[crank[(int)(key & 31ull)]] + cpos[r] : pos[r];
                const int a = 32767 - (int)((key >> 8) & 0xFFFFull);
                int bx[4], nb[4];
                decode_box(loc[r], anc[r], bx);
                nms_roundtrip(bx, nb);
                okey[q] = key;
                box[q] = make_int4(bx[0], bx[1], bx[2], bx[3]);
                nbox[q] = make_int4(nb[0], nb[1], nb[2], nb[3]);
                alive[q] = 1;
            }
        }
        __syncthreads();
        for (int r = wave; r < s_npresent; r += DET_WAVES) nms_segment(nbox, alive, segstart[r], segstart[r + 1] - segstart[r], lane);
        __syncthreads();
        detect_emit(p, b, m, alive, [&](int q) { return okey[q]; }, [&](int q) { return box[q]; }, s_wtot);
        return;
    }

    // ================= general path: bitonic sorts, boxes in (L2-resident) global memory =================
    u64* lkeys = reinterpret_cast<u64*>(smem);
    unsigned char* alive = smem + DET_LDS_KEYS * 8;
    u64* g2 = p.keys2 + (size_t)b * p.A2;
    int4* box = reinterpret_cast<int4*>(p.box + (size_t)b * p.A * 4);
    int4* nbox = reinterpret_cast<int4*>(p.nbox + (size_t)b * p.A * 4);

    // ---- sort 1: confidence descending, anchor ascending ------------------------------
    const int n2 = next_pow2(n > 1 ? n : 1);
    if (n2 <= DET_LDS_KEYS) {
        for (int i = tid; i < n2; i += DET_THREADS) lkeys[i] = i < n ? candidate(i) : 0ull;
        __syncthreads();
        bitonic_desc(lkeys, n2);
        for (int i = tid; i < n; i += DET_THREADS) g1[i] = lkeys[i];
    } else {
        for (int i = tid; i < n2; i += DET_THREADS) g1[i] = i < n ? candidate(i) : 0ull;
        __syncthreads();
        bitonic_desc(g1, n2);
    }
    __syncthreads();
    const int m = p.cap >= 0 ? min(n, p.cap) : n;          // detections_cap (ssdutils.py:207-210)

    // ---- class groups in first-appearance order (defaultdict, ssdutils.py:311-314) ----
    if (tid < 32) { firstpos[tid] = INT_MAX; ccount[tid] = 0; }
    __syncthreads();
    for (int i = tid; i < m; i += DET_THREADS) {
        const int c = (int)(g1[i] & 31ull);
        atomicMin(&firstpos[c], i);
        atomicAdd(&ccount[c], 1);
    }
    __syncthreads();
    if (tid < 32) {
        int r = 0;
        for (int c = 0; c < 32; ++c)
            if (firstpos[c] < firstpos[tid]) ++r;
        crank[tid] = (firstpos[tid] == INT_MAX) ? 31 : (p.do_nms ? r : 0);
    }
    __syncthreads();
    if (tid == 0) {
        int np = 0;
        if (p.do_nms) {     // decode-only mode: one group in confidence order, no suppression
            for (int c = 0; c < 32; ++c)
                if (firstpos[c] != INT_MAX) { order_cls[crank[c]] = c; ++np; }
            int acc = 0;
            for (int r = 0; r < np; ++r) { segstart[r] = acc; acc += ccount[order_cls[r]]; }
            segstart[np] = acc;
        }
        s_npresent = np;
    }
    __syncthreads();

    // ---- sort 2: (class rank, position) ascending == descending of the complement -----
    const int m2 = next_pow2(m > 1 ? m : 1);
    u64* k2 = m2 <= DET_LDS_KEYS ? lkeys : g2;
    for (int i = tid; i < m2; i += DET_THREADS) {
        u64 key = ~0ull;
        if (i < m) key = ((u64)crank[(int)(g1[i] & 31ull)] << 32) | (u64)i;
        k2[i] = ~key;
    }
    __syncthreads();
    bitonic_desc(k2, m2);

    // ---- decode every candidate in class-grouped order ----------------------------------
    for (int q = tid; q < m; q += DET_THREADS) {
        const int pos = (int)((~k2[q]) & 0xFFFFFFFFull);
        const u64 key = g1[pos];
        const int a = 32767 - (int)((key >> 8) & 0xFFFFull);
        int bx[4], nb[4];
        decode_box(p.pred + ((size_t)b * p.A + a) * p.nv + (p.nv - 4), p.anchors + (size_t)a * 4, bx);
        nms_roundtrip(bx, nb);
        box[q] = make_int4(bx[0], bx[1], bx[2], bx[3]);
        nbox[q] = make_int4(nb[0], nb[1], nb[2], nb[3]);
        if (q < DET_MAX_ALIVE) alive[q] = 1;
    }
    __syncthreads();

    // ---- greedy NMS: one wave per class segment ---
    for (int r = wave; r < s_npresent; r += DET_WAVES) nms_segment(nbox, alive, segstart[r], segstart[r + 1] - segstart[r], lane);
    __syncthreads();

    // ---- compact survivors in order; the caller's [:max_out] -------------------------------
    detect_emit(p, b, m, alive, [&](int q) { return g1[(int)((~k2[q]) & 0xFFFFFFFFull)]; }, [&](int q) { return box[q]; }, s_wtot);
}

__global__ __launch_bounds__(DET_THREADS) void detect_image_kernel(DetectArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[DET_SMEM];
    detect_image_body(p, blockIdx.x, smem);
}

// (Round 4 also built the pass as ONE launch -- scan workgroups taking a ticket per image, the last arriver running that image's
// per-image phase beside the scan: bit-identical and slower, 92 us against 17.6 + 26.6 us (profiles/r04_y_detect_fused_probe.txt:
// the scan half alone took 34 us in that form, and a chain of dependent loads beside an HBM-bound scan pays the loaded memory
// system's latency at every link).  Removed in round 5.)

static int pow2_ge(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

static size_t det_bcount_bytes(int B, int A) {      // bcount [scan workgroups][2]
    const size_t blocks = ((size_t)B * A + SCAN_ROWS - 1) / SCAN_ROWS;
    return (blocks * 8 + 255) / 256 * 256 + 256;
}
constexpr int DET_MAX_BATCH = 4096;
static size_t det_head_bytes(int B, int A) { return det_bcount_bytes(B, A); }

size_t detect_ws_bytes(int B, int A) {
    const size_t A2 = pow2_ge(A);
    return det_head_bytes(B, A) + ((size_t)B * A * 8 + 255) / 256 * 256 + 2 * (size_t)B * A2 * 8 + 2 * (size_t)B * A * 16;
}

void detect(int A, int num_classes, const double* anchors, const float* pred, int B, float conf_thr, int cap, int max_out,
            int out_cap, bool nms, const DetectOut& out, void* ws, hipStream_t s) {
    SSD_REQUIRE(A <= 32767 && A <= DET_MAX_ALIVE, "detect: at most 32767 anchors (got %d)", A);
    SSD_REQUIRE(num_classes >= 1 && num_classes <= 27, "detect: 1..27 classes");
    SSD_REQUIRE(A >= SCAN_ROWS, "detect: at least %d anchors", SCAN_ROWS);
    SSD_REQUIRE(out_cap >= 1, "detect: out_cap must be >= 1");
    SSD_REQUIRE((long long)B * A < (1LL << 31), "detect: batch * anchors must stay below 2^31");
    const int nv = num_classes + 5;
    const int A2 = pow2_ge(A);
    SSD_REQUIRE(B <= DET_MAX_BATCH, "detect: at most %d images per pass (got %d)", DET_MAX_BATCH, B);
    char* base = (char*)ws;
    int* bcount = (int*)base; base += det_head_bytes(B, A);
    u64* dense = (u64*)base; base += ((size_t)B * A * 8 + 255) / 256 * 256;
    u64* keys1 = (u64*)base; base += (size_t)B * A2 * 8;
    u64* keys2 = (u64*)base; base += (size_t)B * A2 * 8;
    int* box = (int*)base; base += (size_t)B * A * 16;
    int* nbox = (int*)base;
    const size_t rows = (size_t)B * A;
    const int blocks = (int)((rows + SCAN_ROWS - 1) / SCAN_ROWS);
    {
        ProfScope prof("detect_scan", 0.0, (double)rows * nv * 4.0, s);
        hipLaunchKernelGGL(detect_scan_kernel, dim3(blocks), dim3(256), (size_t)SCAN_ROWS * nv * sizeof(float), s, A, nv, B, pred,
                           conf_thr, dense, bcount);
    }
    DetectArgs a{};
    a.A = A; a.A2 = A2; a.nv = nv; a.B = B; a.anchors = anchors; a.pred = pred;
    a.cap = cap; a.max_out = max_out; a.out_cap = out_cap; a.do_nms = nms ? 1 : 0; a.bcount = bcount; a.dense = dense;
    a.keys1 = keys1; a.keys2 = keys2; a.box = box; a.nbox = nbox; a.out = out;
    {
        ProfScope prof("detect_image", 0.0, 0.0, s);
        hipLaunchKernelGGL(detect_image_kernel, dim3(B), dim3(DET_THREADS), 0, s, a);
    }
    HIP_OK(hipGetLastError());
}

// =================================================================================
// non_maximum_suppression / suppress_overlaps on an arbitrary box list (ssdutils.py:232-318)
// =================================================================================
// One workgroup: sort by (group ascending, confidence descending, input index ascending), greedy NMS per group
// (one wave per group, IoU as the reference's f64 quotient against an arbitrary threshold), survivors in order.
constexpr int NMSB_THREADS = 512;

__device__ __forceinline__ unsigned float_order_bits(float f) {      // monotone map float -> unsigned (any sign)
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(NMSB_THREADS) void nms_boxes_kernel(int n, int n2, int ngroups, const int4* __restrict__ boxes,
                                                                 const float* __restrict__ conf, const int* __restrict__ group,
                                                                 double thr, u64* __restrict__ keys, int* __restrict__ gstart,
                                                                 unsigned char* __restrict__ alive, int* __restrict__ keep) {
    __shared__ int s_wtot[NMSB_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n2; i += NMSB_THREADS) {
        u64 k = 0ull;        // padding sorts last
        if (i < n) k = ((u64)(65535 - group[i]) << 48) | ((u64)float_order_bits(conf[i]) << 16) | (u64)(65535 - i);
        keys[i] = k;
        if (i < n) alive[i] = 1;
    }
    for (int g = tid; g <= ngroups; g += NMSB_THREADS) gstart[g] = n;
    __syncthreads();
    bitonic_desc(keys, n2);
    // segment starts: first sorted position of every group (groups without boxes keep the next start)
    for (int q = tid; q < n; q += NMSB_THREADS) {
        const int g = 65535 - (int)(keys[q] >> 48);
        if (q == 0 || (65535 - (int)(keys[q - 1] >> 48)) != g) gstart[g] = q;
    }
    __syncthreads();
    if (tid == 0)
        for (int g = ngroups - 1; g >= 0; --g)
            if (gstart[g] == n) gstart[g] = gstart[g + 1];
    __syncthreads();
    volatile unsigned char* al = alive;
    for (int g = wave; g < ngroups; g += NMSB_THREADS / 64) {
        const int s0 = gstart[g], len = gstart[g + 1] - s0;
        for (int i = 0; i < len; ++i) {
            if (!al[s0 + i]) continue;
            const int4 bi = boxes[65535 - (int)(keys[s0 + i] & 0xFFFFull)];
            const long long area_i = (long long)(bi.y - bi.x + 1) * (bi.w - bi.z + 1);
            for (int j = i + 1 + lane; j < len; j += 64) {
                if (!al[s0 + j]) continue;
                const int4 bj = boxes[65535 - (int)(keys[s0 + j] & 0xFFFFull)];
                const long long w = max(0, min(bi.y, bj.y) - max(bi.x, bj.x) + 1);
                const long long h = max(0, min(bi.w, bj.w) - max(bi.z, bj.z) + 1);
                const long long inter = w * h;
                const long long uni = area_i + (long long)(bj.y - bj.x + 1) * (bj.w - bj.z + 1) - inter;
                // intersection / union as numpy divides two int64 arrays: IEEE f64 quotient (ssdutils.py:290-292)
                if (__ddiv_rn((double)inter, (double)uni) > thr) al[s0 + j] = 0;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    int total = 0;
    for (int q0 = 0; q0 < n; q0 += NMSB_THREADS) {
        const int q = q0 + tid;
        const bool kp = q < n && alive[q];
        const u64 bal = __ballot(kp);
        if (lane == 0) s_wtot[wave] = __popcll(bal);
        __syncthreads();
        int wbase = total, tot = 0;
#pragma unroll
        for (int i = 0; i < NMSB_THREADS / 64; ++i) {
            if (i < wave) wbase += s_wtot[i];
            tot += s_wtot[i];
        }
        if (kp) keep[1 + wbase + __popcll(bal & ((1ull << lane) - 1ull))] = 65535 - (int)(keys[q] & 0xFFFFull);
        total += tot;
        __syncthreads();
    }
    if (tid == 0) keep[0] = total;
}

size_t nms_boxes_ws_bytes(int n, int ngroups) {
    const size_t n2 = pow2_ge(n > 1 ? n : 1);
    return n2 * 8 + ((size_t)(ngroups + 2) * 4 + 15) / 16 * 16 + (size_t)n + 64;
}

void nms_boxes_device(int n, int ngroups, const int* boxes, const float* conf, const int* group, double thr, int* keep, void* ws,
                      hipStream_t s) {
    SSD_REQUIRE(n >= 1 && n <= 65535 && ngroups >= 1 && ngroups <= 65535, "nms_boxes: 1..65535 boxes and groups");
    const int n2 = pow2_ge(n);
    char* base = (char*)ws;
    u64* keys = (u64*)base; base += (size_t)n2 * 8;
    int* gstart = (int*)base; base += ((size_t)(ngroups + 2) * 4 + 15) / 16 * 16;
    unsigned char* alive = (unsigned char*)base;
    hipLaunchKernelGGL(nms_boxes_kernel, dim3(1), dim3(NMSB_THREADS), 0, s, n, n2, ngroups, reinterpret_cast<const int4*>(boxes), conf,
                       group, thr, keys, gstart, alive, keep);
    HIP_OK(hipGetLastError());
}

}  // namespace ssd
