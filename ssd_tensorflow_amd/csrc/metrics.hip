// VOC07 11-point average precision per class (average_precision.py:84-176) on gfx950.
// One workgroup per class: compact the class's detections, sort them by confidence (bitonic, L2
// resident), then ONE lane walks them in order -- the greedy matching of a detection against the
// still-unmatched ground truth of its image is inherently sequential -- keeping the running maximum
// precision for each of the 11 recall thresholds.  All arithmetic in IEEE f64: bit-exact with numpy.
#include "metrics.h"
#include <climits>

namespace ssd {

typedef unsigned long long u64;

struct APArgs {
    int n_det, n_gt, ncls, n2;
    const float* det_box;      // [n_det][4] xmin,xmax,ymin,ymax (the reference casts them to float32)
    const float* det_conf;
    const int* det_cls;
    const int* det_sample;
    const double* gt_box;      // [n_gt][4], rows grouped by sample id (ascending)
    const int* gt_cls;
    const int* gt_sample;
    double minoverlap;
    u64* keys;                 // [ncls][n2]
    unsigned char* matched;    // [n_gt], zeroed
    double* ap;                // [ncls]
    int* present;              // [ncls]: 1 if the class has ground truth
};

__device__ static void bitonic_desc_u64(u64* keys, int n2) {
    for (int k = 2; k <= n2; k <<= 1)
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

__global__ __launch_bounds__(256) void ap_class_kernel(APArgs p) {
    __shared__ int s_w[4], s_cnt;
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    u64* keys = p.keys + (size_t)k * p.n2;
    // ---- ground-truth count of this class -----------------------------------------------------
    int c = 0;
    for (int g = tid; g < p.n_gt; g += 256) c += p.gt_cls[g] == k;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) s_w[wv] = c;
    __syncthreads();
    const int count = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    if (count == 0) {
        if (tid == 0) { p.present[k] = 0; p.ap[k] = 0.0; }
        return;
    }
    // ---- this class's detections: key = order-preserving confidence bits << 32 | ~index --------
    int n = 0;
    for (int d0 = 0; d0 < p.n_det; d0 += 256) {
        const int d = d0 + tid;
        const bool mine = d < p.n_det && p.det_cls[d] == k;
        const u64 bal = __ballot(mine);
        if (lane == 0) s_w[wv] = __popcll(bal);
        __syncthreads();
        int base = n, tot = 0;
        for (int i = 0; i < 4; ++i) { if (i < wv) base += s_w[i]; tot += s_w[i]; }
        if (mine) {
            unsigned u = __float_as_uint(p.det_conf[d]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            keys[base + __popcll(bal & ((1ull << lane) - 1ull))] = ((u64)u << 32) | (u64)(0xFFFFFFFFu - (unsigned)d);
        }
        n += tot;
        __syncthreads();
    }
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (int i = n + tid; i < n2; i += 256) keys[i] = 0ull;
    __syncthreads();
    bitonic_desc_u64(keys, n2);
    if (tid != 0) return;
    // ---- greedy matching in confidence order (average_precision.py:130-161) -----------------------
    double tp = 0.0, fp = 0.0, best[11];
    bool any[11];
    for (int j = 0; j < 11; ++j) { best[j] = 0.0; any[j] = false; }
    for (int i = 0; i < n; ++i) {
        const int d = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
        const int s = p.det_sample[d];
        int lo = 0, hi = p.n_gt;                       // first ground-truth row of sample s
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (p.gt_sample[mid] < s) lo = mid + 1; else hi = mid; }
        const double b0 = p.det_box[d * 4], b1 = p.det_box[d * 4 + 1], b2 = p.det_box[d * 4 + 2], b3 = p.det_box[d * 4 + 3];
        const double areab = __dmul_rn(__dadd_rn(__dsub_rn(b1, b0), 1.0), __dadd_rn(__dsub_rn(b3, b2), 1.0));
        int arg = -1;
        double iou_max = 0.0;
        for (int g = lo; g < p.n_gt && p.gt_sample[g] == s; ++g) {
            if (p.gt_cls[g] != k) continue;
            const double a0 = p.gt_box[g * 4], a1 = p.gt_box[g * 4 + 1], a2 = p.gt_box[g * 4 + 2], a3 = p.gt_box[g * 4 + 3];
            const double areaa = __dmul_rn(__dadd_rn(__dsub_rn(a1, a0), 1.0), __dadd_rn(__dsub_rn(a3, a2), 1.0));
            const double w = fmax(0.0, __dadd_rn(__dsub_rn(fmin(b1, a1), fmax(b0, a0)), 1.0));
            const double h = fmax(0.0, __dadd_rn(__dsub_rn(fmin(b3, a3), fmax(b2, a2)), 1.0));
            const double inter = __dmul_rn(w, h);
            const double iou = __ddiv_rn(inter, __dsub_rn(__dadd_rn(areab, areaa), inter));
            if (arg < 0 || iou > iou_max) { iou_max = iou; arg = g; }    // np.argmax: first maximum
        }
        if (arg < 0 || iou_max < p.minoverlap || p.matched[arg]) fp += 1.0;
        else { tp += 1.0; p.matched[arg] = 1; }
        const double recall = __ddiv_rn(tp, (double)count);
        const double prec = __ddiv_rn(tp, __dadd_rn(tp, fp));
        for (int j = 0; j < 11; ++j)
            if (recall >= __dmul_rn((double)j, 0.1)) {                     // np.arange(0, 1.1, 0.1)[j]
                if (!any[j] || prec > best[j]) best[j] = prec;
                any[j] = true;
            }
    }
    double ap = 0.0;
    for (int j = 0; j < 11; ++j)
        if (any[j]) ap = __dadd_rn(ap, best[j]);
    p.ap[k] = __ddiv_rn(ap, 11.0);
    p.present[k] = 1;
}

void average_precision_device(int n_det, int n_gt, int ncls, const float* det_box, const float* det_conf, const int* det_cls,
                              const int* det_sample, const double* gt_box, const int* gt_cls, const int* gt_sample,
                              double minoverlap, u64* keys, int n2, unsigned char* matched, double* ap, int* present,
                              hipStream_t s) {
    APArgs a{n_det, n_gt, ncls, n2, det_box, det_conf, det_cls, det_sample, gt_box, gt_cls, gt_sample, minoverlap, keys, matched, ap, present};
    HIP_OK(hipMemsetAsync(matched, 0, n_gt > 0 ? n_gt : 1, s));
    hipLaunchKernelGGL(ap_class_kernel, dim3(ncls), dim3(256), 0, s, a);
    HIP_OK(hipGetLastError());
}

}  // namespace ssd
