// HBM-bound layers around the convolutions: pooling, L2 normalisation, the multibox
// head assembly + loss, the optimizer.  Host interface (all pointers are device memory).
#pragma once
#include "common.h"

namespace ssd {

// ---- max pooling, NHWC, TF SAME semantics (padded cells never win) ---------------
// Replaces the SavedModel's pool1-4 and tf.nn.max_pool at ssdvgg.py:234-236.
struct PoolDesc {
    int B, Hi, Wi, C, Ho, Wo, k, stride, pad_h, pad_w;
};
struct bf16_t;      // bf16.h; every tensor-typed op below exists for fp32 and for bf16 storage (fp32 math)
void maxpool_fwd(const PoolDesc& d, const float* x, float* y, hipStream_t s);
void maxpool_fwd(const PoolDesc& d, const bf16_t* x, bf16_t* y, hipStream_t s);
// dx[cell] = sum of dy over the windows whose FIRST maximum (scan order) is this cell.
// accumulate: += existing dx first; relu_mask: zero where x <= 0 (x is a relu output).
// ws: optional scratch of maxpool_bwd_ws_bytes() for overlapping windows (per-window argmax bytes);
// nullptr = recompute every window's maximum per cell.
size_t maxpool_bwd_ws_bytes(const PoolDesc& d);
void maxpool_bwd(const PoolDesc& d, const float* x, const float* dy, float* dx, bool accumulate, bool relu_mask,
                 void* ws, hipStream_t s);
void maxpool_bwd(const PoolDesc& d, const bf16_t* x, const bf16_t* dy, bf16_t* dx, bool accumulate, bool relu_mask,
                 void* ws, hipStream_t s);

// 3x3 stride-1 pooling whose forward pass keeps the windows' first-maximum taps in `arg` (maxpool_bwd_ws_bytes): backward from that
// record is maxpool_bwd's second pass alone
bool maxpool_arg_applicable(const PoolDesc& d);
void maxpool_fwd_arg(const PoolDesc& d, const float* x, float* y, void* arg, hipStream_t s);
void maxpool_fwd_arg(const PoolDesc& d, const bf16_t* x, bf16_t* y, void* arg, hipStream_t s);
void maxpool_bwd_arg(const PoolDesc& d, const float* x, const void* arg, const float* dy, float* dx, bool accumulate, bool relu_mask, hipStream_t s);
void maxpool_bwd_arg(const PoolDesc& d, const bf16_t* x, const void* arg, const bf16_t* dy, bf16_t* dx, bool accumulate, bool relu_mask, hipStream_t s);

// 2x2 stride-2 pooling with a forward-written record (per window and channel: first maximum's cell + "maximum is
// positive"): backward = record + dy -> dx without re-reading the pooled tensor.  For a pooled tensor with a single
// consumer (dx is overwritten, not accumulated); relu_mask as above.
bool maxpool_rec_applicable(const PoolDesc& d);
size_t maxpool_rec_bytes(const PoolDesc& d);
void maxpool_fwd_rec(const PoolDesc& d, const float* x, float* y, void* rec, hipStream_t s);
void maxpool_fwd_rec(const PoolDesc& d, const bf16_t* x, bf16_t* y, void* rec, hipStream_t s);
void maxpool_bwd_rec(const PoolDesc& d, const void* rec, const float* dy, float* dx, bool relu_mask, hipStream_t s);
void maxpool_bwd_rec(const PoolDesc& d, const void* rec, const bf16_t* dy, bf16_t* dx, bool relu_mask, hipStream_t s);

// ---- l2_normalization (ssdvgg.py:80-84): y = scale * x * rsqrt(max(sum_c x^2, 1e-12))
void l2norm_fwd(int npix, int C, const float* x, const float* scale, float* y, hipStream_t s);
void l2norm_fwd(int npix, int C, const bf16_t* x, const float* scale, bf16_t* y, hipStream_t s);
size_t l2norm_bwd_ws_floats(int npix, int C);
void l2norm_bwd(int npix, int C, const float* x, const float* scale, const float* dy, float* dx, float* dscale,
                float* ws, hipStream_t s);
void l2norm_bwd(int npix, int C, const bf16_t* x, const float* scale, const bf16_t* dy, bf16_t* dx, float* dscale,
                float* ws, hipStream_t s);

// ---- multibox heads: layout + softmax + loss (ssdvgg.py:353-372, 375-580) ----------
constexpr int MAX_MAPS = 8;
struct HeadLayout {
    int nmaps, A, nvars;             // nvars = num_classes + 5
    int hw[MAX_MAPS];                // cells per map
    int nj[MAX_MAPS];                // box types per map
    int ld[MAX_MAPS];                // row stride of the fused head buffer (nj*nvars rounded up to 8)
    int off[MAX_MAPS + 1];           // first anchor of each map
    float* buf[MAX_MAPS];            // [B*hw][ld] raw fused head conv outputs
    void* dbuf[MAX_MAPS];            // same shape, gradients: fp32, or bf16 when grad_bf16
    int grad_bf16;
};
// result[b][a][:] = (softmax(logits), loc) in the reference's anchor order
// (map -> box type -> row -> col, ssdvgg.py:63,365 == ssdutils.py:104-116).
void heads_result(const HeadLayout& L, int B, float* result, hipStream_t s);

struct LossWork {                    // per-step scratch, all device
    float* ce;                       // [B][A] cross entropy
    float* sl1;                      // [B][A] smooth-L1 summed over the 4 offsets
    unsigned char* pos;              // [B][A] 1 = positive anchor
    unsigned char* sel;              // [B][A] 1 = contributes to the confidence loss
    float* sample;                   // [B][4]: conf_b, loc_b, weight_b (1/(pos_n*B) or 0), pos_n
    float* partial;                  // [1024] sum-of-squares partials
    float* losses;                   // [4] total, localization, confidence, l2 (the executor points this at mapped host memory)
    unsigned* ticket;                // completion ticket of the per-sample workgroups (zero between launches)
};
size_t loss_work_bytes(int B, int A);
void loss_work_carve(LossWork& w, void* base, int B, int A);
// Forward of the loss; labels [B][A][nvars] device.  result must hold heads_result's output.
// bnorm: the batch size the per-sample losses are averaged over (<= 0: this step's own B).  A data-parallel
// caller whose shards are unequal passes global_samples / world, so that the mean over ranks is the global mean.
// A step may compute its loss in several launches over disjoint sample ranges (forward lanes): L.buf, result and labels
// point at the range's first sample, B = samples of this launch, b_off = index of its first sample, B_total = the step's.
void multibox_loss(const HeadLayout& L, int B, int b_off, int B_total, const float* result, const float* labels, LossWork& w,
                   float weight_decay, float bnorm, hipStream_t s);
// sum of squares of the filter region into w.partial (the l2 term of multibox_loss, which must follow it in stream
// order): 4 bytes per parameter, independent of the forward pass, so the step runs it beside the first layers
void l2_partials(const float* filters, size_t nfilters, LossWork& w, hipStream_t s);
// d(loss)/d(head outputs) written into L.dbuf (pad columns stay zero).
// (lane form like multibox_loss: L.dbuf, result and labels point at sample b_off, B samples)
void multibox_loss_grad(const HeadLayout& L, int B, int b_off, const float* result, const float* labels, const LossWork& w,
                        hipStream_t s);

// ---- MomentumOptimizer without Nesterov (ssdvgg.py:586-588): acc = m*acc + g; w -= lr*acc
void momentum_update(float* w, float* acc, const float* g, size_t n, float lr, float momentum, float gscale,
                     hipStream_t s);

// gradients of a step without samples (a data-parallel rank whose shard of a short last batch is empty)
void null_gradients(const float* w, float* g, size_t nfilters, size_t n, float wd, hipStream_t s);
// fp32 gradient range <-> bf16 message buffer of the data-parallel all-reduce (halves the bytes that cross xGMI)
void grads_to_bf16(const float* g, void* out, size_t n, hipStream_t s);
void grads_from_bf16(const void* in, float* g, size_t n, hipStream_t s);

void clock_monitor(unsigned* out, int nsamples, unsigned period, hipStream_t s);
void fill_zero(void* p, size_t bytes, hipStream_t s);

}  // namespace ssd
