// im2col-free direct convolution on gfx950 fp32 MFMA -- host interface.
// Replaces the TensorFlow kernels behind tf.nn.conv2d / atrous_conv2d / bias_add /
// relu at ssdvgg.py:48-50, 61-62, 260-262, 287-290 and their gradients.
#pragma once
#include <vector>
#include "common.h"

namespace ssd {

// NHWC activations, HWIO filters ([KH*KW][Ci][Co] row-major).  Every channel
// count is a multiple of 4 except Ci == 3 (conv1_1, scalar-gather path).
struct ConvDesc {
    int B, Hi, Wi, Ci;
    int Ho, Wo, Co;
    int KH, KW, stride, dil;
    int pad_h, pad_w;   // zero rows/cols BEFORE the image (TF SAME: total/2; VALID: 0)
};

void conv_fwd(const ConvDesc& d, const float* x, const float* w, const float* bias,
              float* y, bool relu, hipStream_t s);

// dx = conv^T(dy, w).  mask != nullptr: dx is zeroed where mask <= 0 (relu of the
// producer, mask has dx's shape).  accumulate: dx += (before the mask).
void conv_dgrad(const ConvDesc& d, const float* dy, const float* w, float* dx,
                const float* mask, bool accumulate, hipStream_t s);

// dw = x^T * dy (+ weight_decay * w), dbias = column sums of dy.
// Two stages: split-M partial slabs into ws, then a fixed-order reduce (deterministic).
size_t conv_wgrad_ws_floats(const ConvDesc& d);
void conv_wgrad(const ConvDesc& d, const float* x, const float* dy, float* dw, float* dbias,
                const float* w, float weight_decay, float* ws, hipStream_t s);

// ---- 2x2 stride-2 max-pool fused into its neighbours (round 5; ops.h maxpool_*_rec is the unfused form) ----
// Forward: y_pool[b][oh/2][ow/2][c] = max over the window of relu(conv + bias), first maximum in scan order wins, cells outside
// the image never (TF SAME, no leading padding); rec (may be nullptr) = the pool's 12-bit record per (window, 4 channels).  The
// convolution's own output is NOT written.  3x3 / stride 1 / SAME convolutions.  Bit-identical to conv_fwd + maxpool_fwd_rec.
bool conv_fwd_pool_supported(const ConvDesc& d);
void conv_fwd_pool(const ConvDesc& d, const float* x, const float* w, const float* bias, float* y_pool, void* rec, hipStream_t s);
// Backward: the data gradient of a conv whose INPUT is the pooled tensor, scattered through the pool's record into the
// [B, UH, UW, Ci] gradient of the pool's input: cell = the recorded first maximum and (relu of the producer) the maximum was
// positive ? dx : 0; every cell of dx_unpooled is written.  Bit-identical to conv_dgrad + maxpool_bwd_rec(relu_mask = true).
bool conv_dgrad_unpool_supported(const ConvDesc& d);
void conv_dgrad_unpool(const ConvDesc& d, const float* dy, const float* w, float* dx_unpooled, const void* rec, int UH, int UW,
                       hipStream_t s);

// dedicated fp32 first-layer forward (conv_first.hip): Ci*taps <= 32, Co == 64
bool conv_first_fwd_f32_applicable(const ConvDesc& d);
void conv_first_fwd_f32(const ConvDesc& d, const float* x, const float* w, const float* bias, float* y, bool relu, hipStream_t s);
// dedicated fp32 first-layer weight gradient (conv_first.hip): Ci*taps <= 30, Co == 64; operands straight from global memory
bool conv_first_wgrad_f32_applicable(const ConvDesc& d);
size_t conv_first_wgrad_f32_ws_floats(const ConvDesc& d);
void conv_first_wgrad_f32(const ConvDesc& d, const float* x, const float* dy, float* dw, float* dbias, const float* w,
                          float weight_decay, float* ws, hipStream_t s);

// ---- Round 6: Winograd F(4x4, 3x3) for the fp32 3x3 / stride 1 / SAME layers (winograd.hip) ------------------------------------------
// Transformed tensors are position-major [36][tiles][C]; *_ps = elements between two positions (so a forward lane can own a row range
// of a full-batch tensor).  U = the filter's transform [36][Ci][Co] (forward), Uflip = [36][Co][Ci] of the rotated filter (data gradient).
bool wino_applicable(const ConvDesc& d);
int wino_tiles(const ConvDesc& d);                         // B * ceil(H / 4) * ceil(W / 4) (dilation d: per residue class, see winograd.hip)
int wino_kpad(int c);                                      // c rounded up to the GEMMs' k granularity (32): row length of the transformed dy,
                                                           // rows per position of Uflip (pad rows / columns are zero; Uflip's are the caller's to clear once)
void wino_filter(const ConvDesc& d, const float* w, float* U, float* Uflip, hipStream_t s);      // either may be nullptr
// every Winograd layer's filter transforms in one launch per kind (the step runs them at the start of forward, beside conv1_x)
struct WinoFilterPlan {
    static constexpr int MAX = 24;
    struct Item {
        const float* w;
        float *U, *Uf;
        int Ci, Co, Cop, blk0;
    } it[MAX];
    int n = 0, blocks = 0;
    double elems = 0;
    void add(const float* w, float* U, float* Uf, int Ci, int Co);
};
void wino_filter_plan(const WinoFilterPlan& plan, bool forward, bool flipped, hipStream_t s);
size_t wino_fwd_ws_floats(const ConvDesc& d);              // Mws: 36 * tiles * Co
// y = relu?(conv(x) + bias); V [36][.][Ci] receives the input's transform (kept by a training step for wino_wgrad).  y_pool != nullptr:
// the fused 2x2 pool of conv_fwd_pool instead of y (same values, same record).
// relu_bits (optional, tiles * Ci / 4 64-bit words): which of x's values are positive, per tile and 4 channels -- what wino_dgrad of the
// SAME layer takes as mask_bits in place of a second read of x
void wino_fwd(const ConvDesc& d, const float* x, const float* U, const float* bias, float* y, bool relu, float* V, size_t v_ps,
              float* Mws, float* y_pool, void* pool_rec, hipStream_t s, void* relu_bits = nullptr);
// dy -> Yt (B^T dy B, the data gradient's operand) and / or Ya (A dy A^T, the weight gradient's), each [36][tiles][wino_kpad(Co)]; nullptr skips one
void wino_bwd_transform(const ConvDesc& d, const float* dy, float* Yt, float* Ya, hipStream_t s);
size_t wino_dgrad_ws_floats(const ConvDesc& d);            // Xws: 36 * tiles * Ci
// conv_dgrad's semantics (mask, accumulate) from the transformed dy; unpool_rec != nullptr: conv_dgrad_unpool's
void wino_dgrad(const ConvDesc& d, const float* Yt, const float* Uflip, float* dx, const float* mask, bool accumulate, float* Xws,
                const void* unpool_rec, int UH, int UW, hipStream_t s, const void* mask_bits = nullptr);
size_t wino_wgrad_ws_floats(const ConvDesc& d);
// conv_wgrad's semantics from the forward's V and the transformed dy
void wino_wgrad(const ConvDesc& d, const float* V, size_t v_ps, const float* Ya, float* dw, float* dbias, const float* w,
                float weight_decay, float* ws, hipStream_t s);

// ---- bf16 configuration (conv_bf16.hip): bf16 activations / gradients / filter mirrors, fp32 accumulate ----
struct bf16_t;

// conv1_1 (Ci = 3) stays on the fp32 kernels: fp32 image and master filter in, bf16 out / bf16 dy in
void conv_fwd_smallc_bf16out(const ConvDesc& d, const float* x, const float* w, const float* bias, bf16_t* y, bool relu,
                             hipStream_t s);
void conv_wgrad_smallc_bf16dy(const ConvDesc& d, const float* x, const bf16_t* dy, float* dw, float* dbias, const float* w,
                              float weight_decay, float* ws, hipStream_t s);

// dedicated first-layer kernels (conv_first_bf16.hip): Ci*taps <= 32 and Co == 64; image and filter are rounded to bf16
void conv_first_fwd_bf16(const ConvDesc& d, const float* x, const float* w, const float* bias, bf16_t* y, bool relu, hipStream_t s);
size_t conv_first_wgrad_bf16_ws_floats(const ConvDesc& d);
void conv_first_wgrad_bf16(const ConvDesc& d, const float* x, const bf16_t* dy, float* dw, float* dbias, const float* w,
                           float weight_decay, float* ws, hipStream_t s);

// One launch mirrors every layer's fp32 filter [tap][Ci][Co] as bf16 in the same order (io, the data
// gradient's operand) and transposed [tap][Co][Ci] (oi, the forward operand), at the same offsets.
struct FilterCastPlan {
    static constexpr int MAX_LAYERS = 48;
    struct Layer {
        size_t off;
        int taps, ci, co;
    } L[MAX_LAYERS];
    int n = 0;
    void add(size_t off, int taps, int ci, int co);
};
void cast_filters(const FilterCastPlan& plan, const float* w, bf16_t* io, bf16_t* oi, hipStream_t s);

// launches of the same shape the caller runs side by side on other streams (tile choice: conv_bf16.hip gather_rows_n64)
extern thread_local int g_conv_lanes;
// y: bf16 [B,Ho,Wo,Co], or fp32 when y_f32 (the multibox heads feed the fp32 loss)
void conv_fwd_bf16(const ConvDesc& d, const bf16_t* x, const bf16_t* w_oi, const float* bias, void* y, bool y_f32, bool relu,
                   hipStream_t s);
void conv_dgrad_bf16(const ConvDesc& d, const bf16_t* dy, const bf16_t* w_io, bf16_t* dx, const bf16_t* mask, bool accumulate,
                     hipStream_t s);
// the bf16 forms of the fused pool (above); forward: est. fraction of the tile rows that carry pixels is part of "supported"
bool conv_fwd_pool_bf16_supported(const ConvDesc& d);
void conv_fwd_pool_bf16(const ConvDesc& d, const bf16_t* x, const bf16_t* w_oi, const float* bias, bf16_t* y_pool, void* rec, hipStream_t s);
bool conv_dgrad_unpool_bf16_supported(const ConvDesc& d);
void conv_dgrad_unpool_bf16(const ConvDesc& d, const bf16_t* dy, const bf16_t* w_io, bf16_t* dx_unpooled, const void* rec, int UH, int UW,
                            hipStream_t s);
// Round 5: the data gradient of the 64 -> 64 layer d (conv1_2) with the WEIGHT gradient of the first layer d1 below it (conv1_1:
// 3 input channels, 3x3) computed from the dx tiles while they are in LDS: dx is never written, conv1_1's own weight-gradient
// kernel is not launched.  dw1 / dbias1 = sum over pixels (+ weight_decay * w1), reduced from one slab per workgroup in ws
// (conv_dgrad_first_wgrad_bf16_ws_floats).  mask = the first layer's output (relu mask of dx).
bool conv_dgrad_first_wgrad_bf16_applicable(const ConvDesc& d, const ConvDesc& d1);
size_t conv_dgrad_first_wgrad_bf16_ws_floats(const ConvDesc& d1);
void conv_dgrad_first_wgrad_bf16(const ConvDesc& d, const bf16_t* dy, const bf16_t* w_io, const bf16_t* mask, const ConvDesc& d1,
                                 const float* image, float* dw1, float* dbias1, const float* w1, float weight_decay, float* ws,
                                 hipStream_t s);
size_t conv_wgrad_bf16_ws_floats(const ConvDesc& d);
void conv_wgrad_bf16(const ConvDesc& d, const bf16_t* x, const bf16_t* dy, float* dw, float* dbias, const float* w,
                     float weight_decay, float* ws, hipStream_t s);

// ---- Round 6: the latency-bound tail (conv9_1 ... conv11_2 / conv12_2 and the small maps' heads) as one launch per direction ----
// tail_bf16.hip: a chain of convolution stages (forward, or data gradients in backward order), each image's chain walked by ONE
// workgroup with workgroup barriers between the stages -- nothing below the 10x10 map couples two images.  Pointers are those of
// the FIRST image of the launch; d.B is ignored (nimg images, [nimg][H][W][C] tensors).
struct TailStage {
    ConvDesc d;             // the convolution's geometry (forward sense), as conv_fwd_bf16 / conv_dgrad_bf16 take it
    bool dgrad;             // false: dst = relu?(conv(src) + bias);  true: dst (+)= conv^T(src), masked by `mask` > 0
    const void* src;        // forward: x [.][Hi][Wi][Ci];  data gradient: dy [.][Ho][Wo][Co]   (bf16)
    const void* wgt_packed; // the stage's filter in the chain kernel's fragment order: tail_chain_pack_filter of the [tap][Co][Ci] mirror
                            // (forward) or of the [tap][Ci][Co] mirror (data gradient); tail_chain_packed_elems bf16 elements
    const float* bias;      // forward
    const void* mask;       // data gradient: the relu mask tensor (dx's shape, bf16) or nullptr
    void* dst;              // forward: y (bf16, or fp32 when out_f32);  data gradient: dx (bf16)
    bool relu, accum, out_f32;
};
int tail_chain_max_stages();
bool tail_chain_stage_supported(const ConvDesc& d);
size_t tail_chain_packed_elems(const ConvDesc& d, bool dgrad);
struct TailPackItem {
    ConvDesc d;
    bool dgrad;             // false: `mirror` is the [tap][Co][Ci] image (forward stage); true: the [tap][Ci][Co] image (data-gradient stage)
    const void* mirror;
    void* packed;
};
void tail_chain_pack_filters(const TailPackItem* items, int n, hipStream_t s);      // one launch
void tail_chain_bf16(const TailStage* stages, int nstages, int nimg, const char* label, hipStream_t s);
// the weight gradients of several small layers in ONE launch, each with a single pixel split and the direct epilogue
// (dw = x^T dy + weight_decay * w, dbias = column sums of dy): no slabs, no reduce launches
struct WgradGroupItem {
    ConvDesc d;
    const bf16_t* x;
    const bf16_t* dy;
    float* dw;
    float* dbias;
    const float* w;
};
int conv_wgrad_group_bf16_max();
void conv_wgrad_group_bf16(const WgradGroupItem* items, int n, float weight_decay, hipStream_t s);

}  // namespace ssd
