// The SSD-VGG step executor: layer table, HBM arenas, forward / backward schedules.
// Mirrors the graph SSDVGG.build_from_vgg assembles (ssdvgg.py:96-118, 190-372) and the
// loss/optimizer of build_optimizer (ssdvgg.py:375-599).
#pragma once
#include "common.h"
#include "conv.h"
#include "ops.h"
#include "boxes.h"
#include "bf16.h"
#include <string>
#include <vector>
#include <map>

namespace ssd {

struct Tensor {
    std::string name;        // TF scope of the producing op ("conv4_3", "pool3", "norm_conv4_3", "head2")
    int H = 0, W = 0, C = 0;
    bool relu_out = false;   // produced by conv+relu: gradients written into it get the relu mask
    int consumers = 0;       // ops reading it in forward
    int done = 0;            // backward bookkeeping
    int gstream = 0;         // backward: stream class (0 main, 1 side) of the last kernel that wrote grad ...
    long gseq = 0;           // ... and its number in that class' issue order (net.hip bw_need)
    hipEvent_t gev = nullptr;    // optional: recorded behind every kernel that writes grad, so a reader on the other class waits for
    bool gev_set = false;        // THAT kernel instead of everything its class has been given (the small heads' feature maps)
    bool data_f32 = true;    // storage of data: fp32, or bf16 (bf16 configuration, every tensor but the image and the head outputs)
    bool grad_f32 = true;    // storage of grad
    void* data = nullptr;
    void* grad = nullptr;
    size_t per_image() const { return (size_t)H * W * C; }
    const float* f() const { return static_cast<const float*>(data); }
    float* gf() const { return static_cast<float*>(grad); }
    const bf16_t* h() const { return static_cast<const bf16_t*>(data); }
    bf16_t* gh() const { return static_cast<bf16_t*>(grad); }
};

enum OpKind { OP_CONV, OP_POOL, OP_L2NORM };

struct Op {
    OpKind kind;
    std::string name;        // TF variable scope for conv ops
    int in = -1, out = -1;   // tensor ids
    // conv
    int KH = 1, KW = 1, stride = 1, dil = 1, pad_h = 0, pad_w = 0;
    bool relu = true;
    size_t w_off = 0, b_off = 0;     // offsets (floats) into the arenas
    size_t ws_off = 0;               // this layer's region of the weight-gradient slab workspace (floats)
    // pool
    int k = 2;
    void* pool_rec = nullptr;        // forward-written argmax / sign record of a single-consumer 2x2 pool (ops.h)
    // pool fusion (net.hip plan_pool_fusion): a pool written by its producer's forward epilogue / whose backward is the scatter
    // in its consumer's data gradient; on conv ops: the pool behind this conv / the pool this conv's data gradient un-pools
    bool fused_fwd = false, fused_bwd = false;
    int pool_after = -1, unpool = -1;
    // head op: index of the feature map, else -1
    int head = -1;
    // Round 6 (fp32): the Winograd F(4x4, 3x3) form of this layer's passes (net.hip plan_winograd; conv.h wino_*): forward, data
    // gradient, weight gradient; the layer's filter transforms [36][Ci][Co] / [36][Co][Ci] and its input's transform [36][tiles][Ci]
    // (written by forward, read by the weight gradient)
    bool wino_f = false, wino_d = false, wino_w = false;
    float *wino_U = nullptr, *wino_Uf = nullptr, *wino_V = nullptr;
    void* wino_bits = nullptr;       // the relu mask of the layer's input as bits (forward's input transform -> the data gradient's output transform)
    long long wino_bits_step = -1;   // the forward pass that wrote them (Net::fwd_serial_)
    long long wino_v_step = -1;      // the forward pass that wrote wino_V
};

struct Variable {
    std::string name;        // reference TF variable name
    int ndim;
    int shape[4];
    // location inside the arena: `rows` rows of `width` floats, `pitch` apart, from `off`
    size_t off, rows, width, pitch;
    size_t count() const { return rows * width; }
};

// one output slot of the decode + NMS pass: packed [count | conf | cls | idx | box] in HBM + pinned host mirror
struct DetectSlot {
    char* dev = nullptr;
    char* host = nullptr;
    size_t bytes = 0, used = 0;
    int b = 0, out_cap = 0;
    bool mapped = false;            // dev IS the device's view of the pinned host buffer (no HBM copy, no transfer)
    hipEvent_t ready = nullptr;     // recorded behind the pass (behind the device-to-host copy when !mapped)
};

class Net {
public:
    // dtype 0: fp32 everywhere (BASELINE.json configs[1]); 1: bf16 activations / gradients / filter mirrors with fp32
    // master weights, fp32 accumulation and fp32 loss (configs[2])
    Net(const char* preset, int num_classes, int max_batch, int device, bool training, unsigned long long seed,
        float* ext_params, float* ext_grads, float* ext_momentum, int dtype = 0);
    ~Net();

    static size_t arena_floats(const char* preset, int num_classes);

    void set_stream(hipStream_t s) { stream_ = s; }
    hipStream_t stream() const { return stream_; }

    // steps; x/y device pointers
    void forward(const float* x, int b, bool train_mode, const float* y);
    void backward(int b, const float* y);
    void backward_begin(int b, const float* y);
    // runs reverse ops until >= min_floats of filter gradients are newly final; [off, off+count) is that range
    bool backward_step(size_t min_floats, size_t* off, size_t* count, bool sync_main);
    // the (offset, count) ranges backward_step(min_floats, ...) reports, in order, without running anything: a rank whose
    // shard is empty issues the collectives of the ranks that do run backward (parallel.train_step_dp)
    std::vector<std::pair<size_t, size_t>> backward_ranges(size_t min_floats) const;
    void set_wgrad_stream(hipStream_t s);      // caller-owned side stream for the weight gradients
    void apply_gradients(float grad_scale);
    void backward_apply(int b, const float* y, float grad_scale);      // backward + update, the optimizer overlapped with backward's tail
    void set_loss_normalizer(float bnorm) { loss_bnorm_ = bnorm; }      // <= 0: every step's own batch size
    void null_gradients_step();                                         // gradient arena of a step without samples
    void set_optimizer(const float* lr_values, const long long* bounds, int n, float momentum, float wd);

    // host-buffer conveniences
    void upload_xy(const float* x, const float* y, int b);
    const float* x_stage() const { return x_stage_; }
    const float* y_stage() const { return y_stage_; }

    void get_losses(float out[4]);                       // the last step's; waits for everything on the stream
    // the losses of the last step (steps_back = 0) or of one of the LOSS_RING - 1 steps before it: waits for THAT step's
    // forward only, so a caller can book step k - 1 while step k runs (train.py)
    void get_losses_step(int steps_back, float out[4]);
    void copy_result(float* out, int b);
    void set_result(const float* pred_dev, int b);

    const std::vector<Variable>& variables() const { return vars_; }
    void load_variable(const char* name, const float* host, size_t count, int which);   // 0 params, 2 momentum
    void save_variable(const char* name, float* host, size_t count, int which);         // 0 params, 1 grads, 2 momentum
    void activation(const char* name, int b, float* out, size_t count);
    void activation_shape(const char* name, int* H, int* W, int* C) const;
    void pool_fusion(int* out, int cap, int* count) const;      // per 2x2 stride-2 pool in graph order: bit 0 fused forward, bit 1 fused backward

    void detect_last(int b, float thr, int cap, int max_out, int out_cap, bool nms, int* count, float* conf, int* cls, int* idx, int* box);
    // asynchronous form: kernels + one device-to-host copy enqueued; dev_out (optional) = the HBM arrays of the slot
    const DetectSlot& detect_last_dev(int b, float thr, int cap, int max_out, int out_cap, bool nms, DetectOut* dev_out);
    // host copy of the latest (which = 0) or the previous (which = 1) pass; waits for that slot's copy only
    void detect_fetch(int which, int* count, float* conf, int* cls, int* idx, int* box);
    // the pinned host mirror of that slot itself (valid until the second-next pass); waits for the slot's copy only
    void detect_host(int which, DetectOut* host, int* b, int* out_cap);

    const Preset& preset() const { return *preset_; }
    int num_classes() const { return C_; }
    int nvars() const { return C_ + 5; }
    int max_batch() const { return Bmax_; }
    bool training() const { return training_; }
    int dtype() const { return bf16_ ? 1 : 0; }
    int device() const { return device_; }
    float* params() { return params_; }
    float* grads() { return grads_; }
    float* momentum() { return mom_; }
    size_t nparams() const { return nparams_; }
    size_t nfilters() const { return nfilters_; }
    float* result() { return result_; }
    long long global_step = 0;
    Profiler& profiler() { return prof_; }
    void set_overlap(bool on) { overlap_ = on; }

private:
    int add_tensor(const std::string& name, int H, int W, int C, bool relu_out);
    void build_graph();
    void alloc();
    void init_weights(unsigned long long seed);
    ConvDesc conv_desc(const Op& op, int b) const;
    float current_lr() const;
    const Variable& find_var(const char* name) const;
    void* dalloc(size_t bytes);

    const Preset* preset_;
    int C_, Bmax_, device_;
    bool training_;
    bool bf16_ = false;
    bf16_t *wq_io_ = nullptr, *wq_oi_ = nullptr;     // bf16 mirrors of the filter region: [tap][Ci][Co] and [tap][Co][Ci]
    FilterCastPlan cast_plan_;
    hipStream_t stream_ = nullptr;
    hipStream_t wstream_ = nullptr;        // side stream of the weight gradients (SSD_OVERLAP_WGRAD=0 disables)
    hipEvent_t ev_dy_ = nullptr, ev_w_ = nullptr;
    hipStream_t hstream_ = nullptr;        // side stream of the multibox heads in forward
    hipEvent_t ev_h_ = nullptr, ev_cast_ = nullptr, ev_fmap_[MAX_MAPS] = {};
    hipStream_t s2_ = nullptr;                  // second forward lane: its main stream, which also runs its heads (= wstream_ in a training handle)
    hipEvent_t ev2_h_ = nullptr, ev_l2_ = nullptr, ev_join_ = nullptr, ev2_fmap_[MAX_MAPS] = {};
    int tail_first_ = 0;                 // op index of conv8_1: the extra layers behind it form backward's side chain
    bool stop_events_ = true;            // data gradients carry their gradient tensor's event (common.h g_stop_event)
    // Issue orders (net.hip build_orders): forward walks fwd_order_ (every multibox head right behind its feature map), backward
    // walks bwd_order_ (reverse graph order)
    std::vector<int> fwd_order_, bwd_order_;
    std::vector<char> bw_conv_done_;     // backward: conv ops processed so far (indexed like ops_)
    // backward stream classes: 0 = a lane's main stream, 1 = its side stream.  bw_issued_[c] counts the kernels class c has been
    // given that write a gradient, bw_seen_[x][y] is the count of class y that class x has already been made to wait for
    long bw_issued_[2] = {0, 0}, bw_seen_[2][2] = {{0, 0}, {0, 0}};
    hipEvent_t ev_m2s_ = nullptr;        // main -> side hand-off
    bool bw_first_on_main_ = false;      // the first layer's weight gradient was issued on the main stream (backward_step)
    bool fuse_first_wgrad_ = false;      // bf16: conv1_1's weight gradient inside conv1_2's data gradient where the kernel applies (plan_pool_fusion)
    bool bw_first_fused_ = false;        // ... and this backward pass took it
    int fused_first_out_ = -1;           // the tensor whose gradient that pass did not materialise (conv1_1's output)
    void launch_wgrad(int op_index, int b, hipStream_t ws);
    // Round 6 (bf16): the latency-bound tail as one launch per direction (conv.h TailStage; net.hip plan_tail_chain).  chain_first_ =
    // op index of the first 1x1 layer below the 10x10 map (vgg300: conv9_1, vgg512: conv10_1), -1: off; in_chain_[op] marks the
    // extra layers from there on and the multibox heads that read them.
    int chain_first_ = -1;
    std::vector<char> in_chain_;
    bf16_t* wq_tail_ = nullptr;                          // the chain layers' filters in the chain kernel's fragment order (conv.h tail_chain_pack_filters) ...
    std::vector<long long> tail_fwd_off_, tail_bwd_off_; // ... element offsets per op of the forward / data-gradient form (-1: none)
    void pack_tail_filters(hipStream_t s);               // from the fresh bf16 mirrors, one launch (forward, behind cast_filters)
    bool chain_fwd_ = false, chain_bwd_ = false;      // SSD_TAIL_FUSE bit 0 / bit 1: the chain in forward / in backward
    std::vector<char> in_wgroup_;        // ops whose weight gradient came out of this backward pass' grouped launch
    bool bw_chain_done_ = false;         // this backward pass has issued the chain's data gradients and grouped weight gradients
    void plan_tail_chain();
    // Round 6 (fp32): Winograd layers.  Scratch shared by every such layer: the GEMM results of a forward lane (wino_m_), the
    // transformed dy / dx of the data gradient (main stream: wino_yt_, wino_xw_), the transformed dy and the slabs of the weight
    // gradient (weight-gradient stream: wino_ya_, wino_slab_).  One stream runs each kind in order, so one buffer per kind suffices.
    void plan_winograd();
    long long fwd_serial_ = 0;           // forward passes so far
    WinoFilterPlan wino_plan_, wino_plan_first_, wino_plan_flip_;      // forward transforms of all layers but the first / of the first / flipped ones
    bool wino_flip_pending_ = false;
    bool wino_any_f_ = false, wino_any_d_ = false;
    float *wino_m_[3] = {nullptr, nullptr, nullptr}, *wino_vs_[3] = {nullptr, nullptr, nullptr}, *wino_yt_ = nullptr, *wino_xw_ = nullptr, *wino_ya_ = nullptr, *wino_slab_ = nullptr;
    hipEvent_t ev_wino_ = nullptr, ev_wino_first_ = nullptr, ev_wino_flip_ = nullptr;
    void launch_tail_forward(int b0, int nb, hipStream_t s);
    void launch_tail_backward(int b, bool* side_used);
    void build_orders();
    void plan_pool_fusion();
    int bw_class(const Op& op, int op_index) const;
    void bw_sync(int x, int y);          // class x waits for everything class y has been given so far
    void bw_need(int x, const Tensor& t);
    void bw_wrote(int x, Tensor& t, bool carried = false);
    size_t bw_final_lo() const;          // lowest arena offset such that every filter at or above it has its final gradient
    bool overlap_ = true;
    bool own_wstream_ = true;
    bool s2_is_w_ = false;                 // the second forward lane's stream IS the weight-gradient stream (four streams in all)

    std::vector<Tensor> tensors_;
    std::vector<Op> ops_;
    std::vector<Variable> vars_;
    int input_t_ = -1;
    std::vector<int> head_t_;             // head output tensors per map
    HeadLayout heads_{};

    size_t nparams_ = 0, nfilters_ = 0, scale_off_ = 0;
    float *params_ = nullptr, *grads_ = nullptr, *mom_ = nullptr;
    bool own_params_ = false, own_grads_ = false, own_mom_ = false;

    float* result_ = nullptr;
    float *x_stage_ = nullptr, *y_stage_ = nullptr;
    float* wgrad_ws_ = nullptr;
    float* l2_ws_ = nullptr;
    void* pool_ws_ = nullptr;
    int pool_arg_op_ = -1;               // the 3x3 stride-1 pool whose first-maximum taps the last training forward left in pool_ws_
    void* loss_ws_ = nullptr;
    LossWork lw_{};
    static constexpr int LOSS_RING = 4;
    float* losses_host_ = nullptr;        // pinned, device-mapped: LOSS_RING slots of 4 floats, slot = loss_seq_ % LOSS_RING
    float* losses_dev_ = nullptr;         // the device's view of the same memory
    hipEvent_t ev_loss_[LOSS_RING] = {};  // recorded behind the forward pass that wrote the slot
    long long loss_seq_ = -1;             // forward passes with a loss so far - 1
    float* begin_loss_slot();             // next slot: waits until its previous use (LOSS_RING passes ago) has finished
    double* anchors_dev_ = nullptr;
    int* anchors_abs_dev_ = nullptr;
    void* detect_ws_ = nullptr;
    DetectSlot det_slot_[2];
    int det_cur_ = 0;
    void detect_slot_carve(const DetectSlot& sl, DetectOut& d, char* base) const;
    std::vector<void*> allocs_;

    std::vector<float> lr_values_{0.001f};
    std::vector<long long> lr_bounds_;
    float momentum_ = 0.9f, wd_ = 0.0005f;
    float loss_bnorm_ = 0.f;
    Profiler prof_;
    int bw_pos_ = 0, bw_b_ = 0;          // next entry of bwd_order_
    size_t bw_done_off_ = 0;
};

}  // namespace ssd
