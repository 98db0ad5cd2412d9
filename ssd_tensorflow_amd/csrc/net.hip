// SSD-VGG step executor for gfx950.  See net.h.
#include "net.h"
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <mutex>

namespace ssd {

static bool first_layer_kernel(const ConvDesc& d) { return d.Ci * d.KH * d.KW <= 32 && d.Co == 64; }

// fused head width: a multiple of 8 channels = whole 16-byte pieces in fp32 and in bf16 rows
static int round8(int n) { return (n + 7) / 8 * 8; }

// ---------------------------------------------------------------------------------
// graph (ssdvgg.py:190-372)
// ---------------------------------------------------------------------------------
int Net::add_tensor(const std::string& name, int H, int W, int C, bool relu_out) {
    Tensor t;
    t.name = name; t.H = H; t.W = W; t.C = C; t.relu_out = relu_out;
    tensors_.push_back(t);
    return (int)tensors_.size() - 1;
}

enum PadMode { PAD_SAME, PAD_VALID, PAD_BR1_VALID };

void Net::build_graph() {
    const Preset& p = *preset_;
    const int nv = C_ + 5;
    input_t_ = add_tensor("image_input", p.image_h, p.image_w, 3, false);
    int cur = input_t_;

    auto conv = [&](const std::string& name, int cout, int k, int stride, PadMode pm, int dil, bool relu, int head,
                    int from) -> int {
        const Tensor in = tensors_[from];
        Op op;
        op.kind = OP_CONV; op.name = name; op.in = from; op.KH = op.KW = k; op.stride = stride; op.dil = dil;
        op.relu = relu; op.head = head;
        int Ho, Wo;
        if (pm == PAD_SAME) {
            tf_same(in.H, k, stride, dil, &op.pad_h, &Ho);
            tf_same(in.W, k, stride, dil, &op.pad_w, &Wo);
        } else {
            const int keff = (k - 1) * dil + 1;
            const int extra = pm == PAD_BR1_VALID ? 1 : 0;     // tf.pad bottom/right +1 (ssdvgg.py:328-329)
            op.pad_h = op.pad_w = 0;
            Ho = (in.H + extra - keff) / stride + 1;
            Wo = (in.W + extra - keff) / stride + 1;
        }
        op.out = add_tensor(head >= 0 ? "head" + std::to_string(head) : name, Ho, Wo, cout, relu);
        tensors_[from].consumers++;
        ops_.push_back(op);
        return op.out;
    };
    auto pool = [&](const std::string& name, int k, int stride, int from) -> int {
        const Tensor in = tensors_[from];
        Op op;
        op.kind = OP_POOL; op.name = name; op.in = from; op.k = k; op.stride = stride;
        int Ho, Wo;
        tf_same(in.H, k, stride, 1, &op.pad_h, &Ho);
        tf_same(in.W, k, stride, 1, &op.pad_w, &Wo);
        op.out = add_tensor(name, Ho, Wo, in.C, false);
        tensors_[from].consumers++;
        ops_.push_back(op);
        return op.out;
    };

    // VGG-16 trunk (external SavedModel in the reference; ssdvgg.py:195-207)
    cur = conv("conv1_1", 64, 3, 1, PAD_SAME, 1, true, -1, cur);
    cur = conv("conv1_2", 64, 3, 1, PAD_SAME, 1, true, -1, cur);
    cur = pool("pool1", 2, 2, cur);
    cur = conv("conv2_1", 128, 3, 1, PAD_SAME, 1, true, -1, cur);
    cur = conv("conv2_2", 128, 3, 1, PAD_SAME, 1, true, -1, cur);
    cur = pool("pool2", 2, 2, cur);
    cur = conv("conv3_1", 256, 3, 1, PAD_SAME, 1, true, -1, cur);
    cur = conv("conv3_2", 256, 3, 1, PAD_SAME, 1, true, -1, cur);
    cur = conv("conv3_3", 256, 3, 1, PAD_SAME, 1, true, -1, cur);
    cur = pool("pool3", 2, 2, cur);
    cur = conv("conv4_1", 512, 3, 1, PAD_SAME, 1, true, -1, cur);
    cur = conv("conv4_2", 512, 3, 1, PAD_SAME, 1, true, -1, cur);
    const int c43 = cur = conv("conv4_3", 512, 3, 1, PAD_SAME, 1, true, -1, cur);
    cur = pool("pool4", 2, 2, cur);
    cur = conv("conv5_1", 512, 3, 1, PAD_SAME, 1, true, -1, cur);
    cur = conv("conv5_2", 512, 3, 1, PAD_SAME, 1, true, -1, cur);
    cur = conv("conv5_3", 512, 3, 1, PAD_SAME, 1, true, -1, cur);
    // a-trous modifications (ssdvgg.py:231-292)
    cur = pool("mod_pool5", 3, 1, cur);
    cur = conv("mod_conv6", 1024, 3, 1, PAD_SAME, 6, true, -1, cur);
    const int c7 = cur = conv("mod_conv7", 1024, 1, 1, PAD_SAME, 1, true, -1, cur);
    // extra feature layers (ssdvgg.py:300-332)
    const bool big = p.nmaps >= 7;
    std::vector<int> fmaps{-1, c7};
    cur = conv("conv8_1", 256, 1, 1, PAD_SAME, 1, true, -1, cur);
    cur = conv("conv8_2", 512, 3, 2, PAD_SAME, 1, true, -1, cur); fmaps.push_back(cur);
    cur = conv("conv9_1", 128, 1, 1, PAD_SAME, 1, true, -1, cur);
    cur = conv("conv9_2", 256, 3, 2, PAD_SAME, 1, true, -1, cur); fmaps.push_back(cur);
    cur = conv("conv10_1", 128, 1, 1, PAD_SAME, 1, true, -1, cur);
    cur = conv("conv10_2", 256, 3, big ? 2 : 1, big ? PAD_SAME : PAD_VALID, 1, true, -1, cur); fmaps.push_back(cur);
    cur = conv("conv11_1", 128, 1, 1, PAD_SAME, 1, true, -1, cur);
    cur = conv("conv11_2", 256, 3, 1, PAD_VALID, 1, true, -1, cur); fmaps.push_back(cur);
    if (big) {
        cur = conv("conv12_1", 128, 1, 1, PAD_SAME, 1, true, -1, cur);
        cur = conv("conv12_2", 256, 3, 1, PAD_BR1_VALID, 1, true, -1, cur); fmaps.push_back(cur);
    }
    // l2 norm on conv4_3 (ssdvgg.py:335-337), then the fused multibox heads (ssdvgg.py:353-365)
    {
        Op op;
        op.kind = OP_L2NORM; op.name = "l2_norm_conv4_3"; op.in = c43;
        op.out = add_tensor("norm_conv4_3", tensors_[c43].H, tensors_[c43].W, tensors_[c43].C, false);
        tensors_[c43].consumers++;
        ops_.push_back(op);
        fmaps[0] = op.out;
    }
    SSD_REQUIRE((int)fmaps.size() == p.nmaps, "feature map count mismatch");
    heads_ = HeadLayout{};
    heads_.nmaps = p.nmaps; heads_.A = p.num_anchors; heads_.nvars = nv;
    for (int i = 0; i < p.nmaps; ++i) {
        SSD_REQUIRE(tensors_[fmaps[i]].H == p.map_size[i], "feature map %d is %d, preset says %d", i, tensors_[fmaps[i]].H,
                    p.map_size[i]);
        const int co = round8(p.ntypes[i] * nv);
        const int t = conv("heads/map" + std::to_string(i), co, 3, 1, PAD_SAME, 1, false, i, fmaps[i]);
        head_t_.push_back(t);
        heads_.hw[i] = p.map_size[i] * p.map_size[i];
        heads_.nj[i] = p.ntypes[i];
        heads_.ld[i] = co;
        heads_.off[i] = p.off[i];
    }
    heads_.off[p.nmaps] = p.off[p.nmaps];
    heads_.grad_bf16 = bf16_ ? 1 : 0;
    if (bf16_) {
        for (size_t i = 0; i < tensors_.size(); ++i) tensors_[i].data_f32 = tensors_[i].grad_f32 = false;
        tensors_[input_t_].data_f32 = true;                 // the image is consumed as fed (fp32)
        for (int t : head_t_) tensors_[t].data_f32 = true;   // the loss reads fp32 logits / offsets
    }

    tail_first_ = (int)ops_.size();
    for (size_t i = 0; i < ops_.size(); ++i)
        if (ops_[i].name == "conv8_1") tail_first_ = (int)i;
    build_orders();
    // ---- arena layout: all filters (forward order), all biases, the l2-norm scale ----
    size_t off = 0;
    for (auto& op : ops_)
        if (op.kind == OP_CONV) {
            op.w_off = off;
            off += (size_t)op.KH * op.KW * tensors_[op.in].C * tensors_[op.out].C;
        }
    nfilters_ = off;
    for (auto& op : ops_)
        if (op.kind == OP_CONV) cast_plan_.add(op.w_off, op.KH * op.KW, tensors_[op.in].C, tensors_[op.out].C);
    for (auto& op : ops_)
        if (op.kind == OP_CONV) {
            op.b_off = off;
            off += tensors_[op.out].C;
        }
    scale_off_ = off;
    off += 512;
    nparams_ = off;

    // ---- variables under the reference's names ------------------------------------------
    auto add_var = [&](const std::string& n, int nd, int s0, int s1, int s2, int s3, size_t o, size_t rows, size_t width,
                       size_t pitch) {
        Variable v;
        v.name = n; v.ndim = nd; v.shape[0] = s0; v.shape[1] = s1; v.shape[2] = s2; v.shape[3] = s3;
        v.off = o; v.rows = rows; v.width = width; v.pitch = pitch;
        vars_.push_back(v);
    };
    for (auto& op : ops_) {
        if (op.kind != OP_CONV) continue;
        const int ci = tensors_[op.in].C, co = tensors_[op.out].C;
        if (op.head < 0) {
            const size_t n = (size_t)op.KH * op.KW * ci * co;
            add_var(op.name + "/filter", 4, op.KH, op.KW, ci, co, op.w_off, 1, n, n);
            add_var(op.name + "/biases", 1, co, 0, 0, 0, op.b_off, 1, co, co);
        } else {
            for (int j = 0; j < p.ntypes[op.head]; ++j) {
                const std::string base = "classifiers/classifier" + std::to_string(op.head) + "_" + std::to_string(j);
                add_var(base + "/filter", 4, 3, 3, ci, nv, op.w_off + (size_t)j * nv, (size_t)9 * ci, nv, co);
                add_var(base + "/biases", 1, nv, 0, 0, 0, op.b_off + (size_t)j * nv, 1, nv, nv);
            }
        }
    }
    add_var("l2_norm_conv4_3/scale", 1, 512, 0, 0, 0, scale_off_, 1, 512, 512);
}

// Measurement aid (tools/step_time.py), never set in a product run: ssd_debug_set_ablate("<tokens>") drops groups of launches
// from the step so their price INSIDE the overlapped step can be read off (results are wrong by construction; a warning goes
// to stderr whenever the set changes).  Tokens: pool, tail (conv8_2 ... conv11_2 and the small maps' heads), heads01, conv1,
// conv5, l2norm, reduce (conv_igemm.hip).
static std::string g_ablate;
static std::mutex g_ablate_mu;                     // (set from one thread while another runs a step: the string is never read unlocked)
static std::atomic<bool> g_ablate_on{false};       // the fast path of every launch site: one relaxed load
void set_ablate(const char* tokens) {
    std::lock_guard<std::mutex> lock(g_ablate_mu);
    g_ablate = tokens ? tokens : "";
    g_ablate_on.store(!g_ablate.empty(), std::memory_order_release);
    if (!g_ablate.empty())
        fprintf(stderr, "[ssdvgg_hip] WARNING: ablation '%s' is active: the step SKIPS launches, every result is WRONG (timing aid only)\n",
                g_ablate.c_str());
}
bool ablated(const char* token) {
    if (!g_ablate_on.load(std::memory_order_acquire)) return false;
    std::lock_guard<std::mutex> lock(g_ablate_mu);
    return g_ablate.find(token) != std::string::npos;
}
static bool op_ablated(const std::string& name, int kind, int head, int k) {
    if (!g_ablate_on.load(std::memory_order_acquire)) return false;
    if (kind == 1) return k == 2 && ablated("pool");
    if (kind == 2) return ablated("l2norm");
    if (head >= 2) return ablated("tail");
    if (head >= 0) return ablated("heads01");
    if (name.rfind("conv1_", 0) == 0 && name.size() == 7) return ablated("conv1");
    if (name.rfind("conv5_", 0) == 0) return ablated("conv5");
    if (name.rfind("conv8_2", 0) == 0 || name.rfind("conv9_", 0) == 0 || name.rfind("conv10_", 0) == 0 || name.rfind("conv11_", 0) == 0 ||
        name.rfind("conv12_", 0) == 0)
        return ablated("tail");
    return false;
}

static int env_i(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// Issue orders.  The op list is the reference's graph order (ssdvgg.py:190-372: trunk, extra layers, l2-norm, classifiers); the
// arena layout and the variables follow it.  What is ISSUED when is a separate matter:
//  * Forward: a branch op (the l2-norm, a fused multibox head) is issued right behind the tensor it reads.  In graph order
//    all of them sat behind conv11_2, and because the heads of a lane share one in-order side stream whose first entry
//    (head 0) needs the l2-norm, the six heads ran back to back AFTER the trunk on a nearly empty chip: 340 us of a
//    7.47 ms bf16 step with fewer than 256 workgroups in flight (profiles/r04_a_timeline_bf16.txt; fp32: 0.6 ms).
//  * Backward: reverse graph order, in bf16 with the big heads behind the chain (below).  (Round 4 built and measured another
//    alternative -- head 0 + l2-norm deferred beside mod_conv6 on the side stream: chain 450 -> 264 us, step 7.22 -> 7.37 ms,
//    profiles/r04_g_ab_schedule_bf16.txt; re-measured in round 5: 6.864 -> 6.885 ms.  The bookkeeping (bw_need / bw_sync /
//    bw_final_lo) serves any valid order.)
void Net::build_orders() {
    const int n = (int)ops_.size();
    fwd_order_.clear();
    std::vector<char> placed(n, 0);
    std::vector<int> stack;
    for (int i0 = 0; i0 < n; ++i0) {
        if (placed[i0]) continue;
        stack.assign(1, i0);
        while (!stack.empty()) {
            const int i = stack.back();
            stack.pop_back();
            if (placed[i]) continue;
            placed[i] = 1;
            fwd_order_.push_back(i);
            for (int j = n - 1; j > i; --j)      // (pushed in reverse: popped in graph order)
                if (!placed[j] && ops_[j].in == ops_[i].out && (ops_[j].kind == OP_L2NORM || ops_[j].head >= 0)) stack.push_back(j);
        }
    }
    bwd_order_.clear();
    for (int i = n - 1; i >= 0; --i) bwd_order_.push_back(i);
    // (Round 5 ran the bf16 step with the two BIG heads and the l2-norm BEHIND the chain conv11_2 ... conv8_2 -- in graph order their data
    // gradients filled every CU while the chain's first small links were due: 6.864 -> 6.810 ms, profiles/r05_r_ab_bw_defer_bf16.txt.  With
    // round 6's one-launch tail there is no chain of small launches left to starve and graph order measures 0.7 - 0.9 % faster in bf16,
    // equal in fp32 (profiles/r06_bd_ab_bw_order_*.txt, r06_am_ab_bw_defer_sync_bf16.txt): the reordering and its switch are gone.)
    // every op exactly once
    SSD_REQUIRE((int)bwd_order_.size() == n && (int)fwd_order_.size() == n, "issue orders lost an op (%zu forward, %zu backward of %d)",
                fwd_order_.size(), bwd_order_.size(), n);
}

// stream class of an op's data gradient in backward: 0 = main stream, 1 = side stream (see build_orders)
// Round 5: of the small maps' heads only the LAST one's data gradient starts the side chain (conv11_2's needs it); the others'
// (maps 2 .. n-2) run on the MAIN stream in front of the two big heads.  On the one side stream the four of them ran back to back
// -- 200 us at 30-66 us each beside the big heads' kernels -- before conv11_2's data gradient could start, and the main stream then
// waited 316 us for the chain's end (profiles/r05_l_timeline_merged_tail_bf16.txt).  The chain picks their results up through an
// event per feature map (Tensor::gev), not through a join of the streams.  SSD_BW_SMALL_HEADS_MAIN=0: all of them on the side stream.
int Net::bw_class(const Op& op, int op_index) const {
    if (!(hstream_ && overlap_)) return 0;
    static const bool small_main = env_i("SSD_BW_SMALL_HEADS_MAIN", 1) != 0;
    if (op.kind == OP_CONV && op.head >= 2) return (small_main && op.head != heads_.nmaps - 1) ? 0 : 1;      // small maps' heads: a few workgroups each
    if (op.kind == OP_CONV && op.head < 0 && op_index > tail_first_) return 1;     // conv11_2 ... conv8_2 behind them
    return 0;
}

void Net::bw_sync(int x, int y) {
    hipEvent_t ev = y == 1 ? ev_h_ : ev_m2s_;
    HIP_OK(hipEventRecord(ev, y == 1 ? hstream_ : stream_));
    HIP_OK(hipStreamWaitEvent(x == 1 ? hstream_ : stream_, ev, 0));
    bw_seen_[x][y] = bw_issued_[y];
}

void Net::bw_need(int x, const Tensor& t) {
    if (t.gstream == x || t.gseq <= bw_seen_[x][t.gstream]) return;
    if (t.gev && t.gev_set) HIP_OK(hipStreamWaitEvent(x == 1 ? hstream_ : stream_, t.gev, 0));      // exactly the kernel that wrote it
    else bw_sync(x, t.gstream);
}

void Net::bw_wrote(int x, Tensor& t, bool carried) {
    t.gstream = x;
    t.gseq = ++bw_issued_[x];
    if (t.gev) {
        if (!carried) HIP_OK(hipEventRecord(t.gev, x == 1 ? hstream_ : stream_));      // (carried: the kernel's own stop event, common.h)
        t.gev_set = true;
    }
}

size_t Net::bw_final_lo() const {
    size_t lo = nfilters_;
    for (int i = (int)ops_.size() - 1; i >= 0; --i) {
        if (ops_[i].kind != OP_CONV) continue;
        if (!bw_conv_done_[i]) break;
        lo = ops_[i].w_off;      // conv ops own descending, adjacent filter ranges
    }
    return lo;
}

// Pool fusion (round 5).  A 2x2 stride-2 pool whose input has no other consumer (pool1-3: conv4_3 also feeds the l2-norm) is
// pure HBM traffic between two convolutions: forward reads the producer's output once more to write a quarter of it, backward
// reads the consumer's data gradient once more to write four times as much.  Where the kernels support it
//  * the PRODUCER's forward epilogue takes the maxima and writes the pooled tensor and the pool's record; its own output is
//    then never written -- nothing reads it in backward either (the record carries the argmax and the relu sign);
//  * the CONSUMER's data gradient scatters through the record straight into the producer's output gradient.
// SSD_POOL_FUSE: bit 0 forward, bit 1 backward, bit 2 (bf16) conv1_1's weight gradient inside conv1_2's data gradient
// (backward_step); default 7; 0 = the separate kernels, for the A/B and the identity checks.
void Net::plan_pool_fusion() {
    const int mode = env_i("SSD_POOL_FUSE", 7);      // (read per handle: the tests build a fused and an unfused handle in one process)
    fuse_first_wgrad_ = (mode & 4) != 0 && bf16_ && training_;
    for (int i = 0; i < (int)ops_.size(); ++i) {
        Op& pl = ops_[i];
        if (pl.kind != OP_POOL) continue;
        const Tensor& in = tensors_[pl.in];
        const Tensor& out = tensors_[pl.out];
        PoolDesc pd{Bmax_, in.H, in.W, in.C, out.H, out.W, pl.k, pl.stride, pl.pad_h, pl.pad_w};
        if (!maxpool_rec_applicable(pd) || in.consumers != 1 || out.consumers != 1) continue;
        int prod = -1, cons = -1;
        for (int j = 0; j < (int)ops_.size(); ++j) {
            if (ops_[j].out == pl.in) prod = j;
            if (ops_[j].in == pl.out) cons = j;
        }
        if (prod < 0 || cons < 0 || ops_[prod].kind != OP_CONV || ops_[cons].kind != OP_CONV) continue;
        if (!ops_[prod].relu || ops_[prod].head >= 0) continue;
        const ConvDesc dp = conv_desc(ops_[prod], Bmax_), dc = conv_desc(ops_[cons], Bmax_);
        const bool in_bf16 = bf16_ && !tensors_[ops_[prod].in].data_f32;
        if ((mode & 1) && (bf16_ ? (in_bf16 && conv_fwd_pool_bf16_supported(dp)) : conv_fwd_pool_supported(dp))) {
            ops_[prod].pool_after = i;
            pl.fused_fwd = true;
        }
        if ((mode & 2) && training_ && pl.pool_rec && (bf16_ ? conv_dgrad_unpool_bf16_supported(dc) : conv_dgrad_unpool_supported(dc))) {
            ops_[cons].unpool = i;
            pl.fused_bwd = true;
        }
    }
}

// Round 6, the tail as one launch per direction (bf16; conv.h TailStage, tail_bf16.hip).  The chain starts at the first 1x1
// "convN_1" layer behind conv8_2 whose input map is at most 10 x 10 -- below that a layer is a tile or two per image and nothing
// couples two images -- and holds every extra layer from there on plus the multibox heads that read their outputs:
//   forward   conv9_1, conv9_2, head3, conv10_1, conv10_2, head4, conv11_1, conv11_2, head5                    (vgg300)
//   backward  data gradients of head5, head4, head3, conv11_2, conv11_1, conv10_2, conv10_1, conv9_2 in ONE launch (the first
//             layer's own data gradient accumulates into a tensor the 10x10 head also writes: it stays a launch), then the
//             weight gradients of all nine layers in ONE grouped launch with a direct epilogue (no slabs, no reduces).
// 27 launches of round 5 (9 forward, 8 + 10 backward, not counting 9 reduces) become 3 (+ one that packs the filters).
// SSD_TAIL_FUSE: bit 0 forward, bit 1 backward; default 3; 0 = the per-layer launches.
void Net::plan_tail_chain() {
    chain_first_ = -1;
    in_chain_.assign(ops_.size(), 0);
    const int mode = env_i("SSD_TAIL_FUSE", 3);      // (read per handle) bit 0: forward, bit 1: backward
    chain_fwd_ = chain_bwd_ = false;
    if (!bf16_ || (mode & 3) == 0) return;
    const int n = (int)ops_.size();
    for (int i = tail_first_ + 1; i < n; ++i) {
        const Op& op = ops_[i];
        if (op.kind != OP_CONV || op.head >= 0) continue;
        const Tensor& in = tensors_[op.in];
        if (op.KH == 1 && op.KW == 1 && in.H * in.W <= 100) { chain_first_ = i; break; }
    }
    if (chain_first_ < 0) return;
    std::vector<char> produced(tensors_.size(), 0);
    int count = 0;
    for (int i = chain_first_; i < n; ++i) {
        const Op& op = ops_[i];
        if (op.kind != OP_CONV) continue;
        const bool member = op.head < 0 ? true : produced[op.in] != 0;      // the extra layers; the heads that read one of them
        if (!member) continue;
        if (!tail_chain_stage_supported(conv_desc(op, 1)) || tensors_[op.in].data_f32) { count = -1; break; }
        in_chain_[i] = 1;
        if (op.head < 0) produced[op.out] = 1;
        ++count;
    }
    if (count < 2 || count > tail_chain_max_stages() || count > conv_wgrad_group_bf16_max()) {
        chain_first_ = -1;
        in_chain_.assign(ops_.size(), 0);
        return;
    }
    // the chain kernel reads its filters in fragment order (one contiguous KB per wave and k step): a third image of these few
    // layers' filters, refreshed with the bf16 mirrors every forward pass
    tail_fwd_off_.assign(ops_.size(), -1);
    tail_bwd_off_.assign(ops_.size(), -1);
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (!in_chain_[i]) continue;
        const ConvDesc d = conv_desc(ops_[i], 1);
        tail_fwd_off_[i] = (long long)total;
        total += tail_chain_packed_elems(d, false);
        if (training_ && i != chain_first_) {
            tail_bwd_off_[i] = (long long)total;
            total += tail_chain_packed_elems(d, true);
        }
    }
    wq_tail_ = (bf16_t*)dalloc(total * 2);
    chain_fwd_ = (mode & 1) != 0;
    chain_bwd_ = (mode & 2) != 0 && training_;
}

void Net::pack_tail_filters(hipStream_t s) {
    std::vector<TailPackItem> items;
    for (int i = 0; i < (int)ops_.size(); ++i) {
        if (!in_chain_[i]) continue;
        const ConvDesc d = conv_desc(ops_[i], 1);
        items.push_back(TailPackItem{d, false, wq_oi_ + ops_[i].w_off, wq_tail_ + tail_fwd_off_[i]});
        if (tail_bwd_off_[i] >= 0) items.push_back(TailPackItem{d, true, wq_io_ + ops_[i].w_off, wq_tail_ + tail_bwd_off_[i]});
    }
    tail_chain_pack_filters(items.data(), (int)items.size(), s);
}

void Net::launch_tail_forward(int b0, int nb, hipStream_t s) {
    std::vector<TailStage> st;
    for (const int oi : fwd_order_) {
        if (!in_chain_[oi]) continue;
        const Op& op = ops_[oi];
        const Tensor& in = tensors_[op.in];
        const Tensor& out = tensors_[op.out];
        TailStage t{};
        t.d = conv_desc(op, 1);
        t.dgrad = false;
        t.src = static_cast<const char*>(in.data) + (size_t)b0 * in.per_image() * 2;
        t.wgt_packed = wq_tail_ + tail_fwd_off_[oi];
        t.bias = params_ + op.b_off;
        t.dst = static_cast<char*>(out.data) + (size_t)b0 * out.per_image() * (out.data_f32 ? 4 : 2);
        t.relu = op.relu; t.accum = false; t.out_f32 = out.data_f32;
        st.push_back(t);
    }
    prof_.layer = "tail";
    tail_chain_bf16(st.data(), (int)st.size(), nb, "tail_fwd_bf16", s);
}

// The chain's part of backward (see plan_tail_chain): called when backward_step meets the chain's first op in its order.
void Net::launch_tail_backward(int b, bool* side_used) {
    const bool side = hstream_ && overlap_;
    const int cls = side ? 1 : 0;
    hipStream_t ds = cls == 1 ? hstream_ : stream_;
    std::vector<TailStage> st;
    std::vector<Tensor*> written;
    Tensor* first_out = &tensors_[ops_[chain_first_].out];      // the gradient the first layer's own data gradient (a launch) reads
    for (const int oi : bwd_order_) {
        if (!in_chain_[oi] || oi == chain_first_) continue;
        const Op& op = ops_[oi];
        Tensor& in = tensors_[op.in];
        const Tensor& out = tensors_[op.out];
        bw_need(cls, out);                                      // (the loss gradient of the heads: main stream)
        const bool last = in.done + 1 == in.consumers;
        TailStage t{};
        t.d = conv_desc(op, 1);
        t.dgrad = true;
        t.src = out.grad;
        t.wgt_packed = wq_tail_ + tail_bwd_off_[oi];
        t.mask = (last && in.relu_out) ? in.data : nullptr;
        t.dst = in.grad;
        t.accum = in.done > 0;
        st.push_back(t);
        in.done++;
        if (std::find(written.begin(), written.end(), &in) == written.end()) written.push_back(&in);
    }
    prof_.layer = "tail";
    const bool carry = stop_events_ && first_out->gev != nullptr;
    g_stop_event = carry ? first_out->gev : nullptr;
    {
        struct Disarm { ~Disarm() { g_stop_event = nullptr; } } disarm;
        tail_chain_bf16(st.data(), (int)st.size(), b, "tail_dgrad_bf16", ds);
    }
    for (Tensor* t : written) {      // one kernel wrote them all: the carried event (if any) stands for every one of them
        t->gstream = cls;
        t->gseq = ++bw_issued_[cls];
        t->gev_set = false;
    }
    if (carry) first_out->gev_set = true;
    // ---- the weight gradients of every layer of the chain, one launch behind the data gradients
    const bool wside = wstream_ && overlap_;
    hipStream_t ws = wside ? wstream_ : stream_;
    if (wside || cls == 1) {
        if (carry) {
            HIP_OK(hipStreamWaitEvent(ws, first_out->gev, 0));
        } else {
            HIP_OK(hipEventRecord(ev_dy_, ds));
            HIP_OK(hipStreamWaitEvent(ws, ev_dy_, 0));
        }
        if (!wside) bw_seen_[0][1] = bw_issued_[1];      // (the main stream has now waited for the side stream's chain)
    }
    // (the chain's first layer, on the 10x10 map -- 3200 pixels at batch 32 -- would be the group's straggler with its single pixel
    // split: 50 iterations on 8 workgroups; it keeps its own launch and slab reduce, issued when its turn comes)
    std::vector<WgradGroupItem> items;
    in_wgroup_.assign(ops_.size(), 0);
    for (int oi = 0; oi < (int)ops_.size(); ++oi) {
        if (!in_chain_[oi]) continue;
        const Op& op = ops_[oi];
        const Tensor& in = tensors_[op.in];
        const Tensor& out = tensors_[op.out];
        WgradGroupItem it{};
        it.d = conv_desc(op, b);
        if (oi == chain_first_) continue;      // (whatever the batch: backward_ranges, which knows no batch, mirrors this)
        in_wgroup_[oi] = 1;
        it.x = in.h(); it.dy = out.gh();
        it.dw = grads_ + op.w_off; it.dbias = grads_ + op.b_off; it.w = params_ + op.w_off;
        items.push_back(it);
        bw_conv_done_[oi] = 1;
    }
    conv_wgrad_group_bf16(items.data(), (int)items.size(), wd_, ws);
    if (wside) *side_used = true;
}

// Round 6 (fp32): which layers take the Winograd form (conv.h wino_*).  SSD_WINOGRAD = bits 0 forward / 1 data gradient / 2 weight
// gradient / 3 the big maps' multibox heads too (default 15; the weight gradient reads the forward's transform of the layer's input, so bit 2 needs bit 0);
// SSD_WINO_MIN_CC = the smallest Ci * Co that takes it (default 4096: conv1_2 and up).  The transforms move 2.25x the activations per
// pass and the 36 GEMMs have k = the channel count: at 64 -> 64 channels they are HBM-bound (16 FLOP per byte), and only with 64-wide
// tiles does conv1_2 beat its direct kernels -- 1.55 / 1.65 / 1.14 ms forward / data / weight gradient against 1.95 / 1.91 / 2.04
// (profiles/r06_aw_bench_conv_narrow_tiles.txt; with 128-wide tiles, half of them zeros: 1.90 / 2.18 / 2.2,
// profiles/r06_ao_per_layer_f32_wino_all.txt).  The trunk's 3x3 / stride 1 / SAME layers with Ci in multiples of 32.
void Net::plan_winograd() {
    if (bf16_) return;
    const int mode = env_i("SSD_WINOGRAD", 15), min_cc = env_i("SSD_WINO_MIN_CC", 4096), head_min_hw = env_i("SSD_WINO_HEAD_MIN_HW", 16);
    if (!(mode & 7)) return;
    size_t m_max = 0, v_max = 0, yt_max = 0, xw_max = 0, slab_max = 0;
    for (Op& op : ops_) {
        if (op.kind != OP_CONV) continue;
        const ConvDesc d = conv_desc(op, Bmax_);
        if (!wino_applicable(d) || d.Ci * d.Co < min_cc) continue;
        // the multibox heads of the big maps (38x38 / 19x19; vgg512: 64x64 ... 16x16): below that a layer is a handful of tiles
        // whose three launches cost more than the one they replace
        if (op.head >= 0 && (d.Ho < head_min_hw || !(mode & 8))) continue;
        op.wino_f = (mode & 1) != 0;
        op.wino_d = (mode & 2) != 0 && training_ && op.in != input_t_;
        op.wino_w = (mode & 4) != 0 && training_ && op.wino_f;
        if (!(op.wino_f || op.wino_d)) continue;
        const size_t u = (size_t)36 * d.Ci * d.Co, uf = (size_t)36 * d.Ci * wino_kpad(d.Co), t = (size_t)36 * wino_tiles(d);
        if (op.wino_f) {
            op.wino_U = (float*)dalloc(u * sizeof(float));
            // the input's transform: kept per layer by a training handle (the weight gradient reads it), per-stream scratch otherwise
            if (training_) op.wino_V = (float*)dalloc(t * d.Ci * sizeof(float));
            else v_max = std::max(v_max, t * d.Ci);
            m_max = std::max(m_max, t * d.Co);
        }
        if (op.wino_f && op.wino_d) op.wino_bits = dalloc((size_t)wino_tiles(d) * (d.Ci / 4) * 8);      // (against the fp32 mask: 23.47 -> 23.02 ms, profiles/r06_be_*)
        if (op.wino_d) {
            op.wino_Uf = (float*)dalloc(uf * sizeof(float));
            HIP_OK(hipMemset(op.wino_Uf, 0, uf * sizeof(float)));      // (rows Co ... kpad(Co) - 1 of every position stay zero)
            yt_max = std::max(yt_max, t * wino_kpad(d.Co));
            xw_max = std::max(xw_max, t * d.Ci);
        }
        if (op.wino_w) {
            yt_max = std::max(yt_max, t * wino_kpad(d.Co));
            for (int b = 1; b <= Bmax_; ++b) slab_max = std::max(slab_max, wino_wgrad_ws_floats(conv_desc(op, b)));
        }
        // (the first Winograd layer's forward transform is a launch of its own: the lanes wait for a 16 KB filter, not for all of them)
        (wino_plan_first_.n == 0 && op.wino_f ? wino_plan_first_ : wino_plan_).add(params_ + op.w_off, op.wino_U, nullptr, d.Ci, d.Co);
        if (op.wino_Uf) wino_plan_flip_.add(params_ + op.w_off, nullptr, op.wino_Uf, d.Ci, d.Co);
        wino_any_f_ |= op.wino_f;
        wino_any_d_ |= op.wino_d;
    }
    if (wino_plan_.n + wino_plan_first_.n + wino_plan_flip_.n == 0) return;
    HIP_OK(hipEventCreateWithFlags(&ev_wino_, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ev_wino_first_, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ev_wino_flip_, hipEventDisableTiming));
    if (m_max) for (int l = 0; l < 3; ++l) wino_m_[l] = (float*)dalloc(m_max * sizeof(float));
    if (v_max) for (int l = 0; l < 3; ++l) wino_vs_[l] = (float*)dalloc(v_max * sizeof(float));
    if (training_) {
        wino_yt_ = (float*)dalloc(yt_max * sizeof(float));
        wino_xw_ = (float*)dalloc(xw_max * sizeof(float));
        wino_ya_ = (float*)dalloc(yt_max * sizeof(float));
        wino_slab_ = (float*)dalloc(slab_max * sizeof(float));
    }
}

void Net::pool_fusion(int* out, int cap, int* count) const {
    int k = 0;
    for (const Op& op : ops_)
        if (op.kind == OP_POOL && op.k == 2 && op.stride == 2) {
            if (out && k < cap) out[k] = (op.fused_fwd ? 1 : 0) | (op.fused_bwd ? 2 : 0);
            ++k;
        }
    if (count) *count = k;
}

size_t Net::arena_floats(const char* preset, int num_classes) {
    // cheap: build the graph description only
    const Preset& p = get_preset(preset);
    const int nv = num_classes + 5;
    size_t n = 0;
    auto cv = [&](int k, int ci, int co) { n += (size_t)k * k * ci * co + co; };
    cv(3, 3, 64); cv(3, 64, 64); cv(3, 64, 128); cv(3, 128, 128); cv(3, 128, 256); cv(3, 256, 256); cv(3, 256, 256);
    cv(3, 256, 512); cv(3, 512, 512); cv(3, 512, 512); cv(3, 512, 512); cv(3, 512, 512); cv(3, 512, 512);
    cv(3, 512, 1024); cv(1, 1024, 1024);
    cv(1, 1024, 256); cv(3, 256, 512); cv(1, 512, 128); cv(3, 128, 256); cv(1, 256, 128); cv(3, 128, 256);
    cv(1, 256, 128); cv(3, 128, 256);
    if (p.nmaps >= 7) { cv(1, 256, 128); cv(3, 128, 256); }
    static const int fch[] = {512, 1024, 512, 256, 256, 256, 256};
    for (int i = 0; i < p.nmaps; ++i) cv(3, fch[i], round8(p.ntypes[i] * nv));
    return n + 512;
}

// ---------------------------------------------------------------------------------
// memory
// ---------------------------------------------------------------------------------
void* Net::dalloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    HIP_OK(hipMalloc(&p, bytes));
    allocs_.push_back(p);
    return p;
}

ConvDesc Net::conv_desc(const Op& op, int b) const {
    const Tensor& in = tensors_[op.in];
    const Tensor& out = tensors_[op.out];
    ConvDesc d;
    d.B = b; d.Hi = in.H; d.Wi = in.W; d.Ci = in.C; d.Ho = out.H; d.Wo = out.W; d.Co = out.C;
    d.KH = op.KH; d.KW = op.KW; d.stride = op.stride; d.dil = op.dil; d.pad_h = op.pad_h; d.pad_w = op.pad_w;
    return d;
}

void Net::alloc() {
    const int B = Bmax_;
    const int nv = C_ + 5;
    const int A = preset_->num_anchors;
    for (size_t i = 0; i < tensors_.size(); ++i) {
        Tensor& t = tensors_[i];
        if ((int)i == input_t_) continue;
        t.data = dalloc(t.per_image() * B * (t.data_f32 ? 4 : 2));
        HIP_OK(hipMemset(t.data, 0, t.per_image() * B * (t.data_f32 ? 4 : 2)));
        if (training_) {
            t.grad = dalloc(t.per_image() * B * (t.grad_f32 ? 4 : 2));
            HIP_OK(hipMemset(t.grad, 0, t.per_image() * B * (t.grad_f32 ? 4 : 2)));   // head pad columns stay 0 forever
        }
    }
    for (int i = 0; i < heads_.nmaps; ++i) {
        heads_.buf[i] = static_cast<float*>(tensors_[head_t_[i]].data);
        heads_.dbuf[i] = tensors_[head_t_[i]].grad;
    }
    if (bf16_) {
        wq_io_ = (bf16_t*)dalloc(nfilters_ * 2);
        wq_oi_ = (bf16_t*)dalloc(nfilters_ * 2);
    }
    if (!params_) { params_ = (float*)dalloc(nparams_ * sizeof(float)); own_params_ = true; }
    if (training_) {
        if (!grads_) { grads_ = (float*)dalloc(nparams_ * sizeof(float)); own_grads_ = true; }
        if (!mom_) { mom_ = (float*)dalloc(nparams_ * sizeof(float)); own_mom_ = true; }
        HIP_OK(hipMemset(grads_, 0, nparams_ * sizeof(float)));
        HIP_OK(hipMemset(mom_, 0, nparams_ * sizeof(float)));
        // split-M slab workspace: one region per layer (the reduces of a backward stage run as one grouped launch at its
        // end, so a layer's slabs must survive the next layer's weight gradient); sized for the worst batch <= B
        size_t ws_total = 0;
        for (auto& op : ops_)
            if (op.kind == OP_CONV) {
                size_t ws = 0;
                for (int b = 1; b <= B; ++b) {
                    const ConvDesc d = conv_desc(op, b);
                    ws = std::max(ws, (bf16_ && d.Ci % 8 == 0) ? conv_wgrad_bf16_ws_floats(d) : conv_wgrad_ws_floats(d));
                    if (bf16_ && first_layer_kernel(d)) ws = std::max(ws, std::max(conv_first_wgrad_bf16_ws_floats(d), conv_dgrad_first_wgrad_bf16_ws_floats(d)));
                }
                op.ws_off = ws_total;
                ws_total += (ws + 63) / 64 * 64;
            }
        wgrad_ws_ = (float*)dalloc(ws_total * sizeof(float));
        l2_ws_ = (float*)dalloc(l2norm_bwd_ws_floats(B * 64 * 64, 512) * sizeof(float));
        size_t pws = 0;
        for (auto& op : ops_)
            if (op.kind == OP_POOL) {
                const Tensor& in = tensors_[op.in];
                const Tensor& out = tensors_[op.out];
                PoolDesc d{B, in.H, in.W, in.C, out.H, out.W, op.k, op.stride, op.pad_h, op.pad_w};
                pws = std::max(pws, maxpool_bwd_ws_bytes(d));
            }
        pool_ws_ = dalloc(pws);
        // 2x2 pools whose input has no other consumer keep a forward record for their backward (ops.h)
        for (auto& op : ops_)
            if (op.kind == OP_POOL) {
                const Tensor& in = tensors_[op.in];
                const Tensor& out = tensors_[op.out];
                PoolDesc d{B, in.H, in.W, in.C, out.H, out.W, op.k, op.stride, op.pad_h, op.pad_w};
                if (maxpool_rec_applicable(d) && in.consumers == 1) op.pool_rec = dalloc(maxpool_rec_bytes(d));
            }
    }
    result_ = (float*)dalloc((size_t)B * A * nv * sizeof(float));
    x_stage_ = (float*)dalloc((size_t)B * preset_->image_h * preset_->image_w * 3 * sizeof(float));
    y_stage_ = (float*)dalloc((size_t)B * A * nv * sizeof(float));
    loss_ws_ = dalloc(loss_work_bytes(B, A));
    loss_work_carve(lw_, loss_ws_, B, A);
    HIP_OK(hipMemset(loss_ws_, 0, loss_work_bytes(B, A)));
    // The four losses are written by the loss kernel's final block straight into pinned, device-mapped host memory: a
    // 16-byte device-to-host copy would be a blit kernel of its own on the critical path between the loss and its
    // gradient (12 us of launch gap + the kernel in the rocprofv3 trace, tools/trace_gaps.py).
    // A ring of LOSS_RING such slots (one per forward pass) lets the host read step k - 1's losses while step k runs.
    HIP_OK(hipHostMalloc((void**)&losses_host_, LOSS_RING * 4 * sizeof(float), hipHostMallocMapped));
    for (int i = 0; i < LOSS_RING * 4; ++i) losses_host_[i] = 0.f;
    {
        void* dp = nullptr;
        HIP_OK(hipHostGetDevicePointer(&dp, losses_host_, 0));
        losses_dev_ = static_cast<float*>(dp);
        lw_.losses = losses_dev_;
    }
    for (int i = 0; i < LOSS_RING; ++i) HIP_OK(hipEventCreateWithFlags(&ev_loss_[i], hipEventDisableTiming));
    anchors_dev_ = (double*)dalloc((size_t)A * 4 * sizeof(double));
    anchors_abs_dev_ = (int*)dalloc((size_t)A * 4 * sizeof(int));
    anchors_device(*preset_, anchors_dev_, anchors_abs_dev_, nullptr);
    HIP_OK(hipDeviceSynchronize());
}

// splitmix64 -> uniform [0,1)
static inline double urand(unsigned long long& s) {
    unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

void Net::init_weights(unsigned long long seed) {
    std::vector<float> h(nparams_, 0.f);
    unsigned long long s = seed * 0x2545F4914F6CDD1Dull + 1;
    for (const Variable& v : vars_) {
        if (v.ndim == 4) {   // xavier_initializer (uniform): limit = sqrt(6 / (fan_in + fan_out)), ssdvgg.py:46
            const double fan_in = (double)v.shape[0] * v.shape[1] * v.shape[2];
            const double fan_out = (double)v.shape[0] * v.shape[1] * v.shape[3];
            const double lim = std::sqrt(6.0 / (fan_in + fan_out));
            for (size_t r = 0; r < v.rows; ++r)
                for (size_t c = 0; c < v.width; ++c) h[v.off + r * v.pitch + c] = (float)((urand(s) * 2.0 - 1.0) * lim);
        } else if (v.name == "l2_norm_conv4_3/scale") {
            for (size_t c = 0; c < v.width; ++c) h[v.off + c] = 20.f;   // ssdvgg.py:336
        }
    }
    HIP_OK(hipMemcpy(params_, h.data(), nparams_ * sizeof(float), hipMemcpyHostToDevice));
}

Net::Net(const char* preset, int num_classes, int max_batch, int device, bool training, unsigned long long seed,
         float* ext_params, float* ext_grads, float* ext_momentum, int dtype)
    : preset_(&get_preset(preset)), C_(num_classes), Bmax_(max_batch), device_(device), training_(training), bf16_(dtype == 1) {
    SSD_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (fp32) or 1 (bf16), got %d", dtype);
    SSD_REQUIRE(num_classes >= 1 && num_classes <= 27, "num_classes must be in 1..27 (got %d)", num_classes);
    SSD_REQUIRE(max_batch >= 1, "max_batch must be >= 1");
    HIP_OK(hipSetDevice(device));
    params_ = ext_params; grads_ = ext_grads; mom_ = ext_momentum;
    build_graph();
    alloc();
    init_weights(seed);
    const char* ov = getenv("SSD_OVERLAP_WGRAD");
    overlap_ = !(ov && ov[0] == '0');
    HIP_OK(hipStreamCreateWithFlags(&hstream_, hipStreamNonBlocking));      // (a high-priority side stream was measured again in round 5: +-0)
    HIP_OK(hipEventCreateWithFlags(&ev_h_, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ev_cast_, hipEventDisableTiming));

    HIP_OK(hipEventCreateWithFlags(&ev2_h_, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ev_l2_, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ev_m2s_, hipEventDisableTiming));
    for (int i = 0; i < MAX_MAPS; ++i) HIP_OK(hipEventCreateWithFlags(&ev2_fmap_[i], hipEventDisableTiming));
    for (int i = 0; i < MAX_MAPS; ++i) HIP_OK(hipEventCreateWithFlags(&ev_fmap_[i], hipEventDisableTiming));
    if (training_) {
        HIP_OK(hipStreamCreateWithFlags(&wstream_, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&ev_dy_, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&ev_w_, hipEventDisableTiming));
    }
    // FEW streams: the hardware runs four queues side by side (ROCm's default GPU_MAX_HW_QUEUES); a fifth
    // active stream is time-multiplexed onto one of them, and which two streams then share a queue depends on creation
    // order.  Measured (gpurun r02_v / r02_w, bf16 step): every stream on its own queue (8 queues) 3075 images/s, the
    // default four queues 4016 (and 3850 for a net created after others in the same process), four STREAMS 4080-4120
    // whatever the queue count.  The second forward lane and the weight gradients are never busy at the same time
    // (forward / backward), so they are ONE stream, and the second lane runs its heads on its own main stream as well: THREE
    // streams beside the caller's measure the same as four (bf16 4142 vs 4081 images/s, r02_w) and leave the fourth queue to
    // a data-parallel caller's collective stream.
    if (training_) {
        s2_ = wstream_;
        s2_is_w_ = true;
    } else {
        HIP_OK(hipStreamCreateWithFlags(&s2_, hipStreamNonBlocking));
    }
    plan_pool_fusion();
    plan_tail_chain();
    plan_winograd();
    // Round 5: an event per gradient tensor (round 4: the small heads' feature maps only).  The data-gradient kernel that writes it
    // CARRIES the event (g_stop_event), and the weight-gradient stream / the other class wait for exactly that kernel: no event
    // packet between two data gradients on the main stream, none inside the side chain.  SSD_STOP_EVENTS=0: round 4's records.
    stop_events_ = env_i("SSD_STOP_EVENTS", 1) != 0;
    if (training_)
        for (const Op& op : ops_)
            if ((stop_events_ ? op.in != input_t_ : (op.kind == OP_CONV && op.head >= 2)) && !tensors_[op.in].gev)
                HIP_OK(hipEventCreateWithFlags(&tensors_[op.in].gev, hipEventDisableTiming));
}

Net::~Net() {
    if (g_prof == &prof_) g_prof = nullptr;
    int prev_dev = device_;
    (void)hipGetDevice(&prev_dev);
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();
    for (DetectSlot& sl : det_slot_) {
        if (sl.dev && !sl.mapped) (void)hipFree(sl.dev);
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.ready) (void)hipEventDestroy(sl.ready);
    }
    if (hstream_) {
        (void)hipStreamDestroy(hstream_);
        (void)hipEventDestroy(ev_h_);
        (void)hipEventDestroy(ev_cast_);
        if (ev_wino_) (void)hipEventDestroy(ev_wino_);
        if (ev_wino_first_) (void)hipEventDestroy(ev_wino_first_);
        if (ev_wino_flip_) (void)hipEventDestroy(ev_wino_flip_);
        if (!s2_is_w_) (void)hipStreamDestroy(s2_);
        (void)hipEventDestroy(ev2_h_);
        (void)hipEventDestroy(ev_l2_);
        (void)hipEventDestroy(ev_join_);
        (void)hipEventDestroy(ev_m2s_);
        for (int i = 0; i < MAX_MAPS; ++i) (void)hipEventDestroy(ev2_fmap_[i]);
        for (int i = 0; i < MAX_MAPS; ++i) (void)hipEventDestroy(ev_fmap_[i]);
    }
    if (wstream_) {
        if (own_wstream_) (void)hipStreamDestroy(wstream_);
        (void)hipEventDestroy(ev_dy_);
        (void)hipEventDestroy(ev_w_);
    }
    for (Tensor& t : tensors_)
        if (t.gev) (void)hipEventDestroy(t.gev);
    for (void* p : allocs_) (void)hipFree(p);
    if (losses_host_) (void)hipHostFree(losses_host_);
    for (int i = 0; i < LOSS_RING; ++i)
        if (ev_loss_[i]) (void)hipEventDestroy(ev_loss_[i]);
    if (prev_dev != device_) (void)hipSetDevice(prev_dev);
}

// ---------------------------------------------------------------------------------
// steps
// ---------------------------------------------------------------------------------
// Forward runs the op list on up to two LANES: the batch is cut in two halves that walk the network side by side (lane 0
// on the caller's stream + the side stream for its heads, lane 1 with its heads on s2_, which is the
// weight-gradient stream: see the constructor), so that the tail of one half's kernel (its last, partial round of workgroups) is filled by the other half's
// kernel instead of idle CUs.  Measured on one box (SSD_FWD_LANES=1 / 2, gpurun r02): training step +0.5 % fp32 and
// +2.7 % bf16, inference at batch 128 +1.6 %.  Nothing in forward couples the samples except the loss's final
// reduction, which the last per-sample workgroup of either lane performs (ops.hip).
void Net::forward(const float* x, int b, bool train_mode, const float* y) {
    SSD_REQUIRE(b >= 1 && b <= Bmax_, "batch %d outside 1..%d (max_batch)", b, Bmax_);
    g_prof = &prof_;
    tensors_[input_t_].data = const_cast<float*>(x);
    pool_arg_op_ = -1;
    ++fwd_serial_;
    const bool side = hstream_ && overlap_;
    static const int lanes_env = [] { const char* v = getenv("SSD_FWD_LANES"); return v ? atoi(v) : 0; }();
    const int want_lanes = lanes_env > 0 ? lanes_env : 2;
    const int nl = (want_lanes >= 2 && side && s2_ && b >= 8) ? 2 : 1;
    const bool heads_full = nl == 2;      // with two lanes the multibox heads run as full-batch launches (see the head ops below)
    struct Lane {
        hipStream_t s, h;
        hipEvent_t* ev_fmap;
        hipEvent_t ev_h;
        int b0, nb;
        bool heads_on_side, cast_pending;
        int wino_pending;                 // filter-transform events this lane has still to wait for: 2 = the first layer's, then 1 = the rest
        bool fmap_carried[MAX_MAPS];      // the feature map's producer carried ev_fmap[head] itself (g_stop_event)
    } lane[2] = {{stream_, hstream_, ev_fmap_, ev_h_, 0, nl == 2 ? (b + 1) / 2 : b, false, false, 0, {}},
                 {s2_, s2_, ev2_fmap_, ev2_h_, (b + 1) / 2, b - (b + 1) / 2, false, false, 0, {}}};
    // feature map tensor -> the head that reads it (-1: none), for the carried events
    auto head_of = [&](int tensor) {
        for (const Op& o : ops_)
            if (o.kind == OP_CONV && o.head >= 0 && o.in == tensor) return o.head;
        return -1;
    };
    auto at = [](const Tensor& t, int b0, bool grad = false) -> char* {      // first element of sample b0
        return static_cast<char*>(grad ? t.grad : t.data) + (size_t)b0 * t.per_image() * ((grad ? t.grad_f32 : t.data_f32) ? 4 : 2);
    };
    if (side && (bf16_ || train_mode || nl == 2)) {
        HIP_OK(hipEventRecord(ev_fmap_[MAX_MAPS - 1], stream_));      // everything issued so far (the previous step's update)
        HIP_OK(hipStreamWaitEvent(hstream_, ev_fmap_[MAX_MAPS - 1], 0));
        if (nl == 2) HIP_OK(hipStreamWaitEvent(s2_, ev_fmap_[MAX_MAPS - 1], 0));
        lane[0].heads_on_side = true;
    }
    if (wino_plan_.n + wino_plan_first_.n > 0) {
        // Winograd layers (plan_winograd): the filters' transforms, fresh from the fp32 masters like the bf16 mirrors below, on the side
        // stream beside conv1_1: the first such layer's own (a launch of microseconds: what the lanes wait for), the other layers', and --
        // training -- the flipped forms of the data gradients, which only backward waits for (backward_begin)
        prof_.layer = "filters";
        hipStream_t fs = side ? hstream_ : stream_;
        wino_filter_plan(wino_plan_first_, true, false, fs);
        if (side) HIP_OK(hipEventRecord(ev_wino_first_, fs));
        wino_filter_plan(wino_plan_, true, false, fs);
        if (side) HIP_OK(hipEventRecord(ev_wino_, fs));
        if (train_mode && wino_plan_flip_.n > 0) {
            wino_filter_plan(wino_plan_flip_, false, true, fs);
            if (side) HIP_OK(hipEventRecord(ev_wino_flip_, fs));
            wino_flip_pending_ = side;
        }
        if (side) lane[0].wino_pending = lane[1].wino_pending = 2;
    }
    if (bf16_) {
        // the fp32 masters may have been updated by the optimizer, a variable load or the caller (external
        // arena): refresh both bf16 filter mirrors, one launch -- on the side stream, beside conv1_1 (which reads
        // the fp32 master itself); the first layer that reads a mirror waits for it
        prof_.layer = "filters";
        cast_filters(cast_plan_, params_, wq_io_, wq_oi_, side ? hstream_ : stream_);
        if (chain_first_ >= 0) pack_tail_filters(side ? hstream_ : stream_);
        if (side) {
            HIP_OK(hipEventRecord(ev_cast_, hstream_));
            lane[0].cast_pending = lane[1].cast_pending = true;
        }
    }
    // A forward pass that fails after it has taken a slot of the loss ring gives the slot back: its event was never
    // recorded, and a reader one step late (get_losses_step) would otherwise be handed a slot of four passes ago.
    struct LossSlotGuard {
        long long& seq;
        bool armed = false;
        bool done = false;
        explicit LossSlotGuard(long long& s) : seq(s) {}
        void failed() { if (armed && !done) { --seq; armed = false; } }
        ~LossSlotGuard() { if (armed && !done) --seq; }
    } fwd_guard(loss_seq_);
    if (train_mode) {
        begin_loss_slot();
        fwd_guard.armed = true;
        // the l2 term reads every filter once (105 MB): on the side stream beside the first (matrix-bound) layers
        prof_.layer = "loss";
        l2_partials(params_, nfilters_, lw_, side ? hstream_ : stream_);
        if (nl == 2) HIP_OK(hipEventRecord(ev_l2_, hstream_));
    }
    // Round 5: the lanes MERGE at conv8_1.  The extra layers behind it are a chain of sixteen dependent launches of a handful of
    // workgroups each (conv8_1 ... conv11_2 on two lanes: 6-37 us per launch, every one of them latency- not work-bound): a
    // timeline of the step (profiles/r05_j_timeline_bf16.txt) shows 285 us between mod_conv7 and the loss with a nearly empty chip.
    // One lane runs the eight layers once, at twice the rows per launch and the same latency; the second lane's stream -- idle
    // from here on -- takes every other small head, which used to queue behind each other on the one side stream.
    // (Round 4 merged at the 19x19 maps, i.e. incl. conv5_x / mod_conv6, where two lanes fill each other's partial rounds: slower.)
    static const bool merge_tail = env_i("SSD_FWD_MERGE_TAIL", 1) != 0;      // A/B switch
    int nl_cur = nl;
    int small_heads = 0;
    for (const int op_index : fwd_order_) {
        const Op& op = ops_[op_index];
        const Tensor& in = tensors_[op.in];
        const Tensor& out = tensors_[op.out];
        prof_.layer = op.name.c_str();
        if (op_ablated(op.name, op.kind, op.head, op.k)) continue;
        if (chain_fwd_ && in_chain_[op_index]) {      // Round 6 (plan_tail_chain): the chain's layers and heads run as ONE launch per lane, issued at its first op
            if (op_index == chain_first_)
                for (int li = 0; li < nl_cur; ++li) {
                    Lane& ln = lane[li];
                    if (ln.cast_pending) {
                        HIP_OK(hipStreamWaitEvent(ln.s, ev_cast_, 0));
                        ln.cast_pending = false;
                    }
                    launch_tail_forward(ln.b0, ln.nb, ln.s);
                }
            continue;
        }
        if (nl_cur == 2 && merge_tail && heads_full && op_index == tail_first_) {
            HIP_OK(hipEventRecord(ev_join_, s2_));
            HIP_OK(hipStreamWaitEvent(stream_, ev_join_, 0));
            lane[0].nb = b;
            nl_cur = 1;
        }
        for (int li = 0; li < nl_cur; ++li) {
            Lane& ln = lane[li];
            const int nb = ln.nb;
            switch (op.kind) {
            case OP_CONV: {
                hipStream_t cs = ln.s;
                int run_nb = nb, run_b0 = ln.b0;
                if (op.head >= 0 && side) {
                    // the multibox heads hang off the trunk: they run on a side stream behind their feature
                    // map and fill the CUs the trunk's kernels leave idle between waves of workgroups
                    if (!ln.fmap_carried[op.head]) HIP_OK(hipEventRecord(ln.ev_fmap[op.head], ln.s));
                    ln.fmap_carried[op.head] = false;
                    if (heads_full) {
                        // ONE launch over the whole batch on lane 0's side stream, behind both lanes' feature maps: half the
                        // launches, twice the workgroups each, and lane 1's trunk (whose own "side" stream is its main
                        // stream) does not queue behind its heads
                        if (li < nl_cur - 1) break;
                        // (after the merge: the small maps' heads alternate between the side stream and the idle second lane's)
                        cs = (nl_cur == 1 && op.head >= 2 && (small_heads++ & 1)) ? s2_ : hstream_;
                        for (int l = 0; l < nl_cur; ++l) HIP_OK(hipStreamWaitEvent(cs, lane[l].ev_fmap[op.head], 0));
                        run_nb = b; run_b0 = 0;
                        lane[0].heads_on_side = true;
                    } else {
                        HIP_OK(hipStreamWaitEvent(ln.h, ln.ev_fmap[op.head], 0));
                        cs = ln.h;
                        ln.heads_on_side = true;
                    }
                }
                const ConvDesc d = conv_desc(op, run_nb);
                struct LanesScope {      // (tile choice of the bf16 kernel-row gather: the lanes share the chip's workgroup slots)
                    LanesScope(int n) { g_conv_lanes = n; }
                    ~LanesScope() { g_conv_lanes = 1; }
                } lanes_scope(run_nb == nb ? nl_cur : 1);
                if (ln.cast_pending && !in.data_f32) {
                    HIP_OK(hipStreamWaitEvent(ln.s, ev_cast_, 0));
                    ln.cast_pending = false;
                }
                const float* xin = reinterpret_cast<const float*>(at(in, run_b0));
                void* yout = at(out, run_b0);
                // Round 6: the Winograd form (plan_winograd).  A lane owns the tile rows of its images in the layer's full-batch
                // transform V (which the weight gradient reads) and its own GEMM-result scratch.
                // (GEMM-result scratch: one per stream that runs such layers -- the lanes' main streams and the side stream of the heads)
                float* const wino_M = cs == ln.s ? wino_m_[li] : cs == hstream_ ? wino_m_[2] : nullptr;
                const bool wino = op.wino_f && wino_M;
                float* wino_V = nullptr;
                void* wino_bits = nullptr;
                size_t wino_vps = 0;
                if (wino) {
                    if (cs == ln.s && ln.wino_pending) {      // (the side stream ran the filter transforms itself)
                        // (waiting for ALL transforms incl. the flipped ones before the first layer: 24.00 against 23.69 ms median,
                        // profiles/r06_bc_ab_filter_split_f32.txt)
                        HIP_OK(hipStreamWaitEvent(ln.s, ln.wino_pending == 2 ? ev_wino_first_ : ev_wino_, 0));
                        --ln.wino_pending;
                    }
                    const size_t tpi = (size_t)wino_tiles(d) / d.B;
                    if (op.wino_V) {      // a training handle: this launch's rows of the layer's full-batch transform
                        wino_V = op.wino_V + (size_t)run_b0 * tpi * d.Ci;
                        wino_vps = (size_t)b * tpi * d.Ci;
                        if (train_mode) ops_[op_index].wino_v_step = fwd_serial_;
                    } else {              // an inference handle: the stream's own scratch
                        wino_V = wino_vs_[cs == ln.s ? li : 2];
                        wino_vps = (size_t)run_nb * tpi * d.Ci;
                    }
                    // training: the input transform notes the relu mask of this layer's input for its own data gradient (conv.h)
                    if (train_mode && op.wino_bits) {
                        wino_bits = static_cast<char*>(op.wino_bits) + (size_t)run_b0 * tpi * (d.Ci / 4) * 8;
                        ops_[op_index].wino_bits_step = fwd_serial_;
                    }
                }
                if (op.pool_after >= 0) {
                    // Pool fusion (round 5): this conv's epilogue takes the 2x2 maxima itself and writes the POOLED tensor (+ the
                    // pool's 12-bit record in training); its own output has no other reader -- forward or backward -- and is
                    // never written (plan_pool_fusion)
                    const Op& pl = ops_[op.pool_after];
                    const Tensor& pt = tensors_[pl.out];
                    void* rec = (pl.pool_rec && train_mode)
                                    ? static_cast<char*>(pl.pool_rec) + (size_t)run_b0 * pt.H * pt.W * (pt.C / 4) * sizeof(unsigned short) : nullptr;
                    if (wino)
                        wino_fwd(d, xin, op.wino_U, params_ + op.b_off, nullptr, true, wino_V, wino_vps, wino_M,
                                 reinterpret_cast<float*>(at(pt, run_b0)), rec, cs, wino_bits);
                    else if (!bf16_)
                        conv_fwd_pool(d, xin, params_ + op.w_off, params_ + op.b_off, reinterpret_cast<float*>(at(pt, run_b0)), rec, cs);
                    else
                        conv_fwd_pool_bf16(d, reinterpret_cast<const bf16_t*>(xin), wq_oi_ + op.w_off, params_ + op.b_off,
                                           reinterpret_cast<bf16_t*>(at(pt, run_b0)), rec, cs);
                    break;
                }
                // a feature map's producer carries the event its head waits for (common.h g_stop_event; backward_step does the same)
                const int fh = (stop_events_ && side && op.head < 0 && cs == ln.s) ? head_of(op.out) : -1;
                if (fh >= 0) g_stop_event = ln.ev_fmap[fh];
                struct CarryScope {
                    bool* flag;
                    ~CarryScope() { if (flag) *flag = g_stop_event == nullptr; g_stop_event = nullptr; }
                } carry_scope{fh >= 0 ? &ln.fmap_carried[fh] : nullptr};
                if (wino)
                    wino_fwd(d, xin, op.wino_U, params_ + op.b_off, static_cast<float*>(yout), op.relu, wino_V, wino_vps, wino_M,
                             nullptr, nullptr, cs, wino_bits);
                else if (!bf16_)
                    conv_fwd(d, xin, params_ + op.w_off, params_ + op.b_off, static_cast<float*>(yout), op.relu, cs);
                else if (in.data_f32 && first_layer_kernel(d))      // conv1_1: fp32 image and master filter in, bf16 out
                    conv_first_fwd_bf16(d, xin, params_ + op.w_off, params_ + op.b_off, static_cast<bf16_t*>(yout), op.relu, cs);
                else if (in.data_f32)
                    conv_fwd_smallc_bf16out(d, xin, params_ + op.w_off, params_ + op.b_off, static_cast<bf16_t*>(yout), op.relu, cs);
                else
                    conv_fwd_bf16(d, reinterpret_cast<const bf16_t*>(xin), wq_oi_ + op.w_off, params_ + op.b_off, yout, out.data_f32, op.relu, cs);
                break;
            }
            case OP_POOL: {
                if (op.fused_fwd) break;      // written by its producer's epilogue
                PoolDesc d{nb, in.H, in.W, in.C, out.H, out.W, op.k, op.stride, op.pad_h, op.pad_w};
                if (op.pool_rec && train_mode) {
                    void* rec = static_cast<char*>(op.pool_rec) + (size_t)ln.b0 * out.H * out.W * (in.C / 4) * sizeof(unsigned short);
                    if (bf16_) maxpool_fwd_rec(d, reinterpret_cast<const bf16_t*>(at(in, ln.b0)), reinterpret_cast<bf16_t*>(at(out, ln.b0)), rec, ln.s);
                    else maxpool_fwd_rec(d, reinterpret_cast<const float*>(at(in, ln.b0)), reinterpret_cast<float*>(at(out, ln.b0)), rec, ln.s);
                } else if (train_mode && pool_ws_ && maxpool_arg_applicable(d)) {
                    // mod_pool5: the windows' first-maximum taps stay in pool_ws_ (this pool is its only user) for backward
                    void* arg = static_cast<char*>(pool_ws_) + (size_t)ln.b0 * out.H * out.W * (in.C / 4) * sizeof(unsigned);
                    if (bf16_) maxpool_fwd_arg(d, reinterpret_cast<const bf16_t*>(at(in, ln.b0)), reinterpret_cast<bf16_t*>(at(out, ln.b0)), arg, ln.s);
                    else maxpool_fwd_arg(d, reinterpret_cast<const float*>(at(in, ln.b0)), reinterpret_cast<float*>(at(out, ln.b0)), arg, ln.s);
                    pool_arg_op_ = op_index;
                } else if (bf16_) maxpool_fwd(d, reinterpret_cast<const bf16_t*>(at(in, ln.b0)), reinterpret_cast<bf16_t*>(at(out, ln.b0)), ln.s);
                else maxpool_fwd(d, reinterpret_cast<const float*>(at(in, ln.b0)), reinterpret_cast<float*>(at(out, ln.b0)), ln.s);
                break;
            }
            case OP_L2NORM:
                if (bf16_) l2norm_fwd(nb * in.H * in.W, in.C, reinterpret_cast<const bf16_t*>(at(in, ln.b0)), params_ + scale_off_,
                                      reinterpret_cast<bf16_t*>(at(out, ln.b0)), ln.s);
                else l2norm_fwd(nb * in.H * in.W, in.C, reinterpret_cast<const float*>(at(in, ln.b0)), params_ + scale_off_,
                                reinterpret_cast<float*>(at(out, ln.b0)), ln.s);
                break;
            }
        }
    }
    prof_.layer = "loss";
    const int A = preset_->num_anchors, nv = C_ + 5;
    // The lanes' loss launches share a self-resetting completion ticket (ops.hip): if a launch fails after another
    // lane's has been enqueued, the count never reaches the step's total and the ticket would stay non-zero for the life
    // of the handle -- drain the device and clear it before the error leaves.
    try {
    if (heads_full) {      // full-batch heads: one loss / result launch on the main stream behind the side stream and lane 1
        HIP_OK(hipEventRecord(ev_h_, hstream_));
        HIP_OK(hipStreamWaitEvent(stream_, ev_h_, 0));
        HIP_OK(hipEventRecord(ev_join_, s2_));
        HIP_OK(hipStreamWaitEvent(stream_, ev_join_, 0));
        if (train_mode) multibox_loss(heads_, b, 0, b, result_, y, lw_, wd_, loss_bnorm_, stream_);
        else heads_result(heads_, b, result_, stream_);
    }
    for (int li = 0; li < nl && !heads_full; ++li) {
        Lane& ln = lane[li];
        if (ln.heads_on_side) {
            HIP_OK(hipEventRecord(ln.ev_h, ln.h));
            HIP_OK(hipStreamWaitEvent(ln.s, ln.ev_h, 0));
        }
        HeadLayout hl = heads_;
        for (int i = 0; i < hl.nmaps; ++i) hl.buf[i] = heads_.buf[i] + (size_t)ln.b0 * hl.hw[i] * hl.ld[i];
        float* res = result_ + (size_t)ln.b0 * A * nv;
        if (train_mode) {
            if (li == 1) HIP_OK(hipStreamWaitEvent(ln.s, ev_l2_, 0));      // the final reduction may fall to this lane
            multibox_loss(hl, ln.nb, ln.b0, b, res, y + (size_t)ln.b0 * A * nv, lw_, wd_, loss_bnorm_, ln.s);
        } else {
            heads_result(hl, ln.nb, res, ln.s);
        }
    }
    } catch (...) {
        if (train_mode && lw_.ticket) {
            (void)hipDeviceSynchronize();
            (void)hipMemset(lw_.ticket, 0, sizeof(unsigned));
        }
        fwd_guard.failed();
        throw;
    }
    if (nl == 2) {
        HIP_OK(hipEventRecord(ev_join_, s2_));
        HIP_OK(hipStreamWaitEvent(stream_, ev_join_, 0));
    }
    if (train_mode) HIP_OK(hipEventRecord(ev_loss_[loss_seq_ % LOSS_RING], stream_));
    fwd_guard.done = true;
}

// Backward runs the op list in reverse.  It can be driven in stages so a data-parallel caller can
// start all-reducing finished gradient ranges while the rest of backward still runs: filters sit
// in the arena in forward order, so reverse execution completes them from the END of the filter
// region towards its beginning, always as one contiguous, growing suffix.
// The data-gradient chain is ONE lane (rounds 2-4 also carried a two-lane form, half batches on two sets of streams:
// fp32 -0.3 %, bf16 -7 %, profiles/r02_l_ab_bwd_lanes_*.txt -- backward already has the weight-gradient stream filling the
// data gradients' tails; removed in round 5).
void Net::launch_wgrad(int op_index, int b, hipStream_t ws) {
    const Op& op = ops_[op_index];
    const Tensor& in = tensors_[op.in];
    const Tensor& out = tensors_[op.out];
    const ConvDesc d = conv_desc(op, b);
    float* slab = wgrad_ws_ + op.ws_off;
    prof_.layer = op.name.c_str();
    // Round 6: from the forward's transform of the input and the transformed dy (conv.h wino_wgrad) -- if the forward pass of THIS step
    // left that transform (a launch that ran on a stream without Winograd scratch took the direct kernel: the direct weight gradient then)
    if (op.wino_w && op.wino_v_step == fwd_serial_) {
        const size_t tpi = (size_t)wino_tiles(d) / d.B;
        wino_bwd_transform(d, out.gf(), nullptr, wino_ya_, ws);
        wino_wgrad(d, op.wino_V, (size_t)b * tpi * d.Ci, wino_ya_, grads_ + op.w_off, grads_ + op.b_off, params_ + op.w_off, wd_,
                   wino_slab_, ws);
    } else if (!bf16_) {
        conv_wgrad(d, in.f(), out.gf(), grads_ + op.w_off, grads_ + op.b_off, params_ + op.w_off, wd_, slab, ws);
    } else if (in.data_f32) {   // conv1_1
        if (first_layer_kernel(d))
            conv_first_wgrad_bf16(d, in.f(), out.gh(), grads_ + op.w_off, grads_ + op.b_off, params_ + op.w_off, wd_, slab, ws);
        else
            conv_wgrad_smallc_bf16dy(d, in.f(), out.gh(), grads_ + op.w_off, grads_ + op.b_off, params_ + op.w_off, wd_, slab, ws);
    } else {
        conv_wgrad_bf16(d, in.h(), out.gh(), grads_ + op.w_off, grads_ + op.b_off, params_ + op.w_off, wd_, slab, ws);
    }
}

void Net::backward_begin(int b, const float* y) {
    SSD_REQUIRE(training_, "handle was created with training = 0");
    g_prof = &prof_;
    prof_.layer = "loss";
    const bool side = hstream_ && overlap_;
    for (Tensor& t : tensors_) { t.done = 0; t.gstream = 0; t.gseq = 0; t.gev_set = false; }
    bw_issued_[0] = bw_issued_[1] = 0;
    bw_seen_[0][0] = bw_seen_[0][1] = bw_seen_[1][0] = bw_seen_[1][1] = 0;
    if (wino_flip_pending_) {      // the data gradients' filter transforms (side stream, issued by forward)
        HIP_OK(hipStreamWaitEvent(stream_, ev_wino_flip_, 0));
        wino_flip_pending_ = false;
    }
    multibox_loss_grad(heads_, b, 0, result_, y, lw_, stream_);
    for (int t : head_t_) bw_wrote(0, tensors_[t]);      // the loss gradient, on the main stream
    if (side) bw_sync(1, 0);      // the side stream starts behind it (the small maps' head data gradients: backward_step)
    bw_conv_done_.assign(ops_.size(), 0);
    bw_first_on_main_ = false;
    bw_first_fused_ = false;
    bw_chain_done_ = false;
    in_wgroup_.assign(ops_.size(), 0);
    bw_pos_ = 0;
    bw_b_ = b;
    bw_done_off_ = nfilters_;
}

bool Net::backward_step(size_t min_floats, size_t* off, size_t* count, bool sync_main) {
    const int b = bw_b_;
    const size_t hi = bw_done_off_;
    size_t lo = hi;
    bool side_used = false;
    const int n_order = (int)bwd_order_.size();
    while (bw_pos_ < n_order && hi - lo < min_floats) {
        const int op_index = bwd_order_[bw_pos_++];
        const Op& op = ops_[op_index];
        Tensor& in = tensors_[op.in];
        const Tensor& out = tensors_[op.out];
        const bool need_dx = op.in != input_t_;
        const bool last = in.done + 1 == in.consumers;
        prof_.layer = op.name.c_str();
        // Where this op's data gradient runs (bw_class): the small maps' heads and the chain of extra layers behind them on the
        // side stream, everything else on the main stream.  Hand-offs between the two are events placed where a kernel reads
        // or accumulates into a gradient that the other class wrote last (bw_need): conv8_1 joins the chain.
        const int cls = bw_class(op, op_index);
        hipStream_t ds = cls == 1 ? hstream_ : stream_;
        if (chain_bwd_ && in_chain_[op_index] && !op_ablated(op.name, op.kind, op.head, op.k)) {
            // Round 6 (plan_tail_chain): the first chain op met issues the chain's data gradients (one launch) and every chain layer's
            // weight gradient (one grouped launch); the other members are skipped when their turn comes -- all but the chain's first
            // layer, whose data gradient is an ordinary launch below (its weight gradient came out of the group)
            if (!bw_chain_done_) {
                launch_tail_backward(b, &side_used);
                bw_chain_done_ = true;
                lo = bw_final_lo();
                prof_.layer = op.name.c_str();
            }
            if (op_index != chain_first_) continue;
        }
        if (op_ablated(op.name, op.kind, op.head, op.k)) {
            if (op.kind == OP_CONV) { bw_conv_done_[op_index] = 1; lo = bw_final_lo(); }
            in.done++;
            continue;
        }
        switch (op.kind) {
        case OP_CONV: {
            const ConvDesc d = conv_desc(op, b);
            // The weight gradient only feeds the optimizer; the data gradient is on the critical
            // path.  They read the same dy and write disjoint buffers, so the weight gradient goes
            // to a side stream: its workgroups fill the CUs the data-gradient's last partial wave
            // of workgroups leaves idle (and vice versa).
            // The first layer's weight gradient (no data gradient behind it: the main stream has gone idle by then) runs on the
            // MAIN stream beside conv1_2's weight gradient instead of queueing behind it -- both are HBM-bound at half the
            // achievable bandwidth (profiles/r04_a_timeline_bf16.txt: 7013 .. 7391 us of the step ran one such kernel at a time)
            const bool side = wstream_ && overlap_;
            const bool on_main = side && !need_dx;
            hipStream_t ws = (side && !on_main) ? wstream_ : stream_;
            if (side && !on_main) {
                // (dy written on the main stream but already waited for by the side stream, and this op lives there: the
                // side stream is the one that is less far ahead -- a small head's weight gradient need not wait for mod_conv7)
                if (stop_events_ && out.gev && out.gev_set) {
                    HIP_OK(hipStreamWaitEvent(wstream_, out.gev, 0));      // the kernel that wrote dy last (it waited for the earlier writers)
                } else {
                    const bool from_side = out.gstream == 1 || (cls == 1 && bw_seen_[1][0] >= out.gseq);
                    HIP_OK(hipEventRecord(ev_dy_, from_side ? hstream_ : stream_));
                    HIP_OK(hipStreamWaitEvent(wstream_, ev_dy_, 0));
                }
                side_used = true;
            } else {
                bw_need(0, out);
                if (on_main) bw_first_on_main_ = true;
            }
            if (!(bw_first_fused_ && !need_dx) && !(chain_bwd_ && bw_chain_done_ && in_wgroup_[op_index]))      // (conv1_1's came out of conv1_2's data gradient: below; the chain's first layer's out of the grouped launch)
                launch_wgrad(op_index, b, ws);
            if (need_dx) {
                bw_need(cls, out);
                // Pool fusion (round 5): when this conv reads a recorded 2x2 pool's output, its data gradient routes every
                // pooled pixel's dx through the record straight into the four cells of the POOL'S INPUT gradient (relu mask of
                // that tensor's producer included, from the record's sign bit): the pooled tensor's own gradient is never
                // written and the pool's backward pass is not launched.
                const Op* up = op.unpool >= 0 ? &ops_[op.unpool] : nullptr;
                Tensor& dst = up ? tensors_[up->in] : in;
                if (!up && in.done > 0) bw_need(cls, in);      // accumulates into what the other class wrote
                const bool mask = !up && last && in.relu_out;
                // conv1_2 in bf16: the layer below is the first layer, whose only use for dx is its weight gradient -- computed
                // here from the dx tiles while they are in LDS (conv.h conv_dgrad_first_wgrad_bf16); dx is never written and
                // conv1_1's own weight-gradient kernel is skipped when its turn comes
                int first = -1;
                if (fuse_first_wgrad_ && !up && mask && in.done == 0)
                    for (int j = 0; j < (int)ops_.size(); ++j)
                        if (ops_[j].kind == OP_CONV && ops_[j].out == op.in && ops_[j].in == input_t_) first = j;
                if (first >= 0 && cls == 0 && conv_dgrad_first_wgrad_bf16_applicable(d, conv_desc(ops_[first], b))) {
                    const Op& f = ops_[first];
                    prof_.layer = "conv1_2+conv1_1";
                    conv_dgrad_first_wgrad_bf16(d, out.gh(), wq_io_ + op.w_off, in.h(), conv_desc(f, b), tensors_[input_t_].f(),
                                                grads_ + f.w_off, grads_ + f.b_off, params_ + f.w_off, wd_, wgrad_ws_ + f.ws_off, ds);
                    bw_first_fused_ = true;
                    fused_first_out_ = op.in;
                    bw_wrote(cls, in);
                    bw_conv_done_[op_index] = 1;
                    lo = bw_final_lo();
                    break;
                }
                g_stop_event = stop_events_ ? dst.gev : nullptr;      // the launch carries dst's event (taken by the gather launchers)
                struct Disarm { ~Disarm() { g_stop_event = nullptr; } } disarm;      // (also when the launch throws)
                if (op.wino_d && cls == 0) {      // Round 6: the Winograd form (its scratch belongs to the main stream)
                    wino_bwd_transform(d, out.gf(), wino_yt_, nullptr, ds);
                    wino_dgrad(d, wino_yt_, op.wino_Uf, up ? dst.gf() : in.gf(), mask ? in.f() : nullptr, !up && in.done > 0, wino_xw_,
                               up ? up->pool_rec : nullptr, dst.H, dst.W, ds,
                               op.wino_bits && op.wino_bits_step == fwd_serial_ ? op.wino_bits : nullptr);      // (the forward of THIS step wrote them)
                } else if (!bf16_) {
                    if (up) conv_dgrad_unpool(d, out.gf(), params_ + op.w_off, dst.gf(), up->pool_rec, dst.H, dst.W, ds);
                    else conv_dgrad(d, out.gf(), params_ + op.w_off, in.gf(), mask ? in.f() : nullptr, in.done > 0, ds);
                } else {
                    if (up) conv_dgrad_unpool_bf16(d, out.gh(), wq_io_ + op.w_off, dst.gh(), up->pool_rec, dst.H, dst.W, ds);
                    else conv_dgrad_bf16(d, out.gh(), wq_io_ + op.w_off, in.gh(), mask ? in.h() : nullptr, in.done > 0, ds);
                }
                const bool carried = stop_events_ && dst.gev && g_stop_event == nullptr;
                bw_wrote(cls, dst, carried);
            }
            bw_conv_done_[op_index] = 1;
            lo = bw_final_lo();
            break;
        }
        case OP_POOL: {
            if (op.fused_bwd) break;      // its consumer's data gradient has already written in.grad through the record
            bw_need(0, out);
            if (in.done > 0) bw_need(0, in);
            PoolDesc d{b, in.H, in.W, in.C, out.H, out.W, op.k, op.stride, op.pad_h, op.pad_w};
            void* pws = maxpool_bwd_ws_bytes(d) ? pool_ws_ : nullptr;
            if (op.pool_rec && in.done == 0 && last) {      // single consumer: dx is overwritten from the forward record
                if (bf16_) maxpool_bwd_rec(d, op.pool_rec, out.gh(), in.gh(), in.relu_out, stream_);
                else maxpool_bwd_rec(d, op.pool_rec, out.gf(), in.gf(), in.relu_out, stream_);
            } else if (pool_arg_op_ == op_index && pws && maxpool_arg_applicable(d)) {      // the forward pass of this step left the record
                if (bf16_) maxpool_bwd_arg(d, in.h(), pws, out.gh(), in.gh(), in.done > 0, last && in.relu_out, stream_);
                else maxpool_bwd_arg(d, in.f(), pws, out.gf(), in.gf(), in.done > 0, last && in.relu_out, stream_);
            } else if (bf16_)
                maxpool_bwd(d, in.h(), out.gh(), in.gh(), in.done > 0, last && in.relu_out, pws, stream_);
            else
                maxpool_bwd(d, in.f(), out.gf(), in.gf(), in.done > 0, last && in.relu_out, pws, stream_);
            bw_wrote(0, in);
            break;
        }
        case OP_L2NORM: {
            SSD_REQUIRE(in.done == 0 && !last, "l2norm backward must be the first of several consumers");
            bw_need(cls, out);
            if (bf16_)
                l2norm_bwd(b * in.H * in.W, in.C, in.h(), params_ + scale_off_, out.gh(), in.gh(), grads_ + scale_off_, l2_ws_, ds);
            else
                l2norm_bwd(b * in.H * in.W, in.C, in.f(), params_ + scale_off_, out.gf(), in.gf(), grads_ + scale_off_, l2_ws_, ds);
            bw_wrote(cls, in);
            break;
        }
        }
        in.done++;
    }
    const bool finished = bw_pos_ >= n_order;
    if (finished && bw_issued_[1] > bw_seen_[0][1]) bw_sync(0, 1);      // (every side-stream result has a main-stream reader: a no-op)
    // The returned range is final in the weight-gradient stream's order.  Make it final in
    // main-stream order too unless the caller consumes it on the weight-gradient stream itself
    // (sync_main = false keeps the data gradients running ahead); the last stage always joins.
    if (wstream_ && overlap_ && (side_used || finished) && (sync_main || finished)) {
        HIP_OK(hipEventRecord(ev_w_, wstream_));
        HIP_OK(hipStreamWaitEvent(stream_, ev_w_, 0));
    }
    // Overlap switched off (ssd_set_overlap(0) / SSD_OVERLAP_WGRAD=0): the weight gradients ran on the main
    // stream.  A caller that consumes the range on the weight-gradient stream (sync_main = false) must still
    // find it final there, so that stream waits for the main stream instead.  The same holds for the first layer's weight
    // gradient, which is issued on the main stream (above).
    if (wstream_ && !sync_main && (!overlap_ || (finished && bw_first_on_main_))) {
        HIP_OK(hipEventRecord(ev_dy_, stream_));
        HIP_OK(hipStreamWaitEvent(wstream_, ev_dy_, 0));
    }
    bw_done_off_ = lo;
    *off = lo;
    *count = hi - lo;
    return !finished;
}

std::vector<std::pair<size_t, size_t>> Net::backward_ranges(size_t min_floats) const {
    std::vector<std::pair<size_t, size_t>> out;
    std::vector<char> done(ops_.size(), 0);
    auto final_lo = [&] {
        size_t lo = nfilters_;
        for (int i = (int)ops_.size() - 1; i >= 0; --i) {
            if (ops_[i].kind != OP_CONV) continue;
            if (!done[i]) break;
            lo = ops_[i].w_off;
        }
        return lo;
    };
    size_t hi = nfilters_, pos = 0;
    bool chain_seen = false;
    while (pos < bwd_order_.size()) {
        size_t lo = hi;
        while (pos < bwd_order_.size() && hi - lo < min_floats) {      // exactly backward_step's loop
            const int i = bwd_order_[pos++];
            if (ops_[i].kind != OP_CONV) continue;
            if (chain_bwd_ && in_chain_[i]) {      // the chain's layers finish together, when its first member is met (backward_step)
                if (chain_seen && i != chain_first_) continue;
                if (!chain_seen) {
                    for (size_t j = 0; j < ops_.size(); ++j)
                        if (in_chain_[j] && (int)j != chain_first_) done[j] = 1;
                    chain_seen = true;
                    lo = final_lo();
                    if (i != chain_first_) continue;
                }
            }
            done[i] = 1;
            lo = final_lo();
        }
        if (hi > lo) out.emplace_back(lo, hi - lo);
        hi = lo;
    }
    return out;
}

void Net::set_wgrad_stream(hipStream_t s) {
    SSD_REQUIRE(training_, "handle was created with training = 0");
    if (own_wstream_ && wstream_) (void)hipStreamDestroy(wstream_);
    wstream_ = s;
    own_wstream_ = false;
    if (s2_is_w_) s2_ = s;      // the second forward lane lives on the weight-gradient stream
}

void Net::backward(int b, const float* y) {
    backward_begin(b, y);
    size_t off, count;
    while (backward_step(nparams_, &off, &count, true)) {}
}

float Net::current_lr() const {
    for (size_t i = 0; i < lr_bounds_.size(); ++i)
        if (global_step <= lr_bounds_[i]) return lr_values_[i];
    return lr_values_[lr_bounds_.size()];
}

void Net::apply_gradients(float grad_scale) {
    SSD_REQUIRE(training_, "handle was created with training = 0");
    g_prof = &prof_;
    prof_.layer = "optimizer";
    momentum_update(params_, mom_, grads_, nparams_, current_lr(), momentum_, grad_scale, stream_);
    ++global_step;
}

// backward + update of a single-GPU step.  (Rounds 2-4 carried an "early update" of the finished arena suffix on the
// weight-gradient stream: +-0.1 %, profiles/r02_h -- backward ends with conv1_1's weight gradient alone on that stream, both
// are HBM-bound, there is nothing for the update to hide behind.  Removed in round 5.)
void Net::backward_apply(int b, const float* y, float grad_scale) {
    SSD_REQUIRE(training_, "handle was created with training = 0");
    backward(b, y);
    apply_gradients(grad_scale);
}

void Net::null_gradients_step() {
    SSD_REQUIRE(training_, "handle was created with training = 0");
    null_gradients(params_, grads_, nfilters_, nparams_, wd_, stream_);
    // a step without samples has no loss kernel: its slot of the ring is written here, after the slot's previous user
    // (LOSS_RING passes ago) has finished -- no kernel in flight writes to it
    float* slot = begin_loss_slot();
    for (int i = 0; i < 4; ++i) slot[i] = 0.f;
    HIP_OK(hipEventRecord(ev_loss_[loss_seq_ % LOSS_RING], stream_));
}

void Net::set_optimizer(const float* lr_values, const long long* bounds, int n, float momentum, float wd) {
    SSD_REQUIRE(n >= 1, "need at least one learning rate value");
    lr_values_.assign(lr_values, lr_values + n);
    lr_bounds_.assign(bounds, bounds + (n - 1));
    momentum_ = momentum;
    wd_ = wd;
}

void Net::upload_xy(const float* x, const float* y, int b) {
    SSD_REQUIRE(b >= 1 && b <= Bmax_, "batch %d outside 1..%d (max_batch)", b, Bmax_);
    const size_t nx = (size_t)b * preset_->image_h * preset_->image_w * 3;
    HIP_OK(hipMemcpyAsync(x_stage_, x, nx * sizeof(float), hipMemcpyHostToDevice, stream_));
    if (y) {
        const size_t ny = (size_t)b * preset_->num_anchors * (C_ + 5);
        HIP_OK(hipMemcpyAsync(y_stage_, y, ny * sizeof(float), hipMemcpyHostToDevice, stream_));
    }
}

float* Net::begin_loss_slot() {
    ++loss_seq_;
    const int slot = (int)(loss_seq_ % LOSS_RING);
    if (loss_seq_ >= LOSS_RING) HIP_OK(hipEventSynchronize(ev_loss_[slot]));      // its previous pass (long finished)
    lw_.losses = losses_dev_ + 4 * slot;
    return losses_host_ + 4 * slot;
}

void Net::get_losses(float out[4]) {
    HIP_OK(hipStreamSynchronize(stream_));
    const int slot = loss_seq_ < 0 ? 0 : (int)(loss_seq_ % LOSS_RING);
    for (int i = 0; i < 4; ++i) out[i] = losses_host_[4 * slot + i];
}

void Net::get_losses_step(int steps_back, float out[4]) {
    SSD_REQUIRE(steps_back >= 0 && steps_back < LOSS_RING - 1, "steps_back must be in 0..%d", LOSS_RING - 2);
    SSD_REQUIRE(loss_seq_ - steps_back >= 0, "no step with a loss has run %d step(s) back", steps_back);
    const int slot = (int)((loss_seq_ - steps_back) % LOSS_RING);
    HIP_OK(hipEventSynchronize(ev_loss_[slot]));
    for (int i = 0; i < 4; ++i) out[i] = losses_host_[4 * slot + i];
}

void Net::set_result(const float* pred_dev, int b) {
    SSD_REQUIRE(b >= 1 && b <= Bmax_, "batch %d outside 1..%d", b, Bmax_);
    HIP_OK(hipMemcpyAsync(result_, pred_dev, (size_t)b * preset_->num_anchors * (C_ + 5) * sizeof(float), hipMemcpyDeviceToDevice, stream_));
}

void Net::copy_result(float* out, int b) {
    const size_t n = (size_t)b * preset_->num_anchors * (C_ + 5);
    HIP_OK(hipMemcpyAsync(out, result_, n * sizeof(float), hipMemcpyDeviceToHost, stream_));
    HIP_OK(hipStreamSynchronize(stream_));
}

// ---------------------------------------------------------------------------------
// variables
// ---------------------------------------------------------------------------------
const Variable& Net::find_var(const char* name) const {
    for (const Variable& v : vars_)
        if (v.name == name) return v;
    fail("no such variable: %s", name ? name : "(null)");
    return vars_[0];
}

void Net::load_variable(const char* name, const float* host, size_t count, int which) {
    const Variable& v = find_var(name);
    SSD_REQUIRE(count == v.count(), "variable %s holds %zu floats, got %zu", name, v.count(), count);
    float* base = which == 0 ? params_ : mom_;
    SSD_REQUIRE(base != nullptr, "arena not allocated (training = 0?)");
    HIP_OK(hipStreamSynchronize(stream_));
    HIP_OK(hipMemcpy2D(base + v.off, v.pitch * sizeof(float), host, v.width * sizeof(float), v.width * sizeof(float), v.rows,
                       hipMemcpyHostToDevice));
}

void Net::save_variable(const char* name, float* host, size_t count, int which) {
    const Variable& v = find_var(name);
    SSD_REQUIRE(count == v.count(), "variable %s holds %zu floats, got %zu", name, v.count(), count);
    const float* base = which == 0 ? params_ : (which == 1 ? grads_ : mom_);
    SSD_REQUIRE(base != nullptr, "arena not allocated (training = 0?)");
    HIP_OK(hipStreamSynchronize(stream_));
    HIP_OK(hipMemcpy2D(host, v.width * sizeof(float), base + v.off, v.pitch * sizeof(float), v.width * sizeof(float), v.rows,
                       hipMemcpyDeviceToHost));
}

void Net::activation_shape(const char* name, int* H, int* W, int* C) const {
    if (name && !strncmp(name, "grad:", 5)) name += 5;
    for (const Tensor& t : tensors_)
        if (t.name == name) {
            *H = t.H; *W = t.W; *C = t.C;
            return;
        }
    fail("no such activation: %s", name ? name : "(null)");
}

void Net::activation(const char* name, int b, float* out, size_t count) {
    // "grad:<scope>" returns d(loss)/d(pre-activation) of that layer from the last backward
    const bool want_grad = name && !strncmp(name, "grad:", 5);
    if (want_grad) name += 5;
    for (size_t ti = 0; ti < tensors_.size(); ++ti) {
        const Tensor& t = tensors_[ti];
        if (t.name != name || !t.data) continue;
        SSD_REQUIRE(!want_grad || t.grad != nullptr, "no gradient storage (training = 0?)");
        SSD_REQUIRE(!(want_grad && bw_first_fused_ && fused_first_out_ == (int)ti),
                    "gradient of %s is not materialised: the first layer's weight gradient was computed inside the next layer's data gradient (SSD_POOL_FUSE=3 keeps it)", name);
        for (const Op& op : ops_) {
            SSD_REQUIRE(!(op.kind == OP_POOL && op.fused_fwd && op.in == (int)ti && !want_grad),
                        "activation %s is not materialised: its 2x2 pool is fused into the convolution (SSD_POOL_FUSE=0 keeps it)", name);
            SSD_REQUIRE(!(op.kind == OP_POOL && op.fused_bwd && op.out == (int)ti && want_grad),
                        "gradient of %s is not materialised: the pool's backward is fused into its consumer's data gradient (SSD_POOL_FUSE=0 keeps it)", name);
        }
        SSD_REQUIRE(count == t.per_image() * b, "activation %s holds %zu floats for b=%d, got %zu", name, t.per_image() * b, b,
                    count);
        const void* src = want_grad ? t.grad : t.data;
        HIP_OK(hipStreamSynchronize(stream_));
        if (want_grad ? t.grad_f32 : t.data_f32) {
            HIP_OK(hipMemcpy(out, src, count * sizeof(float), hipMemcpyDeviceToHost));
        } else {        // bf16 storage: widen on the host (exact)
            std::vector<unsigned short> hb(count);
            HIP_OK(hipMemcpy(hb.data(), src, count * 2, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < count; ++i) {
                const unsigned u = (unsigned)hb[i] << 16;
                memcpy(out + i, &u, 4);
            }
        }
        return;
    }
    fail("no such activation: %s", name ? name : "(null)");
}

// ---------------------------------------------------------------------------------
// decode + NMS of the last result
// ---------------------------------------------------------------------------------
// Two output slots alternate, each one packed device buffer [count | conf | cls | idx | box] with a pinned
// host mirror: the kernels and ONE device-to-host copy are enqueued, an event marks the copy, and the
// caller collects a slot later (detect_fetch) -- e.g. after it has launched the next batch -- or at once.
static size_t det_count_bytes(int b) { return ((size_t)b * 4 + 255) / 256 * 256; }

void Net::detect_slot_carve(const DetectSlot& sl, DetectOut& d, char* base) const {
    const size_t n = (size_t)sl.b * sl.out_cap;
    d.count = (int*)base;
    d.conf = (float*)(base + det_count_bytes(sl.b));
    d.cls = (int*)(d.conf + n);
    d.idx = d.cls + n;
    d.box = d.idx + n;
}

const DetectSlot& Net::detect_last_dev(int b, float thr, int cap, int max_out, int out_cap, bool nms, DetectOut* dev_out) {
    SSD_REQUIRE(b >= 1 && b <= Bmax_, "batch %d outside 1..%d", b, Bmax_);
    SSD_REQUIRE(out_cap >= 1, "out_cap must be >= 1");
    const int A = preset_->num_anchors;
    if (!detect_ws_) {
        detect_ws_ = dalloc(detect_ws_bytes(Bmax_, A));
    }
    det_cur_ ^= 1;
    DetectSlot& sl = det_slot_[det_cur_];
    if (!sl.ready) {
        HIP_OK(hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
    } else {
        HIP_OK(hipEventSynchronize(sl.ready));      // the slot's previous copy must have landed before it is reused
    }
    // The survivors (a few hundred KB per batch, only count-many entries per image are written) go STRAIGHT into pinned,
    // device-mapped host memory from the per-image kernel's ordered emit -- like the four losses (alloc()): a
    // device-to-host copy of the whole [b][out_cap] arrays is a blit KERNEL of its own behind the pass (13 us per batch of
    // 128 in the rocprofv3 trace, a third of the pass) plus one more launch for the host to issue.
    const bool mapped = true;
    const size_t need = det_count_bytes(b) + (size_t)b * out_cap * 28;
    if (need > sl.bytes) {
        if (sl.dev && !sl.mapped) HIP_OK(hipFree(sl.dev));
        if (sl.host) HIP_OK(hipHostFree(sl.host));
        sl.dev = sl.host = nullptr;
        sl.bytes = 0;
        const size_t grow = std::max(need, det_count_bytes(Bmax_) + (size_t)Bmax_ * std::min(out_cap, 200) * 28);
        sl.mapped = mapped;
        if (mapped) {
            HIP_OK(hipHostMalloc((void**)&sl.host, grow, hipHostMallocMapped));
            void* dp = nullptr;
            HIP_OK(hipHostGetDevicePointer(&dp, sl.host, 0));
            sl.dev = static_cast<char*>(dp);
        } else {
            HIP_OK(hipMalloc((void**)&sl.dev, grow));
            HIP_OK(hipHostMalloc((void**)&sl.host, grow));
        }
        sl.bytes = grow;
    }
    sl.b = b; sl.out_cap = out_cap; sl.used = need;
    g_prof = &prof_;
    prof_.layer = "detect";
    DetectOut d;
    detect_slot_carve(sl, d, sl.dev);
    detect(A, C_, anchors_dev_, result_, b, thr, cap, max_out, out_cap, nms, d, detect_ws_, stream_);
    if (!sl.mapped) HIP_OK(hipMemcpyAsync(sl.host, sl.dev, need, hipMemcpyDeviceToHost, stream_));
    HIP_OK(hipEventRecord(sl.ready, stream_));
    if (dev_out) *dev_out = d;
    return sl;
}

void Net::detect_fetch(int which, int* count, float* conf, int* cls, int* idx, int* box) {
    SSD_REQUIRE(which == 0 || which == 1, "which must be 0 (latest) or 1 (the one before)");
    DetectSlot& sl = det_slot_[det_cur_ ^ which];
    SSD_REQUIRE(sl.ready != nullptr && sl.b > 0, "no detection pass in that slot");
    HIP_OK(hipEventSynchronize(sl.ready));
    DetectOut h;
    detect_slot_carve(sl, h, sl.host);
    const size_t n = (size_t)sl.b * sl.out_cap;
    if (count) memcpy(count, h.count, (size_t)sl.b * 4);
    if (conf) memcpy(conf, h.conf, n * 4);
    if (cls) memcpy(cls, h.cls, n * 4);
    if (idx) memcpy(idx, h.idx, n * 4);
    if (box) memcpy(box, h.box, n * 16);
}

void Net::detect_host(int which, DetectOut* host, int* b, int* out_cap) {
    SSD_REQUIRE(which == 0 || which == 1, "which must be 0 (latest) or 1 (the one before)");
    SSD_REQUIRE(host != nullptr, "null argument");
    DetectSlot& sl = det_slot_[det_cur_ ^ which];
    SSD_REQUIRE(sl.ready != nullptr && sl.b > 0, "no detection pass in that slot");
    HIP_OK(hipEventSynchronize(sl.ready));
    detect_slot_carve(sl, *host, sl.host);
    if (b) *b = sl.b;
    if (out_cap) *out_cap = sl.out_cap;
}

void Net::detect_last(int b, float thr, int cap, int max_out, int out_cap, bool nms, int* count, float* conf, int* cls,
                      int* idx, int* box) {
    detect_last_dev(b, thr, cap, max_out, out_cap, nms, nullptr);
    detect_fetch(0, count, conf, cls, idx, box);
}

}  // namespace ssd
