// extern "C" boundary of libssdvgg_hip.so (include/ssdvgg_hip.h).
#include "../../include/ssdvgg_hip.h"
#include "net.h"
#include "augment.h"
#include "metrics.h"
#include <vector>
#include <map>
#include <mutex>
#include <algorithm>

namespace ssd {
static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}
const char* get_error() { return g_err.c_str(); }

thread_local Profiler* g_prof = nullptr;
thread_local hipEvent_t g_stop_event = nullptr;
hipEvent_t Profiler::get() {
    if (used == pool.size()) {
        hipEvent_t e;
        HIP_OK(hipEventCreate(&e));
        pool.push_back(e);
    }
    return pool[used++];
}
Profiler::~Profiler() {
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
}
ProfScope::ProfScope(const char* kernel, double flops, double bytes, hipStream_t stream) : s(stream), active(false), idx(0) {
    Profiler* p = g_prof;
    if (!p || !p->on) return;
    Profiler::Rec r{p->detailed ? std::string(kernel) + ":" + p->layer : std::string(kernel), flops, bytes, p->get(), p->get()};
    idx = p->recs.size();
    p->recs.push_back(r);
    active = true;
    (void)hipEventRecord(r.e0, s);
}
ProfScope::~ProfScope() {
    if (active && g_prof) (void)hipEventRecord(g_prof->recs[idx].e1, s);
}
}  // namespace ssd

using namespace ssd;

struct ssd_net {
    Net* net;
};

#define API_BEGIN try {
#define API_END                                  \
    return 0;                                    \
    }                                            \
    catch (const std::exception& e) {            \
        ssd::set_error("%s", e.what());          \
        return 1;                                \
    }                                            \
    catch (...) {                                \
        ssd::set_error("unknown error");         \
        return 1;                                \
    }

static Net& N(ssd_handle h) {
    if (!h || !h->net) fail("null handle");
    return *h->net;
}
// handle-based entry points: the handle's GPU is current for the call, the caller's device is restored on return
#define API_BEGIN_NET(h) \
    try {                \
        Net& n = N(h);   \
        DeviceGuard dev_guard_(n.device());

namespace {
struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) { HIP_OK(hipMalloc(&p, bytes ? bytes : 16)); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <typename T> T* as() { return (T*)p; }
};
}  // namespace

// Label encoding is called once per training batch: the anchor tables of a (preset, device) pair and a small
// growable scratch per device are kept for the life of the process instead of six hipMalloc / hipFree pairs
// (each one a device synchronisation) per call.
namespace {
struct EncodeCache {
    struct Anchors { double* anc = nullptr; int* aabs = nullptr; };
    struct Scratch { char* p = nullptr; size_t bytes = 0; };
    std::mutex mu;
    std::map<std::pair<std::string, int>, Anchors> anchors;
    std::map<int, Scratch> scratch;
};
EncodeCache& encode_cache() {
    static EncodeCache* c = new EncodeCache();      // leaked on purpose: no HIP calls in static destructors
    return *c;
}
}  // namespace

extern "C" {

const char* ssd_last_error(void) { return ssd::get_error(); }
const char* ssd_version(void) { return "ssdvgg_hip 0.1 gfx950 fp32-mfma(v_mfma_f32_32x32x2_f32) bf16-mfma(v_mfma_f32_32x32x16_bf16)"; }

int ssd_preset_info(const char* preset, int* image_w, int* image_h, int* num_anchors, int* num_maps) {
    API_BEGIN
    const Preset& p = get_preset(preset);
    if (image_w) *image_w = p.image_w;
    if (image_h) *image_h = p.image_h;
    if (num_anchors) *num_anchors = p.num_anchors;
    if (num_maps) *num_maps = p.nmaps;
    API_END
}

int ssd_preset_map(const char* preset, int map, int* size, double* scale, int* num_types) {
    API_BEGIN
    const Preset& p = get_preset(preset);
    SSD_REQUIRE(map >= 0 && map < p.nmaps, "map %d outside 0..%d", map, p.nmaps - 1);
    if (size) *size = p.map_size[map];
    if (scale) *scale = p.scale[map];
    if (num_types) *num_types = p.ntypes[map];
    API_END
}

int ssd_anchors_dev(const char* preset, double* anchors_dev, int* anchors_abs_dev, void* stream) {
    API_BEGIN
    anchors_device(get_preset(preset), anchors_dev, anchors_abs_dev, (hipStream_t)stream);
    API_END
}

static void anchors_host(const char* preset, int device, double* out, int* out_abs) {
    const Preset& p = get_preset(preset);
    DeviceGuard dev_guard_(device);
    const size_t A = p.num_anchors;
    DevBuf a(A * 4 * sizeof(double)), b(A * 4 * sizeof(int));
    anchors_device(p, a.as<double>(), b.as<int>(), nullptr);
    if (out) HIP_OK(hipMemcpy(out, a.p, A * 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (out_abs) HIP_OK(hipMemcpy(out_abs, b.p, A * 4 * sizeof(int), hipMemcpyDeviceToHost));
}

int ssd_anchors(const char* preset, int device, double* out) {
    API_BEGIN
    anchors_host(preset, device, out, nullptr);
    API_END
}

int ssd_anchors_abs(const char* preset, int device, int* out) {
    API_BEGIN
    anchors_host(preset, device, nullptr, out);
    API_END
}

int ssd_jaccard_overlap(int device, const double* box, const double* boxes, int n, double* iou_out) {
    API_BEGIN
    SSD_REQUIRE(n >= 0, "n must be >= 0");
    if (n > 0) {
        DeviceGuard dev_guard_(device);
        DevBuf db(4 * sizeof(double)), da((size_t)n * 4 * sizeof(double)), di((size_t)n * sizeof(double));
        HIP_OK(hipMemcpy(db.p, box, 4 * sizeof(double), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(da.p, boxes, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice));
        jaccard_device(db.as<double>(), da.as<double>(), n, di.as<double>(), nullptr);
        HIP_OK(hipMemcpy(iou_out, di.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    }
    API_END
}

// the preset's anchor tables on `device`, built once (caller holds ec.mu)
static EncodeCache::Anchors& encode_anchors(EncodeCache& ec, const Preset& p, int device) {
    const size_t A = p.num_anchors;
    EncodeCache::Anchors& an = ec.anchors[{p.name, device}];
    if (!an.anc) {
        // built into locals and published only once complete: a failure half way must not leave a table that looks ready
        double* anc = nullptr;
        int* aabs = nullptr;
        try {
            HIP_OK(hipMalloc((void**)&anc, A * 4 * sizeof(double)));
            HIP_OK(hipMalloc((void**)&aabs, A * 4 * sizeof(int)));
            anchors_device(p, anc, aabs, nullptr);
            HIP_OK(hipDeviceSynchronize());
        } catch (...) {
            if (anc) (void)hipFree(anc);
            if (aabs) (void)hipFree(aabs);
            throw;
        }
        an.aabs = aabs;
        an.anc = anc;
    }
    return an;
}

static void encode_labels_impl(const char* preset, int num_classes, int device, const double* gt, const int* cls,
                               const int* offsets, int b, float* vec_dev, float* vec_host, hipStream_t s) {
    const Preset& p = get_preset(preset);
    SSD_REQUIRE(b >= 1, "batch must be >= 1");
    SSD_REQUIRE(num_classes >= 1 && num_classes <= 27, "num_classes must be in 1..27");
    DeviceGuard dev_guard_(device);
    const int ntot = offsets[b];
    SSD_REQUIRE(offsets[0] == 0 && ntot >= 0, "gt_offsets must start at 0 and be non-decreasing");
    for (int i = 0; i < ntot; ++i)
        SSD_REQUIRE(cls[i] >= 0 && cls[i] < num_classes, "gt_cls[%d] = %d outside 0..%d", i, cls[i], num_classes - 1);
    const size_t A = p.num_anchors;
    EncodeCache& ec = encode_cache();
    std::lock_guard<std::mutex> lock(ec.mu);
    EncodeCache::Anchors& an = encode_anchors(ec, p, device);
    const size_t nt = ntot ? ntot : 1;
    const size_t n = (size_t)b * A * (num_classes + 5);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_gt = 0, o_cls = o_gt + up(nt * 4 * sizeof(double)), o_off = o_cls + up(nt * sizeof(int)),
                 o_ws = o_off + up((size_t)(b + 1) * sizeof(int)), o_tmp = o_ws + up(encode_labels_ws_bytes(ntot)),
                 need = o_tmp + (vec_dev ? 0 : n * sizeof(float));
    EncodeCache::Scratch& sc = ec.scratch[device];
    if (need > sc.bytes) {
        if (sc.p) HIP_OK(hipFree(sc.p));
        sc.p = nullptr; sc.bytes = 0;
        HIP_OK(hipMalloc((void**)&sc.p, need));
        sc.bytes = need;
    }
    double* dgt = (double*)(sc.p + o_gt);
    int* dcls = (int*)(sc.p + o_cls);
    int* doff = (int*)(sc.p + o_off);
    float* out = vec_dev ? vec_dev : (float*)(sc.p + o_tmp);
    try {
        if (ntot) {
            HIP_OK(hipMemcpyAsync(dgt, gt, (size_t)ntot * 4 * sizeof(double), hipMemcpyHostToDevice, s));
            HIP_OK(hipMemcpyAsync(dcls, cls, (size_t)ntot * sizeof(int), hipMemcpyHostToDevice, s));
        }
        HIP_OK(hipMemcpyAsync(doff, offsets, (size_t)(b + 1) * sizeof(int), hipMemcpyHostToDevice, s));
        encode_labels(p, num_classes, an.anc, an.aabs, dgt, dcls, doff, b, ntot, out, sc.p + o_ws, s);
        if (vec_host) HIP_OK(hipMemcpyAsync(vec_host, out, n * sizeof(float), hipMemcpyDeviceToHost, s));
    } catch (...) {
        (void)hipStreamSynchronize(s);   // nothing of this call may still be using the shared scratch when the lock is released
        throw;
    }
    HIP_OK(hipStreamSynchronize(s));     // the shared scratch is free for the next call
}

int ssd_encode_labels(const char* preset, int num_classes, int device, const double* gt_boxes, const int* gt_cls,
                      const int* gt_offsets, int b, float* vec_out) {
    API_BEGIN
    encode_labels_impl(preset, num_classes, device, gt_boxes, gt_cls, gt_offsets, b, nullptr, vec_out, nullptr);
    API_END
}

int ssd_encode_labels_dev(const char* preset, int num_classes, int device, const double* gt_boxes, const int* gt_cls,
                          const int* gt_offsets, int b, float* vec_out_dev, void* stream) {
    API_BEGIN
    encode_labels_impl(preset, num_classes, device, gt_boxes, gt_cls, gt_offsets, b, vec_out_dev, nullptr, (hipStream_t)stream);
    API_END
}

size_t ssd_encode_labels_ws_bytes(int ntot) { return (encode_labels_ws_bytes(ntot) + 255) / 256 * 256; }

int ssd_encode_labels_resident(const char* preset, int num_classes, int device, const double* gt_boxes_dev, const int* gt_cls_dev,
                               const int* gt_offsets_dev, int b, int ntot, float* vec_out_dev, void* ws_dev, void* stream) {
    API_BEGIN
    const Preset& p = get_preset(preset);
    SSD_REQUIRE(b >= 1 && ntot >= 0, "batch must be >= 1 and ntot >= 0");
    SSD_REQUIRE(num_classes >= 1 && num_classes <= 27, "num_classes must be in 1..27");
    SSD_REQUIRE(gt_offsets_dev && vec_out_dev && ws_dev && (ntot == 0 || (gt_boxes_dev && gt_cls_dev)), "null argument");
    DeviceGuard dev_guard_(device);
    const double* anc;
    const int* aabs;
    {
        EncodeCache& ec = encode_cache();
        std::lock_guard<std::mutex> lock(ec.mu);
        EncodeCache::Anchors& an = encode_anchors(ec, p, device);
        anc = an.anc; aabs = an.aabs;
    }
    encode_labels(p, num_classes, anc, aabs, gt_boxes_dev, gt_cls_dev, gt_offsets_dev, b, ntot, vec_out_dev, ws_dev, (hipStream_t)stream);
    API_END
}

size_t ssd_decode_nms_ws_bytes(const char* preset, int b) {
    try {
        return detect_ws_bytes(b, get_preset(preset).num_anchors);
    } catch (const std::exception& e) {
        ssd::set_error("%s", e.what());
        return 0;
    }
}

int ssd_decode_nms_dev(const char* preset, int num_classes, const double* anchors_dev, const float* pred_dev, int b,
                       float conf_thr, int cap, int max_out, int out_cap, int nms, int* count_dev, float* conf_dev,
                       int* cls_dev, int* idx_dev, int* box_dev, void* ws_dev, void* stream) {
    API_BEGIN
    const Preset& p = get_preset(preset);
    DetectOut o{count_dev, conf_dev, cls_dev, idx_dev, box_dev};
    detect(p.num_anchors, num_classes, anchors_dev, pred_dev, b, conf_thr, cap, max_out, out_cap, nms != 0, o, ws_dev,
           (hipStream_t)stream);
    API_END
}

int ssd_decode_nms(const char* preset, int num_classes, int device, const float* pred, int b, float conf_thr, int cap,
                   int max_out, int out_cap, int nms, int* count, float* conf, int* cls, int* idx, int* box) {
    API_BEGIN
    const Preset& p = get_preset(preset);
    SSD_REQUIRE(b >= 1 && out_cap >= 1, "batch and out_cap must be >= 1");
    DeviceGuard dev_guard_(device);
    const size_t A = p.num_anchors, nv = num_classes + 5, n = (size_t)b * out_cap;
    DevBuf anc(A * 4 * sizeof(double)), dpred((size_t)b * A * nv * sizeof(float)), ws(detect_ws_bytes(b, (int)A));
    DevBuf dcount((size_t)b * 4), dconf(n * 4), dcls(n * 4), didx(n * 4), dbox(n * 16);
    anchors_device(p, anc.as<double>(), nullptr, nullptr);
    HIP_OK(hipMemcpy(dpred.p, pred, (size_t)b * A * nv * sizeof(float), hipMemcpyHostToDevice));
    DetectOut o{dcount.as<int>(), dconf.as<float>(), dcls.as<int>(), didx.as<int>(), dbox.as<int>()};
    detect((int)A, num_classes, anc.as<double>(), dpred.as<float>(), b, conf_thr, cap, max_out, out_cap, nms != 0, o, ws.p, nullptr);
    HIP_OK(hipMemcpy(count, dcount.p, (size_t)b * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(conf, dconf.p, n * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(cls, dcls.p, n * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(idx, didx.p, n * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(box, dbox.p, n * 16, hipMemcpyDeviceToHost));
    API_END
}

int ssd_nms_boxes(int device, int n, const int* box_abs, const float* conf, const int* group, double iou_thr, int* keep_out,
                  int* n_keep) {
    API_BEGIN
    SSD_REQUIRE(n >= 0 && n <= 65535, "ssd_nms_boxes: 0..65535 boxes (got %d)", n);
    SSD_REQUIRE(n_keep != nullptr, "n_keep is null");
    *n_keep = 0;
    if (n > 0) {
        SSD_REQUIRE(box_abs && conf && keep_out, "null argument");
        int ngroups = 1;
        if (group)
            for (int i = 0; i < n; ++i) {
                SSD_REQUIRE(group[i] >= 0 && group[i] < 65535, "group[%d] = %d outside 0..65534", i, group[i]);
                ngroups = std::max(ngroups, group[i] + 1);
            }
        DeviceGuard dev_guard_(device);
        DevBuf dbox((size_t)n * 16), dconf((size_t)n * 4), dgrp((size_t)n * 4), dkeep((size_t)n * 4 + 4), ws(nms_boxes_ws_bytes(n, ngroups));
        HIP_OK(hipMemcpy(dbox.p, box_abs, (size_t)n * 16, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dconf.p, conf, (size_t)n * 4, hipMemcpyHostToDevice));
        if (group) HIP_OK(hipMemcpy(dgrp.p, group, (size_t)n * 4, hipMemcpyHostToDevice));
        else HIP_OK(hipMemset(dgrp.p, 0, (size_t)n * 4));
        nms_boxes_device(n, ngroups, dbox.as<int>(), dconf.as<float>(), dgrp.as<int>(), iou_thr, dkeep.as<int>(), ws.p, nullptr);
        std::vector<int> tmp((size_t)n + 1);
        HIP_OK(hipMemcpy(tmp.data(), dkeep.p, (size_t)n * 4 + 4, hipMemcpyDeviceToHost));
        *n_keep = tmp[0];
        for (int i = 0; i < tmp[0]; ++i) keep_out[i] = tmp[1 + i];
    }
    API_END
}

int ssd_average_precision(int device, int n_det, const float* det_box, const float* det_conf, const int* det_cls,
                          const int* det_sample, int n_gt, const double* gt_box, const int* gt_cls, const int* gt_sample,
                          int num_classes, double minoverlap, double* ap_out, int* present_out) {
    API_BEGIN
    SSD_REQUIRE(num_classes >= 1 && n_det >= 0 && n_gt >= 0, "bad sizes");
    for (int g = 1; g < n_gt; ++g) SSD_REQUIRE(gt_sample[g] >= gt_sample[g - 1], "ground truth must be grouped by ascending sample id");
    DeviceGuard dev_guard_(device);
    int n2 = 1;
    while (n2 < n_det) n2 <<= 1;
    const size_t nd = n_det ? n_det : 1, ng = n_gt ? n_gt : 1;
    DevBuf dbox(nd * 16), dconf(nd * 4), dcls(nd * 4), dsmp(nd * 4), gbox(ng * 32), gcls(ng * 4), gsmp(ng * 4);
    DevBuf keys((size_t)num_classes * n2 * 8), matched(ng), ap((size_t)num_classes * 8), present((size_t)num_classes * 4);
    if (n_det) {
        HIP_OK(hipMemcpy(dbox.p, det_box, nd * 16, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dconf.p, det_conf, nd * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dcls.p, det_cls, nd * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dsmp.p, det_sample, nd * 4, hipMemcpyHostToDevice));
    }
    if (n_gt) {
        HIP_OK(hipMemcpy(gbox.p, gt_box, ng * 32, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(gcls.p, gt_cls, ng * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(gsmp.p, gt_sample, ng * 4, hipMemcpyHostToDevice));
    }
    average_precision_device(n_det, n_gt, num_classes, dbox.as<float>(), dconf.as<float>(), dcls.as<int>(), dsmp.as<int>(),
                             gbox.as<double>(), gcls.as<int>(), gsmp.as<int>(), minoverlap, keys.as<unsigned long long>(), n2,
                             matched.as<unsigned char>(), ap.as<double>(), present.as<int>(), nullptr);
    HIP_OK(hipMemcpy(ap_out, ap.p, (size_t)num_classes * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(present_out, present.p, (size_t)num_classes * 4, hipMemcpyDeviceToHost));
    API_END
}

// ------------------------------------------------------------------------------ model
size_t ssd_arena_floats(const char* preset, int num_classes) {
    try {
        return Net::arena_floats(preset, num_classes);
    } catch (const std::exception& e) {
        ssd::set_error("%s", e.what());
        return 0;
    }
}

size_t ssd_augment_ws_bytes(int b, int out_w, int out_h) { return augment_ws_bytes(b, out_w, out_h); }

int ssd_augment_batch_dev(const unsigned char* images_dev, const ssd_augment_params* params, int b, int out_w, int out_h,
                          float* out_dev, void* ws_dev, void* stream) {
    API_BEGIN
    SSD_REQUIRE(images_dev && params && out_dev && ws_dev, "null argument");
    augment_batch(images_dev, params, b, out_w, out_h, out_dev, ws_dev, (hipStream_t)stream);
    API_END
}

int ssd_create(const char* preset, int num_classes, int max_batch, int device, int training, unsigned long long seed,
               float* ext_params_dev, float* ext_grads_dev, float* ext_momentum_dev, ssd_handle* out) {
    API_BEGIN
    SSD_REQUIRE(out != nullptr, "out handle pointer is null");
    *out = nullptr;
    DeviceGuard dev_guard_(device);
    Net* n = new Net(preset, num_classes, max_batch, device, training != 0, seed, ext_params_dev, ext_grads_dev,
                     ext_momentum_dev);
    SSD_REQUIRE(n->nparams() == Net::arena_floats(preset, num_classes), "arena size mismatch");
    *out = new ssd_net{n};
    API_END
}

int ssd_create_dtype(const char* preset, int num_classes, int max_batch, int device, int training, unsigned long long seed,
                     float* ext_params_dev, float* ext_grads_dev, float* ext_momentum_dev, int dtype, ssd_handle* out) {
    API_BEGIN
    SSD_REQUIRE(out != nullptr, "out handle pointer is null");
    *out = nullptr;
    DeviceGuard dev_guard_(device);
    Net* n = new Net(preset, num_classes, max_batch, device, training != 0, seed, ext_params_dev, ext_grads_dev,
                     ext_momentum_dev, dtype);
    SSD_REQUIRE(n->nparams() == Net::arena_floats(preset, num_classes), "arena size mismatch");
    *out = new ssd_net{n};
    API_END
}

int ssd_get_dtype(ssd_handle h, int* dtype) {
    API_BEGIN_NET(h)
    SSD_REQUIRE(h != nullptr && dtype != nullptr, "null argument");
    *dtype = n.dtype();
    API_END
}

int ssd_destroy(ssd_handle h) {
    API_BEGIN
    if (h) {
        delete h->net;
        delete h;
    }
    API_END
}

int ssd_set_stream(ssd_handle h, void* stream) {
    API_BEGIN_NET(h)
    n.set_stream((hipStream_t)stream);
    API_END
}

int ssd_num_variables(ssd_handle h) {
    try {
        return (int)N(h).variables().size();
    } catch (const std::exception& e) {
        ssd::set_error("%s", e.what());
        return -1;
    }
}

int ssd_variable_info(ssd_handle h, int i, char* name, int name_cap, int* ndim, int shape[4]) {
    API_BEGIN_NET(h)
    const auto& vars = n.variables();
    SSD_REQUIRE(i >= 0 && i < (int)vars.size(), "variable index %d outside 0..%zu", i, vars.size() - 1);
    const Variable& v = vars[i];
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", v.name.c_str());
    if (ndim) *ndim = v.ndim;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = v.shape[k];
    API_END
}

int ssd_load_variable(ssd_handle h, const char* name, const float* data, size_t count) {
    API_BEGIN_NET(h)
    n.load_variable(name, data, count, 0);
    API_END
}
int ssd_save_variable(ssd_handle h, const char* name, float* data, size_t count) {
    API_BEGIN_NET(h)
    n.save_variable(name, data, count, 0);
    API_END
}
int ssd_save_gradient(ssd_handle h, const char* name, float* data, size_t count) {
    API_BEGIN_NET(h)
    n.save_variable(name, data, count, 1);
    API_END
}
int ssd_save_momentum(ssd_handle h, const char* name, float* data, size_t count) {
    API_BEGIN_NET(h)
    n.save_variable(name, data, count, 2);
    API_END
}
int ssd_load_momentum(ssd_handle h, const char* name, const float* data, size_t count) {
    API_BEGIN_NET(h)
    n.load_variable(name, data, count, 2);
    API_END
}

int ssd_set_optimizer(ssd_handle h, const float* lr_values, const long long* lr_boundaries, int n_values, float momentum,
                      float weight_decay) {
    API_BEGIN_NET(h)
    n.set_optimizer(lr_values, lr_boundaries, n_values, momentum, weight_decay);
    API_END
}
int ssd_get_global_step(ssd_handle h, long long* step) {
    API_BEGIN_NET(h)
    *step = n.global_step;
    API_END
}
int ssd_set_global_step(ssd_handle h, long long step) {
    API_BEGIN_NET(h)
    n.global_step = step;
    API_END
}

int ssd_forward_backward_dev(ssd_handle h, const float* x_dev, const float* y_dev, int b) {
    API_BEGIN_NET(h)
    n.forward(x_dev, b, true, y_dev);
    n.backward(b, y_dev);
    API_END
}
int ssd_forward_dev(ssd_handle h, const float* x_dev, const float* y_dev, int b) {
    API_BEGIN_NET(h)
    n.forward(x_dev, b, true, y_dev);
    API_END
}
int ssd_backward_begin_dev(ssd_handle h, const float* y_dev, int b) {
    API_BEGIN_NET(h)
    n.backward_begin(b, y_dev);
    API_END
}
int ssd_backward_next_dev(ssd_handle h, size_t min_floats, int sync_main, size_t* offset, size_t* count, int* more) {
    API_BEGIN_NET(h)
    SSD_REQUIRE(offset && count && more, "null output pointer");
    *more = n.backward_step(min_floats, offset, count, sync_main != 0) ? 1 : 0;
    API_END
}
int ssd_backward_ranges(ssd_handle h, size_t min_floats, size_t* offsets, size_t* counts, int cap, int* n_out) {
    API_BEGIN_NET(h)
    SSD_REQUIRE(n_out != nullptr, "null argument");
    const auto r = n.backward_ranges(min_floats);
    *n_out = (int)r.size();
    for (int i = 0; i < (int)r.size() && i < cap; ++i) {
        if (offsets) offsets[i] = r[i].first;
        if (counts) counts[i] = r[i].second;
    }
    API_END
}
int ssd_set_wgrad_stream(ssd_handle h, void* stream) {
    API_BEGIN_NET(h)
    n.set_wgrad_stream((hipStream_t)stream);
    API_END
}
int ssd_apply_gradients_dev(ssd_handle h, float grad_scale) {
    API_BEGIN_NET(h)
    n.apply_gradients(grad_scale);
    API_END
}
int ssd_set_loss_normalizer(ssd_handle h, float batch) {
    API_BEGIN_NET(h)
    n.set_loss_normalizer(batch);
    API_END
}
int ssd_null_gradients_dev(ssd_handle h) {
    API_BEGIN_NET(h)
    n.null_gradients_step();
    API_END
}
int ssd_train_step_dev(ssd_handle h, const float* x_dev, const float* y_dev, int b) {
    API_BEGIN_NET(h)
    n.forward(x_dev, b, true, y_dev);
    n.backward_apply(b, y_dev, 1.f);
    API_END
}
int ssd_eval_step_dev(ssd_handle h, const float* x_dev, const float* y_dev, int b) {
    API_BEGIN_NET(h)
    n.forward(x_dev, b, true, y_dev);
    API_END
}
int ssd_infer_dev(ssd_handle h, const float* x_dev, int b) {
    API_BEGIN_NET(h)
    n.forward(x_dev, b, false, nullptr);
    API_END
}
int ssd_result_dev(ssd_handle h, const float** result_dev) {
    API_BEGIN_NET(h)
    *result_dev = n.result();
    API_END
}
int ssd_set_result_dev(ssd_handle h, const float* pred_dev, int b) {
    API_BEGIN_NET(h)
    n.set_result(pred_dev, b);
    API_END
}
int ssd_get_result(ssd_handle h, int b, float* result_out) {
    API_BEGIN_NET(h)
    SSD_REQUIRE(h != nullptr && result_out != nullptr, "null argument");
    SSD_REQUIRE(b >= 1 && b <= n.max_batch(), "batch %d outside 1..%d", b, n.max_batch());
    n.copy_result(result_out, b);
    API_END
}

int ssd_get_losses(ssd_handle h, float losses_out[4]) {
    API_BEGIN_NET(h)
    n.get_losses(losses_out);
    API_END
}
int ssd_get_losses_step(ssd_handle h, int steps_back, float losses_out[4]) {
    API_BEGIN_NET(h)
    SSD_REQUIRE(losses_out != nullptr, "null argument");
    n.get_losses_step(steps_back, losses_out);
    API_END
}
int ssd_arenas(ssd_handle h, float** params_dev, float** grads_dev, float** momentum_dev, size_t* floats,
               size_t* filter_floats) {
    API_BEGIN_NET(h)
    if (params_dev) *params_dev = n.params();
    if (grads_dev) *grads_dev = n.grads();
    if (momentum_dev) *momentum_dev = n.momentum();
    if (floats) *floats = n.nparams();
    if (filter_floats) *filter_floats = n.nfilters();
    API_END
}

int ssd_train_step(ssd_handle h, const float* x, const float* y, int b, float* result_out, float losses_out[4]) {
    API_BEGIN_NET(h)
    n.upload_xy(x, y, b);
    n.forward(n.x_stage(), b, true, n.y_stage());
    n.backward_apply(b, n.y_stage(), 1.f);
    if (result_out) n.copy_result(result_out, b);
    if (losses_out) n.get_losses(losses_out);
    HIP_OK(hipStreamSynchronize(n.stream()));
    API_END
}
int ssd_eval_step(ssd_handle h, const float* x, const float* y, int b, float* result_out, float losses_out[4]) {
    API_BEGIN_NET(h)
    n.upload_xy(x, y, b);
    n.forward(n.x_stage(), b, true, n.y_stage());
    if (result_out) n.copy_result(result_out, b);
    if (losses_out) n.get_losses(losses_out);
    HIP_OK(hipStreamSynchronize(n.stream()));
    API_END
}
int ssd_infer(ssd_handle h, const float* x, int b, float* result_out) {
    API_BEGIN_NET(h)
    n.upload_xy(x, nullptr, b);
    n.forward(n.x_stage(), b, false, nullptr);
    if (result_out) n.copy_result(result_out, b);
    HIP_OK(hipStreamSynchronize(n.stream()));
    API_END
}

int ssd_detect_last(ssd_handle h, int b, float conf_thr, int cap, int max_out, int out_cap, int nms, int* count, float* conf,
                    int* cls, int* idx, int* box) {
    API_BEGIN_NET(h)
    n.detect_last(b, conf_thr, cap, max_out, out_cap, nms != 0, count, conf, cls, idx, box);
    API_END
}

int ssd_detect_last_dev(ssd_handle h, int b, float conf_thr, int cap, int max_out, int out_cap, int nms, int** count_dev,
                        float** conf_dev, int** cls_dev, int** idx_dev, int** box_dev) {
    API_BEGIN_NET(h)
    DetectOut d{};
    n.detect_last_dev(b, conf_thr, cap, max_out, out_cap, nms != 0, &d);
    if (count_dev) *count_dev = d.count;
    if (conf_dev) *conf_dev = d.conf;
    if (cls_dev) *cls_dev = d.cls;
    if (idx_dev) *idx_dev = d.idx;
    if (box_dev) *box_dev = d.box;
    API_END
}

int ssd_detect_fetch(ssd_handle h, int which, int* count, float* conf, int* cls, int* idx, int* box) {
    API_BEGIN_NET(h)
    n.detect_fetch(which, count, conf, cls, idx, box);
    API_END
}

int ssd_detect_host(ssd_handle h, int which, const int** count, const float** conf, const int** cls, const int** idx,
                    const int** box, int* b, int* out_cap) {
    API_BEGIN_NET(h)
    DetectOut d;
    n.detect_host(which, &d, b, out_cap);
    if (count) *count = d.count;
    if (conf) *conf = d.conf;
    if (cls) *cls = d.cls;
    if (idx) *idx = d.idx;
    if (box) *box = d.box;
    API_END
}

int ssd_set_overlap(ssd_handle h, int on) {
    API_BEGIN_NET(h)
    n.set_overlap(on != 0);
    API_END
}

int ssd_profile_enable(ssd_handle h, int on) {
    API_BEGIN_NET(h)
    n.profiler().on = on != 0;
    n.profiler().detailed = on == 2;
    n.profiler().reset();
    API_END
}

// one line per kernel label: "label\tlaunches\ttotal_ms\ttotal_flops\ttotal_bytes\n"
int ssd_profile_report(ssd_handle h, char* buf, size_t cap) {
    API_BEGIN_NET(h)
    HIP_OK(hipStreamSynchronize(n.stream()));
    struct Agg { long long cnt = 0; double ms = 0, fl = 0, by = 0; };
    std::map<std::string, Agg> agg;
    for (const auto& r : n.profiler().recs) {
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, r.e0, r.e1));
        Agg& a = agg[r.kernel];
        a.cnt++; a.ms += ms; a.fl += r.flops; a.by += r.bytes;
    }
    std::string out;
    char line[384];
    for (const auto& kv : agg) {
        snprintf(line, sizeof line, "%s\t%lld\t%.6f\t%.6e\t%.6e\n", kv.first.c_str(), kv.second.cnt, kv.second.ms, kv.second.fl,
                 kv.second.by);
        out += line;
    }
    SSD_REQUIRE(buf && cap > out.size(), "report needs %zu bytes", out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    n.profiler().reset();
    API_END
}

int ssd_activation_shape(ssd_handle h, const char* name, int* height, int* width, int* channels) {
    API_BEGIN_NET(h)
    n.activation_shape(name, height, width, channels);
    API_END
}

int ssd_activation(ssd_handle h, const char* name, int b, float* out, size_t count) {
    API_BEGIN_NET(h)
    n.activation(name, b, out, count);
    API_END
}

// --------------------------------------------------------------------- single kernels
static ConvDesc mk(int b, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil, int pad_h,
                   int pad_w) {
    ConvDesc d;
    d.B = b; d.Hi = hi; d.Wi = wi; d.Ci = ci; d.Ho = ho; d.Wo = wo; d.Co = co;
    d.KH = kh; d.KW = kw; d.stride = stride; d.dil = dil; d.pad_h = pad_h; d.pad_w = pad_w;
    return d;
}

int ssd_op_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int b, int hi, int wi, int ci, int ho,
                      int wo, int co, int kh, int kw, int stride, int dil, int pad_h, int pad_w, int relu, void* stream) {
    API_BEGIN
    conv_fwd(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w), x, w, bias, y, relu != 0, (hipStream_t)stream);
    API_END
}
int ssd_op_conv2d_dgrad(const float* dy, const float* w, float* dx, const float* mask, int accumulate, int b, int hi, int wi,
                        int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil, int pad_h, int pad_w,
                        void* stream) {
    API_BEGIN
    conv_dgrad(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w), dy, w, dx, mask, accumulate != 0,
               (hipStream_t)stream);
    API_END
}
int ssd_op_cast_filter(const float* w, void* w_io_bf16, void* w_oi_bf16, int taps, int ci, int co, void* stream) {
    API_BEGIN
    FilterCastPlan plan;
    plan.add(0, taps, ci, co);
    cast_filters(plan, w, (bf16_t*)w_io_bf16, (bf16_t*)w_oi_bf16, (hipStream_t)stream);
    API_END
}
int ssd_op_conv2d_fwd_bf16(const void* x, const void* w_oi, const float* bias, void* y, int y_f32, int b, int hi, int wi, int ci,
                           int ho, int wo, int co, int kh, int kw, int stride, int dil, int pad_h, int pad_w, int relu,
                           void* stream) {
    API_BEGIN
    conv_fwd_bf16(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w), (const bf16_t*)x, (const bf16_t*)w_oi, bias, y,
                  y_f32 != 0, relu != 0, (hipStream_t)stream);
    API_END
}
int ssd_op_conv2d_dgrad_bf16(const void* dy, const void* w_io, void* dx, const void* mask, int accumulate, int b, int hi, int wi,
                             int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil, int pad_h, int pad_w,
                             void* stream) {
    API_BEGIN
    conv_dgrad_bf16(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w), (const bf16_t*)dy, (const bf16_t*)w_io,
                    (bf16_t*)dx, (const bf16_t*)mask, accumulate != 0, (hipStream_t)stream);
    API_END
}
size_t ssd_op_conv2d_wgrad_bf16_ws_floats(int b, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride,
                                          int dil, int pad_h, int pad_w) {
    return conv_wgrad_bf16_ws_floats(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w));
}
int ssd_op_conv2d_wgrad_bf16(const void* x, const void* dy, float* dw, float* dbias, const float* w, float weight_decay,
                             float* ws, int b, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride,
                             int dil, int pad_h, int pad_w, void* stream) {
    API_BEGIN
    conv_wgrad_bf16(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w), (const bf16_t*)x, (const bf16_t*)dy, dw,
                    dbias, w, weight_decay, ws, (hipStream_t)stream);
    API_END
}
namespace {
struct TailPacked {      // a filter mirror packed for the chain kernel into a temporary buffer
    void* p = nullptr;
    TailPacked(const ConvDesc& d, bool dgrad, const void* mirror, hipStream_t s) {
        HIP_OK(hipMalloc(&p, tail_chain_packed_elems(d, dgrad) * 2));
        const TailPackItem it{d, dgrad, mirror, p};
        tail_chain_pack_filters(&it, 1, s);
    }
    ~TailPacked() { if (p) (void)hipFree(p); }
};
}  // namespace
int ssd_op_conv2d_fwd_bf16_chain(const void* x, const void* w_oi, const float* bias, void* y, int y_f32, int b, int hi, int wi, int ci,
                                 int ho, int wo, int co, int kh, int kw, int stride, int dil, int pad_h, int pad_w, int relu,
                                 void* stream) {
    API_BEGIN
    TailStage t{};
    t.d = mk(1, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w);
    t.dgrad = false; t.src = x; t.bias = bias; t.dst = y; t.relu = relu != 0; t.out_f32 = y_f32 != 0;
    TailPacked tmp(t.d, false, w_oi, (hipStream_t)stream);      // (unit-parity entry point: the step packs once per weight update)
    t.wgt_packed = tmp.p;
    tail_chain_bf16(&t, 1, b, "tail_fwd_bf16", (hipStream_t)stream);
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    API_END
}
int ssd_op_conv2d_dgrad_bf16_chain(const void* dy, const void* w_io, void* dx, const void* mask, int accumulate, int b, int hi, int wi,
                                   int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil, int pad_h, int pad_w,
                                   void* stream) {
    API_BEGIN
    TailStage t{};
    t.d = mk(1, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w);
    t.dgrad = true; t.src = dy; t.mask = mask; t.dst = dx; t.accum = accumulate != 0;
    TailPacked tmp(t.d, true, w_io, (hipStream_t)stream);
    t.wgt_packed = tmp.p;
    tail_chain_bf16(&t, 1, b, "tail_dgrad_bf16", (hipStream_t)stream);
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    API_END
}
int ssd_op_conv2d_wgrad_bf16_direct(const void* x, const void* dy, float* dw, float* dbias, const float* w, float weight_decay, int b,
                                    int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil, int pad_h,
                                    int pad_w, void* stream) {
    API_BEGIN
    WgradGroupItem it{};
    it.d = mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w);
    it.x = (const bf16_t*)x; it.dy = (const bf16_t*)dy; it.dw = dw; it.dbias = dbias; it.w = w;
    conv_wgrad_group_bf16(&it, 1, weight_decay, (hipStream_t)stream);
    API_END
}
int ssd_op_conv2d_first_fwd_bf16(const float* x, const float* w, const float* bias, void* y, int b, int hi, int wi, int ci, int ho,
                                 int wo, int co, int kh, int kw, int stride, int dil, int pad_h, int pad_w, int relu,
                                 void* stream) {
    API_BEGIN
    conv_first_fwd_bf16(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w), x, w, bias, (bf16_t*)y, relu != 0,
                        (hipStream_t)stream);
    API_END
}
size_t ssd_op_conv2d_first_wgrad_bf16_ws_floats(int b, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride,
                                                int dil, int pad_h, int pad_w) {
    return conv_first_wgrad_bf16_ws_floats(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w));
}
int ssd_op_conv2d_first_wgrad_bf16(const float* x, const void* dy, float* dw, float* dbias, const float* w, float weight_decay,
                                   float* ws, int b, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride,
                                   int dil, int pad_h, int pad_w, void* stream) {
    API_BEGIN
    conv_first_wgrad_bf16(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w), x, (const bf16_t*)dy, dw, dbias, w,
                          weight_decay, ws, (hipStream_t)stream);
    API_END
}
namespace {
struct WinoWs {      // the op-level scratch: [U][Uflip][V][Mx][Yt][Ya][slabs]
    float *U, *Uf, *V, *Mx, *Yt, *Ya, *slabs;
    size_t total;
    WinoWs(const ConvDesc& d, float* ws) {
        const size_t cop = wino_kpad(d.Co), u = (size_t)36 * d.Ci * cop, t = (size_t)36 * wino_tiles(d);
        U = ws; Uf = U + u; V = Uf + u; Mx = V + t * d.Ci; Yt = Mx + t * std::max((size_t)d.Ci, cop); Ya = Yt + t * cop; slabs = Ya + t * cop;
        total = (size_t)(slabs - ws) + wino_wgrad_ws_floats(d);
    }
};
}  // namespace
size_t ssd_op_conv2d_wino_ws_floats(int b, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil,
                                    int pad_h, int pad_w) {
    const ConvDesc d = mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w);
    return wino_applicable(d) ? WinoWs(d, nullptr).total : 0;
}
size_t ssd_op_conv2d_wino_bits_words(int b, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil,
                                     int pad_h, int pad_w) {
    const ConvDesc d = mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w);
    return wino_applicable(d) ? (size_t)wino_tiles(d) * (d.Ci / 4) : 0;
}
int ssd_op_conv2d_wino_fwd(const float* x, const float* w, const float* bias, float* y, float* y_pool, void* rec, void* relu_bits, float* ws,
                           int flags, int b, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil,
                           int pad_h, int pad_w, int relu, void* stream) {
    API_BEGIN
    const ConvDesc d = mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w);
    SSD_REQUIRE(wino_applicable(d), "winograd: 3x3 / stride 1 / SAME layers (any dilation), Ci in multiples of 32, Co of 4");
    const WinoWs k(d, ws);
    if (!(flags & 1)) {
        HIP_OK(hipMemsetAsync(k.Uf, 0, (size_t)36 * d.Ci * wino_kpad(d.Co) * sizeof(float), (hipStream_t)stream));      // the pad rows
        wino_filter(d, w, k.U, k.Uf, (hipStream_t)stream);
    }
    wino_fwd(d, x, k.U, bias, y, relu != 0, k.V, (size_t)wino_tiles(d) * d.Ci, k.Mx, y_pool, rec, (hipStream_t)stream, relu_bits);
    API_END
}
int ssd_op_conv2d_wino_dgrad(const float* dy, const float* w, float* dx, const float* mask, const void* mask_bits, int accumulate,
                             const void* rec, int uh, int uw, float* ws, int flags, int b, int hi, int wi, int ci, int ho, int wo, int co,
                             int kh, int kw, int stride, int dil, int pad_h, int pad_w, void* stream) {
    API_BEGIN
    const ConvDesc d = mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w);
    SSD_REQUIRE(wino_applicable(d), "winograd: 3x3 / stride 1 / SAME layers (any dilation), Ci in multiples of 32, Co of 4");
    const WinoWs k(d, ws);
    if (!(flags & 1)) {
        HIP_OK(hipMemsetAsync(k.Uf, 0, (size_t)36 * d.Ci * wino_kpad(d.Co) * sizeof(float), (hipStream_t)stream));      // the pad rows
        wino_filter(d, w, k.U, k.Uf, (hipStream_t)stream);
    }
    wino_bwd_transform(d, dy, k.Yt, nullptr, (hipStream_t)stream);
    wino_dgrad(d, k.Yt, k.Uf, dx, mask, accumulate != 0, k.Mx, rec, uh, uw, (hipStream_t)stream, mask_bits);
    API_END
}
int ssd_op_conv2d_wino_wgrad(const float* x, const float* dy, float* dw, float* dbias, const float* w, float weight_decay, float* ws,
                             int flags, int b, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil,
                             int pad_h, int pad_w, void* stream) {
    API_BEGIN
    const ConvDesc d = mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w);
    SSD_REQUIRE(wino_applicable(d), "winograd: 3x3 / stride 1 / SAME layers (any dilation), Ci in multiples of 32, Co of 4");
    const WinoWs k(d, ws);
    const size_t vps = (size_t)wino_tiles(d) * d.Ci;
    if (!(flags & 2)) {      // the input's transform: through wino_fwd's first kernel would need a filter; use the dy transform's B^T form on x
        ConvDesc dx = d;
        dx.Co = d.Ci; dx.Ho = d.Hi; dx.Wo = d.Wi;
        wino_bwd_transform(dx, x, k.V, nullptr, (hipStream_t)stream);
    }
    wino_bwd_transform(d, dy, nullptr, k.Ya, (hipStream_t)stream);
    wino_wgrad(d, k.V, vps, k.Ya, dw, dbias, w, weight_decay, k.slabs, (hipStream_t)stream);
    API_END
}
size_t ssd_op_conv2d_wgrad_ws_floats(int b, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride,
                                     int dil, int pad_h, int pad_w) {
    return conv_wgrad_ws_floats(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w));
}
int ssd_op_conv2d_wgrad(const float* x, const float* dy, float* dw, float* dbias, const float* w, float weight_decay,
                        float* ws, int b, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil,
                        int pad_h, int pad_w, void* stream) {
    API_BEGIN
    conv_wgrad(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w), x, dy, dw, dbias, w, weight_decay, ws,
               (hipStream_t)stream);
    API_END
}
int ssd_op_maxpool_fwd(const float* x, float* y, int b, int hi, int wi, int c, int ho, int wo, int k, int stride, int pad_h,
                       int pad_w, void* stream) {
    API_BEGIN
    PoolDesc d{b, hi, wi, c, ho, wo, k, stride, pad_h, pad_w};
    maxpool_fwd(d, x, y, (hipStream_t)stream);
    API_END
}
int ssd_op_maxpool_bwd(const float* x, const float* dy, float* dx, int accumulate, int relu_mask, int b, int hi, int wi, int c,
                       int ho, int wo, int k, int stride, int pad_h, int pad_w, void* stream) {
    API_BEGIN
    PoolDesc d{b, hi, wi, c, ho, wo, k, stride, pad_h, pad_w};
    DevBuf ws(maxpool_bwd_ws_bytes(d));
    maxpool_bwd(d, x, dy, dx, accumulate != 0, relu_mask != 0, maxpool_bwd_ws_bytes(d) ? ws.p : nullptr, (hipStream_t)stream);
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));     // the scratch dies here
    API_END
}
// 2x2 stride-2 pooling with the forward-written record (ops.h), fp32 (bf16 = 0) or bf16 storage: the unfused form of the
// fused-pool entry points below (tests compare the two bit for bit)
int ssd_op_maxpool_rec_fwd(const void* x, void* y, void* rec, int bf16, int b, int hi, int wi, int c, void* stream) {
    API_BEGIN
    PoolDesc d{b, hi, wi, c, (hi + 1) / 2, (wi + 1) / 2, 2, 2, 0, 0};
    if (bf16) maxpool_fwd_rec(d, (const bf16_t*)x, (bf16_t*)y, rec, (hipStream_t)stream);
    else maxpool_fwd_rec(d, (const float*)x, (float*)y, rec, (hipStream_t)stream);
    API_END
}
int ssd_op_maxpool_rec_bwd(const void* rec, const void* dy, void* dx, int relu_mask, int bf16, int b, int hi, int wi, int c,
                           void* stream) {
    API_BEGIN
    PoolDesc d{b, hi, wi, c, (hi + 1) / 2, (wi + 1) / 2, 2, 2, 0, 0};
    if (bf16) maxpool_bwd_rec(d, rec, (const bf16_t*)dy, (bf16_t*)dx, relu_mask != 0, (hipStream_t)stream);
    else maxpool_bwd_rec(d, rec, (const float*)dy, (float*)dx, relu_mask != 0, (hipStream_t)stream);
    API_END
}
// conv (3x3 stride 1 SAME) + bias + relu + 2x2 stride-2 max-pool in ONE kernel: y_pool [b][(ho+1)/2][(wo+1)/2][co], rec = the
// pool's record or NULL; the convolution's own output is not written (conv.h conv_fwd_pool)
int ssd_op_conv2d_fwd_pool(const float* x, const float* w, const float* bias, float* y_pool, void* rec, int b, int hi, int wi,
                           int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil, int pad_h, int pad_w, void* stream) {
    API_BEGIN
    conv_fwd_pool(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w), x, w, bias, y_pool, rec, (hipStream_t)stream);
    API_END
}
int ssd_op_conv2d_fwd_pool_bf16(const void* x, const void* w_oi, const float* bias, void* y_pool, void* rec, int b, int hi, int wi,
                                int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil, int pad_h, int pad_w,
                                void* stream) {
    API_BEGIN
    conv_fwd_pool_bf16(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w), (const bf16_t*)x, (const bf16_t*)w_oi, bias,
                       (bf16_t*)y_pool, rec, (hipStream_t)stream);
    API_END
}
// data gradient of a conv whose input is a pooled tensor, routed through the pool's record into the pool's input gradient
// [b][uh][uw][ci] (relu mask of that tensor's producer from the record's sign bit); the geometry is the convolution's
int ssd_op_conv2d_dgrad_unpool(const float* dy, const float* w, float* dx_unpooled, const void* rec, int uh, int uw, int b, int hi,
                               int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil, int pad_h, int pad_w,
                               void* stream) {
    API_BEGIN
    conv_dgrad_unpool(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w), dy, w, dx_unpooled, rec, uh, uw,
                      (hipStream_t)stream);
    API_END
}
int ssd_op_conv2d_dgrad_unpool_bf16(const void* dy, const void* w_io, void* dx_unpooled, const void* rec, int uh, int uw, int b,
                                    int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int dil, int pad_h,
                                    int pad_w, void* stream) {
    API_BEGIN
    conv_dgrad_unpool_bf16(mk(b, hi, wi, ci, ho, wo, co, kh, kw, stride, dil, pad_h, pad_w), (const bf16_t*)dy, (const bf16_t*)w_io,
                           (bf16_t*)dx_unpooled, rec, uh, uw, (hipStream_t)stream);
    API_END
}
// data gradient of a 64 -> 64 3x3 layer (geometry b, h, w) with the weight gradient of the 3-channel 3x3 first layer below it
// fused in (conv.h conv_dgrad_first_wgrad_bf16): dw1 [3][3][3][64], dbias1 [64]; dx itself is not produced
size_t ssd_op_conv2d_dgrad_first_wgrad_bf16_ws_floats(int b, int h, int w) {
    return conv_dgrad_first_wgrad_bf16_ws_floats(mk(b, h, w, 3, h, w, 64, 3, 3, 1, 1, 1, 1));
}
int ssd_op_conv2d_dgrad_first_wgrad_bf16(const void* dy, const void* w_io, const void* mask, const float* image, float* dw1, float* dbias1,
                                         const float* w1, float weight_decay, float* ws, int b, int h, int w, void* stream) {
    API_BEGIN
    conv_dgrad_first_wgrad_bf16(mk(b, h, w, 64, h, w, 64, 3, 3, 1, 1, 1, 1), (const bf16_t*)dy, (const bf16_t*)w_io, (const bf16_t*)mask,
                                mk(b, h, w, 3, h, w, 64, 3, 3, 1, 1, 1, 1), image, dw1, dbias1, w1, weight_decay, ws, (hipStream_t)stream);
    API_END
}
// which of its 2x2 pools the handle runs fused: bit 0 of out[i] = forward (producer's epilogue), bit 1 = backward (consumer's
// data gradient); i counts the 2x2 stride-2 pools in graph order; *count = their number
int ssd_pool_fusion(ssd_handle h, int* out, int cap, int* count) {
    API_BEGIN_NET(h)
    n.pool_fusion(out, cap, count);
    API_END
}
// measurement aid (tools/step_time.py): drop groups of launches from every handle's step ("" / NULL = none); results are WRONG
int ssd_debug_set_ablate(const char* tokens) {
    API_BEGIN
    set_ablate(tokens);
    API_END
}
int ssd_grads_to_bf16(int device, const float* grads_dev, void* msg_dev, size_t count, void* stream) {
    API_BEGIN
    DeviceGuard guard(device);
    grads_to_bf16(grads_dev, msg_dev, count, (hipStream_t)stream);
    API_END
}
int ssd_grads_from_bf16(int device, const void* msg_dev, float* grads_dev, size_t count, void* stream) {
    API_BEGIN
    DeviceGuard guard(device);
    grads_from_bf16(msg_dev, grads_dev, count, (hipStream_t)stream);
    API_END
}
int ssd_op_clock_monitor(unsigned* out_dev, int nsamples, unsigned period_ticks, void* stream) {
    API_BEGIN
    SSD_REQUIRE(out_dev && nsamples >= 1 && period_ticks >= 100, "bad arguments");
    clock_monitor(out_dev, nsamples, period_ticks, (hipStream_t)stream);
    API_END
}
int ssd_op_l2norm_fwd(const float* x, const float* scale, float* y, int npix, int c, void* stream) {
    API_BEGIN
    l2norm_fwd(npix, c, x, scale, y, (hipStream_t)stream);
    API_END
}
size_t ssd_op_l2norm_bwd_ws_floats(int npix, int c) { return l2norm_bwd_ws_floats(npix, c); }
int ssd_op_l2norm_bwd(const float* x, const float* scale, const float* dy, float* dx, float* dscale, float* ws, int npix,
                      int c, void* stream) {
    API_BEGIN
    l2norm_bwd(npix, c, x, scale, dy, dx, dscale, ws, (hipStream_t)stream);
    API_END
}

}  // extern "C"
