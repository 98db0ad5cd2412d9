// Box math of the SSD hot path on gfx950: anchors, label encoding, decode, per-class NMS.
// Replaces the numpy loops of ssdutils.py:76-318 and transforms.py:57-114.
#pragma once
#include "common.h"

namespace ssd {

constexpr int PRESET_MAX_MAPS = 8;
constexpr int PRESET_MAX_TYPES = 8;

struct Preset {
    const char* name;
    int image_w, image_h;
    int nmaps;
    int map_size[PRESET_MAX_MAPS];
    double scale[PRESET_MAX_MAPS];
    int nratios[PRESET_MAX_MAPS];
    double ratios[PRESET_MAX_MAPS][4];
    double extra_scale;
    int num_anchors;
    // derived
    int ntypes[PRESET_MAX_MAPS];                          // 2 + nratios
    int off[PRESET_MAX_MAPS + 1];                         // first anchor of each map
    double bw[PRESET_MAX_MAPS][PRESET_MAX_TYPES];         // box sizes (host libm sqrt: correctly rounded)
    double bh[PRESET_MAX_MAPS][PRESET_MAX_TYPES];
};

// ssdutils.py:70-73; throws ssd::Error("No such preset: ...") like the reference's RuntimeError.
const Preset& get_preset(const char* name);

// [A][4] f64 (cx, cy, w, h) and [A][4] i32 (xmin, xmax, ymin, ymax on the 1000 grid), device.
void anchors_device(const Preset& p, double* anchors, int* anchors_abs, hipStream_t s);

// jaccard_overlap of one box vs n boxes (f64, +1 pixel convention); device pointers
void jaccard_device(const double* box, const double* arr, int n, double* iou, hipStream_t s);

// LabelCreatorTransform for a batch.  gt [ntot][4] f64 (cx,cy,w,h), cls [ntot], offsets [B+1]
// (CSR) all device; anchors/anchors_abs from anchors_device.  vec [B][A][C+5] f32 device.
size_t encode_labels_ws_bytes(int ntot);
void encode_labels(const Preset& p, int num_classes, const double* anchors, const int* anchors_abs, const double* gt,
                   const int* cls, const int* offsets, int B, int ntot, float* vec, void* ws, hipStream_t s);

// decode_boxes + suppress_overlaps (+ the caller's [:max_out]) for a batch.
struct DetectOut {
    int* count;    // [B] boxes kept (before the out_cap clip)
    float* conf;   // [B][out_cap]
    int* cls;      // [B][out_cap]
    int* idx;      // [B][out_cap] anchor index
    int* box;      // [B][out_cap][4] xmin,xmax,ymin,ymax: normalize_box's integers
};
size_t detect_ws_bytes(int B, int A);
// nms = false: decode_boxes only (confidence order, nothing suppressed)
void detect(int A, int num_classes, const double* anchors, const float* pred, int B, float conf_thr, int cap,
            int max_out, int out_cap, bool nms, const DetectOut& out, void* ws, hipStream_t s);

// non_maximum_suppression / suppress_overlaps on an arbitrary list: boxes [n][4] i32 (xmin,xmax,ymin,ymax),
// conf [n], group [n] (0..ngroups-1); keep [n+1]: keep[0] = count, then the selected input indices in output order.
size_t nms_boxes_ws_bytes(int n, int ngroups);
void nms_boxes_device(int n, int ngroups, const int* boxes, const float* conf, const int* group, double thr, int* keep, void* ws,
                      hipStream_t s);

}  // namespace ssd
