// Training augmentation on the GPU: the PIXEL half of the reference's recipe
// (process_dataset.py:66-140; transforms.py:117-232, 276-301, 345-359, 378-392) for a whole batch in two
// launches.  The decisions (which transforms fire, the expand / crop windows, the surviving boxes) are
// made on the host by the transform mirror (ssd_tensorflow_amd/transforms.py) from Python's `random`
// exactly as the reference draws them; what arrives here is one ssd_augment_params per image.
//
// The reference runs brightness -> distort chain -> channel reorder on the loaded uint8 image, then
// expand (float64 canvas of the mean value) -> crop -> flip -> cv2.resize to the preset's size, then
// astype(float32) (training_data.py:100).  All photometric steps are pointwise, all geometric steps are
// index maps, and every cv2.resize kernel is separable, so the whole chain collapses into a gather:
//   kernel 1: per image and axis, the resize taps (source index + weight) of every output coordinate
//             in the cropped / flipped / expanded frame (OpenCV's conventions, see oracle/augment.py);
//   kernel 2: one thread per output pixel: sum over its taps of photometric(source pixel) or the mean
//             value outside the pasted image; round-and-clamp when the image never left uint8.
// HBM-bound: 3 x 4 bytes written per output pixel, a few source bytes read per tap through L2.
// Compiled with -ffp-contract=off: the HSV round trip and the truncations to uint8 are not FMA-safe.
#include "augment.h"

namespace ssd {

constexpr int AUG_TMAX = 16;          // taps per axis (INTER_AREA shrinking by up to 14x)
enum { ALG_NEAREST = 0, ALG_LINEAR = 1, ALG_CUBIC = 2, ALG_AREA = 3, ALG_LANCZOS4 = 4 };

// TAP-MAJOR: the taps of neighbouring output coordinates sit next to each other, so a wave of the gather kernel (64
// consecutive output pixels of a row) reads tap i of its 64 columns as ONE 256-byte run.  Round 2's coordinate-major
// layout ([dst][AUG_TMAX]) put every lane on its own 64-byte line for each of the two loads per tap.
struct TapTable {
    int* idx;         // [b][2][AUG_TMAX][pitch]   pitch = max(out_w, out_h)
    float* w;         // same
    int* n;           // [b][2][pitch]
};

__device__ static void cubic_w(float x, float* c) {
    const float a = -0.75f;
    // oracle/_cubic_w evaluates in double on a float32 x and rounds the weights to float32
    const double X = x;
    const double c0 = ((a * (X + 1) - 5 * a) * (X + 1) + 8 * a) * (X + 1) - 4 * a;
    const double c1 = ((a + 2) * X - (a + 3)) * X * X + 1;
    const double c2 = ((a + 2) * (1 - X) - (a + 3)) * (1 - X) * (1 - X) + 1;
    c[0] = (float)c0; c[1] = (float)c1; c[2] = (float)c2; c[3] = (float)(1 - c0 - c1 - c2);
}

__device__ static void lanczos4_w(float x, float* c) {
    const double s45 = 0.70710678118654752440084436210485;
    const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    if (x < 1.1920928955078125e-07f) {
        for (int i = 0; i < 8; ++i) c[i] = 0.f;
        c[3] = 1.f;
        return;
    }
    const double X = x;
    const double y0 = -(X + 3) * 3.14159265358979323846 * 0.25;
    const double s0 = sin(y0), c0 = cos(y0);
    double sum = 0;
    float t[8];
    for (int i = 0; i < 8; ++i) {
        const double y = -(X + 3 - i) * 3.14159265358979323846 * 0.25;
        t[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += (double)t[i];
    }
    for (int i = 0; i < 8; ++i) c[i] = (float)((double)t[i] / sum);
}

// taps of destination coordinate d along one axis (src -> dst samples), OpenCV conventions
__device__ static int axis_taps(int src, int dst, int alg, int d, int* idx, float* w) {
    const double scale = (double)src / (double)dst;
    if (alg == ALG_NEAREST) {
        int s = (int)floor(d * scale);
        idx[0] = s < src - 1 ? s : src - 1;
        w[0] = 1.f;
        return 1;
    }
    if (alg == ALG_AREA && scale >= 1.0) {
        const double f1 = d * scale, f2 = f1 + scale;
        const double cell = fmin(scale, src - f1);
        int s1 = (int)ceil(f1), s2 = (int)floor(f2);
        if (s2 > src - 1) s2 = src - 1;
        if (s1 > s2) s1 = s2;
        int k = 0;
        if (s1 - f1 > 1e-3) { idx[k] = s1 - 1; w[k] = (float)((s1 - f1) / cell); ++k; }
        for (int s = s1; s < s2 && k < AUG_TMAX; ++s) { idx[k] = s; w[k] = (float)(1.0 / cell); ++k; }
        if (f2 - s2 > 1e-3 && k < AUG_TMAX) { idx[k] = s2; w[k] = (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell); ++k; }
        return k;
    }
    if (alg == ALG_LINEAR || alg == ALG_AREA) {
        // OpenCV rounds the source coordinate to float BEFORE taking its floor (fx = (float)(...); sx = cvFloor(fx);
        // fx -= sx): a coordinate a hair below an integer lands ON it, not at fraction 1.0 (where Lanczos divides by 0)
        int s;
        float f;
        if (alg == ALG_AREA) {      // enlarging with INTER_AREA: the linear path with the area-mode coefficient
            s = (int)floor(d * scale);
            f = (float)((d + 1) - (s + 1) / scale);
            f = f <= 0.f ? 0.f : f - floorf(f);
        } else {
            const float fx = (float)((d + 0.5) * scale - 0.5);
            s = (int)floorf(fx);
            f = fx - floorf(fx);
        }
        if (s < 0) { f = 0; s = 0; }
        if (s >= src - 1) { f = 0; s = src - 1; }
        idx[0] = s; idx[1] = s + 1 < src - 1 ? s + 1 : src - 1;
        w[0] = 1.f - f; w[1] = f;
        return 2;
    }
    const float fx = (float)((d + 0.5) * scale - 0.5);
    const int s = (int)floorf(fx);
    const float f = fx - floorf(fx);
    const int T = alg == ALG_CUBIC ? 4 : 8, first = alg == ALG_CUBIC ? -1 : -3;
    if (alg == ALG_CUBIC) cubic_w(f, w); else lanczos4_w(f, w);
    for (int i = 0; i < T; ++i) {
        int q = s + first + i;
        idx[i] = q < 0 ? 0 : (q > src - 1 ? src - 1 : q);
    }
    return T;
}

__global__ __launch_bounds__(256) void augment_taps_kernel(const ssd_augment_params* __restrict__ prm, int b, int out_w, int out_h, TapTable t) {
    const int per = out_w + out_h;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= b * per) return;
    const int img = gid / per, r = gid - img * per;
    const int axis = r >= out_w;                 // 0: x, 1: y
    const int d = axis ? r - out_w : r;
    const int dst = axis ? out_h : out_w;
    const ssd_augment_params& p = prm[img];
    const int src = axis ? p.crop_h : p.crop_w;
    const int pitch = out_w > out_h ? out_w : out_h;
    const size_t plane = (size_t)(img * 2 + axis);
    int li[AUG_TMAX];
    float lw[AUG_TMAX];
    const int nt = axis_taps(src, dst, p.resize_alg, d, li, lw);
    t.n[plane * pitch + d] = nt;
    for (int i = 0; i < nt; ++i) {
        t.idx[(plane * AUG_TMAX + i) * pitch + d] = li[i];
        t.w[(plane * AUG_TMAX + i) * pitch + d] = lw[i];
    }
}

// ---- photometric chain on one uint8 BGR pixel (oracle/augment.py brightness / contrast / hue / saturation) ----
__device__ __forceinline__ float u8f(float v) {       // clip to [0, 255], astype(uint8) truncates
    v = v > 255.f ? 255.f : v;
    v = v < 0.f ? 0.f : v;
    return floorf(v);
}
__device__ static void bgr2hsv_u8(float* px) {
    const float b = px[0], g = px[1], r = px[2];
    const float v = fmaxf(fmaxf(b, g), r), mn = fminf(fminf(b, g), r);
    const float diff = v - mn;
    const float s = v > 0.f ? diff * 255.0f / v : 0.f;
    const float safe = diff > 0.f ? diff : 1.f;
    float h = v == r ? (g - b) / safe * 60.0f : (v == g ? 120.0f + (b - r) / safe * 60.0f : 240.0f + (r - g) / safe * 60.0f);
    h = diff > 0.f ? h : 0.f;
    h = h < 0.f ? h + 360.0f : h;
    int hi = (int)floorf(h / 2.f + 0.5f) % 180;
    float si = floorf(s + 0.5f);
    si = si < 0.f ? 0.f : (si > 255.f ? 255.f : si);
    px[0] = (float)hi; px[1] = si; px[2] = v;
}
__device__ static void hsv2bgr_u8(float* px) {
    const float h = px[0] * 2.0f, s = px[1] / 255.0f, v = px[2];
    const float hh = h / 60.0f;
    const float fl = floorf(hh);
    const int sector = ((int)fl) % 6;
    const float f = hh - fl;
    const float p_ = v * (1 - s), q_ = v * (1 - s * f), t_ = v * (1 - s * (1 - f));
    float r, g, b;
    switch (sector) {
    case 0: r = v; g = t_; b = p_; break;
    case 1: r = q_; g = v; b = p_; break;
    case 2: r = p_; g = v; b = t_; break;
    case 3: r = p_; g = q_; b = v; break;
    case 4: r = t_; g = p_; b = v; break;
    default: r = v; g = p_; b = q_; break;
    }
    const float o[3] = {b, g, r};
    for (int c = 0; c < 3; ++c) {
        float x = floorf(o[c] + 0.5f);
        px[c] = x < 0.f ? 0.f : (x > 255.f ? 255.f : x);
    }
}

// one step of the chain on a pixel of image row `row`: kind 0 contrast, 1 saturation, 2 hue, 3 brightness, 4 channel reorder
__device__ static void photometric_step(int kind, float val, int row, float* px) {
    {
        if (kind == 3) {                                   // brightness
            for (int c = 0; c < 3; ++c) px[c] = u8f(px[c] + val);
        } else if (kind == 4) {                            // reorder: out channel c = in channel (code >> 2c) & 3
            const int code = (int)val;
            const float q[3] = {px[0], px[1], px[2]};
            for (int c = 0; c < 3; ++c) {
                const int sc = (code >> (2 * c)) & 3;
                px[c] = sc == 0 ? q[0] : (sc == 1 ? q[1] : q[2]);
            }
        } else if (kind == 0) {                            // contrast
            for (int c = 0; c < 3; ++c) px[c] = u8f(px[c] * val);
        } else if (kind == 1) {                            // saturation: HSV round trip; image ROW 1 is scaled (reference quirk)
            bgr2hsv_u8(px);
            if (row == 1)
                for (int c = 0; c < 3; ++c) { float x = px[c] * val; x = x > 255.f ? 255.f : x; x = x < 0.f ? 0.f : x; px[c] = floorf(x); }
            hsv2bgr_u8(px);
        } else {                                           // hue: HSV round trip; image ROW 0 is shifted (reference quirk)
            bgr2hsv_u8(px);
            if (row == 0)
                for (int c = 0; c < 3; ++c) {
                    float x = px[c] + val;
                    if (x > 180.f) x -= 180.f;
                    if (x < 0.f) x += 180.f;
                    px[c] = (float)(unsigned char)(int)x;   // astype(uint8) of a value in [0, 255+)
                }
            hsv2bgr_u8(px);
        }
    }
}

// the whole chain: brightness -> distort chain -> channel reorder (the reference's recipe order) -> the extra steps of a second pass
// (round 6: an extra step's Hue / Saturation rows are rows of the array the step was handed -- source row extra_r0[i] is its row 0)
__device__ static void photometric(const ssd_augment_params& p, int row, float* px) {
    if (p.brightness_on) photometric_step(3, (float)p.brightness_delta, row, px);
    for (int i = 0; i < p.n_distort; ++i) photometric_step(p.distort_kind[i], p.distort_val[i], row, px);
    {
        const float q[3] = {px[0], px[1], px[2]};
        for (int c = 0; c < 3; ++c) px[c] = p.reorder[c] == 0 ? q[0] : (p.reorder[c] == 1 ? q[1] : q[2]);
    }
    for (int i = 0; i < p.n_extra; ++i) photometric_step(p.extra_kind[i], p.extra_val[i], row - p.extra_r0[i], px);
}
// a canvas pixel (ExpandTransform's mean value) at source-frame row `row`: the steps taken behind the expand
__device__ static void photometric_fill(const ssd_augment_params& p, int row, double* px) {
    if (p.fill_from >= p.n_extra) return;
    float f[3] = {(float)px[0], (float)px[1], (float)px[2]};      // (astype(float32) of the float64 canvas, transforms.py:168,182)
    for (int i = p.fill_from; i < p.n_extra; ++i) photometric_step(p.extra_kind[i], p.extra_val[i], row - p.extra_r0[i], f);
    for (int c = 0; c < 3; ++c) px[c] = (double)f[c];
}

// Pass 0 (round 3): the photometric chain depends on the SOURCE pixel only, yet the gather evaluated it per tap -- up to 64
// taps per output pixel (Lanczos), two HSV round trips each.  Its result is an exact uint8 value (every step ends in a
// truncation / rounding to 0..255), so images whose chain is not the identity are transformed once, pixel by pixel, into a
// uint8 copy in the workspace (AUG_PRE_BYTES per image; a larger image keeps the per-tap path) and the gather reads bytes.
constexpr size_t AUG_PRE_BYTES = 1310720;      // 1.25 MiB: a 660 x 660 BGR image

// (a plan whose chain is a bare channel reorder reads the source bytes and permutes them in the gather)
__device__ __forceinline__ bool aug_has_photometric(const ssd_augment_params& p) { return p.brightness_on || p.n_distort > 0 || p.n_extra > 0; }
__device__ __forceinline__ bool aug_uses_pre(const ssd_augment_params& p) {
    return aug_has_photometric(p) && (size_t)p.src_w * p.src_h * 3 <= AUG_PRE_BYTES;
}

__global__ __launch_bounds__(256) void augment_photometric_kernel(const unsigned char* __restrict__ images, const ssd_augment_params* __restrict__ prm,
                                                                  unsigned char* __restrict__ pre) {
    const int img = blockIdx.y;
    const ssd_augment_params& p = prm[img];
    if (!aug_uses_pre(p)) return;
    const int npix = p.src_w * p.src_h;
    const unsigned char* src = images + p.src_off;            // 16-byte aligned (the packing pads every image to 16 bytes)
    unsigned char* dst = pre + (size_t)img * AUG_PRE_BYTES;
    // (one pixel per thread: the chain is ALU-bound -- two HSV round trips -- not byte-bound; four pixels per thread with dword
    // loads and stores measured 55 instead of 42 us per batch)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const unsigned char* q = src + (size_t)i * 3;
        float raw[3] = {(float)q[0], (float)q[1], (float)q[2]};
        photometric(p, i / p.src_w, raw);
        for (int c = 0; c < 3; ++c) dst[(size_t)i * 3 + c] = (unsigned char)(int)raw[c];      // exact: integers in 0..255
    }
}

// One output pixel per thread, a workgroup = 256 consecutive pixels of ONE image (blockIdx.y).  Round 6: the x taps of a pixel
// (source column + weight) do not depend on the row tap, yet the round-3 loop re-read them for every one of the ny rows and walked
// the nx columns as a dynamic loop of dependent loads (index -> address -> pixel -> convert -> three double FMAs): the kernel's time
// was the latency of nx * ny such chains.  Here a workgroup looks up its image's tap count once, picks the instantiation with NX =
// 1, 2, 4, 8 or 16 column taps (INTER_AREA's 3 .. 15 are padded with zero-weight repeats of the last tap: x + 0.0 * v = x), keeps
// the NX indices and weights in registers and issues the NX pixel loads of a row back to back from clamped addresses -- visibility
// and the canvas colour are selected afterwards -- before the NX * 3 FMAs in the order the oracle prescribes.  Same sums bit for bit
// (tests/test_gpu_augment.py).  An image whose photometric chain runs per tap (larger than the pre-pass buffer) keeps the loop form.
template <int NX>
__device__ __forceinline__ void augment_gather_body(const unsigned char* __restrict__ images, const ssd_augment_params& p, int img, int ox, int oy,
                                                    int out_w, int out_h, const TapTable& t, const unsigned char* __restrict__ pre,
                                                    float* __restrict__ out) {
    const int pitch = out_w > out_h ? out_w : out_h;
    const size_t px_ = (size_t)(img * 2 + 0), py_ = (size_t)(img * 2 + 1);
    const int nx = t.n[px_ * pitch + ox], ny = t.n[py_ * pitch + oy];
    const int* xi = t.idx + px_ * AUG_TMAX * pitch + ox;       // tap i at xi[i * pitch]
    const float* xw = t.w + px_ * AUG_TMAX * pitch + ox;
    const int* yi = t.idx + py_ * AUG_TMAX * pitch + oy;
    const float* yw = t.w + py_ * AUG_TMAX * pitch + oy;
    const bool from_pre = aug_uses_pre(p);                 // the chain has been applied by pass 0
    const bool per_tap = aug_has_photometric(p) && !from_pre;
    const unsigned char* src = from_pre ? pre + (size_t)img * AUG_PRE_BYTES : images + p.src_off;
    const bool fill_steps = p.fill_from < p.n_extra;
    typedef unsigned u32_unaligned __attribute__((aligned(1)));
    double acc[3] = {0, 0, 0};
    if constexpr (NX > 0) {
        // column taps: source column in the loaded image (x0), its visibility, the weight -- once per pixel
        int x0[NX];
        double wx[NX];
        bool vx[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int ii = i < nx ? i : nx - 1;
            int sx = xi[(size_t)ii * pitch];
            if (p.flip) sx = p.crop_w - 1 - sx;
            x0[i] = sx + p.crop_x0 - p.exp_woff;
            vx[i] = (unsigned)(x0[i] - p.clip_x0) < (unsigned)(p.clip_x1 - p.clip_x0);
            wx[i] = i < nx ? (double)xw[(size_t)ii * pitch] : 0.0;
        }
        const int last_x = p.src_w - 1, last_y = p.src_h - 1;
        for (int j = 0; j < ny; ++j) {
            const int y0 = yi[(size_t)j * pitch] + p.crop_y0 - p.exp_hoff;      // row in the loaded image
            const double wy = yw[(size_t)j * pitch];
            const bool vy = (unsigned)(y0 - p.clip_y0) < (unsigned)(p.clip_y1 - p.clip_y0);
            const unsigned char* rowp = src + (size_t)(vy ? y0 : 0) * p.src_w * 3;
            unsigned v[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int xc = vx[i] ? x0[i] : 0;
                const unsigned char* q = rowp + (size_t)xc * 3;
                // one (unaligned) dword instead of three byte loads -- except for the image's very last pixel, whose fourth byte may
                // lie outside the caller's buffer
                if (vy && y0 == last_y && xc == last_x) v[i] = (unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16);
                else v[i] = *reinterpret_cast<const u32_unaligned*>(q);
            }
            double fillc[3] = {p.mean[0], p.mean[1], p.mean[2]};      // the canvas colour of this row
            if (fill_steps) photometric_fill(p, y0, fillc);
            double rowacc[3] = {0, 0, 0};
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double px[3];
                if (vy && vx[i]) {
                    const float raw[3] = {(float)(v[i] & 255u), (float)((v[i] >> 8) & 255u), (float)((v[i] >> 16) & 255u)};
                    if (from_pre) { px[0] = raw[0]; px[1] = raw[1]; px[2] = raw[2]; }
                    else for (int c = 0; c < 3; ++c) px[c] = raw[p.reorder[c]];
                } else {
                    px[0] = fillc[0]; px[1] = fillc[1]; px[2] = fillc[2];
                }
                for (int c = 0; c < 3; ++c) rowacc[c] += wx[i] * px[c];
            }
            for (int c = 0; c < 3; ++c) acc[c] += wy * rowacc[c];
        }
    } else {
        for (int j = 0; j < ny; ++j) {
            const int sy = yi[(size_t)j * pitch] + p.crop_y0;                  // row in the (expanded) frame
            const double wy = yw[(size_t)j * pitch];
            double rowacc[3] = {0, 0, 0};
            for (int i = 0; i < nx; ++i) {
                int sx = xi[(size_t)i * pitch];
                if (p.flip) sx = p.crop_w - 1 - sx;
                sx += p.crop_x0;
                const double wx = xw[(size_t)i * pitch];
                const int y0 = sy - p.exp_hoff, x0 = sx - p.exp_woff;         // position in the loaded image (offsets are 0 when not expanded)
                double px[3];
                if ((unsigned)(y0 - p.clip_y0) < (unsigned)(p.clip_y1 - p.clip_y0) && (unsigned)(x0 - p.clip_x0) < (unsigned)(p.clip_x1 - p.clip_x0)) {
                    const unsigned char* q = src + ((size_t)y0 * p.src_w + x0) * 3;
                    const bool last_px = (y0 == p.src_h - 1) & (x0 == p.src_w - 1);
                    unsigned v;
                    if (last_px) v = (unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16);
                    else v = *reinterpret_cast<const u32_unaligned*>(q);
                    float raw[3] = {(float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u)};
                    if (per_tap) photometric(p, y0, raw);      // (incl. the reorder)
                    if (per_tap || from_pre) { px[0] = raw[0]; px[1] = raw[1]; px[2] = raw[2]; }
                    else for (int c = 0; c < 3; ++c) px[c] = raw[p.reorder[c]];
                } else {
                    for (int c = 0; c < 3; ++c) px[c] = p.mean[c];
                    if (fill_steps) photometric_fill(p, y0, px);
                }
                for (int c = 0; c < 3; ++c) rowacc[c] += wx * px[c];
            }
            for (int c = 0; c < 3; ++c) acc[c] += wy * rowacc[c];
        }
    }
    float res[3];
    for (int c = 0; c < 3; ++c) {
        double v = acc[c];
        if (!p.is_float) {                    // the image is still uint8 in the reference: cv2.resize saturates to uint8
            v = floor(v + 0.5);
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        res[c] = (float)v;
    }
    // steps behind ResizeTransform act on the resized array (its rows 0 / 1), a flip behind it mirrors the output columns
    for (int i = 0; i < p.n_post; ++i) photometric_step(p.post_kind[i], p.post_val[i], oy, res);
    const size_t gid = ((size_t)img * out_h + oy) * out_w + (p.out_flip ? out_w - 1 - ox : ox);
    float* o = out + gid * 3;
    for (int c = 0; c < 3; ++c) o[c] = res[c];
}

__global__ __launch_bounds__(256) void augment_gather_kernel(const unsigned char* __restrict__ images, const ssd_augment_params* __restrict__ prm,
                                                             int b, int out_w, int out_h, TapTable t, const unsigned char* __restrict__ pre,
                                                             float* __restrict__ out) {
    const int img = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= out_w * out_h) return;
    const int ox = pix % out_w, oy = pix / out_w;
    const ssd_augment_params& p = prm[img];
    // the widest column tap count of the image: uniform per workgroup (nearest 1, linear / enlarging area 2, cubic 4, Lanczos 8,
    // shrinking area up to 16)
    int nxmax;
    const double sx = (double)p.crop_w / (double)out_w;
    if (p.resize_alg == ALG_NEAREST) nxmax = 1;
    else if (p.resize_alg == ALG_LINEAR || (p.resize_alg == ALG_AREA && sx < 1.0)) nxmax = 2;
    else if (p.resize_alg == ALG_CUBIC) nxmax = 4;
    else if (p.resize_alg == ALG_LANCZOS4) nxmax = 8;
    else nxmax = AUG_TMAX;
    if (p.resize_alg == ALG_AREA && sx >= 1.0) nxmax = sx <= 2.0 ? 4 : (sx <= 6.0 ? 8 : 16);      // at most floor(scale) + 2 taps
    const bool per_tap = aug_has_photometric(p) && !aug_uses_pre(p);
    const int pitch = out_w > out_h ? out_w : out_h;
    const int nx = t.n[(size_t)(img * 2) * pitch + ox];
    // (shrinking by more than 6x -- 16 taps, 2000-pixel sources -- keeps the loop form: its 16 doubles of weights would set the
    // register allocation, and with it the resident waves, of every other case)
    if (per_tap || nxmax > 8 || nx > nxmax) augment_gather_body<0>(images, p, img, ox, oy, out_w, out_h, t, pre, out);
    else if (nxmax == 1) augment_gather_body<1>(images, p, img, ox, oy, out_w, out_h, t, pre, out);
    else if (nxmax == 2) augment_gather_body<2>(images, p, img, ox, oy, out_w, out_h, t, pre, out);
    else if (nxmax == 4) augment_gather_body<4>(images, p, img, ox, oy, out_w, out_h, t, pre, out);
    else augment_gather_body<8>(images, p, img, ox, oy, out_w, out_h, t, pre, out);
}

size_t augment_ws_bytes(int b, int out_w, int out_h) {
    const size_t rows = (size_t)b * 2 * (out_w > out_h ? out_w : out_h);
    return rows * AUG_TMAX * (sizeof(int) + sizeof(float)) + rows * sizeof(int) + ((size_t)b * sizeof(ssd_augment_params) + 255) / 256 * 256 + 512 +
           (size_t)b * AUG_PRE_BYTES;
}

void augment_batch(const unsigned char* images_dev, const ssd_augment_params* params_host, int b, int out_w, int out_h, float* out_dev,
                   void* ws, hipStream_t s) {
    SSD_REQUIRE(b >= 1 && out_w >= 1 && out_h >= 1, "augment: empty batch");
    for (int i = 0; i < b; ++i) {
        const ssd_augment_params& p = params_host[i];
        SSD_REQUIRE(p.src_w >= 1 && p.src_h >= 1 && p.crop_w >= 1 && p.crop_h >= 1, "augment: image %d has an empty source or crop window", i);
        SSD_REQUIRE(p.resize_alg >= 0 && p.resize_alg <= 4, "augment: image %d: unknown resize algorithm %d", i, p.resize_alg);
        SSD_REQUIRE(p.n_distort >= 0 && p.n_distort <= 3, "augment: image %d: distort chain length %d", i, p.n_distort);
        SSD_REQUIRE(p.n_extra >= 0 && p.n_extra <= 16, "augment: image %d: %d extra photometric steps", i, p.n_extra);
        SSD_REQUIRE(p.n_post >= 0 && p.n_post <= 4, "augment: image %d: %d steps behind the resize", i, p.n_post);
        for (int k = 0; k < p.n_post; ++k) SSD_REQUIRE(p.post_kind[k] >= 0 && p.post_kind[k] <= 4, "augment: image %d: post step kind %d", i, p.post_kind[k]);
        SSD_REQUIRE(p.fill_from >= 0 && p.fill_from <= p.n_extra, "augment: image %d: fill_from %d of %d extra steps", i, p.fill_from, p.n_extra);
        SSD_REQUIRE(p.clip_x0 >= 0 && p.clip_y0 >= 0 && p.clip_x1 <= p.src_w && p.clip_y1 <= p.src_h && p.clip_x0 <= p.clip_x1 && p.clip_y0 <= p.clip_y1,
                    "augment: image %d: visible window %d..%d x %d..%d outside the %d x %d image", i, p.clip_x0, p.clip_x1, p.clip_y0, p.clip_y1, p.src_w, p.src_h);
        for (int k = 0; k < p.n_extra; ++k) SSD_REQUIRE(p.extra_kind[k] >= 0 && p.extra_kind[k] <= 4, "augment: image %d: extra step kind %d", i, p.extra_kind[k]);
        for (int c = 0; c < 3; ++c) SSD_REQUIRE(p.reorder[c] >= 0 && p.reorder[c] <= 2, "augment: image %d: channel permutation", i);
        const double sx = (double)p.crop_w / out_w, sy = (double)p.crop_h / out_h;
        SSD_REQUIRE(p.resize_alg != ALG_AREA || (sx <= AUG_TMAX - 2 && sy <= AUG_TMAX - 2), "augment: image %d shrinks by more than %dx (INTER_AREA tap table)",
                    i, AUG_TMAX - 2);
        const int fw = p.expand_on ? p.exp_w : p.src_w, fh = p.expand_on ? p.exp_h : p.src_h;
        SSD_REQUIRE(p.crop_x0 >= 0 && p.crop_y0 >= 0 && p.crop_x0 + p.crop_w <= fw && p.crop_y0 + p.crop_h <= fh, "augment: image %d: crop window outside the frame", i);
    }
    const size_t rows = (size_t)b * 2 * (out_w > out_h ? out_w : out_h);
    char* base = static_cast<char*>(ws);
    TapTable t;
    t.idx = reinterpret_cast<int*>(base);
    t.w = reinterpret_cast<float*>(base + rows * AUG_TMAX * sizeof(int));
    t.n = reinterpret_cast<int*>(base + rows * AUG_TMAX * (sizeof(int) + sizeof(float)));
    const size_t prm_off = ((rows * AUG_TMAX * 8 + rows * 4 + 255) / 256) * 256;
    ssd_augment_params* prm = reinterpret_cast<ssd_augment_params*>(base + prm_off);
    unsigned char* pre = reinterpret_cast<unsigned char*>(base + prm_off + ((size_t)b * sizeof(ssd_augment_params) + 255) / 256 * 256);
    HIP_OK(hipMemcpyAsync(prm, params_host, (size_t)b * sizeof(ssd_augment_params), hipMemcpyHostToDevice, s));
    int pre_pixels = 0;      // the largest image pass 0 has to transform (0: no image needs it)
    for (int i = 0; i < b; ++i) {
        const ssd_augment_params& p = params_host[i];
        const size_t bytes = (size_t)p.src_w * p.src_h * 3;
        if ((p.brightness_on || p.n_distort > 0 || p.n_extra > 0) && bytes <= AUG_PRE_BYTES && p.src_w * p.src_h > pre_pixels) pre_pixels = p.src_w * p.src_h;
    }
    if (pre_pixels > 0) {
        ProfScope prof("augment_photometric", 0.0, (double)b * pre_pixels * 6, s);
        const int gx = cdiv(pre_pixels, 256 * 4) < 1 ? 1 : cdiv(pre_pixels, 256 * 4);      // four strided pixels per thread
        hipLaunchKernelGGL(augment_photometric_kernel, dim3(gx, b), dim3(256), 0, s, images_dev, prm, pre);
    }
    {
        const int n = b * (out_w + out_h);
        ProfScope prof("augment_taps", 0.0, (double)n * AUG_TMAX * 8, s);
        hipLaunchKernelGGL(augment_taps_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, prm, b, out_w, out_h, t);
    }
    {
        const size_t total = (size_t)b * out_w * out_h;
        ProfScope prof("augment_gather", 0.0, (double)total * 12, s);
        hipLaunchKernelGGL(augment_gather_kernel, dim3(cdiv(out_w * out_h, 256), b), dim3(256), 0, s, images_dev, prm, b, out_w, out_h, t, pre, out_dev);
    }
    HIP_OK(hipGetLastError());
}

}  // namespace ssd
