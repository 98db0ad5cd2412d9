// Host half of the feeder, native: the trial loop of the reference's SamplerTransform (transforms.py:304-359) for all the
// samplers of a SamplePickerTransform (transforms.py:362-376) in one call.
//
// The reference draws a crop window up to 50 times per sampler (7 samplers per image, process_dataset.py:107-119) and
// tests its best IoU with the ground-truth boxes: ~210 trials per image, each a handful of Python float operations and
// a numpy call -- 3 to 5 ms per image, the whole cost of planning a batch.  The draws come from Python's `random`
// (Mersenne Twister); the Python mirror (ssd_tensorflow_amd/transforms.py) keeps that contract -- same draws, same
// order -- so this loop takes the generator's state (the 624 words + index of random.getstate()), draws exactly what
// random.uniform would, and hands the advanced state back.  No GPU call: the feeder's forked workers run this.
//
// Every floating-point operation is spelled as the interpreter performs it (IEEE double, no contraction: compiled with
// -ffp-contract=off): x ** 2 is pow(x, 2.0), random.uniform(a, b) is a + (b - a) * random(), random() is
// (a >> 5, b >> 6) -> (a * 67108864 + b) / 2^53, int() truncates toward zero, the IoU uses the +1 pixel convention of
// ssdutils.py:139-149 in float64.
#include "common.h"

#include <cmath>
#include <cstdint>

#include "../../include/ssdvgg_hip.h"

namespace {

struct MT {      // MT19937 exactly as CPython's _randommodule.c
    uint32_t* mt;      // 624 words
    uint32_t* idx;     // position
    uint32_t next() {
        static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
        uint32_t y;
        if (*idx >= 624) {
            int kk;
            for (kk = 0; kk < 624 - 397; kk++) {
                y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 0x1u];
            }
            for (; kk < 623; kk++) {
                y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 0x1u];
            }
            y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 0x1u];
            *idx = 0;
        }
        y = mt[(*idx)++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double random() {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
    }
    double uniform(double a, double b) { return a + (b - a) * random(); }
};

inline long long trunc_ll(double v) { return (long long)v; }      // int(): toward zero

}  // namespace

extern "C" int ssd_sampler_trials(unsigned int* mt_state, int n_samplers, const double* params, const int* max_trials, int img_w,
                                  int img_h, const double* gt_px, int n_gt, long long* windows_out, int* found_out) {
    using namespace ssd;
    try {
        SSD_REQUIRE(mt_state && params && max_trials && windows_out && found_out, "null argument");
        SSD_REQUIRE(n_samplers >= 0 && n_gt >= 0 && (n_gt == 0 || gt_px), "bad sizes");
        SSD_REQUIRE(mt_state[624] <= 624, "not a Mersenne Twister state (index %u)", mt_state[624]);
        MT g{mt_state, mt_state + 624};
        const double W = (double)img_w, H = (double)img_h;
        for (int s = 0; s < n_samplers; ++s) {
            const double min_scale = params[5 * s], max_scale = params[5 * s + 1], min_ar = params[5 * s + 2], max_ar = params[5 * s + 3],
                         min_jaccard = params[5 * s + 4];
            found_out[s] = 0;
            for (int t = 0; t < max_trials[s]; ++t) {
                const double scale = g.uniform(min_scale, max_scale);
                double ratio = g.uniform(min_ar, max_ar);
                const double s2 = std::pow(scale, 2.0);
                if (s2 > ratio) ratio = s2;                     // max(ratio, scale ** 2)
                const double inv = 1.0 / std::pow(scale, 2.0);
                if (inv < ratio) ratio = inv;                   // min(ratio, 1 / scale ** 2)
                const double width = scale * std::sqrt(ratio), height = scale / std::sqrt(ratio);
                const double cx = 0.5 * width + g.uniform(0.0, 1.0 - width);
                const double cy = 0.5 * height + g.uniform(0.0, 1.0 - height);
                // prop2abs (utils.py:100-108)
                const double w2 = width * W / 2, h2 = height * H / 2, ax = cx * W, ay = cy * H;
                const long long x0 = trunc_ll(ax - w2), x1 = trunc_ll(ax + w2), y0 = trunc_ll(ay - h2), y1 = trunc_ll(ay + h2);
                SSD_REQUIRE(n_gt > 0, "SamplerTransform: no ground-truth box left to test the window against");
                const double areab = (double)((x1 - x0 + 1) * (y1 - y0 + 1));
                double best = -INFINITY;
                for (int k = 0; k < n_gt; ++k) {
                    const double* o = gt_px + 4 * k;      // xmin, xmax, ymin, ymax
                    const double areaa = (o[1] - o[0] + 1) * (o[3] - o[2] + 1);
                    const double xxmin = (double)x0 > o[0] ? (double)x0 : o[0], xxmax = (double)x1 < o[1] ? (double)x1 : o[1];
                    const double yymin = (double)y0 > o[2] ? (double)y0 : o[2], yymax = (double)y1 < o[3] ? (double)y1 : o[3];
                    double w = xxmax - xxmin + 1, h = yymax - yymin + 1;
                    if (!(w > 0)) w = 0;
                    if (!(h > 0)) h = 0;
                    const double inter = w * h;
                    const double iou = inter / (areab + areaa - inter);
                    if (iou > best) best = iou;
                }
                if (best > 0 && best >= min_jaccard) {
                    found_out[s] = 1;
                    windows_out[4 * s] = x0; windows_out[4 * s + 1] = x1; windows_out[4 * s + 2] = y0; windows_out[4 * s + 3] = y1;
                    break;
                }
            }
        }
        return 0;
    } catch (const std::exception& e) {
        ssd::set_error("%s", e.what());
        return 1;
    } catch (...) {
        ssd::set_error("unknown error");
        return 1;
    }
}
