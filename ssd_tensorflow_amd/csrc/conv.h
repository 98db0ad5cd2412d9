// im2col-free direct convolution on gfx950 fp32 MFMA -- host interface.
// Replaces the TensorFlow kernels behind tf.nn.conv2d / atrous_conv2d / bias_add /
// relu at ssdvgg.py:48-50, 61-62, 260-262, 287-290 and their gradients.
#pragma once
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

}  // namespace ssd
