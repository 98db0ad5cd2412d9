// GPU training augmentation (augment.hip) -- host interface.
#pragma once
#include "common.h"
#include "../../include/ssdvgg_hip.h"

namespace ssd {
size_t augment_ws_bytes(int b, int out_w, int out_h);
void augment_batch(const unsigned char* images_dev, const ssd_augment_params* params_host, int b, int out_w, int out_h, float* out_dev,
                   void* ws, hipStream_t s);
}  // namespace ssd
