// VOC07 11-point average precision on the GPU (metrics.hip).
#pragma once
#include "common.h"

namespace ssd {
void average_precision_device(int n_det, int n_gt, int ncls, const float* det_box, const float* det_conf, const int* det_cls,
                              const int* det_sample, const double* gt_box, const int* gt_cls, const int* gt_sample,
                              double minoverlap, unsigned long long* keys, int n2, unsigned char* matched, double* ap,
                              int* present, hipStream_t s);
}
