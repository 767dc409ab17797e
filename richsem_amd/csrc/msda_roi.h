// msda_roi.h -- ROIAlign forward: the bilinear box pooling RichSem applies to the frozen CLIP feature
// map to form its distillation targets (reference models/richsem/richsem.py:750, :878:
// detectron2.layers.ROIAlign(output_size = 7, spatial_scale = 1/32, sampling_ratio = 0, aligned = True)).  detectron2 is a
// third-party dependency that is not part of the reference tree (and is absent from the image); the algorithm restated here is
// its published ROIAlign forward (detectron2/layers/csrc/ROIAlignRotated ... ROIAlign_cuda.cu = torchvision's roi_align):
//   * box (x1, y1, x2, y2) in input pixels, scaled by spatial_scale, shifted by -0.5 when aligned;
//   * every output bin averages a grid of sampling_ratio^2 (or ceil(roi / bins)^2 when sampling_ratio = 0) bilinear samples;
//   * a sample more than one pixel outside the map contributes 0, coordinates are clamped into [0, size - 1] otherwise.
// Forward only: the teacher features carry no gradient.
#pragma once

#include <stdint.h>

#include "msda_common.h"

namespace msda {

template <typename T>
__device__ __forceinline__ T roi_bilinear(const T *__restrict__ plane, int height, int width, T y, T x)
{
    if (y < (T)-1.0 || y > (T)height || x < (T)-1.0 || x > (T)width) return (T)0;
    if (y <= (T)0) y = (T)0;
    if (x <= (T)0) x = (T)0;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= height - 1) { y_high = y_low = height - 1; y = (T)y_low; } else { y_high = y_low + 1; }
    if (x_low >= width - 1) { x_high = x_low = width - 1; x = (T)x_low; } else { x_high = x_low + 1; }
    const T ly = y - (T)y_low, lx = x - (T)x_low, hy = (T)1 - ly, hx = (T)1 - lx;
    const T v1 = plane[y_low * width + x_low], v2 = plane[y_low * width + x_high];
    const T v3 = plane[y_high * width + x_low], v4 = plane[y_high * width + x_high];
    return hy * hx * v1 + hy * lx * v2 + ly * hx * v3 + ly * lx * v4;
}

// input (N, C, H, W); rois (K, 5) = (batch index, x1, y1, x2, y2); output (K, C, PH, PW).  One thread per output element.
template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(const T *__restrict__ input, const T *__restrict__ rois, int64_t n_out,
                                                            int N, int C, int H, int W, int PH, int PW, T spatial_scale,
                                                            int sampling_ratio, int aligned, T *__restrict__ output)
{
    for (int64_t index = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; index < n_out; index += (int64_t)gridDim.x * blockDim.x) {
        const int pw = (int)(index % PW), ph = (int)((index / PW) % PH), c = (int)((index / PW / PH) % C);
        const int64_t k = index / PW / PH / C;
        const T *roi = rois + k * 5;
        const int b = (int)roi[0];
        if (b < 0 || b >= N) {   // (the reference would read out of bounds; here: zeros)
            output[index] = (T)0;
            continue;
        }
        const T offset = aligned ? (T)0.5 : (T)0.0;
        const T start_w = roi[1] * spatial_scale - offset, start_h = roi[2] * spatial_scale - offset;
        const T end_w = roi[3] * spatial_scale - offset, end_h = roi[4] * spatial_scale - offset;
        T roi_w = end_w - start_w, roi_h = end_h - start_h;
        if (!aligned) {
            roi_w = roi_w > (T)1 ? roi_w : (T)1;
            roi_h = roi_h > (T)1 ? roi_h : (T)1;
        }
        const T bin_h = roi_h / (T)PH, bin_w = roi_w / (T)PW;
        const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceil(roi_h / (T)PH);
        const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceil(roi_w / (T)PW);
        const T count = (T)(grid_h * grid_w > 1 ? grid_h * grid_w : 1);
        const T *plane = input + ((int64_t)b * C + c) * H * W;
        T acc = (T)0;
        for (int iy = 0; iy < grid_h; ++iy) {
            const T y = start_h + (T)ph * bin_h + ((T)iy + (T)0.5) * bin_h / (T)grid_h;
            for (int ix = 0; ix < grid_w; ++ix) {
                const T x = start_w + (T)pw * bin_w + ((T)ix + (T)0.5) * bin_w / (T)grid_w;
                acc += roi_bilinear<T>(plane, H, W, y, x);
            }
        }
        output[index] = acc / count;
    }
}

}  // namespace msda
