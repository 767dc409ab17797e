"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, plain loops) of detectron2's ROIAlign forward, the third-party operator
the reference pools its CLIP feature map with (models/richsem/richsem.py:25 ``from detectron2.layers.roi_align import ROIAlign``,
:878 ``ROIAlign(output_size, spatial_scale, 0, aligned=True)``).  detectron2 is not vendored in the reference tree (its
requirements name it without a pinned version) and is absent from this image; the algorithm below is its published
``ROIAlign_cpu.cpp`` / ``ROIAlign_cuda.cu`` (identical to torchvision.ops.roi_align).

PARITY UNPINNED: neither detectron2 nor torchvision can be imported here and the reference holds no fixture for this operator.
"""
import math

import numpy as np


def _bilinear(plane, y, x):
    H, W = plane.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return 0.0
    y, x = max(y, 0.0), max(x, 0.0)
    y_low, x_low = int(y), int(x)
    if y_low >= H - 1:
        y_high = y_low = H - 1
        y = float(y_low)
    else:
        y_high = y_low + 1
    if x_low >= W - 1:
        x_high = x_low = W - 1
        x = float(x_low)
    else:
        x_high = x_low + 1
    ly, lx = y - y_low, x - x_low
    hy, hx = 1.0 - ly, 1.0 - lx
    return (hy * hx * plane[y_low, x_low] + hy * lx * plane[y_low, x_high] + ly * hx * plane[y_high, x_low]
            + ly * lx * plane[y_high, x_high])


def roi_align(inp, rois, output_size, spatial_scale, sampling_ratio=0, aligned=True):
    inp, rois = np.asarray(inp, np.float64), np.asarray(rois, np.float64)
    ph_n, pw_n = (output_size, output_size) if isinstance(output_size, int) else output_size
    K, C = rois.shape[0], inp.shape[1]
    out = np.zeros((K, C, ph_n, pw_n))
    for k in range(K):
        b = int(rois[k, 0])
        off = 0.5 if aligned else 0.0
        sw, sh, ew, eh = (rois[k, 1] * spatial_scale - off, rois[k, 2] * spatial_scale - off, rois[k, 3] * spatial_scale - off,
                          rois[k, 4] * spatial_scale - off)
        rw, rh = ew - sw, eh - sh
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / ph_n, rw / pw_n
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph_n))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw_n))
        count = max(gh * gw, 1)
        for c in range(C):
            plane = inp[b, c]
            for ph in range(ph_n):
                for pw in range(pw_n):
                    acc = 0.0
                    for iy in range(gh):
                        y = sh + ph * bh + (iy + 0.5) * bh / gh
                        for ix in range(gw):
                            x = sw + pw * bw + (ix + 0.5) * bw / gw
                            acc += _bilinear(plane, y, x)
                    out[k, c, ph, pw] = acc / count
    return out
