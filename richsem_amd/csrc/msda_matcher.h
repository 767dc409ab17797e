// msda_matcher.h -- the Hungarian matcher's cost matrices on the device (criterion plumbing).
//
// Reference: models/richsem/matcher.py:49-78 (HungarianMatcher.forward) with util/box_ops.py:9-59 (box_cxcywh_to_xyxy, box_iou,
// generalized_box_iou).  The reference forms the cost of EVERY query of the batch against EVERY target of the batch
// ((bs * nq) x sum(T_b)), copies all of it to the host with a synchronising .cpu() and keeps the diagonal blocks only; it does
// that once per decoder output (main + 5 auxiliary + intermediate = 7 times a step, richsem.py:1136, :1204, :1255).  Here one
// thread forms one entry of a diagonal block (query q of image b against target t of the same image), with the reference's
// arithmetic in the reference's order:
//     p          = sigmoid(logit[b, q, label_t])
//     cost_class = alpha * (1 - p)^2 * (-log(p + 1e-8))  -  (1 - alpha) * p^2 * (-log(1 - p + 1e-8))
//     cost_bbox  = sum_i |box_q[i] - box_t[i]|                                      (torch.cdist, p = 1)
//     cost_giou  = -(iou - (hull - union) / (hull + 1e-6)),  iou = inter / (union + 1e-6)   (boxes as xyxy corners)
//     C          = w_bbox * cost_bbox + w_class * cost_class + w_giou * cost_giou
// and the blocks of several decoder outputs land in ONE buffer that the caller brings to the host with ONE asynchronous copy.
// Block layout: image b's (Q x T_b) block, row-major, starts at element Q * tgt_offsets[b].
#pragma once

#include <stdint.h>

#include "msda_common.h"

namespace msda {

template <typename T>
__device__ __forceinline__ T matcher_log(T x);
template <>
__device__ __forceinline__ float matcher_log<float>(float x) { return logf(x); }
template <>
__device__ __forceinline__ double matcher_log<double>(double x) { return log(x); }
template <typename T>
__device__ __forceinline__ T matcher_exp(T x);
template <>
__device__ __forceinline__ float matcher_exp<float>(float x) { return expf(x); }
template <>
__device__ __forceinline__ double matcher_exp<double>(double x) { return exp(x); }

// logits (B, Q, C); boxes (B, Q, 4) cxcywh; tgt_ids (Ttot) int64; tgt_boxes (Ttot, 4) cxcywh; tgt_offsets (B + 1) int64 prefix
// of the per-image target counts; cost: Q * Ttot elements.  A label outside [0, C) gives NaN in its column (the reference
// raises an index error there).
template <typename T>
__global__ __launch_bounds__(256) void matcher_cost_kernel(const T *__restrict__ logits, const T *__restrict__ boxes,
                                                           const int64_t *__restrict__ tgt_ids, const T *__restrict__ tgt_boxes,
                                                           const int64_t *__restrict__ tgt_offsets, int B, int Q, int C, T w_class,
                                                           T w_bbox, T w_giou, T alpha, T *__restrict__ cost)
{
    const int64_t total = (int64_t)Q * tgt_offsets[B];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int b = 0;
        while (b + 1 < B && i >= (int64_t)Q * tgt_offsets[b + 1]) ++b;
        const int64_t t0 = tgt_offsets[b];
        const int Tb = (int)(tgt_offsets[b + 1] - t0);
        const int64_t r = i - (int64_t)Q * t0;
        const int q = (int)(r / Tb), t = (int)(r - (int64_t)q * Tb);
        const int64_t label = tgt_ids[t0 + t];
        const T *ob = boxes + ((int64_t)b * Q + q) * 4, *tb = tgt_boxes + (t0 + t) * 4;
        T cc;
        if (label < 0 || label >= C) {
            cc = (T)NAN;
        } else {
            const T x = logits[((int64_t)b * Q + q) * C + label];
            const T p = (T)1 / ((T)1 + matcher_exp<T>(-x));
            const T neg = ((T)1 - alpha) * (p * p) * (-matcher_log<T>((T)1 - p + (T)1e-8));
            const T pos = alpha * (((T)1 - p) * ((T)1 - p)) * (-matcher_log<T>(p + (T)1e-8));
            cc = pos - neg;
        }
        const T ocx = ob[0], ocy = ob[1], ow = ob[2], oh = ob[3], tcx = tb[0], tcy = tb[1], tw = tb[2], th = tb[3];
        const T cb = fabs(ocx - tcx) + fabs(ocy - tcy) + fabs(ow - tw) + fabs(oh - th);
        const T ox0 = ocx - (T)0.5 * ow, oy0 = ocy - (T)0.5 * oh, ox1 = ocx + (T)0.5 * ow, oy1 = ocy + (T)0.5 * oh;
        const T tx0 = tcx - (T)0.5 * tw, ty0 = tcy - (T)0.5 * th, tx1 = tcx + (T)0.5 * tw, ty1 = tcy + (T)0.5 * th;
        const T area_o = (ox1 - ox0) * (oy1 - oy0), area_t = (tx1 - tx0) * (ty1 - ty0);
        const T iw = fmax(fmin(ox1, tx1) - fmax(ox0, tx0), (T)0), ih = fmax(fmin(oy1, ty1) - fmax(oy0, ty0), (T)0);
        const T inter = iw * ih;
        const T uni = area_o + area_t - inter;
        const T iou = inter / (uni + (T)1e-6);
        const T hw = fmax(fmax(ox1, tx1) - fmin(ox0, tx0), (T)0), hh = fmax(fmax(oy1, ty1) - fmin(oy0, ty0), (T)0);
        const T hull = hw * hh;
        const T giou = iou - (hull - uni) / (hull + (T)1e-6);
        cost[i] = w_bbox * cb + w_class * cc + w_giou * (-giou);
    }
}

}  // namespace msda
