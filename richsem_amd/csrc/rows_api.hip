// rows_api.hip -- C ABI of the rows around the operator that SURVEY.md section 8 marks "next" (declared in
// include/richsem_msda.h): the matcher's cost blocks (section 8f rank 4), the attention-pool core (rank 3).  A translation unit of its own so that the operator's
// kernels (msda_api.hip) are not rebuilt with it.  Error reporting: return codes only (msda_last_error covers msda_api.hip's calls).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "../../include/richsem_msda.h"
#include "msda_attnpool.h"
#include "msda_matcher.h"

namespace {

template <typename T>
int matcher_cost_impl(const T *logits, const T *boxes, const int64_t *tgt_ids, const T *tgt_boxes, const int64_t *tgt_offsets, int B,
                      int Q, int C, int64_t n_targets, double w_class, double w_bbox, double w_giou, double alpha, T *cost,
                      msda_stream_t stream)
{
    if (!logits || !boxes || !tgt_offsets || !cost) return MSDA_ERR_NULL_POINTER;
    if (B < 1 || Q < 0 || C < 1 || n_targets < 0) return MSDA_ERR_BAD_DIMS;
    if (n_targets > 0 && (!tgt_ids || !tgt_boxes)) return MSDA_ERR_NULL_POINTER;
    const int64_t total = (int64_t)Q * n_targets;
    if (total == 0) return MSDA_OK;
    if (total >= ((int64_t)1 << 40)) return MSDA_ERR_TOO_LARGE;
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(msda::matcher_cost_kernel<T>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), logits, boxes, tgt_ids,
                       tgt_boxes, tgt_offsets, B, Q, C, (T)w_class, (T)w_bbox, (T)w_giou, (T)alpha, cost);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

template <typename T>
int attnpool_core_impl(const T *u, const T *feat, const T *pos, const T *spos, int K, int H, int C, int Tn, int head_major, T *z,
                       msda_stream_t stream)
{
    if (!u || !feat || !pos || !spos || !z) return MSDA_ERR_NULL_POINTER;
    if (K < 0 || H < 1 || C < 1 || Tn < 1 || Tn > msda::kAttnPoolMaxT) return MSDA_ERR_BAD_DIMS;
    if (K == 0) return MSDA_OK;
    if ((int64_t)K * H >= ((int64_t)1 << 31) || (int64_t)K * C * Tn >= ((int64_t)1 << 40)) return MSDA_ERR_TOO_LARGE;
    const int waves = msda::kAttnPoolThreads / msda::kWave;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (H % 4 == 0) {
        const size_t lds = (size_t)4 * (waves + 1) * (Tn + 1) * sizeof(T);
        hipLaunchKernelGGL((msda::attnpool_core_kernel<T, 4>), dim3(K * (H / 4)), dim3(msda::kAttnPoolThreads), lds, st, u, feat, pos, spos, K,
                           H, C, Tn, head_major, z);
    } else {
        const size_t lds = (size_t)(waves + 1) * (Tn + 1) * sizeof(T);
        hipLaunchKernelGGL((msda::attnpool_core_kernel<T, 1>), dim3(K * H), dim3(msda::kAttnPoolThreads), lds, st, u, feat, pos, spos, K, H, C,
                           Tn, head_major, z);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

}  // namespace

extern "C" {

int msda_attnpool_core_f32(const float *u, const float *feat, const float *pos, const float *spos, int K, int H, int C, int T,
                           int head_major, float *z, msda_stream_t stream)
{
    return attnpool_core_impl<float>(u, feat, pos, spos, K, H, C, T, head_major, z, stream);
}
int msda_attnpool_core_f64(const double *u, const double *feat, const double *pos, const double *spos, int K, int H, int C, int T,
                           int head_major, double *z, msda_stream_t stream)
{
    return attnpool_core_impl<double>(u, feat, pos, spos, K, H, C, T, head_major, z, stream);
}

int msda_matcher_cost_f32(const float *logits, const float *boxes, const int64_t *tgt_ids, const float *tgt_boxes,
                          const int64_t *tgt_offsets, int B, int Q, int C, int64_t n_targets, double w_class, double w_bbox,
                          double w_giou, double alpha, float *cost, msda_stream_t stream)
{
    return matcher_cost_impl<float>(logits, boxes, tgt_ids, tgt_boxes, tgt_offsets, B, Q, C, n_targets, w_class, w_bbox, w_giou, alpha,
                                    cost, stream);
}
int msda_matcher_cost_f64(const double *logits, const double *boxes, const int64_t *tgt_ids, const double *tgt_boxes,
                          const int64_t *tgt_offsets, int B, int Q, int C, int64_t n_targets, double w_class, double w_bbox,
                          double w_giou, double alpha, double *cost, msda_stream_t stream)
{
    return matcher_cost_impl<double>(logits, boxes, tgt_ids, tgt_boxes, tgt_offsets, B, Q, C, n_targets, w_class, w_bbox, w_giou, alpha,
                                     cost, stream);
}

}  // extern "C"
