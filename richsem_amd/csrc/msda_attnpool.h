// msda_attnpool.h -- the attention core of CLIP's AttentionPool2d for a single query token (reference
// clip/model.py:58-91, called on the ROI features of the ground-truth boxes at models/richsem/richsem.py:753).
//
// The reference runs F.multi_head_attention_forward with query = token 0 only (the mean token) and keys = values = all HW + 1
// tokens, which projects every token twice ((HW + 1) x C x C products for k and for v).  Only ONE query row per head exists, so the
// two projections can be moved to the other side of the attention:
//     score[h, t] = q_h . (Wk_h x_t + bk_h) = (Wk_h^T q_h) . x_t + const            -> u_h = Wk_h^T q_h       (one C x C product per ROI)
//     out_h       = sum_t a[h, t] (Wv_h x_t + bv_h) = Wv_h (sum_t a[h, t] x_t) + bv_h -> z_h = sum_t a[h, t] x_t (then one C x C product)
// (the constant q_h . bk_h does not change the softmax).  That is 3 instead of 2 (HW + 1) + 1 C x C products per ROI: 33 x fewer
// FLOPs at HW = 49.  This kernel is the part between the products: tokens x_0 = mean_t f_t + pos_0, x_{t+1} = f_t + pos_{t+1} are
// never materialised, scores -> softmax -> z per (ROI, head) in one workgroup.
//   u    (K, H, C)   per-head key-side query (already scaled by head_dim^-1/2)
//   feat (K, C, T)   ROI features as the ROIAlign kernel writes them (channel-major, T = HW positions)
//   pos  (T + 1, C)  positional embedding
//   z    (K, H, C)   attention-weighted token sum
#pragma once

#include <stdint.h>

#include "msda_common.h"

namespace msda {

constexpr int kAttnPoolThreads = 256;
constexpr int kAttnPoolMaxT = 512;       // (2 waves' worth of partial rows + scores in f64: 37 KB of the 64 KB a launch gets without opting in)

template <typename T>
__device__ __forceinline__ T attnpool_exp(T x);
template <>
__device__ __forceinline__ float attnpool_exp<float>(float x) { return expf(x); }
template <>
__device__ __forceinline__ double attnpool_exp<double>(double x) { return exp(x); }

template <typename T>
__global__ __launch_bounds__(kAttnPoolThreads) void attnpool_core_kernel(const T *__restrict__ u, const T *__restrict__ feat,
                                                                          const T *__restrict__ pos, int H, int C, int Tn,
                                                                          T *__restrict__ z)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char attnpool_smem[];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave, waves = kAttnPoolThreads / kWave;
    const int stride = Tn + 1;
    T *part_f = reinterpret_cast<T *>(attnpool_smem);     // [waves][stride]: partial u . f_t at [1 + t]
    T *part_p = part_f + waves * stride;                   // [waves][stride]: partial u . pos_t' at [t']
    T *sc = part_p + waves * stride;                       // [stride]: scores, then attention weights
    const int k = blockIdx.x / H, h = blockIdx.x % H;
    const T *uh = u + ((int64_t)k * H + h) * C;
    const T *fk = feat + (int64_t)k * C * Tn;

    // lane = token, a wave takes every waves-th channel: u . f_t and u . pos_{t+1} side by side
    for (int t0 = 0; t0 < Tn; t0 += kWave) {
        const int t = t0 + lane;
        if (t < Tn) {
            T af = (T)0, ap = (T)0;
            for (int c = wave; c < C; c += waves) {
                const T uc = uh[c];
                af += uc * fk[(int64_t)c * Tn + t];
                ap += uc * pos[(int64_t)(t + 1) * C + c];
            }
            part_f[wave * stride + 1 + t] = af;
            part_p[wave * stride + 1 + t] = ap;
        }
    }
    {   // u . pos_0
        T acc = (T)0;
        for (int c = lane + wave * kWave; c < C; c += kAttnPoolThreads) acc += uh[c] * pos[c];
        for (int o = kWave / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
        if (lane == 0) part_p[wave * stride] = acc;
    }
    __syncthreads();
    // fold the waves' partial sums; the feature part stays in part_f's first row for the mean token's score
    for (int t = threadIdx.x; t <= Tn; t += kAttnPoolThreads) {
        T sf = (T)0, sp = (T)0;
        for (int w = 0; w < waves; ++w) {
            if (t > 0) sf += part_f[w * stride + t];
            sp += part_p[w * stride + t];
        }
        if (t > 0) part_f[t] = sf;          // (row 0, entry t: read above by this thread only)
        sc[t] = sf + sp;
    }
    __syncthreads();
    if (wave == 0) {   // score of the mean token: u . (mean_t f_t + pos_0) = mean_t (u . f_t) + u . pos_0
        T m = (T)0;
        for (int t = lane; t < Tn; t += kWave) m += part_f[1 + t];
        for (int o = kWave / 2; o > 0; o >>= 1) m += __shfl_xor(m, o, kWave);
        if (lane == 0) sc[0] += m / (T)Tn;
    }
    __syncthreads();
    // softmax over the Tn + 1 scores (one wave)
    if (wave == 0) {
        T mx = -INFINITY;
        for (int t = lane; t <= Tn; t += kWave) mx = fmax(mx, sc[t]);
        for (int o = kWave / 2; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, kWave));
        T sum = (T)0;
        for (int t = lane; t <= Tn; t += kWave) {
            const T e = attnpool_exp<T>(sc[t] - mx);
            sc[t] = e;
            sum += e;
        }
        for (int o = kWave / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, kWave);
        const T inv = (T)1 / sum;
        for (int t = lane; t <= Tn; t += kWave) sc[t] *= inv;
    }
    __syncthreads();
    // z[c] = a_0 (mean_t f[c, t] + pos_0[c]) + sum_t a_{t+1} (f[c, t] + pos_{t+1}[c]): one channel per thread
    T *zh = z + ((int64_t)k * H + h) * C;
    const T a0 = sc[0];
    for (int c = threadIdx.x; c < C; c += kAttnPoolThreads) {
        const T *row = fk + (int64_t)c * Tn;
        T mean = (T)0, acc = (T)0;
        for (int t = 0; t < Tn; ++t) {
            const T f = row[t];
            mean += f;
            acc += sc[1 + t] * (f + pos[(int64_t)(t + 1) * C + c]);
        }
        zh[c] = a0 * (mean / (T)Tn + pos[c]) + acc;
    }
}

}  // namespace msda
