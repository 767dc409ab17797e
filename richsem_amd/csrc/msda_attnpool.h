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
constexpr int kAttnPoolMaxT = 256;       // (4 heads x (4 waves of partial rows + scores) in f64: 41 KB of the 64 KB a launch gets without opting in)

template <typename T>
__device__ __forceinline__ T attnpool_exp(T x);
template <>
__device__ __forceinline__ float attnpool_exp<float>(float x) { return expf(x); }
template <>
__device__ __forceinline__ double attnpool_exp<double>(double x) { return exp(x); }

// HG heads per workgroup: the ROI's features (C x Tn, 400 KB at CLIP-RN50's size) are read once per HG heads instead of once per head
// (HG = 1 made the kernel a stream of L2 reads: 25 MB per ROI)
template <typename T, int HG>
__global__ __launch_bounds__(kAttnPoolThreads) void attnpool_core_kernel(const T *__restrict__ u, const T *__restrict__ feat,
                                                                          const T *__restrict__ pos, const T *__restrict__ spos, int K,
                                                                          int H, int C, int Tn, int head_major, T *__restrict__ z)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char attnpool_smem[];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave, waves = kAttnPoolThreads / kWave;
    const int stride = Tn + 1;
    T *part_f = reinterpret_cast<T *>(attnpool_smem);     // [HG][waves][stride]: partial u . f_t at [1 + t]
    T *sc = part_f + HG * waves * stride;                  // [HG][stride]: scores, then attention weights
    const int groups = H / HG;
    const int k = blockIdx.x / groups, h0 = (blockIdx.x % groups) * HG;
    // u and z: (K, H, C), or (H, K, C) when head_major (what a batched product over the heads reads and writes without a transpose)
    int64_t rows[HG];
#pragma unroll
    for (int g = 0; g < HG; ++g) rows[g] = head_major ? (int64_t)(h0 + g) * K + k : (int64_t)k * H + h0 + g;
    const T *fk = feat + (int64_t)k * C * Tn;

    // lane = token, a wave takes every waves-th channel: u_g . f_t for the HG heads (the positional part u . pos_t comes in as `spos`:
    // a product the caller forms with one library GEMM -- read here per lane it would be one cache line per token and channel)
    for (int t0 = 0; t0 < Tn; t0 += kWave) {
        const int t = t0 + lane;
        if (t < Tn) {
            T af[HG];
#pragma unroll
            for (int g = 0; g < HG; ++g) af[g] = (T)0;
            for (int c = wave; c < C; c += waves) {
                const T f = fk[(int64_t)c * Tn + t];
#pragma unroll
                for (int g = 0; g < HG; ++g) af[g] += u[rows[g] * C + c] * f;
            }
#pragma unroll
            for (int g = 0; g < HG; ++g) part_f[(g * waves + wave) * stride + 1 + t] = af[g];
        }
    }
    __syncthreads();
    // fold the waves' partial sums; the feature part stays in the group's first row for the mean token's score
    for (int i = threadIdx.x; i < HG * stride; i += kAttnPoolThreads) {
        const int g = i / stride, t = i - g * stride;
        T sf = (T)0;
        if (t > 0) {
            for (int w = 0; w < waves; ++w) sf += part_f[(g * waves + w) * stride + t];
            part_f[g * waves * stride + t] = sf;          // (wave 0's row of head g, entry t: read above by this thread only)
        }
        sc[g * stride + t] = sf + spos[rows[g] * stride + t];
    }
    __syncthreads();
    // one wave per head (round robin): score of the mean token -- u . (mean_t f_t + pos_0) = mean_t (u . f_t) + u . pos_0 -- and softmax
    for (int g = wave; g < HG; g += waves) {
        T *scg = sc + g * stride;
        const T *pf = part_f + g * waves * stride;
        T m = (T)0;
        for (int t = lane; t < Tn; t += kWave) m += pf[1 + t];
        for (int o = kWave / 2; o > 0; o >>= 1) m += __shfl_xor(m, o, kWave);
        const T s0 = scg[0] + m / (T)Tn;
        T mx = s0;
        for (int t = lane; t < Tn; t += kWave) mx = fmax(mx, scg[1 + t]);
        for (int o = kWave / 2; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, kWave));
        T sum = (T)0;
        for (int t = lane; t < Tn; t += kWave) {
            const T e = attnpool_exp<T>(scg[1 + t] - mx);
            scg[1 + t] = e;
            sum += e;
        }
        for (int o = kWave / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, kWave);
        const T e0 = attnpool_exp<T>(s0 - mx);
        const T inv = (T)1 / (sum + e0);
        for (int t = lane; t < Tn; t += kWave) scg[1 + t] *= inv;
        if (lane == 0) scg[0] = e0 * inv;
    }
    __syncthreads();
    // z_g[c] = a_0 (mean_t f[c, t] + pos_0[c]) + sum_t a_{t+1} (f[c, t] + pos_{t+1}[c]): one channel per thread, the HG heads side by side
    for (int c = threadIdx.x; c < C; c += kAttnPoolThreads) {
        const T *frow = fk + (int64_t)c * Tn;
        T mean = (T)0, acc[HG];
#pragma unroll
        for (int g = 0; g < HG; ++g) acc[g] = (T)0;
        for (int t = 0; t < Tn; ++t) {
            const T f = frow[t];
            mean += f;
            const T fp = f + pos[(int64_t)(t + 1) * C + c];
#pragma unroll
            for (int g = 0; g < HG; ++g) acc[g] += sc[g * stride + 1 + t] * fp;
        }
        const T m0 = mean / (T)Tn + pos[c];
#pragma unroll
        for (int g = 0; g < HG; ++g) z[rows[g] * C + c] = sc[g * stride] * m0 + acc[g];
    }
}

}  // namespace msda
