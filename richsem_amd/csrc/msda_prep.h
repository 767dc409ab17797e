// msda_prep.h -- the element-wise work of the MSDeformAttn MODULE around the operator, as three gfx950 kernels.
//
// The reference module (models/richsem/ops/modules/ms_deform_attn.py:94-114) leaves this to separate PyTorch ops:
//   value.masked_fill(padding_mask)                       a full read + write of value for a handful of padded columns
//   softmax over the L*P logits of a (query, head)        read + write of (N, Lq, M, L*P)
//   offsets / normaliser (or / P * wh * 0.5), + reference three more passes over (N, Lq, M, L, P, 2)
// and autograd replays each of them backwards.  Here:
//   prep_forward   one pass: raw offsets + logits (+ reference points) -> sampling_loc, attn_weight.  A (query, head) is a
//                  group of L*P lanes; its softmax is two shuffle reductions (max, sum) inside the group.
//   prep_backward  one pass: grad_sampling_loc, grad_attn_weight -> grad_offsets, grad_logits (+ grad_reference_points).
//                  A lane group walks the M heads of a query, so the reference-point gradient (a sum over heads and points)
//                  needs no atomics.
//   mask_rows      zero the value rows (or grad_value rows) of padded pixels IN PLACE: touches only those rows.
// Offsets and logits are addressed with a row stride, so the two projections can be ONE GEMM (256 -> 384 for RichSem) whose
// output is read in place -- and whose gradient is written in place.
#pragma once

#include "msda_common.h"

namespace msda {

constexpr int kPrepMaxL = 16;

struct PrepGeom {
    int N, Lq, M, L, P;
    int ref_dim;                  // 2: (x, y) reference points; 4: (x, y, w, h) reference boxes
    int G;                        // lanes per (query, head): power of two >= L*P, <= 64
    long long off_stride, log_stride;     // elements between consecutive (n, q) rows of offsets / logits
    long long goff_stride, glog_stride;   // the same for their gradients
    float W[kPrepMaxL], H[kPrepMaxL];     // offset normaliser per level (2-d reference points)
};

// element load / store of the projection tensor in compute type T
template <typename T, typename TP>
__device__ __forceinline__ T prep_ld(const TP *p) { return (T)*p; }
template <>
__device__ __forceinline__ float prep_ld<float, bf16_t>(const bf16_t *p) { return __bfloat162float(*p); }
template <typename TP, typename T>
__device__ __forceinline__ void prep_st(TP *p, T v) { *p = (TP)v; }
__device__ __forceinline__ void prep_st(bf16_t *p, float v) { *p = __float2bfloat16(v); }

// width-G (power of two) butterfly reductions inside a wave
template <typename T>
__device__ __forceinline__ T group_max(T v, int G)
{
    for (int s = 1; s < G; s <<= 1) {
        const T o = __shfl_xor(v, s, kWave);
        v = o > v ? o : v;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T group_sum(T v, int G)
{
    for (int s = 1; s < G; s <<= 1) v += __shfl_xor(v, s, kWave);
    return v;
}

// sampling_loc (N, Lq, M, L, P, 2) and attn_weight (N, Lq, M, L, P) from raw offsets, logits and reference points.
// reference ms_deform_attn.py:100 (softmax), :102-109 (location arithmetic).
// TP: storage type of the raw projection (offsets, logits) and of its gradient: T, or bf16_t with T = float (a bf16 GEMM feeds
// the kernel; locations, weights and reference points stay T).
template <typename T, typename TP = T>
__global__ __launch_bounds__(256) void prep_forward_kernel(const TP *__restrict__ offsets, const TP *__restrict__ logits,
                                                           const T *__restrict__ ref, T *__restrict__ loc, T *__restrict__ aw,
                                                           const PrepGeom g)
{
    const int LP = g.L * g.P;
    const int lane_g = threadIdx.x & (g.G - 1);
    const long long groups_per_block = blockDim.x / g.G;
    const long long n_items = (long long)g.N * g.Lq * g.M;
    for (long long item = blockIdx.x * groups_per_block + threadIdx.x / g.G; item < n_items; item += gridDim.x * groups_per_block) {
        const long long nq = item / g.M;
        const int m = (int)(item - nq * g.M);
        const bool live = lane_g < LP;
        const int l = live ? lane_g / g.P : 0;
        T x = (T)-3.0e38;   // padding lanes: exp() = 0
        T ox = 0, oy = 0, rx = 0, ry = 0, rw = 0, rh = 0;
        if (live) {
            x = prep_ld<T>(logits + nq * g.log_stride + (long long)m * LP + lane_g);
            const TP *o = offsets + nq * g.off_stride + ((long long)m * LP + lane_g) * 2;
            ox = prep_ld<T>(o);
            oy = prep_ld<T>(o + 1);
            const T *r = ref + (nq * g.L + l) * g.ref_dim;
            rx = r[0];
            ry = r[1];
            if (g.ref_dim == 4) { rw = r[2]; rh = r[3]; }
        }
        const T mx = group_max<T>(x, g.G);
        const T e = live ? exp(x - mx) : (T)0;
        const T sum = group_sum<T>(e, g.G);
        if (live) {
            aw[item * LP + lane_g] = e / sum;
            T lx, ly;
            if (g.ref_dim == 2) {
                lx = rx + ox / (T)g.W[l];
                ly = ry + oy / (T)g.H[l];
            } else {   // same operation order as the reference: offsets / n_points * wh * 0.5
                lx = rx + ox / (T)g.P * rw * (T)0.5;
                ly = ry + oy / (T)g.P * rh * (T)0.5;
            }
            T *d = loc + (item * LP + lane_g) * 2;
            d[0] = lx;
            d[1] = ly;
        }
    }
}

// grad_offsets, grad_logits (+ grad_reference_points when grad_ref != nullptr) from grad_sampling_loc and grad_attn_weight.
// A lane group takes one (image, query) and walks its M heads.
template <typename T, typename TP = T>
__global__ __launch_bounds__(256) void prep_backward_kernel(const T *__restrict__ grad_loc, const T *__restrict__ grad_aw,
                                                            const T *__restrict__ aw, const TP *__restrict__ offsets,
                                                            const T *__restrict__ ref, TP *__restrict__ grad_offsets,
                                                            TP *__restrict__ grad_logits, T *__restrict__ grad_ref, const PrepGeom g)
{
    const int LP = g.L * g.P;
    const int lane_g = threadIdx.x & (g.G - 1);
    const long long groups_per_block = blockDim.x / g.G;
    const long long n_q = (long long)g.N * g.Lq;
    for (long long nq = blockIdx.x * groups_per_block + threadIdx.x / g.G; nq < n_q; nq += gridDim.x * groups_per_block) {
        const bool live = lane_g < LP;
        const int l = live ? lane_g / g.P : 0;
        T rw = 0, rh = 0;
        if (live && g.ref_dim == 4) {
            const T *r = ref + (nq * g.L + l) * 4;
            rw = r[2];
            rh = r[3];
        }
        T sx = 0, sy = 0, sw = 0, sh = 0;   // this lane's share of the reference-point gradient of its level
        for (int m = 0; m < g.M; ++m) {
            const long long item = nq * g.M + m;
            T ga = 0, a = 0, gx = 0, gy = 0;
            if (live) {
                ga = grad_aw[item * LP + lane_g];
                a = aw[item * LP + lane_g];
                const T *gl = grad_loc + (item * LP + lane_g) * 2;
                gx = gl[0];
                gy = gl[1];
            }
            // softmax backward: dL/dlogit_p = a_p * (g_p - sum_j a_j g_j)
            const T dot = group_sum<T>(a * ga, g.G);
            if (live) {
                prep_st(grad_logits + nq * g.glog_stride + (long long)m * LP + lane_g, a * (ga - dot));
                T dox, doy;
                if (g.ref_dim == 2) {
                    dox = gx / (T)g.W[l];
                    doy = gy / (T)g.H[l];
                } else {
                    dox = gx / (T)g.P * rw * (T)0.5;
                    doy = gy / (T)g.P * rh * (T)0.5;
                    if (grad_ref) {
                        const TP *o = offsets + nq * g.off_stride + ((long long)m * LP + lane_g) * 2;
                        sw += gx * (prep_ld<T>(o) / (T)g.P) * (T)0.5;
                        sh += gy * (prep_ld<T>(o + 1) / (T)g.P) * (T)0.5;
                    }
                }
                TP *d = grad_offsets + nq * g.goff_stride + ((long long)m * LP + lane_g) * 2;
                prep_st(d, dox);
                prep_st(d + 1, doy);
                sx += gx;
                sy += gy;
            }
        }
        if (grad_ref) {   // sum over the P lanes of a level (uniform over the wave: P need not be a power of two)
            for (int lv = 0; lv < g.L; ++lv) {
                const bool mine = live && l == lv;
                const T tx = group_sum<T>(mine ? sx : (T)0, g.G), ty = group_sum<T>(mine ? sy : (T)0, g.G);
                const T tw = group_sum<T>(mine ? sw : (T)0, g.G), th = group_sum<T>(mine ? sh : (T)0, g.G);
                if (lane_g == 0) {
                    T *d = grad_ref + (nq * g.L + lv) * g.ref_dim;
                    d[0] = tx;
                    d[1] = ty;
                    if (g.ref_dim == 4) { d[2] = tw; d[3] = th; }
                }
            }
        }
    }
}

// Zero, in place, the rows of a (rows, row_elems) tensor whose mask byte is non-zero (padded pixels).  One wave looks at 64
// mask bytes; only masked rows cost any traffic.  reference ms_deform_attn.py:95-96 (masked_fill) and its backward.
template <typename T>
__global__ __launch_bounds__(256) void mask_rows_kernel(T *__restrict__ x, const unsigned char *__restrict__ mask, long long rows,
                                                        int row_elems)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) / kWave, n_waves = gridDim.x * (long long)blockDim.x / kWave;
    for (long long r0 = wave * kWave; r0 < rows; r0 += n_waves * kWave) {
        const long long r = r0 + lane;
        unsigned long long hit = __ballot(r < rows && mask[r] != 0);
        while (hit) {   // (uniform) all 64 lanes clear one masked row together
            const int k = __ffsll((long long)hit) - 1;
            hit &= hit - 1;
            T *row = x + (r0 + k) * row_elems;
            for (int c = lane; c < row_elems; c += kWave) row[c] = (T)0;
        }
    }
}

}  // namespace msda
