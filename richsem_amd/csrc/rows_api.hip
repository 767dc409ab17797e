// rows_api.hip -- C ABI of the rows around the operator that SURVEY.md section 8 marks "next" (declared in
// include/richsem_msda.h): the matcher's cost blocks (section 8f rank 4), the attention-pool core (rank 3).  A translation unit of its own so that the operator's
// kernels (msda_api.hip) are not rebuilt with it.  Error reporting: return codes + msda_note_error() (msda_api.hip), so msda_last_error() names the failed call.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include <algorithm>
#include <cstdint>

#include "../../include/richsem_msda.h"

extern "C" int msda_note_error(int code, const char *entry);      // msda_api.hip: sets msda_last_error()
#include "msda_attnpool.h"
#include "msda_matcher.h"

namespace {

template <typename T>
int matcher_cost_impl(const T *logits, const T *boxes, const int64_t *tgt_ids, const T *tgt_boxes, const int64_t *tgt_offsets, int B,
                      int Q, int C, int64_t n_targets, double w_class, double w_bbox, double w_giou, double alpha, T *cost,
                      msda_stream_t stream)
{
    if (!logits || !boxes || !tgt_offsets || !cost) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (B < 1 || Q < 0 || C < 1 || n_targets < 0) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if (n_targets > 0 && (!tgt_ids || !tgt_boxes)) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    const int64_t total = (int64_t)Q * n_targets;
    if (total == 0) return MSDA_OK;
    if (total >= ((int64_t)1 << 40)) return msda_note_error(MSDA_ERR_TOO_LARGE, __func__);
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
    if (!u || !feat || !pos || !spos || !z) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (K < 0 || H < 1 || C < 1 || Tn < 1 || Tn > msda::kAttnPoolMaxT) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if (K == 0) return MSDA_OK;
    if ((int64_t)K * H >= ((int64_t)1 << 31) || (int64_t)K * C * Tn >= ((int64_t)1 << 40)) return msda_note_error(MSDA_ERR_TOO_LARGE, __func__);
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

// gen_sineembed_for_position (models/richsem/utils.py:142-168): one thread per (token, pair of channels) writes sin | cos of
// coordinate * 2 pi / T^(2 i / pe_dim) as a packed pair of bf16
__global__ __launch_bounds__(256) void sine_embed_kernel(const float *__restrict__ boxes, int ld, int tokens, int dims, int pe_dim,
                                                         float log2_temperature, unsigned *__restrict__ out)
{
    const int half = pe_dim / 2, per_tok = dims * half;
    const long long n = (long long)tokens * per_tok;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
        const int t = (int)(i / per_tok), r = (int)(i - (long long)t * per_tok);
        const int part = r / half, k = r - part * half;                 // output block `part` = (y, x, w, h); channels 2k, 2k + 1
        const int src = part < 2 ? 1 - part : part;                      // ... of the box's (x, y, w, h)
        const float v = boxes[(long long)t * ld + src] * 6.283185307179586f;
        const float p = v * exp2f(-log2_temperature * (float)(2 * k) / (float)pe_dim);
        const __hip_bfloat16 a = __float2bfloat16(sinf(p)), b = __float2bfloat16(cosf(p));
        out[i] = (unsigned)__bfloat16_as_ushort(a) | (unsigned)__bfloat16_as_ushort(b) << 16;
    }
}

// the decoder's box update (deformable_transformer.py:779-804 / richsem.py:705-715): y = sigmoid(delta + inverse_sigmoid(ref)) with
// inverse_sigmoid(r) = log(max(clamp(r, 0, 1), eps) / max(1 - clamp(r, 0, 1), eps)) (util/misc.py:605-609); and its gradient w.r.t. delta
template <bool BF16>
__global__ __launch_bounds__(256) void box_refine_kernel(const void *__restrict__ delta, const float *__restrict__ ref, float eps, long long n,
                                                         float *__restrict__ out)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
        const float d = BF16 ? __bfloat162float(static_cast<const __hip_bfloat16 *>(delta)[i]) : static_cast<const float *>(delta)[i];
        const float r = fminf(fmaxf(ref[i], 0.f), 1.f);
        const float u = d + logf(fmaxf(r, eps) / fmaxf(1.f - r, eps));
        out[i] = 1.f / (1.f + expf(-u));
    }
}

// ... and, when `ref` carries a gradient too (the heads' boxes of decoder layers 1..5, richsem.py:705-715: the "look forward twice"
// reference is not detached there), w.r.t. ref: d inverse_sigmoid / d r = [r_c >= eps] / max(r_c, eps) + [1 - r_c >= eps] / max(1 - r_c, eps)
// inside [0, 1] (torch's clamp passes the gradient where the bound is not active, bounds included), 0 outside
template <bool BF16>
__global__ __launch_bounds__(256) void box_refine_grad_kernel(const float *__restrict__ gy, const float *__restrict__ y, long long n,
                                                              void *__restrict__ gdelta, const float *__restrict__ ref, float eps,
                                                              float *__restrict__ gref)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
        const float v = gy[i] * y[i] * (1.f - y[i]);
        if (BF16) static_cast<__hip_bfloat16 *>(gdelta)[i] = __float2bfloat16(v);
        else static_cast<float *>(gdelta)[i] = v;
        if (gref) {
            const float r0 = ref[i], r = fminf(fmaxf(r0, 0.f), 1.f);
            const float dr = (r >= eps ? 1.f / fmaxf(r, eps) : 0.f) + (1.f - r >= eps ? 1.f / fmaxf(1.f - r, eps) : 0.f);
            gref[i] = (r0 >= 0.f && r0 <= 1.f) ? v * dr : 0.f;
        }
    }
}

// Backward of a 256 -> n linear layer with n <= 8 (the box heads' last layer, 256 -> 4): the input gradient is an n-term sum per channel,
// the weight gradient n x 256 token sums -- streams of x / dx, not GEMMs (the library's GEMM takes 150 us for the 44646-token one, padded to 64).
__global__ __launch_bounds__(256) void narrow_linear_dx_kernel(const uint16_t *__restrict__ dy, const float *__restrict__ w, int T, int n,
                                                               uint16_t *__restrict__ dx)
{
    __shared__ float ws[8 * 256];
    for (int i = threadIdx.x; i < n * 256; i += 256) ws[i] = w[i];
    __syncthreads();
    const long long total = (long long)T * 32;      // a thread: one token, 8 channels
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
        const long long t = i >> 5;
        const int c0 = 8 * (int)(i & 31);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < n; ++j) {
            const float d = __uint_as_float((unsigned)dy[t * n + j] << 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(d, ws[j * 256 + c0 + e], acc[e]);
        }
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = (unsigned)__bfloat16_as_ushort(__float2bfloat16(acc[2 * e])) | (unsigned)__bfloat16_as_ushort(__float2bfloat16(acc[2 * e + 1])) << 16;
        *reinterpret_cast<uint4 *>(dx + t * 256 + c0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// dw[j][c] += sum over the workgroup's tokens of dy[t][j] x[t][c] (thread = channel c), db[j] likewise (threads j < n).
// Eight tokens per trip with all of their loads issued before the first product (token index clamped, the surplus tokens weighted 0): as a
// plain loop over the tokens every trip waited for its own two loads -- 42.6 us for 2184 tokens (round 4: the composed step's twelve calls).
__global__ __launch_bounds__(256) void narrow_linear_dw_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x, int T, int n,
                                                               int chunk, float *__restrict__ dw, float *__restrict__ db)
{
    const int c = threadIdx.x;
    const long long t0 = (long long)blockIdx.x * chunk, t1 = t0 + chunk < T ? t0 + chunk : T;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const int cb = c < n ? c : 0;
    for (long long t = t0; t < t1; t += 8) {
        float xv[8], bv[8];
        unsigned short dv[8][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long tt = t + u < t1 ? t + u : t1 - 1;
            xv[u] = __uint_as_float((unsigned)x[tt * 256 + c] << 16);
            bv[u] = __uint_as_float((unsigned)dy[tt * n + cb] << 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) dv[u][j] = dy[tt * n + (j < n ? j : 0)];      // (wave-uniform addresses: scalar loads)
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float live = t + u < t1 ? 1.f : 0.f;
            const float xs = xv[u] * live;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < n) acc[j] = fmaf(__uint_as_float((unsigned)dv[u][j] << 16), xs, acc[j]);
            bsum += bv[u] * live;
        }
    }
    for (int j = 0; j < n; ++j) atomicAdd(dw + j * 256 + c, acc[j]);
    if (db && c < n) atomicAdd(db + c, bsum);
}

}  // namespace


// ---- the all-negative term of the sigmoid focal loss over a whole logit tensor (criterion plumbing) ------------
// sigmoid_focal_loss (reference models/richsem/utils.py / richsem.py:1124-1160) at a NEGATIVE entry is (1 - alpha) p^2 softplus(x), p = sigmoid(x);
// the criterion sums it over every (output, image, query, class) with a weight per ROW (1 / num_boxes for the matching queries, 1 / (num_boxes
// x groups) for the denoising queries' positive slots, 0 for the slots that carry no loss) and corrects the few positive entries separately.
// As PyTorch ops that is sigmoid, softplus, three products, a slice and a sum forward and as many kernels backward over 63 MB of logits; here one
// pass each way.   forward: partial[block] = sum_rows w[row] sum_c (1 - alpha) p^2 softplus(x)   (fp64 partials, summed by the caller)
//                  backward: grad_x = g w[row] (1 - alpha) p^2 (2 (1 - p) softplus(x) + p)
__device__ __forceinline__ float focal_neg(float x, float &dfdx)
{
    const float p = 1.f / (1.f + __expf(-x));
    const float sp = x > 20.f ? x : log1pf(__expf(x));      // softplus, as torch (threshold 20)
    const float pp = p * p;
    dfdx = pp * (2.f * (1.f - p) * sp + p);
    return pp * sp;
}

__global__ __launch_bounds__(256) void focal_neg_sum_kernel(const float *__restrict__ x, const float *__restrict__ w, long long rows, int C, float one_m_alpha,
                                                            double *__restrict__ partial)
{
    __shared__ double red[256];
    double acc = 0.0;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const float wr = w[r];
        if (wr == 0.f) continue;      // (block-uniform)
        float s = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) {
            float d;
            s += focal_neg(x[r * C + c], d);
        }
        acc += (double)(s * wr);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] * (double)one_m_alpha;
}

__global__ __launch_bounds__(256) void focal_neg_grad_kernel(const float *__restrict__ x, const float *__restrict__ w, long long rows, int C, float one_m_alpha,
                                                             const float *__restrict__ gscale, float *__restrict__ gx)
{
    const float g = gscale[0] * one_m_alpha;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const float wr = w[r] * g;
        for (int c = threadIdx.x; c < C; c += 256) {
            float d = 0.f;
            if (wr != 0.f) focal_neg(x[r * C + c], d);
            gx[r * C + c] = d * wr;
        }
    }
}

// ---- the criterion's per-pair tails as one kernel each (verdict item 3: "one kernel for the stacked focal + L1 + GIoU tails") -------------
// Box loss of K matched pairs (reference SetCriterion.loss_boxes, models/richsem/richsem.py:1162-1188 with util/box_ops.py:9-64 on the
// diagonal):   sum_k w[k] * ( c_l1 * |p_k - t_k|_1 + c_giou * (1 - GIoU(xyxy(p_k), xyxy(t_k))) ),   p, t = (cx, cy, w, h)
// and its gradient w.r.t. p, by forward-mode differentiation with four tangents per value (torch's conventions: |x|' = sign x, a maximum /
// minimum passes the gradient to the larger / smaller operand and halves it on a tie, clamp(min = 0) passes it where x >= 0).
struct Dual4 {
    float v, d[4];
};
__device__ __forceinline__ Dual4 d4_const(float v) { return Dual4{v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Dual4 d4_var(float v, int i) { Dual4 r = d4_const(v); r.d[i] = 1.f; return r; }
__device__ __forceinline__ Dual4 operator+(Dual4 a, Dual4 b) { return Dual4{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2], a.d[3] + b.d[3]}}; }
__device__ __forceinline__ Dual4 operator-(Dual4 a, Dual4 b) { return Dual4{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2], a.d[3] - b.d[3]}}; }
__device__ __forceinline__ Dual4 operator*(Dual4 a, Dual4 b)
{
    return Dual4{a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2], a.d[3] * b.v + a.v * b.d[3]}};
}
__device__ __forceinline__ Dual4 operator/(Dual4 a, Dual4 b)
{
    const float q = a.v / b.v, ib = 1.f / b.v;
    return Dual4{q, {(a.d[0] - q * b.d[0]) * ib, (a.d[1] - q * b.d[1]) * ib, (a.d[2] - q * b.d[2]) * ib, (a.d[3] - q * b.d[3]) * ib}};
}
__device__ __forceinline__ Dual4 d4_scale(Dual4 a, float s) { return Dual4{a.v * s, {a.d[0] * s, a.d[1] * s, a.d[2] * s, a.d[3] * s}}; }
__device__ __forceinline__ Dual4 d4_max(Dual4 a, Dual4 b) { return a.v > b.v ? a : (b.v > a.v ? b : d4_scale(a + b, 0.5f)); }
__device__ __forceinline__ Dual4 d4_min(Dual4 a, Dual4 b) { return a.v < b.v ? a : (b.v < a.v ? b : d4_scale(a + b, 0.5f)); }
__device__ __forceinline__ Dual4 d4_relu(Dual4 a) { return a.v >= 0.f ? a : d4_const(0.f); }
__device__ __forceinline__ Dual4 d4_abs(Dual4 a) { return a.v > 0.f ? a : (a.v < 0.f ? d4_scale(a, -1.f) : d4_const(0.f)); }

__device__ __forceinline__ float block_sum_1024(float v, float *red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x < 64) {
        s = threadIdx.x < blockDim.x / 64 ? red[threadIdx.x] : 0.f;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    }
    return s;      // (valid in thread 0)
}

// one workgroup (K is thousands: the matched, two-stage and denoising pairs of a step): loss[0] and grad_p (K, 4) = d loss / d p
__global__ __launch_bounds__(1024) void box_pair_loss_kernel(const float *__restrict__ p, const float *__restrict__ t, const float *__restrict__ w, int K,
                                                             float c_l1, float c_giou, float *__restrict__ loss, float *__restrict__ grad_p)
{
    __shared__ float red[16];
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float4 pv = reinterpret_cast<const float4 *>(p)[k], tv = reinterpret_cast<const float4 *>(t)[k];
        const Dual4 cx = d4_var(pv.x, 0), cy = d4_var(pv.y, 1), pw = d4_var(pv.z, 2), ph = d4_var(pv.w, 3);
        const Dual4 l1 = d4_abs(cx - d4_const(tv.x)) + d4_abs(cy - d4_const(tv.y)) + d4_abs(pw - d4_const(tv.z)) + d4_abs(ph - d4_const(tv.w));
        const Dual4 half = d4_const(0.5f);
        const Dual4 ax0 = cx - half * pw, ay0 = cy - half * ph, ax1 = cx + half * pw, ay1 = cy + half * ph;
        const Dual4 bx0 = d4_const(tv.x - 0.5f * tv.z), by0 = d4_const(tv.y - 0.5f * tv.w), bx1 = d4_const(tv.x + 0.5f * tv.z), by1 = d4_const(tv.y + 0.5f * tv.w);
        const Dual4 area_a = (ax1 - ax0) * (ay1 - ay0), area_b = (bx1 - bx0) * (by1 - by0);
        const Dual4 iw = d4_relu(d4_min(ax1, bx1) - d4_max(ax0, bx0)), ih = d4_relu(d4_min(ay1, by1) - d4_max(ay0, by0));
        const Dual4 inter = iw * ih, uni = area_a + area_b - inter;
        const Dual4 iou = inter / (uni + d4_const(1e-6f));
        const Dual4 hw = d4_relu(d4_max(ax1, bx1) - d4_min(ax0, bx0)), hh = d4_relu(d4_max(ay1, by1) - d4_min(ay0, by0));
        const Dual4 hull = hw * hh;
        const Dual4 giou = iou - (hull - uni) / (hull + d4_const(1e-6f));
        const Dual4 l = d4_scale(l1, c_l1) + d4_scale(d4_const(1.f) - giou, c_giou);
        const float wk = w[k];
        acc += l.v * wk;
        reinterpret_cast<float4 *>(grad_p)[k] = make_float4(l.d[0] * wk, l.d[1] * wk, l.d[2] * wk, l.d[3] * wk);
    }
    const float s = block_sum_1024(acc, red);
    if (threadIdx.x == 0) loss[0] = s;
}

// What the positive entries of a sigmoid focal loss contribute INSTEAD of the all-negative term FocalNegativeSum has counted for them
// (sigmoid_focal_loss, models/richsem/richsem.py:1124-1160 with util: alpha (1 - q)^2 softplus(-x) for a positive, (1 - alpha) q^2
// softplus(x) for a negative, q = sigmoid(x)):   sum_k w[k] * ( alpha (1 - q_k)^2 softplus(-x_k) - (1 - alpha) q_k^2 softplus(x_k) )
__global__ __launch_bounds__(1024) void focal_pos_sum_kernel(const float *__restrict__ x, const float *__restrict__ w, int K, float alpha,
                                                             float *__restrict__ loss, float *__restrict__ grad_x)
{
    __shared__ float red[16];
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float v = x[k], q = 1.f / (1.f + expf(-v));
        const float sp_pos = v > 20.f ? v : log1pf(expf(v)), sp_neg = -v > 20.f ? -v : log1pf(expf(-v));      // softplus(x), softplus(-x) (torch's threshold 20)
        const float omq = 1.f - q;
        const float l = alpha * omq * omq * sp_neg - (1.f - alpha) * q * q * sp_pos;
        // d/dx: q' = q (1 - q); softplus(x)' = q; softplus(-x)' = -(1 - q)
        const float dl = alpha * (-2.f * omq * q * omq * sp_neg - omq * omq * omq) - (1.f - alpha) * (2.f * q * q * omq * sp_pos + q * q * q);
        const float wk = w[k];
        acc += l * wk;
        grad_x[k] = dl * wk;
    }
    const float s = block_sum_1024(acc, red);
    if (threadIdx.x == 0) loss[0] = s;
}

extern "C" {

/* msda_focal_neg_sum_f32: partial (grid doubles, grid = the value returned through n_partial) <- weighted all-negative focal sums; the caller adds
 * them up.  msda_focal_neg_grad_f32: grad_logits <- gscale[0] * d/dx of that sum (every element written). */
int msda_focal_neg_sum_f32(const float *logits, const float *row_weight, int64_t rows, int C, float alpha, double *partial, int max_partial,
                           int *n_partial, msda_stream_t stream)
{
    if (!logits || !row_weight || !partial || !n_partial) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (rows < 1 || C < 1 || max_partial < 1) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    const int grid = (int)std::min<int64_t>(std::min<int64_t>(rows, 4096), max_partial);
    *n_partial = grid;
    hipLaunchKernelGGL(focal_neg_sum_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), logits, row_weight, (long long)rows, C,
                       1.f - alpha, partial);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}
int msda_focal_neg_grad_f32(const float *logits, const float *row_weight, int64_t rows, int C, float alpha, const float *gscale,
                            float *grad_logits, msda_stream_t stream)
{
    if (!logits || !row_weight || !gscale || !grad_logits) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (rows < 1 || C < 1) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    const int grid = (int)std::min<int64_t>(rows, 8192);
    hipLaunchKernelGGL(focal_neg_grad_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), logits, row_weight, (long long)rows, C,
                       1.f - alpha, gscale, grad_logits);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

/* The criterion's per-pair tails, one launch each (K pairs, float32, one workgroup): loss[0] <- the weighted sum, grad (K, 4) / (K) <- its
 * gradient w.r.t. the predictions (the caller multiplies by the incoming scalar gradient).
 * msda_box_pair_loss_f32: sum_k w[k] (c_l1 |p_k - t_k|_1 + c_giou (1 - GIoU(p_k, t_k))), boxes (cx, cy, w, h) (SetCriterion.loss_boxes);
 * msda_focal_pos_sum_f32: sum_k w[k] (alpha (1 - q)^2 softplus(-x_k) - (1 - alpha) q^2 softplus(x_k)), q = sigmoid(x_k): what a positive entry
 * contributes to the sigmoid focal loss instead of the all-negative term msda_focal_neg_sum_f32 counted for it. */
int msda_box_pair_loss_f32(const float *pred, const float *target, const float *weight, int K, float c_l1, float c_giou, float *loss, float *grad_pred,
                           msda_stream_t stream)
{
    if (!pred || !target || !weight || !loss || !grad_pred) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (K < 1) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(grad_pred)) & 15)
        return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    hipLaunchKernelGGL(box_pair_loss_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), pred, target, weight, K, c_l1, c_giou, loss, grad_pred);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}
int msda_focal_pos_sum_f32(const float *x, const float *weight, int K, float alpha, float *loss, float *grad_x, msda_stream_t stream)
{
    if (!x || !weight || !loss || !grad_x) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (K < 1) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    hipLaunchKernelGGL(focal_pos_sum_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), x, weight, K, alpha, loss, grad_x);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

/* Backward of y = x W^T + b for a 256 -> n layer, n <= 8: dy (T, n) bf16 contiguous, x (T, 256) bf16, w (n, 256) f32 -> dx (T, 256) bf16
 * (or NULL), dw (n, 256) f32 and db (n) f32 (or NULL), both overwritten */
int msda_narrow_linear_backward_bf16(const uint16_t *dy, const uint16_t *x, const float *w, int T, int n, uint16_t *dx, float *dw, float *db,
                                     msda_stream_t stream)
{
    if (!dy || !x || !w || !dw) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (T < 1 || n < 1 || n > 8) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(x)) & 15) return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * n * 256, st);
    if (e == hipSuccess && db) e = hipMemsetAsync(db, 0, sizeof(float) * n, st);
    if (e != hipSuccess) return (int)e;
    if (dx) {
        const long long total = (long long)T * 32;
        hipLaunchKernelGGL(narrow_linear_dx_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, st, dy,
                           w, T, n, dx);
    }
    const int chunk = T >= 65536 ? 256 : (T >= 8192 ? 128 : 32);
    hipLaunchKernelGGL(narrow_linear_dw_kernel, dim3((unsigned)((T + chunk - 1) / chunk)), dim3(256), 0, st, dy, x, T, n, chunk, dw, db);
    e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

/* y = sigmoid(delta + inverse_sigmoid(ref)): delta (n) bf16 or f32, ref (n) f32, y (n) f32 */
int msda_box_refine_forward(const void *delta, int delta_is_bf16, const float *ref, float eps, int64_t n, float *y, msda_stream_t stream)
{
    if (!delta || !ref || !y) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (n < 1) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (delta_is_bf16)
        hipLaunchKernelGGL(box_refine_kernel<true>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), delta, ref, eps, (long long)n, y);
    else
        hipLaunchKernelGGL(box_refine_kernel<false>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), delta, ref, eps, (long long)n, y);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

/* grad_delta = grad_y * y * (1 - y), written in delta's type */
int msda_box_refine_backward(const float *grad_y, const float *y, int64_t n, void *grad_delta, int delta_is_bf16, msda_stream_t stream)
{
    return msda_box_refine_backward_ref(grad_y, y, n, grad_delta, delta_is_bf16, nullptr, 0.f, nullptr, stream);
}

/* ... and grad_ref (n) f32 = grad_delta * d inverse_sigmoid(ref) / d ref (the clamps' gradients as torch takes them); ref, grad_ref may be NULL */
int msda_box_refine_backward_ref(const float *grad_y, const float *y, int64_t n, void *grad_delta, int delta_is_bf16, const float *ref, float eps,
                                 float *grad_ref, msda_stream_t stream)
{
    if (!grad_y || !y || !grad_delta || (grad_ref && !ref)) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (n < 1) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (delta_is_bf16)
        hipLaunchKernelGGL(box_refine_grad_kernel<true>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), grad_y, y, (long long)n, grad_delta,
                           ref, eps, grad_ref);
    else
        hipLaunchKernelGGL(box_refine_grad_kernel<false>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), grad_y, y, (long long)n, grad_delta,
                           ref, eps, grad_ref);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

/* the decoder's positional query embedding: boxes (tokens, >= dims) f32 with row stride ld (floats), dims = 2 | 4 -> out (tokens,
 * dims * pe_dim) bf16 */
int msda_sine_embed_bf16(const float *boxes, int ld, int tokens, int dims, int pe_dim, float temperature, uint16_t *out, msda_stream_t stream)
{
    if (!boxes || !out) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (tokens < 1 || (dims != 2 && dims != 4) || ld < dims || pe_dim < 2 || (pe_dim & 1) || !(temperature > 0.f)) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if (reinterpret_cast<uintptr_t>(out) & 3) return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    const long long n = (long long)tokens * dims * (pe_dim / 2);
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(sine_embed_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), boxes, ld, tokens, dims, pe_dim,
                       log2f(temperature), reinterpret_cast<unsigned *>(out));
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

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
