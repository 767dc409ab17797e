// attn_mfma.hip -- masked multi-head self-attention of the decoder's queries on the gfx950 matrix cores, forward and backward
// (nn.MultiheadAttention of DeformableTransformerDecoderLayer, reference
// models/richsem/deformable_transformer.py:907, :974-978; head dimension 32, ~1.1 k queries per image with the boolean
// denoising mask of dn_components.py:155-176).  bf16 storage, fp32 softmax and accumulation.  New capability next to the
// reference, which runs torch's fp32 multi_head_attention_forward.
//
//   softmax(q k^T / sqrt(32) + mask) v   per (image, head), never materialising the (nq x nq) score matrix:
//
// forward    a workgroup owns 16 queries; its four waves each walk a quarter of the keys in blocks of 32 with an online softmax and
//            merge their results through LDS.  The products are taken
//            TRANSPOSED (mfma_f32_16x16x32_bf16): S^T = K . Q^T puts the query on the lane and the keys in the registers, so the
//            row maximum is a register maximum plus two lane exchanges, and the probabilities -- an accumulator tile -- are, converted
//            to bf16, directly the B operand of O^T += V^T . P^T (the k order of that operand is permuted; V^T is read from a
//            transposed copy in the matching order).  Saves log2-sum-exp per query.
// backward   recomputes the probabilities from the saved log-sum-exp in two kernels, each with the tensor it owns on the lanes:
//            keys on the lanes for dK^T += Q^T . dS and dV^T += dO^T . P, queries on the lanes for dQ^T += K^T . dS^T --
//            no atomics, every output element written once.  delta = rowsum(dO * O) comes from a small kernel of its own.
// The boolean mask travels as bits: word (q, kb) holds the 32 keys of block kb (and, for the key-owning kernel, word (k, qb)
// the 32 queries of block qb), built once per mask by the caller.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "../../include/richsem_msda.h"

extern "C" int msda_note_error(int code, const char *entry);      // msda_api.hip: sets msda_last_error()

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int kHd = 32;            // head dimension
constexpr int kQW = 16;            // queries (or keys) per wave
constexpr int kWaves = 4;

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    const bf16x2_t p = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ bf16x8 pack8(const float (&p)[8])
{
    const unsigned u0 = pack_bf16(p[0], p[1]), u1 = pack_bf16(p[2], p[3]), u2 = pack_bf16(p[4], p[5]), u3 = pack_bf16(p[6], p[7]);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(bf16x8, (u32x4){u0, u1, u2, u3});
}
__device__ __forceinline__ float xor16(float v) { return __shfl_xor(v, 16, 64); }
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

// fragment of a transposed copy X^T (32 x nqp) for the product that sums over an accumulator tile's rows: lane (row d, group g),
// element j <-> column 32 blk + 16 (j >> 2) + 4 g + (j & 3)
__device__ __forceinline__ bf16x8 frag_t(const uint16_t *xt, int nqp, int d, int blk, int g)
{
    const uint16_t *p = xt + (size_t)d * nqp + 32 * blk + 4 * g;
    const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(p), hi = *reinterpret_cast<const bf16x4 *>(p + 16);
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// X (token (i, b) at row i * si + b * sb, `ld` elements apart; head h at column 32 h) -> X^T[(b, h)][d][i], i < nqp zero-padded
__global__ __launch_bounds__(256) void attn_transpose_kernel(const uint16_t *__restrict__ x, int ld, int nq, int nqp, int si, int sb, int H,
                                                             uint16_t *__restrict__ xt)
{
    __shared__ uint16_t tile[kHd][64 + 2];
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int i0 = blockIdx.x * 64, t = threadIdx.x;
    {
        const int i = i0 + (t >> 2), part = t & 3;
        bf16x8 v = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (i < nq) v = *reinterpret_cast<const bf16x8 *>(x + ((size_t)i * si + (size_t)b * sb) * ld + h * kHd + 8 * part);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[8 * part + j][t >> 2] = (uint16_t)v[j];
    }
    __syncthreads();
    const int d = t >> 3, c = (t & 7) * 8;
    if (i0 + c < nqp) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (short)tile[d][c + j];
        *reinterpret_cast<bf16x8 *>(xt + ((size_t)bh * kHd + d) * nqp + i0 + c) = o;
    }
}

// delta[(b, h)][q] = sum_d dO * O (zero beyond nq)
__global__ __launch_bounds__(256) void attn_delta_kernel(const uint16_t *__restrict__ o, const uint16_t *__restrict__ dout, int nq, int nqp,
                                                         int bs, int si, int sb, int H, float *__restrict__ delta)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= bs * H * nqp) return;
    const int bh = idx / nqp, i = idx - bh * nqp, b = bh / H, h = bh - b * H;
    float s = 0.f;
    if (i < nq) {
        const size_t off = ((size_t)i * si + (size_t)b * sb) * (H * kHd) + h * kHd;
#pragma unroll
        for (int part = 0; part < 4; ++part) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(o + off + 8 * part), g = *reinterpret_cast<const bf16x8 *>(dout + off + 8 * part);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                s += __uint_as_float((unsigned)(uint16_t)a[j] << 16) * __uint_as_float((unsigned)(uint16_t)g[j] << 16);
        }
    }
    delta[idx] = s;
}

struct Tok {       // token (i, b) sits at row i * si + b * sb (sequence-first: si = bs, sb = 1; batch-first: si = 1, sb = nq)
    int si, sb;
    __device__ __forceinline__ size_t row(int i, int b) const { return (size_t)i * si + (size_t)b * sb; }
};
__device__ __forceinline__ bf16x8 load_row(const uint16_t *x, int ld, int i, int nq, Tok tk, int b, int h, int g)
{
    return *reinterpret_cast<const bf16x8 *>(x + tk.row(min(i, nq - 1), b) * ld + h * kHd + 8 * g);
}

// ---- forward ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWaves * 64) void attn_fwd_kernel(const uint16_t *__restrict__ q, int ldq, const uint16_t *__restrict__ k, int ldk,
                                                               const uint16_t *__restrict__ vt, const uint32_t *__restrict__ mask_bits,
                                                               int nq, int nkb, Tok tk, int H, float scale2, uint16_t *__restrict__ out,
                                                               float *__restrict__ lse)
{
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int nqp = nkb * 32;
    const int q0 = blockIdx.x * kQW;      // the workgroup's 16 queries; its four waves share out the key blocks
    const int qi = q0 + c;
    if (q0 >= nq) {   // (whole workgroup) only the padding of the log-sum-exp: +inf makes the backward's probabilities zero
        if (wave == 0 && g == 0 && qi < nqp) lse[(size_t)bh * nqp + qi] = __builtin_inff();
        return;
    }
    __shared__ float part[kWaves][10][64];      // per wave and lane: two output tiles, running maximum, running sum
    const bf16x8 qf = load_row(q, ldq, qi, nq, tk, b, h, g);
    const uint16_t *vt_bh = vt + (size_t)bh * kHd * nqp;
    const uint32_t *mrow = mask_bits ? mask_bits + (size_t)min(qi, nq - 1) * nkb : nullptr;
    float m = -__builtin_inff(), l = 0.f;
    f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kb = wave; kb < nkb; kb += kWaves) {
        const bf16x8 k0 = load_row(k, ldk, 32 * kb + c, nq, tk, b, h, g), k1 = load_row(k, ldk, 32 * kb + 16 + c, nq, tk, b, h, g);
        const uint32_t bits = mrow ? mrow[kb] : 0u;
        const bf16x8 v0 = frag_t(vt_bh, nqp, c, kb, g), v1 = frag_t(vt_bh, nqp, 16 + c, kb, g);
        const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf, zero, 0, 0, 0);
        const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf, zero, 0, 0, 0);
        float s[8];
        float mx = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 16 * (j >> 2) + 4 * g + (j & 3);      // key inside the block
            const bool dead = 32 * kb + kk >= nq || ((bits >> kk) & 1u);
            s[j] = dead ? -__builtin_inff() : (j < 4 ? s0[j & 3] : s1[j & 3]) * scale2;
            mx = fmaxf(mx, s[j]);
        }
        mx = fmaxf(mx, xor16(mx));
        mx = fmaxf(mx, xor32(mx));
        const float m_new = fmaxf(m, mx);
        const float m_use = m_new == -__builtin_inff() ? 0.f : m_new;
        const float alpha = exp2f(m - m_use);     // (m = -inf: 0)
        float p[8], ps = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            p[j] = exp2f(s[j] - m_use);
            ps += p[j];
        }
        l = l * alpha + ps;
        m = m_new;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o0[i] *= alpha;
            o1[i] *= alpha;
        }
        const bf16x8 pb = pack8(p);
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, pb, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, pb, o1, 0, 0, 0);
    }
    l += xor16(l);
    l += xor32(l);
    // the four waves' partial results (each over a quarter of the keys) are merged by wave 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        part[wave][i][lane] = o0[i];
        part[wave][4 + i][lane] = o1[i];
    }
    part[wave][8][lane] = m;
    part[wave][9][lane] = l;
    __syncthreads();
    if (wave != 0) return;
    float mm = m;
#pragma unroll
    for (int w = 1; w < kWaves; ++w) mm = fmaxf(mm, part[w][8][lane]);
    const float mu = mm == -__builtin_inff() ? 0.f : mm;
    float a0 = exp2f(m - mu);
    l *= a0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o0[i] *= a0;
        o1[i] *= a0;
    }
#pragma unroll
    for (int w = 1; w < kWaves; ++w) {
        const float aw = exp2f(part[w][8][lane] - mu);
        l += aw * part[w][9][lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o0[i] += aw * part[w][i][lane];
            o1[i] += aw * part[w][4 + i][lane];
        }
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if (qi < nq) {
        uint16_t *op = out + tk.row(qi, b) * (H * kHd) + h * kHd + 4 * g;
        *reinterpret_cast<uint2 *>(op) = make_uint2(pack_bf16(o0[0] * inv, o0[1] * inv), pack_bf16(o0[2] * inv, o0[3] * inv));
        *reinterpret_cast<uint2 *>(op + 16) = make_uint2(pack_bf16(o1[0] * inv, o1[1] * inv), pack_bf16(o1[2] * inv, o1[3] * inv));
    }
    if (g == 0 && qi < nqp) lse[(size_t)bh * nqp + qi] = qi < nq && l > 0.f ? mm + log2f(l) : __builtin_inff();
}

// ---- backward, keys on the lanes: dK, dV ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWaves * 64) void attn_bwd_kv_kernel(
    const uint16_t *__restrict__ q, int ldq, const uint16_t *__restrict__ k, int ldk, const uint16_t *__restrict__ v, int ldv,
    const uint16_t *__restrict__ dout, const uint16_t *__restrict__ qt, const uint16_t *__restrict__ dot, const float *__restrict__ lse,
    const float *__restrict__ delta, const uint32_t *__restrict__ maskt_bits, int nq, int nkb, Tok tk, int H, float scale, float scale2,
    uint16_t *__restrict__ dk, int lddk, uint16_t *__restrict__ dv, int lddv)
{
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int nqp = nkb * 32;
    const int k0i = blockIdx.x * kQW;     // the workgroup's 16 keys; its four waves share out the query blocks
    if (k0i >= nq) return;
    const int ki = k0i + c;
    __shared__ float part[kWaves - 1][16][64];
    const bf16x8 kf = load_row(k, ldk, ki, nq, tk, b, h, g), vf = load_row(v, ldv, ki, nq, tk, b, h, g);
    const uint16_t *qt_bh = qt + (size_t)bh * kHd * nqp, *dot_bh = dot + (size_t)bh * kHd * nqp;
    const float *lse_bh = lse + (size_t)bh * nqp, *delta_bh = delta + (size_t)bh * nqp;
    const uint32_t *mrow = maskt_bits ? maskt_bits + (size_t)min(ki, nq - 1) * nkb : nullptr;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 dv0 = zero, dv1 = zero, dk0 = zero, dk1 = zero;
    const int dout_ld = H * kHd;
    for (int qb = wave; qb < nkb; qb += kWaves) {
        const bf16x8 q0 = load_row(q, ldq, 32 * qb + c, nq, tk, b, h, g), q1 = load_row(q, ldq, 32 * qb + 16 + c, nq, tk, b, h, g);
        const bf16x8 g0 = load_row(dout, dout_ld, 32 * qb + c, nq, tk, b, h, g), g1 = load_row(dout, dout_ld, 32 * qb + 16 + c, nq, tk, b, h, g);
        const uint32_t bits = mrow ? mrow[qb] : 0u;
        const f32x4 l0 = *reinterpret_cast<const f32x4 *>(lse_bh + 32 * qb + 4 * g), l1 = *reinterpret_cast<const f32x4 *>(lse_bh + 32 * qb + 16 + 4 * g);
        const f32x4 e0 = *reinterpret_cast<const f32x4 *>(delta_bh + 32 * qb + 4 * g), e1 = *reinterpret_cast<const f32x4 *>(delta_bh + 32 * qb + 16 + 4 * g);
        const bf16x8 a_do0 = frag_t(dot_bh, nqp, c, qb, g), a_do1 = frag_t(dot_bh, nqp, 16 + c, qb, g);
        const bf16x8 a_q0 = frag_t(qt_bh, nqp, c, qb, g), a_q1 = frag_t(qt_bh, nqp, 16 + c, qb, g);
        const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0, kf, zero, 0, 0, 0);
        const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1, kf, zero, 0, 0, 0);
        const f32x4 p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g0, vf, zero, 0, 0, 0);
        const f32x4 p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g1, vf, zero, 0, 0, 0);
        float p[8], ds[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int qq = 16 * (j >> 2) + 4 * g + (j & 3);      // query inside the block
            const bool dead = ki >= nq || ((bits >> qq) & 1u);
            const float sv = (j < 4 ? s0[j & 3] : s1[j & 3]) * scale2, lv = j < 4 ? l0[j & 3] : l1[j & 3];
            const float dp = j < 4 ? p0[j & 3] : p1[j & 3], dl = j < 4 ? e0[j & 3] : e1[j & 3];
            p[j] = dead ? 0.f : exp2f(sv - lv);                   // (rows beyond nq: lse = +inf -> 0)
            ds[j] = p[j] * (dp - dl) * scale;
        }
        const bf16x8 pb = pack8(p), dsb = pack8(ds);
        dv0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_do0, pb, dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_do1, pb, dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_q0, dsb, dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_q1, dsb, dk1, 0, 0, 0);
    }
    if (wave != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            part[wave - 1][i][lane] = dk0[i];
            part[wave - 1][4 + i][lane] = dk1[i];
            part[wave - 1][8 + i][lane] = dv0[i];
            part[wave - 1][12 + i][lane] = dv1[i];
        }
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < kWaves - 1; ++w)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dk0[i] += part[w][i][lane];
            dk1[i] += part[w][4 + i][lane];
            dv0[i] += part[w][8 + i][lane];
            dv1[i] += part[w][12 + i][lane];
        }
    if (ki < nq) {
        uint16_t *pk = dk + tk.row(ki, b) * lddk + h * kHd + 4 * g, *pv = dv + tk.row(ki, b) * lddv + h * kHd + 4 * g;
        *reinterpret_cast<uint2 *>(pk) = make_uint2(pack_bf16(dk0[0], dk0[1]), pack_bf16(dk0[2], dk0[3]));
        *reinterpret_cast<uint2 *>(pk + 16) = make_uint2(pack_bf16(dk1[0], dk1[1]), pack_bf16(dk1[2], dk1[3]));
        *reinterpret_cast<uint2 *>(pv) = make_uint2(pack_bf16(dv0[0], dv0[1]), pack_bf16(dv0[2], dv0[3]));
        *reinterpret_cast<uint2 *>(pv + 16) = make_uint2(pack_bf16(dv1[0], dv1[1]), pack_bf16(dv1[2], dv1[3]));
    }
}

// ---- backward, queries on the lanes: dQ ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWaves * 64) void attn_bwd_q_kernel(
    const uint16_t *__restrict__ q, int ldq, const uint16_t *__restrict__ k, int ldk, const uint16_t *__restrict__ v, int ldv,
    const uint16_t *__restrict__ dout, const uint16_t *__restrict__ kt, const float *__restrict__ lse, const float *__restrict__ delta,
    const uint32_t *__restrict__ mask_bits, int nq, int nkb, Tok tk, int H, float scale, float scale2, uint16_t *__restrict__ dq, int lddq)
{
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int nqp = nkb * 32;
    const int q0 = blockIdx.x * kQW;      // the workgroup's 16 queries; its four waves share out the key blocks
    if (q0 >= nq) return;
    const int qi = q0 + c;
    __shared__ float part[kWaves - 1][8][64];
    const bf16x8 qf = load_row(q, ldq, qi, nq, tk, b, h, g), gf = load_row(dout, H * kHd, qi, nq, tk, b, h, g);
    const uint16_t *kt_bh = kt + (size_t)bh * kHd * nqp;
    const float lq = lse[(size_t)bh * nqp + min(qi, nqp - 1)], dl = delta[(size_t)bh * nqp + min(qi, nqp - 1)];
    const uint32_t *mrow = mask_bits ? mask_bits + (size_t)min(qi, nq - 1) * nkb : nullptr;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 dq0 = zero, dq1 = zero;
    for (int kb = wave; kb < nkb; kb += kWaves) {
        const bf16x8 k0 = load_row(k, ldk, 32 * kb + c, nq, tk, b, h, g), k1 = load_row(k, ldk, 32 * kb + 16 + c, nq, tk, b, h, g);
        const bf16x8 v0 = load_row(v, ldv, 32 * kb + c, nq, tk, b, h, g), v1 = load_row(v, ldv, 32 * kb + 16 + c, nq, tk, b, h, g);
        const uint32_t bits = mrow ? mrow[kb] : 0u;
        const bf16x8 a_k0 = frag_t(kt_bh, nqp, c, kb, g), a_k1 = frag_t(kt_bh, nqp, 16 + c, kb, g);
        const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf, zero, 0, 0, 0);
        const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf, zero, 0, 0, 0);
        const f32x4 p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, gf, zero, 0, 0, 0);
        const f32x4 p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, gf, zero, 0, 0, 0);
        float ds[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 16 * (j >> 2) + 4 * g + (j & 3);
            const bool dead = 32 * kb + kk >= nq || ((bits >> kk) & 1u);
            const float sv = (j < 4 ? s0[j & 3] : s1[j & 3]) * scale2, dp = j < 4 ? p0[j & 3] : p1[j & 3];
            const float pj = dead ? 0.f : exp2f(sv - lq);
            ds[j] = pj * (dp - dl) * scale;
        }
        const bf16x8 dsb = pack8(ds);
        dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_k0, dsb, dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_k1, dsb, dq1, 0, 0, 0);
    }
    if (wave != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            part[wave - 1][i][lane] = dq0[i];
            part[wave - 1][4 + i][lane] = dq1[i];
        }
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < kWaves - 1; ++w)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dq0[i] += part[w][i][lane];
            dq1[i] += part[w][4 + i][lane];
        }
    if (qi < nq) {
        uint16_t *pq = dq + tk.row(qi, b) * lddq + h * kHd + 4 * g;
        *reinterpret_cast<uint2 *>(pq) = make_uint2(pack_bf16(dq0[0], dq0[1]), pack_bf16(dq0[2], dq0[3]));
        *reinterpret_cast<uint2 *>(pq + 16) = make_uint2(pack_bf16(dq1[0], dq1[1]), pack_bf16(dq1[2], dq1[3]));
    }
}

bool bad_ld(int ld, int H) { return ld < H * kHd || (ld & 7); }
bool misaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }
int finish()
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

}  // namespace

extern "C" {

int64_t msda_attn_workspace_bytes(int nq, int bs, int heads)
{
    const int64_t nqp = ((int64_t)nq + 31) / 32 * 32;
    return 4 * (int64_t)bs * heads * kHd * nqp * 2 + (int64_t)bs * heads * nqp * 4;      // four transposed copies (bf16) + delta (f32)
}

int msda_attn_forward_bf16(const uint16_t *q, int ldq, const uint16_t *k, int ldk, const uint16_t *v, int ldv, const uint32_t *mask_bits,
                           int nq, int bs, int batch_first, int heads, uint16_t *out, float *lse, void *workspace, msda_stream_t stream)
{
    const Tok tk = batch_first ? Tok{1, nq} : Tok{bs, 1};
    if (!q || !k || !v || !out || !lse || !workspace) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (nq <= 0 || bs <= 0 || heads <= 0 || bad_ld(ldq, heads) || bad_ld(ldk, heads) || bad_ld(ldv, heads)) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if (misaligned(q) || misaligned(k) || misaligned(v) || misaligned(out) || misaligned(lse) || misaligned(workspace)) return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nkb = (nq + 31) / 32, nqp = nkb * 32;
    uint16_t *vt = static_cast<uint16_t *>(workspace);
    const dim3 tgrid((nqp + 63) / 64, bs * heads), grid(nqp / kQW, bs * heads);
    hipLaunchKernelGGL(attn_transpose_kernel, tgrid, dim3(256), 0, st, v, ldv, nq, nqp, tk.si, tk.sb, heads, vt);
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(kWaves * 64), 0, st, q, ldq, k, ldk, vt, mask_bits, nq, nkb, tk, heads,
                       1.4426950408889634f / sqrtf((float)kHd), out, lse);
    return finish();
}

int msda_attn_backward_bf16(const uint16_t *q, int ldq, const uint16_t *k, int ldk, const uint16_t *v, int ldv, const uint16_t *out,
                            const uint16_t *dout, const float *lse, const uint32_t *mask_bits, const uint32_t *maskt_bits, int nq, int bs,
                            int batch_first, int heads, uint16_t *dq, int lddq, uint16_t *dk, int lddk, uint16_t *dv, int lddv, void *workspace,
                            msda_stream_t stream)
{
    const Tok tk = batch_first ? Tok{1, nq} : Tok{bs, 1};
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !workspace) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if ((mask_bits == nullptr) != (maskt_bits == nullptr)) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (nq <= 0 || bs <= 0 || heads <= 0 || bad_ld(ldq, heads) || bad_ld(ldk, heads) || bad_ld(ldv, heads) || bad_ld(lddq, heads) ||
        bad_ld(lddk, heads) || bad_ld(lddv, heads))
        return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if (misaligned(q) || misaligned(k) || misaligned(v) || misaligned(out) || misaligned(dout) || misaligned(dq) || misaligned(dk) ||
        misaligned(dv) || misaligned(workspace) || misaligned(lse))
        return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nkb = (nq + 31) / 32, nqp = nkb * 32;
    const size_t tsz = (size_t)bs * heads * kHd * nqp;
    uint16_t *qt = static_cast<uint16_t *>(workspace), *kt = qt + tsz, *dot = kt + tsz;
    float *delta = reinterpret_cast<float *>(dot + 2 * tsz);      // (the workspace has four transposed-tensor slots; the forward uses slot 0 for V^T, the backward -- a call of its own, with its own
                                                                  // workspace -- slots 0..2 for Q^T, K^T, dO^T; the row sums follow the fourth)
    const dim3 tgrid((nqp + 63) / 64, bs * heads), grid(nqp / kQW, bs * heads);
    hipLaunchKernelGGL(attn_transpose_kernel, tgrid, dim3(256), 0, st, q, ldq, nq, nqp, tk.si, tk.sb, heads, qt);
    hipLaunchKernelGGL(attn_transpose_kernel, tgrid, dim3(256), 0, st, k, ldk, nq, nqp, tk.si, tk.sb, heads, kt);
    hipLaunchKernelGGL(attn_transpose_kernel, tgrid, dim3(256), 0, st, dout, heads * kHd, nq, nqp, tk.si, tk.sb, heads, dot);
    hipLaunchKernelGGL(attn_delta_kernel, dim3((bs * heads * nqp + 255) / 256), dim3(256), 0, st, out, dout, nq, nqp, bs, tk.si, tk.sb, heads, delta);
    const float scale = 1.f / sqrtf((float)kHd), scale2 = 1.4426950408889634f * scale;
    hipLaunchKernelGGL(attn_bwd_kv_kernel, grid, dim3(kWaves * 64), 0, st, q, ldq, k, ldk, v, ldv, dout, qt, dot, lse, delta, maskt_bits, nq,
                       nkb, tk, heads, scale, scale2, dk, lddk, dv, lddv);
    hipLaunchKernelGGL(attn_bwd_q_kernel, grid, dim3(kWaves * 64), 0, st, q, ldq, k, ldk, v, ldv, dout, kt, lse, delta, mask_bits, nq, nkb,
                       tk, heads, scale, scale2, dq, lddq);
    return finish();
}

}  // extern "C"
