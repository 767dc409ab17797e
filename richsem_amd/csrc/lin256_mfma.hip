// lin256_mfma.hip -- out = act(x W^T + b) for a 256-wide input on the gfx950 matrix cores (bf16 storage, fp32 accumulation), with the
// two epilogues the feed-forward block's backward needs (reference forward_ffn,
// models/richsem/deformable_transformer.py:862-866, :940-944):
//     relu:  h  = relu(x W1^T + b1)                    the hidden activation, recomputed in the backward
//     mask:  gh = (dz W2) * (h > 0)                     gradient at the ReLU's input (W = W2^T, no bias)
// (and the plain / biased form).  The library's GEMM runs these K = 256 shapes at ~220 TFLOP/s (210 us for 44646 x 256 x 2048).
//
// Structure of csrc/cls_mfma.hip / ffn_mfma.hip: transposed product, tokens on the lanes (mfma_f32_16x16x32_bf16); a wave owns 48 tokens
// and keeps their x fragments (96 VGPRs) for the whole kernel; W streams once per workgroup through LDS in blocks of 64 output
// channels (32 KB: 4 row tiles x 8 k-steps), double-buffered through registers, one barrier per block (96 MFMAs per wave).  Output
// channels are permuted inside a block (row 4 q + i of tile u = channel 16 q + 4 u + i, msda_lin256_pack) so that a lane holds 16
// consecutive channels of its token: 32-byte stores, whole 128-byte lines per token.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "../../include/richsem_msda.h"

extern "C" int msda_note_error(int code, const char *entry);      // msda_api.hip: sets msda_last_error()

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int kD = 256;
constexpr int kTokWave = 48, kWaves = 4, kTokWg = kTokWave * kWaves;
constexpr int kFragShorts = 512;
constexpr int kBlockRows = 64;                                  // output channels per LDS block
constexpr int kBlockShorts = 4 * 8 * kFragShorts;              // 32 KB

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    const bf16x2_t p = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

// W (N x 256) bf16 row-major -> packed[block][k-step][tile u][lane][8]; lane (r = 4 q' + i, q) of tile u = output channel
// 64 block + 16 q' + 4 u + i, input channels 32 step + 8 q + 0..7
__global__ void lin256_pack_kernel(const uint16_t *__restrict__ w, uint16_t *__restrict__ packed, int N)
{
    const long long n = (long long)N * kD;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), u = (int)((i >> 9) & 3), s = (int)((i >> 11) & 7), blk = (int)(i >> 14);
        const int r = lane & 15, q = lane >> 4;
        const int ch = kBlockRows * blk + 16 * (r >> 2) + 4 * u + (r & 3), k = 32 * s + 8 * q + j;
        packed[i] = w[(long long)ch * kD + k];
    }
}

// EPI 0: out = acc + bias;  1: out = relu(acc + bias);  2: out = acc * (mask > 0);  3: out = acc + bias, rows with a non-zero byte in
// the row mask (`mask` read as one uint8 per token: the padding mask of MSDeformAttn's value projection,
// ops/modules/ms_deform_attn.py:94-96) zeroed
template <int EPI>
__global__ __launch_bounds__(kWaves * 64, 2)      // two workgroups per CU: one's stores / conversions under the other's MFMAs
void lin256_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ packed, const float *__restrict__ bias,
                   const uint16_t *__restrict__ mask, int T, int N, uint16_t *__restrict__ out, long long chunk_elems)
{
    __shared__ __attribute__((aligned(16))) short wbuf[2][kBlockShorts];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int tok0 = blockIdx.x * kTokWg + wave * kTokWave;
    // the output channels are split over gridDim.y workgroups (so that every CU holds two)
    const int nb_all = N / kBlockRows;
    const int b_lo = (int)((long long)nb_all * blockIdx.y / gridDim.y), b_hi = (int)((long long)nb_all * (blockIdx.y + 1) / gridDim.y);

    bf16x8 xf[3][8];
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
        const int tok = min(tok0 + 16 * t3 + c, T - 1);
        const uint16_t *row = x + (size_t)tok * kD + 8 * q;
#pragma unroll
        for (int s = 0; s < 8; ++s) xf[t3][s] = *reinterpret_cast<const bf16x8 *>(row + 32 * s);
    }

    constexpr int kChunks = kBlockShorts * 2 / 16 / (kWaves * 64);      // 8 x 16 bytes per thread and block
    u32x4 stage[kChunks];
    auto fetch = [&](int b) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(packed + (size_t)b * kBlockShorts);
#pragma unroll
        for (int i = 0; i < kChunks; ++i) stage[i] = src[tid + i * (kWaves * 64)];
    };
    auto park = [&](int slot) {
        u32x4 *dst = reinterpret_cast<u32x4 *>(wbuf[slot]);
#pragma unroll
        for (int i = 0; i < kChunks; ++i) dst[tid + i * (kWaves * 64)] = stage[i];
    };
    if (b_lo < b_hi) {
        fetch(b_lo);
        park(b_lo & 1);
    }
    __syncthreads();

    for (int b = b_lo; b < b_hi; ++b) {
        if (b + 1 < b_hi) fetch(b + 1);
        // the mask rows of this block, requested before the products that hide their latency
        u32x4 mreg[EPI == 2 ? 3 : 1][2];
        if (EPI == 2) {
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
                const int tok = min(tok0 + 16 * t3 + c, T - 1);
                const u32x4 *mp = reinterpret_cast<const u32x4 *>(mask + (size_t)tok * N + kBlockRows * b + 16 * q);
                mreg[EPI == 2 ? t3 : 0][0] = mp[0];
                mreg[EPI == 2 ? t3 : 0][1] = mp[1];
            }
        }
        const short *wt = wbuf[b & 1];
        f32x4 acc[3][4];
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[t3][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(wt + (s * 4 + u) * kFragShorts + lane * 8);
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) acc[t3][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xf[t3][s], acc[t3][u], 0, 0, 0);
            }
        // lane (c, q): channels 64 b + 16 q + (4 u + i) of token c
        const int ch0 = kBlockRows * b + 16 * q;
        float bb[16];
        if (EPI != 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 v = bias ? *reinterpret_cast<const f32x4 *>(bias + ch0 + 4 * u) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) bb[4 * u + i] = v[i];
            }
        }
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            const int tok = tok0 + 16 * t3 + c;
            if (tok >= T) continue;
            float y[16];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) y[4 * u + i] = acc[t3][u][i];
            if (EPI == 2) {
                const u32x4 m0 = mreg[EPI == 2 ? t3 : 0][0], m1 = mreg[EPI == 2 ? t3 : 0][1];
                const unsigned mw[8] = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (!(bf16_lo(mw[e]) > 0.f)) y[2 * e] = 0.f;
                    if (!(bf16_hi(mw[e]) > 0.f)) y[2 * e + 1] = 0.f;
                }
            } else {
                const bool dead = EPI == 3 && reinterpret_cast<const uint8_t *>(mask)[tok] != 0;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    y[e] += bb[e];
                    if (EPI == 1) y[e] = fmaxf(y[e], 0.f);
                    if (EPI == 3 && dead) y[e] = 0.f;
                }
            }
            // chunk_elems != 0: the output is N / 256 separate (T, 256) matrices, chunk_elems apart (several 256-wide layers stacked
            // into one product, each with a contiguous result of its own)
            u32x4 *op = reinterpret_cast<u32x4 *>(chunk_elems ? out + (size_t)(ch0 >> 8) * chunk_elems + (size_t)tok * 256 + (ch0 & 255)
                                                              : out + (size_t)tok * N + ch0);
            op[0] = (u32x4){pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]), pack_bf16(y[4], y[5]), pack_bf16(y[6], y[7])};
            op[1] = (u32x4){pack_bf16(y[8], y[9]), pack_bf16(y[10], y[11]), pack_bf16(y[12], y[13]), pack_bf16(y[14], y[15])};
        }
        if (b + 1 < b_hi) park((b + 1) & 1);
        __syncthreads();
    }
}

// ---- fp32 in / fp32 out at fp32-level accuracy: both operands split into bf16 hi + lo parts, hi.hi + lo_w.hi_x + hi_w.lo_x (the scheme
// of csrc/cls_mfma.hip: three bf16 MFMAs instead of the fp32 MFMA's sixteen-fold cost) -- the MSDeformAttn module's fp32 projections
// (reference ops/modules/ms_deform_attn.py:52-56, :94-100), which the library's fp32 GEMM runs at 80-100 TFLOP/s.
constexpr int kF32BlockRows = 32;                                  // output channels per LDS block (2 row tiles, hi + lo parts)
constexpr int kF32BlockShorts = 2 * 8 * 2 * kFragShorts;           // [part][k-step][tile][fragment]: 32 KB

// W (N x 256) fp32 -> packed[block][part][k-step][tile u][lane][8] bf16; row 4 q' + i of tile u = channel 32 block + 8 q' + 4 u + i
__global__ void lin256_pack_f32_kernel(const float *__restrict__ w, uint16_t *__restrict__ packed, int N)
{
    const long long n = (long long)N * kD;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), u = (int)((i >> 9) & 1), s = (int)((i >> 10) & 7), blk = (int)(i >> 13);
        const int r = lane & 15, q = lane >> 4;
        const int ch = kF32BlockRows * blk + 8 * (r >> 2) + 4 * u + (r & 3), k = 32 * s + 8 * q + j;
        const float v = w[(long long)ch * kD + k];
        const unsigned hi = pack_bf16(v, 0.f) & 0xFFFFu;
        const unsigned lo = pack_bf16(v - __uint_as_float(hi << 16), 0.f) & 0xFFFFu;
        const long long base = (long long)blk * kF32BlockShorts + ((long long)(s * 2 + u) * 64 + lane) * 8 + j;
        packed[base] = (uint16_t)hi;
        packed[base + 8 * 2 * kFragShorts] = (uint16_t)lo;
    }
}

__global__ __launch_bounds__(kWaves * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void lin256_f32_kernel(const float *__restrict__ x, const uint16_t *__restrict__ packed, const float *__restrict__ bias, int T, int N,
                       float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) short wbuf[2][kF32BlockShorts];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int tok0 = blockIdx.x * kTokWg + wave * kTokWave;
    const int nb_all = N / kF32BlockRows;
    const int b_lo = (int)((long long)nb_all * blockIdx.y / gridDim.y), b_hi = (int)((long long)nb_all * (blockIdx.y + 1) / gridDim.y);

    bf16x8 xh[3][8], xl[3][8];
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
        const int tok = min(tok0 + 16 * t3 + c, T - 1);
        const float *row = x + (size_t)tok * kD + 8 * q;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float4 a = *reinterpret_cast<const float4 *>(row + 32 * s), b = *reinterpret_cast<const float4 *>(row + 32 * s + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            u32x4 h, l;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                h[p] = pack_bf16(v[2 * p], v[2 * p + 1]);
                l[p] = pack_bf16(v[2 * p] - bf16_lo(h[p]), v[2 * p + 1] - bf16_hi(h[p]));
            }
            xh[t3][s] = __builtin_bit_cast(bf16x8, h);
            xl[t3][s] = __builtin_bit_cast(bf16x8, l);
        }
    }

    constexpr int kChunks = kF32BlockShorts * 2 / 16 / (kWaves * 64);      // 8
    u32x4 stage[kChunks];
    auto fetch = [&](int b) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(packed + (size_t)b * kF32BlockShorts);
#pragma unroll
        for (int i = 0; i < kChunks; ++i) stage[i] = src[tid + i * (kWaves * 64)];
    };
    auto park = [&](int slot) {
        u32x4 *dst = reinterpret_cast<u32x4 *>(wbuf[slot]);
#pragma unroll
        for (int i = 0; i < kChunks; ++i) dst[tid + i * (kWaves * 64)] = stage[i];
    };
    if (b_lo < b_hi) {
        fetch(b_lo);
        park(b_lo & 1);
    }
    __syncthreads();

    for (int b = b_lo; b < b_hi; ++b) {
        if (b + 1 < b_hi) fetch(b + 1);
        const short *wt = wbuf[b & 1];
        f32x4 acc[3][2];
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t3][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(wt + (s * 2 + u) * kFragShorts + lane * 8);
                const bf16x8 al = *reinterpret_cast<const bf16x8 *>(wt + (16 + s * 2 + u) * kFragShorts + lane * 8);
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) acc[t3][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh[t3][s], acc[t3][u], 0, 0, 0);
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) acc[t3][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl[t3][s], acc[t3][u], 0, 0, 0);
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) acc[t3][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh[t3][s], acc[t3][u], 0, 0, 0);
            }
        // lane (c, q): channels 32 b + 8 q + (4 u + i) of token c: two 16-byte stores
        const int ch0 = kF32BlockRows * b + 8 * q;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
            b0 = *reinterpret_cast<const f32x4 *>(bias + ch0);
            b1 = *reinterpret_cast<const f32x4 *>(bias + ch0 + 4);
        }
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            const int tok = tok0 + 16 * t3 + c;
            if (tok >= T) continue;
            f32x4 *op = reinterpret_cast<f32x4 *>(out + (size_t)tok * N + ch0);
            op[0] = acc[t3][0] + b0;
            op[1] = acc[t3][1] + b1;
        }
        if (b + 1 < b_hi) park((b + 1) & 1);
        __syncthreads();
    }
}

template <int EPI>
int launch_lin(const uint16_t *x, const uint16_t *packed, const float *bias, const uint16_t *mask, int T, int N, uint16_t *out,
               hipStream_t st, long long chunk_elems = 0)
{
    const int gx = (T + kTokWg - 1) / kTokWg, nb = N / kBlockRows;
    // the output-channel blocks are split over gy workgroup rows so that the grid is ONE resident wave of workgroups (two per CU on 256
    // CUs): rounding up instead (3 rows for the encoder's 233 token blocks) made a second, mostly empty wave, dealt 4 blocks of a 256-wide
    // output as 1 + 1 + 2 and read x a third time -- 22.3 -> 18.5 us at 44646 x 256 -> 256, 63 -> 54.5 us at -> 1536, 81 -> 75 us at -> 2048
    int gy = 512 / gx;
    if (gy < 1) gy = 1;
    if (gy > nb) gy = nb;
    if (gy > 8) gy = 8;
    hipLaunchKernelGGL(lin256_kernel<EPI>, dim3(gx, gy), dim3(kWaves * 64), 0, st, x, packed, bias, mask, T, N, out, chunk_elems);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

}  // namespace

extern "C" {

int msda_lin256_pack_bf16(const uint16_t *w, int out_features, int in_features, uint16_t *packed, msda_stream_t stream)
{
    if (!w || !packed) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (in_features != kD || out_features < kBlockRows || out_features % kBlockRows != 0) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    hipLaunchKernelGGL(lin256_pack_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream), w, packed, out_features);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

int msda_lin256_forward_bf16(const uint16_t *x, const uint16_t *packed_w, const float *bias, const uint16_t *relu_mask, int epilogue,
                             int tokens, int in_features, int out_features, uint16_t *out, msda_stream_t stream)
{
    if (!x || !packed_w || !out) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (tokens < 0 || in_features != kD || out_features < kBlockRows || out_features % kBlockRows != 0 || epilogue < 0 || epilogue > 3)
        return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if (epilogue >= 2 && !relu_mask) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed_w) | reinterpret_cast<uintptr_t>(out) |
         reinterpret_cast<uintptr_t>(bias) | (epilogue == 2 ? reinterpret_cast<uintptr_t>(relu_mask) : 0)) & 15)
        return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    if (tokens == 0) return MSDA_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (epilogue) {
    case 0: return launch_lin<0>(x, packed_w, bias, nullptr, tokens, out_features, out, st);
    case 1: return launch_lin<1>(x, packed_w, bias, nullptr, tokens, out_features, out, st);
    case 2: return launch_lin<2>(x, packed_w, nullptr, relu_mask, tokens, out_features, out, st);
    default: return launch_lin<3>(x, packed_w, bias, relu_mask, tokens, out_features, out, st);
    }
}

int msda_lin256_forward_stacked_bf16(const uint16_t *x, const uint16_t *packed_w, const float *bias, const uint8_t *row_mask, int tokens,
                                     int in_features, int out_features, uint16_t *out, msda_stream_t stream)
{
    if (!x || !packed_w || !out) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (tokens < 0 || in_features != kD || out_features < 256 || out_features % 256 != 0) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed_w) | reinterpret_cast<uintptr_t>(out) |
         reinterpret_cast<uintptr_t>(bias)) & 15)
        return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    if (tokens == 0) return MSDA_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long chunk = (long long)tokens * 256;
    if (row_mask) return launch_lin<3>(x, packed_w, bias, reinterpret_cast<const uint16_t *>(row_mask), tokens, out_features, out, st, chunk);
    return launch_lin<0>(x, packed_w, bias, nullptr, tokens, out_features, out, st, chunk);
}

int msda_lin256_pack_f32(const float *w, int out_features, int in_features, uint16_t *packed, msda_stream_t stream)
{
    if (!w || !packed) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (in_features != kD || out_features < kF32BlockRows || out_features % kF32BlockRows != 0) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    hipLaunchKernelGGL(lin256_pack_f32_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream), w, packed, out_features);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

int msda_lin256_forward_f32(const float *x, const uint16_t *packed_w, const float *bias, int tokens, int in_features, int out_features,
                            float *out, msda_stream_t stream)
{
    if (!x || !packed_w || !out) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (tokens < 0 || in_features != kD || out_features < kF32BlockRows || out_features % kF32BlockRows != 0) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed_w) | reinterpret_cast<uintptr_t>(out) |
         reinterpret_cast<uintptr_t>(bias)) & 15)
        return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    if (tokens == 0) return MSDA_OK;
    const int gx = (tokens + kTokWg - 1) / kTokWg, nb = out_features / kF32BlockRows;
    int gy = 1;          // (one workgroup per CU: 1 wave per SIMD at its register count; 233 workgroups at the training shape)
    if (gy > nb) gy = nb;
    if (gy > 4) gy = 4;
    hipLaunchKernelGGL(lin256_f32_kernel, dim3(gx, gy), dim3(kWaves * 64), 0, static_cast<hipStream_t>(stream), x, packed_w, bias, tokens,
                       out_features, out);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

}  // extern "C"
