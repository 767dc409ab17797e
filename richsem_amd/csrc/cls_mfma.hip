// cls_mfma.hip -- two-stage query selection score as ONE gfx950 MFMA kernel (second half: the class-logit
// product "with top-k over max-logit without materialising the logits").
//
// Reference: models/richsem/deformable_transformer.py:368-372
//     enc_outputs_class_unselected = self.enc_out_class_embed(output_memory)            # (bs, sum(HW), classes)
//     topk_proposals = torch.topk(enc_outputs_class_unselected.max(-1)[0], topk, dim=1)[1]
// with enc_out_class_embed = the CLIP-text classifier (models/richsem/richsem.py:176-184, shipped configuration: dino_visual_proj is a
// bias-free nn.Linear(256, 1024), :75-83 with use_mlp_cls = False):
//     f = Wp x;  logits_c = exp(logit_scale) * (f / |f|) . (t_c / |t_c|)
// At the training shape that is a (44646 x 256 x 1024) and a (44646 x 1024 x 1204) product and a 215 MB logit tensor of which only the
// row maxima are used.  Both products collapse onto the 256-wide input:
//     max_c logits_c = exp(logit_scale) * max_c (G x)_c / sqrt(x . (A x)),     G = T^ Wp (classes x 256),  A = Wp^T Wp (256 x 256)
// (the normalisation is one positive factor per row, so it commutes with the max), i.e. ONE (classes + 256) x 256 product per token with
// a reducing epilogue: 33 GFLOP instead of 133, nothing but one float per token written.  G and A are formed once per weight update
// by the caller (library GEMMs, fp32) and packed by msda_cls_pack into MFMA fragment order as bf16 hi + lo parts.
//
// Precision: the reference computes in fp32.  x and [G; A] are split into bf16 hi + lo parts and the product is formed as
// hi.hi + lo_w.hi_x + hi_w.lo_x with fp32 accumulation (the dropped lo.lo term is 2^-18 relative): errors at fp32 level (measured in
// tests/test_gpu_cls.py), at three bf16 MFMAs per tile -- the fp32 MFMA would need 16x the cycles of one.  bf16 activations have no lo
// part (two MFMAs); `parts = 1` drops the weights' lo part as well (what a bf16 autocast GEMM computes).
//
// Structure (as csrc/ffn_mfma.hip): transposed product, tokens on the lanes; mfma_f32_16x16x32_bf16; a wave owns 48 tokens (three column
// tiles) and keeps their x fragments in registers for the whole kernel (192 VGPRs with the lo parts, one wave per SIMD); the packed
// operand streams through LDS in tiles of 16 rows (16 KB), double-buffered through registers, one barrier per tile; accumulator
// tile (rows 4 q + i on the registers, token on the lane) -> running max / x . (A x) partial sums in registers, folded over the four
// lane groups once at the end.
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

constexpr int kD = 256;            // width of the encoder memory (8 k-steps of 32)
constexpr int kTokWave = 48;
constexpr int kWaves = 4;
constexpr int kTokWg = kTokWave * kWaves;
constexpr int kFragShorts = 512;   // one MFMA operand fragment: 64 lanes x 8 bf16
constexpr int kTileShorts = 2 * 8 * kFragShorts;   // 16 rows x 256 k, hi + lo parts: 16 KB
constexpr int kMaxClasses = 8192;

__device__ __forceinline__ unsigned pack_bf16(float a, float b)   // one v_cvt_pk_bf16_f32 (round to nearest even)
{
    const bf16x2_t p = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

// [G (classes x 256); zero rows up to a multiple of 16; A (256 x 256)] fp32 -> packed[tile][part][k-step][lane][8] bf16, lane (r, q)
// of a fragment = row 16 tile + r, k = 32 step + 8 q + 0..7; part 0 = bf16(v), part 1 = bf16(v - part 0).
__global__ void cls_pack_kernel(const float *__restrict__ G, int classes, const float *__restrict__ A, uint16_t *__restrict__ packed,
                                int n_tiles)
{
    const int ct = (classes + 15) / 16;
    const long long n = (long long)n_tiles * 8 * 64 * 8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), s = (int)((i >> 9) & 7), t = (int)(i >> 12);
        const int row = 16 * t + (lane & 15), k = 32 * s + 8 * (lane >> 4) + j;
        float v = 0.f;
        if (t < ct) {
            if (row < classes) v = G[(long long)row * kD + k];
        } else {
            v = A[(long long)(row - 16 * ct) * kD + k];
        }
        const unsigned hi = pack_bf16(v, 0.f) & 0xFFFFu;
        const unsigned lo = pack_bf16(v - __uint_as_float(hi << 16), 0.f) & 0xFFFFu;
        const long long base = (long long)t * kTileShorts + (long long)s * kFragShorts + lane * 8 + j;
        packed[base] = (uint16_t)hi;
        packed[base + 8 * kFragShorts] = (uint16_t)lo;
    }
}

// XF32: x is fp32 (split into hi + lo here), else bf16.  WPARTS: 2 = weights' hi + lo parts, 1 = hi part only.
template <bool XF32, int WPARTS>
__global__ __launch_bounds__(kWaves * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void cls_score_kernel(const void *__restrict__ xv, const uint16_t *__restrict__ packed, int T, int classes, float scale,
                      float *__restrict__ scores)
{
    __shared__ __attribute__((aligned(16))) short wbuf[2][WPARTS * 8 * kFragShorts];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int tok0 = blockIdx.x * kTokWg + wave * kTokWave;
    const int ct = (classes + 15) / 16, n_tiles = ct + kD / 16;
    constexpr int kChunks = WPARTS * 8 * kFragShorts * 2 / 16 / (kWaves * 64);   // 16-byte chunks per thread and tile: 4 or 2

    // this wave's x fragments (B operand): lane (c, q) holds x[token c][32 s + 8 q + 0..7]
    bf16x8 xh[3][8], xl[XF32 ? 3 : 1][8];
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
        const int tok = min(tok0 + 16 * t3 + c, T - 1);
        if (XF32) {
            const float *row = static_cast<const float *>(xv) + (size_t)tok * kD + 8 * q;
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
                xl[XF32 ? t3 : 0][s] = __builtin_bit_cast(bf16x8, l);
            }
        } else {
            const uint16_t *row = static_cast<const uint16_t *>(xv) + (size_t)tok * kD + 8 * q;
#pragma unroll
            for (int s = 0; s < 8; ++s) xh[t3][s] = *reinterpret_cast<const bf16x8 *>(row + 32 * s);
        }
    }

    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()}, nrm[3] = {0.f, 0.f, 0.f};

    // tile -> registers -> LDS, one tile ahead of the products
    u32x4 stage[kChunks];
    auto fetch = [&](int t) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(packed + (size_t)t * kTileShorts);
#pragma unroll
        for (int i = 0; i < kChunks; ++i) stage[i] = src[tid + i * (kWaves * 64)];
    };
    auto park = [&](int slot) {
        u32x4 *dst = reinterpret_cast<u32x4 *>(wbuf[slot]);
#pragma unroll
        for (int i = 0; i < kChunks; ++i) dst[tid + i * (kWaves * 64)] = stage[i];
    };
    fetch(0);
    park(0);
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        if (t + 1 < n_tiles) fetch(t + 1);
        const short *wt = wbuf[t & 1];
        f32x4 acc[3];
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) acc[t3] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(wt + s * kFragShorts + lane * 8);
            bf16x8 al;
            if (WPARTS == 2) al = *reinterpret_cast<const bf16x8 *>(wt + (8 + s) * kFragShorts + lane * 8);
            // the small products first; the three token tiles between two MFMAs on the same accumulator
            if (WPARTS == 2) {
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) acc[t3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh[t3][s], acc[t3], 0, 0, 0);
            }
            if (XF32) {
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) acc[t3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl[XF32 ? t3 : 0][s], acc[t3], 0, 0, 0);
            }
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) acc[t3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh[t3][s], acc[t3], 0, 0, 0);
        }
        if (t < ct) {   // class rows 16 t + 4 q + i: running maximum (rows past `classes` are padding)
            const int row0 = 16 * t + 4 * q;
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (row0 + i < classes) mx[t3] = fmaxf(mx[t3], acc[t3][i]);
        } else {        // rows of A x: channels 16 (t - ct) + 4 q + i, times x of the same channels
            const int ch0 = 16 * (t - ct) + 4 * q;
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
                const int tok = min(tok0 + 16 * t3 + c, T - 1);
                float xs[4];
                if (XF32) {
                    const float4 v = *reinterpret_cast<const float4 *>(static_cast<const float *>(xv) + (size_t)tok * kD + ch0);
                    xs[0] = v.x; xs[1] = v.y; xs[2] = v.z; xs[3] = v.w;
                } else {
                    const uint2 v = *reinterpret_cast<const uint2 *>(static_cast<const uint16_t *>(xv) + (size_t)tok * kD + ch0);
                    xs[0] = bf16_lo(v.x); xs[1] = bf16_hi(v.x); xs[2] = bf16_lo(v.y); xs[3] = bf16_hi(v.y);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) nrm[t3] = fmaf(acc[t3][i], xs[i], nrm[t3]);
            }
        }
        if (t + 1 < n_tiles) park((t + 1) & 1);
        __syncthreads();
    }

    // fold the four lane groups (rows 4 q + i live on lane group q), one store per token
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
        float m = mx[t3], n2 = nrm[t3];
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        n2 += __shfl_xor(n2, 16, 64);
        n2 += __shfl_xor(n2, 32, 64);
        const int tok = tok0 + 16 * t3 + c;
        if (q == 0 && tok < T) scores[tok] = scale * m / sqrtf(n2);
    }
}

template <bool XF32, int WPARTS>
int launch_scores(const void *x, const uint16_t *packed, int tokens, int classes, float scale, float *scores, hipStream_t stream)
{
    const int grid = (tokens + kTokWg - 1) / kTokWg;
    hipLaunchKernelGGL((cls_score_kernel<XF32, WPARTS>), dim3(grid), dim3(kWaves * 64), 0, stream, x, packed, tokens, classes, scale, scores);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

}  // namespace

extern "C" {

int msda_cls_packed_elems(int classes, int64_t *elems)
{
    if (!elems) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (classes < 1 || classes > kMaxClasses) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    *elems = (int64_t)((classes + 15) / 16 + kD / 16) * kTileShorts;
    return MSDA_OK;
}

int msda_cls_pack(const float *G, int classes, const float *A, int d_model, uint16_t *packed, msda_stream_t stream)
{
    if (!G || !A || !packed) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (classes < 1 || classes > kMaxClasses || d_model != kD) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    const int n_tiles = (classes + 15) / 16 + kD / 16;
    hipLaunchKernelGGL(cls_pack_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream), G, classes, A, packed, n_tiles);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

int msda_cls_max_scores(const void *x, int x_is_bf16, const uint16_t *packed, int tokens, int d_model, int classes, float scale, int parts,
                        float *scores, msda_stream_t stream)
{
    if (!x || !packed || !scores) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (tokens < 0 || d_model != kD || classes < 1 || classes > kMaxClasses || (parts != 1 && parts != 2)) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed)) & 15) return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    if (tokens == 0) return MSDA_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (x_is_bf16)
        return parts == 2 ? launch_scores<false, 2>(x, packed, tokens, classes, scale, scores, st)
                          : launch_scores<false, 1>(x, packed, tokens, classes, scale, scores, st);
    return parts == 2 ? launch_scores<true, 2>(x, packed, tokens, classes, scale, scores, st)
                      : launch_scores<true, 1>(x, packed, tokens, classes, scale, scores, st);
}

}  // extern "C"
