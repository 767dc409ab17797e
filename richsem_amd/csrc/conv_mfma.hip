// conv_mfma.hip -- 2-d convolution forward as an implicit GEMM on the gfx950 matrix cores, with the per-channel affine that follows
// every convolution of the backbones (frozen BatchNorm), the residual add and the ReLU in its epilogue (SURVEY.md section 8a rows a10 /
// a11: "conv-as-GEMM MFMA target").
//
// Reference call sites: the frozen CLIP teacher clip/model.py:10-56 (Bottleneck: conv 1x1 -> bn -> relu -> conv 3x3 -> bn -> relu ->
// avgpool -> conv 1x1 -> bn -> + identity -> relu) and :94-167 (ModifiedResNet stem and stages), called at
// models/richsem/richsem.py:628; the detector backbone models/richsem/backbone.py:20-56 (FrozenBatchNorm2d: y = x * scale + shift with
// scale = w * rsqrt(var + eps), shift = b - mean * scale) around torchvision's ResNet-50 convolutions, and the 1 x 1 / 3 x 3 input
// projections models/richsem/richsem.py:295-310.  bf16 storage, fp32 accumulation: new capability (the reference runs fp32 cuDNN).
//
// Layout: activations NHWC (channels last: the k dimension of the GEMM -- (kh, kw, ci) -- is contiguous in ci, so an MFMA operand
// fragment is one 16-byte load per lane, and the output tile's four consecutive channels per lane are one 8-byte store).
// Everything is computed TRANSPOSED as in csrc/ffn_mfma.hip: out^T (C_out x pixels) = W (C_out x K) . im2col^T (K x pixels) with
// mfma_f32_16x16x32_bf16: pixels on the lanes, output channels in the registers.
//   * a wave owns 48 output pixels (three column tiles) x up to 256 output channels (CO_TILES row tiles: 192 accumulators);
//   * the weights are packed once (msda_conv_pack_weight) into fragment order per (channel block, k-step, row tile) and stream through
//     LDS, double-buffered through registers, one barrier per k-step of 32; all four waves of a workgroup share them;
//   * the im2col operand is never formed: lane (pixel c, group q) loads x[n, ho s + kh - p, wo s + kw - p, 32 cb + 8 q ..] one k-step
//     ahead of its MFMAs (zero fragment outside the image).
// C_in must be a multiple of 32 (the 3-channel stems go through msda_conv_patches_bf16 first: explicit patches of a few channels,
// padded to 32, then a 1 x 1 convolution); C_out a multiple of 16 * CO_TILES with CO_TILES in {2, 4, 8, 16}.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "../../include/richsem_msda.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int kWaves = 4;
constexpr int kPixWave = 48;
constexpr int kPixWg = kPixWave * kWaves;
constexpr int kFragShorts = 512;

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    const bf16x2_t p = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

struct ConvGeom {
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
};

// weight (Cout, Cin, KH, KW) fp32, torch layout -> packed[co block][k-step][row tile][lane][8] bf16;
// k-step s = (kh KW + kw) (Cin / 32) + cb; lane (r, q) = output channel 16 tile + r, input channel 32 cb + 8 q + 0..7
__global__ void conv_pack_kernel(const float *__restrict__ w, uint16_t *__restrict__ packed, int Cout, int Cin, int KH, int KW, int co_tiles)
{
    const int cpb = Cin / 32, S = KH * KW * cpb;
    const long long n = (long long)Cout * Cin * KH * KW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        long long r = i >> 9;
        const int tile = (int)(r % co_tiles);
        r /= co_tiles;
        const int s = (int)(r % S), blk = (int)(r / S);
        const int co = (blk * co_tiles + tile) * 16 + (lane & 15);
        const int tap = s / cpb, cb = s - tap * cpb;
        const int ci = 32 * cb + 8 * (lane >> 4) + j, kh = tap / KW, kw = tap - kh * KW;
        packed[i] = (uint16_t)(pack_bf16(w[(((long long)co * Cin + ci) * KH + kh) * KW + kw], 0.f) & 0xFFFFu);
    }
}

// x (N, H, W, C) bf16 with small C -> patches (N Ho Wo, Kpad) bf16, k = (kh KW + kw) C + ci, zero beyond KH KW C
__global__ void conv_patches_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ patches, int N, int H, int W, int C, int Ho,
                                    int Wo, int KH, int KW, int stride, int pad, int Kpad)
{
    const long long n = (long long)N * Ho * Wo * Kpad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad);
        const long long p = i / Kpad;
        uint16_t v = 0;
        if (k < KH * KW * C) {
            const int ci = k % C, tap = k / C, kh = tap / KW, kw = tap - kh * KW;
            const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), b = (int)(p / ((long long)Wo * Ho));
            const int hi = ho * stride + kh - pad, wi = wo * stride + kw - pad;
            if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = x[(((long long)b * H + hi) * W + wi) * C + ci];
        }
        patches[i] = v;
    }
}

template <int CO_TILES>
__global__ __launch_bounds__(kWaves * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv_fwd_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ wpk, const float *__restrict__ scale,
                     const float *__restrict__ shift, const uint16_t *__restrict__ residual, uint16_t *__restrict__ out, ConvGeom g,
                     int relu)
{
    __shared__ __attribute__((aligned(16))) short wbuf[2][CO_TILES * kFragShorts];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const long long P = (long long)g.N * g.Ho * g.Wo;
    const long long pix0 = (long long)blockIdx.x * kPixWg + wave * kPixWave;
    const int cpb = g.Cin / 32, S = g.KH * g.KW * cpb;
    const int co0 = blockIdx.y * (16 * CO_TILES);

    // this lane's three output pixels: image base, top-left input coordinate of the receptive field
    long long img[3];
    int hi0[3], wi0[3];
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
        long long p = pix0 + 16 * t3 + c;
        if (p > P - 1) p = P - 1;
        const int wo = (int)(p % g.Wo), ho = (int)((p / g.Wo) % g.Ho);
        img[t3] = (p / ((long long)g.Wo * g.Ho)) * g.H * g.W;
        hi0[t3] = ho * g.stride - g.pad;
        wi0[t3] = wo * g.stride - g.pad;
    }

    f32x4 acc[3][CO_TILES];
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
        for (int t = 0; t < CO_TILES; ++t) acc[t3][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // weights: tile of k-step s for this channel block -> registers -> LDS
    constexpr int kTileChunks = CO_TILES * kFragShorts * 2 / 16;                       // 16-byte chunks per tile
    constexpr int kChunks = (kTileChunks + kWaves * 64 - 1) / (kWaves * 64);
    const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(wpk + (size_t)blockIdx.y * S * CO_TILES * kFragShorts);
    u32x4 stage[kChunks];
    auto fetch = [&](int s) {
#pragma unroll
        for (int i = 0; i < kChunks; ++i) {
            const int idx = tid + i * (kWaves * 64);
            if (idx < kTileChunks) stage[i] = wsrc[(size_t)s * kTileChunks + idx];
        }
    };
    auto park = [&](int slot) {
        u32x4 *dst = reinterpret_cast<u32x4 *>(wbuf[slot]);
#pragma unroll
        for (int i = 0; i < kChunks; ++i) {
            const int idx = tid + i * (kWaves * 64);
            if (idx < kTileChunks) dst[idx] = stage[i];
        }
    };
    // im2col fragment of k-step (tap kh, kw; channel block cb) for the three pixels
    auto gather = [&](int kh, int kw, int cb, bf16x8 *b) {
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            const int hi = hi0[t3] + kh, wi = wi0[t3] + kw;
            const bool in = hi >= 0 && hi < g.H && wi >= 0 && wi < g.W;
            const long long off = in ? ((img[t3] + (long long)hi * g.W + wi) * g.Cin + 32 * cb + 8 * q) : 0;
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + off);
            b[t3] = in ? v : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    };

    bf16x8 bcur[3], bnext[3];
    int kh = 0, kw = 0, cb = 0;
    gather(0, 0, 0, bnext);
    fetch(0);
    park(0);
    __syncthreads();

    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) bcur[t3] = bnext[t3];
        if (s + 1 < S) {
            if (++cb == cpb) {
                cb = 0;
                if (++kw == g.KW) { kw = 0; ++kh; }
            }
            fetch(s + 1);
            gather(kh, kw, cb, bnext);
        }
        const short *wt = wbuf[s & 1];
#pragma unroll
        for (int t = 0; t < CO_TILES; ++t) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(wt + t * kFragShorts + lane * 8);
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) acc[t3][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bcur[t3], acc[t3][t], 0, 0, 0);
        }
        if (s + 1 < S) park((s + 1) & 1);
        __syncthreads();
    }

    // epilogue: lane (c, q) holds channels co0 + 16 t + 4 q + 0..3 of pixel c: affine, residual, relu, 8-byte store
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
        const long long p = pix0 + 16 * t3 + c;
        if (p >= P) continue;
#pragma unroll
        for (int t = 0; t < CO_TILES; ++t) {
            const int co = co0 + 16 * t + 4 * q;
            const float4 sc = *reinterpret_cast<const float4 *>(scale + co), sh = *reinterpret_cast<const float4 *>(shift + co);
            float y0 = fmaf(acc[t3][t][0], sc.x, sh.x), y1 = fmaf(acc[t3][t][1], sc.y, sh.y);
            float y2 = fmaf(acc[t3][t][2], sc.z, sh.z), y3 = fmaf(acc[t3][t][3], sc.w, sh.w);
            if (residual) {
                const uint2 r = *reinterpret_cast<const uint2 *>(residual + p * g.Cout + co);
                y0 += bf16_lo(r.x); y1 += bf16_hi(r.x); y2 += bf16_lo(r.y); y3 += bf16_hi(r.y);
            }
            if (relu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); y2 = fmaxf(y2, 0.f); y3 = fmaxf(y3, 0.f); }
            uint2 o;
            o.x = pack_bf16(y0, y1);
            o.y = pack_bf16(y2, y3);
            *reinterpret_cast<uint2 *>(out + p * g.Cout + co) = o;
        }
    }
}

template <int CO_TILES>
int launch_conv(const uint16_t *x, const uint16_t *wpk, const float *scale, const float *shift, const uint16_t *residual, uint16_t *out,
                const ConvGeom &g, int relu, hipStream_t stream)
{
    const long long P = (long long)g.N * g.Ho * g.Wo;
    const dim3 grid((unsigned)((P + kPixWg - 1) / kPixWg), (unsigned)(g.Cout / (16 * CO_TILES)));
    hipLaunchKernelGGL(conv_fwd_kernel<CO_TILES>, grid, dim3(kWaves * 64), 0, stream, x, wpk, scale, shift, residual, out, g, relu);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

int co_tiles_for(int Cout)
{
    if (Cout % 256 == 0) return 16;
    if (Cout % 128 == 0) return 8;
    if (Cout % 64 == 0) return 4;
    if (Cout % 32 == 0) return 2;
    return 0;
}

}  // namespace

extern "C" {

int msda_conv_pack_weight(const float *weight, int Cout, int Cin, int KH, int KW, uint16_t *packed, msda_stream_t stream)
{
    if (!weight || !packed) return MSDA_ERR_NULL_POINTER;
    const int ct = co_tiles_for(Cout);
    if (Cout < 1 || Cin < 32 || Cin % 32 != 0 || KH < 1 || KW < 1 || KH > 16 || KW > 16 || ct == 0) return MSDA_ERR_BAD_DIMS;
    hipLaunchKernelGGL(conv_pack_kernel, dim3(512), dim3(256), 0, static_cast<hipStream_t>(stream), weight, packed, Cout, Cin, KH, KW, ct);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

int msda_conv_patches_bf16(const uint16_t *x, int N, int H, int W, int C, int KH, int KW, int stride, int pad, int Kpad,
                           uint16_t *patches, msda_stream_t stream)
{
    if (!x || !patches) return MSDA_ERR_NULL_POINTER;
    if (N < 1 || H < 1 || W < 1 || C < 1 || KH < 1 || KW < 1 || stride < 1 || pad < 0 || Kpad < KH * KW * C || Kpad % 32 != 0)
        return MSDA_ERR_BAD_DIMS;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho < 1 || Wo < 1) return MSDA_ERR_BAD_DIMS;
    const long long n = (long long)N * Ho * Wo * Kpad;
    const int grid = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    hipLaunchKernelGGL(conv_patches_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), x, patches, N, H, W, C, Ho, Wo, KH,
                       KW, stride, pad, Kpad);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

int msda_conv_forward_bf16(const uint16_t *x, const uint16_t *packed_weight, const float *scale, const float *shift,
                           const uint16_t *residual, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu,
                           uint16_t *out, msda_stream_t stream)
{
    if (!x || !packed_weight || !scale || !shift || !out) return MSDA_ERR_NULL_POINTER;
    const int ct = co_tiles_for(Cout);
    if (N < 1 || H < 1 || W < 1 || Cin < 32 || Cin % 32 != 0 || ct == 0 || KH < 1 || KW < 1 || KH > 16 || KW > 16 || stride < 1 || pad < 0)
        return MSDA_ERR_BAD_DIMS;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho < 1 || Wo < 1) return MSDA_ERR_BAD_DIMS;
    if ((long long)N * H * W * Cin >= (1ll << 40) || (long long)N * Ho * Wo * Cout >= (1ll << 40)) return MSDA_ERR_TOO_LARGE;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed_weight) | reinterpret_cast<uintptr_t>(scale) |
         reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(residual)) & 15)
        return MSDA_ERR_MISALIGNED;
    const ConvGeom g{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad};
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (ct) {
    case 16: return launch_conv<16>(x, packed_weight, scale, shift, residual, out, g, relu, st);
    case 8: return launch_conv<8>(x, packed_weight, scale, shift, residual, out, g, relu, st);
    case 4: return launch_conv<4>(x, packed_weight, scale, shift, residual, out, g, relu, st);
    default: return launch_conv<2>(x, packed_weight, scale, shift, residual, out, g, relu, st);
    }
}

}  // extern "C"
