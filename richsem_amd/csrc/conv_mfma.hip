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
//   * a wave owns 16 PT output pixels (PT <= 3 column tiles) x 16 CO_TILES <= 256 output channels (192 accumulators at most); the host
//     picks (CO_TILES, PT) per call (choose_tiling: measured -- small tiles, i.e. more waves per SIMD, win almost everywhere);
//   * the weights are packed once (msda_conv_pack_weight) into fragment order per (k-step, row tile) and stream through
//     LDS, double-buffered through registers, one barrier per k-step of 32; all four waves of a workgroup share them;
//   * the im2col operand is never formed: lane (pixel c, group q) loads x[n, ho s + kh - p, wo s + kw - p, 32 cb + 8 q ..] one k-step
//     ahead of its MFMAs (zero fragment outside the image).
// C_in must be a multiple of 32, except for inputs of a few channels (the 3-channel stems): their operand fragments are gathered
// element by element through a k -> (kh, kw, ci) table in LDS (k = (kh KW + kw) C_in + ci, padded to a multiple of 32; the weight is
// packed in that order).  C_out must be a multiple of 16.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include <atomic>
#include <initializer_list>
#include <type_traits>

#include "../../include/richsem_msda.h"

extern "C" int msda_note_error(int code, const char *entry);      // msda_api.hip: sets msda_last_error()

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int kWaves = 4;
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
    int up;   // > 1: the input is read as if zero-upsampled by `up` (transposed convolution = gradient w.r.t. the input of a strided one)
};

// Operand orders that make the memory accesses wide (both are permutations INSIDE the product, fixed by C_in / C_out alone, applied
// by msda_conv_pack_weight and undone nowhere):
//   * output channels: inside every group of G = min(4, largest power of two dividing C_out / 16) row tiles, row 4 q + i of tile u is
//     channel q (4 G) + 4 u + i of the group -- so lane group q of an accumulator holds 4 G CONSECUTIVE channels of its pixel across the
//     group's tiles (32 bytes of bf16 for G = 4: the four lane groups of a pixel write one 128-byte line);
//   * input channels, when C_in % 64 == 0: k-steps are taken in pairs over 64 channels, lane group q covering channels 16 q + 8 h + 0..7
//     in step h of the pair -- so a lane loads 32 consecutive bytes per pair and the four lane groups of a pixel read a whole line.
__host__ __device__ inline int conv_group(int Cout)
{
    const int tiles = Cout / 16;
    return tiles % 4 == 0 ? 4 : (tiles % 2 == 0 ? 2 : 1);
}

// weight (Cout, Cin, KH, KW) fp32, torch layout -> packed[k-step][row tile][lane][8] bf16, K = KH KW Cin padded to Kpad (multiple of 32);
// flat k = (kh KW + kw) Cin + ci, zero beyond KH KW Cin, with the two permutations above.
__global__ void conv_pack_kernel(const float *__restrict__ w, uint16_t *__restrict__ packed, int Cout, int Cin, int KH, int KW, int Kpad)
{
    const int tiles = Cout / 16, K = KH * KW * Cin, G = conv_group(Cout);
    const bool pairs = Cin % 64 == 0;
    const long long n = (long long)Cout * Kpad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const long long r = i >> 9;
        const int tile = (int)(r % tiles), s = (int)(r / tiles);
        const int row = lane & 15, q = lane >> 4;
        const int co = (tile / G) * 16 * G + (row >> 2) * 4 * G + 4 * (tile % G) + (row & 3);
        const int k = pairs ? 64 * (s >> 1) + 16 * q + 8 * (s & 1) + j : 32 * s + 8 * q + j;
        float v = 0.f;
        if (k < K) {
            const int ci = k % Cin, tap = k / Cin, kh = tap / KW, kw = tap - kh * KW;
            v = w[(((long long)co * Cin + ci) * KH + kh) * KW + kw];
        }
        packed[i] = (uint16_t)(pack_bf16(v, 0.f) & 0xFFFFu);
    }
}

// epilogue: lane (c, q) holds, for every group of G tiles, the 4 G consecutive channels co0 + 16 G grp + 4 G q + (4 u + i) of pixel c
// (tile u of the group, register i): affine, residual, relu (or the mask of an input gradient taken THROUGH a ReLU: zero where the
// forward's activation `mask` was not positive), one 8 G-byte store
template <int CO_TILES, int PT>
__device__ __forceinline__ void conv_epilogue(f32x4 (&acc)[PT][CO_TILES], const long long (&pout)[PT], int q, int co0, int Cout,
                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                              const uint16_t *__restrict__ residual, int relu, const uint16_t *__restrict__ mask,
                                              uint16_t *__restrict__ out)
{
    constexpr int G = CO_TILES >= 4 ? 4 : CO_TILES;
#pragma unroll
    for (int t3 = 0; t3 < PT; ++t3) {
        const long long p = pout[t3];      // the lane's output pixel of column tile t3 (-1: none)
        if (p < 0) continue;
#pragma unroll
        for (int grp = 0; grp < CO_TILES / G; ++grp) {
            const int co = co0 + 16 * G * grp + 4 * G * q;
            float y[4 * G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const float4 sc = scale ? *reinterpret_cast<const float4 *>(scale + co + 4 * u) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float4 sh = shift ? *reinterpret_cast<const float4 *>(shift + co + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
                y[4 * u + 0] = fmaf(acc[t3][grp * G + u][0], sc.x, sh.x);
                y[4 * u + 1] = fmaf(acc[t3][grp * G + u][1], sc.y, sh.y);
                y[4 * u + 2] = fmaf(acc[t3][grp * G + u][2], sc.z, sh.z);
                y[4 * u + 3] = fmaf(acc[t3][grp * G + u][3], sc.w, sh.w);
            }
            if (residual) {
                const uint2 *rp = reinterpret_cast<const uint2 *>(residual + p * Cout + co);
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const uint2 r = rp[u];
                    y[4 * u + 0] += bf16_lo(r.x); y[4 * u + 1] += bf16_hi(r.x); y[4 * u + 2] += bf16_lo(r.y); y[4 * u + 3] += bf16_hi(r.y);
                }
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4 * G; ++e) y[e] = fmaxf(y[e], 0.f);
            }
            if (mask) {      // (bf16 > 0  <=>  its bits, read as a signed 16-bit integer, are > 0)
                const uint2 *mp = reinterpret_cast<const uint2 *>(mask + p * Cout + co);
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const uint2 m = mp[u];
                    if ((short)(m.x & 0xFFFFu) <= 0) y[4 * u + 0] = 0.f;
                    if ((short)(m.x >> 16) <= 0) y[4 * u + 1] = 0.f;
                    if ((short)(m.y & 0xFFFFu) <= 0) y[4 * u + 2] = 0.f;
                    if ((short)(m.y >> 16) <= 0) y[4 * u + 3] = 0.f;
                }
            }
            unsigned o[2 * G];
#pragma unroll
            for (int e = 0; e < 2 * G; ++e) o[e] = pack_bf16(y[2 * e], y[2 * e + 1]);
            uint16_t *dst = out + p * Cout + co;
            if (G == 4) {
                reinterpret_cast<u32x4 *>(dst)[0] = (u32x4){o[0], o[1], o[2], o[3]};
                reinterpret_cast<u32x4 *>(dst)[1] = (u32x4){o[4 % (2 * G)], o[5 % (2 * G)], o[6 % (2 * G)], o[7 % (2 * G)]};
            } else if (G == 2) {
                reinterpret_cast<u32x4 *>(dst)[0] = (u32x4){o[0], o[1], o[2 % (2 * G)], o[3 % (2 * G)]};
            } else {
                reinterpret_cast<uint2 *>(dst)[0] = make_uint2(o[0], o[1]);
            }
        }
    }
}


// MODE 0: few-channel input (fragments gathered element by element through a table); 1: C_in % 32 == 0, one k-step per barrier;
// 2: C_in % 64 == 0, two k-steps (one 32-byte load per lane and pixel) per barrier.
template <int CO_TILES, int PT, int MODE, bool KSPLIT = false>
__global__ __launch_bounds__(kWaves * 64)
void conv_fwd_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ wpk, const float *__restrict__ scale,
                     const float *__restrict__ shift, const uint16_t *__restrict__ residual, uint16_t *__restrict__ out, ConvGeom g,
                     int relu, float *__restrict__ ksum, const uint16_t *__restrict__ mask)
{
    constexpr bool SMALLC = MODE == 0;
    constexpr int KT = MODE == 2 ? 2 : 1;
    constexpr int G = CO_TILES >= 4 ? 4 : CO_TILES;          // == conv_group(C_out): the host only launches matching shapes
    __shared__ __attribute__((aligned(16))) short wbuf[2][KT * CO_TILES * kFragShorts + (KSPLIT ? 64 : 0)];      // (+ 2 x 128 B: the k-slice epilogue's padded tile)
    __shared__ unsigned lut[SMALLC ? 512 : 1];     // k -> kh << 20 | kw << 12 | ci   (0xFFFFFFFF: padding)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const long long P = (long long)g.N * g.Ho * g.Wo;
    const long long pix0 = (long long)blockIdx.x * (kWaves * 16 * PT) + wave * (16 * PT);
    const int Ktot = g.KH * g.KW * g.Cin;
    const int cpb = SMALLC ? 1 : g.Cin / (32 * KT);                                    // iterations per tap
    const int S = SMALLC ? (Ktot + 31) / 32 : g.KH * g.KW * cpb;                      // iterations (KT k-steps each)
    const int tiles_all = g.Cout / 16;
    const int co0 = blockIdx.y * (16 * CO_TILES);

    if (SMALLC) {
        for (int k = tid; k < 32 * S; k += kWaves * 64) {
            unsigned v = 0xFFFFFFFFu;
            if (k < Ktot) {
                const int ci = k % g.Cin, tap = k / g.Cin, kh = tap / g.KW, kw = tap - kh * g.KW;
                v = (unsigned)kh << 20 | (unsigned)kw << 12 | (unsigned)ci;
            }
            lut[k] = v;
        }
    }

    // this lane's output pixels: image base, top-left input coordinate of the receptive field
    long long img[PT];
    int hi0[PT], wi0[PT];
#pragma unroll
    for (int t3 = 0; t3 < PT; ++t3) {
        long long p = pix0 + 16 * t3 + c;
        if (p > P - 1) p = P - 1;
        const int wo = (int)(p % g.Wo), ho = (int)((p / g.Wo) % g.Ho);
        img[t3] = (p / ((long long)g.Wo * g.Ho)) * g.H * g.W;
        hi0[t3] = ho * g.stride - g.pad;
        wi0[t3] = wo * g.stride - g.pad;
    }

    f32x4 acc[PT][CO_TILES];
#pragma unroll
    for (int t3 = 0; t3 < PT; ++t3)
#pragma unroll
        for (int t = 0; t < CO_TILES; ++t) acc[t3][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // weights: this channel block's row tiles of the iteration's k-steps -> registers -> LDS
    constexpr int kStepChunks = CO_TILES * kFragShorts * 2 / 16;                       // 16-byte chunks per k-step
    constexpr int kTileChunks = KT * kStepChunks;
    constexpr int kChunks = (kTileChunks + kWaves * 64 - 1) / (kWaves * 64);
    const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(wpk + (size_t)blockIdx.y * CO_TILES * kFragShorts);
    const size_t step_chunks = (size_t)tiles_all * kFragShorts * 2 / 16;
    u32x4 stage[kChunks];
    auto fetch = [&](int it) {
#pragma unroll
        for (int i = 0; i < kChunks; ++i) {
            const int idx = tid + i * (kWaves * 64);
            if (idx < kTileChunks) stage[i] = wsrc[(size_t)(KT * it + idx / kStepChunks) * step_chunks + idx % kStepChunks];
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
    // im2col fragments of an iteration for the lane's pixels: (tap kh, kw; channel block cb), or k-step cb through the table
    auto gather = [&](int kh, int kw, int cb, bf16x8 (*b)[PT]) {
        if (SMALLC) {
            unsigned e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = lut[32 * cb + 8 * q + j];
#pragma unroll
            for (int t3 = 0; t3 < PT; ++t3) {
                unsigned short v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int hi = hi0[t3] + (int)(e[j] >> 20), wi = wi0[t3] + (int)((e[j] >> 12) & 255);
                    const bool in = e[j] != 0xFFFFFFFFu && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W;
                    const long long off = in ? ((img[t3] + (long long)hi * g.W + wi) * g.Cin + (e[j] & 4095)) : 0;
                    const unsigned short u = x[off];
                    v[j] = in ? u : (unsigned short)0;
                }
                b[0][t3] = (bf16x8){(short)v[0], (short)v[1], (short)v[2], (short)v[3], (short)v[4], (short)v[5], (short)v[6], (short)v[7]};
            }
        } else {
#pragma unroll
            for (int t3 = 0; t3 < PT; ++t3) {
                int hi = hi0[t3] + kh, wi = wi0[t3] + kw;
                bool in = hi >= 0 && wi >= 0;
                if (g.up > 1) {      // virtual (zero-upsampled) coordinates: only multiples of `up` hold data
                    in = in && hi % g.up == 0 && wi % g.up == 0;
                    hi /= g.up;
                    wi /= g.up;
                }
                in = in && hi < g.H && wi < g.W;
                const long long off = in ? ((img[t3] + (long long)hi * g.W + wi) * g.Cin + 32 * KT * cb + 8 * KT * q) : 0;
#pragma unroll
                for (int h = 0; h < KT; ++h) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + off + 8 * h);
                    b[h][t3] = in ? v : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
        }
    };

    // k split (few pixels, long k: layer4's 3 x 3, the stride-2 projection of C5): slice blockIdx.z of gridDim.z takes iterations
    // [s0, s1) and ADDS its raw sums to the zeroed fp32 image `ksum`; conv_ksum_finish_kernel applies the epilogue
    const int s0 = !KSPLIT ? 0 : (int)((long long)S * blockIdx.z / gridDim.z), s1 = !KSPLIT ? S : (int)((long long)S * (blockIdx.z + 1) / gridDim.z);
    bf16x8 bcur[KT][PT], bnext[KT][PT];
    int cb = !KSPLIT ? 0 : s0 % cpb, kh = !KSPLIT ? 0 : (s0 / cpb) / g.KW, kw = !KSPLIT ? 0 : (s0 / cpb) % g.KW;
    if (SMALLC) __syncthreads();      // the table
    gather(kh, kw, cb, bnext);
    fetch(s0);
    park(0);
    __syncthreads();

    for (int s = s0; s < s1; ++s) {
#pragma unroll
        for (int h = 0; h < KT; ++h)
#pragma unroll
            for (int t3 = 0; t3 < PT; ++t3) bcur[h][t3] = bnext[h][t3];
        if (s + 1 < s1) {
            if (++cb == cpb && !SMALLC) {
                cb = 0;
                if (++kw == g.KW) { kw = 0; ++kh; }
            }
            fetch(s + 1);
            gather(kh, kw, cb, bnext);
        }
        const short *wt = wbuf[(s - s0) & 1];
#pragma unroll
        for (int h = 0; h < KT; ++h)
#pragma unroll
            for (int t = 0; t < CO_TILES; ++t) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(wt + (h * CO_TILES + t) * kFragShorts + lane * 8);
#pragma unroll
                for (int t3 = 0; t3 < PT; ++t3) acc[t3][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bcur[h][t3], acc[t3][t], 0, 0, 0);
            }
        if (s + 1 < s1) park((s + 1 - s0) & 1);
        __syncthreads();
    }

    if constexpr (KSPLIT) {      // a k slice (host: MODE 2, PT 1): raw sums to the fp32 image, as whole lines
        // A lane holds 4 G consecutive channels of ONE pixel: adding them from there would be 64 single-float atomics on 64 different
        // lines per instruction (measured: layer4's 3 x 3 55 -> 218 us).  The wave's 16 x (16 CO_TILES) tile goes through LDS (the weight
        // buffers are free now) and comes back with the lanes along the channels: one instruction adds 256 contiguous bytes of a pixel.
        constexpr int CH = 16 * CO_TILES, LD = CH + 1;
        static_assert(PT == 1 && MODE == 2 && 2 * (KT * CO_TILES * kFragShorts + 64) * 2 >= kWaves * 16 * LD * 4, "k split: tile does not fit the weight buffers");
        float *tile = reinterpret_cast<float *>(&wbuf[0][0]) + wave * (16 * LD);
#pragma unroll
        for (int grp = 0; grp < CO_TILES / G; ++grp)
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) tile[c * LD + 16 * G * grp + 4 * G * q + 4 * u + i] = acc[0][grp * G + u][i];
        __syncthreads();
        for (int px = 0; px < 16; ++px) {
            const long long p = pix0 + px;
            if (p >= P) break;      // (uniform per wave)
            for (int ch = lane; ch < CH; ch += 64) atomicAdd(ksum + p * g.Cout + co0 + ch, tile[px * LD + ch]);
        }
        return;
    }

    long long pout[PT];
#pragma unroll
    for (int t3 = 0; t3 < PT; ++t3) pout[t3] = pix0 + 16 * t3 + c < P ? pix0 + 16 * t3 + c : -1;
    conv_epilogue<CO_TILES, PT>(acc, pout, q, co0, g.Cout, scale, shift, residual, relu, mask, out);
}

// ---- the same convolution with BOTH operands prefetched several iterations ahead through LDS rings (round 4) -------------------------
// conv_fwd_kernel asks for an iteration's operands one iteration (KT x CO_TILES MFMAs = 256-512 cycles) before it uses them and leaves
// the rest of the 1-2 k cycles of memory latency to the other waves of its SIMD.  The deep layers of a two-image batch (2100-8400
// pixels: 33-132 pixel tiles) give a SIMD ONE wave, so every iteration waits out a memory round trip (layer3's 1 x 1 over 1024 channels:
// 16 iterations, 19.8 us).  Here nothing is staged in registers:
//   * weights: the iteration's KT x CO_TILES fragments go by LDS DMA (global_load_lds_dwordx4, 1 KB per wave instruction, already in
//     fragment order) into slot it % R of a ring shared by the workgroup, each wave bringing a quarter;
//   * activations: the lane's 16 bytes of an im2col fragment ARE its global_load_lds_dwordx4 element -- the fragment lands in the wave's
//     own ring in the order ds_read_b128 hands it back; lanes outside the image (padding, the zero-upsampled gradient of a strided
//     convolution) fetch a zero line instead of branching;
//   * R - 1 iterations are in flight; an iteration begins with s_waitcnt vmcnt((R - 2) x requests per iteration) and ONE bare s_barrier
//     (everybody's share of the weights has landed; everybody has left the slot that is requested next).  Past the last iteration the
//     requests repeat the last one into a free slot, so the count is the same in every iteration.
//   * the operand reads are inline ds_read_b128 (the compiler would wait for every DMA in flight before an LDS read it can see),
//     eight fragments ahead of their MFMAs, released by partial lgkmcnt waits (csrc/ffn_mfma.hip's scheme).
// C_in % 64 == 0 (two k-steps per iteration), CO_TILES >= 2.  Geometry, packing and epilogue are conv_fwd_kernel's.
__device__ __attribute__((aligned(16))) unsigned g_conv_zero_line[4];      // (device globals are zero-initialised)

#define CONV_LDS_READ(dst, addr, byte_off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(byte_off))
#define CONV_LDS_WAIT(n, a) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(n))

// fragments I .. N - 1 of a slot (1 KB apart) -> dst[I .. N - 1]   (compile-time recursion: the offsets and counts are instruction immediates)
template <int I, int N>
__device__ __forceinline__ void ring_read(u32x4 *dst, unsigned addr)
{
    if constexpr (I < N) {
        CONV_LDS_READ(dst[I], addr, I * 1024);
        ring_read<I + 1, N>(dst, addr);
    }
}

// weight fragments p .. WFR - 1 of an iteration against its activation fragments: fragment p lives in fr[p & 7], is waited for with "at
// most min(7, WFR - 1 - p) newer reads in flight", feeds PT MFMAs and is at once replaced by the read of fragment p + 8
template <int P_, int CO_TILES, int PT>
__device__ __forceinline__ void ring_products(f32x4 (&acc)[PT][CO_TILES], u32x4 *fr, const u32x4 *bfr, unsigned wa)
{
    constexpr int WFR = 2 * CO_TILES;
    if constexpr (P_ < WFR) {
        constexpr int h = P_ / CO_TILES, t = P_ % CO_TILES;
        constexpr int newer = (WFR - 1 - P_) < 7 ? (WFR - 1 - P_) : 7;
        CONV_LDS_WAIT(newer, fr[P_ & 7]);      // (the activation fragments were asked for before fragment 0)
        const bf16x8 a = __builtin_bit_cast(bf16x8, fr[P_ & 7]);
#pragma unroll
        for (int t3 = 0; t3 < PT; ++t3)
            acc[t3][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, bfr[h * PT + t3]), acc[t3][t], 0, 0, 0);
        if constexpr (P_ + 8 < WFR) CONV_LDS_READ(fr[P_ & 7], wa, (P_ + 8) * 1024);
        ring_products<P_ + 1, CO_TILES, PT>(acc, fr, bfr, wa);
    }
}

// the request side of conv_ring_kernel: which iteration is asked for next (tap kh, kw; channel block cb) and this lane's sources for it.
// (A __device__ member, not a lambda of the kernel: the LDS-DMA builtin inside a lambda keeps hipcc from emitting the kernel's host stub.)
template <int CO_TILES, int PT>
struct RingRequest {
    static constexpr int KT = 2, WFR = KT * CO_TILES, WPW = WFR / kWaves, AFR = KT * PT;
    const unsigned char *wtap[WPW];      // this wave's weight fragments of tap (0, 0), channel block 0 (the lane's 16 bytes of each)
    const unsigned char *wsrc[WPW];      // ... of the iteration that is requested next
    const uint16_t *asrc[PT];            // the lane's pixel row at the iteration's tap (+ 16 q channels), or null outside the image
    size_t step_bytes;                   // one k-step of the packed weight
    int it, cb, kh, kw;
    int kh0, kw0, kstep;                 // the taps walked: kh0, kh0 + kstep, .. x kw0, kw0 + kstep, ..  (all of them: 0, 0, 1)

    __device__ __forceinline__ void tap_sources(const uint16_t *x, const ConvGeom &g, int cpb, const long long (&img)[PT], const int (&hi0)[PT],
                                                const int (&wi0)[PT], int q)
    {
#pragma unroll
        for (int t3 = 0; t3 < PT; ++t3) {
            int hi = hi0[t3] + kh, wi = wi0[t3] + kw;
            bool in = hi >= 0 && wi >= 0;
            if (g.up > 1) {      // virtual (zero-upsampled) coordinates: only multiples of `up` hold data
                in = in && hi % g.up == 0 && wi % g.up == 0;
                hi /= g.up;
                wi /= g.up;
            }
            in = in && hi < g.H && wi < g.W;
            asrc[t3] = in ? x + (img[t3] + (long long)hi * g.W + wi) * g.Cin + 16 * q : nullptr;
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) wsrc[i] = wtap[i] + (size_t)(kh * g.KW + kw) * cpb * KT * step_bytes;
    }

    template <int R>
    __device__ __forceinline__ void request(int slot, int wave, int S, int cpb, const uint16_t *x, const ConvGeom &g, const long long (&img)[PT],
                                            const int (&hi0)[PT], const int (&wi0)[PT], int q, unsigned char *wring, unsigned char *aring,
                                            const uint16_t *zero)
    {
        unsigned char *wdst = wring + (slot * WFR + wave * WPW) * 1024;
#ifndef CONV_RING_ABLATE      // diagnostic builds (wrong results): 1 = no weight requests, 2 = no activation requests, 3 = neither
#define CONV_RING_ABLATE 0
#endif
#pragma unroll
        for (int i = 0; i < ((CONV_RING_ABLATE & 1) ? 0 : WPW); ++i)
            __builtin_amdgcn_global_load_lds(wsrc[i], reinterpret_cast<__attribute__((address_space(3))) void *>(reinterpret_cast<uintptr_t>(wdst + i * 1024)),
                                             16, 0, 0);
        unsigned char *adst = aring + ((wave * R + slot) * AFR) * 1024;
#pragma unroll
        for (int h = 0; h < ((CONV_RING_ABLATE & 2) ? 0 : KT); ++h)
#pragma unroll
            for (int t3 = 0; t3 < PT; ++t3)
                __builtin_amdgcn_global_load_lds(asrc[t3] ? asrc[t3] + 64 * cb + 8 * h : zero,
                                                 reinterpret_cast<__attribute__((address_space(3))) void *>(reinterpret_cast<uintptr_t>(adst + (h * PT + t3) * 1024)),
                                                 16, 0, 0);
        if (it + 1 < S) {      // (uniform) the next one; past the end the last one is repeated into a free slot
            ++it;
            if (++cb == cpb) {
                cb = 0;
                kw += kstep;
                if (kw >= g.KW) { kw = kw0; kh += kstep; }
                tap_sources(x, g, cpb, img, hi0, wi0, q);
            } else {
#pragma unroll
                for (int i = 0; i < WPW; ++i) wsrc[i] += KT * step_bytes;
            }
        }
    }
};

template <int CO_TILES, int PT, int R>
__global__ __launch_bounds__(kWaves * 64)
void conv_ring_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ wpk, const float *__restrict__ scale,
                      const float *__restrict__ shift, const uint16_t *__restrict__ residual, uint16_t *__restrict__ out, ConvGeom g,
                      int relu, const uint16_t *__restrict__ mask)
{
    constexpr int KT = 2;
    constexpr int WFR = KT * CO_TILES;            // weight fragments (1 KB each) per iteration
    constexpr int WPW = WFR / kWaves;             // ... requested by each wave
    constexpr int AFR = KT * PT;                  // activation fragments per wave and iteration
    constexpr int NPI = WPW + AFR;                // LDS-DMA requests per wave and iteration
    static_assert(WFR % kWaves == 0 && R >= 2 && (R - 2) * NPI < 64, "ring shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];      // [R][WFR] KB of weights, then [kWaves][R][AFR] KB of activations
    unsigned char *wring = ring, *aring = ring + R * WFR * 1024;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    // A zero-upsampled input (g.up > 1: the gradient of a strided convolution) is taken by PARITY CLASS: output pixels with
    // (ho % up, wo % up) = (ph, pw) = blockIdx.z see data only under the taps kh = kh0, kh0 + up, .. (kh0 = (pad - ph) mod up), kw likewise
    // -- 1, 2, 2 and 4 of a 3 x 3 kernel's 9 taps at up = 2, none for three of a 1 x 1 kernel's four classes (zeros, + add, masked).  A
    // class is a convolution of its own over its pixels; walking all taps with three quarters of the lanes on the zero line cost 4 x the
    // products (conv_fwd_kernel still does that for the shapes this kernel does not take).
    const int up = g.up, ph = (int)blockIdx.z / up, pw = (int)blockIdx.z % up;
    const int Hc = (g.Ho - ph + up - 1) / up, Wc = (g.Wo - pw + up - 1) / up;      // the class's pixels: ho = ph + up hc, wo = pw + up wc
    const long long P = (long long)g.N * Hc * Wc;
    const long long pix0 = (long long)blockIdx.x * (kWaves * 16 * PT) + wave * (16 * PT);
    if ((long long)blockIdx.x * (kWaves * 16 * PT) >= P) return;      // (whole workgroup; the grid is sized for class (0, 0), the largest)
    const int cpb = g.Cin / 64;                   // iterations per tap
    const int kh0 = ((g.pad - ph) % up + up) % up, kw0 = ((g.pad - pw) % up + up) % up;
    const int nkh = kh0 < g.KH ? (g.KH - kh0 + up - 1) / up : 0, nkw = kw0 < g.KW ? (g.KW - kw0 + up - 1) / up : 0;
    const int S = nkh * nkw * cpb;
    const int tiles_all = g.Cout / 16;
    const int co0 = blockIdx.y * (16 * CO_TILES);
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(g_conv_zero_line);

    long long img[PT], pout[PT];
    int hi0[PT], wi0[PT];
#pragma unroll
    for (int t3 = 0; t3 < PT; ++t3) {
        long long p = pix0 + 16 * t3 + c;
        const bool valid = p < P;
        if (!valid) p = P - 1;
        const int wo = (int)(p % Wc) * up + pw, ho = (int)((p / Wc) % Hc) * up + ph;
        const long long n = p / ((long long)Wc * Hc);
        img[t3] = n * g.H * g.W;
        hi0[t3] = ho * g.stride - g.pad;
        wi0[t3] = wo * g.stride - g.pad;
        pout[t3] = valid ? (n * g.Ho + ho) * g.Wo + wo : -1;
    }

    f32x4 acc[PT][CO_TILES];
#pragma unroll
    for (int t3 = 0; t3 < PT; ++t3)
#pragma unroll
        for (int t = 0; t < CO_TILES; ++t) acc[t3][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // request side: the iteration that is asked for next -- its tap, its channel block, this lane's sources
    RingRequest<CO_TILES, PT> rq;
    rq.step_bytes = (size_t)tiles_all * 1024;
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int f = wave * WPW + i, h = f / CO_TILES, t = f % CO_TILES;
        rq.wtap[i] = reinterpret_cast<const unsigned char *>(wpk) + ((size_t)blockIdx.y * CO_TILES + t) * 1024 + h * rq.step_bytes + lane * 16;
    }
    rq.it = rq.cb = 0;
    rq.kh = rq.kh0 = kh0;
    rq.kw = rq.kw0 = kw0;
    rq.kstep = up;
    if (S > 0) {      // (uniform; a class without taps: zeros through the epilogue)
        rq.tap_sources(x, g, cpb, img, hi0, wi0, q);
#pragma unroll
        for (int r = 0; r < R - 1; ++r) rq.template request<R>(r, wave, S, cpb, x, g, img, hi0, wi0, q, wring, aring, zero);
    }

    const unsigned wbase = (unsigned)(uintptr_t)wring + lane * 16, abase = (unsigned)(uintptr_t)aring + (wave * R * AFR) * 1024 + lane * 16;
    int slot = 0;
    for (int s = 0; s < S; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CONV_RING_ABLATE ? 0 : (R - 2) * NPI) : "memory");      // iteration s has landed (this wave's requests)
        __builtin_amdgcn_s_barrier();                                              // ... everybody's; and slot (s - 1) % R is free
        rq.template request<R>(slot == 0 ? R - 1 : slot - 1, wave, S, cpb, x, g, img, hi0, wi0, q, wring, aring, zero);
        const unsigned wa = wbase + slot * (WFR * 1024), ba = abase + slot * (AFR * 1024);
        u32x4 bfr[AFR], fr[8];
        ring_read<0, AFR>(bfr, ba);
        ring_read<0, (WFR < 8 ? WFR : 8)>(fr, wa);
        ring_products<0, CO_TILES, PT>(acc, fr, bfr, wa);
        slot = slot + 1 == R ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the repeated requests of the last iterations)
    conv_epilogue<CO_TILES, PT>(acc, pout, q, co0, g.Cout, scale, shift, residual, relu, mask, out);
}


// k x k pooling (stride s, padding p) on NHWC bf16: a thread owns 8 channels (16 bytes) of one output pixel.  MAX: maximum over the taps
// inside the image (torch's MaxPool2d with implicit -inf padding); else the mean over all k * k taps, which must lie inside the image
// (nn.AvgPool2d(k): stride k, no padding -- clip/model.py:24, :36, :115).
template <bool MAX>
__global__ __launch_bounds__(256) void pool_nhwc_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ out, int N, int H, int W, int C,
                                                        int Ho, int Wo, int k, int stride, int pad)
{
    const int cv = C / 8;
    const long long n = (long long)N * Ho * Wo * cv;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cv);
        long long p = i / cv;
        const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), b = (int)(p / ((long long)Wo * Ho));
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = MAX ? -__builtin_inff() : 0.f;
        for (int kh = 0; kh < k; ++kh) {
            const int hi = ho * stride + kh - pad;
            if (hi < 0 || hi >= H) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int wi = wo * stride + kw - pad;
                if (wi < 0 || wi >= W) continue;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(x + (((long long)b * H + hi) * W + wi) * C + 8 * c8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = bf16_lo(v[e]), hi2 = bf16_hi(v[e]);
                    acc[2 * e] = MAX ? fmaxf(acc[2 * e], lo) : acc[2 * e] + lo;
                    acc[2 * e + 1] = MAX ? fmaxf(acc[2 * e + 1], hi2) : acc[2 * e + 1] + hi2;
                }
            }
        }
        const float sc = MAX ? 1.f : 1.f / (float)(k * k);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf16(acc[2 * e] * sc, acc[2 * e + 1] * sc);
        *reinterpret_cast<u32x4 *>(out + p * C + 8 * c8) = o;
    }
}

// GroupNorm over NHWC bf16 with 8 channels per group (nn.GroupNorm(32, 256) of the input projections, richsem.py:301, :307): a pixel's
// group is one 16-byte vector.  Pass 1: sums of x and x^2 per (image, group) -- per-thread fp32 partials over a strip of pixels, folded
// per workgroup and added as doubles; pass 2: (x - mean) * rstd * gamma + beta to fp32 and / or bf16.
__global__ __launch_bounds__(256) void gn8_stats_kernel(const uint16_t *__restrict__ x, double *__restrict__ stats, int HW, int C)
{
    __shared__ float red[2][256];
    const int groups = C / 8;
    const int g = blockIdx.y, n = blockIdx.z;
    float s1 = 0.f, s2 = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(x + ((long long)n * HW + p) * C + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = bf16_lo(v[e]), b = bf16_hi(v[e]);
            s1 += a + b;
            s2 += a * a + b * b;
        }
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(stats + ((long long)n * groups + g) * 2, (double)red[0][0]);
        atomicAdd(stats + ((long long)n * groups + g) * 2 + 1, (double)red[1][0]);
    }
}

__global__ __launch_bounds__(256) void gn8_apply_kernel(const uint16_t *__restrict__ x, const double *__restrict__ stats,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta, float eps, int N,
                                                        int HW, int C, float *__restrict__ out_f32, uint16_t *__restrict__ out_bf16)
{
    const int groups = C / 8;
    const long long n_vec = (long long)N * HW * groups;
    const double cnt = (double)HW * 8.0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n_vec; i += gridDim.x * 256ll) {
        const int g = (int)(i % groups);
        const long long p = i / groups;
        const int n = (int)(p / HW);
        const double m = stats[((long long)n * groups + g) * 2] / cnt;
        const double var = stats[((long long)n * groups + g) * 2 + 1] / cnt - m * m;
        const float mean = (float)m, rstd = rsqrtf((float)(var > 0.0 ? var : 0.0) + eps);
        const u32x4 v = *reinterpret_cast<const u32x4 *>(x + p * C + 8 * g);
        const f32x4 g0 = *reinterpret_cast<const f32x4 *>(gamma + 8 * g), g1 = *reinterpret_cast<const f32x4 *>(gamma + 8 * g + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(beta + 8 * g), b1 = *reinterpret_cast<const f32x4 *>(beta + 8 * g + 4);
        float y[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[2 * e] = (bf16_lo(v[e]) - mean) * rstd;
            y[2 * e + 1] = (bf16_hi(v[e]) - mean) * rstd;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] = y[e] * g0[e] + b0[e];
            y[4 + e] = y[4 + e] * g1[e] + b1[e];
        }
        if (out_f32) {
            f32x4 *o = reinterpret_cast<f32x4 *>(out_f32 + p * C + 8 * g);
            o[0] = (f32x4){y[0], y[1], y[2], y[3]};
            o[1] = (f32x4){y[4], y[5], y[6], y[7]};
        }
        if (out_bf16)
            *reinterpret_cast<u32x4 *>(out_bf16 + p * C + 8 * g) =
                (u32x4){pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]), pack_bf16(y[4], y[5]), pack_bf16(y[6], y[7])};
    }
}

// GroupNorm(8 channels per group) backward, pass 1: per (image, group) the per-channel sums  sum_p dy  and  sum_p dy * xhat  (xhat from
// the forward's statistics), 16 values, added in fp64 to bstats[(n, g)][16]
__global__ __launch_bounds__(256) void gn8_bwd_stats_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                            const double *__restrict__ stats, float eps, double *__restrict__ bstats, int HW,
                                                            int C)
{
    const int groups = C / 8;
    const int g = blockIdx.y, n = blockIdx.z;
    const double cnt = (double)HW * 8.0;
    const double m = stats[((long long)n * groups + g) * 2] / cnt;
    const double var = stats[((long long)n * groups + g) * 2 + 1] / cnt - m * m;
    const float mean = (float)m, rstd = rsqrtf((float)(var > 0.0 ? var : 0.0) + eps);
    float s[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const long long at = ((long long)n * HW + p) * C + 8 * g;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(x + at), d = *reinterpret_cast<const u32x4 *>(dy + at);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d0 = bf16_lo(d[e]), d1 = bf16_hi(d[e]);
            s[2 * e] += d0;
            s[2 * e + 1] += d1;
            s[8 + 2 * e] += d0 * ((bf16_lo(v[e]) - mean) * rstd);
            s[8 + 2 * e + 1] += d1 * ((bf16_hi(v[e]) - mean) * rstd);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float t = s[e];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(bstats + ((long long)n * groups + g) * 16 + e, (double)t);
    }
}

// pass 2:  dx = rstd * (gamma dy - (A + xhat B) / m)  with  A = sum_c gamma_c sum_p dy,  B = sum_c gamma_c sum_p dy xhat  of the group;
// workgroup 0 also writes  dgamma_c = sum_n sum_p dy xhat,  dbeta_c = sum_n sum_p dy
__global__ __launch_bounds__(256) void gn8_bwd_apply_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                            const double *__restrict__ stats, const double *__restrict__ bstats,
                                                            const float *__restrict__ gamma, float eps, int N, int HW, int C,
                                                            uint16_t *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int groups = C / 8;
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += 256) {
            double a = 0.0, b = 0.0;
            for (int n = 0; n < N; ++n) {
                a += bstats[((long long)n * groups + c / 8) * 16 + (c & 7)];
                b += bstats[((long long)n * groups + c / 8) * 16 + 8 + (c & 7)];
            }
            if (dbeta) dbeta[c] = (float)a;
            if (dgamma) dgamma[c] = (float)b;
        }
    const long long n_vec = (long long)N * HW * groups;
    const double cnt = (double)HW * 8.0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n_vec; i += gridDim.x * 256ll) {
        const int g = (int)(i % groups);
        const long long p = i / groups;
        const int n = (int)(p / HW);
        const double m = stats[((long long)n * groups + g) * 2] / cnt;
        const double var = stats[((long long)n * groups + g) * 2 + 1] / cnt - m * m;
        const float mean = (float)m, rstd = rsqrtf((float)(var > 0.0 ? var : 0.0) + eps);
        const f32x4 g0 = *reinterpret_cast<const f32x4 *>(gamma + 8 * g), g1 = *reinterpret_cast<const f32x4 *>(gamma + 8 * g + 4);
        const double *bs = bstats + ((long long)n * groups + g) * 16;
        double A = 0.0, B = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            A += (double)g0[e] * bs[e] + (double)g1[e] * bs[4 + e];
            B += (double)g0[e] * bs[8 + e] + (double)g1[e] * bs[12 + e];
        }
        const float a = (float)(A / cnt), b = (float)(B / cnt);
        const u32x4 v = *reinterpret_cast<const u32x4 *>(x + p * C + 8 * g), d = *reinterpret_cast<const u32x4 *>(dy + p * C + 8 * g);
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ga = e < 2 ? g0[2 * e] : g1[2 * e - 4], gb = e < 2 ? g0[2 * e + 1] : g1[2 * e - 3];
            o[2 * e] = rstd * (ga * bf16_lo(d[e]) - a - (bf16_lo(v[e]) - mean) * rstd * b);
            o[2 * e + 1] = rstd * (gb * bf16_hi(d[e]) - a - (bf16_hi(v[e]) - mean) * rstd * b);
        }
        *reinterpret_cast<u32x4 *>(dx + p * C + 8 * g) =
            (u32x4){pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
    }
}

// epilogue of a k-split convolution: out = act(scale * ksum + shift (+ residual)), 8 channels per thread
__global__ __launch_bounds__(256) void conv_ksum_finish_kernel(const float *__restrict__ ksum, const float *__restrict__ scale,
                                                               const float *__restrict__ shift, const uint16_t *__restrict__ residual,
                                                               uint16_t *__restrict__ out, long long P, int Cout, int relu,
                                                               const uint16_t *__restrict__ mask)
{
    const int cv = Cout / 8;
    const long long n = P * cv;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
        const int co = 8 * (int)(i % cv);
        const long long at = (i / cv) * Cout + co;
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(ksum + at), a1 = *reinterpret_cast<const f32x4 *>(ksum + at + 4);
        float y[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = fmaf(y[e], scale ? scale[co + e] : 1.f, shift ? shift[co + e] : 0.f);
        if (residual) {
            const u32x4 r = *reinterpret_cast<const u32x4 *>(residual + at);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[2 * e] += bf16_lo(r[e]);
                y[2 * e + 1] += bf16_hi(r[e]);
            }
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = fmaxf(y[e], 0.f);
        }
        if (mask) {
            const u32x4 m = *reinterpret_cast<const u32x4 *>(mask + at);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if ((short)(m[e] & 0xFFFFu) <= 0) y[2 * e] = 0.f;
                if ((short)(m[e] >> 16) <= 0) y[2 * e + 1] = 0.f;
            }
        }
        *reinterpret_cast<u32x4 *>(out + at) = (u32x4){pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]), pack_bf16(y[4], y[5]), pack_bf16(y[6], y[7])};
    }
}

struct ConvArgs {
    const uint16_t *x, *wpk;
    const float *scale, *shift;
    const uint16_t *residual;
    uint16_t *out;
    ConvGeom g;
    int relu;
    hipStream_t stream;
    float *ksum = nullptr;      // zeroed (P, C_out) fp32 image for a k split, or nullptr
    int nz = 1;                 // k slices
    const uint16_t *mask = nullptr;      // input gradient through a ReLU: the forward's activation (result zeroed where it is not positive)
    int ring = 0;               // > 0: conv_ring_kernel with this many slots (C_in % 64 == 0, CO_TILES >= 2, PT <= 2, no k split)
};

template <int CO_TILES, int PT, int R>
int launch_ring(const ConvArgs &a, const dim3 &grid)
{
    constexpr int bytes = R * (2 * CO_TILES + kWaves * 2 * PT) * 1024;
    if constexpr (bytes <= 160 * 1024) {
        static std::atomic<int> prepared{0};
        if (bytes > 64 * 1024 && !prepared.load()) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_ring_kernel<CO_TILES, PT, R>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return (int)e;
            prepared = 1;
        }
        hipLaunchKernelGGL((conv_ring_kernel<CO_TILES, PT, R>), grid, dim3(kWaves * 64), bytes, a.stream, a.x, a.wpk, a.scale, a.shift,
                           a.residual, a.out, a.g, a.relu, a.mask);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? MSDA_OK : (int)e;
    } else {
        return MSDA_ERR_BAD_OPTION;
    }
}

template <int CO_TILES, int PT, int MODE>
int launch_conv(const ConvArgs &a)
{
    const long long P = (long long)a.g.N * a.g.Ho * a.g.Wo;
    const bool split = a.ksum != nullptr && a.nz > 1 && MODE == 2 && PT == 1;
    const dim3 grid((unsigned)((P + kWaves * 16 * PT - 1) / (kWaves * 16 * PT)), (unsigned)(a.g.Cout / (16 * CO_TILES)), split ? a.nz : 1);
    if constexpr (MODE == 2 && PT <= 2 && CO_TILES >= 2) {
        if (!split && a.ring > 0) {
            const int up = a.g.up;
            const long long Pc = (long long)a.g.N * ((a.g.Ho + up - 1) / up) * ((a.g.Wo + up - 1) / up);      // class (0, 0)'s pixels
            const dim3 grid((unsigned)((Pc + kWaves * 16 * PT - 1) / (kWaves * 16 * PT)), (unsigned)(a.g.Cout / (16 * CO_TILES)), (unsigned)(up * up));
            int ring = a.ring;
            while (ring > 3 && ring * (2 * CO_TILES + kWaves * 2 * PT) > 160) ring = ring > 4 ? 4 : 3;      // (what fits the CU's 160 KB)
            switch (ring) {
            case 3: return launch_ring<CO_TILES, PT, 3>(a, grid);
            case 4: return launch_ring<CO_TILES, PT, 4>(a, grid);
            default: return launch_ring<CO_TILES, PT, 6>(a, grid);
            }
        }
    }
    if constexpr (MODE == 2 && PT == 1) {
        if (split)
            hipLaunchKernelGGL((conv_fwd_kernel<CO_TILES, PT, MODE, true>), grid, dim3(kWaves * 64), 0, a.stream, a.x, a.wpk, a.scale, a.shift,
                               a.residual, a.out, a.g, a.relu, a.ksum, nullptr);
    }
    if (!split)
        hipLaunchKernelGGL((conv_fwd_kernel<CO_TILES, PT, MODE>), grid, dim3(kWaves * 64), 0, a.stream, a.x, a.wpk, a.scale, a.shift,
                           a.residual, a.out, a.g, a.relu, nullptr, a.mask);
    if (split) {
        const long long n = P * (a.g.Cout / 8);
        hipLaunchKernelGGL(conv_ksum_finish_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, a.stream,
                           a.ksum, a.scale, a.shift, a.residual, a.out, P, a.g.Cout, a.relu, a.mask);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

template <int CO_TILES, int MODE>
int launch_pt(const ConvArgs &a, int pt)
{
    switch (pt) {
    case 3: return launch_conv<CO_TILES, 3, MODE>(a);
    case 2: return launch_conv<CO_TILES, 2, MODE>(a);
    default: return launch_conv<CO_TILES, 1, MODE>(a);
    }
}

template <int MODE>
int launch_ct(const ConvArgs &a, int ct, int pt)
{
    switch (ct) {
    case 16: return launch_pt<16, MODE>(a, pt);
    case 8: return launch_pt<8, MODE>(a, pt);
    case 4: return launch_pt<4, MODE>(a, pt);
    case 2: return launch_pt<2, MODE>(a, pt);
    default: return launch_pt<1, MODE>(a, pt);
    }
}

std::atomic<int> g_force_ct{0}, g_force_pt{0};   // tuning: msda_conv_set_tiling
std::atomic<int> g_ring{0};                       // msda_conv_set_ring: -1 never, 0 automatic (choose_ring), 3 / 4 / 6 slots wherever the kernel applies

// Tile choice (tools/time_conv.py --sweep on MI355X, ResNet-50 shapes at 2 x 800 x 1344): the kernel is bound by latency, not by operand
// re-use -- small tiles (more waves per SIMD, more workgroups) win almost everywhere: 16 pixels per wave, 128 output channels per wave
// while that still gives two workgroups per CU, else 64; only wide, shallow 3 x 3 layers (thousands of workgroups) take 48 pixels.
void choose_tiling(long long P, int Cout, int K, int &ct, int &pt)
{
    const int tiles = Cout / 16, G = conv_group(Cout);
    auto n_wg = [&](int c, int p) { return ((P + 64 * p - 1) / (64 * p)) * (tiles / c); };
    if (G < 4) {
        ct = G;
    } else {
        ct = (tiles % 8 == 0 && n_wg(8, 1) >= 512) ? 8 : 4;
    }
    pt = (K >= 288 && n_wg(ct, 1) > 2048) ? 3 : 1;
}

// k slices for a problem at the chosen tiling: only where the (pixel, channel) grid leaves most of the chip idle and the k loop is
// long -- about three workgroups per CU in all, at least 8 iterations per slice (0 / 1: no split)
int choose_ksplit(long long P, int Cout, int Cin, int KH, int KW, int ct, int pt)
{
    if (Cin % 64 != 0 || pt != 1) return 1;      // (the kernel's k-slice epilogue: two k-steps per barrier, one pixel tile per wave)
    const long long wgs = ((P + 64 * pt - 1) / (64 * pt)) * (Cout / (16 * ct));
    const int S = KH * KW * (Cin / 64);
    if (wgs > 64 || S < 64) return 1;      // (measured: with 264 or 528 workgroups -- layer4's / layer3's shapes -- the split only adds its zero-fill, atomics and second launch)
    long long nz = (768 + wgs / 2) / wgs;
    if (nz > S / 8) nz = S / 8;
    if (nz > 16) nz = 16;
    return nz < 2 ? 1 : (int)nz;
}

// Operand rings for a problem (0: conv_fwd_kernel), and the tile that goes with them.  Measured per shape (tools/time_conv.py --ring,
// MI355X, 2 x 800 x 1344): the ring wins where a SIMD gets one or two waves and the k loop is long -- layer3's / layer4's 3 x 3 and
// channel-reducing 1 x 1 convolutions (26.6 against 34.4, 15.7 / 18.6, 32.6 / 54.0, 16.7 / 31.8 us) and every strided input gradient
// (three quarters of whose lanes fetch the zero line) -- and loses where thousands of workgroups hide the latency by themselves (layer1,
// layer2) or the loop is 4-8 iterations (256 -> 1024, 512 -> 2048).  Three slots: deeper rings cost occupancy and gain nothing.
int choose_ring(long long P, int Cout, int Cin, int KH, int KW, int up, bool forced_tile, int &ct, int &pt)
{
    const int r = g_ring.load();
    if (r < 0 || Cin % 64 != 0) return 0;
    const int tiles = Cout / 16;
    if (r == 0) {
        if ((long long)KH * KW * Cin < 1024 || (P > 16384 && up == 1)) return 0;
        if (!forced_tile) {
            ct = tiles % 8 == 0 ? 8 : (tiles % 4 == 0 ? 4 : conv_group(Cout));
            pt = (P >= 4096 && up == 1) ? 2 : 1;
        }
    }
    if (ct < 2 || pt > 2) return 0;
    return r > 0 ? r : 3;
}

}  // namespace

extern "C" {

/* Tuning / tests: the operand rings of conv_ring_kernel: -1 never, 0 automatic, 3 / 4 / 6 slots wherever the kernel applies. */
int msda_conv_set_ring(int slots)
{
    if (slots != -1 && slots != 0 && slots != 3 && slots != 4 && slots != 6) return msda_note_error(MSDA_ERR_BAD_OPTION, __func__);
    g_ring = slots;
    return MSDA_OK;
}

/* Tuning / tests: force the tile shape (channel tiles per wave in {1, 2, 4, 8, 16}, pixel tiles per wave in {1, 2, 3}); 0 = automatic. */
int msda_conv_set_tiling(int co_tiles, int pixel_tiles)
{
    if ((co_tiles != 0 && co_tiles != 1 && co_tiles != 2 && co_tiles != 4 && co_tiles != 8 && co_tiles != 16) || pixel_tiles < 0 ||
        pixel_tiles > 3)
        return msda_note_error(MSDA_ERR_BAD_OPTION, __func__);
    g_force_ct = co_tiles;
    g_force_pt = pixel_tiles;
    return MSDA_OK;
}

int msda_conv_packed_elems(int Cout, int Cin, int KH, int KW, int64_t *elems)
{
    if (!elems) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (Cout < 16 || Cout % 16 != 0 || Cin < 1 || KH < 1 || KW < 1 || KH > 16 || KW > 16) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    const int64_t K = (int64_t)KH * KW * Cin;
    if (Cin % 32 != 0 && K > 512) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);      // few-channel inputs only (table of 512 entries)
    *elems = (int64_t)Cout * ((K + 31) / 32 * 32);
    return MSDA_OK;
}

int msda_conv_pack_weight(const float *weight, int Cout, int Cin, int KH, int KW, uint16_t *packed, msda_stream_t stream)
{
    if (!weight || !packed) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    int64_t n = 0;
    const int rc = msda_conv_packed_elems(Cout, Cin, KH, KW, &n);
    if (rc != MSDA_OK) return rc;
    hipLaunchKernelGGL(conv_pack_kernel, dim3(512), dim3(256), 0, static_cast<hipStream_t>(stream), weight, packed, Cout, Cin, KH, KW,
                       (int)(n / Cout));
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

/* bytes of zeroed fp32 workspace with which msda_conv_forward_ws_bf16 splits the k loop of this problem (0: it does not) */
int msda_conv_forward_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t *bytes)
{
    if (!bytes) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    int64_t n = 0;
    const int rc = msda_conv_packed_elems(Cout, Cin, KH, KW, &n);
    if (rc != MSDA_OK) return rc;
    if (N < 1 || H < 1 || W < 1 || stride < 1 || pad < 0) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho < 1 || Wo < 1) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    int ct, pt;
    choose_tiling((long long)N * Ho * Wo, Cout, KH * KW * Cin, ct, pt);
    *bytes = choose_ksplit((long long)N * Ho * Wo, Cout, Cin, KH, KW, ct, pt) > 1 ? (int64_t)N * Ho * Wo * Cout * (int64_t)sizeof(float) : 0;
    return MSDA_OK;
}

int msda_conv_forward_bf16(const uint16_t *x, const uint16_t *packed_weight, const float *scale, const float *shift,
                           const uint16_t *residual, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu,
                           uint16_t *out, msda_stream_t stream)
{
    return msda_conv_forward_ws_bf16(x, packed_weight, scale, shift, residual, N, H, W, Cin, Cout, KH, KW, stride, pad, relu, out, nullptr,
                                     stream);
}

/* msda_conv_forward_bf16 with an optional workspace: `workspace` = msda_conv_forward_workspace_bytes() bytes of ZEROED device memory
 * (the k loop is then split over several workgroups that add their sums there, and a second kernel applies the epilogue), or NULL */
int msda_conv_forward_ws_bf16(const uint16_t *x, const uint16_t *packed_weight, const float *scale, const float *shift,
                              const uint16_t *residual, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu,
                              uint16_t *out, void *workspace, msda_stream_t stream)
{
    if (!x || !packed_weight || !scale || !shift || !out) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    int64_t n = 0;
    const int rc = msda_conv_packed_elems(Cout, Cin, KH, KW, &n);
    if (rc != MSDA_OK) return rc;
    if (N < 1 || H < 1 || W < 1 || stride < 1 || pad < 0) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho < 1 || Wo < 1) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((long long)N * H * W * Cin >= (1ll << 40) || (long long)N * Ho * Wo * Cout >= (1ll << 40)) return msda_note_error(MSDA_ERR_TOO_LARGE, __func__);
    const bool small_c = Cin % 32 != 0;
    if ((reinterpret_cast<uintptr_t>(packed_weight) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) |
         reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(residual) | (small_c ? 0 : reinterpret_cast<uintptr_t>(x))) & 15)
        return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    int ct, pt;
    choose_tiling((long long)N * Ho * Wo, Cout, KH * KW * Cin, ct, pt);
    const int fct = g_force_ct.load(), fpt = g_force_pt.load();
    if (fct && (Cout / 16) % fct == 0 && fct % conv_group(Cout) == 0) ct = fct;
    if (fpt) pt = fpt;
    ConvArgs a{x, packed_weight, scale, shift, residual, out, ConvGeom{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, 1}, relu,
               static_cast<hipStream_t>(stream)};
    if (workspace && !fct && !fpt) {
        if (reinterpret_cast<uintptr_t>(workspace) & 15) return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
        a.nz = choose_ksplit((long long)N * Ho * Wo, Cout, Cin, KH, KW, ct, pt);
        a.ksum = a.nz > 1 ? static_cast<float *>(workspace) : nullptr;
    }
    if (!a.ksum) a.ring = choose_ring((long long)N * Ho * Wo, Cout, Cin, KH, KW, 1, fct || fpt, ct, pt);      // (a k split where the grid is tiny)
    if (small_c) return launch_ct<0>(a, ct, pt);
    return Cin % 64 == 0 ? launch_ct<2>(a, ct, pt) : launch_ct<1>(a, ct, pt);
}

/* GroupNorm with 8 channels per group on NHWC bf16 (nn.GroupNorm(32, 256) of the input projections, models/richsem/richsem.py:301, :307):
 * x (N, H W, C) bf16, C = 8 * groups; gamma, beta (C) f32; stats: N * groups * 2 doubles of scratch (zeroed here); out_f32 and / or
 * out_bf16 (N, H W, C), either may be NULL.  Forward only. */
int msda_groupnorm8_nhwc_bf16(const uint16_t *x, const float *gamma, const float *beta, float eps, int N, int HW, int C, double *stats,
                              float *out_f32, uint16_t *out_bf16, msda_stream_t stream)
{
    if (!x || !gamma || !beta || !stats || (!out_f32 && !out_bf16)) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (N < 1 || HW < 1 || C < 8 || C % 8 != 0 || C / 8 > 65535 || N > 65535) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
         reinterpret_cast<uintptr_t>(out_f32) | reinterpret_cast<uintptr_t>(out_bf16)) & 15)
        return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int groups = C / 8;
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * 2 * N * groups, st);
    if (e != hipSuccess) return (int)e;
    int strips = (HW + 2047) / 2048;
    if (strips > 64) strips = 64;
    hipLaunchKernelGGL(gn8_stats_kernel, dim3(strips, groups, N), dim3(256), 0, st, x, stats, HW, C);
    const long long n_vec = (long long)N * HW * groups;
    const int grid = (int)((n_vec + 255) / 256 < 65536 ? (n_vec + 255) / 256 : 65536);
    hipLaunchKernelGGL(gn8_apply_kernel, dim3(grid), dim3(256), 0, st, x, stats, gamma, beta, eps, N, HW, C, out_f32, out_bf16);
    e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

/* Backward of msda_groupnorm8_nhwc_bf16: x, dy (N, HW, C) bf16; stats as the forward left them; bstats: N * (C / 8) * 16 doubles of
 * device scratch (zeroed here); dx (N, HW, C) bf16; dgamma, dbeta (C) f32 (either may be NULL). */
int msda_groupnorm8_backward_nhwc_bf16(const uint16_t *x, const uint16_t *dy, const float *gamma, float eps, int N, int HW, int C,
                                       const double *stats, double *bstats, uint16_t *dx, float *dgamma, float *dbeta, msda_stream_t stream)
{
    if (!x || !dy || !gamma || !stats || !bstats || !dx) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (N < 1 || HW < 1 || C < 8 || C % 8 != 0 || C / 8 > 65535 || N > 65535) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(dx)) & 15)
        return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int groups = C / 8;
    hipError_t e = hipMemsetAsync(bstats, 0, sizeof(double) * 16 * N * groups, st);
    if (e != hipSuccess) return (int)e;
    int strips = (HW + 2047) / 2048;
    if (strips > 64) strips = 64;
    hipLaunchKernelGGL(gn8_bwd_stats_kernel, dim3(strips, groups, N), dim3(256), 0, st, x, dy, stats, eps, bstats, HW, C);
    const long long n_vec = (long long)N * HW * groups;
    const int grid = (int)((n_vec + 255) / 256 < 65536 ? (n_vec + 255) / 256 : 65536);
    hipLaunchKernelGGL(gn8_bwd_apply_kernel, dim3(grid), dim3(256), 0, st, x, dy, stats, bstats, gamma, eps, N, HW, C, dx, dgamma, dbeta);
    e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

/* Pooling on NHWC bf16 (the pools around the backbones' convolutions: nn.AvgPool2d(k) of the CLIP ResNet, clip/model.py:24, :36, :115;
 * torchvision's MaxPool2d(3, 2, 1) after the ResNet stem).  is_max = 0: mean over k x k, stride, padding 0 (H, W multiples are not
 * required: Ho = (H - k) / stride + 1); is_max = 1: maximum with implicit -inf padding `pad`.  C % 8 == 0.  Forward only. */
int msda_pool_nhwc_bf16(const uint16_t *x, int N, int H, int W, int C, int k, int stride, int pad, int is_max, uint16_t *out,
                        msda_stream_t stream)
{
    if (!x || !out) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (N < 1 || H < 1 || W < 1 || C < 8 || C % 8 != 0 || k < 1 || stride < 1 || pad < 0 || (!is_max && pad != 0) || pad >= k)
        return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (Ho < 1 || Wo < 1) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    const long long n = (long long)N * Ho * Wo * (C / 8);
    const int grid = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (is_max)
        hipLaunchKernelGGL(pool_nhwc_kernel<true>, dim3(grid), dim3(256), 0, st, x, out, N, H, W, C, Ho, Wo, k, stride, pad);
    else
        hipLaunchKernelGGL(pool_nhwc_kernel<false>, dim3(grid), dim3(256), 0, st, x, out, N, H, W, C, Ho, Wo, k, stride, pad);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

/* Gradient of msda_conv_forward_bf16 w.r.t. its input, by the same kernel: a stride-1 convolution of the (zero-upsampled, for a
 * strided forward) output gradient with the flipped, transposed weight.  dy (N, Ho, Wo, Cout) bf16; packed_weight_t =
 * msda_conv_pack_weight of w_t[ci][co][kh][kw] = w[co][ci][KH-1-kh][KW-1-kw] (times the forward's scale[co] if the gradient is taken
 * before the affine); dx (N, H, W, Cin) bf16, every element written.  Cout % 32 == 0 (it is the k dimension here), Cin % 16 == 0. */
int msda_conv_dgrad_bf16(const uint16_t *dy, const uint16_t *packed_weight_t, int N, int Ho, int Wo, int Cout, int Cin, int KH, int KW,
                         int stride, int pad, int H, int W, uint16_t *dx, msda_stream_t stream)
{
    return msda_conv_dgrad_ws_bf16(dy, packed_weight_t, N, Ho, Wo, Cout, Cin, KH, KW, stride, pad, H, W, dx, nullptr, stream);
}

/* bytes of zeroed fp32 workspace with which msda_conv_dgrad_ws_bf16 splits the k loop (k = KH KW C_out here) of this problem (0: none) */
int msda_conv_dgrad_workspace_bytes(int N, int Ho, int Wo, int Cout, int Cin, int KH, int KW, int stride, int pad, int H, int W, int64_t *bytes)
{
    if (!bytes) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (N < 1 || Ho < 1 || Wo < 1 || H < 1 || W < 1 || Cout < 32 || Cout % 32 != 0 || Cin < 16 || Cin % 16 != 0 || KH < 1 || KW < 1 ||
        KH > 16 || KW > 16 || stride < 1 || pad < 0)
        return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    int ct, pt;
    choose_tiling((long long)N * H * W, Cin, KH * KW * Cout, ct, pt);
    *bytes = choose_ksplit((long long)N * H * W, Cin, Cout, KH, KW, ct, pt) > 1 ? (int64_t)N * H * W * Cin * (int64_t)sizeof(float) : 0;
    return MSDA_OK;
}

/* msda_conv_dgrad_bf16 with an optional ZEROED workspace of msda_conv_dgrad_workspace_bytes() bytes (k split), or NULL */
int msda_conv_dgrad_ws_bf16(const uint16_t *dy, const uint16_t *packed_weight_t, int N, int Ho, int Wo, int Cout, int Cin, int KH, int KW,
                            int stride, int pad, int H, int W, uint16_t *dx, void *workspace, msda_stream_t stream)
{
    return msda_conv_dgrad_fused_bf16(dy, packed_weight_t, N, Ho, Wo, Cout, Cin, KH, KW, stride, pad, H, W, nullptr, nullptr, dx, workspace,
                                      stream);
}

/* msda_conv_dgrad_ws_bf16 with the element-wise work that follows an input gradient in a residual network done in its epilogue:
 *   dx = mask(conv_dgrad(dy) + add)
 * `add` (N, H, W, Cin) bf16 or NULL: a second gradient of the same tensor (the identity branch of a bottleneck, torchvision's
 * Bottleneck.forward `out += identity`); `relu_out` (N, H, W, Cin) bf16 or NULL: the tensor whose gradient this is, when it is the OUTPUT
 * OF A RELU -- dx is zeroed where it is not positive, i.e. dx is the gradient at the ReLU's input (aten::threshold_backward). */
int msda_conv_dgrad_fused_bf16(const uint16_t *dy, const uint16_t *packed_weight_t, int N, int Ho, int Wo, int Cout, int Cin, int KH, int KW,
                               int stride, int pad, int H, int W, const uint16_t *add, const uint16_t *relu_out, uint16_t *dx,
                               void *workspace, msda_stream_t stream)
{
    if (!dy || !packed_weight_t || !dx) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (N < 1 || Ho < 1 || Wo < 1 || H < 1 || W < 1 || Cout < 32 || Cout % 32 != 0 || Cin < 16 || Cin % 16 != 0 || KH < 1 || KW < 1 ||
        KH > 16 || KW > 16 || stride < 1 || pad < 0 || pad > KH - 1 || pad > KW - 1)
        return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((H + 2 * pad - KH) / stride + 1 != Ho || (W + 2 * pad - KW) / stride + 1 != Wo) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((long long)N * H * W * Cin >= (1ll << 40) || (long long)N * Ho * Wo * Cout >= (1ll << 40)) return msda_note_error(MSDA_ERR_TOO_LARGE, __func__);
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(packed_weight_t) | reinterpret_cast<uintptr_t>(dx) |
         reinterpret_cast<uintptr_t>(add) | reinterpret_cast<uintptr_t>(relu_out)) & 15)
        return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    // as a forward call: input dy with Cout channels, output dx with Cin channels and H x W pixels, padding KH - 1 - pad
    // (KH == KW is not required: the column padding is KW - 1 - pad, see below), virtual input upsampled by `stride`
    if (KH - 1 - pad != KW - 1 - pad) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);      // one padding value in the kernel's geometry: square kernels
    int ct, pt;
    choose_tiling((long long)N * H * W, Cin, KH * KW * Cout, ct, pt);
    const int fct = g_force_ct.load(), fpt = g_force_pt.load();
    if (fct && (Cin / 16) % fct == 0 && fct % conv_group(Cin) == 0) ct = fct;
    if (fpt) pt = fpt;
    ConvArgs a{dy, packed_weight_t, nullptr, nullptr, add, dx, ConvGeom{N, Ho, Wo, Cout, H, W, Cin, KH, KW, 1, KH - 1 - pad, stride},
               0, static_cast<hipStream_t>(stream)};
    a.mask = relu_out;
    if (workspace) {
        if (reinterpret_cast<uintptr_t>(workspace) & 15) return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
        a.nz = choose_ksplit((long long)N * H * W, Cin, Cout, KH, KW, ct, pt);
        a.ksum = a.nz > 1 ? static_cast<float *>(workspace) : nullptr;
    }
    if (!a.ksum) a.ring = choose_ring((long long)N * H * W, Cin, Cout, KH, KW, stride, fct || fpt, ct, pt);
    return Cout % 64 == 0 ? launch_ct<2>(a, ct, pt) : launch_ct<1>(a, ct, pt);
}

}  // extern "C"
