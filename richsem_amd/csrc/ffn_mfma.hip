// ffn_mfma.hip -- the transformer layers' feed-forward block as ONE gfx950 MFMA kernel (bf16 storage, fp32 accumulation):
//
//     out = LayerNorm(x + W2 . relu(W1 . x + b1) + b2)            reference: models/richsem/deformable_transformer.py:862-866
//                                                                  (encoder forward_ffn) and :940-944 (decoder forward_ffn)
//
// RichSem: d_model = 256, d_ffn = 2048; the encoder calls it on N*S = 44646 tokens per layer: 93.6 GFLOP, the largest dense
// contraction of the transformer.  The hidden activation (44646 x 2048: 183 MB in bf16) never
// leaves the chip.
//
// Everything is computed TRANSPOSED, so that tokens sit on the lanes and channels in the registers:
//     H^T   (hidden x tokens) = W1 (hidden x 256)  . x^T        A operand = nn.Linear's own row-major weight, B = x rows
//     out^T (256 x tokens)   += W2 (256 x hidden)  . relu(H^T)   A operand = nn.Linear's own weight again
// with mfma_f32_16x16x32_bf16.  An accumulator tile has its column (token) on the lane (l & 15) and rows 4 (l >> 4) + reg in its
// four registers, and the second product sums over exactly those rows -- so relu(H^T) of two stacked 16-row tiles, converted
// to bf16, IS the B operand (k = 32) of the second product: no LDS round trip, no lane movement (guide: "an accumulator
// tile as the next MFMA's operand").  The k order of such an operand is permuted -- lane group q holds hidden units 4 q .. 4 q + 3
// and 16 + 4 q .. 16 + 4 q + 3 of the tile -- so W2 is repacked once (msda_ffn_pack_w2_bf16) and then read with plain 16-byte
// fragments.  The LayerNorm runs over registers: a lane holds 64 of its token's 256 channels, lanes l ^ 16, l ^ 32, l ^ 48 the rest.
//
// Work decomposition: a wave owns 48 tokens (three MFMA column tiles: every weight fragment it reads feeds three MFMAs;
// 44646 tokens = 931 waves for the chip's 1024 SIMDs), a workgroup is four waves -- one per SIMD: 192 accumulators for out^T,
// 96 registers for the wave's x fragments, 24 for the hidden tile -- and has its CU to itself.  The weights stream through
// LDS in tiles of 32 hidden units (16 KB of W1 + 16 KB of W2), a ring of three, by LDS DMA (global_load_lds_dwordx4: no
// registers) straight into FRAGMENT ORDER -- fragment f is 1 KB, lane l's 16 bytes at l*16 -- so every operand read is a
// conflict-free ds_read_b128.
// The pipeline is hand-scheduled where hipcc would drain it: tiles are requested TWO ahead and waited for with a partial
// vmcnt before a bare s_barrier (a __syncthreads, or any LDS read the compiler can see, waits for every LDS DMA in flight),
// and the operand reads are inline ds_read_b128 seven fragments ahead of the MFMAs that use them, released by partial lgkmcnt
// waits (a wave has its SIMD to itself, so nothing else hides the LDS latency).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include <atomic>
#include <initializer_list>

#include "../../include/richsem_msda.h"

extern "C" int msda_note_error(int code, const char *entry);      // msda_api.hip: sets msda_last_error()

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kD = 256;            // d_model (fixed: 8 output row tiles, 16 k-steps)
constexpr int kTokWave = 48;       // tokens per wave (three 16-token MFMA column tiles)
constexpr int kWaves = 4;
constexpr int kRing = 3;           // weight tiles in LDS
constexpr int kTokWg = kTokWave * kWaves;
constexpr int kHT = 32;            // hidden units per weight tile
constexpr int kFragShorts = 512;   // one MFMA operand fragment: 64 lanes x 8 bf16
constexpr int kTileFrags = 32;     // 16 of W1 (k-steps over d_model) + 16 of W2 (8 row tiles x 2 k-steps)
constexpr int kMaxFfn = 4096;

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b)   // one v_cvt_pk_bf16_f32 (round to nearest even)
{
    const bf16x2_t p = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }   // one v_med3_f32
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

// W2 (256 x F, row-major) -> the k order the accumulator-as-operand idiom needs: inside every 32 hidden columns, position
// 8 q + j holds column 4 q + j (j < 4) or 16 + 4 q + (j - 4) (j >= 4).
__global__ void pack_w2_kernel(const uint16_t *__restrict__ w2, uint16_t *__restrict__ w2p, long long n)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i & 31), q = p >> 3, j = p & 7;
        const int src = j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4);
        w2p[i] = w2[(i & ~31ll) + src];
    }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one operand fragment from LDS, not visible to the compiler's wait-count bookkeeping (see the header)
#define FFN_READ(dst, addr, byte_off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(byte_off))
// "at most n LDS reads still in flight": everything older has arrived.  The operand ties the fragment to the wait.
#define FFN_WAIT(n, a) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(n))

__global__ __launch_bounds__(kWaves * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ffn_fwd_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w1, const float *__restrict__ b1,
                    const uint16_t *__restrict__ w2p, const float *__restrict__ b2, const float *__restrict__ gamma,
                    const float *__restrict__ beta, float eps, int T, int F, uint16_t *__restrict__ out,
                    unsigned long long *__restrict__ stamps, float *__restrict__ rstd_out, uint16_t *__restrict__ yhat_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    short *wbuf = reinterpret_cast<short *>(smem);                                          // [kRing][kTileFrags * kFragShorts]
    float *lb1 = reinterpret_cast<float *>(smem + kRing * kTileFrags * kFragShorts * 2);    // [F]
    float *lb2 = lb1 + F, *lgam = lb2 + kD, *lbet = lgam + kD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;   // MFMA column (token) / lane group
    const int tok0 = blockIdx.x * kTokWg + wave * kTokWave;
    const int nt = F / kHT;

    // biases and LayerNorm parameters -> LDS (ordinary loads: all of them retire before the first LDS DMA is issued)
    for (int i = tid; i < F; i += kWaves * 64) lb1[i] = b1[i];
    for (int i = tid; i < kD; i += kWaves * 64) { lb2[i] = b2[i]; lgam[i] = gamma[i]; lbet[i] = beta[i]; }

    // this wave's x fragments (B operand of the first product): lane (c, q) holds x[token c][32 s + 8 q + 0..7]
    bf16x8 xf[3][8];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        const int tok = min(tok0 + 16 * ct + c, T - 1);
        const uint16_t *row = x + (size_t)tok * kD + 8 * q;
#pragma unroll
        for (int s = 0; s < 8; ++s) xf[ct][s] = *reinterpret_cast<const bf16x8 *>(row + 32 * s);
    }

    f32x4 acc[3][16];   // out^T: [token tile][16-channel row tile]; lane (c, q) holds channels 16 t + 4 q + 0..3 of token c
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[ct][t][i] = 0.f;

    // Weight tile `ht` -> ring slot, in fragment order (lane (r, q) of a fragment = row r, k = 8 q ..).  W1 fragment 2 s + rt =
    // (k-step s, 16-row tile rt), W2 fragment 16 + t = 16-channel row tile t.  Wave w brings k-steps 2 w, 2 w + 1 of W1 and row
    // tiles 4 w .. 4 w + 3 of W2: 8 DMAs of 1 KB per wave and tile.
    const uint16_t *p1 = w1 + (size_t)c * kD + 8 * q + 64 * wave;
    const uint16_t *p2 = w2p + (size_t)(64 * wave + c) * F + 8 * q;
    auto stage_piece = [&](int ht, int slot, int i) {   // piece i of 8 (i < 4: W1, else W2)
        short *dst = wbuf + slot * (kTileFrags * kFragShorts);
        if (i < 4)   // i = 2 (s - 2 w) + rt
            __builtin_amdgcn_global_load_lds(p1 + (size_t)ht * (kHT * kD) + (size_t)(i & 1) * 16 * kD + 32 * (i >> 1),
                                             reinterpret_cast<__attribute__((address_space(3))) void *>(
                                                 reinterpret_cast<uintptr_t>(dst + (4 * wave + i) * kFragShorts)), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds(p2 + ht * kHT + (size_t)(i - 4) * 16 * F,
                                             reinterpret_cast<__attribute__((address_space(3))) void *>(
                                                 reinterpret_cast<uintptr_t>(dst + (16 + 4 * wave + (i - 4)) * kFragShorts)), 16, 0, 0);
    };
    auto stage = [&](int ht, int slot) {
#pragma unroll
        for (int i = 0; i < 8; ++i) stage_piece(ht, slot, i);
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the ordinary loads above have retired
    __syncthreads();                      // (parameters in LDS)
    stage(0, 0);
    if (nt > 1) stage(1, 1);

    // diagnostic (msda_ffn_debug_stamps): shader-clock totals of the loop's stages, wave 0 of every workgroup
    unsigned long long t_last = 0, t_acc[6] = {0, 0, 0, 0, 0, 0};
#define FFN_STAMP(i)                                               \
    if (stamps) {                                                  \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        t_acc[i] += now_ - t_last;                                 \
        t_last = now_;                                             \
    }
    if (stamps) t_last = __builtin_amdgcn_s_memtime();
    int slot = 0;
    for (int ht = 0; ht < nt; ++ht) {
        // tile ht has landed: at most the 8 DMAs of tile ht + 1 may still be in flight
        if (ht + 1 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FFN_STAMP(0)   // wait for the tile's DMAs
        __builtin_amdgcn_s_barrier();   // ... everybody's share of it; and everybody is done with tile ht - 1, whose slot is refilled now
        FFN_STAMP(1)   // barrier
        // the 8 pieces of tile ht + 2 are requested one by one between the MFMA groups below (an LDS-DMA instruction costs its wave
        // 60-185 cycles of issue depending on what surrounds it: guide, "LDS-DMA piece issue cost")
        const bool refill = ht + 2 < nt;
        const int rslot = slot == 0 ? 2 : slot - 1;
#define FFN_STAGE(I) if (refill) stage_piece(ht + 2, rslot, I);
        const unsigned wt = (unsigned)(uintptr_t)(wbuf + slot * (kTileFrags * kFragShorts)) + lane * 16;   // LDS byte address of this lane's piece of fragment 0

        // The operand stream of a tile: a ring of eight fragment registers; fragment p lives in fr[p & 7], is waited for with
        // "at most 7 newer reads in flight", feeds three MFMAs and is at once replaced by the read of fragment p + 8.
        u32x4 fr[8];
        f32x4 bias[2];   // rows 16 rt + 4 q + 0..3 of the tile: read in the same stream, ahead of fragment 0 (whose wait covers them)
        {
            const unsigned ba = (unsigned)(uintptr_t)lb1 + (unsigned)(ht * kHT + 4 * q) * 4u;
            FFN_READ(bias[0], ba, 0);
            FFN_READ(bias[1], ba, 64);
        }
        FFN_READ(fr[0], wt, 0 * 1024); FFN_READ(fr[1], wt, 1 * 1024); FFN_READ(fr[2], wt, 2 * 1024); FFN_READ(fr[3], wt, 3 * 1024);
        FFN_READ(fr[4], wt, 4 * 1024); FFN_READ(fr[5], wt, 5 * 1024); FFN_READ(fr[6], wt, 6 * 1024); FFN_READ(fr[7], wt, 7 * 1024);

        // ---- H^T tile (two 16-row tiles) = W1 tile . x^T + b1 ---------------------------------------------------------------
        // The MFMAs of this product are inline: their accumulators must be VGPRs (relu and the conversion read them; hipcc
        // would put them into AGPRs, on top of six out^T tiles that it then parks in VGPRs around every hidden tile: ~100 register
        // moves per tile), and the first k-step takes the bias as its C operand instead of copying it into the accumulators.
        f32x4 hacc[3][2];
        bf16x8 hb[3];
#define FFN_MFMA1(D, A, B, C) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=v"(D) : "v"(A), "v"(B), "v"(C))
#define FFN_MFMA1_ACC(D, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(D) : "v"(A), "v"(B))
#define FFN_USE1(P)                                                                                                          \
    {                                                                                                                        \
        constexpr int s_ = (P) >> 1, rt_ = (P) & 1;                                                                          \
        if (s_ == 0) {                                                                                                       \
            FFN_MFMA1(hacc[0][rt_], fr[(P) & 7], xf[0][s_], bias[rt_]);                                                      \
            FFN_MFMA1(hacc[1][rt_], fr[(P) & 7], xf[1][s_], bias[rt_]);                                                      \
            FFN_MFMA1(hacc[2][rt_], fr[(P) & 7], xf[2][s_], bias[rt_]);                                                      \
        } else {                                                                                                             \
            FFN_MFMA1_ACC(hacc[0][rt_], fr[(P) & 7], xf[0][s_]);                                                             \
            FFN_MFMA1_ACC(hacc[1][rt_], fr[(P) & 7], xf[1][s_]);                                                             \
            FFN_MFMA1_ACC(hacc[2][rt_], fr[(P) & 7], xf[2][s_]);                                                             \
        }                                                                                                                    \
    }
#define FFN_USE2(P)                                                                                                          \
    {                                                                                                                        \
        const bf16x8 a_ = __builtin_bit_cast(bf16x8, fr[(P) & 7]);                                                           \
        constexpr int t_ = (P) - 16;                                                                                         \
        acc[0][t_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, hb[0], acc[0][t_], 0, 0, 0);                                \
        acc[1][t_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, hb[1], acc[1][t_], 0, 0, 0);                                \
        acc[2][t_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, hb[2], acc[2][t_], 0, 0, 0);                                \
    }
        FFN_WAIT(7, fr[0]);
        asm volatile("" : "+v"(bias[0]), "+v"(bias[1]));
        FFN_USE1(0) FFN_READ(fr[0], wt, 8 * 1024);
        FFN_WAIT(7, fr[1]); FFN_USE1(1) FFN_READ(fr[1], wt, 9 * 1024);
        FFN_WAIT(7, fr[2]); FFN_USE1(2) FFN_READ(fr[2], wt, 10 * 1024);
        FFN_STAGE(0)
        FFN_WAIT(7, fr[3]); FFN_USE1(3) FFN_READ(fr[3], wt, 11 * 1024);
        FFN_WAIT(7, fr[4]); FFN_USE1(4) FFN_READ(fr[4], wt, 12 * 1024);
        FFN_WAIT(7, fr[5]); FFN_USE1(5) FFN_READ(fr[5], wt, 13 * 1024);
        FFN_WAIT(7, fr[6]); FFN_USE1(6) FFN_READ(fr[6], wt, 14 * 1024);
        FFN_STAGE(1)
        FFN_WAIT(7, fr[7]); FFN_USE1(7) FFN_READ(fr[7], wt, 15 * 1024);
        FFN_WAIT(7, fr[0]); FFN_USE1(8) FFN_READ(fr[0], wt, 16 * 1024);
        FFN_WAIT(7, fr[1]); FFN_USE1(9) FFN_READ(fr[1], wt, 17 * 1024);
        FFN_WAIT(7, fr[2]); FFN_USE1(10) FFN_READ(fr[2], wt, 18 * 1024);
        FFN_STAGE(2)
        FFN_WAIT(7, fr[3]); FFN_USE1(11) FFN_READ(fr[3], wt, 19 * 1024);
        FFN_WAIT(7, fr[4]); FFN_USE1(12) FFN_READ(fr[4], wt, 20 * 1024);
        FFN_WAIT(7, fr[5]); FFN_USE1(13) FFN_READ(fr[5], wt, 21 * 1024);
        FFN_WAIT(7, fr[6]); FFN_USE1(14) FFN_READ(fr[6], wt, 22 * 1024);
        FFN_STAGE(3)
        FFN_WAIT(7, fr[7]); FFN_USE1(15) FFN_READ(fr[7], wt, 23 * 1024);
        FFN_STAMP(2)   // first product
        // (results of an inline MFMA: the compiler does not know to keep its distance -- a 4-pass MFMA's result may be read by a
        // vector instruction 11 wait states later)
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(hacc[0][0]), "+v"(hacc[0][1]), "+v"(hacc[1][0]), "+v"(hacc[1][1]), "+v"(hacc[2][0]), "+v"(hacc[2][1]));
        // ---- relu, to bf16: elements 0..3 = rows 4 q + 0..3 of row tile 0, elements 4..7 = the same rows of row tile 1 ------------
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            u32x4 u;
            u[0] = pack_bf16(relu1(hacc[ct][0][0]), relu1(hacc[ct][0][1]));
            u[1] = pack_bf16(relu1(hacc[ct][0][2]), relu1(hacc[ct][0][3]));
            u[2] = pack_bf16(relu1(hacc[ct][1][0]), relu1(hacc[ct][1][1]));
            u[3] = pack_bf16(relu1(hacc[ct][1][2]), relu1(hacc[ct][1][3]));
            hb[ct] = __builtin_bit_cast(bf16x8, u);
        }
        FFN_STAMP(3)   // relu + conversion
        // ---- out^T += W2 tile . relu(H^T): fragment 16 + t = 16-channel row tile t ------------------------------------------------
        FFN_WAIT(7, fr[0]); FFN_USE2(16) FFN_READ(fr[0], wt, 24 * 1024);
        FFN_WAIT(7, fr[1]); FFN_USE2(17) FFN_READ(fr[1], wt, 25 * 1024);
        FFN_WAIT(7, fr[2]); FFN_USE2(18) FFN_READ(fr[2], wt, 26 * 1024);
        FFN_STAGE(4)
        FFN_WAIT(7, fr[3]); FFN_USE2(19) FFN_READ(fr[3], wt, 27 * 1024);
        FFN_WAIT(7, fr[4]); FFN_USE2(20) FFN_READ(fr[4], wt, 28 * 1024);
        FFN_WAIT(7, fr[5]); FFN_USE2(21) FFN_READ(fr[5], wt, 29 * 1024);
        FFN_WAIT(7, fr[6]); FFN_USE2(22) FFN_READ(fr[6], wt, 30 * 1024);
        FFN_STAGE(5)
        FFN_WAIT(7, fr[7]); FFN_USE2(23) FFN_READ(fr[7], wt, 31 * 1024);
        FFN_WAIT(7, fr[0]); FFN_USE2(24)
        FFN_WAIT(6, fr[1]); FFN_USE2(25)
        FFN_WAIT(5, fr[2]); FFN_USE2(26)
        FFN_STAGE(6)
        FFN_WAIT(4, fr[3]); FFN_USE2(27)
        FFN_WAIT(3, fr[4]); FFN_USE2(28)
        FFN_WAIT(2, fr[5]); FFN_USE2(29)
        FFN_WAIT(1, fr[6]); FFN_USE2(30)
        FFN_STAGE(7)
        FFN_WAIT(0, fr[7]); FFN_USE2(31)
#undef FFN_USE1
#undef FFN_USE2
#undef FFN_STAGE
#undef FFN_MFMA1
#undef FFN_MFMA1_ACC
        FFN_STAMP(4)   // second product
        slot = slot == kRing - 1 ? 0 : slot + 1;
    }

    if (stamps && lane == 0 && wave == 0)
        for (int i = 0; i < 5; ++i) stamps[(size_t)blockIdx.x * 8 + i] = t_acc[i];
#undef FFN_STAMP

    // ---- epilogue: + b2 + x, LayerNorm over the 256 channels of a token (64 in this lane, the rest in lanes ^16, ^32, ^48), bf16 store ---
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        const int tok = tok0 + 16 * ct + c;
        const bool live = tok < T;
        const uint16_t *xrow = x + (size_t)min(tok, T - 1) * kD;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int ch = 16 * t + 4 * q;   // channels ch .. ch + 3 <-> the four registers of row tile t
            const uint2 xr = *reinterpret_cast<const uint2 *>(xrow + ch);
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(lb2 + ch);
            const float xv[4] = {bf16_lo(xr.x), bf16_hi(xr.x), bf16_lo(xr.y), bf16_hi(xr.y)};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[ct][t][i] += bb[i] + xv[i];
                sum += acc[ct][t][i];
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.f / kD);
        float var = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = acc[ct][t][i] - mean;
                var += d * d;
            }
        var += __shfl_xor(var, 16, 64);
        var += __shfl_xor(var, 32, 64);
        const float rstd = rsqrtf(var * (1.f / kD) + eps);
        if (rstd_out && live && q == 0) rstd_out[tok] = rstd;      // for the backward (msda_ffn_ln_backward_bf16)
        if (live) {
            uint16_t *orow = out + (size_t)tok * kD;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int ch = 16 * t + 4 * q;
                const f32x4 ga = *reinterpret_cast<const f32x4 *>(lgam + ch), be = *reinterpret_cast<const f32x4 *>(lbet + ch);
                float y[4], yh[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    yh[i] = (acc[ct][t][i] - mean) * rstd;
                    y[i] = yh[i] * ga[i] + be[i];
                }
                uint2 o;
                o.x = pack_bf16(y[0], y[1]);
                o.y = pack_bf16(y[2], y[3]);
                *reinterpret_cast<uint2 *>(orow + ch) = o;
                if (yhat_out) {      // training: the normalised pre-affine value, for the LayerNorm's backward
                    uint2 h;
                    h.x = pack_bf16(yh[0], yh[1]);
                    h.y = pack_bf16(yh[2], yh[3]);
                    *reinterpret_cast<uint2 *>(yhat_out + (size_t)tok * kD + ch) = h;
                }
            }
        }
    }
}

// ---- backward, first step: gradient at the LayerNorm's input ------------------------------------------------------------------------
// out = yhat * gamma + beta with yhat = (y - mean) * rstd, y = x + W2 relu(W1 x + b1) + b2.  Given dy (gradient of out), out and rstd:
//     yhat                                  (stored by the forward in bf16: recovering it as (out - beta) / gamma would amplify the
//                                            output's rounding by |out / gamma| and fail for a zero gamma)
//     g    = dy * gamma;   dz = rstd * (g - mean_c(g) - yhat * mean_c(g * yhat))          (= gradient of y: of the residual x, of b2, ...)
//     dgamma += dy * yhat;  dbeta += dy;  db2 += dz                                         (summed over the tokens)
// One token per wave and iteration (lane = 4 channels), sums over the tokens in registers, folded through LDS and added with atomics
// (3 x 256 per workgroup) to the zeroed fp32 results -- the launch is kept at 512 workgroups for them: 59 us at 2048 workgroups, 105 us
// at 4096, 33 us at 512, 40 us at 256.
__global__ __launch_bounds__(256) void ffn_ln_backward_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ yhat,
                                                              const float *__restrict__ rstd, const float *__restrict__ gamma, int T,
                                                              uint16_t *__restrict__ dz,
                                                              float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ db2)
{
    __shared__ float red[4][3][kD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ch = 4 * lane;
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + ch);
    float s_g[4] = {0.f, 0.f, 0.f, 0.f}, s_b[4] = {0.f, 0.f, 0.f, 0.f}, s_z[4] = {0.f, 0.f, 0.f, 0.f};
    for (int tok = blockIdx.x * 4 + wave; tok < T; tok += gridDim.x * 4) {
        const uint2 d = *reinterpret_cast<const uint2 *>(dy + (size_t)tok * kD + ch), o = *reinterpret_cast<const uint2 *>(yhat + (size_t)tok * kD + ch);
        const float dv[4] = {bf16_lo(d.x), bf16_hi(d.x), bf16_lo(d.y), bf16_hi(d.y)};
        const float ov[4] = {bf16_lo(o.x), bf16_hi(o.x), bf16_lo(o.y), bf16_hi(o.y)};
        float yh[4], g[4], a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            yh[i] = ov[i];
            g[i] = dv[i] * ga[i];
            a += g[i];
            b += g[i] * yh[i];
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            a += __shfl_xor(a, m, 64);
            b += __shfl_xor(b, m, 64);
        }
        const float r = rstd[tok], ma = a * (1.f / kD), mb = b * (1.f / kD);
        float z[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            z[i] = r * (g[i] - ma - yh[i] * mb);
            s_g[i] += dv[i] * yh[i];
            s_b[i] += dv[i];
            s_z[i] += z[i];
        }
        uint2 zo;
        zo.x = pack_bf16(z[0], z[1]);
        zo.y = pack_bf16(z[2], z[3]);
        *reinterpret_cast<uint2 *>(dz + (size_t)tok * kD + ch) = zo;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[wave][0][ch + i] = s_g[i];
        red[wave][1][ch + i] = s_b[i];
        red[wave][2][ch + i] = s_z[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * kD; i += 256) {
        const int which = i / kD, c = i - which * kD;
        const float v = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
        atomicAdd((which == 0 ? dgamma : which == 1 ? dbeta : db2) + c, v);
    }
}

// ---- residual add + LayerNorm (the norm1 of the transformer layers: reference deformable_transformer.py:876-877, src = norm1(src + src2)) ----
// out = LayerNorm(a + b) * gamma + beta over 256 channels, bf16 in / out, fp32 statistics; one token per wave and iteration (lane = 4
// channels).  Also writes rstd and the normalised value yhat for the backward (msda_ffn_ln_backward_bf16 serves it: its dz is the
// gradient of both a and b).
__global__ __launch_bounds__(256) void add_layernorm_kernel(const uint16_t *__restrict__ a, const uint16_t *__restrict__ b,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta, float eps, int T,
                                                            uint16_t *__restrict__ out, float *__restrict__ rstd_out,
                                                            uint16_t *__restrict__ yhat_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ch = 4 * lane;
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + ch), be = *reinterpret_cast<const f32x4 *>(beta + ch);
    for (int tok = blockIdx.x * 4 + wave; tok < T; tok += gridDim.x * 4) {
        const uint2 ua = *reinterpret_cast<const uint2 *>(a + (size_t)tok * kD + ch);
        float v[4] = {bf16_lo(ua.x), bf16_hi(ua.x), bf16_lo(ua.y), bf16_hi(ua.y)};
        if (b) {
            const uint2 ub = *reinterpret_cast<const uint2 *>(b + (size_t)tok * kD + ch);
            v[0] += bf16_lo(ub.x); v[1] += bf16_hi(ub.x); v[2] += bf16_lo(ub.y); v[3] += bf16_hi(ub.y);
        }
        float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
        const float mean = s * (1.f / kD);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] -= mean;
            q += v[i] * v[i];
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) q += __shfl_xor(q, m, 64);
        const float r = rsqrtf(q * (1.f / kD) + eps);
        float yh[4], y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            yh[i] = v[i] * r;
            y[i] = yh[i] * ga[i] + be[i];
        }
        uint2 o;
        o.x = pack_bf16(y[0], y[1]);
        o.y = pack_bf16(y[2], y[3]);
        *reinterpret_cast<uint2 *>(out + (size_t)tok * kD + ch) = o;
        if (yhat_out) {
            uint2 h;
            h.x = pack_bf16(yh[0], yh[1]);
            h.y = pack_bf16(yh[2], yh[3]);
            *reinterpret_cast<uint2 *>(yhat_out + (size_t)tok * kD + ch) = h;
        }
        if (rstd_out && lane == 0) rstd_out[tok] = r;
    }
}

size_t ffn_lds_bytes(int F) { return (size_t)kRing * kTileFrags * kFragShorts * 2 + (size_t)(F + 3 * kD) * 4; }

}  // namespace

unsigned long long *g_ffn_stamps = nullptr;

extern "C" {

/* Diagnostic: when `device_buffer` is non-NULL the feed-forward kernel adds up the shader clocks its wave 0 spends per loop stage
 * (wait for the weight tile, barrier, first product, relu + conversion, second product) into 8 x 8 bytes per workgroup. */
int msda_ffn_debug_stamps(void *device_buffer)
{
    g_ffn_stamps = static_cast<unsigned long long *>(device_buffer);
    return MSDA_OK;
}

int msda_ffn_pack_w2_bf16(const uint16_t *w2, int d_model, int d_ffn, uint16_t *w2_packed, msda_stream_t stream)
{
    if (!w2 || !w2_packed) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (d_model != kD || d_ffn < kHT || d_ffn % kHT != 0 || d_ffn > kMaxFfn) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    const long long n = (long long)d_model * d_ffn;
    hipLaunchKernelGGL(pack_w2_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream), w2, w2_packed, n);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

/* msda_ffn_forward_bf16 that also writes what the backward needs besides x: the LayerNorm's 1 / sqrt(var + eps) per token (rstd, `tokens`
 * floats) and its normalised input yhat = (y - mean) * rstd (tokens x 256 bf16); either may be NULL */
int msda_ffn_forward_train_bf16(const uint16_t *x, const uint16_t *w1, const float *b1, const uint16_t *w2_packed, const float *b2,
                                const float *ln_weight, const float *ln_bias, float eps, int tokens, int d_model, int d_ffn,
                                uint16_t *out, float *rstd, uint16_t *yhat, msda_stream_t stream);

int msda_ffn_forward_bf16(const uint16_t *x, const uint16_t *w1, const float *b1, const uint16_t *w2_packed, const float *b2,
                          const float *ln_weight, const float *ln_bias, float eps, int tokens, int d_model, int d_ffn,
                          uint16_t *out, msda_stream_t stream)
{
    return msda_ffn_forward_train_bf16(x, w1, b1, w2_packed, b2, ln_weight, ln_bias, eps, tokens, d_model, d_ffn, out, nullptr, nullptr, stream);
}

int msda_ffn_forward_train_bf16(const uint16_t *x, const uint16_t *w1, const float *b1, const uint16_t *w2_packed, const float *b2,
                                const float *ln_weight, const float *ln_bias, float eps, int tokens, int d_model, int d_ffn,
                                uint16_t *out, float *rstd, uint16_t *yhat, msda_stream_t stream)
{
    if (!x || !w1 || !b1 || !w2_packed || !b2 || !ln_weight || !ln_bias || !out) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (d_model != kD || d_ffn < kHT || d_ffn % kHT != 0 || d_ffn > kMaxFfn || tokens < 0) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2_packed) |
         reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(b2) |
         reinterpret_cast<uintptr_t>(ln_weight) | reinterpret_cast<uintptr_t>(ln_bias)) & 15)
        return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    if (tokens == 0) return MSDA_OK;
    const size_t lds = ffn_lds_bytes(d_ffn);
    static std::atomic<bool> raised[64];   // per device: the dynamic-LDS limit is a property of (function, device)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return msda_note_error(MSDA_ERR_NO_DEVICE, __func__);
    if (!raised[dev].load()) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ffn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        raised[dev] = true;
    }
    const int grid = (tokens + kTokWg - 1) / kTokWg;
    hipLaunchKernelGGL(ffn_fwd_kernel, dim3(grid), dim3(kWaves * 64), lds, static_cast<hipStream_t>(stream), x, w1, b1, w2_packed, b2,
                       ln_weight, ln_bias, eps, tokens, d_ffn, out, g_ffn_stamps, rstd, yhat);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

/* Backward, first step (see ffn_ln_backward_kernel): dy, yhat (tokens, 256) bf16; rstd (tokens), yhat from msda_ffn_forward_train_bf16;
 * ln_weight (256) f32 -> dz (tokens, 256) bf16 = gradient at the LayerNorm's input, and the three 256-vectors grad_ln_weight,
 * grad_ln_bias, grad_b2 (f32, zeroed here). */
int msda_ffn_ln_backward_bf16(const uint16_t *dy, const uint16_t *yhat, const float *rstd, const float *ln_weight, int tokens, int d_model,
                              uint16_t *dz, float *grad_ln_weight, float *grad_ln_bias, float *grad_b2, msda_stream_t stream)
{
    if (!dy || !yhat || !rstd || !ln_weight || !dz || !grad_ln_weight || !grad_ln_bias || !grad_b2) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (tokens < 0 || d_model != kD) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(yhat) | reinterpret_cast<uintptr_t>(dz) |
         reinterpret_cast<uintptr_t>(ln_weight)) & 15)
        return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (float *p : {grad_ln_weight, grad_ln_bias, grad_b2}) {
        const hipError_t e = hipMemsetAsync(p, 0, kD * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    if (tokens == 0) return MSDA_OK;
    const int grid = (tokens + 3) / 4 < 512 ? (tokens + 3) / 4 : 512;
    hipLaunchKernelGGL(ffn_ln_backward_kernel, dim3(grid), dim3(256), 0, st, dy, yhat, rstd, ln_weight, tokens, dz, grad_ln_weight, grad_ln_bias,
                       grad_b2);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

/* out = LayerNorm(a + b) (256 channels; b may be NULL): a, b, out (tokens, 256) bf16; gamma, beta f32; rstd (tokens) f32 and yhat
 * (tokens, 256) bf16 for the backward, either may be NULL.  The backward is msda_ffn_ln_backward_bf16 (dz = gradient of a and of b). */
int msda_add_layernorm_forward_bf16(const uint16_t *a, const uint16_t *b, const float *ln_weight, const float *ln_bias, float eps, int tokens,
                                    int d_model, uint16_t *out, float *rstd, uint16_t *yhat, msda_stream_t stream)
{
    if (!a || !ln_weight || !ln_bias || !out) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (tokens < 0 || d_model != kD) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(yhat) |
         reinterpret_cast<uintptr_t>(ln_weight) | reinterpret_cast<uintptr_t>(ln_bias)) & 15)
        return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    if (tokens == 0) return MSDA_OK;
    const int grid = (tokens + 3) / 4 < 4096 ? (tokens + 3) / 4 : 4096;
    hipLaunchKernelGGL(add_layernorm_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, ln_weight, ln_bias, eps, tokens,
                       out, rstd, yhat);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

}  // extern "C"
