// conv_wgrad.hip -- gradient of the convolution w.r.t. its weight on the gfx950 matrix cores (the trained
// backbone stages layer2-4, models/richsem/backbone.py:65-67).
//
//     dW[co][kh][kw][ci] = sum over output pixels p = (n, ho, wo) of  dz[p][co] * x[n, ho s + kh - pad, wo s + kw - pad][ci]
//
// (dz = gradient at the convolution's output, i.e. after the ReLU mask and the frozen affine's scale).  Per tap (kh, kw) that is a GEMM
// dz^T (C_out x P) . x_shifted (P x C_in) whose contraction index is the PIXEL -- the strided dimension of both NHWC operands.  The
// operands are therefore staged through LDS in their natural [pixel][channel] layout (coalesced 16-byte loads of whole pixel rows, any
// stride / shift / padding handled at the load) and read back TRANSPOSED by gfx950's ds_read_b64_tr_b16 (tools/tr_probe.hip checks what
// it delivers): a 16-lane group fetches 4 pixels x 16 channels and every lane receives its channel's 4 pixels, two such reads make
// one mfma_f32_16x16x32_bf16 operand (A: row = output channel, B: column = input channel, k = 8 consecutive pixels per lane group).
// The LDS image is the guide's swizzled 256-byte-row layout (16-byte chunk ch of row r at 256 r + 16 (ch ^ ((r & 3) << 2 | (r >> 2) & 3)))
// so that neither the row-wise stores nor the transposed reads conflict.
//   workgroup = (pixel chunk, tap, 128 output channels, 128 input channels), four waves of 64 x 64 (16 accumulator tiles each);
//   64 pixels per stage (two k-steps), double-buffered through registers, one barrier per stage;
//   split-K over pixel chunks only where the (tap, channel block) grid alone cannot fill the chip; the chunks' partial results go to a
//   caller-provided workspace with plain stores and are summed by a second kernel (fp32 atomics into the result were tried first: the
//   L2 atomic rate made them the whole run time -- 39 us of a 40 us launch for layer4's 3 x 3).  Result layout (C_out, KH, KW, C_in).
// bf16 operands, fp32 accumulation.  C_out % 128 == 0 and C_in % 128 == 0 (all convolutions of ResNet-50's layer2-4).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include <atomic>

#include "../../include/richsem_msda.h"

extern "C" int msda_note_error(int code, const char *entry);      // msda_api.hip: sets msda_last_error()

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 128, kBN = 128;     // output / input channels per workgroup
constexpr int kStagePx = 64;            // pixels per stage (two k-steps of 32)
constexpr int kThreads = 256;
constexpr int kImageBytes = kStagePx * 256;   // one operand's stage: 64 rows of 128 bf16

struct WgradGeom {
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    long long P, chunk;     // output pixels; pixels per workgroup (multiple of kStagePx)
    int torch_layout;       // final result as (C_out, C_in, KH, KW) (nn.Conv2d's) instead of (C_out, KH, KW, C_in)
};

__device__ __forceinline__ int lds_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// one workgroup's work: pixel chunk `bx`, (tap, channel blocks) `by_` of problem g
__device__ __forceinline__ void wgrad_block(const uint16_t *__restrict__ dz, const uint16_t *__restrict__ x, float *__restrict__ dw,
                                            float *__restrict__ dbias, const float *__restrict__ scale, int direct, const WgradGeom &g,
                                            int bx, int by_, unsigned char (*lds)[2 * kImageBytes])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb_ci = g.Cin / kBN, nb_co = g.Cout / kBM;
    int by = by_;
    const int cib = by % nb_ci;
    by /= nb_ci;
    const int cob = by % nb_co, tap = by / nb_co;
    const int kh = tap / g.KW, kw = tap - kh * g.KW;
    const long long p0 = (long long)bx * g.chunk;
    const long long p1 = p0 + g.chunk < g.P ? p0 + g.chunk : g.P;
    const int n_stage = (int)((p1 - p0 + kStagePx - 1) / kStagePx);

    // staging: thread -> chunk column ch = tid % 16 of rows tid / 16 + 16 i (i < 4), for both operands.  The rows' pixels are decoded
    // once (32-bit: the host checks P < 2^31) and advanced by a stage's 64 pixels from then on -- no division in the loop.
    const int ch = tid & 15, row0 = tid >> 4;
    const int P1 = (int)p1;
    int rp[4], rn[4], rho[4], rwo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rp[i] = (int)p0 + row0 + 16 * i;
        const int pc = rp[i] < (int)g.P ? rp[i] : 0;
        rwo[i] = pc % g.Wo;
        rho[i] = (pc / g.Wo) % g.Ho;
        rn[i] = pc / (g.Wo * g.Ho);
    }
    const uint16_t *dz_col = dz + cob * kBM + ch * 8, *x_col = x + cib * kBN + ch * 8;
    // one stage ahead in one register set (a second set -- two stages in flight -- spills 195 registers at the 256-register budget of
    // two workgroups per CU and doubles the run time; with 512 registers the kernel runs one wave per SIMD and gains nothing)
    u32x4 sa[4], sb[4];
    // optional: the bias gradient sum_p dz[p][co], formed on the way by the workgroups of tap 0 / input block 0 (this thread: its 8 channels)
    const bool want_bias = dbias != nullptr && tap == 0 && cib == 0;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto fetch = [&]() {      // the next stage's rows; advances the row coordinates
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 va = {0u, 0u, 0u, 0u}, vb = {0u, 0u, 0u, 0u};
            if (rp[i] < P1) {
                va = *reinterpret_cast<const u32x4 *>(dz_col + (size_t)rp[i] * g.Cout);
                const int hi = rho[i] * g.stride + kh - g.pad, wi = rwo[i] * g.stride + kw - g.pad;
                if (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W)
                    vb = *reinterpret_cast<const u32x4 *>(x_col + ((size_t)(rn[i] * g.H + hi) * g.W + wi) * g.Cin);
            }
            sa[i] = va;
            sb[i] = vb;
            rp[i] += kStagePx;
            rwo[i] += kStagePx;
            while (rwo[i] >= g.Wo) {
                rwo[i] -= g.Wo;
                if (++rho[i] == g.Ho) {
                    rho[i] = 0;
                    ++rn[i];
                }
            }
        }
    };
    auto park = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = lds_off(row0 + 16 * i, ch);
            *reinterpret_cast<u32x4 *>(lds[slot] + off) = sa[i];
            *reinterpret_cast<u32x4 *>(lds[slot] + kImageBytes + off) = sb[i];
            if (want_bias) {      // (here, where the rows have arrived anyway: summing them at the load would wait for it)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += __uint_as_float(sa[i][e] << 16);
                    bsum[2 * e + 1] += __uint_as_float(sa[i][e] & 0xFFFF0000u);
                }
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transposed operand reads: lane = (group gI = lane / 16, i = lane % 16 = 4 q + p); the group's block = pixels 8 gI (+ 4) .. of the
    // k-step, channels of the tile; this lane supplies row q, 16-byte chunk 2 tile + (p >> 1), half p & 1
    const int gI = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int wm = wave & 1, wn = wave >> 1;
    auto frag = [&](unsigned char *image, int ks, int tile) {
        const int r = 32 * ks + 8 * gI + q;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4 *)(image + lds_off(r, 2 * tile + (pp >> 1)) + 8 * (pp & 1)));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4 *)(image + lds_off(r + 4, 2 * tile + (pp >> 1)) + 8 * (pp & 1)));
        return (bf16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    };

    auto products = [&](int buf) {
        unsigned char *img_a = lds[buf], *img_b = lds[buf] + kImageBytes;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = frag(img_a, ks, 4 * wm + t);
                fb[t] = frag(img_b, ks, 4 * wn + t);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    };
    if (n_stage > 0) {
        fetch();
        park(0);
    }
    __syncthreads();
    for (int s = 0; s < n_stage; ++s) {
        if (s + 1 < n_stage) fetch();
        products(s & 1);
        if (s + 1 < n_stage) park((s + 1) & 1);
        __syncthreads();
    }

    // accumulator tile: lane (c = lane & 15, gI) holds rows (output channels) 4 gI + i, column (input channel) c; pixel chunk
    // blockIdx.x writes its own slice of the output (the result itself when there is one chunk, else the workspace)
    const int taps = g.KH * g.KW;
    float *out = dw + (size_t)bx * g.Cout * taps * g.Cin;
    if (want_bias) {      // fold the 16 row groups (threads with equal ch) through LDS; chunk blockIdx.x writes its slice of dbias
        float *red = reinterpret_cast<float *>(lds[0]);     // (the last barrier of the loop has passed: the images are free)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(row0 * 16 + ch) * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < kBM) {
            float v = 0.f;
            for (int r = 0; r < 16; ++r) v += red[(r * 16 + (tid >> 3)) * 8 + (tid & 7)];
            dbias[(size_t)bx * g.Cout + cob * kBM + tid] = v;
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int ci = cib * kBN + 16 * (4 * wn + b) + li;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = cob * kBM + 16 * (4 * wm + a) + 4 * gI + i;
                if (direct)      // one chunk: the final result, scaled per output channel and in the layout asked for
                    out[g.torch_layout ? ((long long)co * g.Cin + ci) * taps + tap : ((long long)co * taps + tap) * g.Cin + ci] =
                        acc[a][b][i] * (scale ? scale[co] : 1.f);
                else
                    out[((long long)co * taps + tap) * g.Cin + ci] = acc[a][b][i];
            }
        }
}

// ---- the same workgroup with its operand stages prefetched THREE stages ahead through an LDS ring filled by LDS DMA (round 4) ---------
// wgrad_block asks for a stage one stage ahead (registers: a second set spills) and a stage is 512 MFMA cycles per wave: a workgroup's stage
// takes what a memory round trip takes (2.07 us per stage at the feed-forward block's 44646 tokens: the matrix pipe 20 % busy at two
// workgroups per CU).  Here a stage is 32 pixels (one k-step: 8 KB of each operand, the same swizzled 256-byte-row image), the ring four
// stages (the same 64 KB), and the rows go from memory to LDS without passing through registers: global_load_lds_dwordx4 writes lane l's
// 16 bytes at (base + 16 l), i.e. row (l / 16) and chunk SLOT (l % 16) of four consecutive rows -- so the lane fetches the chunk that
// BELONGS in that slot, ch = slot ^ swizzle(row) (the swizzle is applied on the memory side; rows beyond the chunk's pixels or in the
// padding come from a zero line).  A wave brings rows 8 w .. 8 w + 7 of both images: four requests per stage.  A stage begins with
// s_waitcnt vmcnt(8) (two younger stages may be in flight) and one bare s_barrier; the transposed operand reads are inline
// ds_read_b64_tr_b16 (the compiler would wait for every DMA in flight before an LDS read it can see), released in four steps by lgkmcnt.
// The bias gradient, which wgrad_block sums from its staging registers, is a fifth column of products here: A fragments x a fragment of ones.
__device__ __attribute__((aligned(16))) unsigned g_wgrad_zero_line[4];      // (device globals are zero-initialised)

constexpr int kRingStagePx = 32, kRingSlots = 4, kRingImage = kRingStagePx * 256, kRingSlotBytes = 2 * kRingImage;
static_assert(kRingSlots * kRingSlotBytes == 2 * 2 * kImageBytes, "the ring is the old double buffer's LDS");

#ifndef WGRAD_RING_ABLATE      // diagnostic builds (wrong results): 1 = no requests inside the loop, 2 = no products, 4 = no barrier, 8 = plain 8-byte reads, 16 = no MFMAs
#define WGRAD_RING_ABLATE 0
#endif
#if WGRAD_RING_ABLATE & 8
#define WG_TR_READ(dst, addr, byte_off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(byte_off))
#else
#define WG_TR_READ(dst, addr, byte_off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(byte_off))
#endif
#define WG_LDS_WAIT(n, a, b) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(n))

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct WgradRing {      // a workgroup's request side: the two rows (of four consecutive image rows each) a lane requests per stage and operand
    int rp[2], rn[2], rho[2], rwo[2];
    const uint16_t *acol[2], *bcol[2];      // the lane's chunk of a dz row / an x row (channel block and swizzled chunk folded in)
    unsigned char *ring;
    const uint16_t *zero;
    int wave, P1, kh, kw;

    // requests of one stage into ring slot `slot`; advances the rows by a stage (past the chunk's last pixel: zero lines)
    __device__ __forceinline__ void request(int slot, const WgradGeom &g)
    {
        request_row(0, slot, g);
        request_row(1, slot, g);
    }

    // ... one of its two halves (a dz row and an x row of the lane: two requests), so that the loop can place them between its MFMA groups
    __device__ __forceinline__ void request_row(int j, int slot, const WgradGeom &g)
    {
        {
            const bool live = rp[j] < P1;
            const int hi = rho[j] * g.stride + kh - g.pad, wi = rwo[j] * g.stride + kw - g.pad;
            const bool in = live && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W;
            const uint16_t *pa = live ? acol[j] + (size_t)rp[j] * g.Cout : zero;
            const uint16_t *pb = in ? bcol[j] + ((size_t)(rn[j] * g.H + hi) * g.W + wi) * g.Cin : zero;
            unsigned char *dst = ring + slot * kRingSlotBytes + (8 * wave + 4 * j) * 256;
            __builtin_amdgcn_global_load_lds(pa, reinterpret_cast<__attribute__((address_space(3))) void *>(reinterpret_cast<uintptr_t>(dst)), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(pb, reinterpret_cast<__attribute__((address_space(3))) void *>(reinterpret_cast<uintptr_t>(dst + kRingImage)), 16,
                                             0, 0);
            rp[j] += kRingStagePx;
            rwo[j] += kRingStagePx;
            while (rwo[j] >= g.Wo) {
                rwo[j] -= g.Wo;
                if (++rho[j] == g.Ho) {
                    rho[j] = 0;
                    ++rn[j];
                }
            }
        }
    }
};

// the products of the stage in the ring slot at byte offset `so` (a run-time slot: unrolling the loop over the four slots for immediate
// offsets gave every slot its own set of accumulators, 64 register moves per trip and spills whose reloads drained the DMA queue)
template <bool BIAS>
__device__ __forceinline__ void wgrad_ring_products(f32x4 (&acc)[4][4], f32x4 (&bacc)[4], const unsigned (&aa)[2][4], const unsigned (&ab)[2][4], unsigned so,
                                                    bool want_bias, WgradRing &rq, int req_slot, const WgradGeom &g)
{
    constexpr int oa = 0, ob = kRingImage;
    u32x2 fa[4][2], fb[4][2];
    // order of the reads = order of the waits: A tile 0, the four B tiles, then A tiles 1-3
    WG_TR_READ(fa[0][0], aa[0][0] + so, oa); WG_TR_READ(fa[0][1], aa[1][0] + so, oa);
#pragma unroll
    for (int t = 0; t < 4; ++t) { WG_TR_READ(fb[t][0], ab[0][t] + so, ob); WG_TR_READ(fb[t][1], ab[1][t] + so, ob); }
#pragma unroll
    for (int t = 1; t < 4; ++t) { WG_TR_READ(fa[t][0], aa[0][t] + so, oa); WG_TR_READ(fa[t][1], aa[1][t] + so, oa); }
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
#define WG_FRAG(f) __builtin_bit_cast(bf16x8, (u32x4){(f)[0][0], (f)[0][1], (f)[1][0], (f)[1][1]})
#define WG_ROW(A_, NEWER)                                                                                                      \
    WG_LDS_WAIT(NEWER, fa[A_][0], fa[A_][1]);                                                                                  \
    {                                                                                                                          \
        const bf16x8 a = WG_FRAG(fa[A_]);                                                                                      \
        _Pragma("unroll") for (int b = 0; b < ((WGRAD_RING_ABLATE & 16) ? 1 : 4); ++b)                                         \
            acc[A_][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, WG_FRAG(fb[b]), acc[A_][b], 0, 0, 0);                      \
        if (BIAS && want_bias) bacc[A_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, ones, bacc[A_], 0, 0, 0);                 \
    }
    asm volatile("" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]));      // (the B fragments are read by the MFMAs only behind ...
    asm volatile("" : "+v"(fb[2][0]), "+v"(fb[2][1]), "+v"(fb[3][0]), "+v"(fb[3][1]));
    // the next-but-two stage's requests go between the MFMA groups (an LDS-DMA instruction holds its wave for ~100 cycles of issue: under
    // the matrix pipe's work instead of in front of it); the scheduling barriers keep the groups where they are written
    WG_ROW(0, 6)      // ... this wait: at most A tiles 1-3 (six reads) still in flight)
    __builtin_amdgcn_sched_barrier(0);
    if (!(WGRAD_RING_ABLATE & 1)) rq.request_row(0, req_slot, g);
    __builtin_amdgcn_sched_barrier(0);
    WG_ROW(1, 4)
    __builtin_amdgcn_sched_barrier(0);
    if (!(WGRAD_RING_ABLATE & 1)) rq.request_row(1, req_slot, g);
    __builtin_amdgcn_sched_barrier(0);
    WG_ROW(2, 2)
    __builtin_amdgcn_sched_barrier(0);
    WG_ROW(3, 0)
#undef WG_ROW
#undef WG_FRAG
}

template <bool BIAS>
__device__ __forceinline__ void wgrad_block_ring(const uint16_t *__restrict__ dz, const uint16_t *__restrict__ x, float *__restrict__ dw,
                                                 float *__restrict__ dbias, const float *__restrict__ scale, int direct, const WgradGeom &g, int bx,
                                                 int by_, unsigned char *ring)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb_ci = g.Cin / kBN, nb_co = g.Cout / kBM;
    int by = by_;
    const int cib = by % nb_ci;
    by /= nb_ci;
    const int cob = by % nb_co, tap = by / nb_co;
    const long long p0 = (long long)bx * g.chunk;
    const long long p1 = p0 + g.chunk < g.P ? p0 + g.chunk : g.P;
    const int n_stage = (int)((p1 - p0 + kRingStagePx - 1) / kRingStagePx);

    // request side: rows 8 wave + 4 j + lane / 16 of a stage, chunk slot lane % 16 -> chunk (slot ^ swizzle(row)) of the pixel's 256-byte row
    WgradRing rq;
    rq.ring = ring;
    rq.zero = reinterpret_cast<const uint16_t *>(g_wgrad_zero_line);
    rq.wave = wave;
    rq.P1 = (int)p1;
    rq.kh = tap / g.KW;
    rq.kw = tap - rq.kh * g.KW;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * wave + 4 * j + (lane >> 4);
        const int chunk = (lane & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
        rq.acol[j] = dz + cob * kBM + chunk * 8;
        rq.bcol[j] = x + cib * kBN + chunk * 8;
        rq.rp[j] = (int)p0 + r;
        const int pc = rq.rp[j] < (int)g.P ? rq.rp[j] : 0;
        rq.rwo[j] = pc % g.Wo;
        rq.rho[j] = (pc / g.Wo) % g.Ho;
        rq.rn[j] = pc / (g.Wo * g.Ho);
    }

    f32x4 acc[4][4], bacc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        bacc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // read side: lane = (group gI = lane / 16, i = lane % 16 = 4 q + p): rows r = 8 gI + q and r + 4 of the stage, chunk 2 tile + (p >> 1),
    // half p & 1 (wgrad_block's frag)
    const int gI = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int wm = wave & 1, wn = wave >> 1;
    unsigned aa[2][4], ab[2][4];
    const unsigned base = (unsigned)(uintptr_t)ring;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int r = 8 * gI + q + 4 * v;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            aa[v][t] = base + lds_off(r, 2 * (4 * wm + t) + (pp >> 1)) + 8 * (pp & 1);
            ab[v][t] = base + lds_off(r, 2 * (4 * wn + t) + (pp >> 1)) + 8 * (pp & 1);
        }
    }

    // three stages ahead (stages past the chunk's end are zero lines: the request count per stage is a constant)
#pragma unroll
    for (int r = 0; r < kRingSlots - 1; ++r) rq.request(r, g);
    const bool want_bias = BIAS && dbias != nullptr && tap == 0 && cib == 0 && wn == 0;
    int slot = 0;
    for (int st = 0; st < n_stage; ++st) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // this stage has landed (this wave's requests; two younger stages may be in flight)
        if (!(WGRAD_RING_ABLATE & 4)) __builtin_amdgcn_s_barrier();      // ... everybody's; and the slot of stage st - 1 is free
        if (WGRAD_RING_ABLATE & 2) { rq.request((slot + kRingSlots - 1) % kRingSlots, g); slot = (slot + 1) % kRingSlots; continue; }
        wgrad_ring_products<BIAS>(acc, bacc, aa, ab, (unsigned)slot * kRingSlotBytes, want_bias, rq, (slot + kRingSlots - 1) % kRingSlots, g);      // (+ stage st + 3's requests)
        slot = (slot + 1) % kRingSlots;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the zero-line requests past the end)

    const int taps = g.KH * g.KW;
    float *out = dw + (size_t)bx * g.Cout * taps * g.Cin;
    if (want_bias) {      // every column of the ones product is the row sum: column 0's lanes write it (rows 16 (4 wm + a) + 4 gI + i)
        if (li == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i) dbias[(size_t)bx * g.Cout + cob * kBM + 16 * (4 * wm + a) + 4 * gI + i] = bacc[a][i];
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int ci = cib * kBN + 16 * (4 * wn + b) + li;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = cob * kBM + 16 * (4 * wm + a) + 4 * gI + i;
                if (direct)
                    out[g.torch_layout ? ((long long)co * g.Cin + ci) * taps + tap : ((long long)co * taps + tap) * g.Cin + ci] =
                        acc[a][b][i] * (scale ? scale[co] : 1.f);
                else
                    out[((long long)co * taps + tap) * g.Cin + ci] = acc[a][b][i];
            }
        }
}

template <bool RING, bool BIAS>
__global__ __launch_bounds__(kThreads, 2)      // (two workgroups per CU: without the hint the unrolled ring loop is given a fresh set of accumulators per slot)
void conv_wgrad_kernel(const uint16_t *__restrict__ dz, const uint16_t *__restrict__ x,
                                                              float *__restrict__ dw, float *__restrict__ dbias,
                                                              const float *__restrict__ scale, int direct, WgradGeom g)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2 * kImageBytes];   // [buffer][A image | B image], or the ring's four slots
    if constexpr (RING)
        wgrad_block_ring<BIAS>(dz, x, dw, dbias, scale, direct, g, blockIdx.x, blockIdx.y, &lds[0][0]);
    else
        wgrad_block(dz, x, dw, dbias, scale, direct, g, blockIdx.x, blockIdx.y, lds);
}

// Several weight gradients in ONE launch (a bottleneck block's three or four: msda_conv_wgrad_group_bf16).  Each problem alone would be
// split into ~512 workgroups of 4-8 stages; together they share the chip: a seventh of the partial sums, loops of 15-30 stages, one launch
// (and one reduction launch) instead of three or four of each.  Workgroup b belongs to the problem j with first[j] <= b < first[j + 1].
constexpr int kMaxGroup = 8;
struct WgradGroup {
    const uint16_t *dz[kMaxGroup], *x[kMaxGroup];
    float *part[kMaxGroup];          // where the chunks' slices go: the workspace, or the result itself for a problem of one chunk
    float *dw[kMaxGroup];
    float *bpart[kMaxGroup], *dbias[kMaxGroup];      // bias gradient (or null): where the chunks' slices go, and the result
    const float *scale[kMaxGroup];
    WgradGeom g[kMaxGroup];
    int split[kMaxGroup], first[kMaxGroup + 1];
    long long red_first[kMaxGroup + 1];      // reduction: float4 elements of the problems with more than one chunk, concatenated
    int n;
};

template <bool RING, bool BIAS>
__global__ __launch_bounds__(kThreads, 2) void conv_wgrad_group_kernel(WgradGroup grp)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2 * kImageBytes];
    int j = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroup; ++i)
        if (i < grp.n && (int)blockIdx.x >= grp.first[i]) j = i;      // (uniform)
    const int local = (int)blockIdx.x - grp.first[j], split = grp.split[j];
    if constexpr (RING)
        wgrad_block_ring<BIAS>(grp.dz[j], grp.x[j], grp.part[j], BIAS ? grp.bpart[j] : nullptr, grp.scale[j], split > 1 ? 0 : 1, grp.g[j], local % split,
                               local / split, &lds[0][0]);
    else
        wgrad_block(grp.dz[j], grp.x[j], grp.part[j], grp.bpart[j], grp.scale[j], split > 1 ? 0 : 1, grp.g[j], local % split, local / split, lds);
}

// the reduction of a group's split problems: 64 float4 elements per workgroup-iteration as in conv_wgrad_reduce_kernel
__global__ __launch_bounds__(256) void conv_wgrad_group_reduce_kernel(WgradGroup grp)
{
    __shared__ float4 red[4][64];
    const int col = threadIdx.x & 63, phase = threadIdx.x >> 6;
    // the bias partials of the split problems: 64 channels per workgroup, the same four phases (as conv_wgrad_reduce_kernel)
    for (int j = 0; j < grp.n; ++j) {
        const int cout = grp.g[j].Cout, split = grp.split[j];
        if (!grp.dbias[j] || split < 2 || (int)blockIdx.x * 64 >= cout) continue;      // (uniform)
        __shared__ float bred[4][64];
        const int c = blockIdx.x * 64 + col;
        float v = 0.f;
        if (c < cout)
            for (int s = phase; s < split; s += 4) v += grp.bpart[j][(size_t)s * cout + c];
        bred[phase][col] = v;
        __syncthreads();
        if (phase == 0 && c < cout) grp.dbias[j][c] = bred[0][col] + bred[1][col] + bred[2][col] + bred[3][col];
        __syncthreads();
    }
    const long long total = grp.red_first[grp.n];
    for (long long i0 = (long long)blockIdx.x * 64; i0 < total; i0 += (long long)gridDim.x * 64) {
        const long long ig = i0 + col;
        int j = 0;
#pragma unroll
        for (int k = 1; k < kMaxGroup; ++k)
            if (k < grp.n && ig >= grp.red_first[k]) j = k;
        const bool live = ig < total;
        const long long i = ig - grp.red_first[j], n4 = grp.red_first[j + 1] - grp.red_first[j];
        const int split = grp.split[j];
        const float4 *ws = reinterpret_cast<const float4 *>(grp.part[j]);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (live) {
            int s = phase;
            for (; s + 4 < split; s += 8) {
                const float4 u = ws[(long long)s * n4 + i], v = ws[(long long)(s + 4) * n4 + i];
                a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
                b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
            }
            if (s < split) {
                const float4 u = ws[(long long)s * n4 + i];
                a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            }
        }
        red[phase][col] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        __syncthreads();
        if (phase == 0 && live) {
            float4 r = red[0][col];
#pragma unroll
            for (int p = 1; p < 4; ++p) {
                r.x += red[p][col].x; r.y += red[p][col].y; r.z += red[p][col].z; r.w += red[p][col].w;
            }
            const WgradGeom &g = grp.g[j];
            const int taps = g.KH * g.KW, cin = g.Cin;
            const long long e0 = 4 * i;
            const int co = (int)(e0 / ((long long)taps * cin));
            if (grp.scale[j]) {
                const float sc = grp.scale[j][co];
                r.x *= sc; r.y *= sc; r.z *= sc; r.w *= sc;
            }
            float *dw = grp.dw[j];
            if (g.torch_layout && taps > 1) {
                const int rem = (int)(e0 - (long long)co * taps * cin), tap = rem / cin, ci = rem - tap * cin;
                float *d = dw + ((long long)co * cin + ci) * taps + tap;
                d[0] = r.x; d[taps] = r.y; d[2 * taps] = r.z; d[3 * taps] = r.w;
            } else {
                reinterpret_cast<float4 *>(dw)[i] = r;
            }
        }
        __syncthreads();
    }
}

// dw[i] = sum over the chunks' slices.  64 float4 elements per workgroup, four phases of threads per element (phase p sums slices
// p, p + 4, ... with the loads of several slices in flight; the first version -- one thread per element walking all slices -- took
// longer than the products it follows: 31 us for 64 slices of 256 KB), folded through LDS.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw, long long n4, int split,
                                                                const float *__restrict__ ws_bias, float *__restrict__ dbias, int cout,
                                                                const float *__restrict__ scale, int taps, int cin, int torch_layout)
{
    __shared__ float4 red[4][64];
    const int col = threadIdx.x & 63, phase = threadIdx.x >> 6;
    if (ws_bias && (int)blockIdx.x * 64 < cout) {      // the bias partials: 64 channels per workgroup, the same four phases
        __shared__ float bred[4][64];
        const int c = blockIdx.x * 64 + col;
        float v = 0.f;
        if (c < cout)
            for (int s = phase; s < split; s += 4) v += ws_bias[(size_t)s * cout + c];
        bred[phase][col] = v;
        __syncthreads();
        if (phase == 0 && c < cout) dbias[c] = bred[0][col] + bred[1][col] + bred[2][col] + bred[3][col];
    }
    for (long long i0 = (long long)blockIdx.x * 64; i0 < n4; i0 += (long long)gridDim.x * 64) {
        const long long i = i0 + col;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (i < n4) {
            int s = phase;
            for (; s + 4 < split; s += 8) {
                const float4 u = reinterpret_cast<const float4 *>(ws)[(long long)s * n4 + i];
                const float4 v = reinterpret_cast<const float4 *>(ws)[(long long)(s + 4) * n4 + i];
                a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
                b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
            }
            if (s < split) {
                const float4 u = reinterpret_cast<const float4 *>(ws)[(long long)s * n4 + i];
                a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            }
        }
        red[phase][col] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        __syncthreads();
        if (phase == 0 && i < n4) {
            float4 r = red[0][col];
#pragma unroll
            for (int p = 1; p < 4; ++p) {
                r.x += red[p][col].x; r.y += red[p][col].y; r.z += red[p][col].z; r.w += red[p][col].w;
            }
            // element 4 i of the (C_out, taps, C_in) order: scale per output channel, final layout
            const long long e0 = 4 * i;
            const int co = (int)(e0 / ((long long)taps * cin));
            if (scale) {
                const float sc = scale[co];
                r.x *= sc; r.y *= sc; r.z *= sc; r.w *= sc;
            }
            if (torch_layout && taps > 1) {
                const int rem = (int)(e0 - (long long)co * taps * cin), tap = rem / cin, ci = rem - tap * cin;
                float *d = dw + ((long long)co * cin + ci) * taps + tap;
                d[0] = r.x; d[taps] = r.y; d[2 * taps] = r.z; d[3 * taps] = r.w;
            } else {
                reinterpret_cast<float4 *>(dw)[i] = r;
            }
        }
        __syncthreads();
    }
}

// pixel chunks for a problem: at most two workgroups per CU (512: one resident wave) together with the (tap, channel block) grid, of at least
// 512 pixels each (measured: 256 and 1024 workgroups are both slower on the linear layers' shapes; a grid of 144 blocks still gains from 4 chunks)
void wgrad_split(const WgradGeom &g, long long &split, long long &chunk)
{
    const long long blocks_y = (long long)g.KH * g.KW * (g.Cout / kBM) * (g.Cin / kBN);
    split = 512 / blocks_y;      // rounded DOWN (measured: the ResNet shape set 432-437 -> 410 us, layer3's 3 x 3 43 -> 34.5 us): 512 workgroups are resident (two per CU); 540 would run as a full wave + a wave of 28
    // (few pixels -- the decoder's ~2 k tokens: chunks of 128, or 20 workgroups would each walk 512 pixels one stage after the other)
    const long long min_chunk = g.P >= 16384 ? 512 : 128;
    const long long max_split = (g.P + min_chunk - 1) / min_chunk;
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    chunk = ((g.P + split - 1) / split + kStagePx - 1) / kStagePx * kStagePx;
    split = (g.P + chunk - 1) / chunk;
}

std::atomic<int> g_wgrad_ring{1};      // msda_conv_set_wgrad_ring: 1 = operand stages through the LDS ring (wgrad_block_ring), 0 = register-staged

// a group's plan: the problems share ~512 workgroups in proportion to their work (pixels x taps x channel blocks), each chunk at least
// 256 pixels (4 stages)
int plan_group(const msda_wgrad_problem *probs, int n, WgradGroup &grp, int64_t &ws_floats)
{
    if (!probs || n < 1 || n > kMaxGroup) return MSDA_ERR_BAD_DIMS;
    double work[kMaxGroup], total = 0;
    for (int j = 0; j < n; ++j) {
        const msda_wgrad_problem &p = probs[j];
        if (p.N < 1 || p.H < 1 || p.W < 1 || p.Cin < kBN || p.Cin % kBN != 0 || p.Cout < kBM || p.Cout % kBM != 0 || p.KH < 1 || p.KW < 1 ||
            p.KH > 16 || p.KW > 16 || p.stride < 1 || p.pad < 0)
            return MSDA_ERR_BAD_DIMS;
        const int Ho = (p.H + 2 * p.pad - p.KH) / p.stride + 1, Wo = (p.W + 2 * p.pad - p.KW) / p.stride + 1;
        if (Ho < 1 || Wo < 1) return MSDA_ERR_BAD_DIMS;
        grp.g[j] = WgradGeom{p.N, p.H, p.W, p.Cin, Ho, Wo, p.Cout, p.KH, p.KW, p.stride, p.pad, (long long)p.N * Ho * Wo, 0, 1};
        if (grp.g[j].P >= (1ll << 31) - (1 << 20) || (long long)p.N * p.H * p.W >= (1ll << 31)) return MSDA_ERR_TOO_LARGE;
        work[j] = (double)grp.g[j].P * p.KH * p.KW * (p.Cout / kBM) * (p.Cin / kBN);
        total += work[j];
    }
    grp.n = n;
    grp.first[0] = 0;
    grp.red_first[0] = 0;
    ws_floats = 0;
    for (int j = 0; j < n; ++j) {
        const WgradGeom &g = grp.g[j];
        const long long blocks_y = (long long)g.KH * g.KW * (g.Cout / kBM) * (g.Cin / kBN);
        long long split = (long long)(512.0 * work[j] / total / (double)blocks_y + 0.5);
        const long long max_split = (g.P + 255) / 256;
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
        long long chunk = ((g.P + split - 1) / split + kStagePx - 1) / kStagePx * kStagePx;
        split = (g.P + chunk - 1) / chunk;
        grp.g[j].chunk = chunk;
        grp.split[j] = (int)split;
        grp.first[j + 1] = grp.first[j] + (int)(split * blocks_y);
        const long long n_dw = (long long)g.Cout * g.KH * g.KW * g.Cin;
        grp.red_first[j + 1] = grp.red_first[j] + (split > 1 ? n_dw / 4 : 0);
        if (split > 1) ws_floats += split * n_dw + (probs[j].dbias ? split * (long long)g.Cout : 0);
    }
    return MSDA_OK;
}

}  // namespace

extern "C" {

/* Several weight gradients (msda_conv_wgrad_bf16 problems with the result in nn.Conv2d's layout, scaled; no bias gradients) in one launch
 * of the product kernel and one of the reduction: the problems share the chip instead of each being cut into ~512 short workgroups.
 * n <= 8; workspace: msda_conv_wgrad_group_workspace_bytes() bytes (may be NULL when that is 0). */
int msda_conv_wgrad_group_workspace_bytes(const msda_wgrad_problem *problems, int n, int64_t *bytes)
{
    if (!bytes) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    WgradGroup grp;
    int64_t fl = 0;
    const int rc = plan_group(problems, n, grp, fl);
    if (rc != MSDA_OK) return msda_note_error(rc, __func__);
    *bytes = fl * (int64_t)sizeof(float);
    return MSDA_OK;
}

int msda_conv_wgrad_group_bf16(const msda_wgrad_problem *problems, int n, void *workspace, msda_stream_t stream)
{
    WgradGroup grp;
    int64_t fl = 0;
    const int rc = plan_group(problems, n, grp, fl);
    if (rc != MSDA_OK) return msda_note_error(rc, __func__);
    if (fl > 0 && (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15))) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    float *ws = static_cast<float *>(workspace);
    bool any_bias = false;
    int max_cout = 0;
    for (int j = 0; j < n; ++j) {
        const msda_wgrad_problem &p = problems[j];
        if (!p.dz || !p.x || !p.dw) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
        if ((reinterpret_cast<uintptr_t>(p.dz) | reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.dw)) & 15)
            return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
        grp.dz[j] = p.dz;
        grp.x[j] = p.x;
        grp.dw[j] = p.dw;
        grp.scale[j] = p.scale;
        grp.part[j] = grp.split[j] > 1 ? ws : p.dw;
        if (grp.split[j] > 1) ws += (int64_t)grp.split[j] * p.Cout * p.KH * p.KW * p.Cin;
        grp.dbias[j] = p.dbias;
        grp.bpart[j] = !p.dbias ? nullptr : (grp.split[j] > 1 ? ws : p.dbias);
        if (p.dbias && grp.split[j] > 1) ws += (int64_t)grp.split[j] * p.Cout;
        any_bias = any_bias || p.dbias != nullptr;
        max_cout = p.dbias && grp.split[j] > 1 && p.Cout > max_cout ? p.Cout : max_cout;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 ggrid((unsigned)grp.first[n]);
    if (!g_wgrad_ring.load())
        hipLaunchKernelGGL((conv_wgrad_group_kernel<false, true>), ggrid, dim3(kThreads), 0, st, grp);
    else if (any_bias)
        hipLaunchKernelGGL((conv_wgrad_group_kernel<true, true>), ggrid, dim3(kThreads), 0, st, grp);
    else
        hipLaunchKernelGGL((conv_wgrad_group_kernel<true, false>), ggrid, dim3(kThreads), 0, st, grp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const long long total = grp.red_first[n];
    if (total > 0 || max_cout > 0) {
        long long blocks = (total + 63) / 64 < 8192 ? (total + 63) / 64 : 8192;
        if (blocks < (max_cout + 63) / 64) blocks = (max_cout + 63) / 64;      // (the bias partials: 64 channels per workgroup)
        const int grid = (int)blocks;
        hipLaunchKernelGGL(conv_wgrad_group_reduce_kernel, dim3(grid), dim3(256), 0, st, grp);
        e = hipGetLastError();
    }
    return e == hipSuccess ? MSDA_OK : (int)e;
}

/* Tuning / tests: 1 (default) = the weight-gradient kernels prefetch their operand stages three ahead through an LDS ring (LDS DMA);
 * 0 = the register-staged form (one stage ahead).  Same products, same order: same results. */
int msda_conv_set_wgrad_ring(int on)
{
    if (on != 0 && on != 1) return msda_note_error(MSDA_ERR_BAD_OPTION, __func__);
    g_wgrad_ring = on;
    return MSDA_OK;
}

/* bytes of workspace msda_conv_wgrad_bf16 needs for a problem (0: none) */
int msda_conv_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t *bytes)
{
    if (!bytes) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (N < 1 || H < 1 || W < 1 || Cin < kBN || Cin % kBN != 0 || Cout < kBM || Cout % kBM != 0 || KH < 1 || KW < 1 || KH > 16 || KW > 16 ||
        stride < 1 || pad < 0)
        return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho < 1 || Wo < 1) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    WgradGeom g{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, (long long)N * Ho * Wo, 0, 0};
    long long split, chunk;
    wgrad_split(g, split, chunk);
    *bytes = split > 1 ? (int64_t)split * ((int64_t)Cout * KH * KW * Cin + Cout) * (int64_t)sizeof(float) : 0;      // (+ bias partials)
    return MSDA_OK;
}

int msda_conv_wgrad_bf16(const uint16_t *dz, const uint16_t *x, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                         int pad, float *dw, float *dbias, const float *scale, int torch_layout, void *workspace, msda_stream_t stream)
{
    if (!dz || !x || !dw) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (N < 1 || H < 1 || W < 1 || Cin < kBN || Cin % kBN != 0 || Cout < kBM || Cout % kBM != 0 || KH < 1 || KW < 1 || KH > 16 || KW > 16 ||
        stride < 1 || pad < 0)
        return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho < 1 || Wo < 1) return msda_note_error(MSDA_ERR_BAD_DIMS, __func__);
    if ((long long)N * H * W * Cin >= (1ll << 40) || (long long)N * Ho * Wo * Cout >= (1ll << 40)) return msda_note_error(MSDA_ERR_TOO_LARGE, __func__);
    if ((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dw)) & 15) return msda_note_error(MSDA_ERR_MISALIGNED, __func__);
    hipStream_t st = static_cast<hipStream_t>(stream);
    WgradGeom g{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, (long long)N * Ho * Wo, 0, torch_layout};
    if (g.P >= (1ll << 31) - (1 << 20) || (long long)N * H * W >= (1ll << 31)) return msda_note_error(MSDA_ERR_TOO_LARGE, __func__);      // 32-bit pixel counters
    long long split;
    wgrad_split(g, split, g.chunk);
    const long long blocks_y = (long long)KH * KW * (Cout / kBM) * (Cin / kBN);
    const long long n_dw = (long long)Cout * KH * KW * Cin;
    if (split > 1 && (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15))) return msda_note_error(MSDA_ERR_NULL_POINTER, __func__);
    if (blocks_y > 65535) return msda_note_error(MSDA_ERR_TOO_LARGE, __func__);
    float *ws_bias = split > 1 ? static_cast<float *>(workspace) + split * n_dw : nullptr;
    const dim3 grid((unsigned)split, (unsigned)blocks_y);
    float *part = split > 1 ? static_cast<float *>(workspace) : dw, *bpart = dbias ? (split > 1 ? ws_bias : dbias) : nullptr;
    if (!g_wgrad_ring.load())
        hipLaunchKernelGGL((conv_wgrad_kernel<false, true>), grid, dim3(kThreads), 0, st, dz, x, part, bpart, scale, split > 1 ? 0 : 1, g);
    else if (dbias)
        hipLaunchKernelGGL((conv_wgrad_kernel<true, true>), grid, dim3(kThreads), 0, st, dz, x, part, bpart, scale, split > 1 ? 0 : 1, g);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<true, false>), grid, dim3(kThreads), 0, st, dz, x, part, bpart, scale, split > 1 ? 0 : 1, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (split > 1) {
        const long long n4 = n_dw / 4;
        const int grid = (int)((n4 + 63) / 64 < 8192 ? (n4 + 63) / 64 : 8192);
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(grid), dim3(256), 0, st, static_cast<const float *>(workspace), dw, n4, (int)split,
                           dbias ? ws_bias : nullptr, dbias, Cout, scale, KH * KW, Cin, torch_layout);
    }
    e = hipGetLastError();
    return e == hipSuccess ? MSDA_OK : (int)e;
}

}  // extern "C"
