// msda_levelsum.h -- backward grad_value of WHOLE small levels, accumulated in LDS (gfx950, fp32).
//
// When a call sends many sampling points into a small map (decoder calls: 1092 queries x 4 points per head hit a
// 13x21, 25x42 or 50x84 level), the direct backward kernel spends its time in global float atomics that mostly collide
// on the same few rows.  This kernel takes such a level away from it: one workgroup owns (image, head, level, slice of
// kLsChan channels), keeps that whole slice of the level as an f64 window in LDS (ds_add_f64 is the native LDS float
// atomic on gfx950; ds_add_f32 is serialised, tools/lds_atomic_bench.hip), walks ALL queries of the call for that level
// and writes every pixel of its slice exactly once with plain stores -- no global atomics, no dependence on the
// zero-fill.  The f64 window also makes the sum exact to fp32 rounding whatever the order.
// Semantics: reference ms_deform_im2col_cuda.cuh:87-159 (the grad_value part), products formed in fp32 as there.
#pragma once

#include <atomic>

#include "msda_common.h"

namespace msda {

constexpr int kLsChan = 4;          // channels per slice = lanes per sampling point
constexpr int kLsThreads = 1024;
constexpr int kLsMaxLevels = 32;    // (level, row band) entries one launch can take
constexpr int kLsUnroll = 4;        // points in flight per lane group
constexpr int kLsLdsBudget = 150 * 1024;
inline std::atomic<int> &levelsum_lds_kb()   // window size: 150 KB = one workgroup per CU, 75 KB = two
{
    static std::atomic<int> kb{150};
    return kb;
}

struct LevelSumGeom {
    int N, S, M, D, L, Lq, P;
    int nlev, nslices;   // entries, channel slices
    int dbg;             // diagnostic (wrong results): 1 skip the walk, 2 skip the flush, 4 skip zeroing
    // entry = a band of rows [r0, r0 + nr) of level lev (the whole level when it fits LDS)
    int lev[kLsMaxLevels], H[kLsMaxLevels], W[kLsMaxLevels], start[kLsMaxLevels], r0[kLsMaxLevels], nr[kLsMaxLevels];
};

// Which levels of a call this kernel should take (bit l of the result), and the launch geometry for them.
// First choice: ALL levels -- a level too large for LDS is cut into row bands, each band one window -- because the direct
// backward kernel then issues no global atomic at all, needs no zero-fill and can keep 4 channels per lane.  That is
// taken when every workgroup can afford to walk all Lq*P points of its level: cheap for decoder-shaped calls, and for
// encoder-sized calls with scattered sampling points (the direct path's job) still twice as fast as 2.9 GB of row atomics
// (MI355X, call E: 2248 -> 1142 us on uniform-random locations).
// Otherwise: only the levels that fit LDS whole and receive at least two sampling points per pixel (there is something
// to merge); the rest stays with the direct kernel's row atomics.
inline unsigned plan_levelsum(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes,
                              const int64_t *lsi, LevelSumGeom &g, size_t &lds_bytes)
{
    lds_bytes = 0;
    if (L > 32 || D > 128 || (int64_t)Lq * P > 1048576) { g = LevelSumGeom{}; return 0; }
    const size_t px_bytes = (size_t)kLsChan * sizeof(double);
    const size_t budget = std::min<size_t>(kLsLdsBudget, (size_t)std::max(8, levelsum_lds_kb().load()) * 1024);
    for (int all = 1; all >= 0; --all) {
        g = LevelSumGeom{};
        g.N = N; g.S = S; g.M = M; g.D = D; g.L = L; g.Lq = Lq; g.P = P;
        g.nslices = (D + kLsChan - 1) / kLsChan;
        lds_bytes = 0;
        unsigned mask = 0;
        bool ok = true;
        for (int l = 0; l < L && ok; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const int64_t px = (int64_t)H * W;
            const bool whole = px * px_bytes <= budget;
            if (!all && (!whole || (int64_t)Lq * P < 2 * px)) continue;
            const int rows_max = (int)(budget / (px_bytes * W));   // rows of this level one window holds
            if (rows_max < 1) { ok = false; break; }
            const int bands = whole ? 1 : (H + rows_max - 1) / rows_max;
            const int rows = (H + bands - 1) / bands;
            if (bands > 16 || g.nlev + bands > kLsMaxLevels) { ok = false; break; }
            for (int r0 = 0; r0 < H; r0 += rows) {
                const int e = g.nlev++;
                g.lev[e] = l; g.H[e] = H; g.W[e] = W; g.start[e] = (int)lsi[l];
                g.r0[e] = r0;
                g.nr[e] = H - r0 < rows ? H - r0 : rows;
                const size_t bytes = (size_t)g.nr[e] * W * px_bytes;
                lds_bytes = bytes > lds_bytes ? bytes : lds_bytes;
            }
            mask |= 1u << l;
        }
        if (all && (!ok || mask != (L >= 32 ? ~0u : (1u << L) - 1))) continue;   // fall back to the dense-levels rule
        return mask;
    }
    g = LevelSumGeom{};
    return 0;
}

inline int levelsum_grid(const LevelSumGeom &g)
{
    return kXcds * ((g.N * g.M + kXcds - 1) / kXcds) * g.nlev * g.nslices;
}

// P4: exactly four points per level (RichSem) -- their locations / weights are loaded as whole vectors.
// TV: storage type of grad_out / grad_value (float, or bf16_t: the f64 window is rounded to bf16 once, at the store).
template <bool P4, typename TV = float>
__global__ __launch_bounds__(kLsThreads) void bwd_levelsum_kernel(const float *__restrict__ loc,
                                                                  const float *__restrict__ aw,
                                                                  const TV *__restrict__ grad_out,
                                                                  TV *__restrict__ grad_value, const LevelSumGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *win = reinterpret_cast<double *>(smem);

    int pair, t;
    if (!decode_block(blockIdx.x, g.N * g.M, g.nlev * g.nslices, pair, t)) return;
    // the slices of one level are neighbours in the XCD-local order: they read the same loc / attn lines
    // (entries are taken LAST FIRST -- round 5: the coarse levels, whose single window takes every point of the call, are the long workgroups;
    // launched behind the 512 short row-band workgroups of level 0 they ran as a half-empty last round)
    const int li = g.nlev - 1 - t / g.nslices, slice = t - (t / g.nslices) * g.nslices;
    const int b = pair / g.M, m = pair - b * g.M;
    const int l = g.lev[li], H = g.H[li], W = g.W[li], r0 = g.r0[li], nr = g.nr[li];
    const int npx = nr * W;   // the window: rows [r0, r0 + nr) of the level
    const int tid = threadIdx.x;
    const int j = tid & (kLsChan - 1), grp = tid / kLsChan;
    constexpr int kGroups = kLsThreads / kLsChan;
    const int ch = slice * kLsChan + j;
    const bool has_ch = ch < g.D;

    if (!(g.dbg & 4)) for (int e = tid; e < npx * kLsChan; e += kLsThreads) win[e] = 0.0;
    __syncthreads();

    const int LP = g.L * g.P;
    // query-major walk: a lane group takes a query and its P points of this level (kLsUnroll at a time, all loads first) -- no
    // index divisions in the loop, grad_out read once per query
    for (int q = grp; q < ((g.dbg & 1) ? 0 : g.Lq); q += kGroups) {
        const unsigned item = (unsigned)((b * g.Lq + q) * g.M + m);
        const unsigned pt0 = item * (unsigned)LP + (unsigned)(l * g.P);
        const float go = has_ch ? ld1(grad_out + item * (unsigned)g.D + ch) : 0.f;
        if (P4) {
            // Lane j of the group resolves point j ONCE and the group shares it by quad broadcasts (DPP); each lane then adds
            // its own channel of the four corners -- the four lanes of a group still hit four consecutive doubles.
            static_assert(kLsChan == 4, "one quad per query");
            const float2 xy = *reinterpret_cast<const float2 *>(loc + 2u * (pt0 + (unsigned)j));
            const float a = aw[pt0 + (unsigned)j];
            const float h_im = xy.y * (float)H - 0.5f, w_im = xy.x * (float)W - 0.5f;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)hf, w_low = (int)wf;
            const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
            const bool top = h_low >= r0 && h_low < r0 + nr, bot = h_low + 1 >= r0 && h_low + 1 < r0 + nr;
            const bool alive = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W && (top || bot);
            const bool lef = w_low >= 0, rig = w_low + 1 <= W - 1;
            const int rt = (top ? h_low : h_low + 1) - r0, rb = (bot ? h_low + 1 : h_low) - r0;
            const int cl = lef ? w_low : w_low + 1, cr = rig ? w_low + 1 : w_low;
            // element offsets of the four corners (clamped to valid addresses) and their weights (0 for a corner that does not count)
            const int my_o[4] = {(rt * W + cl) * kLsChan, (rt * W + cr) * kLsChan, (rb * W + cl) * kLsChan, (rb * W + cr) * kLsChan};
            const float my_w[4] = {top && lef ? hh * hw : 0.f, top && rig ? hh * lw : 0.f, bot && lef ? lh * hw : 0.f, bot && rig ? lh * lw : 0.f};
#define LS_BCAST_I(V, K) __builtin_amdgcn_update_dpp(0, (V), (K) * 0x55, 0xF, 0xF, true)
#define LS_BCAST_F(V, K) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(V), (K) * 0x55, 0xF, 0xF, true))
#define LS_POINT(K)                                                                                     \
    if (LS_BCAST_I((int)alive, K)) {                                                                    \
        const float ga_ = go * LS_BCAST_F(a, K); /* top_grad * attn_weight (ms_deform_im2col_cuda.cuh:117) */ \
        atomicAdd(win + LS_BCAST_I(my_o[0], K) + j, (double)(LS_BCAST_F(my_w[0], K) * ga_));            \
        atomicAdd(win + LS_BCAST_I(my_o[1], K) + j, (double)(LS_BCAST_F(my_w[1], K) * ga_));            \
        atomicAdd(win + LS_BCAST_I(my_o[2], K) + j, (double)(LS_BCAST_F(my_w[2], K) * ga_));            \
        atomicAdd(win + LS_BCAST_I(my_o[3], K) + j, (double)(LS_BCAST_F(my_w[3], K) * ga_));            \
    }
            LS_POINT(0)
            LS_POINT(1)
            LS_POINT(2)
            LS_POINT(3)
#undef LS_POINT
#undef LS_BCAST_I
#undef LS_BCAST_F
            continue;
        }
        for (int p0 = 0; p0 < g.P; p0 += kLsUnroll) {
            float x[kLsUnroll], y[kLsUnroll], ga[kLsUnroll];
            {
#pragma unroll
                for (int u = 0; u < kLsUnroll; ++u) {
                    const bool live = p0 + u < g.P;
                    const unsigned pt = pt0 + (unsigned)(live ? p0 + u : 0);
                    const float2 xy = *reinterpret_cast<const float2 *>(loc + 2u * pt);
                    const float a = aw[pt];
                    x[u] = live ? xy.x : -4.f;            // (-4: dropped by the range test below)
                    y[u] = xy.y;
                    ga[u] = go * a;                       // top_grad * attn_weight (ms_deform_im2col_cuda.cuh:117)
                }
            }
#pragma unroll
            for (int u = 0; u < kLsUnroll; ++u) {
                const float h_im = y[u] * (float)H - 0.5f, w_im = x[u] * (float)W - 0.5f;
                if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) continue;
                const float hf = floorf(h_im), wf = floorf(w_im);
                const int h_low = (int)hf, w_low = (int)wf;
                const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                // a corner counts if it is inside the map AND inside this workgroup's row band.  One branch for "not in this
                // band at all"; after it every corner is added unconditionally -- a corner that does not count adds 0.0 to a
                // clamped (valid) address, which costs less than four more branches
                const bool top = h_low >= r0 && h_low < r0 + nr, bot = h_low + 1 >= r0 && h_low + 1 < r0 + nr;
                if (!(top || bot)) continue;
                const bool lef = w_low >= 0, rig = w_low + 1 <= W - 1;
                const int rt = (top ? h_low : h_low + 1) - r0, rb = (bot ? h_low + 1 : h_low) - r0;
                const int cl = lef ? w_low : w_low + 1, cr = rig ? w_low + 1 : w_low;
                double *pt_ = win + (rt * W) * kLsChan + j, *pb_ = win + (rb * W) * kLsChan + j;
                atomicAdd(pt_ + cl * kLsChan, (double)(top && lef ? hh * hw * ga[u] : 0.f));
                atomicAdd(pt_ + cr * kLsChan, (double)(top && rig ? hh * lw * ga[u] : 0.f));
                atomicAdd(pb_ + cl * kLsChan, (double)(bot && lef ? lh * hw * ga[u] : 0.f));
                atomicAdd(pb_ + cr * kLsChan, (double)(bot && rig ? lh * lw * ga[u] : 0.f));
            }
        }
    }
    __syncthreads();

    // every pixel of the slice, once: 16 B per lane group
    TV *dst = grad_value + ((int64_t)(b * g.S + g.start[li] + r0 * W) * g.M + m) * g.D + ch;
    if (has_ch && !(g.dbg & 2))
        for (int px = grp; px < npx; px += kGroups) dst[(int64_t)px * g.M * g.D] = to_storage<TV, float>((float)win[px * kLsChan + j]);
}

}  // namespace msda
