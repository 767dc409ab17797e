// msda_tiled.h -- LDS-window MSDeformAttn kernels for gfx950, for encoder-shaped calls
// (queries = the pixels of the pyramid, Lq == S; D = 32 fp32; L <= 4): the shape that carries
// >95 % of the path's bytes (call "E").
//
// Why: in an encoder call every query samples around its own position, at every level.  The direct
// kernels fetch each of the 64 bilinear corners of a (query, head) from L2 (2.9 GB of 128-B row
// requests per call, L2-rate bound at ~180 us) and add each corner's gradient with a global float
// atomic (2.9 GB of atomics, chip-rate bound at ~2.3 ms).  Here a workgroup owns ONE image REGION
// of ONE (image, head) pair:
//   * its queries are the pixels of every level whose centre falls in the region (~16x16 level-0
//     pixels -> ~340 queries), so they share their sampling neighbourhoods;
//   * per level it stages the window (region footprint +- margin) of the value slice in LDS --
//     128 B per pixel, filled with whole-row 16-B loads -- and gathers the corners from LDS;
//   * backward accumulates grad_value into an LDS window with LDS float atomics and flushes each
//     touched pixel ONCE per region with a 128-B-row global atomic, instead of once per corner.
// The window is only a cache: every corner is tested against it and corners outside (large learned
// offsets, samples near the map border) fall back to global loads / global atomics, so results
// are exact for ARBITRARY sampling locations; only speed depends on locality.
// The query <-> region assignment is pure geometry on the level sizes; it needs the call to be
// encoder-shaped only to be profitable, never to be correct.
//
// Semantics: identical to msda_direct.h (spec: reference ms_deform_im2col_cuda.cuh:33-159, 237-403).
#pragma once

#include <atomic>
#include <map>
#include <mutex>
#include <utility>

#include "msda_common.h"

namespace msda {

constexpr int kTL = 4;               // levels supported by the tiled kernels
constexpr int kTD = 32;              // channels per head
constexpr int kMaxRegionQueries = 512;
constexpr int kMaxGrid = 24;         // region rows / columns covered by the host-made geometry tables
// Gather kernels (forward; backward location / attention gradients): a workgroup works on ONE CHANNEL HALF at a time
// -- 16 channels = 64 B per window pixel -- so that its LDS stays under half a CU's and two workgroups share a CU: one
// waits on its window fill (HBM-paced) while the other gathers.
constexpr int kFwdGC = 16;           // forward: channels per workgroup (a channel half), 4 lanes per query, 512 threads
constexpr int kFwdLdsBudget = 75 * 1024;   // + header < 80 KiB: two workgroups per CU
// Backward location / attention gradients need all 32 channels of a sample at once: 8 lanes per query, 1024 threads,
// one workgroup per CU (two 16-channel passes with a read-modify-write of the per-point results measured slower).
constexpr int kBwdGC = 32;
constexpr int kBwdCPL = 8;            // backward gather: channels per lane (4 lanes per query)
constexpr int kBwdLdsBudget = 152 * 1024;   // region 20 fits its finest-level window in one phase
// ... alternatively on channel halves like the forward (two workgroups per CU, the first half's per-point results kept
// in registers): needs one level per work item and P <= 4 (tile option "bwd_gather_halves")   // three phases = three independent workgroups per region (measured faster than two)
// Lane layout of the gather kernels: GC channels per workgroup pass, CPL channels per lane (4 or 8), so GL = GC/CPL lanes
// per query; QPG queries per lane group so that one pass over k covers the largest region (kMaxRegionQueries).
template <int GC, int CPL>
struct GatherCfg {
    static constexpr int GL = GC / CPL;
    static constexpr int NV = CPL / 4;                       // float4 vectors per lane
    static constexpr int kThreads = GC == 16 ? 512 : 1024;
    static constexpr int kGroups = kThreads / GL;            // queries in flight per pass over k
    static constexpr int QPG = 512 / kGroups;                // = kMaxRegionQueries / kGroups
};
// Scatter kernel (backward grad_value): 16 channels x f64 = 128 B per window pixel, one workgroup per CU.
constexpr int kTiledThreads = 1024;
constexpr int kSD = 16;              // channels per scatter workgroup (two workgroups per region: channel halves)
constexpr int kScatterGroups = kTiledThreads / kSD;   // 16-lane groups, one query each
constexpr int kLdsBudgetBytes = 152 * 1024;           // windows; the header takes the rest
constexpr int kScatterBatch = 2;                      // queries per group whose operands are fetched together

struct TiledGeom {
    int N, S, M, Lq, L, P;
    int GY, GX;            // region grid over the normalised image plane
    int margin;            // window margin, in pixels of the sampled level (base value)
    int margin_l[kTL];     // per level, after growing into unused LDS
    int H[kTL], W[kTL], start[kTL];
    int phase[kTL];        // levels are processed in phases; the windows of one phase share the LDS
    int nphases;
    // per level and region row / column, precomputed on the host (level_rect): first query row, window origin, window
    // extent -- the workgroup's header is table look-ups instead of ~30 integer divisions
    short rq0[kTL][kMaxGrid + 1], rw0[kTL][kMaxGrid], rwn[kTL][kMaxGrid];
    short cq0[kTL][kMaxGrid + 1], cw0[kTL][kMaxGrid], cwn[kTL][kMaxGrid];
    unsigned long long *stamps;   // diagnostic builds of a run only: per-workgroup s_memtime stamps (16 per workgroup), or null
    unsigned *stats;              // forward, optional: stats[0] += points that missed their window (locality monitor), or null
    int dbg;               // diagnostic: bits 4..5 select the kernel that writes stage stamps (0 = all, 1 = scatter, 2 = gather)
};

// ---- region geometry (host and device) --------------------------------------------------------------
// Level-l pixel row r belongs to region row gy iff its centre (r+0.5)/H lies in [gy/GY, (gy+1)/GY):
// first row of region gy:
__host__ __device__ inline int region_first(int H, int gy, int GY) { return (2 * H * gy + GY - 1) / (2 * GY); }

__host__ __device__ inline int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

struct LevelRect {
    int qr0, qc0, qnr, qnc;   // queries of this level inside the region
    int wr0, wc0, nwr, nwc;   // window (inclusive origin, size) of this level
};

__host__ __device__ inline LevelRect level_rect(int H, int W, int gy, int gx, int GY, int GX, int margin)
{
    LevelRect r;
    r.qr0 = region_first(H, gy, GY);
    r.qnr = region_first(H, gy + 1, GY) - r.qr0;
    r.qc0 = region_first(W, gx, GX);
    r.qnc = region_first(W, gx + 1, GX) - r.qc0;
    // reference positions of the region, in level pixel coordinates: [g*H/G - 0.5, (g+1)*H/G - 0.5)
    // The window may reach one pixel beyond the map on every side (rows -1 and H, columns -1 and W): that apron is
    // zero-filled, so a sample whose corners straddle the map border -- frequent on the coarse levels -- still takes
    // the in-window path and reads the zero the bilinear kernel prescribes for outside corners.
    int lo = floor_div(2 * gy * H - GY, 2 * GY) - margin;
    int hi = floor_div(2 * (gy + 1) * H - GY, 2 * GY) + 1 + margin;
    lo = lo < -1 ? -1 : lo;
    hi = hi > H ? H : hi;
    r.wr0 = lo;
    r.nwr = hi - lo + 1;
    lo = floor_div(2 * gx * W - GX, 2 * GX) - margin;
    hi = floor_div(2 * (gx + 1) * W - GX, 2 * GX) + 1 + margin;
    lo = lo < -1 ? -1 : lo;
    hi = hi > W ? W : hi;
    r.wc0 = lo;
    r.nwc = hi - lo + 1;
    return r;
}

// Per-workgroup header at the start of LDS: the region's geometry and its query list.  Kept in LDS (not in
// registers) so that the level loop can stay a run-time loop; fields are re-read as wave-uniform scalars.
struct TileHeader {
    LevelRect r[kTL];
    int qpre[kTL + 1];   // prefix sums of the per-level query counts
    int lds_px[kTL];     // first LDS pixel of the level's window inside its phase
    int H[kTL], W[kTL], start[kTL], phase[kTL];
    int pad[3];
    int qid[kMaxRegionQueries];   // global query index of the region's i-th query
    float gmax[kMaxRegionQueries];   // integer-accumulation scatter: max_c |grad_out[q, c]| of the region's i-th query
};
static_assert(sizeof(TileHeader) % 16 == 0, "windows must stay 16-byte aligned behind the header");

// Set by msda_set_option (any thread), read by every launch (any thread): plain atomics, like the other options.
struct TiledOptions {
    std::atomic<int> region_px{20};   // finest-level pixels per region side (swept on MI355X: 20 beats 16 by ~15 %; larger does not fit LDS)
    std::atomic<int> margin{6};
    std::atomic<int> bwd_halves{0};   // backward location/attention gradients: 0 = 32 channels at once (faster as measured), 1 = two 16-channel passes
    std::atomic<int> persist{512};    // 0 = one workgroup per work item; n > 0 = at most n workgroups (n/2 for the 1024-thread kernels)
                                      // walking the items (2 x 256 CUs by default: no per-item launch ramp)
    std::atomic<int> grow{1};         // 1 = windows grow into the LDS their phase leaves unused (per-level margins)
    std::atomic<int> accum{2};        // grad_value: 0 = f64 LDS atomic window, 1 = integer block-floating-point window, 2 = sorted reduction
    std::atomic<int> dbg{0};
    std::atomic<unsigned long long *> stamps{nullptr};
    std::atomic<unsigned *> stats{nullptr};   // diagnostic override of the locality counter (msda_debug_stats)
};
inline TiledOptions &tiled_options()
{
    static TiledOptions o;
    return o;
}

// ---- host planner ------------------------------------------------------------------------------------
struct TiledPlan {
    bool ok = false;
    TiledGeom g{};
    size_t lds_bytes = 0;
    int grid = 0;
    int max_px = 0;   // integer-accumulation scatter: largest single-level window
    int max_q = 0;    // most queries in one region
};

inline TiledPlan plan_tiled(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes, const int64_t *lsi,
                            int region_px, int margin, int budget_bytes, int px_bytes, int per_query_bytes = 0)
{
    TiledPlan pl;
    if (D != kTD || L > kTL || L < 1 || Lq != S || P < 1 || L * P > 16) return pl;
    int64_t pre = 0;
    for (int l = 0; l < L; ++l) {   // queries must be exactly the pixels: levels tile [0,S) in order
        if (lsi[l] != pre) return pl;
        pre += shapes[2 * l] * shapes[2 * l + 1];
    }
    if (pre != S) return pl;
    TiledGeom &g = pl.g;
    g.N = N; g.S = S; g.M = M; g.Lq = Lq; g.L = L; g.P = P;
    g.margin = margin;
    g.dbg = tiled_options().dbg;
    g.stamps = tiled_options().stamps;
    g.stats = tiled_options().stats;
    int Hmax = 0, Wmax = 0;
    for (int l = 0; l < L; ++l) {
        g.H[l] = (int)shapes[2 * l];
        g.W[l] = (int)shapes[2 * l + 1];
        g.start[l] = (int)lsi[l];
        Hmax = g.H[l] > Hmax ? g.H[l] : Hmax;
        Wmax = g.W[l] > Wmax ? g.W[l] : Wmax;
    }
    int cap_px = budget_bytes / px_bytes;
    // largest window of level l over all regions, for a given margin
    auto worst_window = [&g](int l, int mg) {
        int worst = 0;
        for (int gy = 0; gy < g.GY; ++gy)
            for (int gx = 0; gx < g.GX; ++gx) {
                const LevelRect r = level_rect(g.H[l], g.W[l], gy, gx, g.GY, g.GX, mg);
                worst = r.nwr * r.nwc > worst ? r.nwr * r.nwc : worst;
            }
        return worst;
    };
    // region grid: ~region_px pixels of the finest level per side; refine until queries and windows fit
    for (int rp = region_px; rp >= 4; rp -= 2) {
        g.GY = (Hmax + rp - 1) / rp;
        g.GX = (Wmax + rp - 1) / rp;
        if (g.GY > kMaxGrid || g.GX > kMaxGrid || Hmax >= 32768 || Wmax >= 32768) return pl;
        int max_q = 0, max_win[kTL] = {0, 0, 0, 0};
        for (int gy = 0; gy < g.GY; ++gy)
            for (int gx = 0; gx < g.GX; ++gx) {
                int nq = 0;
                for (int l = 0; l < L; ++l) {
                    const LevelRect r = level_rect(g.H[l], g.W[l], gy, gx, g.GY, g.GX, margin);
                    nq += r.qnr * r.qnc;
                }
                max_q = nq > max_q ? nq : max_q;
            }
        for (int l = 0; l < L; ++l) max_win[l] = worst_window(l, margin);
        cap_px = (budget_bytes - max_q * per_query_bytes) / px_bytes;   // LDS also holds per-query data in some kernels
        bool fits = max_q <= kMaxRegionQueries && cap_px > 0;
        for (int l = 0; l < L; ++l) fits = fits && max_win[l] <= cap_px;
        if (!fits) continue;
        // greedy phases: consecutive levels share the LDS while their worst-case windows fit together
        int ph = 0, used = 0, max_phase_px = 0;
        for (int l = 0; l < L; ++l) {
            if (used + max_win[l] > cap_px) { ++ph; used = 0; }
            g.phase[l] = ph;
            used += max_win[l];
            max_phase_px = used > max_phase_px ? used : max_phase_px;
        }
        for (int l = L; l < kTL; ++l) { g.phase[l] = -1; g.H[l] = g.W[l] = 1; g.start[l] = 0; }
        g.nphases = ph + 1;
        // per-level margins.  With `grow` the LDS a phase leaves unused widens the windows of its levels, coarsest
        // first (a pixel of margin is cheapest there and the queries of a coarse level sit closest to their region's
        // edge), until the window is the whole map: fewer points on the general path when the offsets reach far.
        int mg[kTL];
        for (int l = 0; l < kTL; ++l) mg[l] = margin;
        if (tiled_options().grow) {
            for (int p = 0; p <= ph; ++p) {
                int used_p = 0;
                for (int l = 0; l < L; ++l) used_p += g.phase[l] == p ? max_win[l] : 0;
                bool grew = true;
                while (grew) {
                    grew = false;
                    for (int l = L - 1; l >= 0; --l) {
                        if (g.phase[l] != p || mg[l] >= 64) continue;
                        const int w = worst_window(l, mg[l] + 1);
                        if (w == max_win[l] || used_p - max_win[l] + w > cap_px) continue;   // whole map / no room
                        used_p += w - max_win[l];
                        max_win[l] = w;
                        ++mg[l];
                        grew = true;
                    }
                }
                max_phase_px = used_p > max_phase_px ? used_p : max_phase_px;
            }
        }
        for (int l = 0; l < kTL; ++l) g.margin_l[l] = mg[l];
        for (int l = 0; l < kTL; ++l) {
            const int margin = mg[l];
            for (int gy = 0; gy <= g.GY; ++gy) {
                const LevelRect r = level_rect(g.H[l], g.W[l], gy < g.GY ? gy : g.GY - 1, 0, g.GY, g.GX, margin);
                g.rq0[l][gy] = (short)(gy < g.GY ? r.qr0 : r.qr0 + r.qnr);
                if (gy < g.GY) { g.rw0[l][gy] = (short)r.wr0; g.rwn[l][gy] = (short)r.nwr; }
            }
            for (int gx = 0; gx <= g.GX; ++gx) {
                const LevelRect r = level_rect(g.H[l], g.W[l], 0, gx < g.GX ? gx : g.GX - 1, g.GY, g.GX, margin);
                g.cq0[l][gx] = (short)(gx < g.GX ? r.qc0 : r.qc0 + r.qnc);
                if (gx < g.GX) { g.cw0[l][gx] = (short)r.wc0; g.cwn[l][gx] = (short)r.nwc; }
            }
        }
        pl.lds_bytes = sizeof(TileHeader) + (size_t)max_phase_px * px_bytes + (size_t)max_q * per_query_bytes;
        pl.max_q = max_q;
        pl.grid = kXcds * ((N * M + kXcds - 1) / kXcds) * g.GY * g.GX;
        pl.ok = true;
        return pl;
    }
    return pl;
}

// ---- device helpers -------------------------------------------------------------------------------------
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Diagnostic only (g.stamps is null in normal runs): slot i of this workgroup's stamp row <- shader clock.
template <int KERNEL>   // 1 = scatter, 2 = gather; dbg bits 4..5 select one kernel (0 = all)
__device__ __forceinline__ void stamp(const TiledGeom &g, int i)
{
    if (g.stamps && threadIdx.x == 0 && (((g.dbg >> 4) & 3) == 0 || ((g.dbg >> 4) & 3) == KERNEL)) {
        const unsigned long long wg = blockIdx.x + (unsigned long long)gridDim.x * (blockIdx.y + (unsigned long long)gridDim.y * blockIdx.z);
        g.stamps[wg * 16 + i] = __builtin_amdgcn_s_memtime();
    }
}

// Builds the header (all threads of the workgroup must call it).  Returns the number of queries.
__device__ __forceinline__ int build_header(TileHeader *h, const TiledGeom &g, int gy, int gx)
{
    const int l = threadIdx.x;
    if (l < kTL) {   // one lane per level; geometry comes from the host-made tables (uniform row / column index)
        int H = 1, W = 1, st = 0, ph = -1;
        LevelRect r = LevelRect{0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
        for (int i = 0; i < kTL; ++i)
            if (i == l) {
                H = g.H[i]; W = g.W[i]; st = g.start[i]; ph = g.phase[i];
                r.qr0 = g.rq0[i][gy];
                r.qnr = g.rq0[i][gy + 1] - r.qr0;
                r.qc0 = g.cq0[i][gx];
                r.qnc = g.cq0[i][gx + 1] - r.qc0;
                r.wr0 = g.rw0[i][gy];
                r.nwr = g.rwn[i][gy];
                r.wc0 = g.cw0[i][gx];
                r.nwc = g.cwn[i][gx];
            }
        if (l >= g.L) r = LevelRect{0, 0, 0, 1, 0, 0, 0, 1};
        h->r[l] = r;
        h->H[l] = H;
        h->W[l] = W;
        h->start[l] = st;
        h->phase[l] = l < g.L ? ph : -1;
    }
    __syncthreads();
    // every thread derives the prefix sums from one batch of (pipelined) LDS reads -- a single thread walking the levels
    // with dependent LDS round trips cost ~3k cycles per workgroup
    int qpre[kTL + 1], ldspx[kTL];
    {
        int qn[kTL], wn[kTL], phs[kTL];
#pragma unroll
        for (int i = 0; i < kTL; ++i) {
            qn[i] = h->r[i].qnr * h->r[i].qnc;
            wn[i] = h->r[i].nwr * h->r[i].nwc;
            phs[i] = h->phase[i];
        }
        qpre[0] = 0;
        int used = 0, cur = 0;
#pragma unroll
        for (int i = 0; i < kTL; ++i) {
            const bool on = i < g.L;
            qpre[i + 1] = qpre[i] + (on ? qn[i] : 0);
            if (on && phs[i] != cur) { cur = phs[i]; used = 0; }
            ldspx[i] = used;
            if (on) used += wn[i];
        }
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < kTL; ++i) {
            h->qpre[i] = qpre[i];
            h->lds_px[i] = ldspx[i];
        }
        h->qpre[kTL] = qpre[kTL];
        h->pad[0] = 0;   // forward: number of queries with points that missed their window (tiled_gather_kernel)
    }
    const int nq = qpre[kTL];
    for (int i = threadIdx.x; i < nq; i += blockDim.x) {   // level-major, row-major inside the level
        int q = -1;
#pragma unroll
        for (int lv = 0; lv < kTL; ++lv) {
            if (i >= qpre[lv] && i < qpre[lv + 1]) {
                const int k = i - qpre[lv];
                const int qnc = h->r[lv].qnc;
                const int rr = k / qnc, cc = k - rr * qnc;
                q = h->start[lv] + (h->r[lv].qr0 + rr) * h->W[lv] + h->r[lv].qc0 + cc;
            }
        }
        h->qid[i] = q;
    }
    __syncthreads();
    return uni(nq);
}

// DPP quad broadcast: every lane of a quad receives the value held by lane `SRC` of its quad.
template <int SRC>
__device__ __forceinline__ int quad_bcast_i(int v)
{
    return __builtin_amdgcn_mov_dpp(v, SRC | (SRC << 2) | (SRC << 4) | (SRC << 6), 0xF, 0xF, true);
}
template <int SRC>
__device__ __forceinline__ float quad_bcast_f(float v)
{
    return __int_as_float(quad_bcast_i<SRC>(__float_as_int(v)));
}

// Wave-uniform description of one sampled level inside the current region.
struct LevelCtx {
    int H, W, wr0, wc0, nwr, nwc, lds_base /* float index of the window */, base_row /* element offset of value[b, start, m, 0] */;
};

// ---- forward, and the location / attention gradients of backward ------------------------------------------
// 8 lanes x 4 channels per query; lane i of each quad resolves sampling point pc+i and the quad shares it by
// DPP broadcast.  BWD = false: out.  BWD = true: grad_loc, grad_attn (grad_value is the scatter kernel's job).
typedef float v2f __attribute__((ext_vector_type(2)));

// Sum over the 4 lanes of a query (one quad) with DPP only: no LDS crossbar, no barrier.
__device__ __forceinline__ float quad_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}

// The four corner rows of an IN-WINDOW sampling point for this lane's 4*NV channels: straight-line LDS reads.
template <int GC, int NV>
__device__ __forceinline__ void lds_corners(const float *win, int nwc, int j, int mode, float4 (&v)[4][NV])
{
    const float *p = win + mode + 4 * NV * j;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        v[0][n] = *reinterpret_cast<const float4 *>(p + 4 * n);
        v[1][n] = *reinterpret_cast<const float4 *>(p + GC + 4 * n);
        v[2][n] = *reinterpret_cast<const float4 *>(p + nwc * GC + 4 * n);
        v[3][n] = *reinterpret_cast<const float4 *>(p + nwc * GC + GC + 4 * n);
    }
}

// Sum over the GL lanes of a query: the quad, plus the neighbouring quad (row_half_mirror) when a query spans 8 lanes.
template <int GL>
__device__ __forceinline__ float query_sum(float v)
{
    v = quad_sum(v);
    if (GL == 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    return v;
}

// General point (a corner outside the window): corners from global memory, zero outside the map.
// `chan` = first channel of this lane inside the head.
template <int NV>
__device__ __forceinline__ void global_corners(const float *__restrict__ value, const LevelCtx &lc, int row_elems, int chan,
                                               float x, float y, float4 (&v)[4][NV])
{
    int o[4];
    float lh2, lw2;
    resolve_point<float>(x, y, lc.H, lc.W, lc.base_row, row_elems, o, lh2, lw2);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < NV; ++n)
            v[c][n] = o[c] >= 0 ? *reinterpret_cast<const float4 *>(value + o[c] + chan + 4 * n) : z;
}

__device__ __forceinline__ void fwd_accumulate(float w1, float w2, float w3, float w4, const float4 &v1, const float4 &v2,
                                               const float4 &v3, const float4 &v4, v2f &acc_lo, v2f &acc_hi)
{
    acc_lo += w1 * (v2f){v1.x, v1.y} + w2 * (v2f){v2.x, v2.y} + w3 * (v2f){v3.x, v3.y} + w4 * (v2f){v4.x, v4.y};
    acc_hi += w1 * (v2f){v1.z, v1.w} + w2 * (v2f){v2.z, v2.w} + w3 * (v2f){v3.z, v3.w} + w4 * (v2f){v4.z, v4.w};
}

// Backward location / attention gradients of one point are three channel sums that are all linear in the four
// "corner dots"  D_k = sum_c grad_out[c] * v_k[c]:
//   grad_attn = w1*D1 + w2*D2 + w3*D3 + w4*D4,  d/dlw = hh*(D2-D1) + lh*(D4-D3),  d/dlh = hw*(D3-D1) + lw*(D4-D2)
// so a lane only forms the four dots over its 4 channels (8 packed FMAs); the dots are summed over the query's lanes and
// combined with the bilinear coefficients afterwards.
template <int NV>
__device__ __forceinline__ void corner_dots(const float4 (&gq)[NV], const float4 (&v)[4][NV], float &d1, float &d2, float &d3,
                                            float &d4)
{
    v2f t[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        t[c] = (v2f){gq[0].x, gq[0].y} * (v2f){v[c][0].x, v[c][0].y} + (v2f){gq[0].z, gq[0].w} * (v2f){v[c][0].z, v[c][0].w};
#pragma unroll
        for (int n = 1; n < NV; ++n)
            t[c] += (v2f){gq[n].x, gq[n].y} * (v2f){v[c][n].x, v[c][n].y} + (v2f){gq[n].z, gq[n].w} * (v2f){v[c][n].z, v[c][n].w};
    }
    d1 = t[0].x + t[0].y;
    d2 = t[1].x + t[1].y;
    d3 = t[2].x + t[2].y;
    d4 = t[3].x + t[3].y;
}

__device__ __forceinline__ void combine_dots(float lh, float lw, float d1, float d2, float d3, float d4, float &s_a,
                                             float &s_w, float &s_h)
{
    const float hh = 1.f - lh, hw = 1.f - lw;
    s_a = hh * hw * d1 + hh * lw * d2 + lh * hw * d3 + lh * lw * d4;
    s_w = hh * (d2 - d1) + lh * (d4 - d3);
    s_h = hw * (d3 - d1) + lw * (d4 - d2);
}

// This lane's sampling point (lane i of the quad holds point i of the level's first four) for each of the quad's
// queries.  Loaded one level AHEAD of its use, so the global-memory latency hides behind the window fill / the
// previous level's gather.
template <int QPG>
struct LevelOps {
    float2 xy[QPG];
    float a[QPG];
};

template <int QPG>
__device__ __forceinline__ void load_level_ops(const float *__restrict__ loc, const float *__restrict__ aw,
                                               const unsigned (&item)[QPG], unsigned LP, unsigned lvl_pt0, int P, int j,
                                               LevelOps<QPG> &o)
{
    const unsigned mp = lvl_pt0 + ((j & 3) < P ? (j & 3) : 0);
#pragma unroll
    for (int k = 0; k < QPG; ++k) {   // queries without a slot read a valid address and are masked later
        o.xy[k] = *reinterpret_cast<const float2 *>(loc + 2u * (item[k] * LP + mp));
        o.a[k] = aw[item[k] * LP + mp];
    }
}

// backward: a per-point result of this channel half; the second half adds to what the first half stored (same lane,
// same address, program order).
template <bool ACC>
__device__ __forceinline__ void store_point_grads(float *__restrict__ grad_loc, float *__restrict__ grad_aw, unsigned pt,
                                                  float s_a, float gx, float gy)
{
    float2 *pl = reinterpret_cast<float2 *>(grad_loc + 2u * pt);
    if (ACC) {
        const float2 o = *pl;
        grad_aw[pt] += s_a;
        *pl = make_float2(o.x + gx, o.y + gy);
    } else {
        grad_aw[pt] = s_a;
        *pl = make_float2(gx, gy);
    }
}

// One sampled level for the kGatherQPG queries of a quad.  P4 = the level has exactly 4 points (RichSem): no
// point-count checks in the unrolled body.  Hot path = in-window points: one wave-divergent branch per point and
// straight-line LDS reads + packed FMAs.  Points with a corner outside the window are rare; they are handled
// afterwards in ONE run-time loop per query (not unrolled), with shuffles instead of DPP.
// MODE (backward): 0 = all 32 channels in one pass: store the per-point results; 1 = first channel half: keep them in
// `part` (lane i of the quad keeps point i); 2 = second channel half: add `part` and store.
template <bool BWD, bool P4, int MODE, int GC, int CPL, typename TV>
__device__ __forceinline__ void gather_level(
    const TV *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ aw, const float *win,
    const LevelCtx &lc, int row_elems, int P_, int j, int chan, const unsigned (&pt0)[GatherCfg<GC, CPL>::QPG],
    const bool (&live)[GatherCfg<GC, CPL>::QPG], const LevelOps<GatherCfg<GC, CPL>::QPG> &pre,
    v2f (&acc_lo)[GatherCfg<GC, CPL>::QPG], v2f (&acc_hi)[GatherCfg<GC, CPL>::QPG],
    const float4 (&gq)[GatherCfg<GC, CPL>::QPG][GatherCfg<GC, CPL>::NV], float (&part)[GatherCfg<GC, CPL>::QPG][3],
    float *__restrict__ grad_loc, float *__restrict__ grad_aw, unsigned &n_general, const int lvl,
    unsigned (&miss)[GatherCfg<GC, CPL>::QPG])
{
    constexpr bool defer = !BWD && P4 && GatherCfg<GC, CPL>::GL == 4;
    constexpr int kGatherQPG = GatherCfg<GC, CPL>::QPG, NV = GatherCfg<GC, CPL>::NV, GL = GatherCfg<GC, CPL>::GL;
    static_assert(BWD || NV == 1, "the forward keeps its accumulators for four channels per lane");
    const int P = P4 ? 4 : P_;
    for (int pc = 0; pc < P; pc += 4) {
        const int myp = pc + (j & 3);
        const bool pv = P4 || myp < P;
        const unsigned mp = pv ? myp : 0;
#pragma unroll
        for (int k = 0; k < kGatherQPG; ++k) {
            if (!live[k]) continue;   // uniform over the quad
            float2 xy = pre.xy[k];
            float a = pre.a[k];
            if (!P4 && pc > 0) {   // more than four points per level: the rest is loaded in place (uniform branch)
                xy = *reinterpret_cast<const float2 *>(loc + 2u * (pt0[k] + mp));
                a = aw[pt0[k] + mp];
            }
            // resolve this lane's point.  mode >= 0: LDS float index of corner (h_low, w_low), all corners in the window
            // (the window's apron beyond the map holds zeros, so border samples qualify too); -1: dropped; -2: general
            int mode = -1;
            float lh = 0.f, lw = 0.f;
            const float h_im = xy.y * (float)lc.H - 0.5f, w_im = xy.x * (float)lc.W - 0.5f;
            if (pv && h_im > -1.f && w_im > -1.f && h_im < (float)lc.H && w_im < (float)lc.W) {
                const float hf = floorf(h_im), wf = floorf(w_im);
                lh = h_im - hf;
                lw = w_im - wf;
                const int rr = (int)hf - lc.wr0, cc = (int)wf - lc.wc0;
                const bool inside = rr >= 0 && rr + 1 < lc.nwr && cc >= 0 && cc + 1 < lc.nwc;
                mode = inside ? lc.lds_base + (rr * lc.nwc + cc) * GC : -2;
            }
            n_general += (mode == -2 && j < 4) ? 1u : 0u;   // locality monitor (lanes 0..3 of a query hold its four points)
            // forward: a point that missed its window is only NOTED here (bit = level*4 + point, kept by the lane that resolved
            // it) and done after the item by whole lane groups (see tiled_gather_kernel) -- its cost is then per point, not
            // per wave iteration
            if (defer && mode == -2) miss[k] |= 1u << (lvl * 4 + (j & 3));
            const float hh = 1.f - lh, hw = 1.f - lw;
            // forward broadcasts finished weights; backward needs lh, lw and the attention weight separately
            const float w1 = hh * hw * a, w2 = hh * lw * a, w3 = lh * hw * a, w4 = lh * lw * a;
            bool any_slow = false;
            float ps_a = 0.f, ps_x = 0.f, ps_y = 0.f;   // backward: this lane's point (lane i <-> point pc + i)
#define MSDA_POINT(I)                                                                                                 \
    if (P4 || pc + I < P) {                                                                                            \
        const int m_ = quad_bcast_i<I>(mode);                                                                          \
        any_slow |= m_ == -2;                                                                                          \
        if (!BWD) {                                                                                                    \
            if (m_ >= 0) {                                                                                             \
                float4 v[4][NV];                                                                                       \
                lds_corners<GC, NV>(win, lc.nwc, j, m_, v);                                                            \
                fwd_accumulate(quad_bcast_f<I>(w1), quad_bcast_f<I>(w2), quad_bcast_f<I>(w3), quad_bcast_f<I>(w4),     \
                               v[0][0], v[1][0], v[2][0], v[3][0], acc_lo[k], acc_hi[k]);                              \
            }                                                                                                          \
        } else {                                                                                                       \
            float d1 = 0.f, d2 = 0.f, d3 = 0.f, d4 = 0.f, s_a, s_w, s_h;                                               \
            if (m_ >= 0) {                                                                                             \
                float4 v[4][NV];                                                                                       \
                lds_corners<GC, NV>(win, lc.nwc, j, m_, v);                                                            \
                corner_dots<NV>(gq[k], v, d1, d2, d3, d4);                                                             \
            }                                                                                                          \
            d1 = query_sum<GL>(d1);                                                                                    \
            d2 = query_sum<GL>(d2);                                                                                    \
            d3 = query_sum<GL>(d3);                                                                                    \
            d4 = query_sum<GL>(d4);                                                                                    \
            combine_dots(quad_bcast_f<I>(lh), quad_bcast_f<I>(lw), d1, d2, d3, d4, s_a, s_w, s_h);                     \
            /* lane I of the quad keeps point I (dropped / general points: zeros here, general ones redone below); */  \
            /* the four points are stored together after the loop: one 16-B and one 32-B segment per query */          \
            if (j == I) {                                                                                              \
                ps_a = s_a;                                                                                            \
                ps_x = (float)lc.W * s_w * a;                                                                          \
                ps_y = (float)lc.H * s_h * a;                                                                          \
            }                                                                                                          \
        }                                                                                                              \
    }
            MSDA_POINT(0)
            MSDA_POINT(1)
            MSDA_POINT(2)
            MSDA_POINT(3)
#undef MSDA_POINT
            if (BWD && j < 4 && (P4 || pc + j < P)) {
                if (MODE == 1) {
                    part[k][0] = ps_a;
                    part[k][1] = ps_x;
                    part[k][2] = ps_y;
                } else {
                    store_point_grads<false>(grad_loc, grad_aw, pt0[k] + pc + j, ps_a + (MODE == 2 ? part[k][0] : 0.f),
                                             ps_x + (MODE == 2 ? part[k][1] : 0.f), ps_y + (MODE == 2 ? part[k][2] : 0.f));
                }
            }
            if (!defer && any_slow) {   // uniform over the quad; rare
                // A wave steps through this body whenever ONE of its lanes holds a general point, so the body is kept short:
                // the owner lane resolves its point once, the quad receives corner offsets and coefficients by DPP
                // (compile-time slots, no cross-lane LDS traffic), and a slot that has no general point anywhere in the wave
                // is skipped by a wave-uniform test.
                const unsigned long long slow_lanes = __ballot(mode == -2);
                int o[4] = {-1, -1, -1, -1};
                if (mode == -2) {
                    float t0, t1;
                    resolve_point<float>(xy.x, xy.y, lc.H, lc.W, lc.base_row, row_elems, o, t0, t1);
                }
#define MSDA_SLOW(I)                                                                                                  \
    if ((slow_lanes & (0x1111111111111111ull << I)) && quad_bcast_i<I>(mode) == -2) {                                  \
        const int o0 = quad_bcast_i<I>(o[0]), o1 = quad_bcast_i<I>(o[1]), o2 = quad_bcast_i<I>(o[2]),                  \
                  o3 = quad_bcast_i<I>(o[3]);                                                                          \
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);                                                              \
        float4 v[4][NV];                                                                                               \
        _Pragma("unroll") for (int n = 0; n < NV; ++n)                                                                 \
        {                                                                                                              \
            v[0][n] = o0 >= 0 ? ld4(value + o0 + chan + 4 * n) : z;                                                    \
            v[1][n] = o1 >= 0 ? ld4(value + o1 + chan + 4 * n) : z;                                                    \
            v[2][n] = o2 >= 0 ? ld4(value + o2 + chan + 4 * n) : z;                                                    \
            v[3][n] = o3 >= 0 ? ld4(value + o3 + chan + 4 * n) : z;                                                    \
        }                                                                                                              \
        if (!BWD) {                                                                                                    \
            fwd_accumulate(quad_bcast_f<I>(w1), quad_bcast_f<I>(w2), quad_bcast_f<I>(w3), quad_bcast_f<I>(w4), v[0][0], \
                           v[1][0], v[2][0], v[3][0], acc_lo[k], acc_hi[k]);                                           \
        } else {                                                                                                       \
            float d1, d2, d3, d4, s_a, s_w, s_h;                                                                       \
            corner_dots<NV>(gq[k], v, d1, d2, d3, d4);                                                                 \
            d1 = query_sum<GL>(d1);                                                                                    \
            d2 = query_sum<GL>(d2);                                                                                    \
            d3 = query_sum<GL>(d3);                                                                                    \
            d4 = query_sum<GL>(d4);                                                                                    \
            combine_dots(quad_bcast_f<I>(lh), quad_bcast_f<I>(lw), d1, d2, d3, d4, s_a, s_w, s_h);                     \
            /* the fast pass gave this point zeros: add ours on top (first half: in `part`, else in memory) */         \
            if (j == I) {                                                                                              \
                const float gx_ = (float)lc.W * s_w * a, gy_ = (float)lc.H * s_h * a;                                  \
                if (MODE == 1) {                                                                                       \
                    part[k][0] += s_a;                                                                                 \
                    part[k][1] += gx_;                                                                                 \
                    part[k][2] += gy_;                                                                                 \
                } else {                                                                                               \
                    store_point_grads<true>(grad_loc, grad_aw, pt0[k] + pc + I, s_a, gx_, gy_);                        \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    }
                MSDA_SLOW(0)
                MSDA_SLOW(1)
                MSDA_SLOW(2)
                MSDA_SLOW(3)
#undef MSDA_SLOW
            }
        }
    }
}

// Forward (BWD = false): workgroup = (image, head, region, channel half); accumulates over the LDS phases in registers.
// Backward (BWD = true): workgroup = (image, head, region, LDS phase); runs the two channel halves one after the other
// (the per-point gradients are sums over all 32 channels).
// TV = storage type of value / grad_out / out (float, or bf16_t: converted at the loads / the store; the LDS windows and
// all arithmetic are fp32 either way)
template <bool BWD, bool P4, int GC, int CPL, typename TV = float>
__global__ __launch_bounds__(GC == 16 ? 512 : 1024, GC == 16 ? 4 : 1) void tiled_gather_kernel(
    const TV *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ aw,
    const TV *__restrict__ grad_out, TV *__restrict__ out, float *__restrict__ grad_loc,
    float *__restrict__ grad_aw, const TiledGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TileHeader *hdr = reinterpret_cast<TileHeader *>(smem);
    float *win = reinterpret_cast<float *>(smem + sizeof(TileHeader));

    // the sub-workgroups of a region are the fastest-varying part of the XCD-local index: they run back to back on one
    // XCD and share loc / attn / grad_out (and the value windows) in its L2
    using Cfg = GatherCfg<GC, CPL>;
    constexpr int GL = Cfg::GL;                              // lanes per query
    constexpr int NV = Cfg::NV;
    constexpr int kThreads = Cfg::kThreads;
    constexpr int kGroups = Cfg::kGroups;                    // queries in flight per pass over k
    constexpr int kGatherQPG = Cfg::QPG;
    constexpr int FL = GC / 4, kFillGroups = kThreads / FL;  // window fill: 16 B per lane, FL lanes per pixel
    // loads in flight per lane and batch.  A/B-timed on MI355X (tools/ab_probe.sh): 3..5 are equal, 8 costs ~4 us per
    // call and 10 (the whole finest-level window in one batch) ~8 us -- deeper batches only queue up behind the L2
    constexpr int kFillBatch = 4;
    constexpr int kHalves = kTD / GC;                        // channel passes per region
    const int nsub = BWD ? g.nphases : kHalves;
    // Persistent form: the grid may be smaller than the number of work items; a workgroup then walks the items
    // vb = blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x is a multiple of 8, so vb keeps the workgroup's XCD).
    const int n_items = kXcds * ((g.N * g.M + kXcds - 1) / kXcds) * g.GY * g.GX * nsub;
    unsigned n_general = 0;   // points of this lane that missed their window
    for (int vb = blockIdx.x; vb < n_items; vb += gridDim.x) {
    int pair, rs;
    if (!decode_block(vb, g.N * g.M, g.GY * g.GX * nsub, pair, rs)) continue;
    const int region = rs / nsub, sub = rs - region * nsub;
    const int b = pair / g.M, m = pair - b * g.M;
    const int gy = region / g.GX, gx = region - gy * g.GX;
    stamp<2>(g, 0);
    const int nq = build_header(hdr, g, gy, gx);
    stamp<2>(g, 1);

    const int tid = threadIdx.x;
    const int j = tid & (GL - 1), grp = tid / GL;
    const int fj = tid & (FL - 1), fgrp = tid / FL;
    const int row_elems = g.M * kTD;
    const int LP = g.L * g.P;

    bool live[kGatherQPG];
    unsigned item[kGatherQPG];   // (b*Lq + q)*M + m; every element offset derived from it fits 32 bits (host-checked)
    v2f acc_lo[kGatherQPG], acc_hi[kGatherQPG];   // forward: output accumulators of the quad's queries
#pragma unroll
    for (int k = 0; k < kGatherQPG; ++k) {
        const int i = grp + k * kGroups;
        live[k] = i < nq;
        item[k] = (unsigned)((b * g.Lq + hdr->qid[live[k] ? i : 0]) * g.M + m);
        acc_lo[k] = acc_hi[k] = (v2f){0.f, 0.f};
    }

    // backward on channel halves keeps the first half's per-point results here; it then needs one level per work item and
    // at most four points per level (host-enforced)
    float part[kGatherQPG][3];
    unsigned miss[kGatherQPG];   // forward: bit (level*4 + point) = that point of query k missed its window (this lane's point only)
#pragma unroll
    for (int k = 0; k < kGatherQPG; ++k) {
        part[k][0] = part[k][1] = part[k][2] = 0.f;
        miss[k] = 0u;
    }
    constexpr bool defer = !BWD && P4 && GL == 4;   // (compile-time: the in-loop general path is not even compiled into this kernel)
    const int ph_begin = BWD ? sub : 0, ph_end = BWD ? ph_begin + 1 : g.nphases;
    const int half_begin = BWD ? 0 : sub, half_end = BWD ? kHalves : sub + 1;
    int st = 2;
    for (int half = half_begin; half < half_end; ++half) {
        const int chan = half * GC + CPL * j;   // this lane's first channel inside the head
        float4 gq[kGatherQPG][NV];               // backward: grad_out of the group's queries, this lane's channels
#pragma unroll
        for (int k = 0; k < kGatherQPG; ++k)
#pragma unroll
            for (int n = 0; n < NV; ++n)
                gq[k][n] = BWD ? ld4(grad_out + item[k] * (unsigned)kTD + chan + 4 * n) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ph = ph_begin; ph < ph_end; ++ph) {
            // levels of this phase are consecutive: [lb, le)
            int lb = g.L, le = 0;
            for (int l = 0; l < g.L; ++l)
                if (uni(hdr->phase[l]) == ph) { lb = l < lb ? l : lb; le = l + 1; }
            LevelOps<kGatherQPG> nxt;
            load_level_ops(loc, aw, item, (unsigned)LP, (unsigned)(lb * g.P), g.P, j, nxt);   // in flight during the fill
            // ---- stage this phase's windows: 64-B pixel half-rows, 16 B per lane -------------------------------------
            for (int l = lb; l < le; ++l) {
                const int wr0 = uni(hdr->r[l].wr0), wc0 = uni(hdr->r[l].wc0), nwc = uni(hdr->r[l].nwc);
                const int npx = uni(hdr->r[l].nwr) * nwc, Wl = uni(hdr->W[l]), Hl = uni(hdr->H[l]);
                const TV *src = value + ((int64_t)(b * g.S + uni(hdr->start[l])) * g.M + m) * kTD + half * GC + 4 * fj;
                float *dst = win + (int64_t)uni(hdr->lds_px[l]) * GC + 4 * fj;
                // kFillBatch independent loads in flight per lane before the first LDS store
                for (int px0 = fgrp; px0 < npx; px0 += kFillBatch * kFillGroups) {
                    float4 v[kFillBatch];
#pragma unroll
                    for (int u = 0; u < kFillBatch; ++u) {
                        const int px = min(px0 + u * kFillGroups, npx - 1);   // clamped; stored only if in range
                        const int rr = px / nwc, cc = px - rr * nwc;
                        const int row = wr0 + rr, col = wc0 + cc;
                        const bool in_map = row >= 0 && row < Hl && col >= 0 && col < Wl;   // else: the zero apron
                        const int rowc = min(max(row, 0), Hl - 1), colc = min(max(col, 0), Wl - 1);
                        const float4 t = ld4(src + (int64_t)(rowc * Wl + colc) * row_elems);
                        v[u] = in_map ? t : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < kFillBatch; ++u) {
                        const int px = px0 + u * kFillGroups;
                        if (px < npx) *reinterpret_cast<float4 *>(dst + px * GC) = v[u];
                    }
                }
            }
            __syncthreads();
            stamp<2>(g, st++);

            // ---- gather -------------------------------------------------------------------------------------
            for (int l = lb; l < le; ++l) {
                const LevelOps<kGatherQPG> cur = nxt;
                if (l + 1 < le) load_level_ops(loc, aw, item, (unsigned)LP, (unsigned)((l + 1) * g.P), g.P, j, nxt);
                LevelCtx lc;
                lc.H = uni(hdr->H[l]);
                lc.W = uni(hdr->W[l]);
                lc.wr0 = uni(hdr->r[l].wr0);
                lc.wc0 = uni(hdr->r[l].wc0);
                lc.nwr = uni(hdr->r[l].nwr);
                lc.nwc = uni(hdr->r[l].nwc);
                lc.lds_base = uni(hdr->lds_px[l]) * GC;
                lc.base_row = (b * g.S + uni(hdr->start[l])) * row_elems + m * kTD;
                unsigned pt0[kGatherQPG];
#pragma unroll
                for (int k = 0; k < kGatherQPG; ++k) pt0[k] = item[k] * (unsigned)LP + (unsigned)(l * g.P);
                if (!BWD || kHalves == 1)
                    gather_level<BWD, P4, 0, GC, CPL, TV>(value, loc, aw, win, lc, row_elems, g.P, j, chan, pt0, live, cur, acc_lo,
                                                 acc_hi, gq, part, grad_loc, grad_aw, n_general, l, miss);
                else if (half == 0)
                    gather_level<BWD, P4, 1, GC, CPL, TV>(value, loc, aw, win, lc, row_elems, g.P, j, chan, pt0, live, cur, acc_lo,
                                                 acc_hi, gq, part, grad_loc, grad_aw, n_general, l, miss);
                else
                    gather_level<BWD, P4, 2, GC, CPL, TV>(value, loc, aw, win, lc, row_elems, g.P, j, chan, pt0, live, cur, acc_lo,
                                                 acc_hi, gq, part, grad_loc, grad_aw, n_general, l, miss);
            }
            // the next fill overwrites the windows (forward, last phase: the barrier below, which also tells whether any point
            // of the workgroup missed its window, takes this one's place)
            if (BWD || !defer || ph + 1 < ph_end || half + 1 < half_end) __syncthreads();
            stamp<2>(g, st++);
        }
    }

    if (!BWD && defer) {
        // ---- points that missed their window: collected per query, then done by 4-lane groups (4 channels per lane of this
        //      channel half) straight from global memory -- every lane works on a missed point, nobody idles beside one.  The
        //      partial rows come back through LDS (the windows are free now) and the owners add them before they store. -------
        int *fx_list = reinterpret_cast<int *>(win);          // [kMaxRegionQueries][2]: (query slot, mask of missed points)
        float *fx_patch = win + 2 * kMaxRegionQueries;        // [kMaxRegionQueries][GC]
        unsigned any_miss = 0u;
#pragma unroll
        for (int k = 0; k < kGatherQPG; ++k) any_miss |= live[k] ? miss[k] : 0u;
        if (__syncthreads_or((int)any_miss)) {   // (uniform over the workgroup; the windows are free behind this barrier)
            int my_row[kGatherQPG];
#pragma unroll
            for (int k = 0; k < kGatherQPG; ++k) {
                unsigned mk = miss[k];
                mk |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mk, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                mk |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mk, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
                int r = -1;
                if (live[k] && mk && j == 0) {
                    r = atomicAdd(&hdr->pad[0], 1);
                    fx_list[2 * r] = grp + k * kGroups;
                    fx_list[2 * r + 1] = (int)mk;
                }
                my_row[k] = quad_bcast_i<0>(r);
            }
            __syncthreads();
            const int n_fix = uni(hdr->pad[0]);
            const int chan = sub * GC + 4 * j;
            for (int e = grp; e < n_fix; e += kGroups) {
                const int slot = fx_list[2 * e];
                unsigned mk = (unsigned)fx_list[2 * e + 1];
                const unsigned itm = (unsigned)((b * g.Lq + hdr->qid[slot]) * g.M + m);
                v2f lo = (v2f){0.f, 0.f}, hi = (v2f){0.f, 0.f};
                while (mk) {
                    const int bit = __ffs((int)mk) - 1;
                    mk &= mk - 1u;
                    const int lv = bit >> 2, pp = bit & 3;
                    const unsigned pt = itm * (unsigned)LP + (unsigned)(lv * g.P + pp);
                    const float2 xy = *reinterpret_cast<const float2 *>(loc + 2u * pt);
                    const float a = aw[pt];
                    int o[4];
                    float lh, lw;
                    resolve_point<float>(xy.x, xy.y, hdr->H[lv], hdr->W[lv], (b * g.S + hdr->start[lv]) * row_elems + m * kTD, row_elems, o, lh, lw);
                    const float hh = 1.f - lh, hw = 1.f - lw;
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 v1 = o[0] >= 0 ? ld4(value + o[0] + chan) : z, v2 = o[1] >= 0 ? ld4(value + o[1] + chan) : z;
                    const float4 v3 = o[2] >= 0 ? ld4(value + o[2] + chan) : z, v4 = o[3] >= 0 ? ld4(value + o[3] + chan) : z;
                    fwd_accumulate(hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a, v1, v2, v3, v4, lo, hi);
                }
                *reinterpret_cast<float4 *>(fx_patch + e * GC + 4 * j) = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kGatherQPG; ++k)
                if (my_row[k] >= 0) {
                    const float4 t = *reinterpret_cast<const float4 *>(fx_patch + my_row[k] * GC + 4 * j);
                    acc_lo[k] += (v2f){t.x, t.y};
                    acc_hi[k] += (v2f){t.z, t.w};
                }
        }
    }
    stamp<2>(g, st++);
    if (!BWD) {
        const int chan = sub * GC + 4 * j;
#pragma unroll
        for (int k = 0; k < kGatherQPG; ++k)
            if (live[k])
                st4(out + item[k] * (unsigned)kTD + chan, make_float4(acc_lo[k].x, acc_lo[k].y, acc_hi[k].x, acc_hi[k].y));
    }
    __syncthreads();   // the next item rebuilds the header
    stamp<2>(g, st++);
    }
    if (g.stats) {   // locality monitor: one atomic per wave (only when the host asked for the count)
        for (int o = kWave / 2; o > 0; o >>= 1) n_general += __shfl_xor(n_general, o, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0 && n_general) atomicAdd(g.stats, n_general);
    }
}

// ---- backward: grad_value ---------------------------------------------------------------------------------
// gfx950 facts this kernel is built on (measured, tools/lds_atomic_bench.hip): the LDS float atomic ds_add_f32 is
// serialised (120-190 CU cycles per wave-instruction) while ds_add_f64 is native (about 5), and global float atomics
// run at ~1.3 TB/s only as whole row segments.  So the window accumulates in f64 (which also makes the in-window sum
// exact to f32 precision whatever the order), a workgroup takes one CHANNEL HALF of a region (16 channels x 8 B =
// 128 B per pixel, the same LDS geometry as the gather kernels) and every touched pixel is flushed once.
// 16 lanes per query (4 quads): lane i of every quad resolves sampling point i of the current level and the quad shares
// it by DPP broadcast; each lane owns one channel of the half.  Persistent workgroups walk (region, channel half, phase).
__global__ __launch_bounds__(kTiledThreads) void tiled_scatter_kernel(
    const float *__restrict__ loc, const float *__restrict__ aw, const float *__restrict__ grad_out,
    float *__restrict__ grad_value, const TiledGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TileHeader *hdr = reinterpret_cast<TileHeader *>(smem);
    double *win = reinterpret_cast<double *>(smem + sizeof(TileHeader));

    // (channel half, phase) of a region = the fastest-varying part of the XCD-local index: the workgroups of one
    // region run back to back on one XCD and share loc / attn / grad_out in its L2
    const int nsub = (kTD / kSD) * g.nphases;
    const int n_items = kXcds * ((g.N * g.M + kXcds - 1) / kXcds) * g.GY * g.GX * nsub;
    for (int vb = blockIdx.x; vb < n_items; vb += gridDim.x) {   // persistent form, see tiled_gather_kernel
    int pair, rs;
    if (!decode_block(vb, g.N * g.M, g.GY * g.GX * nsub, pair, rs)) continue;
    const int region = rs / nsub, sub = rs - region * nsub;
    const int half = sub % (kTD / kSD);
    const int b = pair / g.M, m = pair - b * g.M;
    const int gy = region / g.GX, gx = region - gy * g.GX;
    stamp<1>(g, 0);
    const int nq = build_header(hdr, g, gy, gx);
    stamp<1>(g, 1);

    const int tid = threadIdx.x;
    const int j = tid & (kSD - 1), grp = tid / kSD;
    const int row_elems = g.M * kTD;
    const int LP = g.L * g.P;
    const int ch0 = m * kTD + half * kSD;   // first channel of this workgroup inside a pixel row
    {
        const int ph = sub / (kTD / kSD);   // one LDS phase per workgroup: the phases of a region are independent here
        // ---- clear this phase's accumulation windows ---------------------------------------------------------
        int phase_px = 0;
        for (int l = 0; l < g.L; ++l)
            if (uni(hdr->phase[l]) == ph) phase_px = uni(hdr->lds_px[l]) + uni(hdr->r[l].nwr) * uni(hdr->r[l].nwc);
        for (int i = tid; i < phase_px * (kSD / 2); i += kTiledThreads)
            reinterpret_cast<double2 *>(win)[i] = make_double2(0.0, 0.0);
        __syncthreads();
        stamp<1>(g, 2);

        // ---- accumulate: a query = 16 lanes (4 quads); lane i of every quad resolves point i of the current level and the
        //      quad shares it by DPP broadcast (no LDS records); kScatterBatch queries per group are fetched together ---------
        for (int i0 = grp; i0 < nq; i0 += kScatterBatch * kScatterGroups) {
            unsigned items[kScatterBatch];
            float gks[kScatterBatch];
#pragma unroll
            for (int u = 0; u < kScatterBatch; ++u) {
                const int i = i0 + u * kScatterGroups;
                items[u] = (unsigned)((b * g.Lq + hdr->qid[i < nq ? i : i0]) * g.M + m);   // clamped; masked below
                gks[u] = grad_out[items[u] * (unsigned)kTD + half * kSD + j];
            }
            for (int l = 0; l < g.L; ++l) {
                if (uni(hdr->phase[l]) != ph) continue;
                const int H = uni(hdr->H[l]), W = uni(hdr->W[l]);
                const int wr0 = uni(hdr->r[l].wr0), wc0 = uni(hdr->r[l].wc0), nwr = uni(hdr->r[l].nwr), nwc = uni(hdr->r[l].nwc);
                const int ldsl = uni(hdr->lds_px[l]);
                const int base_l = (b * g.S + uni(hdr->start[l])) * row_elems + ch0;
                const int row2 = nwc * kSD;   // f64 elements between vertically adjacent window pixels
                for (int pc = 0; pc < g.P; pc += 4) {
                    const int myp = pc + (j & 3);
                    const bool pv = myp < g.P;
                    float2 xys[kScatterBatch];
                    float as[kScatterBatch];
#pragma unroll
                    for (int u = 0; u < kScatterBatch; ++u) {
                        const unsigned pt = items[u] * (unsigned)LP + (unsigned)(l * g.P + (pv ? myp : 0));
                        xys[u] = *reinterpret_cast<const float2 *>(loc + 2u * pt);
                        as[u] = aw[pt];
                    }
#pragma unroll
                    for (int u = 0; u < kScatterBatch; ++u) {
                        if (i0 + u * kScatterGroups >= nq) break;   // uniform over the 16-lane group
                        const float gk = gks[u];
                        // resolve this lane's point: base >= 0: LDS f64 index of corner (h_low, w_low), all four corners in the
                        // window (apron included); -1: nothing; -2: general point with per-corner targets t[]
                        int base = -1, t0 = -1, t1 = -1, t2 = -1, t3 = -1;
                        float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
                        const float h_im = xys[u].y * (float)H - 0.5f, w_im = xys[u].x * (float)W - 0.5f;
                        if (pv && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                            const float hf = floorf(h_im), wf = floorf(w_im);
                            const int h_low = (int)hf, w_low = (int)wf;
                            const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                            w0 = hh * hw * as[u];
                            w1 = hh * lw * as[u];
                            w2 = lh * hw * as[u];
                            w3 = lh * lw * as[u];
                            const int rr = h_low - wr0, cc = w_low - wc0;
                            const bool r0 = rr >= 0 && rr < nwr, r1 = rr + 1 >= 0 && rr + 1 < nwr;
                            const bool c0 = cc >= 0 && cc < nwc, c1 = cc + 1 >= 0 && cc + 1 < nwc;
                            const int lbase = (ldsl + rr * nwc + cc) * kSD;
                            if (r0 && r1 && c0 && c1) {
                                base = lbase;
                            } else {
                                base = -2;
                                const bool top = h_low >= 0, bot = h_low + 1 <= H - 1, lef = w_low >= 0, rig = w_low + 1 <= W - 1;
                                const int gbase = base_l + (h_low * W + w_low) * row_elems;
                                if (top && lef) t0 = (r0 && c0) ? lbase : -gbase - 2;
                                if (top && rig) t1 = (r0 && c1) ? lbase + kSD : -(gbase + row_elems) - 2;
                                if (bot && lef) t2 = (r1 && c0) ? lbase + row2 : -(gbase + W * row_elems) - 2;
                                if (bot && rig) t3 = (r1 && c1) ? lbase + row2 + kSD : -(gbase + W * row_elems + row_elems) - 2;
                            }
                        }
#define MSDA_SC_ONE(I)                                                                                                 \
    if (pc + I < g.P) {                                                                                                 \
        const int base_ = quad_bcast_i<I>(base);                                                                        \
        const float v0 = quad_bcast_f<I>(w0) * gk, v1 = quad_bcast_f<I>(w1) * gk;                                       \
        const float v2 = quad_bcast_f<I>(w2) * gk, v3 = quad_bcast_f<I>(w3) * gk;                                       \
        if (base_ >= 0) {                                                                                               \
            double *p0 = win + base_ + j, *p1 = p0 + row2;                                                              \
            atomicAdd(p0, (double)v0);                                                                                  \
            atomicAdd(p0 + kSD, (double)v1);                                                                            \
            atomicAdd(p1, (double)v2);                                                                                  \
            atomicAdd(p1 + kSD, (double)v3);                                                                            \
        } else if (base_ == -2) {                                                                                       \
            const int tt[4] = {quad_bcast_i<I>(t0), quad_bcast_i<I>(t1), quad_bcast_i<I>(t2), quad_bcast_i<I>(t3)};     \
            const float vv[4] = {v0, v1, v2, v3};                                                                       \
            _Pragma("unroll") for (int cn = 0; cn < 4; ++cn)                                                            \
            {                                                                                                           \
                if (tt[cn] >= 0)                                                                                        \
                    atomicAdd(win + tt[cn] + j, (double)vv[cn]);                                                        \
                else if (tt[cn] < -1)                                                                                   \
                    atomicAdd(grad_value + (-(tt[cn] + 2)) + j, vv[cn]);                                                \
            }                                                                                                           \
        }                                                                                                               \
    }
                        MSDA_SC_ONE(0)
                        MSDA_SC_ONE(1)
                        MSDA_SC_ONE(2)
                        MSDA_SC_ONE(3)
#undef MSDA_SC_ONE
                    }
                }
            }
        }
        __syncthreads();
        stamp<1>(g, 3);

        // ---- flush: every touched pixel once, 64-B row segments of global float atomics --------------------------------
        for (int l = 0; l < g.L; ++l) {
            if (uni(hdr->phase[l]) != ph) continue;
            const int fr0 = uni(hdr->r[l].wr0), fc0 = uni(hdr->r[l].wc0), fnc = uni(hdr->r[l].nwc);
            const int npx = uni(hdr->r[l].nwr) * fnc, Wl = uni(hdr->W[l]);
            float *dst = grad_value + (int64_t)(b * g.S + uni(hdr->start[l])) * row_elems + ch0 + j;
            const double *src = win + (int64_t)uni(hdr->lds_px[l]) * kSD + j;
            const int Hl = uni(hdr->H[l]);
            for (int px = grp; px < npx; px += kScatterGroups) {
                const float v = (float)src[px * kSD];
                const int rr = px / fnc, cc = px - rr * fnc;
                const int row = fr0 + rr, col = fc0 + cc;
                if (v != 0.f && row >= 0 && row < Hl && col >= 0 && col < Wl)
                    atomicAdd(dst + (int64_t)(row * Wl + col) * row_elems, v);
            }
        }
        __syncthreads();
        stamp<1>(g, 4);
    }
    }
}

// ---- backward: grad_value, integer accumulation ("block floating point per pixel") --------------------------------------
// ds_add_u32 costs ~2.6 CU cycles per wave instruction against ~13 for ds_add_f64 under this kernel's bank conflicts, and a
// 32-bit accumulator holds all 32 channels of a pixel in 128 B, so one workgroup serves a whole (region, level).  To make
// integer accumulation safe for any input, every window pixel gets ITS OWN scale:
//   pass 0  gmax[q] = max_c |grad_out[q, c]| for the region's queries
//   pass 1  one thread per (query, point): resolve the point once, park it as a record in LDS, and per in-window corner
//           update cnt[pixel] += 1 and maxc[pixel] = max(maxc, |bilinear * attn| * gmax[q]) with integer LDS atomics
//           (non-negative floats order like their bit patterns)
//   scale   2^30 / (cnt * maxc): the quantised contributions of a pixel can never overflow 32 bits; the records'
//           weights are multiplied by their destination pixel's scale
//   pass 2  replay: acc[pixel][channel] += round(weight * grad_out[q, channel])   (ds_add_u32, one channel per lane)
//   flush   acc / scale, one 128-B row global float atomic per touched in-map pixel
// Error per pixel-channel <= cnt^2 * maxc / 2^31 in the worst case, ~sqrt(cnt) * cnt * maxc * 2^-32 typically: for 64
// contributions 2e-6 / 1e-7 of the pixel's largest contribution -- float32-class -- and the in-window sum is bitwise
// reproducible (integer sums do not depend on order).  Points with a corner outside the window use global float atomics.
constexpr int kBfpGroups = kTiledThreads / kTD;          // 32-lane groups, one query each
constexpr int kBfpPxBytes = kTD * 4 + 8;                 // int32 x 32 channels + {maxc | scale, cnt | 1/scale}
constexpr int kBfpBatch = 4;                             // queries per group whose grad_out rows are fetched together

struct alignas(4) BfpRec {   // one sampling point of one of the region's queries, resolved in pass 1
    int base;     // >= 0: window pixel of corner (h_low, w_low), all four corners inside the window (incl. apron);
                  //   -1: nothing to do;  -2: general point, resolved again (per corner) when replayed
    float w[4];   // bilinear weight x attention weight (x destination pixel scale after the scale step)
};

// Per-corner targets of a general point: window pixel (>= 0), -(global element offset) - 2, or -1.
__device__ __forceinline__ void bfp_general_targets(float x, float y, int H, int W, int wr0, int wc0, int nwr, int nwc,
                                                    int lds_px, int base_row, int row_elems, int t[4])
{
    t[0] = t[1] = t[2] = t[3] = -1;
    const float h_im = y * (float)H - 0.5f, w_im = x * (float)W - 0.5f;
    if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) return;
    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
    const int rr = h_low - wr0, cc = w_low - wc0;
    const bool r0 = rr >= 0 && rr < nwr, r1 = rr + 1 >= 0 && rr + 1 < nwr;
    const bool c0 = cc >= 0 && cc < nwc, c1 = cc + 1 >= 0 && cc + 1 < nwc;
    const bool top = h_low >= 0, bot = h_low + 1 <= H - 1, lef = w_low >= 0, rig = w_low + 1 <= W - 1;
    const int lbase = lds_px + rr * nwc + cc;
    const int gbase = base_row + (h_low * W + w_low) * row_elems;
    if (top && lef) t[0] = (r0 && c0) ? lbase : -gbase - 2;
    if (top && rig) t[1] = (r0 && c1) ? lbase + 1 : -(gbase + row_elems) - 2;
    if (bot && lef) t[2] = (r1 && c0) ? lbase + nwc : -(gbase + W * row_elems) - 2;
    if (bot && rig) t[3] = (r1 && c1) ? lbase + nwc + 1 : -(gbase + W * row_elems + row_elems) - 2;
}

template <bool P4>
__global__ __launch_bounds__(kTiledThreads) void tiled_scatter_bfp_kernel(
    const float *__restrict__ loc, const float *__restrict__ aw, const float *__restrict__ grad_out,
    float *__restrict__ grad_value, const TiledGeom g, const int max_phase_px)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TileHeader *hdr = reinterpret_cast<TileHeader *>(smem);
    int *acc = reinterpret_cast<int *>(smem + sizeof(TileHeader));                       // [px][32]
    unsigned *maxc = reinterpret_cast<unsigned *>(acc + (size_t)max_phase_px * kTD);     // pass 1: max bits; then scale
    unsigned *cnt = maxc + max_phase_px;                                                  // pass 1: count; then 1/scale
    BfpRec *recs = reinterpret_cast<BfpRec *>(cnt + max_phase_px);                        // [query][point]

    const int tid = threadIdx.x;
    const int j = tid & (kTD - 1), grp = tid / kTD;
    const int row_elems = g.M * kTD;
    const int LP = g.L * g.P;
    const int nsub = g.nphases;   // one level per phase
    const int n_items = kXcds * ((g.N * g.M + kXcds - 1) / kXcds) * g.GY * g.GX * nsub;
    for (int vb = blockIdx.x; vb < n_items; vb += gridDim.x) {   // persistent form, see tiled_gather_kernel
        int pair, rs;
        if (!decode_block(vb, g.N * g.M, g.GY * g.GX * nsub, pair, rs)) continue;
        const int region = rs / nsub, lv = rs - region * nsub;   // lv: the level this item scatters into
        const int b = pair / g.M, m = pair - b * g.M;
        const int gy = region / g.GX, gx = region - gy * g.GX;
        stamp<1>(g, 0);
        const int nq = build_header(hdr, g, gy, gx);
        stamp<1>(g, 1);

        const int H = uni(hdr->H[lv]), W = uni(hdr->W[lv]), nwc = uni(hdr->r[lv].nwc), nwr = uni(hdr->r[lv].nwr);
        const int wr0 = uni(hdr->r[lv].wr0), wc0 = uni(hdr->r[lv].wc0);
        const int npx = nwr * nwc;   // the window starts at LDS pixel 0 (one level per phase)
        const int base_row = (b * g.S + uni(hdr->start[lv])) * row_elems + m * kTD;

        for (int i = tid; i < npx * (kTD / 4); i += kTiledThreads) reinterpret_cast<int4 *>(acc)[i] = make_int4(0, 0, 0, 0);
        for (int i = tid; i < npx; i += kTiledThreads) { maxc[i] = 0u; cnt[i] = 0u; }

        // ---- pass 0: gmax[q]; 8 lanes x 4 channels per query, 128 queries per sweep -----------------------------------
        for (int i = tid >> 3; i < nq; i += kTiledThreads / 8) {
            const unsigned item = (unsigned)((b * g.Lq + hdr->qid[i]) * g.M + m);
            const float4 v = *reinterpret_cast<const float4 *>(grad_out + item * (unsigned)kTD + 4u * (tid & 7));
            float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
            mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0xB1, 0xF, 0xF, true)));
            mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x4E, 0xF, 0xF, true)));
            mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x141, 0xF, 0xF, true)));
            if ((tid & 7) == 0) hdr->gmax[i] = mx;
        }
        __syncthreads();
        stamp<1>(g, 2);

        // ---- pass 1: one thread per (query, point): record + per-pixel count / largest possible contribution ---------------------
        for (int idx = tid; idx < nq * g.P; idx += kTiledThreads) {
            const int qi = idx / g.P, pp = idx - qi * g.P;
            const unsigned item = (unsigned)((b * g.Lq + hdr->qid[qi]) * g.M + m);
            const unsigned pt = item * (unsigned)LP + (unsigned)(lv * g.P + pp);
            const float2 xy = *reinterpret_cast<const float2 *>(loc + 2u * pt);
            const float a = aw[pt];
            BfpRec r;
            r.base = -1;
            r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0.f;
            const float h_im = xy.y * (float)H - 0.5f, w_im = xy.x * (float)W - 0.5f;
            if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                const float hf = floorf(h_im), wf = floorf(w_im);
                const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                r.w[0] = hh * hw * a;
                r.w[1] = hh * lw * a;
                r.w[2] = lh * hw * a;
                r.w[3] = lh * lw * a;
                const int rr = (int)hf - wr0, cc = (int)wf - wc0;
                const bool inside = rr >= 0 && rr + 1 < nwr && cc >= 0 && cc + 1 < nwc;
                r.base = inside ? rr * nwc + cc : -2;
                const float gm = hdr->gmax[qi];
                int t[4];
                if (inside) {
                    t[0] = r.base; t[1] = r.base + 1; t[2] = r.base + nwc; t[3] = r.base + nwc + 1;
                } else {
                    bfp_general_targets(xy.x, xy.y, H, W, wr0, wc0, nwr, nwc, 0, 0, row_elems, t);
                }
#pragma unroll
                for (int cn = 0; cn < 4; ++cn)
                    if (t[cn] >= 0) {
                        atomicMax(maxc + t[cn], __float_as_uint(fabsf(r.w[cn]) * gm));
                        atomicAdd(cnt + t[cn], 1u);
                    }
            }
            recs[idx] = r;
        }
        __syncthreads();
        // ---- per-pixel scale; a non-finite bound (inf / nan in grad_out or attn) poisons the pixel instead of hiding it ------
        for (int i = tid; i < npx; i += kTiledThreads) {
            const float bound = (float)cnt[i] * __uint_as_float(maxc[i]);
            const bool ok = bound > 0.f && bound < 3.0e38f;
            maxc[i] = __float_as_uint(ok ? 1073741824.f / bound : 0.f);
            cnt[i] = __float_as_uint(bound > 0.f || bound != bound ? bound * (1.f / 1073741824.f) : 0.f);
        }
        __syncthreads();
        const float *scale = reinterpret_cast<const float *>(maxc);
        // fold the destination pixels' scales into the weights of the in-window records
        for (int idx = tid; idx < nq * g.P; idx += kTiledThreads) {
            const int base = recs[idx].base;
            if (base >= 0) {
                recs[idx].w[0] *= scale[base];
                recs[idx].w[1] *= scale[base + 1];
                recs[idx].w[2] *= scale[base + nwc];
                recs[idx].w[3] *= scale[base + nwc + 1];
            }
        }
        __syncthreads();
        stamp<1>(g, 3);

        // ---- pass 2: replay; 32 lanes = 32 channels of one query, kBfpBatch queries' grad_out rows fetched together ------------
        for (int i0 = grp; i0 < nq; i0 += kBfpBatch * kBfpGroups) {
            float gs[kBfpBatch];
#pragma unroll
            for (int u = 0; u < kBfpBatch; ++u) {
                const int i = i0 + u * kBfpGroups;
                const unsigned item = (unsigned)((b * g.Lq + hdr->qid[i < nq ? i : i0]) * g.M + m);
                gs[u] = grad_out[item * (unsigned)kTD + j];
            }
#pragma unroll
            for (int u = 0; u < kBfpBatch; ++u) {
                const int i = i0 + u * kBfpGroups;
                if (i >= nq) break;   // uniform over the 32-lane group
                const float gk = gs[u];
                const int np = P4 ? 4 : g.P;
                BfpRec rq[4];
                if (P4) {   // the four records of the query are read together: one LDS latency per query, not per point
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) rq[pp] = recs[i * 4 + pp];
                }
#pragma unroll
                for (int pp = 0; pp < (P4 ? 4 : 16); ++pp) {
                    if (!P4 && pp >= np) break;
                    const BfpRec r = P4 ? rq[pp & 3] : recs[i * g.P + pp];   // same address in all 32 lanes: LDS broadcast
                    if (r.base >= 0) {
                        int *a0 = acc + r.base * kTD + j, *a1 = a0 + nwc * kTD;
                        atomicAdd(a0, __float2int_rn(r.w[0] * gk));
                        atomicAdd(a0 + kTD, __float2int_rn(r.w[1] * gk));
                        atomicAdd(a1, __float2int_rn(r.w[2] * gk));
                        atomicAdd(a1 + kTD, __float2int_rn(r.w[3] * gk));
                    } else if (r.base == -2) {   // rare: a corner outside the window
                        const unsigned item = (unsigned)((b * g.Lq + hdr->qid[i]) * g.M + m);
                        const float2 xy = *reinterpret_cast<const float2 *>(loc + 2u * (item * (unsigned)LP + (unsigned)(lv * g.P + pp)));
                        int t[4];
                        bfp_general_targets(xy.x, xy.y, H, W, wr0, wc0, nwr, nwc, 0, base_row, row_elems, t);
#pragma unroll
                        for (int cn = 0; cn < 4; ++cn) {
                            if (t[cn] >= 0)
                                atomicAdd(acc + t[cn] * kTD + j, __float2int_rn(r.w[cn] * scale[t[cn]] * gk));
                            else if (t[cn] < -1)
                                atomicAdd(grad_value + (-(t[cn] + 2)) + j, r.w[cn] * gk);
                        }
                    }
                }
            }
        }
        __syncthreads();
        stamp<1>(g, 4);

        // ---- flush: de-quantise, one 128-B row of global float atomics per touched in-map pixel ---------------------------------
        {
            const float *inv = reinterpret_cast<const float *>(cnt);
            float *dst = grad_value + (int64_t)(b * g.S + uni(hdr->start[lv])) * row_elems + m * kTD + j;
            for (int px = grp; px < npx; px += kBfpGroups) {
                const int q = acc[px * kTD + j];
                const float iv = inv[px];
                const int rr = px / nwc, cc = px - rr * nwc;
                const int row = wr0 + rr, col = wc0 + cc;
                const bool poisoned = !(iv == iv) || fabsf(iv) > 3.0e38f;   // non-finite bound: keep the result non-finite
                if ((q != 0 || poisoned) && row >= 0 && row < H && col >= 0 && col < W)
                    atomicAdd(dst + (int64_t)(row * W + col) * row_elems, poisoned ? iv : (float)q * iv);
            }
        }
        stamp<1>(g, 5);
        __syncthreads();   // the next item rebuilds the header and clears the window
    }
}

// ---- backward: grad_value by sorted (segmented) reduction -- no floating-point LDS atomics ---------------------------------------
// Work item = (image, head, region), all 32 channels, its levels one after the other.  Instead of adding every corner of every
// sampling point into an LDS window with a float atomic (the f64 kernel above is bound by the ds_add_f64 issue rate), every
// (point, corner) pair becomes an 8-byte ENTRY {query, weight} that is sorted -- with integer LDS atomics, one lane per point,
// not per channel -- into the list of its destination pixel, and every output pixel then sums its list in registers:
//   header + A  once per region: geometry, query list, grad_out rows of the region's queries -> LDS (gcache)
//   per level (the next level's sampling locations / attention weights are fetched during the current level's E):
//   B  one thread per (query, point): resolve; each in-map corner of an in-window point takes a rank in its pixel's list
//      (ds_add_rtn_u32); points with a corner outside the window go to a side list
//   C  exclusive scan of the per-pixel counts -> list offsets;  D  entries written to their sorted slots
//   E  8 lanes x 4 channels per output pixel: acc += w * gcache[q] over the pixel's list, four entries in flight; the wave
//      then hands its 8 pixel rows over through LDS, re-reads them one channel per lane and adds whole 128-B rows to
//      grad_value (two rows per global atomic instruction)
//   F  side list: 32 lanes per point, row atomics straight to global memory (as the direct kernel)
// Sums are fp32 like the reference's; their order inside a list follows the atomic ranks (run-to-run variation at the
// rounding level, as with the reference's float atomics).
// What the step costs was measured per stage and per level (tools/stage_stamps.py): E is bound by instruction issue and LDS
// traffic per list entry, which is why the entries are per pixel (one 8-B read, no corner / bucket arithmetic in the loop).
constexpr int kSortMaxPx = 1280;            // largest single-level window (pixels) the kernel takes
constexpr int kSortMaxPts = kMaxRegionQueries * 4;

struct alignas(8) SortRec {
    int q;     // index of the query inside the region (row of gcache)
    float w;   // bilinear weight x attention weight of the corner that is this pixel
};

struct SortLds {   // after the TileHeader
    float gcache[kMaxRegionQueries * kTD];
    int offs[kSortMaxPx + 4];               // histogram, then exclusive prefix (offs[npx] = total)
    SortRec sorted[4 * kSortMaxPts];        // one entry per (point, corner), grouped by destination pixel
    unsigned short genlist[kSortMaxPts];    // points with a corner outside the window (index of the point in the region)
    float stage[16][8 * kTD];               // step E: per wave, the 8 pixel rows it hands to the row atomics
    int stage_row[16][8];                   //         and their element offsets in grad_value (-1 = not a map pixel)
    int wave_tot[16];
    int ngen;
    int pad[3];
};
static_assert(sizeof(TileHeader) + sizeof(SortLds) <= 160 * 1024, "sorted scatter: LDS budget");
static_assert(kSortMaxPts <= 65536, "genlist holds 16-bit point indices");

// TV = storage type of grad_out.  grad_value is always accumulated in fp32 (bf16 mode: an fp32 scratch buffer, rounded once).
template <typename TV = float>
__global__ __launch_bounds__(kTiledThreads) void tiled_scatter_sorted_kernel(
    const float *__restrict__ loc, const float *__restrict__ aw, const TV *__restrict__ grad_out,
    float *__restrict__ grad_value, const TiledGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TileHeader *hdr = reinterpret_cast<TileHeader *>(smem);
    SortLds *S = reinterpret_cast<SortLds *>(smem + sizeof(TileHeader));

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int row_elems = g.M * kTD;
    const int LP = g.L * g.P;
    // work item = (image, head, region); its levels go through the tables one after the other, so the header and the
    // grad_out rows are set up once per region and the next level's operands are fetched while the current level is reduced
    const int n_items = kXcds * ((g.N * g.M + kXcds - 1) / kXcds) * g.GY * g.GX;
    for (int vb = blockIdx.x; vb < n_items; vb += gridDim.x) {   // persistent form, see tiled_gather_kernel
        int pair, region;
        if (!decode_block(vb, g.N * g.M, g.GY * g.GX, pair, region)) continue;
        const int b = pair / g.M, m = pair - b * g.M;
        const int gy = region / g.GX, gx = region - gy * g.GX;
        stamp<1>(g, 0);
        const int nq = build_header(hdr, g, gy, gx);
        stamp<1>(g, 1);
        // this thread's (up to) two sampling points per level: (query, point) = idx / P, idx % P
        unsigned pt_base[2];
        bool pt_live[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + u * kTiledThreads;
            pt_live[u] = idx < nq * g.P;
            const int qi = pt_live[u] ? idx / g.P : 0, pp = pt_live[u] ? idx - qi * g.P : 0;
            pt_base[u] = (unsigned)((b * g.Lq + hdr->qid[qi]) * g.M + m) * (unsigned)LP + (unsigned)pp;
        }
        float2 nxt_xy[2];
        float nxt_a[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {   // level 0
            nxt_xy[u] = *reinterpret_cast<const float2 *>(loc + 2u * pt_base[u]);
            nxt_a[u] = aw[pt_base[u]];
        }
        // ---- A: grad_out rows -> LDS ------------------------------------------------------------------------------
        for (int i = tid >> 3; i < nq; i += kTiledThreads / 8) {
            const unsigned item = (unsigned)((b * g.Lq + hdr->qid[i]) * g.M + m);
            *reinterpret_cast<float4 *>(S->gcache + i * kTD + 4 * (tid & 7)) = ld4(grad_out + item * (unsigned)kTD + 4u * (tid & 7));
        }
        for (int lv = 0; lv < g.L; ++lv) {
        const int H = uni(hdr->H[lv]), W = uni(hdr->W[lv]), nwc = uni(hdr->r[lv].nwc), nwr = uni(hdr->r[lv].nwr);
        const int wr0 = uni(hdr->r[lv].wr0), wc0 = uni(hdr->r[lv].wc0);
        const int npx = nwr * nwc;
        const int base_row = (b * g.S + uni(hdr->start[lv])) * row_elems + m * kTD;
        for (int i = tid; i <= npx; i += kTiledThreads) S->offs[i] = 0;   // clear the histogram
        if (tid == 0) S->ngen = 0;
        float2 cur_xy[2];
        float cur_a[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            cur_xy[u] = nxt_xy[u];
            cur_a[u] = nxt_a[u];
            if (lv + 1 < g.L) {   // in flight until the next level's step B
                nxt_xy[u] = *reinterpret_cast<const float2 *>(loc + 2u * (pt_base[u] + (unsigned)((lv + 1) * g.P)));
                nxt_a[u] = aw[pt_base[u] + (unsigned)((lv + 1) * g.P)];
            }
        }
        __syncthreads();

        // ---- B: resolve; every corner of an in-window point takes a rank in the list of its destination pixel ------------
        int r_pix[2];
        int r_rank[2][4];
        float r_w[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + u * kTiledThreads;
            r_pix[u] = -1;
            if (pt_live[u]) {
                const float2 xy = cur_xy[u];
                const float a = cur_a[u];
                const float h_im = xy.y * (float)H - 0.5f, w_im = xy.x * (float)W - 0.5f;
                if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                    const int rr = (int)hf - wr0, cc = (int)wf - wc0;
                    if (rr >= 0 && rr + 1 < nwr && cc >= 0 && cc + 1 < nwc) {   // all four corners in the window (apron incl.)
                        r_pix[u] = rr * nwc + cc;
                        r_w[u][0] = hh * hw * a;
                        r_w[u][1] = hh * lw * a;
                        r_w[u][2] = lh * hw * a;
                        r_w[u][3] = lh * lw * a;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            // corners on the apron (outside the map) are dropped here: their pixels are never flushed
                            const int row = (int)hf + (k >> 1), col = (int)wf + (k & 1);
                            const bool in_map = row >= 0 && row < H && col >= 0 && col < W;
                            r_rank[u][k] = in_map ? atomicAdd(&S->offs[r_pix[u] + (k >> 1) * nwc + (k & 1)], 1) : -1;
                        }
                    } else {
                        S->genlist[atomicAdd(&S->ngen, 1)] = (unsigned short)idx;
                    }
                }
            }
        }
        __syncthreads();
        stamp<1>(g, 2);

        // ---- C: exclusive scan of the histogram (two entries per thread; kSortMaxPx <= 2 * 1024) --------------------------------
        {
            const int e0 = 2 * tid, e1 = 2 * tid + 1;
            const int c0 = e0 < npx ? S->offs[e0] : 0, c1 = e1 < npx ? S->offs[e1] : 0;
            int incl = c0 + c1;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const int t = __shfl_up(incl, d, kWave);
                if (lane >= d) incl += t;
            }
            if (lane == kWave - 1) S->wave_tot[wave] = incl;
            __syncthreads();
            int base = 0;
#pragma unroll
            for (int w = 0; w < kTiledThreads / kWave; ++w) base += w < wave ? S->wave_tot[w] : 0;
            const int excl = base + incl - (c0 + c1);
            if (e0 <= npx) S->offs[e0] = excl;                 // e0 == npx / e1 == npx write the total (sentinel)
            if (e1 <= npx) S->offs[e1] = excl + c0;
        }
        __syncthreads();
        // ---- D: records to their sorted slots -----------------------------------------------------------------------------------
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (r_pix[u] >= 0) {
                const int qi = (tid + u * kTiledThreads) / g.P;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (r_rank[u][k] >= 0)
                        S->sorted[S->offs[r_pix[u] + (k >> 1) * nwc + (k & 1)] + r_rank[u][k]] = SortRec{qi, r_w[u][k]};
            }
        __syncthreads();
        stamp<1>(g, 3);

        // ---- E: per output pixel, 8 lanes x 4 channels; then transpose to one channel per lane and add rows ---------------------
        {
            const int j8 = lane & 7, grp8 = tid >> 3;
            for (int px0 = 0; px0 < npx; px0 += kTiledThreads / 8) {   // wave-uniform trip count
                const int px = px0 + grp8;
                const int rr = px / nwc, cc = px - rr * nwc;
                const int row = wr0 + rr, col = wc0 + cc;
                const bool valid = px < npx && row >= 0 && row < H && col >= 0 && col < W;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid) {
                    const int e1 = S->offs[px + 1];
                    int e = S->offs[px];
                    // several entries in flight: the entry -> gcache row dependency is the latency chain of this loop
                    for (; e + 3 < e1; e += 4) {
                        SortRec r[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) r[u] = S->sorted[e + u];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float4 gv = *reinterpret_cast<const float4 *>(S->gcache + r[u].q * kTD + 4 * j8);
                            acc.x += r[u].w * gv.x;
                            acc.y += r[u].w * gv.y;
                            acc.z += r[u].w * gv.z;
                            acc.w += r[u].w * gv.w;
                        }
                    }
                    for (; e < e1; ++e) {
                        const SortRec r = S->sorted[e];
                        const float4 gv = *reinterpret_cast<const float4 *>(S->gcache + r.q * kTD + 4 * j8);
                        acc.x += r.w * gv.x;
                        acc.y += r.w * gv.y;
                        acc.z += r.w * gv.z;
                        acc.w += r.w * gv.w;
                    }
                }
                // hand-over through LDS: the lane group writes its pixel's 32 channels, then the wave re-reads one channel per
                // lane so that each global atomic instruction adds two whole 128-B rows (same-wave LDS traffic is in order;
                // the barriers only stop compiler motion)
                *reinterpret_cast<float4 *>(S->stage[wave] + (lane >> 3) * kTD + 4 * j8) = acc;
                if (j8 == 0) S->stage_row[wave][lane >> 3] = valid ? base_row + (row * W + col) * row_elems : -1;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 4; ++r) {   // lower / upper half-wave: pixel slot 2r / 2r + 1
                    const int slot = 2 * r + (lane >> 5);
                    const float v = S->stage[wave][slot * kTD + (lane & 31)];
                    const int ro = S->stage_row[wave][slot];
                    if (ro >= 0 && v != 0.f) atomicAdd(grad_value + ro + (lane & 31), v);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        stamp<1>(g, 4);

        // ---- F: points with a corner outside the window: row atomics straight to global memory ----------------------------------
        {
            const int j = tid & (kTD - 1), grp32 = tid / kTD;
            const int ngen = S->ngen;
            for (int gi = grp32; gi < ngen; gi += kTiledThreads / kTD) {
                const int idx = S->genlist[gi];
                const int qi = idx / g.P, pp = idx - qi * g.P;
                const unsigned item = (unsigned)((b * g.Lq + hdr->qid[qi]) * g.M + m);
                const unsigned pt = item * (unsigned)LP + (unsigned)(lv * g.P + pp);
                const float2 xy = *reinterpret_cast<const float2 *>(loc + 2u * pt);
                const float a = aw[pt];
                int o[4];
                float lh, lw;
                resolve_point<float>(xy.x, xy.y, H, W, base_row, row_elems, o, lh, lw);
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float gk = S->gcache[qi * kTD + j] * a;
                const float ww[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
#pragma unroll
                for (int cn = 0; cn < 4; ++cn)
                    if (o[cn] >= 0) atomicAdd(grad_value + o[cn] + j, ww[cn] * gk);
            }
        }
        stamp<1>(g, 5);
        __syncthreads();   // the next level clears the tables; the next item rebuilds the header
        }
    }
}

// ---- host entry points ----------------------------------------------------------------------------------------------
inline TiledPlan plan_gather(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes, const int64_t *lsi)
{
    TiledPlan pl = plan_tiled(N, S, M, D, L, Lq, P, shapes, lsi, tiled_options().region_px, tiled_options().margin, kFwdLdsBudget,
                              kFwdGC * (int)sizeof(float));
    // the forward's fix-up of window misses reuses the window memory: a list of queries and one partial row per query
    const size_t fix = sizeof(TileHeader) + 2 * sizeof(int) * kMaxRegionQueries + (size_t)pl.max_q * kFwdGC * sizeof(float);
    if (pl.ok && pl.lds_bytes < fix) pl.lds_bytes = fix;
    return pl;
}
inline TiledPlan plan_bwd_gather(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes, const int64_t *lsi)
{
    if (tiled_options().bwd_halves && P <= 4) {   // channel halves: forward geometry, one level per work item
        TiledPlan pl = plan_tiled(N, S, M, D, L, Lq, P, shapes, lsi, tiled_options().region_px, tiled_options().margin,
                                  kFwdLdsBudget, kFwdGC * (int)sizeof(float));
        if (pl.ok) {
            int max_px = 0;
            for (int gy = 0; gy < pl.g.GY; ++gy)
                for (int gx = 0; gx < pl.g.GX; ++gx)
                    for (int l = 0; l < L; ++l) {
                        const LevelRect r = level_rect(pl.g.H[l], pl.g.W[l], gy, gx, pl.g.GY, pl.g.GX, pl.g.margin_l[l]);
                        max_px = r.nwr * r.nwc > max_px ? r.nwr * r.nwc : max_px;
                    }
            for (int l = 0; l < L; ++l) pl.g.phase[l] = l;
            pl.g.nphases = L;
            pl.lds_bytes = sizeof(TileHeader) + (size_t)max_px * kFwdGC * sizeof(float);
            pl.max_px = kFwdGC;   // marks the channel-half configuration
        }
        return pl;
    }
    TiledPlan pl = plan_tiled(N, S, M, D, L, Lq, P, shapes, lsi, tiled_options().region_px, tiled_options().margin, kBwdLdsBudget,
                              kBwdGC * (int)sizeof(float));
    pl.max_px = kBwdGC;
    return pl;
}
inline TiledPlan plan_scatter_bfp(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes, const int64_t *lsi)
{
    // LDS: header + window (136 B per pixel) + one 20-B record per (query, point) of the region, within 160 KiB
    const int budget = 160 * 1024 - (int)sizeof(TileHeader) - 256;
    TiledPlan pl = plan_tiled(N, S, M, D, L, Lq, P, shapes, lsi, tiled_options().region_px, tiled_options().margin, budget,
                              kBfpPxBytes, P * (int)sizeof(BfpRec));
    if (pl.ok) {   // one level per phase = per work item; every window starts at LDS pixel 0
        int max_px = 0;
        for (int gy = 0; gy < pl.g.GY; ++gy)
            for (int gx = 0; gx < pl.g.GX; ++gx)
                for (int l = 0; l < L; ++l) {
                    const LevelRect r = level_rect(pl.g.H[l], pl.g.W[l], gy, gx, pl.g.GY, pl.g.GX, pl.g.margin_l[l]);
                    max_px = r.nwr * r.nwc > max_px ? r.nwr * r.nwc : max_px;
                }
        for (int l = 0; l < L; ++l) pl.g.phase[l] = l;
        pl.g.nphases = L;
        pl.lds_bytes = sizeof(TileHeader) + (size_t)max_px * kBfpPxBytes + (size_t)pl.max_q * P * sizeof(BfpRec);
        pl.grid = pl.grid;   // per (pair, region); the launch multiplies by the number of levels
        pl.max_px = max_px;
    }
    return pl;
}
inline TiledPlan plan_scatter_sorted(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes, const int64_t *lsi)
{
    TiledPlan pl;
    if (P > 4) return pl;   // two (query, point) records per thread
    pl = plan_tiled(N, S, M, D, L, Lq, P, shapes, lsi, tiled_options().region_px, tiled_options().margin, kSortMaxPx * 4, 4);
    if (pl.ok) {
        int max_px = 0;
        for (int gy = 0; gy < pl.g.GY; ++gy)
            for (int gx = 0; gx < pl.g.GX; ++gx)
                for (int l = 0; l < L; ++l) {
                    const LevelRect r = level_rect(pl.g.H[l], pl.g.W[l], gy, gx, pl.g.GY, pl.g.GX, pl.g.margin_l[l]);
                    max_px = r.nwr * r.nwc > max_px ? r.nwr * r.nwc : max_px;
                }
        if (max_px > kSortMaxPx) { pl.ok = false; return pl; }
        for (int l = 0; l < L; ++l) pl.g.phase[l] = l;
        pl.g.nphases = L;
        pl.lds_bytes = sizeof(TileHeader) + sizeof(SortLds);
    }
    return pl;
}
inline TiledPlan plan_scatter(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes, const int64_t *lsi)
{
    return plan_tiled(N, S, M, D, L, Lq, P, shapes, lsi, tiled_options().region_px, tiled_options().margin, kLdsBudgetBytes,
                      kSD * (int)sizeof(double));
}

template <typename T>
bool tiled_fwd_applicable(int, int, int, int, int, int, int, const int64_t *, const int64_t *, const T *, const T *)
{
    return false;
}
template <>
inline bool tiled_fwd_applicable<float>(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes,
                                        const int64_t *lsi, const float *value, const float *out)
{
    if ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(out)) & 15) return false;
    return plan_gather(N, S, M, D, L, Lq, P, shapes, lsi).ok;
}

template <typename T>
bool tiled_bwd_applicable(int, int, int, int, int, int, int, const int64_t *, const int64_t *, const T *, const T *,
                          const T *)
{
    return false;
}
template <>
inline bool tiled_bwd_applicable<float>(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes,
                                        const int64_t *lsi, const float *value, const float *grad_out,
                                        const float *grad_value)
{
    if ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(grad_out) |
         reinterpret_cast<uintptr_t>(grad_value)) & 15)
        return false;
    return plan_bwd_gather(N, S, M, D, L, Lq, P, shapes, lsi).ok && plan_scatter(N, S, M, D, L, Lq, P, shapes, lsi).ok;
}

// Persistent grid size: at most `cap` workgroups, a multiple of 8 (XCD affinity), and with cap/8 coprime to the number of
// sub-items per region -- otherwise a workgroup's stride through the item list would always land on the same kind of
// sub-item (e.g. always the three-level phase) and the work would be badly balanced.
inline int persistent_grid(int total, int cap, int nsub)
{
    if (cap <= 0 || total <= cap) return total;
    if (nsub == 1) {   // equal items: the smallest grid that needs no more rounds than `cap` workgroups would
        const int rounds = (total + cap - 1) / cap;
        const int need = (total + rounds - 1) / rounds;
        return (need + kXcds - 1) / kXcds * kXcds;
    }
    auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
    int g8 = cap / kXcds;
    while (g8 > 1 && gcd(g8, nsub) != 1) --g8;
    return g8 * kXcds;
}

// Raise a kernel's dynamic-LDS limit.  Done once per (device, kernel, size class): repeating the runtime call on every
// launch costs time and is not something to issue while the caller captures its stream into a graph.
inline hipError_t set_lds_limit(const void *fn, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> granted;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = granted[std::make_pair(dev, fn)];
    if (have >= bytes) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

// TV = float or bf16_t (storage of value / out / grad_out); loc / attn and their gradients are fp32
template <typename TV>
inline hipError_t launch_fwd_tiled_tv(const TV *value, const float *loc, const float *aw, TV *out, int N, int S, int M, int D,
                                      int L, int Lq, int P, const int64_t *shapes_h, const int64_t *lsi_h,
                                      unsigned *general_points, hipStream_t stream)
{
    TiledPlan pl = plan_gather(N, S, M, D, L, Lq, P, shapes_h, lsi_h);
    if (!pl.ok) return hipErrorInvalidValue;
    if (general_points) pl.g.stats = general_points;   // locality monitor (msda_api.hip); else the diagnostic override
    auto kern = P == 4 ? &tiled_gather_kernel<false, true, kFwdGC, 4, TV> : &tiled_gather_kernel<false, false, kFwdGC, 4, TV>;
    hipError_t e = set_lds_limit(reinterpret_cast<const void *>(kern), pl.lds_bytes);
    if (e != hipSuccess) return e;
    const int grid = persistent_grid(pl.grid * (kTD / kFwdGC), tiled_options().persist, kTD / kFwdGC);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kFwdGC == 16 ? 512 : 1024), pl.lds_bytes, stream, value, loc, aw,
                       (const TV *)nullptr, out, (float *)nullptr, (float *)nullptr, pl.g);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_fwd_tiled(const T *, const int64_t *, const int64_t *, const T *, const T *, T *, int, int, int, int,
                            int, int, int, const int64_t *, const int64_t *, unsigned *, hipStream_t)
{
    return hipErrorNotSupported;
}
template <>
inline hipError_t launch_fwd_tiled<float>(const float *value, const int64_t *, const int64_t *, const float *loc,
                                          const float *aw, float *out, int N, int S, int M, int D, int L, int Lq, int P,
                                          const int64_t *shapes_h, const int64_t *lsi_h, unsigned *general_points,
                                          hipStream_t stream)
{
    return launch_fwd_tiled_tv<float>(value, loc, aw, out, N, S, M, D, L, Lq, P, shapes_h, lsi_h, general_points, stream);
}

// grad_value_f32: where grad_value is accumulated (pre-zeroed by the caller): the output itself for fp32, an fp32 scratch
// buffer for bf16 storage.  The f64-window and integer-window scatter variants exist for fp32 storage only.
template <typename TV>
inline hipError_t launch_bwd_tiled_tv(const TV *value, const float *loc, const float *aw, const TV *grad_out,
                                      float *grad_value_f32, float *grad_loc, float *grad_aw, int N, int S, int M, int D,
                                      int L, int Lq, int P, const int64_t *shapes_h, const int64_t *lsi_h, hipStream_t stream);

template <>
inline hipError_t launch_bwd_tiled_tv<bf16_t>(const bf16_t *value, const float *loc, const float *aw, const bf16_t *grad_out,
                                              float *grad_value_f32, float *grad_loc, float *grad_aw, int N, int S, int M,
                                              int D, int L, int Lq, int P, const int64_t *shapes_h, const int64_t *lsi_h,
                                              hipStream_t stream)
{
    const TiledPlan pg = plan_bwd_gather(N, S, M, D, L, Lq, P, shapes_h, lsi_h);
    const TiledPlan pso = plan_scatter_sorted(N, S, M, D, L, Lq, P, shapes_h, lsi_h);
    if (!pg.ok || !pso.ok || pg.max_px == kFwdGC) return hipErrorNotSupported;
    auto kern = P == 4 ? &tiled_gather_kernel<true, true, kBwdGC, kBwdCPL, bf16_t> : &tiled_gather_kernel<true, false, kBwdGC, kBwdCPL, bf16_t>;
    hipError_t e = set_lds_limit(reinterpret_cast<const void *>(kern), pg.lds_bytes);
    if (e == hipSuccess) e = set_lds_limit(reinterpret_cast<const void *>(&tiled_scatter_sorted_kernel<bf16_t>), pso.lds_bytes);
    if (e != hipSuccess) return e;
    const int sgrid = persistent_grid(pso.grid, tiled_options().persist / 2, 1);
    hipLaunchKernelGGL(tiled_scatter_sorted_kernel<bf16_t>, dim3(sgrid), dim3(kTiledThreads), pso.lds_bytes, stream, loc, aw,
                       grad_out, grad_value_f32, pso.g);
    const int ggrid = persistent_grid(pg.grid * pg.g.nphases, tiled_options().persist / 2, pg.g.nphases);
    hipLaunchKernelGGL(kern, dim3(ggrid), dim3(1024), pg.lds_bytes, stream, value, loc, aw, grad_out, (bf16_t *)nullptr,
                       grad_loc, grad_aw, pg.g);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_bwd_tiled(const T *, const int64_t *, const int64_t *, const T *, const T *, const T *, T *, T *, T *,
                            int, int, int, int, int, int, int, const int64_t *, const int64_t *, hipStream_t)
{
    return hipErrorNotSupported;
}
template <>
inline hipError_t launch_bwd_tiled<float>(const float *value, const int64_t *, const int64_t *, const float *loc,
                                          const float *aw, const float *grad_out, float *grad_value, float *grad_loc,
                                          float *grad_aw, int N, int S, int M, int D, int L, int Lq, int P,
                                          const int64_t *shapes_h, const int64_t *lsi_h, hipStream_t stream)
{
    const TiledPlan pg = plan_bwd_gather(N, S, M, D, L, Lq, P, shapes_h, lsi_h);
    const TiledPlan ps = plan_scatter(N, S, M, D, L, Lq, P, shapes_h, lsi_h);
    if (!pg.ok || !ps.ok) return hipErrorInvalidValue;
    const size_t lds_scatter = ps.lds_bytes;
    const bool halves = pg.max_px == kFwdGC;
    auto kern = halves ? (P == 4 ? &tiled_gather_kernel<true, true, kFwdGC, 4> : &tiled_gather_kernel<true, false, kFwdGC, 4>)
                       : (P == 4 ? &tiled_gather_kernel<true, true, kBwdGC, kBwdCPL> : &tiled_gather_kernel<true, false, kBwdGC, kBwdCPL>);
    hipError_t e = set_lds_limit(reinterpret_cast<const void *>(kern), pg.lds_bytes);
    if (e == hipSuccess) e = set_lds_limit(reinterpret_cast<const void *>(&tiled_scatter_kernel), lds_scatter);
    if (e != hipSuccess) return e;
    // grad_value (pre-zeroed by the caller of this function): LDS accumulation + one flush per touched pixel
    const TiledPlan pb = plan_scatter_bfp(N, S, M, D, L, Lq, P, shapes_h, lsi_h);
    const TiledPlan pso = plan_scatter_sorted(N, S, M, D, L, Lq, P, shapes_h, lsi_h);
    if (tiled_options().accum == 2 && pso.ok) {
        e = set_lds_limit(reinterpret_cast<const void *>(&tiled_scatter_sorted_kernel<float>), pso.lds_bytes);
        if (e != hipSuccess) return e;
        const int sgrid = persistent_grid(pso.grid, tiled_options().persist / 2, 1);
        hipLaunchKernelGGL(tiled_scatter_sorted_kernel<float>, dim3(sgrid), dim3(kTiledThreads), pso.lds_bytes, stream, loc, aw,
                           grad_out, grad_value, pso.g);
    } else if (tiled_options().accum == 1 && pb.ok) {
        auto skern = P == 4 ? &tiled_scatter_bfp_kernel<true> : &tiled_scatter_bfp_kernel<false>;
        e = set_lds_limit(reinterpret_cast<const void *>(skern), pb.lds_bytes);
        if (e != hipSuccess) return e;
        const int sgrid = persistent_grid(pb.grid * pb.g.nphases, tiled_options().persist / 2, pb.g.nphases);
        hipLaunchKernelGGL(skern, dim3(sgrid), dim3(kTiledThreads), pb.lds_bytes, stream, loc, aw, grad_out,
                           grad_value, pb.g, pb.max_px);
    } else {
        const int sgrid = persistent_grid(ps.grid * (kTD / kSD) * ps.g.nphases, tiled_options().persist / 2,
                                          (kTD / kSD) * ps.g.nphases);
        hipLaunchKernelGGL(tiled_scatter_kernel, dim3(sgrid), dim3(kTiledThreads), lds_scatter, stream, loc, aw, grad_out,
                           grad_value, ps.g);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // grad_sampling_loc, grad_attn_weight: gather from LDS windows of value
    const int ggrid = persistent_grid(pg.grid * pg.g.nphases, halves ? tiled_options().persist.load() : tiled_options().persist.load() / 2,
                                      pg.g.nphases);
    hipLaunchKernelGGL(kern, dim3(ggrid), dim3(halves ? 512 : 1024), pg.lds_bytes, stream, value, loc, aw, grad_out,
                       (float *)nullptr, grad_loc, grad_aw, pg.g);
    return hipGetLastError();
}

}  // namespace msda
