// msda_tiled.h -- LDS-window MSDeformAttn FORWARD kernel for gfx950, for encoder-shaped calls
// (queries = the pixels of the pyramid, Lq == S; D = 32 fp32 or bf16 storage; L <= 4): the shape that carries
// >95 % of the path's bytes (call "E").
//
// Why: in an encoder call every query samples around its own position, at every level.  The direct
// kernel fetches each of the 64 bilinear corners of a (query, head) from L2 (2.9 GB of 128-B row
// requests per call).  Here a workgroup owns ONE image REGION of ONE (image, head) pair:
//   * its queries are the pixels of every level whose centre falls in the region (~20x19 level-0
//     pixels), so they share their sampling neighbourhoods;
//   * per level it stages the window (region footprint +- margin) of the value slice in LDS --
//     64 B per pixel and channel half, filled with whole-row 16-B loads -- and gathers the corners from LDS.
// The window is only a cache: every corner is tested against it and points that miss it (large learned
// offsets) are finished per point from global memory after the item, so results are exact for ARBITRARY
// sampling locations; only speed depends on locality (msda_api.hip's locality monitor switches to the direct kernel).
// The query <-> region assignment is pure geometry on the level sizes; it needs the call to be
// encoder-shaped only to be profitable, never to be correct.
// (The backward of encoder-shaped calls is msda_rps.h: routed, pixel-stationary, no windows.)
//
// Semantics: identical to msda_direct.h (spec: reference ms_deform_im2col_cuda.cuh:33-84, 237-299).
#pragma once

#include <atomic>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "msda_common.h"

namespace msda {

constexpr int kTL = 4;               // levels supported by the tiled kernels
constexpr int kTD = 32;              // channels per head
constexpr int kMaxRegionQueries = 512;
constexpr int kMaxGrid = 24;         // region rows / columns covered by the host-made geometry tables
// A workgroup works on ONE CHANNEL HALF at a time -- 16 channels = 64 B per window pixel -- so that its LDS stays under half a
// CU's and two workgroups share a CU: one waits on its window fill (HBM-paced) while the other gathers.
constexpr int kFwdGC = 16;           // channels per workgroup (a channel half), 4 lanes per query, 512 threads
constexpr int kFwdLdsBudget = 75 * 1024;   // + header < 80 KiB: two workgroups per CU
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

static __device__ __attribute__((aligned(16))) unsigned g_tiled_zero_line[4];      // (zero-initialised: the source of apron pixels in the DMA fill)

// Set by msda_set_option (any thread), read by every launch (any thread): plain atomics, like the other options.
struct TiledOptions {
    std::atomic<int> region_px{20};   // finest-level pixels per region side (swept on MI355X: 20 beats 16 by ~15 %; larger does not fit LDS)
    std::atomic<int> margin{6};
    std::atomic<int> persist{512};    // 0 = one workgroup per work item; n > 0 = at most n workgroups (n/2 for the 1024-thread kernels)
                                      // walking the items (2 x 256 CUs by default: no per-item launch ramp)
    std::atomic<int> grow{1};         // 1 = windows grow into the LDS their phase leaves unused (per-level margins)
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

// ---- forward ------------------------------------------------------------------------------------------------
// 4 lanes x 4 channels per query; lane i of each quad resolves sampling point pc+i and the quad shares it by
// DPP broadcast.
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

// ---- fused module path: the kernel reads the RAW projection of the query -- offsets and attention logits,
// one row of M * L * P * 3 values per (image, query) -- and the reference points instead of sampling_loc / attn_weight: the softmax over a
// (query, head)'s L * P logits (reference ops/modules/ms_deform_attn.py:100) is formed by the quad that owns the query -- lane i holds
// point i of every level: a maximum and a sum over its own L values, then two DPP steps over the quad --, the location arithmetic
// (:102-109) by the lane that resolves the point.  sampling_loc / attn_weight are written as BY-PRODUCTS (what the backward needs) by
// the first channel half's workgroup only, or not at all (loc_out == nullptr: inference).  P = 4 only (a quad = the points of a level).
struct TiledPrepSrc {
    const void *offsets, *logits;      // TP = float or bf16_t: offsets[(n, q) * off_stride + (m * LP + pt) * 2 + {0, 1}], logits[(n, q) * log_stride + m * LP + pt]
    long long off_stride, log_stride;
    const float *ref;                  // (N, Lq, L, ref_dim)
    int ref_dim;                       // 2: reference points (x, y); 4: reference boxes (x, y, w, h)
    float *loc_out, *aw_out;           // by-products, or null
};

__device__ __forceinline__ float quad_max(float v)
{
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
}

template <typename TP>
__device__ __forceinline__ float tp_ld(const TP *p);
template <>
__device__ __forceinline__ float tp_ld<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ float tp_ld<bf16_t>(const bf16_t *p) { return __uint_as_float((unsigned)*reinterpret_cast<const unsigned short *>(p) << 16); }
template <typename TP>
__device__ __forceinline__ float2 tp_ld2(const TP *p);      // two consecutive elements (an offset pair): aligned to the pair
template <>
__device__ __forceinline__ float2 tp_ld2<float>(const float *p) { return *reinterpret_cast<const float2 *>(p); }
template <>
__device__ __forceinline__ float2 tp_ld2<bf16_t>(const bf16_t *p)
{
    const unsigned u = *reinterpret_cast<const unsigned *>(p);
    return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u));
}

// The raw operands of one level for the quad's queries: requested a level ahead (like LevelOps), turned into locations / weights where used.
template <int QPG, bool REF4>
struct LevelRaw {
    float2 off[QPG];                 // this lane's offset pair (point j & 3 of the level)
    float2 rxy[QPG];                 // the level's reference point
    float2 rwh[REF4 ? QPG : 1];      // ... and box size (reference boxes only)
    float lg[QPG];                   // this lane's logit
};

template <int QPG, typename TP, bool REF4>
__device__ __forceinline__ void load_level_raw(const TiledPrepSrc &src, const unsigned (&nq)[QPG], int m, int L, int LP, int lvl, int j,
                                               LevelRaw<QPG, REF4> &o)
{
    const int pt = m * LP + lvl * 4 + (j & 3);
#pragma unroll
    for (int k = 0; k < QPG; ++k) {
        o.off[k] = tp_ld2<TP>(reinterpret_cast<const TP *>(src.offsets) + (long long)nq[k] * src.off_stride + 2 * pt);
        o.lg[k] = tp_ld<TP>(reinterpret_cast<const TP *>(src.logits) + (long long)nq[k] * src.log_stride + pt);
        const float *r = src.ref + ((long long)nq[k] * L + lvl) * (REF4 ? 4 : 2);
        o.rxy[k] = *reinterpret_cast<const float2 *>(r);
        if constexpr (REF4) o.rwh[k] = *reinterpret_cast<const float2 *>(r + 2);
    }
}

// One sampled level for the kGatherQPG queries of a quad.  P4 = the level has exactly 4 points (RichSem): no
// point-count checks in the unrolled body.  Hot path = in-window points: one wave-divergent branch per point and
// straight-line LDS reads + packed FMAs.  Points with a corner outside the window are rare; they are handled
// afterwards in ONE run-time loop per query (not unrolled), with shuffles instead of DPP.
template <bool P4, int GC, int CPL, typename TV>
__device__ __forceinline__ void gather_level(
    const TV *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ aw, const float *win,
    const LevelCtx &lc, int row_elems, int P_, int j, int chan, const unsigned (&pt0)[GatherCfg<GC, CPL>::QPG],
    const bool (&live)[GatherCfg<GC, CPL>::QPG], const LevelOps<GatherCfg<GC, CPL>::QPG> &pre,
    v2f (&acc_lo)[GatherCfg<GC, CPL>::QPG], v2f (&acc_hi)[GatherCfg<GC, CPL>::QPG], unsigned &n_general, const int lvl,
    unsigned (&miss)[GatherCfg<GC, CPL>::QPG])
{
    constexpr bool defer = P4 && GatherCfg<GC, CPL>::GL == 4;
    constexpr int kGatherQPG = GatherCfg<GC, CPL>::QPG, NV = GatherCfg<GC, CPL>::NV;
    static_assert(NV == 1, "the accumulators hold four channels per lane");
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
            const float w1 = hh * hw * a, w2 = hh * lw * a, w3 = lh * hw * a, w4 = lh * lw * a;
            bool any_slow = false;
#define MSDA_POINT(I)                                                                                                 \
    if (P4 || pc + I < P) {                                                                                            \
        const int m_ = quad_bcast_i<I>(mode);                                                                          \
        any_slow |= m_ == -2;                                                                                          \
        if (m_ >= 0) {                                                                                                 \
            float4 v[4][NV];                                                                                           \
            lds_corners<GC, NV>(win, lc.nwc, j, m_, v);                                                                \
            fwd_accumulate(quad_bcast_f<I>(w1), quad_bcast_f<I>(w2), quad_bcast_f<I>(w3), quad_bcast_f<I>(w4),         \
                           v[0][0], v[1][0], v[2][0], v[3][0], acc_lo[k], acc_hi[k]);                                  \
        }                                                                                                              \
    }
            MSDA_POINT(0)
            MSDA_POINT(1)
            MSDA_POINT(2)
            MSDA_POINT(3)
#undef MSDA_POINT
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
        const float4 v0_ = o0 >= 0 ? ld4(value + o0 + chan) : z, v1_ = o1 >= 0 ? ld4(value + o1 + chan) : z;           \
        const float4 v2_ = o2 >= 0 ? ld4(value + o2 + chan) : z, v3_ = o3 >= 0 ? ld4(value + o3 + chan) : z;           \
        fwd_accumulate(quad_bcast_f<I>(w1), quad_bcast_f<I>(w2), quad_bcast_f<I>(w3), quad_bcast_f<I>(w4), v0_, v1_,   \
                       v2_, v3_, acc_lo[k], acc_hi[k]);                                                                \
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

// Workgroup = (image, head, region, channel half); accumulates over the LDS phases in registers.
// TV = storage type of value / out (float, or bf16_t: converted at the loads / the store; the LDS windows and all arithmetic
// are fp32 either way)
// TPREP = void: sampling_loc / attn_weight are inputs (the operator's entry points); float / bf16_t: the fused module path -- the raw
// projection of that type in `prep`, `loc` / `aw` unused (see TiledPrepSrc).
template <bool P4, int GC, int CPL, typename TV = float, typename TPREP = void, bool REF4 = false>
__global__ __launch_bounds__(GC == 16 ? 512 : 1024, GC == 16 ? 4 : 1) void tiled_gather_kernel(
    const TV *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ aw, TV *__restrict__ out,
    const TiledGeom g, const TiledPrepSrc prep)
{
    constexpr bool PREP = !std::is_same<TPREP, void>::value;
    static_assert(!PREP || (P4 && GC / CPL == 4), "fused prep: a quad owns a query and a level has four points");
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
    constexpr int nsub = kHalves;
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
    // value layout: the reference's (N, S, M, D) -- a pixel's row of one head every M * D elements -- or, as a measured experiment
    // (round 4, dbg bit 7; SURVEY.md section 8f rank 1 "head-major value layout"), (N, M, S, D): rows of a head contiguous
    const bool head_major = g.dbg & 128;
    const int row_elems = head_major ? kTD : g.M * kTD;
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

    // fused module path: (image, query) row of the quad's queries in the raw projection, and the softmax statistics of their L * P logits
    // (maximum, sum of exponentials: lane i of the quad holds point i of every level)
    unsigned nqv[kGatherQPG];
    float smax[kGatherQPG], ssum[kGatherQPG];
    if constexpr (PREP) {
#pragma unroll
        for (int k = 0; k < kGatherQPG; ++k) {
            nqv[k] = (unsigned)(b * g.Lq + hdr->qid[live[k] ? grp + k * kGroups : 0]);
            float x[kTL];
#pragma unroll
            for (int l = 0; l < kTL; ++l)
                x[l] = l < g.L ? tp_ld<TPREP>(reinterpret_cast<const TPREP *>(prep.logits) + (long long)nqv[k] * prep.log_stride + m * LP + l * 4 + (j & 3))
                               : -3.0e38f;
            const float mx = quad_max(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
            float e = 0.f;
#pragma unroll
            for (int l = 0; l < kTL; ++l) e += l < g.L ? expf(x[l] - mx) : 0.f;
            smax[k] = mx;
            ssum[k] = quad_sum(e);
        }
    }
    unsigned miss[kGatherQPG];   // bit (level*4 + point) = that point of query k missed its window (this lane's point only)
#pragma unroll
    for (int k = 0; k < kGatherQPG; ++k) miss[k] = 0u;
    constexpr bool defer = P4 && GL == 4;   // (compile-time: the in-loop general path is not even compiled into this kernel)
    const int ph_end = g.nphases;
    int st = 2;
    {
        const int half = sub;
        const int chan = half * GC + CPL * j;   // this lane's first channel inside the head
        for (int ph = 0; ph < ph_end; ++ph) {
            // levels of this phase are consecutive: [lb, le)
            int lb = g.L, le = 0;
            for (int l = 0; l < g.L; ++l)
                if (uni(hdr->phase[l]) == ph) { lb = l < lb ? l : lb; le = l + 1; }
            LevelOps<kGatherQPG> nxt;
            LevelRaw<PREP ? kGatherQPG : 1, REF4> rnxt;
            if constexpr (PREP) load_level_raw<kGatherQPG, TPREP, REF4>(prep, nqv, m, g.L, LP, lb, j, rnxt);   // in flight during the fill
            else load_level_ops(loc, aw, item, (unsigned)LP, (unsigned)(lb * g.P), g.P, j, nxt);
            // ---- stage this phase's windows: 64-B pixel half-rows, 16 B per lane -------------------------------------
            for (int l = lb; l < le; ++l) {
#if defined(FWD_ABLATE) && FWD_ABLATE == 1      // (diagnostic: no window fills -- wrong results)
                continue;
#endif
                const int wr0 = uni(hdr->r[l].wr0), wc0 = uni(hdr->r[l].wc0), nwc = uni(hdr->r[l].nwc);
                const int npx = uni(hdr->r[l].nwr) * nwc, Wl = uni(hdr->W[l]), Hl = uni(hdr->H[l]);
                const TV *src = value + (head_major ? ((int64_t)(b * g.M + m) * g.S + uni(hdr->start[l])) * kTD
                                                    : ((int64_t)(b * g.S + uni(hdr->start[l])) * g.M + m) * kTD) + half * GC + 4 * fj;
                float *dst = win + (int64_t)uni(hdr->lds_px[l]) * GC + 4 * fj;
                // kFillBatch independent loads in flight per lane before the first LDS store.  Straight-line code (round 4): the pixel index
                // is clamped for the load AND for the store -- lanes past the window's end re-store its last pixel, the same value to the
                // same place -- so that no branch surrounds a store: with `if (px < npx)` around it hipcc moved the LOAD under the branch
                // too and waited for it there (one or two loads in flight instead of four); the pixel's row comes from a reciprocal
                // (exact: (px + 0.5) / nwc is at least 0.5 / nwc away from an integer), not from a 20-instruction integer division.
                const float inv_nwc = 1.0f / (float)nwc;
                if constexpr (sizeof(TV) == 4 && FL == 4) {
                    if (!(g.dbg & 256)) {
                        // fp32 value (round 4): the window rows by LDS DMA: lane l of a wave writes 16 B at (base + 16 l) = pixel l / 4, quarter
                        // l % 4 of the wave's 16 consecutive window pixels -- no staging registers, no LDS store instructions; lanes past
                        // the window's end are masked off, pixels of the zero apron fetch a zero line.  114.2 -> 108.5 us at the init pattern,
                        // 145.5 -> 140.5 us at sigma = 4 (tools/r04_dma_fill.py, same box; tile_debug bit 8 = the register-staged fill below,
                        // which bf16 storage keeps: its rows are converted on the way)
                        const int wave_px = (tid >> 6) * 16;
                        for (int px0 = 0; px0 < npx; px0 += kFillGroups) {
                            const int px = px0 + fgrp;
                            if (px < npx) {
                                const int rr = (int)(((float)px + 0.5f) * inv_nwc), cc = px - rr * nwc;
                                const int row = wr0 + rr, col = wc0 + cc;
                                const bool in = row >= 0 && row < Hl && col >= 0 && col < Wl;
                                const float *sp = in ? reinterpret_cast<const float *>(src) + (int64_t)(row * Wl + col) * row_elems
                                                     : reinterpret_cast<const float *>(g_tiled_zero_line);
                                float *dp = win + (int64_t)(uni(hdr->lds_px[l]) + px0 + wave_px) * GC;
                                __builtin_amdgcn_global_load_lds(sp, reinterpret_cast<__attribute__((address_space(3))) void *>(reinterpret_cast<uintptr_t>(dp)),
                                                                 16, 0, 0);
                            }
                        }
                        continue;
                    }
                }
                for (int px0 = fgrp; px0 < npx; px0 += kFillBatch * kFillGroups) {
                    float4 v[kFillBatch];
                    int pxs[kFillBatch];
                    bool inm[kFillBatch];
#pragma unroll
                    for (int u = 0; u < kFillBatch; ++u) {
                        const int px = min(px0 + u * kFillGroups, npx - 1);
                        const int rr = (int)(((float)px + 0.5f) * inv_nwc), cc = px - rr * nwc;
                        const int row = wr0 + rr, col = wc0 + cc;
                        inm[u] = row >= 0 && row < Hl && col >= 0 && col < Wl;   // else: the zero apron
                        const int rowc = min(max(row, 0), Hl - 1), colc = min(max(col, 0), Wl - 1);
                        pxs[u] = px;
                        v[u] = ld4(src + (int64_t)(rowc * Wl + colc) * row_elems);
                    }
#pragma unroll
                    for (int u = 0; u < kFillBatch; ++u)
                        *reinterpret_cast<float4 *>(dst + pxs[u] * GC) = inm[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (sizeof(TV) == 4 && FL == 4 && !(g.dbg & 256)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the DMA fill's requests have landed)
            __syncthreads();
            stamp<2>(g, st++);

            // ---- gather -------------------------------------------------------------------------------------
            for (int l = lb; l < le; ++l) {
#if defined(FWD_ABLATE) && FWD_ABLATE == 2      // (diagnostic: no gather -- wrong results)
                continue;
#endif
                LevelOps<kGatherQPG> cur = nxt;
                [[maybe_unused]] const LevelRaw<PREP ? kGatherQPG : 1, REF4> rcur = rnxt;
                if (l + 1 < le) {
                    if constexpr (PREP) load_level_raw<kGatherQPG, TPREP, REF4>(prep, nqv, m, g.L, LP, l + 1, j, rnxt);
                    else load_level_ops(loc, aw, item, (unsigned)LP, (unsigned)((l + 1) * g.P), g.P, j, nxt);
                }
                if constexpr (PREP) {
                    // softmax weight and sampling location of this lane's point (reference ms_deform_attn.py:100, :102-109; the operation
                    // order of msda_prep.h's prep_forward_kernel), written out once -- by the first channel half -- for the backward
                    const float Wl = (float)uni(hdr->W[l]), Hl = (float)uni(hdr->H[l]);
#pragma unroll
                    for (int k = 0; k < kGatherQPG; ++k) {
                        const float a = expf(rcur.lg[k] - smax[k]) / ssum[k];
                        float lx, ly;
                        if constexpr (!REF4) {
                            lx = rcur.rxy[k].x + rcur.off[k].x / Wl;
                            ly = rcur.rxy[k].y + rcur.off[k].y / Hl;
                        } else {
                            lx = rcur.rxy[k].x + rcur.off[k].x / 4.0f * rcur.rwh[k].x * 0.5f;
                            ly = rcur.rxy[k].y + rcur.off[k].y / 4.0f * rcur.rwh[k].y * 0.5f;
                        }
                        cur.xy[k] = make_float2(lx, ly);
                        cur.a[k] = a;
                        if (prep.loc_out && sub == 0 && live[k]) {
                            const unsigned pt = item[k] * (unsigned)LP + (unsigned)(l * 4 + (j & 3));
                            *reinterpret_cast<float2 *>(prep.loc_out + 2u * pt) = make_float2(lx, ly);
                            prep.aw_out[pt] = a;
                        }
                    }
                }
                LevelCtx lc;
                lc.H = uni(hdr->H[l]);
                lc.W = uni(hdr->W[l]);
                lc.wr0 = uni(hdr->r[l].wr0);
                lc.wc0 = uni(hdr->r[l].wc0);
                lc.nwr = uni(hdr->r[l].nwr);
                lc.nwc = uni(hdr->r[l].nwc);
                lc.lds_base = uni(hdr->lds_px[l]) * GC;
                lc.base_row = head_major ? ((b * g.M + m) * g.S + uni(hdr->start[l])) * kTD : (b * g.S + uni(hdr->start[l])) * row_elems + m * kTD;
                unsigned pt0[kGatherQPG];
#pragma unroll
                for (int k = 0; k < kGatherQPG; ++k) pt0[k] = item[k] * (unsigned)LP + (unsigned)(l * g.P);
                gather_level<P4, GC, CPL, TV>(value, loc, aw, win, lc, row_elems, g.P, j, chan, pt0, live, cur, acc_lo, acc_hi, n_general, l,
                                              miss);
            }
            // the next fill overwrites the windows (last phase: the barrier below, which also tells whether any point of the
            // workgroup missed its window, takes this one's place)
            if (!defer || ph + 1 < ph_end) __syncthreads();
            stamp<2>(g, st++);
        }
    }

    if (defer) {
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
                // (fused module path: the query's softmax statistics once more -- any lane group may be handed any query)
                [[maybe_unused]] float fmx = 0.f, fsum = 1.f;
                [[maybe_unused]] const long long fnq = (long long)b * g.Lq + hdr->qid[slot];
                if constexpr (PREP) {
                    float x[kTL];
#pragma unroll
                    for (int l = 0; l < kTL; ++l)
                        x[l] = l < g.L ? tp_ld<TPREP>(reinterpret_cast<const TPREP *>(prep.logits) + fnq * prep.log_stride + m * LP + l * 4 + (j & 3)) : -3.0e38f;
                    fmx = quad_max(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
                    float e = 0.f;
#pragma unroll
                    for (int l = 0; l < kTL; ++l) e += l < g.L ? expf(x[l] - fmx) : 0.f;
                    fsum = quad_sum(e);
                }
                while (mk) {
                    const int bit = __ffs((int)mk) - 1;
                    mk &= mk - 1u;
                    const int lv = bit >> 2, pp = bit & 3;
                    const unsigned pt = itm * (unsigned)LP + (unsigned)(lv * g.P + pp);
                    float2 xy;
                    float a;
                    if constexpr (PREP) {
                        const int rp = m * LP + lv * 4 + pp;
                        const float2 of = tp_ld2<TPREP>(reinterpret_cast<const TPREP *>(prep.offsets) + fnq * prep.off_stride + 2 * rp);
                        a = expf(tp_ld<TPREP>(reinterpret_cast<const TPREP *>(prep.logits) + fnq * prep.log_stride + rp) - fmx) / fsum;
                        const float *r = prep.ref + (fnq * g.L + lv) * (REF4 ? 4 : 2);
                        if constexpr (!REF4) xy = make_float2(r[0] + of.x / (float)hdr->W[lv], r[1] + of.y / (float)hdr->H[lv]);
                        else xy = make_float2(r[0] + of.x / 4.0f * r[2] * 0.5f, r[1] + of.y / 4.0f * r[3] * 0.5f);
                    } else {
                        xy = *reinterpret_cast<const float2 *>(loc + 2u * pt);
                        a = aw[pt];
                    }
                    int o[4];
                    float lh, lw;
                    resolve_point<float>(xy.x, xy.y, hdr->H[lv], hdr->W[lv],
                                         head_major ? ((b * g.M + m) * g.S + hdr->start[lv]) * kTD : (b * g.S + hdr->start[lv]) * row_elems + m * kTD,
                                         row_elems, o, lh, lw);
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
    {
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

// TV = float or bf16_t (storage of value / out); loc / attn are fp32
template <typename TV>
inline hipError_t launch_fwd_tiled_tv(const TV *value, const float *loc, const float *aw, TV *out, int N, int S, int M, int D,
                                      int L, int Lq, int P, const int64_t *shapes_h, const int64_t *lsi_h,
                                      unsigned *general_points, hipStream_t stream)
{
    TiledPlan pl = plan_gather(N, S, M, D, L, Lq, P, shapes_h, lsi_h);
    if (!pl.ok) return hipErrorInvalidValue;
    if (general_points) pl.g.stats = general_points;   // locality monitor (msda_api.hip); else the diagnostic override
    auto kern = P == 4 ? &tiled_gather_kernel<true, kFwdGC, 4, TV> : &tiled_gather_kernel<false, kFwdGC, 4, TV>;
    hipError_t e = set_lds_limit(reinterpret_cast<const void *>(kern), pl.lds_bytes);
    if (e != hipSuccess) return e;
    const int grid = persistent_grid(pl.grid * (kTD / kFwdGC), tiled_options().persist, kTD / kFwdGC);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kFwdGC == 16 ? 512 : 1024), pl.lds_bytes, stream, value, loc, aw, out, pl.g, TiledPrepSrc{});
    return hipGetLastError();
}

// Fused module path (P = 4): raw projection + reference points in, out (+ sampling_loc / attn_weight as by-products) out.
template <typename TV, typename TP>
inline hipError_t launch_fwd_tiled_prep(const TV *value, const TiledPrepSrc &src, TV *out, int N, int S, int M, int D, int L, int Lq, int P,
                                        const int64_t *shapes_h, const int64_t *lsi_h, unsigned *general_points, hipStream_t stream)
{
    if (P != 4) return hipErrorNotSupported;
    TiledPlan pl = plan_gather(N, S, M, D, L, Lq, P, shapes_h, lsi_h);
    if (!pl.ok) return hipErrorNotSupported;
    if (general_points) pl.g.stats = general_points;
    auto kern = src.ref_dim == 4 ? &tiled_gather_kernel<true, kFwdGC, 4, TV, TP, true> : &tiled_gather_kernel<true, kFwdGC, 4, TV, TP, false>;
    hipError_t e = set_lds_limit(reinterpret_cast<const void *>(kern), pl.lds_bytes);
    if (e != hipSuccess) return e;
    const int grid = persistent_grid(pl.grid * (kTD / kFwdGC), tiled_options().persist, kTD / kFwdGC);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), pl.lds_bytes, stream, value, (const float *)nullptr, (const float *)nullptr, out, pl.g, src);
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

}  // namespace msda
