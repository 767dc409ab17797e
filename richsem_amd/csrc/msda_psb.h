// msda_psb.h -- "pixel-stationary" MSDeformAttn backward for gfx950 (fp32, D = 32, P <= 4, L <= 4).
//
// The reference's backward (ms_deform_im2col_cuda.cuh:87-159, 301-403) is query-stationary: a block owns a (query, head),
// and every bilinear corner of every sampling point is ADDED to grad_value with a global float atomic -- 2.9 GB of atomics
// per encoder call, ~2.3 ms at the chip's atomic rate.  Here the ownership is turned round: a workgroup owns a TILE OF
// OUTPUT PIXELS of one (image, head, level) and pulls in every sampling point that lands on it.
//
//   * eight lanes x four channels hold one pixel's value row and its grad_value accumulator IN REGISTERS for the whole
//     item (up to four pixels per lane group: a tile plus a one-pixel apron below / right of it is <= 512 pixels);
//   * the queries that can reach the tile ("candidates") are walked in chunks of 512: their grad_out rows go to LDS,
//     one lane per (query, point) resolves the sample, and every corner that falls on a pixel of the tile becomes an
//     8-byte entry {query, weight} SORTED into that pixel's list with integer LDS atomics (count, scan, place);
//   * every pixel then walks its list:  acc += w * grad_out[q]  (grad_value) and  D = <grad_out[q], value[pixel]>, the
//     "corner dot" that grad_sampling_loc / grad_attn_weight are linear combinations of (D overwrites w in the entry);
//   * the lane that resolved a point reads its four corner dots back and stores the point's two gradients.
// So grad_value is produced by plain stores, once per pixel -- no float atomics, no zero-fill, fp32 sums as in the
// reference -- and the value tensor is read exactly once (no halo: the halo is on the QUERY side, where it costs only
// the resolution of a point, not LDS capacity).
//
// Which queries are candidates is pure geometry for encoder-shaped calls (Lq == S, query i = pixel i of the pyramid):
// a query can only reach pixels within `margin` of the pixel under its own position.  A point that reaches further is
// "far": the tile that is home to the query appends it to a list and psb_far_kernel finishes it with the direct
// method (global row atomics), after this kernel.  Levels with margin < 0 and calls that are not encoder-shaped
// (decoder: 1092 queries anywhere) take every query as a candidate of every tile: nothing is far.  Coarse levels whose
// tile is the whole map split their candidates into slabs over several workgroups; those add their pixel rows to
// (pre-zeroed) grad_value with 128-B row atomics -- a few MB per call.
//
// Exact for arbitrary sampling locations: ownership rules, not distributions, decide who computes what.
#pragma once

#include <algorithm>
#include <atomic>
#include <vector>

#include "msda_common.h"

namespace msda {

constexpr int kPsbThreads = 1024;
constexpr int kPsbGroups = kPsbThreads / 8;     // 8-lane groups: one pixel row = 8 lanes x 4 channels
constexpr int kPsbSlots = 3;                    // base pixels per lane group (four partial sums each)
constexpr int kPsbMaxPx = kPsbGroups * kPsbSlots;   // tile + apron
constexpr int kPsbQC = 512;                     // candidate queries per chunk
constexpr int kPsbMaxP = 4;
constexpr int kPsbMaxL = 4;
constexpr int kPsbMaxEntries = kPsbQC * kPsbMaxP;
constexpr int kPsbMaxUnits = 224;
constexpr int kPsbD = 32;

struct PsbLevel {
    int H, W, start;
    int TH, TW, nty, ntx;   // tile grid: tile (ty, tx) = rows [ty*TH, min(H, ty*TH + TH)) x cols [tx*TW, ...)
    int mg;                 // how far (pixels of this level) a corner may lie from the pixel under its query; < 0: no limit
    int atomic;             // 1: several slabs per tile -> rows are added to pre-zeroed grad_value with atomics
    float sH[kPsbMaxL], sW[kPsbMaxL];   // pixel of this level under the centre of row r of query level lq: (int)((r+0.5f)*sH[lq])
};

struct PsbGeom {
    int N, S, M, Lq, L, P;
    int encoder;            // 1: query i is pixel i of the pyramid (candidates by geometry); 0: every query everywhere
    int nunits, ppx;        // work table; pairs per XCD queue = ceil(N*M / 8)
    PsbLevel lv[kPsbMaxL];
    unsigned units[kPsbMaxUnits];   // level | ty << 2 | tx << 8 | slab << 14 | nslab << 22, heaviest first
    unsigned *ctr;          // workspace: [0..7] per-XCD queue heads, [8] number of far points
    unsigned *far_list;     // workspace: global point indices of the far points
    unsigned long long *stamps;   // diagnostic runs only (msda_debug_stamps): per workgroup 16 x 8 B of shader-clock sums per stage
};

struct alignas(16) PsbEnt {   // one sampling point in the list of its base pixel; overwritten by its four corner dots
    int q;                    // row of gcache | base-grid index << 16
    float lh, lw, a;          // bilinear fractions, attention weight
};

struct PsbLds {
    int rect_lo[2 * kPsbMaxL], rect_hi[2 * kPsbMaxL];   // candidate rows / cols per query level
    int cand_pre[kPsbMaxL + 1], cand_nc[kPsbMaxL];
    int next_item;
    unsigned long long stamp_last, stamp_acc[15];
    int wave_tot[16];
    int pad[2];
    int qid[2][kPsbQC];       // global query index of the chunk's candidates (-1: none); double-buffered over chunks
    int qrc[2][kPsbQC];       // (pixel row << 16 | pixel col) under the candidate at the item's level
    int offs[kPsbMaxPx + 4];  // histogram, then exclusive prefix
    float gcache[kPsbQC * kPsbD];       // grad_out rows of the chunk; stage of the final fold
    PsbEnt ent[kPsbMaxEntries];
    float vtile[kPsbMaxPx * kPsbD];     // value rows of the tile + apron
};
static_assert(sizeof(PsbLds) <= 160 * 1024, "psb: LDS budget");

__device__ __forceinline__ float group8_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return v;
}

__device__ __forceinline__ int psb_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Diagnostic (g.stamps is null in normal runs): thread 0 adds the shader-clock time since the previous stamp to stage i.
#define PSB_STAMP(i)                                                                   \
    if (g.stamps && threadIdx.x == 0) {                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                  \
        S->stamp_acc[i] += now_ - S->stamp_last;                                       \
        S->stamp_last = now_;                                                          \
    }

// pixel (row or column) of a level under the centre of row / column r of another level
__device__ __forceinline__ int psb_under(int r, float scale) { return (int)(((float)r + 0.5f) * scale); }

// Zero the queue heads / far counter and the grad_value ranges of the levels that are flushed with atomics.
__global__ __launch_bounds__(256) void psb_prep_kernel(float *__restrict__ grad_value, const PsbGeom g)
{
    if (blockIdx.x == 0 && threadIdx.x < 16) g.ctr[threadIdx.x] = 0u;
    const int row4 = g.M * kPsbD / 4;   // float4 per pixel
    for (int l = 0; l < g.L; ++l) {
        if (!g.lv[l].atomic) continue;
        const int n4 = g.lv[l].H * g.lv[l].W * row4;
        for (int b = 0; b < g.N; ++b) {
            float4 *dst = reinterpret_cast<float4 *>(grad_value) + (size_t)(b * g.S + g.lv[l].start) * row4;
            for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x)
                dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

template <bool P4>
__global__ __launch_bounds__(kPsbThreads) void psb_kernel(
    const float *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ aw,
    const float *__restrict__ grad_out, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_aw, const PsbGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PsbLds *S = reinterpret_cast<PsbLds *>(smem);

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int j8 = tid & 7, grp = tid >> 3;
    const int P = P4 ? 4 : g.P;
    const int LP = g.L * P;
    const int row_elems = g.M * kPsbD;
    const int pairs = g.N * g.M;
    const int xq = blockIdx.x & (kXcds - 1);   // blocks equal mod 8 share an XCD (observed; speed only)
    const int n_items = g.nunits * g.ppx;
    if (g.stamps && tid == 0) {
        for (int i = 0; i < 15; ++i) S->stamp_acc[i] = 0;
        S->stamp_last = __builtin_amdgcn_s_memtime();
    }

    for (;;) {
        if (tid == 0) S->next_item = (int)atomicAdd(g.ctr + xq, 1u);
        __syncthreads();
        const int item = psb_uni(S->next_item);
        if (item >= n_items) break;
        const unsigned unit = g.units[item / g.ppx];
        const int pair = xq + kXcds * (item % g.ppx);
        if (pair >= pairs) { __syncthreads(); continue; }
        const int l = unit & 3, ty = (unit >> 2) & 63, tx = (unit >> 8) & 63, slab = (unit >> 14) & 255, nslab = (unit >> 22) & 255;
        const int b = pair / g.M, m = pair - b * g.M;
        const int H = g.lv[l].H, W = g.lv[l].W, mg = g.lv[l].mg;
        const int R0 = ty * g.lv[l].TH, R1 = min(H, R0 + g.lv[l].TH), C0 = tx * g.lv[l].TW, C1 = min(W, C0 + g.lv[l].TW);
        // Two grids of the same shape (th+1) x gw:
        //   base grid   (gr, gc) <-> sampling points whose corner (h_low, w_low) is pixel (R0-1+gr, C0-1+gc): one list each
        //   pixel grid  (vr, vc) <-> pixel (R0+vr, C0+vc): the tile plus one apron row / column (value rows for the dots)
        // so the four corners of base p are the pixels p-gw-1, p-gw, p-1, p of the pixel grid.
        const int gw = C1 - C0 + 1;
        const int npx = (R1 - R0 + 1) * gw;
        const bool geo = g.encoder && mg >= 0;            // candidates by geometry
        const bool first_tile = ty == 0 && tx == 0;

        // ---- candidates of this tile: per query level a rectangle of query pixels (or every query) ----------------------
        if (g.encoder && tid < 2 * g.L) {
            const int lq = tid >> 1, is_col = tid & 1;
            const int n = is_col ? g.lv[lq].W : g.lv[lq].H;
            int lo = 0, hi = n - 1;
            if (mg >= 0) {
                const float sc = is_col ? g.lv[l].sW[lq] : g.lv[l].sH[lq];
                // a query whose pixel-under is u has near points with h_low in [u-mg, u+mg-1]; this tile's lists take
                // h_low in [R0-1, R1-1]
                const int a = (is_col ? C0 : R0) - mg, z = (is_col ? C1 : R1) - 1 + mg;
                // psb_under is non-decreasing in r: first r with under(r) >= a, last r with under(r) <= z
                int x0 = 0, x1 = n;
                while (x0 < x1) { const int mid = (x0 + x1) >> 1; if (psb_under(mid, sc) >= a) x1 = mid; else x0 = mid + 1; }
                lo = x0;
                x0 = -1; x1 = n - 1;
                while (x0 < x1) { const int mid = (x0 + x1 + 1) >> 1; if (psb_under(mid, sc) <= z) x0 = mid; else x1 = mid - 1; }
                hi = x0;
            }
            S->rect_lo[tid] = lo;
            S->rect_hi[tid] = hi;
        }
        // ---- the tile's value rows (+ apron, zeros beyond the map) -> LDS ------------------------------------------------
        const int64_t lvl_base = ((int64_t)(b * g.S + g.lv[l].start) * g.M + m) * kPsbD + 4 * j8;
        for (int p = grp; p < npx; p += kPsbGroups) {
            const int row = R0 + p / gw, col = C0 + p % gw;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < H && col < W) t = *reinterpret_cast<const float4 *>(value + lvl_base + (int64_t)(row * W + col) * row_elems);
            *reinterpret_cast<float4 *>(S->vtile + p * kPsbD + 4 * j8) = t;
        }
        __syncthreads();
        if (tid == 0) {   // prefix counts of the candidate rectangles (kept in LDS: only the decode step below needs them)
            int pre = 0;
            for (int lq = 0; lq < kPsbMaxL; ++lq) {
                int nr = 0, nc = 0;
                if (g.encoder && lq < g.L) {
                    nr = max(0, S->rect_hi[2 * lq] - S->rect_lo[2 * lq] + 1);
                    nc = max(0, S->rect_hi[2 * lq + 1] - S->rect_lo[2 * lq + 1] + 1);
                }
                S->cand_pre[lq] = pre;
                S->cand_nc[lq] = nc > 0 ? nc : 1;
                pre += nr * nc;
            }
            S->cand_pre[kPsbMaxL] = pre;
        }
        __syncthreads();
        const int ncand = g.encoder ? psb_uni(S->cand_pre[kPsbMaxL]) : g.Lq;
        const int nchunks = (ncand + kPsbQC - 1) / kPsbQC;
        const int cps = (nchunks + nslab - 1) / nslab;
        const int c_begin = slab * cps, c_end = min(nchunks, c_begin + cps);

        // ---- this lane group's base pixels: four partial sums each (one per corner), in registers for the whole item -----------
        float4 acc[kPsbSlots][4];
#pragma unroll
        for (int s = 0; s < kPsbSlots; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[s][k] = make_float4(0.f, 0.f, 0.f, 0.f);

        // ---- chunk pipeline: the candidates of chunk ch+1 are decoded while chunk ch is sorted, and this lane's two sampling
        //      points of chunk ch+1 are fetched while chunk ch is reduced --------------------------------------------------------------
        auto decode = [&](int ch, int buf) {   // candidate -> (query, pixel under it at this level)
            if (tid < kPsbQC) {
                const int ci = ch * kPsbQC + tid;
                int q = -1, rc = 0;
                if (ci < ncand && ch < c_end) {
                    if (g.encoder) {
                        int lq = 0;
#pragma unroll
                        for (int t = 1; t < kPsbMaxL; ++t) lq += ci >= S->cand_pre[t] ? 1 : 0;
                        const int k = ci - S->cand_pre[lq], nc = S->cand_nc[lq];
                        const int rr = k / nc, cc = k - rr * nc;
                        const int qr = S->rect_lo[2 * lq] + rr, qc = S->rect_lo[2 * lq + 1] + cc;
                        q = g.lv[lq].start + qr * g.lv[lq].W + qc;
                        rc = (psb_under(qr, g.lv[l].sH[lq]) << 16) | psb_under(qc, g.lv[l].sW[lq]);
                    } else {
                        q = ci;
                    }
                }
                S->qid[buf][tid] = q;
                S->qrc[buf][tid] = rc;
            }
        };
        int n_pt[2], n_rc[2];     // next chunk: global point index (-1: none), pixel under the query
        float2 n_xy[2];
        float n_a[2];
        auto fetch_points = [&](int buf) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = tid + u * kPsbThreads;
                const int ql = P4 ? idx >> 2 : idx / P, pp = P4 ? idx & 3 : idx - ql * P;
                const int q = ql < kPsbQC ? S->qid[buf][ql] : -1;
                n_pt[u] = -1;
                n_rc[u] = 0;
                n_xy[u] = make_float2(-4.f, -4.f);
                n_a[u] = 0.f;
                if (q >= 0) {
                    n_pt[u] = ((b * g.Lq + q) * g.M + m) * LP + l * P + pp;
                    n_rc[u] = S->qrc[buf][ql];
                    n_xy[u] = *reinterpret_cast<const float2 *>(loc + 2u * (unsigned)n_pt[u]);
                    n_a[u] = aw[n_pt[u]];
                }
            }
        };
        decode(c_begin, 0);
        __syncthreads();
        fetch_points(0);
        PSB_STAMP(0)
        for (int ch = c_begin; ch < c_end; ++ch) {
            const int cur = (ch - c_begin) & 1;
            // ---- (1) grad_out rows of the chunk in flight; clear the histogram; decode the next chunk ----------------------------
            float4 grow[kPsbQC / kPsbGroups];
#pragma unroll
            for (int i = 0; i < kPsbQC / kPsbGroups; ++i) {
                const int q = S->qid[cur][grp + i * kPsbGroups];
                grow[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q >= 0) grow[i] = *reinterpret_cast<const float4 *>(grad_out + ((int64_t)(b * g.Lq + q) * g.M + m) * kPsbD + 4 * j8);
            }
            for (int i = tid; i <= npx; i += kPsbThreads) S->offs[i] = 0;
            decode(ch + 1, cur ^ 1);
            __syncthreads();
            PSB_STAMP(1)

            // ---- (2) one lane per (query, point): resolve, classify, take a rank in the list of the point's base pixel ------------
            int pos[2], pbase[2];   // rank, then slot in `ent`; base-grid index
            int p_pt[2];            // global point index when this workgroup writes the point's gradients, else -1
            float p_lh[2], p_lw[2], p_a[2];
            unsigned p_valid[2];    // corners inside the map (bit k), only for points whose gradients this workgroup writes
            bool is_far[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pos[u] = -1;
                pbase[u] = 0;
                p_lh[u] = p_lw[u] = 0.f;
                p_a[u] = n_a[u];
                p_valid[u] = 0;
                is_far[u] = false;
                p_pt[u] = -1;
                const int pt = n_pt[u];
                if (pt < 0) continue;
                const int ur = n_rc[u] >> 16, uc = n_rc[u] & 0xFFFF;     // pixel under the query (encoder mode)
                const bool home = geo ? (ur >= R0 && ur < R1 && uc >= C0 && uc < C1) : first_tile;
                const float h_im = n_xy[u].y * (float)H - 0.5f, w_im = n_xy[u].x * (float)W - 0.5f;
                if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) {
                    if (home) p_pt[u] = pt;   // dropped sample: zero gradients, written by the query's home tile
                    continue;
                }
                const float hf = floorf(h_im), wf = floorf(w_im);
                const int h_low = (int)hf, w_low = (int)wf;
                const bool near = !geo || (h_low >= ur - mg && h_low < ur + mg && w_low >= uc - mg && w_low < uc + mg);
                if (!near) {
                    is_far[u] = home;
                    if (home) pbase[u] = pt;   // (reused: the global point index for the far list)
                    continue;
                }
                if (h_low < R0 - 1 || h_low >= R1 || w_low < C0 - 1 || w_low >= C1) continue;   // no corner list of this tile
                const int br = max(h_low, 0), bc = max(w_low, 0);
                const bool owner = br >= R0 && br < R1 && bc >= C0 && bc < C1;
                p_lh[u] = h_im - hf;
                p_lw[u] = w_im - wf;
                if (owner) {
                    p_pt[u] = pt;
                    p_valid[u] = (h_low >= 0 && w_low >= 0 ? 1u : 0u) | (h_low >= 0 && w_low + 1 < W ? 2u : 0u) |
                                 (h_low + 1 < H && w_low >= 0 ? 4u : 0u) | (h_low + 1 < H && w_low + 1 < W ? 8u : 0u);
                }
                pbase[u] = (h_low - R0 + 1) * gw + (w_low - C0 + 1);
                pos[u] = atomicAdd(&S->offs[pbase[u]], 1);
            }
            // far points of queries that are at home here: one list append per wave
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned long long fm = __ballot(is_far[u]);
                if (fm) {
                    const int nfar = __popcll(fm);
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(g.ctr + 8, (unsigned)nfar);
                    base = (unsigned)psb_uni((int)base);
                    if (is_far[u]) g.far_list[base + __popcll(fm & ((1ull << lane) - 1ull))] = (unsigned)pbase[u];
                }
            }
            // the grad_out rows have had the resolve step to arrive
#pragma unroll
            for (int i = 0; i < kPsbQC / kPsbGroups; ++i)
                *reinterpret_cast<float4 *>(S->gcache + (grp + i * kPsbGroups) * kPsbD + 4 * j8) = grow[i];
            __syncthreads();
            PSB_STAMP(2)

            // ---- (3) exclusive scan of the per-list counts (<= 384: one per thread of the first 6 waves) --------------------------
            {
                int c = 0, incl = 0;
                if (tid < kPsbMaxPx) {
                    c = tid < npx ? S->offs[tid] : 0;
                    incl = c;
#pragma unroll
                    for (int d = 1; d < kWave; d <<= 1) {
                        const int t = __shfl_up(incl, d, kWave);
                        if (lane >= d) incl += t;
                    }
                    if (lane == kWave - 1) S->wave_tot[wave] = incl;
                }
                __syncthreads();
                if (tid < kPsbMaxPx) {
                    int base = 0;
#pragma unroll
                    for (int w = 0; w < kPsbMaxPx / kWave; ++w) base += w < wave ? S->wave_tot[w] : 0;
                    const int excl = base + incl - c;
                    if (tid <= npx) S->offs[tid] = excl;      // tid == npx: the total (c = 0 there)
                    if (tid == kPsbMaxPx - 1 && npx == kPsbMaxPx) S->offs[npx] = base + incl;
                }
            }
            __syncthreads();
            PSB_STAMP(3)

            // ---- (4) entries to their sorted slots ----------------------------------------------------------------------------
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (pos[u] >= 0) {
                    pos[u] += S->offs[pbase[u]];
                    S->ent[pos[u]] = PsbEnt{(tid + u * kPsbThreads) / P | pbase[u] << 16, p_lh[u], p_lw[u], p_a[u]};
                }
            __syncthreads();
            PSB_STAMP(4)

            // ---- (5) this lane's sampling points of the next chunk: in flight during the reduction -----------------------------------
            fetch_points(cur ^ 1);

            // ---- (6) every base pixel walks its list: four partial sums (one per corner) -----------------------------------------------
#pragma unroll
            for (int s = 0; s < kPsbSlots; ++s) {
                const int p = s * kPsbGroups + grp;
                if (p < npx) {
                    int e = S->offs[p];
                    const int e1 = S->offs[p + 1];
#define PSB_ADD_POINT(EN, GQ)                                                                                                    \
    {                                                                                                                            \
        const float hh = 1.f - EN.lh, hw = 1.f - EN.lw, ha = hh * EN.a, la = EN.lh * EN.a;                                       \
        const float w0 = ha * hw, w1 = ha * EN.lw, w2 = la * hw, w3 = la * EN.lw;                                                \
        acc[s][0].x += w0 * GQ.x; acc[s][0].y += w0 * GQ.y; acc[s][0].z += w0 * GQ.z; acc[s][0].w += w0 * GQ.w;                 \
        acc[s][1].x += w1 * GQ.x; acc[s][1].y += w1 * GQ.y; acc[s][1].z += w1 * GQ.z; acc[s][1].w += w1 * GQ.w;                 \
        acc[s][2].x += w2 * GQ.x; acc[s][2].y += w2 * GQ.y; acc[s][2].z += w2 * GQ.z; acc[s][2].w += w2 * GQ.w;                 \
        acc[s][3].x += w3 * GQ.x; acc[s][3].y += w3 * GQ.y; acc[s][3].z += w3 * GQ.z; acc[s][3].w += w3 * GQ.w;                 \
    }
                    for (; e + 1 < e1; e += 2) {
                        const PsbEnt en0 = S->ent[e], en1 = S->ent[e + 1];
                        const float4 g0 = *reinterpret_cast<const float4 *>(S->gcache + (en0.q & 0xFFFF) * kPsbD + 4 * j8);
                        const float4 g1 = *reinterpret_cast<const float4 *>(S->gcache + (en1.q & 0xFFFF) * kPsbD + 4 * j8);
                        PSB_ADD_POINT(en0, g0)
                        PSB_ADD_POINT(en1, g1)
                    }
                    if (e < e1) {
                        const PsbEnt en0 = S->ent[e];
                        const float4 g0 = *reinterpret_cast<const float4 *>(S->gcache + (en0.q & 0xFFFF) * kPsbD + 4 * j8);
                        PSB_ADD_POINT(en0, g0)
                    }
#undef PSB_ADD_POINT
                }
            }
            __syncthreads();
            PSB_STAMP(5)

            // ---- (7) corner dots, one sampling point per lane group in list order (balanced): D_k = <grad_out[q], value[corner k]>;
            //      lanes 0..3 write them over the entry ----------------------------------------------------------------------------------
            {
                const int n_ent = S->offs[npx];
                for (int e = grp; e < n_ent; e += kPsbGroups) {
                    const PsbEnt en = S->ent[e];
                    const int pb = en.q >> 16;
                    const float4 gq = *reinterpret_cast<const float4 *>(S->gcache + (en.q & 0xFFFF) * kPsbD + 4 * j8);
                    const int gr = pb / gw, gc = pb - gr * gw;
                    const int r0 = max(gr - 1, 0) * gw, c0 = max(gc - 1, 0);
                    const float4 v0 = *reinterpret_cast<const float4 *>(S->vtile + (r0 + c0) * kPsbD + 4 * j8);
                    const float4 v1 = *reinterpret_cast<const float4 *>(S->vtile + (r0 + gc) * kPsbD + 4 * j8);
                    const float4 v2 = *reinterpret_cast<const float4 *>(S->vtile + (gr * gw + c0) * kPsbD + 4 * j8);
                    const float4 v3 = *reinterpret_cast<const float4 *>(S->vtile + (gr * gw + gc) * kPsbD + 4 * j8);
                    float d0 = gq.x * v0.x + gq.y * v0.y + gq.z * v0.z + gq.w * v0.w;
                    float d1 = gq.x * v1.x + gq.y * v1.y + gq.z * v1.z + gq.w * v1.w;
                    float d2 = gq.x * v2.x + gq.y * v2.y + gq.z * v2.z + gq.w * v2.w;
                    float d3 = gq.x * v3.x + gq.y * v3.y + gq.z * v3.z + gq.w * v3.w;
                    d0 = group8_sum(d0);
                    d1 = group8_sum(d1);
                    d2 = group8_sum(d2);
                    d3 = group8_sum(d3);
                    const float dk = j8 == 0 ? d0 : (j8 == 1 ? d1 : (j8 == 2 ? d2 : d3));
                    if (j8 < 4) reinterpret_cast<float *>(S->ent + e)[j8] = dk;   // (all eight lanes have read the entry)
                }
            }
            __syncthreads();
            PSB_STAMP(6)

            // ---- (8) the resolving lane combines its point's corner dots into the two gradients -------------------------------------
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (p_pt[u] >= 0) {
                    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (pos[u] >= 0) d = *reinterpret_cast<const float4 *>(S->ent + pos[u]);
                    if (!(p_valid[u] & 1u)) d.x = 0.f;
                    if (!(p_valid[u] & 2u)) d.y = 0.f;
                    if (!(p_valid[u] & 4u)) d.z = 0.f;
                    if (!(p_valid[u] & 8u)) d.w = 0.f;
                    const float lh = p_lh[u], lw = p_lw[u], hh = 1.f - lh, hw = 1.f - lw, a = p_a[u];
                    const float s_a = hh * hw * d.x + hh * lw * d.y + lh * hw * d.z + lh * lw * d.w;
                    const float s_w = hh * (d.y - d.x) + lh * (d.w - d.z);
                    const float s_h = hw * (d.z - d.x) + lw * (d.w - d.y);
                    grad_aw[p_pt[u]] = s_a;
                    *reinterpret_cast<float2 *>(grad_loc + 2u * (unsigned)p_pt[u]) = make_float2((float)W * s_w * a, (float)H * s_h * a);
                }
            // (the next chunk rewrites `ent` only after three more barriers; it rewrites gcache after one: the dots are done)
            PSB_STAMP(7)
        }

        // ---- fold the partial sums: pixel x of the pixel grid = BR[x] + BL[x+1] + TR[x+gw] + TL[x+gw+1] of the base grid ---------------
        __syncthreads();   // (the last chunk's combine still reads `ent`, which shares no memory with the stage; gcache is free now)
        {
            float *stage = S->gcache;   // [npx][32]
#pragma unroll
            for (int rnd = 0; rnd < 3; ++rnd) {
                const int k = rnd == 0 ? 2 : (rnd == 1 ? 1 : 0);            // BL, TR, TL
#pragma unroll
                for (int s = 0; s < kPsbSlots; ++s) {
                    const int p = s * kPsbGroups + grp;
                    if (p < npx) *reinterpret_cast<float4 *>(stage + p * kPsbD + 4 * j8) = acc[s][k];
                }
                __syncthreads();
#pragma unroll
                for (int s = 0; s < kPsbSlots; ++s) {
                    const int p = s * kPsbGroups + grp;
                    const int gr = p / gw, gc = p - gr * gw;
                    const int src = rnd == 0 ? p + 1 : (rnd == 1 ? p + gw : p + gw + 1);
                    const bool ok = p < npx && (rnd == 1 || gc + 1 < gw) && (rnd == 0 || gr + 1 <= R1 - R0);
                    if (ok) {
                        const float4 t = *reinterpret_cast<const float4 *>(stage + src * kPsbD + 4 * j8);
                        acc[s][3].x += t.x; acc[s][3].y += t.y; acc[s][3].z += t.z; acc[s][3].w += t.w;
                    }
                }
                __syncthreads();
            }
        }

        // ---- flush the tile: one 128-B row per pixel ----------------------------------------------------------------------------
        if (!g.lv[l].atomic) {
#pragma unroll
            for (int s = 0; s < kPsbSlots; ++s) {
                const int p = s * kPsbGroups + grp;
                const int row = R0 + p / gw, col = C0 + p % gw;
                if (p < npx && row < R1 && col < C1)
                    *reinterpret_cast<float4 *>(grad_value + lvl_base + (int64_t)(row * W + col) * row_elems) = acc[s][3];
            }
        } else {
            // several workgroups share the tile: hand the rows over through LDS and add them one channel per lane, so that a
            // wave instruction adds two whole 128-B rows (32-B atomic segments run ~4x slower)
            float *stage = S->gcache;
#pragma unroll
            for (int s = 0; s < kPsbSlots; ++s) {
                const int p = s * kPsbGroups + grp;
                if (p < npx) *reinterpret_cast<float4 *>(stage + p * kPsbD + 4 * j8) = acc[s][3];
            }
            __syncthreads();
            const int c32 = tid & 31;
            for (int p = tid >> 5; p < npx; p += kPsbThreads / 32) {
                const int row = R0 + p / gw, col = C0 + p % gw;
                const float x = stage[p * kPsbD + c32];
                if (row < R1 && col < C1 && x != 0.f)
                    atomicAdd(grad_value + (lvl_base - 4 * j8) + (int64_t)(row * W + col) * row_elems + c32, x);
            }
        }
        __syncthreads();
        PSB_STAMP(8)
    }
    if (g.stamps && tid == 0) {
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();
        S->stamp_acc[9] += now_ - S->stamp_last;   // waiting at the empty queue
        for (int i = 0; i < 15; ++i) g.stamps[(size_t)blockIdx.x * 16 + i] = S->stamp_acc[i];
    }
}

// Far points (a corner further than the level's margin from the pixel under its query): the direct method, one point per
// half-wave, one channel per lane: row atomics for grad_value, shuffle-reduced location / attention gradients.
__global__ __launch_bounds__(256) void psb_far_kernel(
    const float *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ aw,
    const float *__restrict__ grad_out, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_aw, const PsbGeom g)
{
    const unsigned n = g.ctr[8];
    const int c = threadIdx.x & 31;
    const int LP = g.L * g.P;
    const int row_elems = g.M * kPsbD;
    for (unsigned i = blockIdx.x * (blockDim.x / 32) + (threadIdx.x >> 5); i < n; i += gridDim.x * (blockDim.x / 32)) {
        const unsigned pt = g.far_list[i];
        const unsigned item = pt / (unsigned)LP;          // (b*Lq + q)*M + m
        const int lp = (int)(pt - item * (unsigned)LP), l = lp / g.P;
        const int m = (int)(item % (unsigned)g.M), b = (int)(item / (unsigned)g.M) / g.Lq;
        const float2 xy = *reinterpret_cast<const float2 *>(loc + 2u * pt);
        const float a = aw[pt];
        const int H = g.lv[l].H, W = g.lv[l].W;
        int o[4];
        float lh, lw;
        resolve_point<float>(xy.x, xy.y, H, W, (b * g.S + g.lv[l].start) * row_elems + m * kPsbD, row_elems, o, lh, lw);
        const float hh = 1.f - lh, hw = 1.f - lw;
        const float go = grad_out[item * (unsigned)kPsbD + c];
        const float w4[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
        float d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d[k] = 0.f;
            if (o[k] >= 0) {
                d[k] = go * value[o[k] + c];
                atomicAdd(grad_value + o[k] + c, w4[k] * a * go);
            }
        }
#pragma unroll
        for (int s = 1; s < 32; s <<= 1)
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] += __shfl_xor(d[k], s, kWave);
        if (c == 0) {
            grad_aw[pt] = w4[0] * d[0] + w4[1] * d[1] + w4[2] * d[2] + w4[3] * d[3];
            const float s_w = hh * (d[1] - d[0]) + lh * (d[3] - d[2]);
            const float s_h = hw * (d[2] - d[0]) + lw * (d[3] - d[1]);
            *reinterpret_cast<float2 *>(grad_loc + 2u * pt) = make_float2((float)W * s_w * a, (float)H * s_h * a);
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
struct PsbOptions {
    std::atomic<int> margin{6};        // fine levels, pixels of the sampled level
    std::atomic<int> tile{20};         // largest tile side + 1 (tile + apron <= 384 pixels)
    std::atomic<int> max_chunks{12};   // candidates of one workgroup, in chunks of 512 queries, before a tile is split into slabs
    std::atomic<int> coarse_px{640};   // a level with at most this many pixels takes every query as candidate (no margin)
};
inline PsbOptions &psb_options()
{
    static PsbOptions o;
    return o;
}

struct PsbPlan {
    bool ok = false;
    PsbGeom g{};
    size_t far_cap = 0;   // far-list capacity needed (points)
};

// candidate count of tile rows [a0, a1) (same for columns) at query level of size n: monotone map under(), margin mg
inline int psb_host_range(int n, float sc, int a, int z)
{
    int cnt = 0;
    for (int r = 0; r < n; ++r) {
        const int u = (int)(((float)r + 0.5f) * sc);
        cnt += (u >= a && u <= z) ? 1 : 0;
    }
    return cnt;
}

inline PsbPlan plan_psb(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes, const int64_t *lsi)
{
    PsbPlan pl;
    if (D != kPsbD || L < 1 || L > kPsbMaxL || P < 1 || P > kPsbMaxP) return pl;
    int64_t pre = 0;
    bool tiles_s = true;
    for (int l = 0; l < L; ++l) {
        tiles_s = tiles_s && lsi[l] == pre;
        pre += shapes[2 * l] * shapes[2 * l + 1];
        if (shapes[2 * l] >= 32768 || shapes[2 * l + 1] >= 32768) return pl;
    }
    if (!tiles_s || pre != S) return pl;   // tiles must not overlap in grad_value
    PsbGeom &g = pl.g;
    g.N = N; g.S = S; g.M = M; g.Lq = Lq; g.L = L; g.P = P;
    g.encoder = Lq == S ? 1 : 0;
    g.ppx = (N * M + kXcds - 1) / kXcds;
    const int tmax = psb_options().tile.load() - 1, mg_fine = psb_options().margin.load();
    const int max_chunks = psb_options().max_chunks.load(), coarse_px = psb_options().coarse_px.load();
    struct U { unsigned code; int64_t cost; };
    std::vector<U> units;
    for (int l = 0; l < L; ++l) {
        PsbLevel &v = g.lv[l];
        v.H = (int)shapes[2 * l]; v.W = (int)shapes[2 * l + 1]; v.start = (int)lsi[l];
        for (int t = tmax; t >= 1; --t) {   // balanced tiles of at most t x t pixels whose grid (+1 row / column) fits the lane groups
            v.nty = (v.H + t - 1) / t; v.ntx = (v.W + t - 1) / t;
            v.TH = (v.H + v.nty - 1) / v.nty; v.TW = (v.W + v.ntx - 1) / v.ntx;
            v.nty = (v.H + v.TH - 1) / v.TH; v.ntx = (v.W + v.TW - 1) / v.TW;
            if ((v.TH + 1) * (v.TW + 1) <= kPsbMaxPx) break;
        }
        if ((v.TH + 1) * (v.TW + 1) > kPsbMaxPx || v.nty > 64 || v.ntx > 64) return pl;
        v.mg = (g.encoder && v.H * v.W > coarse_px) ? mg_fine : -1;
        if (v.mg > 32767) v.mg = -1;
        for (int lq = 0; lq < kPsbMaxL; ++lq) {
            v.sH[lq] = lq < L ? (float)v.H / (float)shapes[2 * lq] : 1.f;
            v.sW[lq] = lq < L ? (float)v.W / (float)shapes[2 * lq + 1] : 1.f;
        }
    }
    int64_t far_cap = 0;
    for (int l = 0; l < L; ++l) {
        PsbLevel &v = g.lv[l];
        v.atomic = 0;
        if (v.mg >= 0) far_cap += (int64_t)N * Lq * M * P;
        for (int ty = 0; ty < v.nty; ++ty)
            for (int tx = 0; tx < v.ntx; ++tx) {
                int64_t ncand = Lq;
                if (g.encoder && v.mg >= 0) {
                    ncand = 0;
                    const int R0 = ty * v.TH, R1 = std::min(v.H, R0 + v.TH), C0 = tx * v.TW, C1 = std::min(v.W, C0 + v.TW);
                    for (int lq = 0; lq < L; ++lq)
                        ncand += (int64_t)psb_host_range((int)shapes[2 * lq], v.sH[lq], R0 - v.mg, R1 - 1 + v.mg) *
                                 psb_host_range((int)shapes[2 * lq + 1], v.sW[lq], C0 - v.mg, C1 - 1 + v.mg);
                }
                const int nchunks = (int)((ncand + kPsbQC - 1) / kPsbQC);
                int nslab = (nchunks + max_chunks - 1) / max_chunks;
                nslab = nslab < 1 ? 1 : nslab;
                if (nslab > 255) return pl;
                if (nslab > 1) v.atomic = 1;
                const int cps = (nchunks + nslab - 1) / nslab;
                for (int sl = 0; sl < nslab; ++sl) {
                    const int nch = std::max(0, std::min(nchunks, (sl + 1) * cps) - sl * cps);
                    units.push_back(U{(unsigned)l | (unsigned)ty << 2 | (unsigned)tx << 8 | (unsigned)sl << 14 | (unsigned)nslab << 22,
                                      (int64_t)nch * 16 + 8});
                }
            }
    }
    // a level is flushed either with stores by all its tiles or with atomics by all of them
    for (int l = 0; l < L; ++l)
        if (g.lv[l].atomic) { /* units of tiles with one slab still add atomically: the range is pre-zeroed */ }
    if (units.size() > (size_t)kPsbMaxUnits) return pl;
    std::stable_sort(units.begin(), units.end(), [](const U &a, const U &b) { return a.cost > b.cost; });
    g.nunits = (int)units.size();
    for (int i = 0; i < g.nunits; ++i) g.units[i] = units[i].code;
    pl.far_cap = (size_t)far_cap;
    pl.ok = true;
    return pl;
}

}  // namespace msda
