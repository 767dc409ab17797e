// msda_band.h -- "row-band" MSDeformAttn backward for gfx950: DECODER-shaped calls (few queries onto a large pyramid), D = 32, ONE launch.
//
// The reference's backward (ms_deform_im2col_cuda.cuh:87-159, 301-403) adds every bilinear corner of every sampling point to grad_value
// with a global float atomic.  A decoder call (1092 queries: 280 k sampling points onto 22 k pixels) had two kernels here through round 4:
// bwd_levelsum_kernel -- a workgroup per (image, head, level, 4-channel slice) walking ALL points of its level into an f64 LDS window,
// 48 us: every point is resolved by eight workgroups, and a pixel's 128-B row leaves in eight 16-B pieces -- and bwd_direct_kernel for
// grad_sampling_loc / grad_attn_weight (33 us, the forward's gathers once more).  Here ONE workgroup owns a band of ROWS of one
// (image, head, level) with all 32 channels:
//   zero      the band's window: 32 f64 sums per pixel in LDS (ds_add_f64, gfx950's fast LDS float atomic)
//   scan      every thread tests the four points of a query at this level: which of them touch the band (their upper or lower corner row
//             lies in it)?  Queries with a hit are listed in LDS (4 B each: query, hit mask, owner mask).  The thread of a DROPPED sample
//             (outside the reference's acceptance window, :276-285) writes its zero gradients on the spot (first band of the level only).
//   reduce    a group of EIGHT lanes (4 channels each) takes a listed query: grad_out's row (128 B) once, the four corners' value rows of
//             every point the band owns (the band that holds its upper row) -- two points, 8 rows, requested together --, w * a * grad_out added
//             to the window's sums of the corners inside the band, the four corner dots <grad_out, value> reduced over the group (DPP) and
//             turned into grad_sampling_loc / grad_attn_weight by one lane.
//   flush     every pixel of the band once: a 128-B row per lane group, plain stores -- no zero-fill of grad_value, no global atomics.
// Coarse levels receive ALL points of a call on a few rows; their bands are dealt over several workgroups by query range ("slabs"), which
// add their rows to the pre-zeroed level with 128-B row atomics (a few MB per call).
// grad_value is exact to fp32 rounding whatever the order (f64 sums); products are formed in fp32 as in the reference.
#pragma once

#include <algorithm>
#include <atomic>

#include "msda_common.h"

namespace msda {

constexpr int kBandThreads = 512;
constexpr int kBandD = 32;
constexpr int kBandStride = 36;            // doubles per pixel of the window: channel 4 * j + i sits in slot 8 * i + j, so that the eight lanes of a group
                                           // add to eight CONSECUTIVE doubles, and a pixel's row is 288 B so that the groups of a wave -- eight pixels --
                                           // start on different banks (a 256-B stride puts every pixel on the same banks: 8-way conflicts on every atomic)
constexpr int kBandMaxLevels = 8;
constexpr int kBandMaxEntries = 384;       // (level, band, slab) work items of one (image, head)
constexpr int kBandList = 2048;            // items (query, chunk of four points) one scan round tests and can list

struct BandOptions {
    std::atomic<int> lds_kb{64};           // window budget: 64 KB (+ list) = two workgroups per CU
    std::atomic<int> hits{400};            // expected hits per workgroup above which a band is dealt over slabs
};
inline BandOptions &band_options()
{
    static BandOptions o;
    return o;
}

struct BandGeom {
    int N, S, M, Lq, L, P;
    int nent;
    int dbg;          // diagnostic (wrong results): 1 skip the reduce stage, 2 skip the scan, 4 skip zeroing and flush, 8 no value rows / dots
    unsigned long long *stamps;      // diagnostic runs only (msda_debug_stamps): per workgroup 8 words -- start, end of zero / scan / reduce / flush, entry, items
    int H[kBandMaxLevels], W[kBandMaxLevels], start[kBandMaxLevels];
    // entry e: rows [r0, r0 + nr) of level lev; slab `slab` of `nslab` (the level's points dealt over nslab workgroups by point index)
    unsigned char lev[kBandMaxEntries], nr[kBandMaxEntries], slab[kBandMaxEntries], nslab[kBandMaxEntries];
    unsigned short r0[kBandMaxEntries];
};

struct BandPlan {
    bool ok = false;
    BandGeom g{};
    size_t lds_bytes = 0;
    unsigned atomic_levels = 0;     // levels whose bands are shared by several workgroups: pre-zeroed, accumulated with row atomics
};

inline BandPlan plan_band(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes, const int64_t *lsi)
{
    BandPlan pl;
    if (D != kBandD || L < 1 || L > kBandMaxLevels || P < 1 || (int64_t)Lq * P >= ((int64_t)1 << 24)) return pl;
    BandGeom &g = pl.g;
    g.N = N; g.S = S; g.M = M; g.Lq = Lq; g.L = L; g.P = P;
    const size_t win_budget = (size_t)std::max(16, std::min(150, band_options().lds_kb.load())) * 1024;
    const int px_cap = (int)(win_budget / (kBandStride * sizeof(double)));
    const int64_t npts = (int64_t)Lq * P;
    const int hits_cap = std::max(32, band_options().hits.load());
    int64_t pre = 0, scans = 0;
    int max_px = 0;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        if (lsi[l] != pre || H < 1 || W < 1 || H >= 32768 || W > px_cap) return pl;      // levels tile [0, S) in order; a row fits the window
        pre += (int64_t)H * W;
        g.H[l] = H; g.W[l] = W; g.start[l] = (int)lsi[l];
        const int rows_max = std::min(255, px_cap / W);
        const int bands = (H + rows_max - 1) / rows_max, rows = (H + bands - 1) / bands;
        // hits of a band if the points spread evenly over the level: its rows plus the row above (a point touches two rows)
        const double hits = (double)npts * (rows + 1) / (double)(H + 1);
        const int nslab = (int)std::min<int64_t>(255, std::max<int64_t>(1, (int64_t)((hits + hits_cap - 1) / hits_cap)));
        if (nslab > 1) pl.atomic_levels |= 1u << l;
        for (int r0 = 0; r0 < H; r0 += rows)
            for (int sl = 0; sl < nslab; ++sl) {
                if (g.nent >= kBandMaxEntries) return pl;
                const int e = g.nent++;
                g.lev[e] = (unsigned char)l;
                g.r0[e] = (unsigned short)r0;
                g.nr[e] = (unsigned char)std::min(rows, H - r0);
                g.slab[e] = (unsigned char)sl;
                g.nslab[e] = (unsigned char)nslab;
                scans += (npts + nslab - 1) / nslab;
            }
        max_px = std::max(max_px, rows * W);
    }
    if (pre != S) return pl;
    if (scans * N * M > ((int64_t)1 << 26)) return pl;      // every workgroup tests all points of its level / slab: a decoder-sized job only
    pl.lds_bytes = (size_t)max_px * kBandStride * sizeof(double) + (size_t)kBandList * sizeof(unsigned) + 64;
    pl.ok = true;
    return pl;
}

inline int band_grid(const BandGeom &g) { return kXcds * ((g.N * g.M + kXcds - 1) / kXcds) * g.nent; }

// sum over the 8 lanes of a lane group (DPP only: two quad permutes, one mirror inside the half-row)
__device__ __forceinline__ float band_group_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return v;
}

// TV: storage type of value / grad_out / grad_value (float, or bf16_t with fp32 arithmetic).  grad_acc: where the slabbed levels are
// accumulated with fp32 row atomics -- grad_value itself for TV = float, an fp32 image of it for bf16 (rounded by band_round_kernel).
// Work decomposition of scan and reduce: an ITEM is (query, chunk of four points of this level) -- RichSem's P = 4: one item per query --;
// a thread scans an item (its points' locations are 32 contiguous bytes), a group of EIGHT lanes (4 channels each) reduces a listed item:
// grad_out's row once, then the value rows of the points the band owns, two points (8 rows of 128 B) at a time -- four at a time cost 64
// registers: hipcc spilled, and a spill is a memory round trip in the middle of the loop (154 -> 103 us).
template <typename TV>
__global__ __launch_bounds__(kBandThreads, 4) void bwd_band_kernel(const TV *__restrict__ value, const float *__restrict__ loc,
                                                                   const float *__restrict__ aw, const TV *__restrict__ grad_out,
                                                                   TV *__restrict__ grad_value, float *__restrict__ grad_acc,
                                                                   float *__restrict__ grad_loc, float *__restrict__ grad_aw, const BandGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int pair, e;
    if (!decode_block(blockIdx.x, g.N * g.M, g.nent, pair, e)) return;
    const int b = pair / g.M, m = pair - b * g.M;
    const int l = g.lev[e], H = g.H[l], W = g.W[l], r0 = g.r0[e], nr = g.nr[e], nslab = g.nslab[e];
    const int npx = nr * W;
    double *win = reinterpret_cast<double *>(smem);
    unsigned *list = reinterpret_cast<unsigned *>(smem + (size_t)npx * kBandStride * sizeof(double));
    int *cnt = reinterpret_cast<int *>(list + kBandList);
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int j = tid & 7, grp = tid >> 3;
    constexpr int kGroups = kBandThreads / 8;
    const int c4 = 4 * j;      // this lane's four channels
    const int LP = g.L * g.P;
    const float Hf = (float)H, Wf = (float)W;
    const int nch = (g.P + 3) >> 2;      // chunks of four points per query

    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
    int st_items = 0;
    if (g.stamps) st[0] = __builtin_amdgcn_s_memtime();
    // this workgroup's share of the level's Lq * nch items
    const int n_items = g.Lq * nch;
    const int per = (n_items + nslab - 1) / nslab;
    const int i_lo = min(n_items, (int)g.slab[e] * per), i_hi = min(n_items, i_lo + per);
    const bool first_band = r0 == 0;
    const int64_t lvl_base = (int64_t)(b * g.S + g.start[l]);

    // the four points of an item: index of the first in sampling_loc / attn_weight, how many there are
    auto item_points = [&](int it, unsigned &pt0, int &np) {
        const int q = it / nch, ch = it - q * nch;
        pt0 = (unsigned)((b * g.Lq + q) * g.M + m) * (unsigned)LP + (unsigned)(l * g.P + 4 * ch);
        np = min(4, g.P - 4 * ch);
    };

    constexpr int kPer = kBandList / kBandThreads;      // items a thread scans per round
    for (int c0 = i_lo; c0 < i_hi; c0 += kBandList) {      // (uniform; a decoder call: one round)
        const int s_here = min(kBandList, i_hi - c0);
        // ---- scan: kPer items per thread; ALL their locations are requested first, and the window is zeroed while they travel ------------------
        float2 sxy[kPer][4];
        unsigned spt0[kPer];
        int snp[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int it = c0 + u * kBandThreads + tid;
            item_points(it < c0 + s_here ? it : c0, spt0[u], snp[u]);
#pragma unroll
            for (int p = 0; p < 4; ++p) sxy[u][p] = *reinterpret_cast<const float2 *>(loc + 2u * (spt0[u] + (unsigned)min(p, snp[u] - 1)));
        }
        if (c0 == i_lo && !(g.dbg & 4))
            for (int i = tid; i < npx * kBandStride; i += kBandThreads) win[i] = 0.0;
        if (tid == 0) *cnt = 0;
        __syncthreads();
        if (g.stamps && !st[1]) st[1] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int it = c0 + u * kBandThreads + tid;
            const bool in_range = it < c0 + s_here && !(g.dbg & 2);
            const unsigned pt0 = spt0[u];
            const int np = snp[u];
            unsigned hitm = 0u, ownm = 0u;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const bool there = in_range && p < np;
                const float h_im = sxy[u][p].y * Hf - 0.5f, w_im = sxy[u][p].x * Wf - 0.5f;
                const bool alive = there && h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
                if (there && !alive && first_band) {      // dropped sample: zero gradients (reference .cuh:285: nothing is added to them)
                    grad_aw[pt0 + p] = 0.f;
                    *reinterpret_cast<float2 *>(grad_loc + 2u * (pt0 + p)) = make_float2(0.f, 0.f);
                }
                const int h_low = (int)floorf(h_im);
                const bool top = h_low >= r0 && h_low < r0 + nr, bot = h_low + 1 >= r0 && h_low + 1 < r0 + nr;
                // the band that holds the point's upper row -- or, for a point whose upper row lies above the map, row 0 -- forms its gradients
                const int orow = min(max(h_low, 0), H - 1);
                if (alive && (top || bot)) {
                    hitm |= 1u << p;
                    if (orow >= r0 && orow < r0 + nr) ownm |= 1u << p;
                }
            }
            const unsigned long long mask = __ballot(hitm != 0u);
            if (mask) {      // (uniform) one LDS atomic per wave
                int base = 0;
                if (lane == 0) base = atomicAdd(cnt, __popcll(mask));
                base = __builtin_amdgcn_readfirstlane(base);
                if (hitm) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned)it | hitm << 24 | ownm << 28;
            }
        }
        __syncthreads();
        if (g.stamps) { st[2] = __builtin_amdgcn_s_memtime(); st_items += *cnt; }
        const int n = (g.dbg & 1) ? 0 : *cnt;
        // ---- reduce: a group of eight lanes per listed item; the NEXT item's first loads (its entry, grad_out row, locations, weights) travel
        //      under this item's work -------------------------------------------------------------------------------------------------------------
        struct Head {
            unsigned pt0, hitm, ownm;
            float4 go;
            float2 xy[4];
            float a[4];
        };
        auto fetch = [&](int i) {
            Head h;
            const unsigned w = i < n ? list[i] : list[0] & 0xFFFFFFu;      // (past the end: a valid item with empty masks)
            unsigned pt0;
            int np;
            item_points((int)(w & 0xFFFFFFu), pt0, np);
            h.pt0 = pt0;
            h.hitm = (w >> 24) & 15u;
            h.ownm = w >> 28;
            h.go = ld4(grad_out + (size_t)(pt0 / (unsigned)LP) * kBandD + c4);      // row (b * Lq + q) * M + m
#pragma unroll
            for (int p = 0; p < 4; ++p) {      // (points that are not listed read the item's first point: a valid address)
                const unsigned pt = pt0 + ((h.hitm >> p & 1u) ? (unsigned)p : 0u);
                h.xy[p] = *reinterpret_cast<const float2 *>(loc + 2u * pt);
                h.a[p] = aw[pt];
            }
            return h;
        };
        Head nxt;
        if (n > 0) nxt = fetch(grp);
        for (int i = grp; i < n; i += kGroups) {      // (uniform over the group)
            const Head cur = nxt;
            nxt = fetch(i + kGroups);
            const unsigned pt0 = cur.pt0, hitm = cur.hitm, ownm = cur.ownm;
            const float4 go = cur.go;
            const float2 *xy = cur.xy;
            const float *a = cur.a;
            // two points at a time (the value rows of four points in flight cost 64 registers: hipcc spilled, and a spill is a memory round
            // trip in the middle of the loop)
#pragma unroll
            for (int ph = 0; ph < 4; ph += 2) {
                if (!(hitm >> ph & 3u)) continue;      // (uniform over the group)
                // corners: rows / columns, fractions; value rows of the points this band owns (requested together)
                int row0[2], col0[2];
                float lh[2], lw[2];
                float4 v[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int p = ph + u;
                    const float h_im = xy[p].y * Hf - 0.5f, w_im = xy[p].x * Wf - 0.5f;
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    lh[u] = h_im - hf;
                    lw[u] = w_im - wf;
                    row0[u] = (int)hf;
                    col0[u] = (int)wf;
                    const bool own = (ownm >> p & 1u) && !(g.dbg & 8);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {      // (corners that do not count fetch the level's first row: a cache hit)
                        const int rr = row0[u] + (k >> 1), cc = col0[u] + (k & 1);
                        const bool in = own && rr >= 0 && rr <= H - 1 && cc >= 0 && cc <= W - 1;
                        const float4 x = ld4(value + ((lvl_base + (in ? rr * W + cc : 0)) * g.M + m) * kBandD + c4);
                        v[u][k] = in ? x : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int p = ph + u;
                    if (!(hitm >> p & 1u)) continue;      // (uniform over the group)
                    const float hh = 1.f - lh[u], hwt = 1.f - lw[u];
                    const float wgt[4] = {hh * hwt, hh * lw[u], lh[u] * hwt, lh[u] * lw[u]};
                    // top_grad * attn_weight (ms_deform_im2col_cuda.cuh:117), then * the corner's weight: the reference's order of products
                    const float4 ga = make_float4(go.x * a[p], go.y * a[p], go.z * a[p], go.w * a[p]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int rr = row0[u] + (k >> 1), cc = col0[u] + (k & 1);
                        if (rr >= r0 && rr < r0 + nr && rr <= H - 1 && cc >= 0 && cc <= W - 1 && !(g.dbg & 16)) {      // (rr >= r0 >= 0)
                            double *dst = win + (size_t)((rr - r0) * W + cc) * kBandStride + j;      // slot 8 * i + j <- channel 4 * j + i
                            atomicAdd(dst, (double)(wgt[k] * ga.x));
                            atomicAdd(dst + 8, (double)(wgt[k] * ga.y));
                            atomicAdd(dst + 16, (double)(wgt[k] * ga.z));
                            atomicAdd(dst + 24, (double)(wgt[k] * ga.w));
                        }
                    }
                    if ((ownm >> p & 1u) && !(g.dbg & 32)) {      // (uniform over the group)
                        float d[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            d[k] = band_group_sum(go.x * v[u][k].x + go.y * v[u][k].y + go.z * v[u][k].z + go.w * v[u][k].w);
                        if (j == 0) {
                            const unsigned pt = pt0 + (unsigned)p;
                            grad_aw[pt] = hh * hwt * d[0] + hh * lw[u] * d[1] + lh[u] * hwt * d[2] + lh[u] * lw[u] * d[3];
                            const float s_w = hh * (d[1] - d[0]) + lh[u] * (d[3] - d[2]), s_h = hwt * (d[2] - d[0]) + lw[u] * (d[3] - d[1]);
                            *reinterpret_cast<float2 *>(grad_loc + 2u * pt) = make_float2(Wf * s_w * a[p], Hf * s_h * a[p]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (i_lo >= i_hi) {      // (a slab past the level's last item: nothing to add, but the window is still flushed)
        if (!(g.dbg & 4)) for (int i = tid; i < npx * kBandStride; i += kBandThreads) win[i] = 0.0;
        __syncthreads();
    }
    if (g.stamps) st[3] = __builtin_amdgcn_s_memtime();

    // ---- flush: a pixel's 128-B row per lane group --------------------------------------------------------------------------------------------------
    if (g.dbg & 4) return;
    const int64_t base = (lvl_base + r0 * W) * g.M * kBandD + (int64_t)m * kBandD + c4;
    if (nslab == 1) {
        for (int px = grp; px < npx; px += kGroups) {
            const double *src = win + (size_t)px * kBandStride + j;
            st4(grad_value + base + (int64_t)px * g.M * kBandD, make_float4((float)src[0], (float)src[8], (float)src[16], (float)src[24]));
        }
    } else {
        // several workgroups share the band: its rows are ADDED to the (pre-zeroed) level, one channel per lane, so that a wave instruction
        // adds two whole 128-B rows (narrower atomic segments run several times slower)
        const int c1 = tid & 31;
        const int64_t base1 = (lvl_base + r0 * W) * g.M * kBandD + (int64_t)m * kBandD + c1;
        for (int px = tid >> 5; px < npx; px += kBandThreads / 32) {
            const float x = (float)win[(size_t)px * kBandStride + 8 * (c1 & 3) + (c1 >> 2)];
            if (x != 0.f) atomicAdd(grad_acc + base1 + (int64_t)px * g.M * kBandD, x);
        }
    }
    if (g.stamps && tid == 0 && blockIdx.x < 4096) {
        unsigned long long *o = g.stamps + (size_t)blockIdx.x * 16;
        o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = st[3]; o[4] = __builtin_amdgcn_s_memtime(); o[5] = (unsigned long long)e; o[6] = (unsigned long long)st_items;
    }
}

// The levels several workgroups add to, zeroed in every image (one launch; hipMemset2DAsync takes ~90 us per level on this runtime).
__global__ __launch_bounds__(256) void band_zero_kernel(float *__restrict__ acc, const BandGeom g, unsigned atomic_levels)
{
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    const int row4 = g.M * kBandD / 4;
    for (int l = 0; l < g.L; ++l) {
        if (!(atomic_levels >> l & 1u)) continue;
        const int n4 = g.H[l] * g.W[l] * row4;
        for (int b = 0; b < g.N; ++b) {
            float4 *dst = reinterpret_cast<float4 *>(acc) + (size_t)(b * g.S + g.start[l]) * row4;
            for (int i = gtid; i < n4; i += gsz) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// bf16 storage: the slabbed levels live in the fp32 image `acc`; round them into grad_value once.
__global__ __launch_bounds__(256) void band_round_kernel(const float *__restrict__ acc, bf16_t *__restrict__ grad_value, const BandGeom g,
                                                         unsigned atomic_levels)
{
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    const int row4 = g.M * kBandD / 4;
    for (int l = 0; l < g.L; ++l) {
        if (!(atomic_levels >> l & 1u)) continue;
        const int n4 = g.H[l] * g.W[l] * row4;
        for (int b = 0; b < g.N; ++b) {
            const size_t base = (size_t)(b * g.S + g.start[l]) * row4 * 4;
            for (int i = gtid; i < n4; i += gsz) st4(grad_value + base + 4 * (size_t)i, *reinterpret_cast<const float4 *>(acc + base + 4 * (size_t)i));
        }
    }
}

}  // namespace msda
