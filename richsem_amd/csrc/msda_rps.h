// msda_rps.h -- "routed pixel-stationary" MSDeformAttn backward for gfx950 (fp32 compute, D = 32, L <= 4).
//
// The reference's backward (ms_deform_im2col_cuda.cuh:87-159, 301-403) is query-stationary: every bilinear corner of every
// sampling point is ADDED to grad_value with a global float atomic -- 2.9 GB of atomics per encoder call, ~2.3 ms at the
// chip's atomic rate.  Here the OUTPUT owns the work: grad_value is cut into tiles of <= 15x15 pixels per (image, head,
// level) -- the tile plus one apron row / column is a grid of at most kRpsMaxPx = 256 pixels --, and a tile's workgroup pulls in
// exactly the sampling points that land on it.
//
//   route   (rps_route_kernel, ONE pass)  one lane per sampling point: where does its corner (h_low, w_low) fall?  The point is
//           appended -- a 16-byte record: position code, bilinear fractions, attention weight -- to the bin of THAT tile only
//           (round 5; rounds 2-4 sent a second / third / fourth record to the tiles below / right of it when its lower / right
//           corners cross the tile's edge: 13 % more records and three quarters of the pass's ranking work -- now the owner tile
//           keeps those corners' sums in the apron row / column of its window and ADDS them to the neighbours' first row /
//           column, which every tile flushes with row atomics onto rows the route pass has zeroed).  A workgroup (128
//           queries of one (image, head)) writes the records of each bin it touches as one RUN inside its own stretch of the
//           record pool and announces the run to the bin with one 64-bit atomic (records | runs): a bin is the list of its runs.
//           No counting pass, no prefix pass, no capacity guess, no overflow path.
//   reduce  (rps_tile_kernel)  a workgroup takes a tile: its value rows (+ a one-pixel apron) go to LDS; the bin is
//           walked in chunks of 1536 points: the points are sorted by the pixel under their corner (integer LDS
//           atomics: count, scan, place), every pixel's list is cut into units of at most 16 points, and a quad (four
//           lanes x eight channels) walks a unit:  four partial sums (one per corner) += w * grad_out[q]  and the four
//           "corner dots" <grad_out[q], value[corner]> that grad_sampling_loc / grad_attn_weight are linear in.
//           A unit's partial sums are added to the tile's f64 sums in LDS (ds_add_f64, the native LDS float atomic
//           of gfx950), so units -- not pixels -- are what the waves share out: a chunk's walk takes as long as its
//           points need, not as long as its longest list.
//
// grad_value is written with plain stores, once per pixel, inside the tiles -- no float atomics, no zero-fill there, fp32 sums as in the
// reference -- and value is read once.  Nothing depends on WHERE the points fall: uniform-random locations cost the same
// as local ones, so there is no margin, no "far" path and no locality monitor on this side.
// Coarse levels whose tile is the whole map split their bin into slabs over several workgroups; those add their rows to
// the (pre-zeroed) level with 128-B row atomics -- a few MB per call.  So do all tiles for their first row / column and their apron (the
// pixels they share with their neighbours: 13 % of the rows).
#pragma once

#include <algorithm>
#include <atomic>
#include <vector>

#include "msda_common.h"

namespace msda {

#ifndef RPS_THREADS
#define RPS_THREADS 768
#define RPS_RPL 2
#endif
constexpr int kRpsThreads = RPS_THREADS;               // 12 waves = 3 per SIMD, 168 registers per lane (1024 threads at 128 registers spilled; 512 threads: same speed)
constexpr int kRpsWaves = kRpsThreads / 64;
constexpr int kRpsMaxPx = 256;                  // pixel grid of a tile (tile + one row / column): bound by LDS (f64 sums + value rows)
constexpr int kRpsPpq = (kRpsMaxPx + kRpsThreads / 4 - 1) / (kRpsThreads / 4);   // pixels per quad where a quad stands for a pixel
constexpr int kRpsRpl = RPS_RPL;                      // records per lane and chunk
constexpr int kRpsChunk = kRpsRpl * kRpsThreads;   // sampling points per chunk
constexpr int kRpsSumStride = 36;               // doubles per pixel of the f64 sums (see RpsLds)
constexpr int kRpsSegShift = 3;                 // SMALLEST unit of the list walk: 1 << 3 = 8 points (the units' length is RpsOptions::seg_shift, default 4: <= 16 points)
constexpr int kRpsMaxSegs = kRpsMaxPx + (kRpsChunk >> kRpsSegShift);   // units of a chunk: <= lists + points / 8
constexpr int kRpsMaxL = 4;
constexpr int kRpsMaxUnits = 448;
constexpr int kRpsD = 32;
constexpr int kRpsPad = 32;                     // atomically updated counters sit on lines of their own
constexpr int kRpsDummyWgs = 2048, kRpsDummyBytes = kRpsDummyWgs * 1024;   // 1 KB per workgroup: 16 B per lane of a wave
constexpr int kRpsMaxRuns = 320;               // runs per bin the tile kernel can index (2.5 KB of LDS: what the budget leaves): Lq <= 320 x 128 queries
                                               // with 8-wave route workgroups (the 1280 x 1280 mosaic shape: 266), twice that with 16-wave ones
constexpr int kRpsRunSearch = 256;             // first step of the binary search over a bin's runs: the largest power of two below kRpsMaxRuns
static_assert(kRpsRunSearch < kRpsMaxRuns && 2 * kRpsRunSearch >= kRpsMaxRuns, "run search");
constexpr int kRpsQpBits = 19;                  // entry code: query * P + point below this bit (plan: Lq * P < 2^19)
static_assert(kRpsMaxPx <= 256, "record code: the base-grid index has 8 bits (bits 19..26), the tile kernel masks it with 0xFF");

struct RpsLevel {
    int H, W, start;
    int TH, TW, nty, ntx;   // tile grid: tile (ty, tx) = rows [ty*TH, min(H, ty*TH + TH)) x cols [tx*TW, ...)
    int bin0;               // first bin of this level inside a pair's bins; bin = bin0 + (ty*ntx + tx)*nslab + slab
    int nslab;              // bins ("slabs") per tile: the points of a dense tile are dealt over several bins by query block,
                            // each reduced by its own workgroup (and no single bin counter is hammered by every wave)
    int atomic;             // 1: nslab > 1 -> rows are added to pre-zeroed grad_value with atomics
    float inv_TH, inv_TW;   // tile of pixel row r: (int)((r + 0.5f) * inv_TH) -- exact for r < 2^15 (no integer division per point)
};

// Routed sampling point, as the route pass writes it and the tile kernel streams it.
//   code = (query * P + point)  |  base-grid index in the tile << 19 (8 bits: < kRpsMaxPx)  |  corners inside the map << 27 (4 bits)  |  1 << 31
// (bit 31: this tile forms the point's gradients -- every record since round 5, when a point stopped being sent to the neighbouring tiles
// as well).  The bin fixes (image, head, level).
struct alignas(16) RpsRec {
    unsigned code;
    float lh, lw, a;   // bilinear fractions, attention weight
};

struct RpsGeom {
    int N, S, M, Lq, L, P;
    int nunits, ppx;        // work table; pairs per XCD queue = ceil(N*M / 8)
    int bins_per_pair, nbins;
    RpsLevel lv[kRpsMaxL];
    unsigned units[kRpsMaxUnits];   // level | ty << 2 | tx << 8 | slab << 14 | nslab << 22, heaviest first
    unsigned *ctr;          // workspace: [0..7] per-XCD queue heads (reset by the route pass)
    unsigned long long *bin_state;   // [nbins], one per 128-B line (stride kRpsPad / 2): records of the bin (low word) | runs (high word); zero between calls
    uint2 *runs;            // [nbins * max_runs] a bin's runs in the record pool: {first record, position of the run inside the bin}
    int max_runs;           // runs a bin can get = query blocks of a pair (every route work item adds at most one run to a bin)
    int route_threads;      // threads of a route workgroup (512, or 1024 where 512 would give a bin more than kRpsMaxRuns runs)
    unsigned entries_cap;   // records the pool holds (one per point); indices are clamped to it, so that counters left
                            // dirty by an aborted call can give wrong results but never an access outside the pool
    struct RpsRec *entries;         // one 16-byte record per (point, bin it was routed to)
    int lut_r[kRpsMaxL], lut_c[kRpsMaxL], lut_n;      // route pass: where the levels' row / column tables start in LDS, their total length
    int seg_shift;                  // a pixel's list is walked in units of at most 1 << seg_shift points (>= kRpsSegShift)
    float *dummy;                   // kRpsDummyBytes of scratch: where the lanes that have nothing to store send their stores
                                    // (every store instruction is then issued unconditionally: see rps_tile_kernel)
    unsigned long long *stamps;     // diagnostic runs only (msda_debug_stamps)
    int dbg;                        // diagnostic: bits 4..6 = 1 + level -> only that level's tiles do any work (wrong results); bit 3: walk units in list order (A/B of the length classes);
                                    // bits 8..9, builds with -DRPS_ROUTE_ABLATION only: route-pass ablations (the tile kernel is not launched): 256 no record stores, 512 4-byte records
};

// One sampling point in the list of its base pixel; overwritten by its four corner dots.  The walk is bound by instruction issue (round 5:
// ~690 cycles per step of a wave with three waves per SIMD = 3 x ~57 instructions x 4 cycles), so the entry holds what costs a lane the
// fewest instructions: corner j4 = 2*r + c of a quad's lane weighs the point with (r ? lh : 1 - lh) * (c ? lw : 1 - lw) * a -- one FMA on
// lh (per-lane constants +-1, 0 / 1) times ONE word the lane picks by address (hwa or lwa) -- and the grad_out row is a 32-bit byte offset
// added to a scalar base (6 + 3 instructions per point before).
struct alignas(16) RpsEnt {
    float lh;                 // fraction along the rows
    float hwa, lwa;           // (1 - lw) * a, lw * a: the column fraction's two weights times the attention weight
    unsigned row;             // ((b*Lq + q)*M + m) * D * sizeof(TV): byte offset of the grad_out row -- read ahead of the rest
};

struct RpsLds {
    int item_slot[2];                   // work-queue draws (current / next), double-buffered
    int wave_tot[16];
    int n_segs, pad[3];
    int uhist[16];                      // walk units per (wave of the scan, length class), longest class first
    uint2 runs[kRpsMaxRuns];            // the work item's bin as the route pass left it: {first record, position inside the bin} per run
    unsigned long long stamp_last, stamp_acc[14];
    int offs[kRpsMaxPx + 4];            // histogram, then exclusive prefix: where a base pixel's list starts
    unsigned short seg[kRpsMaxSegs + 16];   // walk units of the chunk: list | segment of the list << 8
    RpsEnt ent[kRpsChunk];              // sorted points of the chunk, then their corner dots
    RpsRec meta[kRpsChunk];             // the routed records in ARRIVAL order (for the gradient combine)
    unsigned short slot[kRpsChunk];     // arrival index -> sorted slot (where the point's four corner dots are found)
    // grad_value of the tile's pixel grid: f64 sums (ds_add_f64), stored once per work item.  A pixel's row is 36 doubles and
    // channel 4*j + i of a half sits in slot 4*i + j: the four lanes of a quad then add to four consecutive doubles, and the
    // 16 quads of a wave -- consecutive pixels, usually -- spread over all 32 bank pairs (a 256-B row stride would put every
    // quad on the same banks: 16-way conflicts on every atomic)
    double sum[kRpsMaxPx * kRpsSumStride];
    float vtile[kRpsMaxPx * kRpsD];     // value rows of the pixel grid
};
static_assert(sizeof(RpsLds) <= 160 * 1024, "rps: LDS budget");

__device__ __forceinline__ float rps_group8_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return v;
}

__device__ __forceinline__ int rps_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Inclusive prefix sum over the 64 lanes of a wave in six DPP adds (shifts inside a row of 16, then the rows' last lanes broadcast to the
// rows behind them): the shuffle form goes through the LDS crossbar six times (ds_bpermute), this one stays in the vector pipe.
__device__ __forceinline__ int rps_wave_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);    // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);    // row_bcast:31 -> rows 2, 3
    return v;
}

#define RPS_STAMP(i)                                                                   \
    if (kStamps && g.stamps && threadIdx.x == 0) {                                     \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                  \
        S->stamp_acc[i] += now_ - S->stamp_last;                                       \
        S->stamp_last = now_;                                                          \
    }

// Where a sampling position falls, per level, comes from two small tables in LDS (round 4) -- one entry per base row r = h_low + 1 in
// [0, H] and one per base column c = w_low + 1 in [0, W] -- instead of ~35 vector instructions of tile arithmetic per point and level
// (the route pass is bound by instruction issue: ~2400 instructions per wave and work item before, profiles/r04_route_ablation.md):
//   row entry     gr (bits 0..4) | inr << 6 | (ty * ntx * nslab) << 8
//   column entry  gc (bits 0..4) | inc << 6 | (tx * nslab) << 8 | gw << 17
// gr / gc: base-grid row / column inside the owner tile (the tile that holds the point's upper-left corner; row / column 0 of the grid is
// the map's edge: a corner above / left of the map); inr / inc: bit 0 = upper / left corner inside the map, bit 1 = lower / right corner
// inside the map; gw: width of the owner tile's grid.  The tile grid is the one the host plans (RpsLevel) and the tile kernel's work items use.
__device__ __forceinline__ unsigned rps_lut_row(const RpsLevel &v, int r)
{
    const int h_low = r - 1, br = max(h_low, 0);
    const int ty = (int)(((float)br + 0.5f) * v.inv_TH);
    const int R0 = ty * v.TH;
    const unsigned gr = (unsigned)(h_low - R0 + 1);
    const unsigned inr = (h_low >= 0 ? 1u : 0u) | (h_low + 1 < v.H ? 2u : 0u);
    return gr | inr << 6 | (unsigned)(ty * v.ntx * v.nslab) << 8;
}
__device__ __forceinline__ unsigned rps_lut_col(const RpsLevel &v, int c)
{
    const int w_low = c - 1, bc = max(w_low, 0);
    const int tx = (int)(((float)bc + 0.5f) * v.inv_TW);
    const int C0 = tx * v.TW, C1 = min(v.W, C0 + v.TW);
    const unsigned gc = (unsigned)(w_low - C0 + 1);
    const unsigned inc = (w_low >= 0 ? 1u : 0u) | (w_low + 1 < v.W ? 2u : 0u);
    const unsigned gw = (unsigned)(C1 - C0 + 1);
    return gc | inc << 6 | (unsigned)(tx * v.nslab) << 8 | gw << 17;
}

// Route pass (ONE pass: no counting pass, no prefix pass).  A workgroup takes a block of consecutive queries of one (image, head);
// lane = (query, point), levels one after the other.  It first sorts out its own points in LDS -- a histogram over the pair's bins;
// the value an LDS atomic returns is the point's rank among the workgroup's points of that bin -- lays them out in ITS stretch of the
// record pool (its bins' records lie one after the other there: a "run" per bin), tells every bin it touched
// where its run is -- one 64-bit atomic per (workgroup, bin) adds the run's length to the bin's record count and one to its run count,
// and what it returns is the run's slot in the bin's run table and the run's position inside the bin -- and writes its 16-byte entries
// (position code, bilinear fractions, attention weight) at run start + rank.  The tile kernel reads a bin as the list of its runs.
// Also here: dropped samples get their (zero) gradients; what the tile kernel flushes with atomics later -- the slabbed levels, and
// the first row / column of every tile of the others -- is zeroed in grad_value; the tile kernel's queue heads are reset.
// Lanes of a wave usually share the owner bin (neighbouring queries, neighbouring points): they are matched with one ballot and
// served by a single LDS atomic; the others take one each.
// kRpsRouteThreads: 512 (8 waves: 128 queries per work item at P <= 4) is the form every shape up to Lq = kRpsMaxRuns x 128 = 40960 takes
// (E: 22323, the 1280 x 1280 mosaic batches: 34000); longer query sets run the same kernel with 1024 threads -- 256 queries per work item,
// so that a bin still gets at most kRpsMaxRuns runs (the tile kernel keeps a bin's run table in LDS) -- at ~12 % more time per query
// (one workgroup of 16 waves per CU instead of two of 8: MI355X, Lq = 34000: 94.3 against 83.8 us).
constexpr int kRpsRouteThreadsMax = 1024;
static_assert(kRpsMaxUnits <= 512, "route pass: one thread per bin of a pair");

// kL / kP: the call's levels / points per level as compile-time constants (0: read from the geometry) -- the pass is bound by instruction
// issue, and with L = P = 4 known the per-level branches, the zero-fills of skipped levels, the divisions by P and a good part of the
// scalar registers the generic form spills go away (round 5: 1150 -> ~700 instructions per wave and work item).  kStamps: the diagnostic
// stage stamps (msda_debug_stamps) are compiled into a second instance only.
template <int kRpsRouteThreads, int kL = 0, int kP = 0, bool kStamps = false>
#ifndef RPS_ROUTE_OCC
#define RPS_ROUTE_OCC 4      // waves per SIMD the 8-wave form is compiled for (two workgroups per CU)
#endif
__global__ __launch_bounds__(kRpsRouteThreads, kRpsRouteThreads == 512 ? RPS_ROUTE_OCC : 4) void rps_route_kernel(const float *__restrict__ loc, const float *__restrict__ aw,
                                                                     float *__restrict__ grad_value, float *__restrict__ grad_loc,
                                                                     float *__restrict__ grad_aw, const RpsGeom g)
{
    __shared__ unsigned hist[kRpsMaxUnits], base[512];
    __shared__ unsigned wtot[kRpsRouteThreads / kWave];
    __shared__ unsigned sink[kWave];           // where the lanes that have nothing to count send their (zero) LDS atomics: see stage A
    extern __shared__ unsigned rps_lut[];      // [g.lut_n]: row tables, then column tables, of the levels (offsets g.lut_r / g.lut_c)
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int B = g.bins_per_pair;
    const int L = kL ? kL : g.L;
    {
        const int gtid = blockIdx.x * blockDim.x + tid, gsz = gridDim.x * blockDim.x;
        if (gtid < 8) g.ctr[gtid] = 0u;
        if (tid < kWave) sink[tid] = 0u;
        const int row4 = g.M * kRpsD / 4;   // float4 per pixel
        for (int l = 0; l < L; ++l) {
            const RpsLevel &v = g.lv[l];
            for (int i = tid; i <= v.H; i += kRpsRouteThreads) rps_lut[g.lut_r[l] + i] = rps_lut_row(v, i);
            for (int i = tid; i <= v.W; i += kRpsRouteThreads) rps_lut[g.lut_c[l] + i] = rps_lut_col(v, i);
            if (!v.atomic) {
                // the first row and the first column of every tile take their neighbours' apron sums by row atomics (rps_tile_kernel's
                // flush): zero them in every image -- nty rows of W pixels, ntx columns of H pixels, all heads of a pixel at once
                const int nb = v.nty * v.W + v.ntx * v.H;
                for (int b = 0; b < g.N; ++b) {
                    float4 *dst = reinterpret_cast<float4 *>(grad_value) + (size_t)(b * g.S + v.start) * row4;
                    for (int i = gtid; i < nb * row4; i += gsz) {
                        const int k = i / row4, c = i - k * row4;
                        int px;
                        if (k < v.nty * v.W) {
                            const int ty = k / v.W;
                            px = ty * v.TH * v.W + (k - ty * v.W);
                        } else {
                            const int k2 = k - v.nty * v.W, tx = k2 / v.H;
                            px = (k2 - tx * v.H) * v.W + tx * v.TW;
                        }
                        dst[(size_t)px * row4 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                continue;
            }
            const int n4 = v.H * v.W * row4;
            for (int b = 0; b < g.N; ++b) {
                float4 *dst = reinterpret_cast<float4 *>(grad_value) + (size_t)(b * g.S + v.start) * row4;
                for (int i = gtid; i < n4; i += gsz) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    const int P = kP ? kP : g.P, LP = L * P;
    const int qpw = P <= 4 ? 16 : (P <= 8 ? 8 : (P <= 16 ? 4 : (P <= 32 ? 2 : 1)));   // queries per wave: qpw * P <= 64 lanes
    const int ql = lane / P, pp = lane - ql * P;
    const int qpb = qpw * (kRpsRouteThreads / kWave);   // queries per workgroup item
    const int qblocks = (g.Lq + qpb - 1) / qpb;
    const int pairs = g.N * g.M;
    const int n_items = pairs * qblocks;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // (diagnostic, msda_debug_stamps: shader cycles of thread 0 per stage, rows 1024 + blockIdx.x of the stamp buffer)
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = kStamps && g.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
#define RPS_RSTAMP(i)                                                  \
    if (kStamps && g.stamps) {                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        st_acc[i] += now_ - st_last;                                   \
        st_last = now_;                                                \
    }
    // The pass is a chain of round trips -- memory for the operands, LDS for the tables, the ballots' shuffles and the ranking atomics --
    // with a few hundred instructions between them (round 4, tools/rps_stamps.py: waiting for the operands was a third of a work item,
    // the per-level chains of stage A a quarter).  So the locations of the NEXT item are requested as soon as this item's are consumed
    // (their registers are free then), and every stage issues the LDS operations of all four levels before it waits for any of them:
    // lanes that have nothing to count add zero to a word of their own (`sink`) instead of branching around the atomic.
    float2 xy[kRpsMaxL];
    // (work item i = query block * pairs + pair; the items of a workgroup are gridDim.x apart: (pair, block) advance by a constant step,
    // without a division per item.  A queue instead -- items drawn one ahead with an atomic, tried in round 5 because the slowest workgroups
    // end 10-15 % after the median -- cost more than the tail: 47 -> 57 us, the wait for the returning draw sits in every item's first stage)
    const int step_pair = (int)(gridDim.x % (unsigned)pairs), step_qb = (int)(gridDim.x / (unsigned)pairs);
    auto item_point = [&](int item_, int pair_, int qb_, bool &live_, int &q_) {      // -> index of the lane's point at level 0
        const int b_ = pair_ / g.M, m_ = pair_ - b_ * g.M;
        q_ = qb_ * qpb + wave * qpw + ql;
        live_ = item_ < n_items && ql < qpw && pp < P && q_ < g.Lq;
        return (unsigned)(((b_ * g.Lq + (live_ ? q_ : 0)) * g.M + m_) * LP + min(pp, P - 1));      // (a valid index whatever the lane)
    };
    // (loads are unconditional, from clamped addresses: hipcc waits for a conditional load right behind it; `live` masks the result)
    auto request_xy = [&](int item_, int pair_, int qb_) {
        bool live_;
        int q_;
        const unsigned pt0_ = item_point(item_, pair_, qb_, live_, q_);
#pragma unroll
        for (int l = 0; l < kRpsMaxL; ++l) xy[l] = *reinterpret_cast<const float2 *>(loc + 2u * (pt0_ + (unsigned)(min(l, L - 1) * P)));
    };
    int pair = (int)(blockIdx.x % (unsigned)pairs), qb = (int)(blockIdx.x / (unsigned)pairs);
    request_xy(blockIdx.x, pair, qb);
    unsigned long long p_old = 0ull;      // the previous work item's announce (see stage D)
    unsigned p_first = 0u, p_cnt = 0u;
    size_t p_gb = 0;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {   // (uniform; neighbouring workgroups: different pairs = different bins)
        for (int i = tid; i < B; i += kRpsRouteThreads) hist[i] = 0u;
        bool live;
        int q;
        const unsigned pt0 = item_point(item, pair, qb, live, q);
        int n_pair = pair + step_pair, n_qb = qb + step_qb;      // the next item of this workgroup
        if (n_pair >= pairs) {
            n_pair -= pairs;
            ++n_qb;
        }
        n_qb = min(n_qb, qblocks - 1);      // (past the table: a valid address, the lanes are not live)
        __syncthreads();      // (the first item: the tables as well)
        RPS_RSTAMP(1)
        // ---- A: ranks inside the workgroup ---------------------------------------------------------------------------------------
        // word[l] = bin | rank << 9 | pbase << 23 of the point's entry in its owner tile; ~0u = no entry
        unsigned word[kRpsMaxL];
        unsigned inmap[kRpsMaxL];
        float lh[kRpsMaxL], lw[kRpsMaxL];
        unsigned re[kRpsMaxL], ce[kRpsMaxL];
        unsigned valid_m = 0u, asked_m = 0u, matched_m = 0u;      // per-lane flags, bit l (kept in vector registers: a bool per level and flag costs a pair of scalar registers, and the kernel spills those)
        unsigned dropped = 0u;      // levels at which the reference drops this point (their zero gradients are stored at the end of the item)
        // A.1: the sampling positions (reference ms_deform_im2col_cuda.cuh:276-285: dropped unless -1 < h_im < H and -1 < w_im < W) and
        //      their table entries
#pragma unroll
        for (int l = 0; l < kRpsMaxL; ++l) {
            lh[l] = lw[l] = 0.f;
            re[l] = ce[l] = 0u;
            if (l >= L) continue;   // (uniform)
            const RpsLevel &v = g.lv[l];
            const float Hf = (float)v.H, Wf = (float)v.W;
            const float h_im = xy[l].y * Hf - 0.5f, w_im = xy[l].x * Wf - 0.5f;
            const bool valid = live && h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
            valid_m |= valid ? 1u << l : 0u;
            const float hf = floorf(h_im), wf = floorf(w_im);
            lh[l] = h_im - hf;
            lw[l] = w_im - wf;
            dropped |= live && !valid ? 1u << l : 0u;
            re[l] = rps_lut[g.lut_r[l] + (valid ? (int)hf + 1 : 0)];
            ce[l] = rps_lut[g.lut_c[l] + (valid ? (int)wf + 1 : 0)];
        }
        // (xy is consumed: the weights of this item -- needed by the record stores of stage C only -- and the next item's locations
        // travel under the rest of this one; no other memory operation sits in a branch between here and stage C, so that the waits
        // for them are counted and never drain the queue)
        request_xy(item + (int)gridDim.x, n_pair, n_qb);
        float at[kRpsMaxL];
#pragma unroll
        for (int l = 0; l < kRpsMaxL; ++l) at[l] = aw[pt0 + (unsigned)(min(l, L - 1) * P)];
        // A.2: owner bins.  Lanes of a wave usually share the owner bin (neighbouring queries, neighbouring points): they are matched
        //      with one ballot and served by the atomic of one of them; the others take one each.
        int ob[kRpsMaxL], lead[kRpsMaxL], lb[kRpsMaxL];
        unsigned own_rank[kRpsMaxL], lt_cnt[kRpsMaxL], n_match[kRpsMaxL];
#pragma unroll
        for (int l = 0; l < kRpsMaxL; ++l) {
            ob[l] = -1;
            lead[l] = 0;
            lb[l] = -1;
            own_rank[l] = lt_cnt[l] = n_match[l] = 0u;
            inmap[l] = 0u;
            word[l] = ~0u;
            if (l >= L) continue;   // (uniform)
            const RpsLevel &v = g.lv[l];
            const int ns = v.nslab;
            const int slab = ns > 1 ? rps_uni(qb % ns) : 0;
            ob[l] = (valid_m >> l & 1u) ? v.bin0 + slab + (int)(re[l] >> 8) + (int)((ce[l] >> 8) & 511u) : -1;
            const unsigned inr = (re[l] >> 6) & 3u, inc = (ce[l] >> 6) & 3u;
            inmap[l] = ((inr & 1u) ? inc : 0u) | ((inr & 2u) ? inc << 2 : 0u);      // bit 0 TL, 1 TR, 2 BL, 3 BR inside the map
            const unsigned long long vote = __ballot(ob[l] >= 0);
            const int ld = vote ? __ffsll((long long)vote) - 1 : 0;      // (uniform)
            lb[l] = __builtin_amdgcn_readlane(ob[l], ld);               // (-1: no lane of the wave has a point at this level)
            const unsigned long long match = __ballot(ob[l] == lb[l]);
            lead[l] = ld;
            matched_m |= ob[l] == lb[l] && lb[l] >= 0 ? 1u << l : 0u;
            lt_cnt[l] = (unsigned)__popcll(match & lt_mask);
            n_match[l] = (unsigned)__popcll(match);
        }
#pragma unroll
        for (int l = 0; l < kRpsMaxL; ++l) {      // (the four atomics back to back: one wait)
            const bool is_lead = lane == lead[l] && lb[l] >= 0, other = ob[l] >= 0 && ob[l] != lb[l];
            const bool asked = is_lead || other;
            asked_m |= asked ? 1u << l : 0u;
            own_rank[l] = atomicAdd(asked ? &hist[is_lead ? lb[l] : ob[l]] : &sink[lane], is_lead ? n_match[l] : (other ? 1u : 0u));
        }
        // A.3: the record's word: bin | rank << 9 (added in stage C) | base-grid index << 23.  (Lower / right corners beyond the tile's edge
        //      stay with this tile: its window has an apron row / column for them, flushed with atomics -- no second record.)
#pragma unroll
        for (int l = 0; l < kRpsMaxL; ++l) {
            if (l >= L) continue;   // (uniform)
            const unsigned gr = re[l] & 31u, gc = ce[l] & 31u;
            if (!(asked_m >> l & 1u)) own_rank[l] = lt_cnt[l];      // + the leader's result, below
            if (ob[l] >= 0) word[l] = (unsigned)ob[l] | (gr * ((ce[l] >> 17) & 31u) + gc) << 23;
        }
        RPS_RSTAMP(2)
        __syncthreads();
        RPS_RSTAMP(3)
        // ---- B: the workgroup's runs: exclusive prefix over its bins' counts (one bin per thread), room in the pool, one 64-bit
        //      atomic per (workgroup, bin) ---------------------------------------------------------------------------------------------
        const unsigned cnt = tid < B ? hist[tid] : 0u;
        const unsigned incl = (unsigned)rps_wave_scan((int)cnt);
        if (lane == kWave - 1) wtot[wave] = incl;
        __syncthreads();
        unsigned before = 0u;
#pragma unroll
        for (int w = 0; w < kRpsRouteThreads / kWave; ++w) before += w < wave ? wtot[w] : 0u;
        // (every work item has its own stretch of the record pool, sized for its points -- one record each --: no cursor to share)
        const unsigned first = (unsigned)item * (unsigned)(qpb * LP) + before + incl - cnt;
        if (cnt) base[tid] = first;      // (known without asking anybody: the entries below do not wait for the bins' counters)
        // ---- C: entries to their slots ------------------------------------------------------------------------------------------
#pragma unroll
        for (int l = 0; l < kRpsMaxL; ++l) {
            if (l >= L) continue;
            unsigned r = own_rank[l];
            const unsigned rl = (unsigned)__builtin_amdgcn_readlane((int)r, lead[l]);
            if ((matched_m >> l & 1u) && lane != lead[l]) r += rl;
            if (word[l] != ~0u) word[l] |= r << 9;
        }
        __syncthreads();
        RPS_RSTAMP(4)
        // ---- D (first half): the runs are announced to their bins -- one 64-bit atomic per (workgroup, bin) -- BEFORE the records are
        //      stored: what the atomic returns (1-2 us later) is needed only for the run-table entry, written behind the record stores
        //      (unconditional -- threads without a run add zero to a scratch word of their own --: behind a branch the wait for the
        //      weights below could not be counted and would wait for the atomic as well)
        const size_t gb = (size_t)pair * B + min(tid, B - 1);
        unsigned long long *const scratch64 = reinterpret_cast<unsigned long long *>(g.dummy + (size_t)(blockIdx.x & (kRpsDummyWgs - 1)) * 256) + lane;
        // (the run-table entry of the PREVIOUS work item: its announce has had a whole item to come back -- round 5; written here, before this
        // item's atomic is issued, so that the two results never live at the same time and no register copy waits for the new one)
        if (p_cnt) g.runs[p_gb * (size_t)g.max_runs + (size_t)min((unsigned)(p_old >> 32), (unsigned)g.max_runs - 1u)] = make_uint2(p_first, (unsigned)p_old);
        const unsigned long long old = atomicAdd(cnt ? g.bin_state + gb * (kRpsPad / 2) : scratch64, cnt ? (1ull << 32) | (unsigned long long)cnt : 0ull);
        const unsigned qp = (unsigned)((live ? q : 0) * P + pp);
        unsigned slot0[kRpsMaxL];
#pragma unroll
        for (int l = 0; l < kRpsMaxL; ++l) slot0[l] = base[word[l] & 511u];      // (all the run starts are read before any of them is waited for; ~0u reads base[511])
        // (the weights are in their registers from here on: the conditional stores below then carry no wait of their own -- behind a
        // branch the compiler cannot count what is outstanding and would wait for EVERY earlier store to be acknowledged)
        asm volatile("" : "+v"(at[0]), "+v"(at[1]), "+v"(at[2]), "+v"(at[3]));
#pragma unroll
        for (int l = 0; l < kRpsMaxL; ++l) {
            const unsigned w = word[l];
            if (w != ~0u) {
                const unsigned slot = min(slot0[l] + ((w >> 9) & 16383u), g.entries_cap - 1u);
                const unsigned code = qp | (w >> 23) << kRpsQpBits | inmap[l] << 27 | 0x80000000u;
#ifdef RPS_ROUTE_ABLATION      // (diagnostic build, profiles/r04_route_ablation.md: what the record stores cost)
                if (g.dbg & 512) reinterpret_cast<unsigned *>(g.entries)[slot] = code;
                else if (!(g.dbg & 256))
#endif
                g.entries[slot] = RpsRec{code, lh[l], lw[l], at[l]};
            }
        }
        if (__ballot(dropped != 0u)) {      // (uniform; rare) dropped samples: zero gradients
#pragma unroll
            for (int l = 0; l < kRpsMaxL; ++l)
                if (dropped >> l & 1u) {
                    const unsigned pt = pt0 + (unsigned)(l * P);
                    grad_aw[pt] = 0.f;
                    *reinterpret_cast<float2 *>(grad_loc + 2u * pt) = make_float2(0.f, 0.f);
                }
        }
        RPS_RSTAMP(5)
        // ---- D (second half): the run's slot in the bin's run table and its position inside the bin -- written one work item later (above)
        p_old = old;
        p_first = first;
        p_cnt = cnt;
        p_gb = gb;
        pair = n_pair;
        qb = n_qb;
        RPS_RSTAMP(6)
        __syncthreads();   // hist / base are reused by the next item
        RPS_RSTAMP(7)
    }
    if (p_cnt) g.runs[p_gb * (size_t)g.max_runs + (size_t)min((unsigned)(p_old >> 32), (unsigned)g.max_runs - 1u)] = make_uint2(p_first, (unsigned)p_old);
    if (kStamps && g.stamps && tid == 0)
        for (int i = 0; i < 8; ++i) g.stamps[(size_t)(1024 + blockIdx.x) * 16 + i] = st_acc[i];
#undef RPS_RSTAMP
}

typedef float rps_v2f __attribute__((ext_vector_type(2)));
typedef float rps_v4f __attribute__((ext_vector_type(4)));

// Lane j of a quad ends up with the quad's sum over its four lanes of "corner j's value" (j = 0..3), where every lane passes ITS values
// in the order x[r] = value of corner r ^ j: two exchange steps -- with lane ^ 1, then lane ^ 2 -- in which a lane keeps the half it is
// responsible for and takes the matching half of its partner: with the rotated order those are the same REGISTERS in every lane, so the
// steps are three DPP adds and no selects (rounds 3-4 kept corner r in register r: six selects per point on top, a fifth of the walk's
// vector instructions once the rest had shrunk).  The caller arranges the rotation where it is free: which value row a register holds.
__device__ __forceinline__ float rps_quad_rotated_sum(float x0, float x1, float x2, float x3)
{
    const float ra = x0 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x1), 0xB1, 0xF, 0xF, true));   // lane ^ 1: its x1 is corner j
    const float rb = x2 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x3), 0xB1, 0xF, 0xF, true));   //           its x3 is corner j ^ 2
    return ra + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(rb), 0x4E, 0xF, 0xF, true));             // lane ^ 2: its rb is corner j
}

// A lane's 8 channels of a grad_out row as they travel from memory: fp32 as two float4, bf16 as two packed 8-byte words that are
// widened only where the point is reduced (half the registers in flight).
template <typename TV>
struct RpsRow;
template <>
struct RpsRow<float> {
    float4 a, b;
    // (the lane's two 16-B pieces of the row at byte offset `off` -- lane offset included -- from the uniform base: one 32-bit add per point)
    __device__ __forceinline__ void load(const float *base, unsigned off)
    {
        a = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + off);
        b = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + off + 64u);
    }
    __device__ __forceinline__ void unpack(rps_v2f (&gq)[4]) const
    {
        gq[0] = (rps_v2f){a.x, a.y}; gq[1] = (rps_v2f){a.z, a.w};
        gq[2] = (rps_v2f){b.x, b.y}; gq[3] = (rps_v2f){b.z, b.w};
    }
};
template <>
struct RpsRow<bf16_t> {
    uint2 a, b;
    __device__ __forceinline__ void load(const bf16_t *base, unsigned off)
    {
        a = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(base) + off);
        b = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(base) + off + 32u);
    }
    __device__ __forceinline__ void unpack(rps_v2f (&gq)[4]) const
    {
        gq[0] = (rps_v2f){__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xFFFF0000u)};
        gq[1] = (rps_v2f){__uint_as_float(a.y << 16), __uint_as_float(a.y & 0xFFFF0000u)};
        gq[2] = (rps_v2f){__uint_as_float(b.x << 16), __uint_as_float(b.x & 0xFFFF0000u)};
        gq[3] = (rps_v2f){__uint_as_float(b.y << 16), __uint_as_float(b.y & 0xFFFF0000u)};
    }
};

// Work decomposition: a QUAD (4 lanes x 8 channels) walks one unit -- at most 16 points of one base pixel's list -- with four
// partial sums (32 registers per lane) and the value rows of the pixel's four corners in registers, packed fp32 arithmetic
// (v_pk_fma_f32) throughout: every FMA carries two channels and the per-point overhead (weights, the dot reduction, the
// entry read) is shared by 8 channels per lane.  After its unit the quad adds the partial sums to the tile's f64 sums in LDS
// and takes the next unit (the waves take groups of 16 units in turn).  Measured before this structure (MI355X, call E,
// init pattern): with ONE quad per pixel and the sums kept in registers for the whole tile, a chunk's walk lasted as long as
// its longest list -- 2.3 to 3 times the mean list, and a quarter of the quads had no list at all -- so the walk ran at
// 23 % of its lanes' capacity.
// Everything a work item needs from memory is requested early: the queue is drawn two items ahead, the bin bounds of the
// next item are requested at the start of the current one, its first chunk of records while the last chunk's gradients are
// formed, the value rows at the start of the item (they are needed only by the walk, two stages later).
// Stores are issued by every lane unconditionally (lanes with nothing to store write to the workgroup's scratch line): the
// compiler can then count them, and a wait for an older load is s_waitcnt vmcnt(n), not vmcnt(0) -- which would hold the
// wave until every scattered store has retired (vector-memory operations retire in order).
// TV: storage type of value / grad_out / grad_value (float, or bf16_t with fp32 arithmetic).  grad_acc: where the levels whose
// tiles are shared by several workgroups are accumulated with fp32 atomics -- grad_value itself for TV = float, an fp32 scratch
// image of it for bf16 (rounded by rps_round_kernel afterwards).
// kStamps: the diagnostic stage stamps (msda_debug_stamps) are compiled into a second instance only
template <bool P4, typename TV = float, bool kStamps = false>
__global__ __launch_bounds__(kRpsThreads, kRpsThreads / 256) void rps_tile_kernel(
    const TV *__restrict__ value, const TV *__restrict__ grad_out, TV *__restrict__ grad_value, float *__restrict__ grad_acc,
    float *__restrict__ grad_loc, float *__restrict__ grad_aw, const RpsGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    RpsLds *S = reinterpret_cast<RpsLds *>(smem);

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int j4 = tid & 3, quad = tid >> 2;
    // this lane's 8 channels: [c_lo, c_lo + 4) and [c_hi, c_hi + 4) -- so that each of a lane's two 16-B accesses to a 128-B
    // row forms, with the other three lanes of its quad, 64 contiguous bytes
    const int c_lo = 4 * j4, c_hi = 16 + 4 * j4;
    // the walk's per-lane constants (see RpsEnt): corner j4 = 2 * r + c of the quad's lane
    const float w_ys = (j4 & 2) ? 1.f : -1.f, w_yc = (j4 & 2) ? 0.f : 1.f;      // (r ? lh : 1 - lh) = w_ys * lh + w_yc
    const unsigned w_xoff = (j4 & 1) ? 8u : 4u;                                  // lwa : hwa
    const unsigned lane_row = (unsigned)(c_lo * sizeof(TV));
    const int P = P4 ? 4 : g.P;
    const int LP = g.L * P;
    const int row_elems = g.M * kRpsD;
    const int pairs = g.N * g.M;
    const int xq = blockIdx.x & (kXcds - 1);   // blocks equal mod 8 share an XCD (observed; speed only)
    const int n_items = g.nunits * g.ppx;
    float *const dummy_w = g.dummy + (size_t)(blockIdx.x & (kRpsDummyWgs - 1)) * 256;   // this workgroup's 1 KB of scratch
    if (kStamps && g.stamps && tid == 0) {
        for (int i = 0; i < 14; ++i) S->stamp_acc[i] = 0;
        S->stamp_last = __builtin_amdgcn_s_memtime();
    }
    // a work item (uniform): tile and pair
    struct Item {
        bool live;
        int l, b, m, H, W, R0, R1, C0, C1, gw, npx;
    };
    auto item_geom = [&](int id) {
        Item it;
        const unsigned unit = (unsigned)rps_uni((int)g.units[min(id, n_items - 1) / g.ppx]);
        const int pr = xq + kXcds * (min(id, n_items - 1) % g.ppx);
        const int pair = min(pr, pairs - 1);
        // (diagnostic: dbg bits 4..6 = 1 + level -> only that level's units do any work)
        it.live = id < n_items && pr < pairs && (((g.dbg >> 4) & 7) == 0 || ((g.dbg >> 4) & 7) == (int)(unit & 3) + 1);
        it.l = unit & 3;
        const int ty = (unit >> 2) & 63, tx = (unit >> 8) & 63;
        it.b = pair / g.M;
        it.m = pair - it.b * g.M;
        it.H = g.lv[it.l].H;
        it.W = g.lv[it.l].W;
        it.R0 = ty * g.lv[it.l].TH;
        it.R1 = min(it.H, it.R0 + g.lv[it.l].TH);
        it.C0 = tx * g.lv[it.l].TW;
        it.C1 = min(it.W, it.C0 + g.lv[it.l].TW);
        // Two grids of the same shape (th+1) x gw:
        //   base grid   (gr, gc) <-> sampling points whose corner (h_low, w_low) is pixel (R0-1+gr, C0-1+gc): one list each
        //   pixel grid  (vr, vc) <-> pixel (R0+vr, C0+vc): the tile plus one apron row / column (value rows for the dots)
        // so the four corners of base p are the pixels p-gw-1, p-gw, p-1, p of the pixel grid.
        it.gw = it.C1 - it.C0 + 1;
        it.npx = it.live ? (it.R1 - it.R0 + 1) * it.gw : 0;
        return it;
    };
    // bin of a work item: its index, number of records and number of runs (0 records for ids past the table or pairs past the batch)
    // (bin_range_raw requests the counter; bin_counts turns it into numbers WHERE THEY ARE FIRST NEEDED -- behind an opaque asm, or the
    // compiler forms them, and waits for the load, at the top of the work item)
    auto bin_range_raw = [&](int id, unsigned &bin_, unsigned &okm_, unsigned long long &st_) {
        const int idc = min(id, n_items - 1);
        const unsigned unit = g.units[idc / g.ppx];
        const int pair = xq + kXcds * (idc % g.ppx);
        const bool ok = id < n_items && pair < pairs;
        const int l = unit & 3, ty = (unit >> 2) & 63, tx = (unit >> 8) & 63, slab = (unit >> 14) & 255, nslab = (unit >> 22) & 255;
        const unsigned bin_c = ok ? (unsigned)(pair * g.bins_per_pair + g.lv[l].bin0 + (ty * g.lv[l].ntx + tx) * nslab + slab) : 0u;
        st_ = g.bin_state[(size_t)bin_c * (kRpsPad / 2)];
        okm_ = ok ? 0xFFFFFFFFu : 0u;
        bin_ = bin_c | ~okm_;                             // (0xFFFFFFFF: no bin)
    };
    auto bin_counts = [&](unsigned long long st_, unsigned okm_, int &n, int &nr) {
        unsigned lo = (unsigned)st_, hi = (unsigned)(st_ >> 32);
        asm volatile("" : "+v"(lo), "+v"(hi));
        n = (int)(min(lo, 0x7FFFFFFFu) & okm_);
        nr = (int)(min(hi, (unsigned)min(g.max_runs, kRpsMaxRuns)) & okm_);
    };
    auto bin_range = [&](int id, unsigned &bin_, int &n, int &nr) {
        // (the counter is read unconditionally -- bin 0's for an id past the table or a pair past the batch, masked afterwards --: hipcc
        // waits for a load that sits in a branch right behind it, and this one is needed a whole work item later)
        const int idc = min(id, n_items - 1);
        const unsigned unit = g.units[idc / g.ppx];
        const int pair = xq + kXcds * (idc % g.ppx);
        const bool ok = id < n_items && pair < pairs;
        const int l = unit & 3, ty = (unit >> 2) & 63, tx = (unit >> 8) & 63, slab = (unit >> 14) & 255, nslab = (unit >> 22) & 255;
        const unsigned bin_c = ok ? (unsigned)(pair * g.bins_per_pair + g.lv[l].bin0 + (ty * g.lv[l].ntx + tx) * nslab + slab) : 0u;
        const unsigned long long st = g.bin_state[(size_t)bin_c * (kRpsPad / 2)];
        const unsigned okm = ok ? 0xFFFFFFFFu : 0u;      // (masks, not selects: a select lets the compiler move the load under `ok`)
        bin_ = bin_c | ~okm;                             // (0xFFFFFFFF: no bin)
        n = (int)(min((unsigned)st, 0x7FFFFFFFu) & okm);
        nr = (int)(min((unsigned)(st >> 32), (unsigned)min(g.max_runs, kRpsMaxRuns)) & okm);
    };
    // a bin's run table: global -> registers (one run per lane) -> LDS
    uint2 run_reg = make_uint2(0u, 0u);
    auto load_runs = [&](unsigned bin_, int nr) {
        const int i = min(tid, max(rps_uni(nr) - 1, 0));
        const unsigned b_ = (unsigned)rps_uni((int)bin_);
        run_reg = g.runs[(size_t)(b_ == 0xFFFFFFFFu ? 0u : b_) * g.max_runs + i];
    };
    auto park_runs = [&]() {
        if (tid < kRpsMaxRuns) S->runs[tid] = run_reg;
    };
    RpsRec n_rec[kRpsRpl];   // this lane's records of the chunk in flight (lanes past the end of the bin: a copy of its last record)
    // The run of record k = the last run that starts at or before it (the runs' positions ascend with their slots).  A binary search over
    // the table is nine DEPENDENT LDS reads per record -- 2.2 k cycles per call with every wave of the workgroup in the same chain, ~20 calls
    // per workgroup: 9 % of the kernel (round 5, stage stamps).  A wave's 64 records are consecutive, so the wave searches together: every
    // lane reads ONE sample of the table (every T-th run), two ballots bracket the wave's first and last record between samples, and a lane
    // finishes inside that bracket -- a few runs -- with a short search of its own: 3-5 dependent reads instead of 10.
    auto fetch_recs = [&](int n_, int nr_, int ch) {      // (the bin's run table is in LDS)
        const int n = rps_uni(n_), nr = rps_uni(nr_);
        const int T = max((nr + kWave - 1) / kWave, 1);      // sample stride (nr <= kRpsMaxRuns: T <= 5)
        const bool s_ok = lane * T < nr;
        const unsigned sy = S->runs[min(lane * T, max(nr - 1, 0))].y;
        const int k_last = max(n - 1, 0);
#pragma unroll
        for (int u = 0; u < kRpsRpl; ++u) {
            const int kb = ch * kRpsChunk + u * kRpsThreads + wave * kWave;      // (uniform) the wave's first record of this slot
            const unsigned k = (unsigned)min(kb + lane, k_last);
            const unsigned k_lo = (unsigned)min(kb, k_last), k_hi = (unsigned)min(kb + kWave - 1, k_last);
            const int c_lo = __popcll(__ballot(s_ok && sy <= k_lo)) - 1, c_hi = __popcll(__ballot(s_ok && sy <= k_hi)) - 1;
            const int ra = max(c_lo, 0) * T, rb = min((max(c_hi, 0) + 1) * T, nr);      // (uniform) the runs of all 64 records lie in [ra, rb)
            int r = ra;
            for (int step = rb - ra > 1 ? 1 << (31 - __builtin_clz(rb - ra - 1)) : 0; step > 0; step >>= 1) {      // (uniform trip count)
                const int cand = r + step;
                if (cand < rb && S->runs[cand].y <= k) r = cand;
            }
            const uint2 run = S->runs[r];
            n_rec[u] = g.entries[n > 0 ? min(run.x + (k - run.y), g.entries_cap - 1u) : 0u];
        }
    };
    // value rows of a work item's pixel grid: 32 B per lane and pixel, pixels `quad` and `quad + 192` (the second only where
    // the grid has that many).  Pixels beyond the map fetch a clamped (valid) row: corners outside the map are masked where the
    // gradients are formed.
    float4 nv[kRpsPpq][2];
    auto fetch_rows = [&](const Item &it) {
#pragma unroll
        for (int r = 0; r < kRpsPpq; ++r) {
            const int px = min(quad + r * (kRpsThreads / 4), max(it.npx - 1, 0));
            const int gr_ = px / it.gw, gc_ = px - gr_ * it.gw;
            const int prow_ = min(it.R0 + gr_, it.H - 1), pcol_ = min(it.C0 + gc_, it.W - 1);
            const TV *src = value + ((int64_t)(it.b * g.S + g.lv[it.l].start + prow_ * it.W + pcol_) * g.M + it.m) * kRpsD;
            nv[r][0] = ld4(src + c_lo);
            nv[r][1] = ld4(src + c_hi);
        }
    };
    auto store_rows = [&]() {
#pragma unroll
        for (int r = 0; r < kRpsPpq; ++r) {
            const int px = quad + r * (kRpsThreads / 4);
            if (px < kRpsMaxPx) {
                *reinterpret_cast<float4 *>(S->vtile + px * kRpsD + c_lo) = nv[r][0];
                *reinterpret_cast<float4 *>(S->vtile + px * kRpsD + c_hi) = nv[r][1];
            }
        }
    };

    // The work queue is drawn by EVERY thread (round 4): thread 0 adds one to its XCD's queue head, the others add zero to a scratch word.
    // Inside `if (tid == 0)` the compiler waits for the returning atomic right behind it -- 1-2 us per work item during which wave 0 kept
    // the whole workgroup at the next barrier, with the value rows and first records of the item it had requested early drained as well.
    unsigned *const draw_at = tid == 0 ? g.ctr + xq : reinterpret_cast<unsigned *>(dummy_w) + (tid & 255);
    const unsigned draw_by = tid == 0 ? 1u : 0u;
    unsigned draw = atomicAdd(draw_at, draw_by);   // (thread 0) the queue draw in flight
    if (tid == 0) S->item_slot[0] = (int)draw;
    draw = atomicAdd(draw_at, draw_by);
    __syncthreads();
    int item_id = rps_uni(S->item_slot[0]);
    unsigned e_bin;
    int n_ent, n_runs;
    bin_range(item_id, e_bin, n_ent, n_runs);
    e_bin = (unsigned)rps_uni((int)e_bin);
    n_ent = rps_uni(n_ent);
    n_runs = rps_uni(n_runs);
    load_runs(e_bin, n_runs);
    park_runs();
    __syncthreads();
    fetch_recs(n_ent, n_runs, 0);
    Item it = item_geom(item_id);
    int par = 0;
    fetch_rows(it);
    for (int i = tid; i < kRpsMaxPx * kRpsSumStride / 2; i += kRpsThreads) reinterpret_cast<double2 *>(S->sum)[i] = make_double2(0.0, 0.0);

    while (item_id < n_items) {
        if (tid == 0) S->item_slot[par ^ 1] = (int)draw;   // issued one item ago
        const int l = it.l, b = it.b, m = it.m, H = it.H, W = it.W, R0 = it.R0, R1 = it.R1, C0 = it.C0, C1 = it.C1, gw = it.gw;
        const int npx = it.npx;
        const int n_chunks = it.live ? (n_ent + kRpsChunk - 1) / kRpsChunk : 0;
        const int bq0 = b * g.Lq;
        float *const vt = S->vtile;
        // (the item's value rows were requested before the previous item's sums were stored -- before the loop for the first item;
        // the f64 sums are zero: cleared before the loop, and by the lanes that read them out at the end of every item)
        __syncthreads();
        const int next_id = rps_uni(S->item_slot[par ^ 1]);
        unsigned next_bin, next_ok;
        unsigned long long next_st;
        int next_n = 0, next_runs = 0;
        bin_range_raw(next_id, next_bin, next_ok, next_st);   // (turned into counts where first used: that waits for the load)
        const Item nit = item_geom(next_id);
        store_rows();   // (read only behind the barriers of the first chunk's sort; an empty bin reads nothing)
        RPS_STAMP(0)

        for (int ch = 0; ch < n_chunks; ++ch) {
            const int n_here = min(kRpsChunk, n_ent - ch * kRpsChunk);
            // ---- (1) this lane's two points of the chunk: rank in the list of their base pixel -----------------------------------
            for (int i = tid; i <= npx; i += kRpsThreads) S->offs[i] = 0;
            int pbase[kRpsRpl], pos[kRpsRpl];
#pragma unroll
            for (int u = 0; u < kRpsRpl; ++u) {
                pos[u] = -1;
                // the record stays in the order the route pass wrote it: neighbouring lanes hold neighbouring points of a query, and the
                // gradient stores of stage (5) -- taken in this order -- fall into 16 / 32 contiguous bytes per (query, level)
                S->meta[u * kRpsThreads + tid] = n_rec[u];
            }
            // (the next queue draw goes out here, in the item's first chunk: returning atomics and loads come back in order, and the three
            // sort stages that follow touch LDS only -- at the top of the item it delayed the value rows, behind the walk the next records)
            if (ch == 0) draw = atomicAdd(draw_at, draw_by);
            __syncthreads();
            RPS_STAMP(1)
#pragma unroll
            for (int u = 0; u < kRpsRpl; ++u) {
                pbase[u] = (int)((n_rec[u].code >> kRpsQpBits) & 0xFFu);
                if (u * kRpsThreads + tid < n_here) pos[u] = atomicAdd(&S->offs[pbase[u]], 1);
            }
            __syncthreads();
            RPS_STAMP(2)

            // ---- (2) exclusive scan of the per-list counts (<= 256: one per thread of the first 4 waves) and of the lists' unit
            //      counts, both in one packed word (count | units << 16); every list writes its units -------------------------------
            //      The units are listed by LENGTH CLASS, longest first: a wave walks 16 units side by side for as long as the longest
            //      of them, so units of like length (four classes) go together (lists vary from 1 to 16+ points around a mean of ~6).
            {
                const int sh = g.seg_shift;      // a unit's length -> quarter of the full length it falls into -> slot (0 = longest)
                int c = 0, v = 0, incl = 0, n_full = 0, part = 0, pslot = 0, n0 = 0, incl0 = 0, rank = 0, mycnt = 0;
                if (tid < kRpsMaxPx) {
                    c = tid < npx ? S->offs[tid] : 0;
                    v = c | ((c + (1 << sh) - 1) >> sh) << 16;
                    n_full = c >> sh;
                    part = c & ((1 << sh) - 1);
                    pslot = part ? ((g.dbg & 8) ? 0 : 3 - ((part - 1) >> (sh - 2))) : -1;   // (dbg 8: A/B, units in list order)
                    n0 = n_full + (pslot == 0 ? 1 : 0);      // this list's units of slot 0: its full ones, then a nearly full last one
                    incl = rps_wave_scan(v);
                    incl0 = rps_wave_scan(n0);
                    if (lane == kWave - 1) S->wave_tot[wave] = incl;
                    // the wave's count per slot (lane k keeps slot k's) and this list's rank among the wave's units of its slot
                    mycnt = __builtin_amdgcn_readlane(incl0, kWave - 1);
                    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
                    for (int k = 1; k < 4; ++k) {
                        const unsigned long long bk = __ballot(pslot == k);
                        if (pslot == k) rank = __popcll(bk & lt);
                        if (lane == k) mycnt = __popcll(bk);
                    }
                    if (lane < 4) S->uhist[wave * 4 + lane] = mycnt;
                }
                __syncthreads();
                if (tid < kRpsMaxPx) {
                    int base = 0;
#pragma unroll
                    for (int w = 0; w < kRpsMaxPx / kWave; ++w) base += w < wave ? S->wave_tot[w] : 0;
                    const int excl = base + incl - v;
                    if (tid <= npx) S->offs[tid] = excl & 0xFFFF;      // tid == npx: the total (c = 0 there)
                    if (tid == kRpsMaxPx - 1) {
                        if (npx == kRpsMaxPx) S->offs[npx] = (base + incl) & 0xFFFF;
                        S->n_segs = (base + incl) >> 16;
                    }
                    // where this wave's units of slot k start: the slots before k of all waves + slot k of the waves before this one --
                    // every lane forms the four values itself from the 16 counts (broadcast reads; no cross-lane traffic)
                    int sbk[4], run_ = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        int tot = 0, mine_before = 0;
#pragma unroll
                        for (int w = 0; w < kRpsMaxPx / kWave; ++w) {
                            const int x = S->uhist[w * 4 + k];
                            tot += x;
                            mine_before += w < wave ? x : 0;
                        }
                        sbk[k] = run_ + mine_before;
                        run_ += tot;
                    }
                    const int at0 = sbk[0] + incl0 - n0;
                    const int atp = (pslot == 1 ? sbk[1] : (pslot == 2 ? sbk[2] : sbk[3])) + rank;
                    for (int sgm = 0; sgm < n_full; ++sgm) S->seg[at0 + sgm] = (unsigned short)(tid | sgm << 8);
                    if (pslot == 0) S->seg[at0 + n_full] = (unsigned short)(tid | n_full << 8);
                    else if (pslot > 0) S->seg[atp] = (unsigned short)(tid | n_full << 8);
                }
            }
            __syncthreads();
            RPS_STAMP(3)

            // ---- (3) records to their sorted slots ------------------------------------------------------------------------------------
#pragma unroll
            for (int u = 0; u < kRpsRpl; ++u)
                if (pos[u] >= 0) {
                    const int e = pos[u] + S->offs[pbase[u]];
                    const unsigned qp = n_rec[u].code & ((1u << kRpsQpBits) - 1u);
                    const int q = P4 ? (int)(qp >> 2) : (int)(qp / (unsigned)P);
                    const float lwa = n_rec[u].lw * n_rec[u].a;
                    S->ent[e] = RpsEnt{n_rec[u].lh, (1.f - n_rec[u].lw) * n_rec[u].a, lwa, (unsigned)((bq0 + q) * g.M + m) * (unsigned)(kRpsD * sizeof(TV))};
                    S->slot[u * kRpsThreads + tid] = (unsigned short)e;
                }
            __syncthreads();
            RPS_STAMP(4)

            // ---- (4) the waves take groups of 16 units in turn; a quad walks its unit: four partial sums and four corner dots per
            //      point, grad_out rows straight from global memory, software-pipelined; then the sums go to the tile's f64 sums ----
            const int n_segs = rps_uni(S->n_segs);
            for (int u0 = wave * 16; u0 < n_segs; u0 += kRpsWaves * 16) {   // (uniform)
                unsigned long long wt0 = 0, wt1 = 0, wt2 = 0;      // (diagnostic: wave 0's time in a group's set-up / point loop / epilogue)
                if (kStamps && g.stamps && tid == 0) wt0 = __builtin_amdgcn_s_memtime();
                const int un = u0 + (lane >> 2);
                if (un < n_segs) {
                    const unsigned sc = S->seg[un];
                    const int my_p = (int)(sc & 0xFFu);
                    int e = S->offs[my_p] + (int)((sc >> 8) << g.seg_shift);
                    const int e1 = min(e + (1 << g.seg_shift), S->offs[my_p + 1]);
                    // value rows of the four corner pixels (clamped into the grid: a corner above / left of row / column 0 of
                    // the grid lies outside the map or belongs to a point this tile does not form gradients for)
                    const int lr = my_p / gw, lc = my_p - lr * gw;
                    const int r0 = max(lr - 1, 0) * gw, c0 = max(lc - 1, 0);
                    // (register set r of lane j4 holds corner r ^ j4 -- see rps_quad_rotated_sum --: the rotation is address arithmetic here,
                    // a per-lane choice between "this row / column" and "the other one")
                    rps_v2f v[4][4];
                    {
                        const int row_hi = lr * gw - r0, col_hi = lc - c0;                    // corner bit set: lower row / right column
                        const int row_same = (j4 & 2) ? row_hi : 0, row_flip = row_hi - row_same;   // bit 1 of r ^ j4 for r = 0, 1 / r = 2, 3
                        const int col_same = (j4 & 1) ? col_hi : 0, col_flip = col_hi - col_same;   // bit 0 of r ^ j4 for r even / odd
                        const float *const v00 = vt + (r0 + c0) * kRpsD;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float *const src = v00 + (((r & 2) ? row_flip : row_same) + ((r & 1) ? col_flip : col_same)) * kRpsD;
                            const float4 a0 = *reinterpret_cast<const float4 *>(src + c_lo);
                            const float4 a1 = *reinterpret_cast<const float4 *>(src + c_hi);
                            v[r][0] = (rps_v2f){a0.x, a0.y}; v[r][1] = (rps_v2f){a0.z, a0.w};
                            v[r][2] = (rps_v2f){a1.x, a1.y}; v[r][3] = (rps_v2f){a1.z, a1.w};
                        }
                    }
                    // partial sums: macc[m][k] = corner k, this lane's channel m (of its 8).  They are formed on the MATRIX pipe (round 4):
                    // v_mfma_f32_4x4x1 is sixteen 4 x 4 outer products, one per quad -- A = the four corner weights (lane k of the quad
                    // supplies corner k's), B = the four lanes' m-th channels, D[k][j] in register k of lane j -- so a point costs 8 MFMAs
                    // and ONE weight per lane instead of 16 packed FMAs and four weights, and the vector pipe keeps the dots.
                    rps_v4f macc[8];
#pragma unroll
                    for (int m_ = 0; m_ < 8; ++m_) macc[m_] = (rps_v4f){0.f, 0.f, 0.f, 0.f};
#define RPS_POINT(ROW, CF, EP)                                                                                                   \
    {                                                                                                                            \
        const float wj = __builtin_fmaf(w_ys, CF.x, w_yc) * CF.y;                                                                \
        rps_v2f gq[4];                                                                                                           \
        ROW.unpack(gq);                                                                                                          \
        _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                            \
        {                                                                                                                        \
            macc[2 * c] = __builtin_amdgcn_mfma_f32_4x4x1f32(wj, gq[c].x, macc[2 * c], 0, 0, 0);                                 \
            macc[2 * c + 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wj, gq[c].y, macc[2 * c + 1], 0, 0, 0);                         \
        }                                                                                                                        \
        float d[4];                                                                                                              \
        _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                                            \
        {                                                                                                                        \
            rps_v2f t = gq[0] * v[k][0];                                                                                         \
            t += gq[1] * v[k][1];                                                                                                \
            t += gq[2] * v[k][2];                                                                                                \
            t += gq[3] * v[k][3];                                                                                                \
            d[k] = t.x + t.y;      /* this lane's part of corner k ^ j4's dot */                                               \
        }                                                                                                                        \
        /* lane k of the quad writes dot k over the entry (all four lanes have read it) */                                       \
        *RPS_LDS(float, (EP) + 4u * (unsigned)j4) = rps_quad_rotated_sum(d[0], d[1], d[2], d[3]);                                \
    }
                    // software pipeline: while point e is reduced, the grad_out row of point e + 1 is in flight and the row offset
                    // of point e + 2 is being read (the chain entry -> row address -> row is what a walk waits for).  Only the
                    // row offset is read ahead; the weight's two words are read when the point is reduced (registers).
                    // (entries are addressed by BYTE offset into the chunk's entry array: one add per two points)
#if defined(RPS_WALK_ABLATE) && RPS_WALK_ABLATE == 1      // (diagnostic: no row requests at all -- wrong results)
#define RPS_ROW(OFF, ROW) asm volatile("" : "+v"(ROW.a.x), "+v"(ROW.a.y), "+v"(ROW.b.x), "+v"(ROW.b.y), "+v"(OFF));
#elif defined(RPS_WALK_ABLATE) && RPS_WALK_ABLATE == 2    // (diagnostic: every request hits one of 64 rows -- wrong results)
#define RPS_ROW(OFF, ROW) ROW.load(grad_out, ((OFF) & 0x1F80u) + lane_row);
#else
#define RPS_ROW(OFF, ROW) ROW.load(grad_out, (OFF) + lane_row);
#endif
#define RPS_OFF(EP) (*RPS_LDS(const unsigned, (EP) + 12u))
                    // (ep = LDS ADDRESS of the entry: the array's base is added once, behind an opaque asm -- derived from the symbol inside
                    // the loop, every address costs an extra add of the "+ 0" the LDS layout pass leaves behind)
#define RPS_LDS(T, A) (reinterpret_cast<__attribute__((address_space(3))) T *>((size_t)(A)))
                    unsigned ent_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)(S->ent);
                    asm volatile("" : "+v"(ent_lds));
                    unsigned ep = ent_lds + (unsigned)e * 16u;
                    const unsigned ep_end = ent_lds + (unsigned)e1 * 16u, ep_last = ep_end - 16u;
                    if (kStamps && g.stamps && wave == 0) {      // (diagnostic: the longest unit of wave 0's group = the steps its point loop takes)
                        int len = e1 - e;
                        for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o));
                        if (tid == 0) S->stamp_acc[13] += (unsigned long long)len;
                    }
                    // (a point's two weight words -- RPS_CF -- are read one iteration ahead as well: with the loop no longer bound by instruction
                    // issue, the two LDS round trips in front of a point's first MFMA were on its critical path.  A ring of THREE rows -- a
                    // row requested two steps ahead -- was tried in round 5 and gained nothing, with or without scheduling barriers: the steps
                    // do not wait for one row's latency but for the CU's vector-memory pipe, see profiles/r05_routed_experiments.md)
#define RPS_CF(EP) make_float2(*RPS_LDS(const float, EP), *RPS_LDS(const float, (EP) + w_xoff))
                    unsigned offA = RPS_OFF(ep), offB = RPS_OFF(min(ep + 16u, ep_last));
                    float2 cA = RPS_CF(ep), cB = RPS_CF(min(ep + 16u, ep_last));
                    RpsRow<TV> gA, gB;
                    RPS_ROW(offA, gA)
                    if (kStamps && g.stamps && tid == 0) wt1 = __builtin_amdgcn_s_memtime();
                    for (; ep + 16u < ep_end; ep += 32u) {
                        RPS_ROW(offB, gB)
                        const unsigned nA = min(ep + 32u, ep_last), nB = min(ep + 48u, ep_last);
                        offA = RPS_OFF(nA);
                        const float2 cA2 = RPS_CF(nA);
                        RPS_POINT(gA, cA, ep)
                        RPS_ROW(offA, gA)
                        offB = RPS_OFF(nB);
                        const float2 cB2 = RPS_CF(nB);
                        RPS_POINT(gB, cB, ep + 16u)
                        cA = cA2;
                        cB = cB2;
                    }
                    if (ep < ep_end) RPS_POINT(gA, cA, ep)
#undef RPS_CF
#undef RPS_LDS
#undef RPS_OFF
#undef RPS_ROW
#undef RPS_POINT
                    if (kStamps && g.stamps && tid == 0) wt2 = __builtin_amdgcn_s_memtime();
                    // the unit's partial sums to the tile's f64 sums: corner k of base pixel p is pixel p-gw-1 / p-gw / p-1 / p of
                    // the pixel grid, where that pixel exists (else it lies outside the map or in the tile above / to the left)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool ok = (k >= 2 || lr >= 1) && ((k & 1) || lc >= 1);
                        if (ok) {
                            double *dst = S->sum + (my_p - (k < 2 ? gw : 0) - ((k & 1) ? 0 : 1)) * kRpsSumStride + j4;
#pragma unroll
#if defined(RPS_EPI_ABLATE) && RPS_EPI_ABLATE == 1      // (diagnostic: no f64 atomics -- wrong results; the sums are kept alive by one plain store)
                            for (int m_ = 0; m_ < 1; ++m_) *reinterpret_cast<float *>(dst) = macc[0][k] + macc[1][k] + macc[2][k] + macc[3][k] + macc[4][k] + macc[5][k] + macc[6][k] + macc[7][k];
#elif defined(RPS_EPI_ABLATE) && RPS_EPI_ABLATE == 2    // (diagnostic: plain f64 read-modify-write instead of atomics -- racy, wrong results)
                            for (int m_ = 0; m_ < 8; ++m_) dst[4 * m_] += (double)macc[m_][k];
#else
                            for (int m_ = 0; m_ < 8; ++m_) atomicAdd(dst + 4 * m_, (double)macc[m_][k]);
#endif
                        }
                    }
                }
                if (kStamps && g.stamps && tid == 0 && wt1) {
                    const unsigned long long wt3 = __builtin_amdgcn_s_memtime();
                    S->stamp_acc[9] += wt1 - wt0;
                    S->stamp_acc[10] += wt2 - wt1;
                    S->stamp_acc[11] += wt3 - wt2;
                    S->stamp_acc[12] += 1;
                }
            }
            // the next chunk -- or the first chunk of the next work item -- is requested now, behind the walk (held across it,
            // the eight registers would spill) and AHEAD of this chunk's gradient stores
            // (both requests are issued whatever the chunk -- the run table of the NEXT item is parked in LDS only behind the last chunk's
            // barrier, the records are those of the next chunk, or a harmless re-read of this one behind the last --: inside
            // `if (last_chunk) ... else ...` the compiler waited for them on the spot, 1-2 us per chunk that were meant to pass under
            // the gradient stage below)
            const bool last_chunk = ch + 1 == n_chunks;
            bin_counts(next_st, next_ok, next_n, next_runs);
            load_runs(next_bin, next_runs);
            fetch_recs(n_ent, n_runs, min(ch + 1, n_chunks - 1));
            __syncthreads();
            if (last_chunk) park_runs();
            RPS_STAMP(5)

            // ---- (5) gradients of the points this tile owns: one lane per record, in arrival order; four store instructions issued
            //      by every lane (see the kernel's header) ---------------------------------------------------------------------------
            float ga_[kRpsRpl], gx_[kRpsRpl], gy_[kRpsRpl];
            unsigned pt_[kRpsRpl];
            bool mine_[kRpsRpl];
#pragma unroll
            for (int u = 0; u < kRpsRpl; ++u) {
                const int k = min(u * kRpsThreads + tid, n_here - 1);
                const RpsRec r = S->meta[k];
                float4 d = *reinterpret_cast<const float4 *>(S->ent + S->slot[k]);
                const unsigned in = r.code >> 27;
                if (!(in & 1u)) d.x = 0.f;
                if (!(in & 2u)) d.y = 0.f;
                if (!(in & 4u)) d.z = 0.f;
                if (!(in & 8u)) d.w = 0.f;
                const float lh = r.lh, lw = r.lw, hh = 1.f - lh, hw = 1.f - lw;
                const float s_a = hh * hw * d.x + hh * lw * d.y + lh * hw * d.z + lh * lw * d.w;
                const float s_w = hh * (d.y - d.x) + lh * (d.w - d.z);
                const float s_h = hw * (d.z - d.x) + lw * (d.w - d.y);
                const unsigned qp = r.code & ((1u << kRpsQpBits) - 1u);
                const unsigned q = P4 ? qp >> 2 : qp / (unsigned)P, pp = qp - q * (unsigned)P;
                pt_[u] = (unsigned)((bq0 + (int)q) * g.M + m) * (unsigned)LP + (unsigned)(l * P) + pp;
                mine_[u] = u * kRpsThreads + tid < n_here && (r.code >> 31);
                ga_[u] = s_a;
                gx_[u] = (float)W * s_w * r.a;
                gy_[u] = (float)H * s_h * r.a;
            }
#pragma unroll
            for (int u = 0; u < kRpsRpl; ++u) {
                *(mine_[u] ? grad_aw + pt_[u] : dummy_w + lane) = ga_[u];
                *(mine_[u] ? reinterpret_cast<float2 *>(grad_loc + 2u * pt_[u]) : reinterpret_cast<float2 *>(dummy_w) + lane) = make_float2(gx_[u], gy_[u]);
            }
            // (the next chunk clears the histogram now -- its last reader was the list walk -- and rewrites the entries only
            // after three more barriers)
            RPS_STAMP(6)
        }
        if (n_chunks == 0) {   // (an empty bin: nothing was requested ahead)
            draw = atomicAdd(draw_at, draw_by);
            bin_counts(next_st, next_ok, next_n, next_runs);
            load_runs(next_bin, next_runs);
            __syncthreads();
            park_runs();
        }
        // the NEXT item's value rows are requested now and travel while this item's sums are stored and the next item's are cleared
        // (they are parked in LDS behind the next item's first barrier)
        fetch_rows(nit);
        // ---- the tile's sums to grad_value: one 128-B row per pixel (two pixels per quad) ------------------------------------------
        __syncthreads();
        fetch_recs(next_n, next_runs, 0);      // the next item's first chunk (its run table is in LDS now) travels under the stores below
        if (tid == 0 && e_bin != 0xFFFFFFFFu) g.bin_state[(size_t)e_bin * (kRpsPad / 2)] = 0ull;      // this bin is consumed: its counter is zero for the next call
        if (!g.lv[l].atomic) {   // (uniform)
            // interior pixels of the tile -- everything but its first row / column -- belong to this workgroup alone: plain stores
#pragma unroll
            for (int r = 0; r < kRpsPpq; ++r) {
                const int px = quad + r * (kRpsThreads / 4);
                const int pxc = min(px, kRpsMaxPx - 1);
                const int gr = pxc / gw, gc = pxc - gr * gw;
                const int prow = R0 + gr, pcol = C0 + gc;
                const bool interior = px < npx && gr >= 1 && gc >= 1 && prow < R1 && pcol < C1;
                double *src = S->sum + pxc * kRpsSumStride + j4;
                const int64_t px_off = ((int64_t)(b * g.S + g.lv[l].start + prow * W + pcol) * g.M + m) * kRpsD;
                TV *const dst = interior ? grad_value + px_off : reinterpret_cast<TV *>(dummy_w) + 4 * lane - c_lo;
                st4(dst + c_lo, make_float4((float)src[0], (float)src[4], (float)src[8], (float)src[12]));
                st4(interior ? dst + c_hi : dst + c_lo, make_float4((float)src[16], (float)src[20], (float)src[24], (float)src[28]));
                if (interior) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) src[4 * k] = 0.0;      // (read out: zero for the next work item)
                }
            }
            // The tile's first row / column and its apron (the row below / the column right of it: the lower / right corners of the points
            // on its last row / column) are shared with the neighbouring tiles: ADDED to rows the route pass has zeroed, one channel per lane,
            // so that a wave instruction adds two whole 128-B rows.  (Rounds 2-4 routed a second record to the neighbour instead.)
            // (the perimeter of the pixel grid, walked without divisions: its first and last row, then the two columns between them)
            const int c32 = tid & 31;
            const int64_t tile_base = ((int64_t)(b * g.S + g.lv[l].start) * g.M + m) * kRpsD + c32;
            const int gh = R1 - R0 + 1, n_border = npx ? 2 * (gw + gh) - 4 : 0;
            for (int k = tid >> 5; k < n_border; k += kRpsThreads / 32) {
                const int k2 = k - 2 * gw;
                const int pr = k < gw ? 0 : (k2 < 0 ? gh - 1 : 1 + (k2 >> 1));
                const int pc = k < gw ? k : (k2 < 0 ? k - gw : ((k2 & 1) ? gw - 1 : 0));
                const int row = R0 + pr, col = C0 + pc;
                double *const sp = S->sum + (pr * gw + pc) * kRpsSumStride + (c32 & 16) + 4 * (c32 & 3) + ((c32 & 15) >> 2);
                const float x = (float)*sp;
                *sp = 0.0;
                if (row < H && col < W && x != 0.f) atomicAdd(grad_acc + tile_base + (int64_t)(row * W + col) * row_elems, x);
            }
        } else if (n_chunks > 0) {
            // several workgroups share the tile: its rows -- apron included -- are ADDED to the (pre-zeroed) level, one channel per lane, so
            // that a wave instruction adds two whole 128-B rows (32-B atomic segments run ~4x slower)
            const int c32 = tid & 31;
            const int64_t tile_base = ((int64_t)(b * g.S + g.lv[l].start) * g.M + m) * kRpsD + c32;
            for (int p = tid >> 5; p < npx; p += kRpsThreads / 32) {
                const int pr = p / gw, row = R0 + pr, col = C0 + (p - pr * gw);
                double *const sp = S->sum + p * kRpsSumStride + (c32 & 16) + 4 * (c32 & 3) + ((c32 & 15) >> 2);
                const float x = (float)*sp;
                *sp = 0.0;
                if (row < H && col < W && x != 0.f) atomicAdd(grad_acc + tile_base + (int64_t)(row * W + col) * row_elems, x);
            }
        }
        __syncthreads();
        RPS_STAMP(7)
        item_id = next_id;
        e_bin = (unsigned)rps_uni((int)next_bin);
        n_ent = rps_uni(next_n);
        n_runs = rps_uni(next_runs);
        it = nit;
        par ^= 1;
    }
    if (kStamps && g.stamps && tid == 0) {
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();
        S->stamp_acc[8] += now_ - S->stamp_last;   // waiting at the empty queue
        for (int i = 0; i < 14; ++i) g.stamps[(size_t)blockIdx.x * 16 + i] = S->stamp_acc[i];
    }
}

// bf16 storage: what is accumulated with atomics lives in the fp32 image `acc` -- the slabbed levels, and the first row / column of every
// tile of the others --; round it into grad_value once.
__global__ __launch_bounds__(256) void rps_round_kernel(const float *__restrict__ acc, bf16_t *__restrict__ grad_value, const RpsGeom g)
{
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    const int row4 = g.M * kRpsD / 4;
    for (int l = 0; l < g.L; ++l) {
        const RpsLevel &v = g.lv[l];
        const int nb = v.nty * v.W + v.ntx * v.H;      // border pixels of the tile grid (see rps_route_kernel's prologue)
        const int n4 = (v.atomic ? v.H * v.W : nb) * row4;
        for (int b = 0; b < g.N; ++b) {
            const size_t base = (size_t)(b * g.S + v.start) * row4 * 4;
            for (int i = gtid; i < n4; i += gsz) {
                size_t at = (size_t)i;
                if (!v.atomic) {
                    const int k = i / row4, c = i - k * row4;
                    int px;
                    if (k < v.nty * v.W) {
                        const int ty = k / v.W;
                        px = ty * v.TH * v.W + (k - ty * v.W);
                    } else {
                        const int k2 = k - v.nty * v.W, tx = k2 / v.H;
                        px = (k2 - tx * v.H) * v.W + tx * v.TW;
                    }
                    at = (size_t)px * row4 + c;
                }
                st4(grad_value + base + 4 * at, *reinterpret_cast<const float4 *>(acc + base + 4 * at));
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
struct RpsOptions {
    std::atomic<int> tile{16};         // largest tile side + 1 (tile + one row / column <= kRpsMaxPx = 256 pixels)
    std::atomic<int> max_chunks{12};   // expected chunks of one workgroup before a tile is split into slabs (MI355X, call E: 6 -> 12 = 463 -> 447 us on uniform locations, equal at the init pattern)
    std::atomic<int> route_wgs{2};     // route pass: workgroups per CU (persistent over the query blocks): two are resident at its 102 registers (MI355X, call E: 2 -> 55.9 us, 3 -> 61.7, 4 -> 58.1, 6 -> 63.1)
    std::atomic<int> order{0};         // work-queue order of the tile kernel: 0 = heaviest tile first, 1 = region-major (see plan_rps)
    std::atomic<int> seg_shift{4};     // units of the list walk: at most 1 << seg_shift points of a pixel's list (3..11; MI355X,
                                       // call E, list walk: 8 -> 315 k cycles per workgroup, 16 -> 301 k, 32 -> 330 k, whole lists -> 347 k)
};
inline RpsOptions &rps_options()
{
    static RpsOptions o;
    return o;
}

struct RpsPlan {
    bool ok = false;
    RpsGeom g{};
    size_t max_entries = 0;   // capacity the entry list needs (every point can sit in up to four bins)
};

inline RpsPlan plan_rps(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes, const int64_t *lsi)
{
    RpsPlan pl;
    if (D != kRpsD || L < 1 || L > kRpsMaxL || P < 1 || P > 64) return pl;
    int64_t pre = 0;
    bool tiles_s = true;
    for (int l = 0; l < L; ++l) {
        tiles_s = tiles_s && lsi[l] == pre;
        pre += shapes[2 * l] * shapes[2 * l + 1];
        if (shapes[2 * l] >= 32768 || shapes[2 * l + 1] >= 32768) return pl;
    }
    if (!tiles_s || pre != S) return pl;   // tiles must not overlap in grad_value
    const int64_t n_pts = (int64_t)N * Lq * M * L * P;
    if (n_pts >= ((int64_t)1 << 30) || (int64_t)Lq * P >= ((int64_t)1 << kRpsQpBits)) return pl;      // (4 x points: 32-bit record indices)
    RpsGeom &g = pl.g;
    g.N = N; g.S = S; g.M = M; g.Lq = Lq; g.L = L; g.P = P;
    g.seg_shift = std::max(kRpsSegShift, std::min(11, rps_options().seg_shift.load()));
    g.ppx = (N * M + kXcds - 1) / kXcds;
    const int tmax = std::max(1, rps_options().tile.load() - 1), max_chunks = std::max(1, rps_options().max_chunks.load());
    struct U { unsigned code; int64_t cost; };
    std::vector<U> units;
    int bins = 0;
    for (int l = 0; l < L; ++l) {
        RpsLevel &v = g.lv[l];
        v.H = (int)shapes[2 * l]; v.W = (int)shapes[2 * l + 1]; v.start = (int)lsi[l];
        for (int t = tmax; t >= 1; --t) {   // balanced tiles of at most t x t pixels whose grid (+1 row / column) fits the lane groups
            v.nty = (v.H + t - 1) / t; v.ntx = (v.W + t - 1) / t;
            v.TH = (v.H + v.nty - 1) / v.nty; v.TW = (v.W + v.ntx - 1) / v.ntx;
            v.nty = (v.H + v.TH - 1) / v.TH; v.ntx = (v.W + v.TW - 1) / v.TW;
            if ((v.TH + 1) * (v.TW + 1) <= kRpsMaxPx) break;
        }
        if ((v.TH + 1) * (v.TW + 1) > kRpsMaxPx || v.nty > 64 || v.ntx > 64) return pl;
        // slabs per tile, from the points a tile receives if they spread evenly over the level (they need not: a slab just
        // takes what lands in its bin)
        const double per_px = (double)Lq * P / ((double)v.H * v.W);
        const int nchunks = (int)(per_px * v.TH * v.TW * 1.05 / kRpsChunk) + 1;
        v.nslab = std::min(255, std::max(1, (nchunks + max_chunks - 1) / max_chunks));
        v.atomic = v.nslab > 1 ? 1 : 0;
        v.inv_TH = 1.0f / (float)v.TH;
        v.inv_TW = 1.0f / (float)v.TW;
        v.bin0 = bins;
        bins += v.nty * v.ntx * v.nslab;
        for (int ty = 0; ty < v.nty; ++ty)
            for (int tx = 0; tx < v.ntx; ++tx)
                for (int sl = 0; sl < v.nslab; ++sl)
                    units.push_back(U{(unsigned)l | (unsigned)ty << 2 | (unsigned)tx << 8 | (unsigned)sl << 14 | (unsigned)v.nslab << 22,
                                      (int64_t)(nchunks + v.nslab - 1) / v.nslab * 16 + 8});
    }
    if (units.size() > (size_t)kRpsMaxUnits) return pl;
    g.lut_n = 0;
    for (int l = 0; l < L; ++l) {      // (H, W <= 64 tiles of <= 15 pixels: at most 4 x 2 x 962 words = 30 KB of LDS)
        g.lut_r[l] = g.lut_n;
        g.lut_n += g.lv[l].H + 1;
        g.lut_c[l] = g.lut_n;
        g.lut_n += g.lv[l].W + 1;
    }
    if (rps_options().order.load() == 0) {
        std::stable_sort(units.begin(), units.end(), [](const U &a, const U &b) { return a.cost > b.cost; });      // heaviest first
    } else {
        // Region-major order (round 4): the slabbed (heaviest) tiles first, heaviest first as before; then every other tile in the order
        // of a coarse-to-fine walk over the image -- a tile of the coarsest level, then the tiles of the next finer level whose centres
        // lie in it, each followed by ITS finer tiles, ... -- so that the tiles the points of one group of queries land on at the
        // different levels are drawn from the queue one after the other and run at about the same time on the same XCD: the four
        // 32-byte pieces of a (query, head)'s 128-byte grad_loc line (one per level, written by four workgroups) then meet in that
        // XCD's L2 instead of going out to memory one by one.
        std::vector<int> order;      // the levels without slabs, coarsest tile grid first
        for (int l = 0; l < L; ++l)
            if (g.lv[l].nslab == 1) order.push_back(l);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return g.lv[a].nty * g.lv[a].ntx < g.lv[b].nty * g.lv[b].ntx; });
        auto key_of = [&](unsigned code) {
            const int l = code & 3, ty = (code >> 2) & 63, tx = (code >> 8) & 63, nslab = (code >> 22) & 255;
            if (nslab > 1) return (unsigned long long)0;      // (kept in front, by cost)
            const RpsLevel &v = g.lv[l];
            const double cy = (std::min(v.H, ty * v.TH + v.TH) + ty * v.TH) * 0.5 / v.H, cx = (std::min(v.W, tx * v.TW + v.TW) + tx * v.TW) * 0.5 / v.W;
            unsigned long long key = 1ull << 60;
            int shift = 42;
            for (int k : order) {
                const RpsLevel &u = g.lv[k];
                unsigned long long part = 0;      // (0: this unit is coarser than level k -- it sorts before the tiles inside it)
                if (u.nty * u.ntx <= v.nty * v.ntx || k == l) {
                    const int uy = k == l ? ty : std::min(u.nty - 1, (int)(cy * u.H / u.TH)), ux = k == l ? tx : std::min(u.ntx - 1, (int)(cx * u.W / u.TW));
                    part = (unsigned long long)((uy + 1) << 7 | (ux + 1));
                }
                key |= part << shift;
                shift -= 14;
                if (k == l) break;
            }
            return key;
        };
        std::stable_sort(units.begin(), units.end(), [&](const U &a, const U &b) {
            const unsigned long long ka = key_of(a.code), kb = key_of(b.code);
            return ka != kb ? ka < kb : a.cost > b.cost;
        });
    }
    g.nunits = (int)units.size();
    for (int i = 0; i < g.nunits; ++i) g.units[i] = units[i].code;
    g.bins_per_pair = bins;
    g.nbins = bins * N * M;
    // runs a bin can get: one per route work item of its pair (the route pass's block of queries: 16 queries per wave at P <= 4).  The
    // tile kernel keeps a bin's run table in LDS (kRpsMaxRuns entries): where 8-wave route workgroups would write more runs than that
    // (Lq > 40960 at P <= 4; the 1280 x 1280 mosaic batches of ImageNet-LVIS, reference datasets/transforms.py:356-357,437-445, have
    // Lq = 34000: 266 runs) the route pass runs with 16 waves per workgroup -- half the runs.
    const int qpw = P <= 4 ? 16 : (P <= 8 ? 8 : (P <= 16 ? 4 : (P <= 32 ? 2 : 1)));
    g.route_threads = 512;
    g.max_runs = (Lq + qpw * 8 - 1) / (qpw * 8);
    if (g.max_runs > kRpsMaxRuns) {
        g.route_threads = kRpsRouteThreadsMax;
        g.max_runs = (Lq + qpw * 16 - 1) / (qpw * 16);
    }
    if (g.max_runs > kRpsMaxRuns) return pl;
    const int qpb = qpw * (g.route_threads / kWave);
    // the record pool: a stretch per route work item (pair, block of qpb queries): one record per point
    pl.max_entries = (size_t)N * M * g.max_runs * (size_t)qpb * L * P;
    if (pl.max_entries >= ((size_t)1 << 32)) return pl;
    pl.ok = true;
    return pl;
}

}  // namespace msda
