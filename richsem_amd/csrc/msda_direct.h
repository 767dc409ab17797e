// msda_direct.h -- "direct" MSDeformAttn kernels for gfx950: every shape, every D.
//
// Work decomposition (both kernels)
//   item   = one (image b, query q, head m): D output channels, L*P sampling points.
//   group  = G lanes (power of two, 1..64) holding one item; lane j owns channels
//            [(chunk*G + j)*C, +C) with C = channels per lane (one 16-B / 8-B / 4-B access).
//            D=32 f32: C=4, G=8 -> a wave works on 8 items at once and every value access is a
//            1 KiB wave-instruction made of eight 128-B rows (the widest gather shape there is).
//   block  = 4 waves; all items of a block belong to ONE (b,m) pair and to one tile of
//            `qtile` consecutive queries; pairs are pinned to XCDs (decode_block).
//   points = resolved ONCE per item (lane j resolves points j, j+G, ...), parked in LDS as
//            PointRec and re-read by the item's lanes as LDS broadcasts -- instead of every
//            channel-thread re-deriving all L*P points from global memory, which is what the
//            reference's one-thread-per-output-scalar layout does (ms_deform_im2col_cuda.cuh:255-297).
//
// Forward  (spec: reference ms_deform_im2col_cuda.cuh:237-299): out = sum_points attn * bilinear.
// Backward (spec: reference ms_deform_im2col_cuda.cuh:87-159, 301-403): grad_value by global
//          float atomics shaped as G*C*sizeof(T)-byte row segments (levels in DirectGeom::gv_skip
//          excepted: whole small levels are summed in LDS by msda_levelsum.h); grad_loc / grad_attn reduced
//          across the group's lanes with wave shuffles (no LDS tree, no barriers, no serial
//          thread-0 loop) and written exactly once -> no zero-fill needed for them.
#pragma once

#include "msda_common.h"
#include "msda_prep.h"      // prep_ld: element loads of the raw projection (fwd_direct_prep_kernel)

namespace msda {

struct DirectGeom {
    int N, S, M, D, L, Lq, P;
    int G, logG, nchunks;  // lanes per item, log2, channel chunks per lane
    int qtile, ntiles;     // queries per block, tiles per (b,m) pair
    int pbatch;            // sampling points staged in LDS per pass: min(L*P, kPointBatch)
    unsigned gv_skip;      // backward: bit l set = grad_value of level l is produced elsewhere (msda_levelsum.h)
    int head_major;        // forward, measured experiment (round 4): value is (N, M, S, D) instead of the reference's (N, S, M, D)
};

constexpr int kDirectThreads = 256;
constexpr int kDirectWaves = kDirectThreads / kWave;
constexpr int kMinGroup = 8;   // lanes per item are never fewer: bounds the items (LDS slots) per wave

// LDS: [L x LevelGeom][waves x items-per-wave x (pbatch + 1) x PointRec<T>]; the +1 record of padding
// makes the items a wave reads together (one LDS broadcast per item) start on different banks.
template <typename T>
inline size_t direct_lds_bytes(const DirectGeom &g)
{
    const int ipw = kWave / g.G;
    return sizeof(LevelGeom) * g.L + sizeof(PointRec<T>) * kDirectWaves * ipw * (g.pbatch + 1);
}

__device__ __forceinline__ void load_levels(LevelGeom *lv, const int64_t *shapes, const int64_t *lsi, int L)
{
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        lv[l].H = (int)shapes[2 * l];
        lv[l].W = (int)shapes[2 * l + 1];
        lv[l].start = (int)lsi[l];
        lv[l].pad = 0;
    }
    __syncthreads();
}

// OCC = waves per SIMD the register budget is cut for.  Measured on MI355X: the kernel waits on its gathers 78 % of the time
// (PMC), so for large calls more waves with half the loads in flight each win (call E: 190 -> 157 us), while a small call
// (decoder: 1092 queries) is faster with the deeper per-wave pipeline of OCC 4 (19.5 vs 23 us).  Large calls use OCC 6:
// at 8 the 64-register budget spills (28 B per lane, 64 MB of scratch writes per E call) and runs 2.5 % slower.
template <typename T, int C, int OCC, typename TV = T>
__global__ __launch_bounds__(kDirectThreads, OCC) void fwd_direct_kernel(
    const TV *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const T *__restrict__ loc, const T *__restrict__ aw, TV *__restrict__ out, const DirectGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LevelGeom *lv = reinterpret_cast<LevelGeom *>(smem);
    PointRec<T> *recs = reinterpret_cast<PointRec<T> *>(smem + sizeof(LevelGeom) * g.L);

    int pair, tile;
    if (!decode_block(blockIdx.x, g.N * g.M, g.ntiles, pair, tile)) return;  // block-uniform
    load_levels(lv, shapes, lsi, g.L);

    const int b = pair / g.M, m = pair - b * g.M;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int ipw = kWave >> g.logG;              // items per wave
    const int j = lane & (g.G - 1);               // lane within the group
    const int slot = wave * ipw + (lane >> g.logG);
    const int LP = g.L * g.P;
    const int row_elems = g.head_major ? g.D : g.M * g.D;
    PointRec<T> *my = recs + slot * (g.pbatch + 1);

    const int q_end = min((tile + 1) * g.qtile, g.Lq);
    for (int q0 = tile * g.qtile; q0 < q_end; q0 += kDirectWaves * ipw) {   // block-uniform trip count
        const int q = q0 + slot;
        const bool live = q < q_end;
        const int item = (b * g.Lq + q) * g.M + m;
        for (int ch = 0; ch < g.nchunks; ++ch) {
            const int c0 = (ch * g.G + j) * C;
            const bool has_c = live && c0 < g.D;
            T acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = (T)0;
            for (int p0 = 0; p0 < LP; p0 += g.pbatch) {
                const int np = min(g.pbatch, LP - p0);
                // (1) resolve this batch of points, one point per lane, into the item's LDS slot
                if (live) {
                    for (int pt = j; pt < np; pt += g.G) {
                        const int gp = p0 + pt, l = gp / g.P;
                        const Pack<T, 2> xy = *reinterpret_cast<const Pack<T, 2> *>(loc + ((int64_t)item * LP + gp) * 2);
                        const T a = aw[(int64_t)item * LP + gp];
                        const LevelGeom G_ = lv[l];
                        PointRec<T> r;
                        T lh, lw;
                        resolve_point<T>(xy.v[0], xy.v[1], G_.H, G_.W,
                                         g.head_major ? ((b * g.M + m) * g.S + G_.start) * g.D : (b * g.S + G_.start) * row_elems + m * g.D,
                                         row_elems, r.o, lh, lw);
                        const T hh = (T)1 - lh, hw = (T)1 - lw;
                        r.f[0] = hh * hw * a;
                        r.f[1] = hh * lw * a;
                        r.f[2] = lh * hw * a;
                        r.f[3] = lh * lw * a;
                        my[pt] = r;
                    }
                }
                __builtin_amdgcn_wave_barrier();  // same-wave LDS traffic is in order; stop compiler motion
                // (2) gather: every lane walks the item's points, C channels each
                if (has_c) {
#pragma unroll(OCC >= 6 ? 2 : 4)
                    for (int pt = 0; pt < np; ++pt) {
                        const PointRec<T> r = my[pt];
                        Pack<T, C> v[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (r.o[k] >= 0) {
                                const Pack<TV, C> raw = *reinterpret_cast<const Pack<TV, C> *>(value + r.o[k] + c0);
#pragma unroll
                                for (int c = 0; c < C; ++c) v[k].v[c] = to_compute<T, TV>(raw.v[c]);
                            } else {
#pragma unroll
                                for (int c = 0; c < C; ++c) v[k].v[c] = (T)0;
                            }
                        }
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            acc[c] += r.f[0] * v[0].v[c] + r.f[1] * v[1].v[c] + r.f[2] * v[2].v[c] + r.f[3] * v[3].v[c];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (has_c) {
                Pack<TV, C> o;
#pragma unroll
                for (int c = 0; c < C; ++c) o.v[c] = to_storage<TV, T>(acc[c]);
                *reinterpret_cast<Pack<TV, C> *>(out + (int64_t)item * g.D + c0) = o;
            }
        }
    }
}

// ---- forward of SMALL calls (decoder-shaped: a thousand queries), D = 32, fp32 compute -------------------------------------------------
// fwd_direct_kernel gives an item (query, head) to 8 lanes that walk its L*P points one after the other, four points' gathers in flight:
// at 1092 queries x 2 images x 8 heads that is 2184 waves -- two per SIMD -- each of which sits through four rounds of memory latency:
// 29 us for 51 MB.  Here an item takes 32 lanes: 8 channel lanes (4 channels each) x 4 point groups, every lane resolves ITS L*P/4 points
// itself (the 8 lanes of a group read the same locations: one broadcast request) and has all their corner gathers in flight at once --
// one round of latency instead of four, four times the waves --, then the four groups' partial sums meet in two exchange steps.
// No LDS but the level table, no barrier after it.  Items in their natural order (image, query, head): a wave holds two heads of a query.
constexpr int kSplitGroups = 4;           // point groups per item
constexpr int kSplitMaxPts = 8;           // points a lane can take: L*P <= 32
template <typename TV, int PPG>           // PPG: points per group (L*P / 4) as a compile-time constant; 0 = g.L * g.P / 4 at run time
__global__ __launch_bounds__(kDirectThreads, 4) void fwd_split_kernel(
    const TV *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ aw, TV *__restrict__ out, const DirectGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LevelGeom *lv = reinterpret_cast<LevelGeom *>(smem);
    load_levels(lv, shapes, lsi, g.L);
    const int lane = threadIdx.x & (kWave - 1);
    const int j = lane & 7, pg = (lane >> 3) & (kSplitGroups - 1);
    const long long n_items = (long long)g.N * g.Lq * g.M;
    const long long item_raw = ((long long)blockIdx.x * kDirectThreads + threadIdx.x) >> 5;
    const bool live = item_raw < n_items;
    const long long item = live ? item_raw : n_items - 1;      // (a valid item whatever the lane: the exchange steps need every lane)
    const int m = (int)(item % g.M);
    const int b = (int)(item / ((long long)g.Lq * g.M));
    const int LP = g.L * g.P, ppg = PPG ? PPG : LP / kSplitGroups;
    const int row_elems = g.M * g.D, c0 = 4 * j;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (PPG ? PPG : kSplitMaxPts); ++i) {
        if (!PPG && i >= ppg) break;
        const int gp = pg * ppg + i, l = gp / g.P;
        const Pack<float, 2> xy = *reinterpret_cast<const Pack<float, 2> *>(loc + (item * LP + gp) * 2);
        const float a = aw[item * LP + gp];
        const LevelGeom G_ = lv[l];
        int o[4];
        float lh, lw;
        resolve_point<float>(xy.v[0], xy.v[1], G_.H, G_.W, (b * g.S + G_.start) * row_elems + m * g.D, row_elems, o, lh, lw);
        const float hh = 1.f - lh, hw = 1.f - lw;
        const float f[4] = {hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // (a corner outside the map reads the item's own first value row and masks it: no branch around the gather)
            const Pack<TV, 4> raw = *reinterpret_cast<const Pack<TV, 4> *>(value + (o[k] >= 0 ? o[k] : (b * g.S) * row_elems + m * g.D) + c0);
            // (the VALUE is masked, not the weight: 0 x inf would be NaN where the reference adds nothing)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += f[k] * (o[k] >= 0 ? to_compute<float, TV>(raw.v[c]) : 0.f);
        }
    }
    // the four point groups of an item: lanes 8 and 16 apart
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        acc[c] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[c]), 0x128, 0xF, 0xF, true));      // row_ror:8 = lane ^ 8 inside a row of 16
        acc[c] += __shfl_xor(acc[c], 16, kWave);
    }
    if (live && pg == 0) {
        Pack<TV, 4> o_;
#pragma unroll
        for (int c = 0; c < 4; ++c) o_.v[c] = to_storage<TV, float>(acc[c]);
        *reinterpret_cast<Pack<TV, 4> *>(out + item * g.D + c0) = o_;
    }
}

// ---- the module's element-wise work FUSED into the gather -----------------------------------------
// fwd_direct_prep_kernel = fwd_direct_kernel whose point resolution reads the RAW projection of the module -- sampling offsets and
// attention logits (reference ops/modules/ms_deform_attn.py:97-99), addressed with a row stride so that both come out of one GEMM -- and
// the reference points, and does what the reference module does between the projections and the operator (:100 softmax over the L*P
// logits of a (query, head); :102-109 location = reference + offset / (W_l, H_l), or reference_xy + offset / P * reference_wh * 0.5):
// lane j of an item's group holds the logits of points j, j + G, ..., the softmax's maximum and sum are two butterfly reductions over the
// group.  sampling_loc / attn_weight are written as a by-product when the caller wants them (training: the backward needs them);
// the gather never re-reads them.  One launch instead of two for decoder-shaped calls (prep 19.5 us + gather 29 us at 1092 queries).
// Needs all L*P points of an item in one pass (L*P <= kPointBatch).
template <typename TP>
struct PrepSrc {
    const TP *offsets, *logits;            // raw projection: offsets[(n, q) * off_stride + (m * LP + pt) * 2 + {0, 1}], logits[(n, q) * log_stride + m * LP + pt]
    long long off_stride, log_stride;
    int ref_dim;                            // 2: reference points (x, y); 4: reference boxes (x, y, w, h)
};

template <typename T, int C, int OCC, typename TV, typename TP>
__global__ __launch_bounds__(kDirectThreads, OCC) void fwd_direct_prep_kernel(
    const TV *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi, const PrepSrc<TP> src,
    const T *__restrict__ ref, T *__restrict__ loc_out, T *__restrict__ aw_out, TV *__restrict__ out, const DirectGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LevelGeom *lv = reinterpret_cast<LevelGeom *>(smem);
    PointRec<T> *recs = reinterpret_cast<PointRec<T> *>(smem + sizeof(LevelGeom) * g.L);

    int pair, tile;
    if (!decode_block(blockIdx.x, g.N * g.M, g.ntiles, pair, tile)) return;  // block-uniform
    load_levels(lv, shapes, lsi, g.L);

    const int b = pair / g.M, m = pair - b * g.M;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int ipw = kWave >> g.logG;              // items per wave
    const int j = lane & (g.G - 1);               // lane within the group
    const int slot = wave * ipw + (lane >> g.logG);
    const int LP = g.L * g.P;                     // (<= pbatch: the host launches this kernel only then)
    const int row_elems = g.M * g.D;
    PointRec<T> *my = recs + slot * (g.pbatch + 1);
    constexpr int kPerLane = kPointBatch / kMinGroup;   // points a lane resolves at most

    const int q_end = min((tile + 1) * g.qtile, g.Lq);
    for (int q0 = tile * g.qtile; q0 < q_end; q0 += kDirectWaves * ipw) {   // block-uniform trip count
        const int q = q0 + slot;
        const bool live = q < q_end;
        const long long nq = (long long)b * g.Lq + (live ? q : q_end - 1);
        const long long item = nq * g.M + m;
        // (1) this lane's points: logits -> softmax over the item's group -> locations -> corner records in the item's LDS slot.
        //     (the group's lanes all belong to one item: the butterflies stay inside it, and `live` is uniform over the group)
        T lg[kPerLane];
        T mx = (T)-3.0e38;
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) {
            const int pt = j + i * g.G;
            lg[i] = pt < LP ? prep_ld<T>(src.logits + nq * src.log_stride + (long long)m * LP + pt) : (T)-3.0e38;
            mx = lg[i] > mx ? lg[i] : mx;
        }
        for (int s_ = 1; s_ < g.G; s_ <<= 1) {
            const T o = __shfl_xor(mx, s_, kWave);
            mx = o > mx ? o : mx;
        }
        T sum = (T)0;
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) {
            lg[i] = j + i * g.G < LP ? exp(lg[i] - mx) : (T)0;
            sum += lg[i];
        }
        for (int s_ = 1; s_ < g.G; s_ <<= 1) sum += __shfl_xor(sum, s_, kWave);
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) {
            const int pt = j + i * g.G;
            if (live && pt < LP) {
                const int l = pt / g.P;
                const T a = lg[i] / sum;
                const TP *o = src.offsets + nq * src.off_stride + ((long long)m * LP + pt) * 2;
                const T ox = prep_ld<T>(o), oy = prep_ld<T>(o + 1);
                const T *r = ref + (nq * g.L + l) * src.ref_dim;
                const LevelGeom G_ = lv[l];
                T lx, ly;
                if (src.ref_dim == 2) {
                    lx = r[0] + ox / (T)G_.W;
                    ly = r[1] + oy / (T)G_.H;
                } else {   // same operation order as the reference: offsets / n_points * wh * 0.5
                    lx = r[0] + ox / (T)g.P * r[2] * (T)0.5;
                    ly = r[1] + oy / (T)g.P * r[3] * (T)0.5;
                }
                if (loc_out) {
                    Pack<T, 2> xy;
                    xy.v[0] = lx;
                    xy.v[1] = ly;
                    *reinterpret_cast<Pack<T, 2> *>(loc_out + (item * LP + pt) * 2) = xy;
                    aw_out[item * LP + pt] = a;
                }
                PointRec<T> rec;
                T lh, lw;
                resolve_point<T>(lx, ly, G_.H, G_.W, (b * g.S + G_.start) * row_elems + m * g.D, row_elems, rec.o, lh, lw);
                const T hh = (T)1 - lh, hw = (T)1 - lw;
                rec.f[0] = hh * hw * a;
                rec.f[1] = hh * lw * a;
                rec.f[2] = lh * hw * a;
                rec.f[3] = lh * lw * a;
                my[pt] = rec;
            }
        }
        __builtin_amdgcn_wave_barrier();  // same-wave LDS traffic is in order; stop compiler motion
        // (2) gather: every lane walks the item's points, C channels each, once per channel chunk
        for (int ch = 0; ch < g.nchunks; ++ch) {
            const int c0 = (ch * g.G + j) * C;
            if (live && c0 < g.D) {
                T acc[C];
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] = (T)0;
#pragma unroll(OCC >= 6 ? 2 : 4)
                for (int pt = 0; pt < LP; ++pt) {
                    const PointRec<T> r = my[pt];
                    Pack<T, C> v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (r.o[k] >= 0) {
                            const Pack<TV, C> raw = *reinterpret_cast<const Pack<TV, C> *>(value + r.o[k] + c0);
#pragma unroll
                            for (int c = 0; c < C; ++c) v[k].v[c] = to_compute<T, TV>(raw.v[c]);
                        } else {
#pragma unroll
                            for (int c = 0; c < C; ++c) v[k].v[c] = (T)0;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        acc[c] += r.f[0] * v[0].v[c] + r.f[1] * v[1].v[c] + r.f[2] * v[2].v[c] + r.f[3] * v[3].v[c];
                }
                Pack<TV, C> o;
#pragma unroll
                for (int c = 0; c < C; ++c) o.v[c] = to_storage<TV, T>(acc[c]);
                *reinterpret_cast<Pack<TV, C> *>(out + item * g.D + c0) = o;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- grad_sampling_loc / grad_attn_weight of SMALL calls whose grad_value is produced elsewhere (decoder-shaped: msda_levelsum.h takes
// every level), D = 32, fp32 compute: the backward twin of fwd_split_kernel.  bwd_direct_kernel walks an item's L*P points one after the
// other -- gathers, then three shuffle reductions, per point: sixteen rounds of memory latency per wave, 33 us at 1092 queries.  Here a
// lane has its L*P/4 points' sixteen corner gathers in flight at once and the three sums of a point meet over the item's 8 channel lanes
// in three DPP steps (no LDS crossbar).
__device__ __forceinline__ float direct_group8_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return v;
}

template <typename TV, int PPG>
__global__ __launch_bounds__(kDirectThreads, 4) void bwd_split_kernel(
    const TV *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ aw, const TV *__restrict__ grad_out,
    float *__restrict__ grad_loc, float *__restrict__ grad_aw, const DirectGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LevelGeom *lv = reinterpret_cast<LevelGeom *>(smem);
    load_levels(lv, shapes, lsi, g.L);
    const int lane = threadIdx.x & (kWave - 1);
    const int j = lane & 7, pg = (lane >> 3) & (kSplitGroups - 1);
    const long long n_items = (long long)g.N * g.Lq * g.M;
    const long long item_raw = ((long long)blockIdx.x * kDirectThreads + threadIdx.x) >> 5;
    const bool live = item_raw < n_items;
    const long long item = live ? item_raw : n_items - 1;      // (a valid item whatever the lane: the reductions need every lane)
    const int m = (int)(item % g.M);
    const int b = (int)(item / ((long long)g.Lq * g.M));
    const int LP = g.L * g.P, ppg = PPG ? PPG : LP / kSplitGroups;
    const int row_elems = g.M * g.D, c0 = 4 * j;
    float go[4];
    {
        const Pack<TV, 4> raw = *reinterpret_cast<const Pack<TV, 4> *>(grad_out + item * g.D + c0);
#pragma unroll
        for (int c = 0; c < 4; ++c) go[c] = to_compute<float, TV>(raw.v[c]);
    }
    float r_a[PPG ? PPG : kSplitMaxPts], r_w[PPG ? PPG : kSplitMaxPts], r_h[PPG ? PPG : kSplitMaxPts];
#pragma unroll
    for (int i = 0; i < (PPG ? PPG : kSplitMaxPts); ++i) {
        if (!PPG && i >= ppg) break;      // (uniform)
        const int gp = pg * ppg + i, l = gp / g.P;
        const Pack<float, 2> xy = *reinterpret_cast<const Pack<float, 2> *>(loc + (item * LP + gp) * 2);
        const float a = aw[item * LP + gp];
        const LevelGeom G_ = lv[l];
        int o[4];
        float lh, lw;
        resolve_point<float>(xy.v[0], xy.v[1], G_.H, G_.W, (b * g.S + G_.start) * row_elems + m * g.D, row_elems, o, lh, lw);
        const float hh = 1.f - lh, hw = 1.f - lw;
        float v[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Pack<TV, 4> raw = *reinterpret_cast<const Pack<TV, 4> *>(value + (o[k] >= 0 ? o[k] : (b * g.S) * row_elems + m * g.D) + c0);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[k][c] = o[k] >= 0 ? to_compute<float, TV>(raw.v[c]) : 0.f;
        }
        float s_a = 0.f, s_w = 0.f, s_h = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float tgv = go[c] * a;
            s_a += go[c] * (hh * hw * v[0][c] + hh * lw * v[1][c] + lh * hw * v[2][c] + lh * lw * v[3][c]);
            s_w += (hh * (v[1][c] - v[0][c]) + lh * (v[3][c] - v[2][c])) * tgv;
            s_h += (hw * (v[2][c] - v[0][c]) + lw * (v[3][c] - v[1][c])) * tgv;
        }
        // (with the point count known the three sums are kept and stored behind the loop: no store sits between the points' gathers)
        r_a[i] = direct_group8_sum(s_a);
        r_w[i] = (float)G_.W * direct_group8_sum(s_w);
        r_h[i] = (float)G_.H * direct_group8_sum(s_h);
        if (!PPG && live && j == 0) {
            grad_aw[item * LP + gp] = r_a[i];
            Pack<float, 2> gl;
            gl.v[0] = r_w[i];
            gl.v[1] = r_h[i];
            *reinterpret_cast<Pack<float, 2> *>(grad_loc + (item * LP + gp) * 2) = gl;
        }
    }
    if (PPG && live && j == 0) {
#pragma unroll
        for (int i = 0; i < PPG; ++i) {
            const int gp = pg * PPG + i;
            grad_aw[item * LP + gp] = r_a[i];
            Pack<float, 2> gl;
            gl.v[0] = r_w[i];
            gl.v[1] = r_h[i];
            *reinterpret_cast<Pack<float, 2> *>(grad_loc + (item * LP + gp) * 2) = gl;
        }
    }
}

// TV = storage type of value / grad_out.  grad_value is accumulated with atomics in the compute type T (in bf16 mode it
// is an fp32 scratch buffer that is rounded to bf16 once, afterwards -- or nothing at all when gv_skip covers every level).
template <typename T, int C, typename TV = T>
__global__ __launch_bounds__(kDirectThreads) void bwd_direct_kernel(
    const TV *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const T *__restrict__ loc, const T *__restrict__ aw, const TV *__restrict__ grad_out,
    T *__restrict__ grad_value, T *__restrict__ grad_loc, T *__restrict__ grad_aw, const DirectGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LevelGeom *lv = reinterpret_cast<LevelGeom *>(smem);
    PointRec<T> *recs = reinterpret_cast<PointRec<T> *>(smem + sizeof(LevelGeom) * g.L);

    int pair, tile;
    if (!decode_block(blockIdx.x, g.N * g.M, g.ntiles, pair, tile)) return;
    load_levels(lv, shapes, lsi, g.L);

    const int b = pair / g.M, m = pair - b * g.M;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int ipw = kWave >> g.logG;
    const int j = lane & (g.G - 1);
    const int slot = wave * ipw + (lane >> g.logG);
    const int LP = g.L * g.P;
    const int row_elems = g.M * g.D;
    PointRec<T> *my = recs + slot * (g.pbatch + 1);

    const int q_end = min((tile + 1) * g.qtile, g.Lq);
    for (int q0 = tile * g.qtile; q0 < q_end; q0 += kDirectWaves * ipw) {
        const int q = q0 + slot;
        const bool live = q < q_end;
        const int item = (b * g.Lq + q) * g.M + m;
        // grad_out of this lane's first channel chunk stays in registers (the only chunk when D <= 64*C)
        Pack<T, C> g0;
#pragma unroll
        for (int c = 0; c < C; ++c) g0.v[c] = (T)0;
        if (live && j * C < g.D) {
            const Pack<TV, C> raw = *reinterpret_cast<const Pack<TV, C> *>(grad_out + (int64_t)item * g.D + j * C);
#pragma unroll
            for (int c = 0; c < C; ++c) g0.v[c] = to_compute<T, TV>(raw.v[c]);
        }

        for (int p0 = 0; p0 < LP; p0 += g.pbatch) {
            const int np = min(g.pbatch, LP - p0);
            if (live) {
                for (int pt = j; pt < np; pt += g.G) {
                    const int gp = p0 + pt, l = gp / g.P;
                    const Pack<T, 2> xy = *reinterpret_cast<const Pack<T, 2> *>(loc + ((int64_t)item * LP + gp) * 2);
                    const LevelGeom G_ = lv[l];
                    PointRec<T> r;
                    resolve_point<T>(xy.v[0], xy.v[1], G_.H, G_.W, (b * g.S + G_.start) * row_elems + m * g.D,
                                     row_elems, r.o, r.f[0], r.f[1]);
                    r.f[2] = aw[(int64_t)item * LP + gp];
                    r.f[3] = (T)0;
                    my[pt] = r;
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int pt = 0; pt < np; ++pt) {           // wave-uniform trip count (shuffles inside)
                const PointRec<T> r = my[pt];
                const bool add_gv = !((g.gv_skip >> ((p0 + pt) / g.P)) & 1u);   // wave-uniform (L <= 32 when set)
                const T lh = r.f[0], lw = r.f[1], a = r.f[2];
                const T hh = (T)1 - lh, hw = (T)1 - lw;
                const T w[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                T s_a = (T)0, s_w = (T)0, s_h = (T)0;
                for (int ch = 0; ch < g.nchunks; ++ch) {
                    const int c0 = (ch * g.G + j) * C;
                    if (!(live && c0 < g.D)) continue;
                    Pack<T, C> tg = g0;
                    if (ch > 0) {
                        const Pack<TV, C> raw = *reinterpret_cast<const Pack<TV, C> *>(grad_out + (int64_t)item * g.D + c0);
#pragma unroll
                        for (int c = 0; c < C; ++c) tg.v[c] = to_compute<T, TV>(raw.v[c]);
                    }
                    Pack<T, C> v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (r.o[k] >= 0) {
                            const Pack<TV, C> raw = *reinterpret_cast<const Pack<TV, C> *>(value + r.o[k] + c0);
#pragma unroll
                            for (int c = 0; c < C; ++c) v[k].v[c] = to_compute<T, TV>(raw.v[c]);
                        } else {
#pragma unroll
                            for (int c = 0; c < C; ++c) v[k].v[c] = (T)0;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const T tgv = tg.v[c] * a;
                        const T v1 = v[0].v[c], v2 = v[1].v[c], v3 = v[2].v[c], v4 = v[3].v[c];
                        s_a += tg.v[c] * (w[0] * v1 + w[1] * v2 + w[2] * v3 + w[3] * v4);
                        s_w += (hh * (v2 - v1) + lh * (v4 - v3)) * tgv;
                        s_h += (hw * (v3 - v1) + lw * (v4 - v2)) * tgv;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (add_gv && r.o[k] >= 0) atomicAdd(grad_value + r.o[k] + c0 + c, w[k] * tgv);
                    }
                }
                // sum the per-lane partials over the item's G lanes
                for (int s = 1; s < g.G; s <<= 1) {
                    s_a += shfl_xor_t(s_a, s);
                    s_w += shfl_xor_t(s_w, s);
                    s_h += shfl_xor_t(s_h, s);
                }
                if (live && j == 0) {
                    const int gp = p0 + pt;
                    const LevelGeom G_ = lv[gp / g.P];
                    grad_aw[(int64_t)item * LP + gp] = s_a;
                    Pack<T, 2> gl;
                    gl.v[0] = (T)G_.W * s_w;
                    gl.v[1] = (T)G_.H * s_h;
                    *reinterpret_cast<Pack<T, 2> *>(grad_loc + ((int64_t)item * LP + gp) * 2) = gl;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace msda
