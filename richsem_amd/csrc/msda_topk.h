// msda_topk.h -- top-k query selection of the two-stage encoder output (reference models/richsem/deformable_transformer.py:370-372:
// topk_proposals = torch.topk(enc_outputs_class_unselected.max(-1)[0], num_queries, dim=1)[1]) as one workgroup per image
// (index work, bit-exact).
//
// A row (one image: 22323 scores, 900 wanted) lives in LDS as order-preserving 32-bit keys.  A four-pass radix select (8 bits per
// pass, LDS histogram) finds the key of the k-th largest score; the elements above it, plus -- in index order -- as many of the
// elements equal to it as are still missing, are compacted and sorted by (score descending, index ascending) with a bitonic
// network.  For pairwise different scores that is exactly torch.topk's (sorted) result; equal scores are taken lowest index first
// (torch.topk leaves their order unspecified).
#pragma once

#include <stdint.h>

#include "msda_common.h"

namespace msda {

constexpr int kTopkThreads = 1024;
constexpr int kTopkMaxK = 1024;
constexpr int kTopkMaxN = 36864;   // keys of a row in LDS: 144 KB

__device__ __forceinline__ unsigned topk_key(float f)   // ascending float order -> ascending unsigned order (NaN above +inf)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(kTopkThreads) void topk_rows_kernel(const float *__restrict__ scores, int n, int k,
                                                                 int64_t *__restrict__ out_idx, float *__restrict__ out_val)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *keys = reinterpret_cast<unsigned *>(smem);                 // [n]
    unsigned *hist = keys + ((n + 3) & ~3);                              // [256]
    unsigned *ckey = hist + 256;                                         // [kTopkMaxK] candidates: key
    int *cidx = reinterpret_cast<int *>(ckey + kTopkMaxK);               // [kTopkMaxK] candidates: index
    unsigned *scan = reinterpret_cast<unsigned *>(cidx + kTopkMaxK);     // [kTopkThreads]
    __shared__ unsigned s_prefix, s_need, s_count;

    const int tid = threadIdx.x;
    const float *row = scores + (size_t)blockIdx.x * n;
    for (int i = tid; i < n; i += kTopkThreads) keys[i] = topk_key(row[i]);
    if (tid == 0) { s_prefix = 0u; s_need = (unsigned)k; }
    __syncthreads();

    // ---- radix select: after the pass over bits [shift, shift + 8) the k-th largest key has the known bits s_prefix above `shift`
    //      and is the s_need-th largest among the keys that share them
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += kTopkThreads) hist[i] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix, hi_mask = shift == 24 ? 0u : ~0u << (shift + 8);
        for (int i = tid; i < n; i += kTopkThreads) {
            const unsigned key = keys[i];
            if ((key & hi_mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned need = s_need, d = 255u;
            for (;; --d) {   // digits from the top: the first one whose count reaches `need` holds the k-th largest
                const unsigned c = hist[d];
                if (c >= need || d == 0u) break;
                need -= c;
            }
            s_prefix = prefix | d << shift;
            s_need = need;
        }
        __syncthreads();
    }
    const unsigned thr = s_prefix, n_equal_wanted = s_need;   // key of the k-th largest; how many elements equal to it belong to the top k

    // ---- compact: everything above thr, and the first n_equal_wanted elements equal to thr in index order ---------------------
    if (tid == 0) s_count = 0u;
    const int per = (n + kTopkThreads - 1) / kTopkThreads, i0 = tid * per, i1 = min(n, i0 + per);   // contiguous chunk per thread
    unsigned eq = 0;
    for (int i = i0; i < i1; ++i) eq += keys[i] == thr ? 1u : 0u;
    scan[tid] = eq;
    __syncthreads();
    for (int d = 1; d < kTopkThreads; d <<= 1) {   // inclusive scan of the per-thread counts of "equal" elements
        const unsigned t = tid >= d ? scan[tid - d] : 0u;
        __syncthreads();
        scan[tid] += t;
        __syncthreads();
    }
    unsigned eq_rank = scan[tid] - eq;   // "equal" elements before this thread's chunk
    for (int i = i0; i < i1; ++i) {
        const unsigned key = keys[i];
        bool take = key > thr;
        if (key == thr) {
            take = eq_rank < n_equal_wanted;
            ++eq_rank;
        }
        if (take) {
            const unsigned slot = atomicAdd(&s_count, 1u);
            ckey[slot] = key;
            cidx[slot] = i;
        }
    }
    __syncthreads();
    for (int i = (int)s_count + tid; i < kTopkMaxK; i += kTopkThreads) {   // padding sorts to the end
        ckey[i] = 0u;
        cidx[i] = 0x7FFFFFFF;
    }
    __syncthreads();

    // ---- bitonic sort of the 1024 (key, index) pairs: key descending, index ascending -------------------------------------------
    for (int size = 2; size <= kTopkMaxK; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int a = tid, b = tid ^ stride;
            if (b > a) {
                const unsigned ka = ckey[a], kb = ckey[b];
                const int ia = cidx[a], ib = cidx[b];
                const bool a_first = ka > kb || (ka == kb && ia < ib);   // a belongs before b in the final order
                const bool up = (a & size) == 0;                          // this sub-sequence is sorted in the final order
                if (up != a_first) {
                    ckey[a] = kb; ckey[b] = ka;
                    cidx[a] = ib; cidx[b] = ia;
                }
            }
            __syncthreads();
        }
    if (tid < k) {
        out_idx[(size_t)blockIdx.x * k + tid] = cidx[tid];
        if (out_val) out_val[(size_t)blockIdx.x * k + tid] = row[cidx[tid]];
    }
}

inline size_t topk_lds_bytes(int n) { return (size_t)((n + 3) & ~3) * 4 + 256 * 4 + kTopkMaxK * 8 + kTopkThreads * 4; }

}  // namespace msda
