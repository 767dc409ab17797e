// msda_dn.h -- the integer part of the reference's contrastive-denoising set-up (prepare_for_cdn, models/richsem/dn_components.py)
// on the device: which image and which slot every denoising query belongs to, and the boolean self-attention mask that
// keeps the denoising groups apart (integer / bool work, bit-exact).
//
//   known_bid        (dn_components.py:47, :53)    image index of every (group, ground-truth box) pair
//   map_known_indice (dn_components.py:137-139)    its slot in the padded query block: index inside the image + single_pad * group
//   attn_mask        (dn_components.py:155-176)    (pad_size + num_queries)^2 bool: matching queries cannot see the denoising block,
//                                                  denoising groups cannot see each other
// Everything here is int64 / bool: the results equal the reference's exactly.
#pragma once

#include <stdint.h>

#include "msda_common.h"

namespace msda {

// cum: exclusive prefix of the per-image box counts, batch + 1 entries (device).  n = total * groups2 entries are written, where
// total = cum[batch] and groups2 = 2 * dn_number: entry i = (group g = i / total, box j = i % total).
__global__ __launch_bounds__(256) void dn_indices_kernel(const int64_t *__restrict__ cum, int batch, int64_t total, int64_t n,
                                                         int64_t single_pad, int64_t *__restrict__ known_bid,
                                                         int64_t *__restrict__ map_known_indice)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = i / total, j = i - g * total;
        int lo = 0, hi = batch;   // image b with cum[b] <= j < cum[b + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cum[mid] <= j) lo = mid;
            else hi = mid;
        }
        known_bid[i] = lo;
        map_known_indice[i] = (j - cum[lo]) + single_pad * g;
    }
}

// attn_mask[r][c] (row-major, tgt x tgt bytes, 1 = masked): dn_components.py:155-176 with group_pad = the size of one group
__global__ __launch_bounds__(256) void dn_attn_mask_kernel(unsigned char *__restrict__ mask, int64_t tgt, int64_t pad_size,
                                                           int64_t group_pad)
{
    const int64_t n = tgt * tgt;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / tgt, c = i - r * tgt;
        unsigned char m = 0;
        if (c < pad_size) {
            if (r >= pad_size) m = 1;                                   // match query cannot see the reconstruct
            else if (group_pad > 0 && r / group_pad != c / group_pad) m = 1;   // reconstruct cannot see each other
        }
        mask[i] = m;
    }
}

}  // namespace msda
