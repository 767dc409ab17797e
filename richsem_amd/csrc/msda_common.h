// msda_common.h -- shared device helpers for the gfx950 MSDeformAttn kernels.
//
// Semantics restated from the reference CUDA path (read as a specification, not translated):
//   sampling position   h_im = loc_y*H - 0.5, w_im = loc_x*W - 0.5, sample dropped unless
//                       h_im > -1 && w_im > -1 && h_im < H && w_im < W
//                       (reference ms_deform_im2col_cuda.cuh:279-291)
//   bilinear corners    (h_low,w_low) (h_low,w_high) (h_high,w_low) (h_high,w_high), each read
//                       only if inside [0,H)x[0,W), weights hh*hw, hh*lw, lh*hw, lh*lw
//                       (reference ms_deform_im2col_cuda.cuh:33-84)
#pragma once

#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msda {

constexpr int kWave = 64;           // gfx950 wavefront
constexpr int kXcds = 8;            // MI355X: 8 XCDs, blocks are dealt round-robin over them
constexpr int kPointBatch = 32;     // sampling points staged in LDS per pass

// C consecutive channels of one lane, loaded/stored with one instruction (16 B for f32x4/f64x2).
template <typename T, int C>
struct alignas(sizeof(T) * C) Pack {
    T v[C];
};

// Storage type of value / out / grad_out / grad_value: the compute type itself (float, double), or bfloat16 with fp32
// compute (msda_*_bf16 entry points: the reference has no half path, ms_deform_attn_cuda.cu:64,134 -- new capability).
// Sampling locations, attention weights, their gradients and every accumulation stay in the compute type.
using bf16_t = __hip_bfloat16;

template <typename T, typename TV>
__device__ __forceinline__ T to_compute(TV x) { return (T)x; }
template <>
__device__ __forceinline__ float to_compute<float, bf16_t>(bf16_t x) { return __bfloat162float(x); }

template <typename TV, typename T>
__device__ __forceinline__ TV to_storage(T x) { return (TV)x; }
template <>
__device__ __forceinline__ bf16_t to_storage<bf16_t, float>(float x) { return __float2bfloat16(x); }   // round to nearest even

// four consecutive channels as fp32, from either storage type (16-B or 8-B aligned address)
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t *p)
{
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float *p, const float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void st4(bf16_t *p, const float4 v)
{
    bf16_t r[4] = {__float2bfloat16(v.x), __float2bfloat16(v.y), __float2bfloat16(v.z), __float2bfloat16(v.w)};
    *reinterpret_cast<uint2 *>(p) = *reinterpret_cast<const uint2 *>(r);
}
__device__ __forceinline__ float ld1(const float *p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t *p) { return __bfloat162float(*p); }

// One sampling point of one (image, query, head), resolved once and shared through LDS by the
// lanes that hold that query's channels: element offsets of the four corners into `value`
// (-1 = corner outside the map or sample dropped) and four T-typed fields whose meaning is
// kernel-specific (forward: corner weights x attention weight; backward: lh, lw, attn, unused).
template <typename T>
struct alignas(16) PointRec {
    int o[4];
    T f[4];
};

struct LevelGeom {  // one level of the pyramid, as int32 (the reference truncates the int64 too)
    int H, W, start, pad;
};

// Resolve one sampling point.  Returns false when the sample is dropped.
//   base_row  = element offset of value[b, level_start, m, 0]
//   row_elems = M*D (elements between horizontally adjacent pixels of one head)
template <typename T>
__device__ __forceinline__ bool resolve_point(T loc_x, T loc_y, int H, int W, int base_row, int row_elems,
                                              int o[4], T &lh, T &lw)
{
    const T h_im = loc_y * (T)H - (T)0.5;
    const T w_im = loc_x * (T)W - (T)0.5;
    o[0] = o[1] = o[2] = o[3] = -1;
    lh = lw = (T)0;
    if (!(h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W)) return false;
    const T hf = floor(h_im), wf = floor(w_im);
    const int h_low = (int)hf, w_low = (int)wf;
    lh = h_im - hf;
    lw = w_im - wf;
    const bool top = h_low >= 0, bot = h_low + 1 <= H - 1;
    const bool lef = w_low >= 0, rig = w_low + 1 <= W - 1;
    const int p00 = base_row + (h_low * W + w_low) * row_elems;
    if (top && lef) o[0] = p00;
    if (top && rig) o[1] = p00 + row_elems;
    if (bot && lef) o[2] = p00 + W * row_elems;
    if (bot && rig) o[3] = p00 + W * row_elems + row_elems;
    return true;
}

// XCD-affine decode of the linear block id.  Blocks b and b+8 share an XCD (observed dispatch
// order; speed only, never correctness), so all tiles of one (image, head) pair are given ids
// that are equal mod 8: that pair's value slice (S*D elements) then stays in ONE XCD's 4 MiB L2.
//   grid = kXcds * ceil(pairs/kXcds) * ntiles
__device__ __forceinline__ bool decode_block(int bid, int pairs, int ntiles, int &pair, int &tile)
{
    const int x = bid % kXcds, t = bid / kXcds;
    const int slot = t / ntiles;
    tile = t - slot * ntiles;
    pair = slot * kXcds + x;
    return pair < pairs;
}

template <typename T>
__device__ __forceinline__ T shfl_xor_t(T v, int mask)
{
    return __shfl_xor(v, mask, kWave);
}

}  // namespace msda
