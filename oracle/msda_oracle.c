/*
 * msda_oracle.c -- CPU restatement of multi-scale deformable attention (MSDeformAttn)
 * forward and backward.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker for the HIP kernels in richsem_amd/csrc.  It is not part of the
 * product path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (richsem_amd/) never falls back to it; without the HIP library the
 * product raises.
 *
 * Parity pin: this restatement is checked (tests/test_oracle_golden.py) against golden
 * vectors produced in the build container by the reference's own pure-PyTorch path
 * ms_deform_attn_core_pytorch (reference models/richsem/ops/functions/ms_deform_attn_func.py:41-61)
 * and fp64 autograd through it, using the input recipe of the reference's only test
 * (models/richsem/ops/test.py:21-36).  Generator: tests/golden/make_golden.py.
 *
 * Algorithm followed (reference file:line, all under models/richsem/ops/src/cuda/):
 *   forward  : ms_deform_im2col_cuda.cuh:237-299  (index decode, level-major then point loop,
 *              h_im = loc_h*H - 0.5, acceptance window  h_im>-1 && w_im>-1 && h_im<H && w_im<W)
 *   bilinear : ms_deform_im2col_cuda.cuh:33-84    (4 corners, zero outside [0,H)x[0,W))
 *   backward : ms_deform_im2col_cuda.cuh:87-159   (per-corner grad_value scatter-add, grad_attn_weight
 *              = top_grad*val, grad_loc = (W*grad_w_weight, H*grad_h_weight) * top_grad*attn_weight)
 *              and :301-403 (per (b,q,m,l,p): sum of the per-channel partials over the D channels)
 *   shapes   : ms_deform_attn_cuda.cu:40-79,107-152 (value (N,S,M,D); loc (N,Lq,M,L,P,2) with (x,y)
 *              order; attn (N,Lq,M,L,P); out (N,Lq,M*D); int64 spatial_shapes (L,2)=(H,W) and
 *              level_start_index (L))
 *
 * Written as plain loops: one (b,q,m) at a time, channels innermost, so the floating-point
 * evaluation order per output element equals the reference kernel's per-thread order
 * (col += bilinear(...) * weight, l-major then p; bilinear = w1*v1 + w2*v2 + w3*v3 + w4*v4).
 *
 * Threading (OpenMP) is used only for the cpu_baseline timing leg: forward is parallel over
 * (b,q); backward over (b,m,level) units, whose grad_value / grad_loc / grad_aw slices are disjoint, so no atomics are
 * needed and the result is deterministic for any thread count.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

void msda_oracle_set_threads(int n) { g_threads = n > 0 ? n : 1; }

int msda_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

#define DEFINE_ORACLE(T, SUFFIX)                                                                       \
void msda_oracle_forward_##SUFFIX(const T *value, const int64_t *shapes, const int64_t *lsi,           \
                                  const T *loc, const T *aw, int N, int S, int M, int D, int L,        \
                                  int Lq, int P, T *out)                                               \
{                                                                                                      \
    const int64_t nq = (int64_t)N * Lq;                                                                \
    _Pragma("omp parallel for num_threads(g_threads) schedule(static)")                                \
    for (int64_t bq = 0; bq < nq; ++bq) {                                                              \
        const int b = (int)(bq / Lq);                                                                  \
        for (int m = 0; m < M; ++m) {                                                                  \
            T *o = out + (bq * M + m) * D;                                                             \
            for (int c = 0; c < D; ++c) o[c] = 0;                                                      \
            const T *lp = loc + (bq * M + m) * (int64_t)L * P * 2;                                     \
            const T *wp = aw + (bq * M + m) * (int64_t)L * P;                                          \
            for (int l = 0; l < L; ++l) {                                                              \
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                          \
                const T *vl = value + ((int64_t)b * S + lsi[l]) * M * D + (int64_t)m * D;              \
                const int64_t ws = (int64_t)M * D, hs = (int64_t)W * ws;                               \
                for (int p = 0; p < P; ++p) {                                                          \
                    const T loc_w = lp[(l * P + p) * 2], loc_h = lp[(l * P + p) * 2 + 1];              \
                    const T weight = wp[l * P + p];                                                    \
                    const T h_im = (T)(loc_h * H - 0.5), w_im = (T)(loc_w * W - 0.5);                  \
                    if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;                   \
                    const int h_low = (int)floor((double)h_im), w_low = (int)floor((double)w_im);      \
                    const int h_high = h_low + 1, w_high = w_low + 1;                                  \
                    const T lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;            \
                    const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                    \
                    const int ok1 = h_low >= 0 && w_low >= 0, ok2 = h_low >= 0 && w_high <= W - 1;     \
                    const int ok3 = h_high <= H - 1 && w_low >= 0;                                     \
                    const int ok4 = h_high <= H - 1 && w_high <= W - 1;                                \
                    const T *p1 = vl + h_low * hs + w_low * ws, *p2 = p1 + ws;                         \
                    const T *p3 = p1 + hs, *p4 = p3 + ws;                                              \
                    for (int c = 0; c < D; ++c) {                                                      \
                        const T v1 = ok1 ? p1[c] : 0, v2 = ok2 ? p2[c] : 0;                            \
                        const T v3 = ok3 ? p3[c] : 0, v4 = ok4 ? p4[c] : 0;                            \
                        o[c] += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * weight;                      \
                    }                                                                                  \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    }                                                                                                  \
}                                                                                                      \
                                                                                                       \
void msda_oracle_backward_##SUFFIX(const T *value, const int64_t *shapes, const int64_t *lsi,          \
                                   const T *loc, const T *aw, const T *grad_out, int N, int S, int M,  \
                                   int D, int L, int Lq, int P, T *grad_value, T *grad_loc,            \
                                   T *grad_aw)                                                         \
{                                                                                                      \
    memset(grad_value, 0, sizeof(T) * (size_t)N * S * M * D);                                          \
    memset(grad_loc, 0, sizeof(T) * (size_t)N * Lq * M * L * P * 2);                                   \
    memset(grad_aw, 0, sizeof(T) * (size_t)N * Lq * M * L * P);                                        \
    const int units = N * M * L; /* (b,m,level): disjoint grad_value / grad_loc / grad_aw slices */    \
    _Pragma("omp parallel for num_threads(g_threads) schedule(dynamic, 1)")                            \
    for (int u = 0; u < units; ++u) {                                                                  \
        const int pr = u / L, l = u % L;                                                               \
        const int b = pr / M, m = pr % M;                                                              \
        for (int q = 0; q < Lq; ++q) {                                                                 \
            const int64_t bqm = ((int64_t)b * Lq + q) * M + m;                                         \
            const T *g = grad_out + bqm * D;                                                           \
            const T *lp = loc + bqm * (int64_t)L * P * 2;                                              \
            const T *wp = aw + bqm * (int64_t)L * P;                                                   \
            T *glp = grad_loc + bqm * (int64_t)L * P * 2;                                              \
            T *gwp = grad_aw + bqm * (int64_t)L * P;                                                   \
            {                                                                                          \
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                          \
                const int64_t base = ((int64_t)b * S + lsi[l]) * M * D + (int64_t)m * D;               \
                const T *vl = value + base;                                                            \
                T *gvl = grad_value + base;                                                            \
                const int64_t ws = (int64_t)M * D, hs = (int64_t)W * ws;                               \
                for (int p = 0; p < P; ++p) {                                                          \
                    const T loc_w = lp[(l * P + p) * 2], loc_h = lp[(l * P + p) * 2 + 1];              \
                    const T weight = wp[l * P + p];                                                    \
                    const T h_im = (T)(loc_h * H - 0.5), w_im = (T)(loc_w * W - 0.5);                  \
                    if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;                   \
                    const int h_low = (int)floor((double)h_im), w_low = (int)floor((double)w_im);      \
                    const int h_high = h_low + 1, w_high = w_low + 1;                                  \
                    const T lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;            \
                    const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                    \
                    const int ok1 = h_low >= 0 && w_low >= 0, ok2 = h_low >= 0 && w_high <= W - 1;     \
                    const int ok3 = h_high <= H - 1 && w_low >= 0;                                     \
                    const int ok4 = h_high <= H - 1 && w_high <= W - 1;                                \
                    const int64_t o1 = h_low * hs + w_low * ws, o2 = o1 + ws, o3 = o1 + hs,            \
                                  o4 = o3 + ws;                                                        \
                    T s_w = 0, s_h = 0, s_a = 0;                                                       \
                    for (int c = 0; c < D; ++c) {                                                      \
                        const T top_grad = g[c];                                                       \
                        const T tgv = top_grad * weight;                                               \
                        T gh = 0, gw = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                              \
                        if (ok1) { v1 = vl[o1 + c]; gh -= hw * v1; gw -= hh * v1;                      \
                                   gvl[o1 + c] += w1 * tgv; }                                          \
                        if (ok2) { v2 = vl[o2 + c]; gh -= lw * v2; gw += hh * v2;                      \
                                   gvl[o2 + c] += w2 * tgv; }                                          \
                        if (ok3) { v3 = vl[o3 + c]; gh += hw * v3; gw -= lh * v3;                      \
                                   gvl[o3 + c] += w3 * tgv; }                                          \
                        if (ok4) { v4 = vl[o4 + c]; gh += lw * v4; gw += lh * v4;                      \
                                   gvl[o4 + c] += w4 * tgv; }                                          \
                        const T val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;                           \
                        s_a += top_grad * val;                                                         \
                        s_w += W * gw * tgv;                                                           \
                        s_h += H * gh * tgv;                                                           \
                    }                                                                                  \
                    gwp[l * P + p] = s_a;                                                              \
                    glp[(l * P + p) * 2] = s_w;                                                        \
                    glp[(l * P + p) * 2 + 1] = s_h;                                                    \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    }                                                                                                  \
}

DEFINE_ORACLE(float, f32)
DEFINE_ORACLE(double, f64)
