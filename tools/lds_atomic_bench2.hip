// Microbenchmark 2 (tuning aid): LDS atomics in the scatter kernels' access patterns on gfx950.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/lds_atomic_bench2 tools/lds_atomic_bench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// pattern: each group of G lanes adds to G consecutive elements of a pseudo-random row (row = 128 B)
template <typename T, int G>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, int iters, int nrows, int with_valu)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *base = reinterpret_cast<T *>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = 0.f;
    __syncthreads();
    const int grp = lane / G, j = lane % G;
    unsigned rng = 1234567u + 7919u * (wave * (64 / G) + grp) + blockIdx.x;
    float f = 1.0f + lane;
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rng = rng * 1664525u + 1013904223u;
            const int row = (rng >> 10) & (nrows - 1);           // row of 128 bytes (nrows: power of two)
            T *p = base + row * (128 / sizeof(T)) + j;
            if (with_valu) { f = f * 1.0001f + 0.5f; f = __builtin_rintf(f); }
            if (sizeof(T) == 8) atomicAdd(reinterpret_cast<double *>(p), (double)f);
            else atomicAdd(reinterpret_cast<unsigned *>(p), (unsigned)(int)f);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

template <typename T, int G>
void run(const char *name, int nrows, int with_valu)
{
    const int blocks = 256, iters = 1000;
    unsigned long long *d;
    (void)hipMalloc(&d, sizeof(unsigned long long) * blocks * 16);
    hipLaunchKernelGGL((k<T, G>), dim3(blocks), dim3(1024), 131072, 0, d, iters, nrows, with_valu);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 16);
    (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * blocks * 16, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double per = sum / h.size() / (iters * 4.0);
    printf("%-34s rows=%4d valu=%d  CU cycles per wave-instr (16 waves) = %6.1f\n", name, nrows, with_valu, per / 16);
    (void)hipFree(d);
}

int main()
{
    for (int valu = 0; valu < 2; ++valu) {
        run<unsigned, 32>("u32: 2 groups x 32 lanes (row each)", 512, valu);
        run<unsigned, 32>("u32: 2 groups x 32 lanes (row each)", 16, valu);
        run<unsigned, 16>("u32: 4 groups x 16 lanes", 512, valu);
        run<double, 16>("f64: 4 groups x 16 lanes (row each)", 512, valu);
        run<double, 16>("f64: 4 groups x 16 lanes (row each)", 16, valu);
    }
    return 0;
}
