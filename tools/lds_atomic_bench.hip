// Microbenchmark (tuning aid): cycles per wave-instruction of LDS update primitives on gfx950.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/lds_atomic_bench tools/lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum Op { ADD_F32, ADD_U32, ADD_U64, ADD_RTN_U32, PLAIN_RMW, ADD_F64, MAX_F32, ADD_F32_CONFLICT2 };

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, int iters, int nwaves_active)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *f = reinterpret_cast<float *>(smem);
    unsigned *u = reinterpret_cast<unsigned *>(smem);
    unsigned long long *u64 = reinterpret_cast<unsigned long long *>(smem);
    double *d = reinterpret_cast<double *>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) f[i] = 0.f;
    __syncthreads();
    if (wave >= nwaves_active) return;
    // each wave walks its own 4 KB region: lane-consecutive dwords (conflict-free), new row every instruction
    unsigned acc = 0;
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = wave * 1024 + ((it * 8 + r) & 15) * 64 + lane;
            if (OP == ADD_F32) atomicAdd(f + idx, 1.0f);
            if (OP == ADD_U32) atomicAdd(u + idx, 1u);
            if (OP == ADD_U64) atomicAdd(u64 + (idx & 2047), 1ull);
            if (OP == ADD_RTN_U32) acc += atomicAdd(u + idx, 1u);
            if (OP == PLAIN_RMW) f[idx] += 1.0f;
            if (OP == ADD_F64) atomicAdd(d + (idx & 2047), 1.0);
            if (OP == MAX_F32) atomicMax(u + idx, (unsigned)it);
            if (OP == ADD_F32_CONFLICT2) atomicAdd(f + wave * 1024 + ((it * 8 + r) & 15) * 64 + (lane & 31), 1.0f);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 16 + wave] = (t1 - t0) + (acc == 0xFFFFFFFFu);
}

template <int OP>
void run(const char *name, int blocks)
{
    unsigned long long *d;
    hipMalloc(&d, sizeof(unsigned long long) * blocks * 16);
    const int iters = 2000;
    for (int nw : {1, 4, 8, 16}) {
        hipMemset(d, 0, sizeof(unsigned long long) * blocks * 16);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 65536, 0, d, iters, nw);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 16);
        hipMemcpy(h.data(), d, sizeof(unsigned long long) * blocks * 16, hipMemcpyDeviceToHost);
        double sum = 0;
        int n = 0;
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < nw; ++w) { sum += (double)h[b * 16 + w]; ++n; }
        const double per_wave_instr = sum / n / (iters * 8.0);
        // CU-level: nw waves issue concurrently; throughput = nw / per_wave_instr instr per cycle
        printf("%-20s waves/CU=%2d  cycles/wave-instr (per wave)=%7.1f   CU cycles per wave-instr=%6.1f\n", name, nw,
               per_wave_instr, per_wave_instr / nw);
    }
    hipFree(d);
}

int main()
{
    const int blocks = 256;
    run<ADD_F32>("ds_add_f32", blocks);
    run<ADD_U32>("ds_add_u32", blocks);
    run<ADD_U64>("ds_add_u64", blocks);
    run<ADD_RTN_U32>("ds_add_rtn_u32", blocks);
    run<PLAIN_RMW>("plain read+add+write", blocks);
    run<ADD_F64>("ds_add_f64", blocks);
    run<MAX_F32>("ds_max_u32", blocks);
    run<ADD_F32_CONFLICT2>("ds_add_f32 2-lane dup", blocks);
    return 0;
}
