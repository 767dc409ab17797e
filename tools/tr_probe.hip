// tools/tr_probe.hip -- what ds_read_b64_tr_b16 delivers on this GPU (the wgrad kernel's operand reads rest on it).
// LDS image [8 rows][32 cols] of 16-bit values row * 100 + col; lane group G (16 lanes) reads the block rows 4 (G & 1).., cols 16 (G >> 1)..:
// lane 4 q + p of the group supplies the address of row q, columns 4 p .. 4 p + 3.  Expected: lane i gets column i, element e = row e.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short *out)
{
    __shared__ short lds[8 * 32];
    for (int i = threadIdx.x; i < 8 * 32; i += 64) lds[i] = (short)((i / 32) * 100 + i % 32);
    __syncthreads();
    const int G = threadIdx.x / 16, i = threadIdx.x % 16, q = i / 4, p = i % 4;
    const int r0 = 4 * (G & 1), c0 = 16 * (G >> 1);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(lds + (r0 + q) * 32 + c0 + 4 * p));
    *(s4 *)(out + threadIdx.x * 4) = v;
}
int main()
{
    short *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int G = l / 16, i = l % 16, r0 = 4 * (G & 1), c0 = 16 * (G >> 1);
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) {
            printf(" %4d", h[4 * l + e]);
            bad += h[4 * l + e] != (r0 + e) * 100 + c0 + i;
        }
        printf("\n");
    }
    printf("mismatches against (row r0 + e, column c0 + i): %d\n", bad);
    return bad != 0;
}
