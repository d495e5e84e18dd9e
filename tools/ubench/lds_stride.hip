// Micro-benchmark: LDS cycles of the mix kernel's pair read (ds_read2_b32 a[i], a[i+1]) as a
// function of the resample ratio ds, for candidate window layouts.  Lanes 0-31 = left ear,
// 32-63 = right ear (window offset D); lane l reads sample floor((16*l + k) * ds) at step k, like
// phase B of spatial_mix.  Addresses are computed on the host.  Prints 2.4 GHz cycles per wave
// instruction with 16 waves on one CU (4.0 = the 128 B/clk LDS peak).  gfx950.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define N_ITER 1000
__global__ void k(float* out, const int* __restrict__ table) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int addr[16];
    const int lds_base = (int)(size_t)(__attribute__((address_space(3))) float*)lds;
    for (int kk = 0; kk < 16; ++kk) addr[kk] = table[lane * 16 + kk] + lds_base;
    for (int it = 0; it < N_ITER; ++it) {
        double d[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(d[kk]) : "v"(addr[kk]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) asm volatile("" :: "v"(d[kk]));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)addr[0] + lds[threadIdx.x];
}
static double run(float* out, int* dtab, const std::vector<int>& tab) {
    (void)hipMemcpy(dtab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, out, dtab);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, out, dtab);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)ms * 1e-3 * 2.4e9 / (N_ITER * 16.0 * 16.0);
}
// layout: 0 plain, 1 one pad per 16, 2 one pad per 8, 3 one pad per 32, 4 xor swizzle ((a>>4)&15)
static int place(int a, int layout) {
    switch (layout) {
    case 1: return a + (a >> 4);
    case 2: return a + (a >> 3);
    case 3: return a + (a >> 5);
    default: return a;
    }
}
int main() {
    float* out; int* dtab;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&dtab, 1024 * sizeof(int));
    printf("# ds plain pad16 pad8 pad32\n");
    for (int i = -40; i <= 40; ++i) {
        const float ds = 1.0f + 0.0025f * (float)i;
        double r[4];
        for (int layout = 0; layout < 4; ++layout) {
            std::vector<int> tab(64 * 16);
            for (int lane = 0; lane < 64; ++lane)
                for (int kk = 0; kk < 16; ++kk) {
                    const int l = lane & 31;
                    const int a = (int)std::floor((float)(16 * l + kk) * ds + 0.37f) + (lane >= 32 ? 13 : 0);
                    tab[lane * 16 + kk] = (place(a, layout) & 16383) * 4;
                }
            r[layout] = run(out, dtab, tab);
        }
        printf("%.4f %.2f %.2f %.2f %.2f\n", ds, r[0], r[1], r[2], r[3]);
        if (false) { float h[64]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); printf("# addr[0] of lanes (last layout):"); for (int q = 0; q < 64; q += 4) printf(" %g", h[q]); printf("\n"); }
    }
    return 0;
}
