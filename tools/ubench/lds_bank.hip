// Micro-benchmark: does the LDS of gfx950 show bank conflicts for ds_read_b32 / ds_read2_b32 /
// ds_read_b64 with lane stride S dwords?  Prints 2.4 GHz cycles per wave instruction (16 waves on one CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2000
template <int OP>
__global__ void k(float* out, int stride) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int addr[8];
    for (int kk = 0; kk < 8; ++kk) addr[kk] = ((lane * stride + 2 * kk) & 16383) * 4;
    for (int it = 0; it < N_ITER; ++it) {
        double d[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (OP == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(*(float*)&d[kk]) : "v"(addr[kk]));
            if (OP == 1) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(d[kk]) : "v"(addr[kk]));
            if (OP == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(d[kk]) : "v"(addr[kk]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) asm volatile("" :: "v"(d[kk]));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)addr[0];
}
template <int OP> double run(float* out, int stride) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(1024), 0, 0, out, stride);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(1024), 0, 0, out, stride);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)ms * 1e-3 * 2.4e9 / (N_ITER * 8.0 * 16.0);
}
int main() {
    float* out; (void)hipMalloc(&out, 1 << 20);
    printf("# stride(dwords) ds_read_b32 ds_read2_b32 ds_read_b64\n");
    const int strides[] = {1, 2, 3, 4, 8, 15, 16, 17, 32, 33, 64, 65, 128, 256};
    for (int s : strides) printf("%4d %.2f %.2f %.2f\n", s, run<0>(out, s), run<1>(out, s), run<2>(out, s));
    return 0;
}
