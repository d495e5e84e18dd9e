// Micro-benchmark: how many plain f32 VALU instructions per nanosecond ONE SIMD of a busy MI355X
// issues, by waves per SIMD (the whole chip runs the same loop, so the clocks are the loaded
// ones).  Settles whether the mix kernel (about 1000 VALU wave-instructions per SIMD per
// "round" of 16 source-tiles per CU) can be VALU-issue bound.  Wall time from hipEvents,
// cycles from s_memtime and the constant 100 MHz s_memrealtime.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4000
#define REP 16

template <int DEP>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* ticks, float a, float b) {
    float x0 = a + threadIdx.x, x1 = b + threadIdx.x, x2 = a * 2 + threadIdx.x, x3 = b * 3 + threadIdx.x;
    float x4 = a * 5 + threadIdx.x, x5 = b * 7 + threadIdx.x, x6 = a * 11 + threadIdx.x, x7 = b * 13 + threadIdx.x;
    const float y0 = a;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = wall_clock64();
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (DEP == 0)   // 8 independent chains: sub, mul, add, mul, ... (the mix loop's op mix)
                asm volatile("v_sub_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y0));
            else            // one dependent chain of 8
                asm volatile("v_sub_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                             "v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n v_add_f32 %0, %0, %1"
                             : "+v"(x0) : "v"(y0));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = r1 - r0; }
}

template <int DEP> void run(const char* name, int threads, int blocks) {
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, (size_t)blocks * threads * sizeof(float)); (void)hipMalloc(&ticks, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<DEP>, dim3(blocks), dim3(threads), 0, 0, out, ticks, 1.0f, 1.0001f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        if (rep < 2) continue;
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t[2]; (void)hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost);
        const double instr_per_wave = (double)N_ITER * REP * 8;
        const double waves_per_simd = (double)threads / 256.0 * (blocks > 256 ? blocks / 256.0 : 1.0);
        const double ns = ms * 1e6;
        printf("%-12s thr=%4d blocks=%4d waves/SIMD=%.0f | wall %.3f ms | per SIMD %.3f instr/ns | s_memtime ticks/instr/wave %.2f (tick %.0f MHz) | realtime %.0f MHz\n",
               name, threads, blocks, waves_per_simd, ms, instr_per_wave * waves_per_simd / ns, (double)t[0] / instr_per_wave, t[0] / (ms * 1e3),
               t[1] / (ms * 1e3));
    }
    (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
    for (int pass = 0; pass < 2; ++pass) {
        run<0>("independent", 256, 256);    // 1 wave / SIMD, every CU busy
        run<0>("independent", 512, 256);    // 2
        run<0>("independent", 1024, 256);   // 4
        run<0>("independent", 1024, 512);   // 8 (two blocks per CU)
        run<1>("dependent", 256, 256);
        run<1>("dependent", 1024, 256);
        run<1>("dependent", 1024, 512);
        printf("\n");
    }
    return 0;
}
