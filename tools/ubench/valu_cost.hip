// Micro-benchmark: issue cost (cycles per wave64 instruction, one wave on one SIMD, and with 2/4
// waves per SIMD) of the VALU/LDS instructions the mix kernel is made of.  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 20000
#define REP 32
template <int OP>
__global__ void k(float* out, long long* cyc, float a, float b) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float x0 = a + threadIdx.x, x1 = b + threadIdx.x, x2 = a * 2 + threadIdx.x, x3 = b * 3 + threadIdx.x;
    float y0 = a, y1 = b, y2 = a, y3 = b;
    int i0 = threadIdx.x, i1 = threadIdx.x * 3, i2 = 5, i3 = 7;
    long long t0 = __builtin_readcyclecounter();
    t0 = clock64();
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(y0)); }                      // dependent chain
            if (OP == 1) { asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0)); }
            if (OP == 2) { asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(*(double*)&x0), "+v"(*(double*)&x2) : "v"(*(double*)&y0)); }
            if (OP == 3) { asm volatile("v_cvt_i32_f32 %0, %4\n v_cvt_i32_f32 %1, %5\n v_cvt_i32_f32 %2, %6\n v_cvt_i32_f32 %3, %7" : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3)); }
            if (OP == 4) { asm volatile("v_fract_f32 %0, %4\n v_fract_f32 %1, %5\n v_fract_f32 %2, %6\n v_fract_f32 %3, %7" : "=v"(y0), "=v"(y1), "=v"(y2), "=v"(y3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3)); }
            if (OP == 5) { asm volatile("v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0)); }
            if (OP == 6) { asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0)); }
            if (OP == 7) { asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0)); }
            if (OP == 8) { asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7" : "=v"(y0), "=v"(y1), "=v"(y2), "=v"(y3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3)); }
            if (OP == 9) { // ds_read_b64 conflict-free (lane*8 bytes), 4 independent
                double d0, d1, d2, d3; int ad = (threadIdx.x & 63) * 8;
                asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n s_waitcnt lgkmcnt(0)" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(ad));
                x0 += (float)d0; }
            if (OP == 10) { // ds_read_b64 stride 16 samples dual-copy-like: lane*64 bytes
                double d0, d1, d2, d3; int ad = (threadIdx.x & 63) * 64;
                asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n s_waitcnt lgkmcnt(0)" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(ad));
                x0 += (float)d0; }
            if (OP == 11) { // ds_read_b64 stride 17.1 samples (68 B -> rounded to 8): (lane*68)&~7
                double d0, d1, d2, d3; int ad = ((threadIdx.x & 63) * 68) & ~7;
                asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n s_waitcnt lgkmcnt(0)" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(ad));
                x0 += (float)d0; }
            if (OP == 12) { // ds_read2_b32 conflict-free stride 17 dwords
                double d0, d1, d2, d3; int ad = (threadIdx.x & 63) * 68;
                asm volatile("ds_read2_b32 %0, %4 offset1:1\n ds_read2_b32 %1, %4 offset0:2 offset1:3\n ds_read2_b32 %2, %4 offset0:4 offset1:5\n ds_read2_b32 %3, %4 offset0:6 offset1:7\n s_waitcnt lgkmcnt(0)" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(ad));
                x0 += (float)d0; }
            if (OP == 13) { asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(*(double*)&x0), "+v"(*(double*)&x2) : "v"(*(double*)&y0)); }
            if (OP == 14) { asm volatile("v_sub_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0)); }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3 + (float)(i0 + i1 + i2 + i3);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, int per_iter_instrs, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, 4096 * sizeof(float)); hipMalloc(&cyc, 8);
    k<OP><<<1, threads>>>(out, cyc, 1.0f, 1.0001f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<OP><<<1, threads>>>(out, cyc, 1.0f, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    if (OP == 1 && threads == 1024) printf("   [calibration] %lld ticks in %.4f ms (incl. launch) -> >= %.1f MHz tick\n", c, ms, c / (ms * 1e3));
    double per = (double)c / (N_ITER * REP * per_iter_instrs);
    printf("%-44s threads=%4d  cycles/instr(per wave)=%.2f  -> per-instr SIMD occupancy=%.2f (with %d waves/SIMD)\n", name, threads, per, per / (threads / 256.0 > 1 ? threads / 256.0 : 1), threads / 256 > 1 ? threads / 256 : 1);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int threads : {64, 256, 512, 1024}) {
        run<0>("v_add_f32 dependent chain", 1, threads);
        run<1>("v_add_f32 x4 independent", 4, threads);
        run<7>("v_mul_f32 x4 independent", 4, threads);
        run<14>("sub/mul/add/mul mix x4", 4, threads);
        run<2>("v_pk_add_f32 x2 independent", 2, threads);
        run<13>("v_pk_mul_f32 x2 independent", 2, threads);
        run<3>("v_cvt_i32_f32 x4", 4, threads);
        run<4>("v_fract_f32 x4", 4, threads);
        run<5>("v_lshl_add_u32 x4", 4, threads);
        run<6>("v_mad_u32_u24 x4", 4, threads);
        run<8>("v_mov_b32 x4", 4, threads);
        run<9>("ds_read_b64 x4 conflict-free + wait", 4, threads);
        run<10>("ds_read_b64 x4 stride 64B + wait", 4, threads);
        run<11>("ds_read_b64 x4 stride ~68B + wait", 4, threads);
        run<12>("ds_read2_b32 x4 stride 17dw + wait", 4, threads);
        printf("\n");
    }
    return 0;
}
