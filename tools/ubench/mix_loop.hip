// Micro-benchmark: the per-sample loop of the mix kernels (kernels.h mix_source_lds) alone -- every SIMD of the chip at the
// kernels' occupancy (8 workgroups of two waves per CU), windows already in LDS, no DMA, no cursor scan, no barrier, no
// records -- and cut-down copies of it, to see what the loop's floor is and which instructions set it.
// One "source" = 16 samples per lane = 1024 frames of one ear per wave, as in spatial_mix_pair.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -o mix_loop mix_loop.hip
#include "../../oddio_amd/csrc/pair_kernels.h"
#include <cstdio>
#include <vector>
using namespace oddio_hip;

// VAR 0: mix_source_lds as spatial_mix_pair's common variant instantiates it.
// VAR 1..: hand-written copies of the same loop with parts left out:
//   1 everything (should equal 0)      2 no LDS read (a, b from registers)     3 no gain ramp op (constant gain)
//   4 no cursor ops (tr, fr constant)   5 lerp + accumulate only                6 LDS read only (+ one add per sample)
template <int VAR>
__global__ __launch_bounds__(128, MIX_WAVES_PER_SIMD) void k(float* out, int n_src, float ds, float dg, float g0) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[PAIR_LDS_TOTAL];
    float* w = reinterpret_cast<float*>(smem);
    for (int i = threadIdx.x; i < PAIR_LDS_TOTAL / 4; i += 128) w[i] = (float)(i & 255) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int cB = lane >> 4, bB = lane & 15;
    const uint32_t frame0 = 16u * (uint32_t)lane;
    float acc[16], fi[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.0f; fi[i] = (float)frame0 + (float)i; }
    const int wrel4 = 4 * (cB * 262);
    float x0 = (float)(16 * bB) * ds + 0.25f;
    for (int s = 0; s < n_src; ++s) {
        asm volatile("" : "+v"(x0), "+v"(ds), "+v"(dg), "+v"(g0));
        const unsigned char* win_bytes = smem + (s & 1) * PAIR_WIN_BYTES;
        if (VAR == 0) {
            mix_source_lds<true, false, true, false, true, false, false, PAIR_WIN_CAP, false, true>(win_bytes, wrel4, x0, bB, 0, 0.0f, acc, fi, frame0, 1024u, 1.0f,
                                                                                                    g0, dg, ds, 1060, nullptr);
        } else {
            const float* wbase = reinterpret_cast<const float*>(win_bytes + wrel4);
            const float gbase = __builtin_fmaf(fi[0], dg, g0);
            float x = x0;
            float a[16], bb[16], fr[16];
#define ISSUE(I)                                                                                          \
    {                                                                                                     \
        int tr = 16 * bB + (I);                                                                           \
        if (VAR != 4 && VAR != 5 && VAR != 6) { tr = (int)x; fr[I] = __builtin_amdgcn_fractf(x); x = x + ds; } \
        else fr[I] = ds;                                                                                  \
        if (VAR == 2) { a[I] = x; bb[I] = fr[I]; }                                                        \
        else { a[I] = wbase[tr]; bb[I] = wbase[tr + 1]; }                                                 \
    }
#pragma unroll
            for (int i = 0; i < MIX_DEPTH; ++i) ISSUE(i)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i + MIX_DEPTH < 16) ISSUE(i + MIX_DEPTH)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" : "+v"(dg));
                if (VAR == 6) { asm("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a[i] + bb[i])); }
                else {
                    const float v = __builtin_fmaf(fr[i], bb[i] - a[i], a[i]);
                    const float g = (VAR == 3 || VAR == 5) ? g0 : __builtin_fmaf((float)i, dg, gbase);
                    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(v), "v"(g));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef ISSUE
        }
    }
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += acc[i];
    out[(size_t)blockIdx.x * 128 + threadIdx.x] = t;
}

template <int VAR> void run(const char* name, int wgs, int n_src) {
    float* out;
    (void)hipMalloc(&out, (size_t)wgs * 128 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<VAR>, dim3(wgs), dim3(128), 0, 0, out, n_src, 1.02f, 1e-7f, 0.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    // the headline callback: 262 144 sources x 2 ears = 524 288 wave-sources
    const double wave_src = (double)wgs * 2 * n_src;
    printf("%-46s wgs=%5d src/wave=%4d | %.4f ms | = %.4f ms per 524288 wave-sources | %.1f cycles/sample/SIMD at 2.2 GHz\n", name, wgs, n_src, best,
           best * 524288.0 / wave_src, best * 1e-3 * 2.2e9 / (wave_src * 16.0 / 1024.0));
    (void)hipFree(out);
}

int main() {
    for (int pass = 0; pass < 2; ++pass) {
        const int wgs = 2048, n = 512;      // 8 workgroups per CU; 4x the headline callback's work per launch
        run<0>("mix_source_lds (pair kernel's common variant)", wgs, n);
        run<1>("hand copy, everything", wgs, n);
        run<2>("  no LDS read", wgs, n);
        run<3>("  no gain-ramp op", wgs, n);
        run<4>("  no cursor ops (cvt, fract, add)", wgs, n);
        run<5>("  lerp + accumulate only (LDS read kept)", wgs, n);
        run<6>("  LDS read + one add", wgs, n);
        run<0>("mix_source_lds, 4 workgroups per CU", 1024, n);
        run<0>("mix_source_lds, 2 workgroups per CU", 512, n);
        printf("\n");
    }
    return 0;
}
