// Micro-benchmark: the cost of ONE wave's chain of dependent v_add_f32_dpp (quad_perm broadcast operand, running sum as
// the plain operand) -- the inner sequence of ordered_sum -- by what separates consecutive adds, and whether the
// result is the exact sequential sum.  One wave per CU on 128 CUs (ordered_sum's geometry).
// hipcc --offload-arch=gfx950 -O3 -o dpp_chain dpp_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define N_ITER 8192
#define A4(V, SEP) \
    "v_add_f32_dpp %0, " V ", %0 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n" SEP \
    "v_add_f32_dpp %0, " V ", %0 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n" SEP \
    "v_add_f32_dpp %0, " V ", %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n" SEP \
    "v_add_f32_dpp %0, " V ", %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n" SEP
#define P4(V, SEP) \
    "v_add_f32 %0, " V ", %0\n" SEP "v_add_f32 %0, " V ", %0\n" SEP "v_add_f32 %0, " V ", %0\n" SEP "v_add_f32 %0, " V ", %0\n" SEP

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* ticks, const float* in) {
    float s = 0.0f;
    const float v0 = in[threadIdx.x], v1 = in[64 + threadIdx.x];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N_ITER; ++it) {
        if (MODE == 0) asm volatile(A4("%1", "") A4("%2", "") : "+v"(s) : "v"(v0), "v"(v1));
        else if (MODE == 1) asm volatile(A4("%1", "s_nop 0\n") A4("%2", "s_nop 0\n") : "+v"(s) : "v"(v0), "v"(v1));
        else if (MODE == 2) asm volatile(A4("%1", "s_nop 1\n") A4("%2", "s_nop 1\n") : "+v"(s) : "v"(v0), "v"(v1));
        else if (MODE == 3) asm volatile(P4("%1", "") P4("%2", "") : "+v"(s) : "v"(v0), "v"(v1));          // plain adds, no DPP
        else asm volatile(P4("%1", "s_nop 0\n") P4("%2", "s_nop 0\n") : "+v"(s) : "v"(v0), "v"(v1));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE> void run(const char* name, const float* d_in, const std::vector<float>& h_in) {
    float* out; unsigned long long* ticks;
    const int blocks = 128;
    (void)hipMalloc(&out, (size_t)blocks * 64 * sizeof(float)); (void)hipMalloc(&ticks, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, ticks, d_in);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    std::vector<float> h(64); (void)hipMemcpy(h.data(), out, 64 * 4, hipMemcpyDeviceToHost);
    // expected: DPP modes add the quad's lanes 3,2,1,0 of v0 then of v1; plain modes add the lane's own v0 x4, v1 x4
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        float s = 0.0f;
        for (int it = 0; it < N_ITER; ++it)
            for (int r = 0; r < 2; ++r)
                for (int m = 3; m >= 0; --m) {
                    volatile float x = (MODE <= 2) ? h_in[r * 64 + (l & ~3) + m] : h_in[r * 64 + l];
                    volatile float y = s + x;
                    s = y;
                }
        if (memcmp(&s, &h[l], 4) != 0) bad++;
    }
    const double adds = (double)N_ITER * 8;
    printf("%-22s wall %.3f ms | %.2f ns/add | %.2f s_memtime ticks/add | lanes with a wrong sum: %d\n", name, ms, ms * 1e6 / adds, (double)t / adds, bad);
    (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
    std::vector<float> h(128);
    for (int i = 0; i < 128; ++i) h[i] = 0.001f * (float)((i * 37) % 101) - 0.03f;
    float* d; (void)hipMalloc(&d, 512); (void)hipMemcpy(d, h.data(), 512, hipMemcpyHostToDevice);
    run<0>("dpp back to back", d, h);
    run<1>("dpp + s_nop 0", d, h);
    run<2>("dpp + s_nop 1", d, h);
    run<3>("plain back to back", d, h);
    run<4>("plain + s_nop 0", d, h);
    return 0;
}
