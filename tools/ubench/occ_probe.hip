// Probe: how many workgroups of a given shape (threads, static LDS, VGPRs) does ONE launch really keep resident per CU
// on MI355X?  Every block records when it started; blocks that did not start in the first microseconds were queued
// behind resident ones.  (ordered_fused's 384-thread / 59 904-byte workgroups: the occupancy API says 2 per CU.)
// hipcc --offload-arch=gfx950 -O3 -o occ_probe occ_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

template <int THREADS, int LDS, int VG>
__global__ __launch_bounds__(THREADS) void k(unsigned long long* rec, float* sink, int spin_us) {
    __shared__ unsigned char lds[LDS];
    const unsigned long long t0 = wall_clock64();
    float r[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) r[i] = (float)(threadIdx.x + i);
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    while (wall_clock64() - t0 < (unsigned long long)spin_us * 100ull) {
#pragma unroll
        for (int i = 0; i < VG; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) % VG]));
    }
    float s = (float)lds[(threadIdx.x * 7) % THREADS];
#pragma unroll
    for (int i = 0; i < VG; ++i) s += r[i];
    if (threadIdx.x == 0) {
        rec[2 * blockIdx.x] = t0;
        rec[2 * blockIdx.x + 1] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg(63492);
    }
    sink[(size_t)blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int THREADS, int LDS, int VG> void run(int blocks) {
    unsigned long long* rec; float* sink;
    (void)hipHostMalloc((void**)&rec, blocks * 16, 0);
    (void)hipMalloc(&sink, (size_t)blocks * THREADS * 4);
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&k<THREADS, LDS, VG>), THREADS, 0);
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k<THREADS, LDS, VG>));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<THREADS, LDS, VG>), dim3(blocks), dim3(THREADS), 0, 0, rec, sink, 200);
        (void)hipDeviceSynchronize();
    }
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < blocks; ++b) t0 = std::min(t0, rec[2 * b]);
    int early = 0; std::map<unsigned, int> per_cu;
    for (int b = 0; b < blocks; ++b) {
        if ((rec[2 * b] - t0) * 0.01 > 50.0) continue;
        early++;
        const unsigned hw = (unsigned)rec[2 * b + 1], xcc = (unsigned)(rec[2 * b + 1] >> 32) & 15u;
        per_cu[(xcc << 16) | (hw & 0xff00u)]++;
    }
    int mx = 0; for (auto& kv : per_cu) mx = std::max(mx, kv.second);
    printf("threads %4d  lds %6d B  vgprs %3d | occupancy API %d/CU | %d blocks launched: %d resident at once on %zu CUs (max %d on one CU)\n",
           THREADS, LDS, fa.numRegs, occ, blocks, early, per_cu.size(), mx);
    (void)hipHostFree(rec); (void)hipFree(sink);
}

int main() {
    run<384, 59904, 8>(768);
    run<384, 59904, 90>(768);
    run<384, 1024, 90>(1536);
    run<384, 1024, 8>(2048);
    run<320, 49920, 90>(768);
    run<256, 39936, 90>(1024);
    run<192, 29952, 90>(1536);
    run<128, 19968, 90>(2304);
    run<128, 19968, 120>(2304);
    run<384, 39936, 90>(1024);
    return 0;
}
