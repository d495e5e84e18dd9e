// Micro-benchmark: memory-level parallelism of the mix kernel's access pattern.  spatial_mix keeps ONE window (2.3 KB) in
// flight per wave while it renders the previous one, 16 waves per CU (LDS: two window buffers + the stream blocks per wave).
// This models that loop -- fetch a random 2304-byte chunk per "source", `busy` dependent FMAs of rendering per source -- with
// DEPTH chunks in flight per wave and the occupancy forced through the workgroup's LDS size, to see what a third window
// buffer (2 in flight, 12 waves per CU) would buy before building it.
// hipcc --offload-arch=gfx950 -O3 -o hbm_mlp hbm_mlp.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(128) void mlp(const unsigned char* __restrict__ base, size_t n_clips, size_t clip_stride, uint32_t per_wave, int busy,
                                           uint32_t salt, float* __restrict__ sink) {
    extern __shared__ unsigned char lds[];
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    auto addr = [&](uint32_t c) {
        uint64_t h = (uint64_t)(wave * per_wave + c + 1) * 0x9E3779B97F4A7C15ull + salt;
        h ^= h >> 29;
        const size_t clip = (size_t)(h % n_clips);
        const size_t off = (size_t)((h >> 24) % (clip_stride - 2304 - 16)) & ~(size_t)15;
        return base + clip * clip_stride + off;
    };
    u32x4 v[DEPTH][3];
    float acc = (float)lane;
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) {
        const unsigned char* p = addr(q);
#pragma unroll
        for (int j = 0; j < 3; ++j) v[q][j] = (j * 1024 + lane * 16) < 2304u ? *reinterpret_cast<const u32x4*>(p + j * 1024 + lane * 16) : u32x4{0, 0, 0, 0};
    }
    for (uint32_t c = 0; c < per_wave; c += DEPTH) {
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) {
            // consume chunk c + q (waits for it), start chunk c + q + DEPTH into the same registers, "render"
            float x = __uint_as_float((v[q][0].x ^ v[q][1].y ^ v[q][2].z) & 0x007fffffu | 0x3f800000u);
            if (c + q + DEPTH < per_wave) {
                const unsigned char* p = addr(c + q + DEPTH);
#pragma unroll
                for (int j = 0; j < 3; ++j) v[q][j] = (j * 1024 + lane * 16) < 2304u ? *reinterpret_cast<const u32x4*>(p + j * 1024 + lane * 16) : u32x4{0, 0, 0, 0};
            }
            for (int k = 0; k < busy; ++k) acc = __builtin_fmaf(acc, 0.999f, x);
        }
    }
    if (acc == 0.12345f) { sink[0] = acc; lds[threadIdx.x] = 1; }
}

static hipEvent_t e0, e1;
template <class F> static double time_ms(F&& launch, int reps = 6) {
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2 && ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t clip_stride = 256 * 1024, n_clips = 131072;     // 32 GiB of "clips"
    unsigned char* buf; float* sink;
    if (hipMalloc(&buf, n_clips * clip_stride) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 1, n_clips * clip_stride);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("# random 2304-byte chunks, GB/s; rows: dependent FMAs of 'rendering' per chunk; columns: waves per CU x chunks in flight per wave\n");
    printf("# busy   16x1   16x2   12x1   12x2   12x3    8x2    8x4\n");
    for (int busy : {0, 300, 600, 900}) {
        printf("%6d", busy);
        auto run = [&](auto kern, unsigned waves_per_cu) {
            const size_t lds = 160 * 1024 / (waves_per_cu / 2) - 512;       // bytes per 2-wave workgroup: exactly waves_per_cu / 2 fit a CU
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const uint32_t waves = 256 * waves_per_cu * 2, per_wave = 96;    // two rounds of resident waves
            uint32_t salt = 1;
            double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(waves / 2), dim3(128), lds, 0, buf, n_clips, clip_stride, per_wave, busy, salt++ * 7919u, sink); });
            printf(" %6.0f", (double)waves * per_wave * 2304 / (ms * 1e-3) / 1e9);
        };
        run(mlp<1>, 16); run(mlp<2>, 16); run(mlp<1>, 12); run(mlp<2>, 12); run(mlp<3>, 12); run(mlp<2>, 8); run(mlp<4>, 8);
        printf("\n");
    }
    return 0;
}
