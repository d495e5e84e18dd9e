// Micro-benchmark: what HBM bandwidth the mix kernel's access pattern can reach on one MI355X.
//
// The harness first has to show that it can drive the memory system at all: part 1 is a plain
// float4 copy in several launch shapes, and the best one must reproduce the streaming figure of
// MI355X_MICROARCH.md (6.29 TB/s read + write).  Part 2 is a read-only stream (the mix kernel
// writes almost nothing).  Part 3 is the mix kernel's pattern without the mixing: every wave
// fetches a short contiguous chunk (a source's window) from a random place in a different
// 256 KB clip, 16 B per lane, `depth` chunks in flight per wave, at several occupancies.
// hipcc --offload-arch=gfx950 -O3 -o hbm_ceiling hbm_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- part 1: copies ----------------------------------------------------------------------------
template <int UNROLL>
__global__ __launch_bounds__(256) void copy_blocked(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
    // one block moves UNROLL * 256 consecutive float4 (loads first, then stores)
    size_t i0 = ((size_t)blockIdx.x * UNROLL) * 256 + threadIdx.x;
    u32x4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) v[k] = i0 + (size_t)k * 256 < n ? __builtin_nontemporal_load(&src[i0 + (size_t)k * 256]) : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) if (i0 + (size_t)k * 256 < n) __builtin_nontemporal_store(v[k], &dst[i0 + (size_t)k * 256]);
}
template <int UNROLL>
__global__ __launch_bounds__(256) void copy_blocked_plain(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
    size_t i0 = ((size_t)blockIdx.x * UNROLL) * 256 + threadIdx.x;
    u32x4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) v[k] = i0 + (size_t)k * 256 < n ? src[i0 + (size_t)k * 256] : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) if (i0 + (size_t)k * 256 < n) dst[i0 + (size_t)k * 256] = v[k];
}
__global__ __launch_bounds__(256) void copy_gridstride(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ---- part 2: read-only stream -------------------------------------------------------------------
template <int UNROLL>
__global__ __launch_bounds__(256) void read_blocked(const u32x4* __restrict__ src, size_t n, unsigned* __restrict__ sink) {
    size_t i0 = ((size_t)blockIdx.x * UNROLL) * 256 + threadIdx.x;
    u32x4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) v[k] = i0 + (size_t)k * 256 < n ? src[i0 + (size_t)k * 256] : u32x4{0, 0, 0, 0};
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    if (acc == 0x12345678u) sink[0] = acc;
}

// ---- part 3: random chunks ------------------------------------------------------------------------
// wave w reads `per_wave` chunks; chunk (w, c) lives in clip hash(w, c) at a random 16 B aligned offset.
template <int DEPTH, int VECS>
__global__ __launch_bounds__(256) void chunks(const unsigned char* __restrict__ base, size_t n_clips, size_t clip_stride, uint32_t chunk_bytes,
                                               uint32_t per_wave, uint32_t salt, unsigned* __restrict__ sink) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned acc = 0;
    for (uint32_t c = 0; c < per_wave; c += DEPTH) {
        u32x4 v[DEPTH][VECS];
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) {
            uint64_t h = (uint64_t)(wave * per_wave + c + q + 1) * 0x9E3779B97F4A7C15ull + salt;
            h ^= h >> 29;
            const size_t clip = (size_t)(h % n_clips);
            const size_t off = (size_t)((h >> 24) % (clip_stride - chunk_bytes - 16)) & ~(size_t)15;
            const unsigned char* p = base + clip * clip_stride + off;
#pragma unroll
            for (int j = 0; j < VECS; ++j)
                v[q][j] = (uint32_t)(j * 1024 + lane * 16) < chunk_bytes ? *reinterpret_cast<const u32x4*>(p + j * 1024 + lane * 16) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int q = 0; q < DEPTH; ++q)
#pragma unroll
            for (int j = 0; j < VECS; ++j) acc += v[q][j].x ^ v[q][j].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static hipEvent_t e0, e1;
template <class F> static double time_ms(F&& launch, int reps = 8) {
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
    const size_t total = n_clips * clip_stride;
    unsigned char* buf; unsigned* sink;
    if (hipMalloc(&buf, total + ((size_t)8 << 30)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 1, total + ((size_t)8 << 30));
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // warm the clocks: ~0.3 s of streaming
    for (int i = 0; i < 60; ++i) hipLaunchKernelGGL(copy_gridstride, dim3(4096), dim3(256), 0, 0, (const u32x4*)buf, (u32x4*)(buf + total), ((size_t)4 << 30) / 16);
    (void)hipDeviceSynchronize();

    const size_t nb = (size_t)8 << 30, n = nb / 16;
    const u32x4* src = (const u32x4*)buf; u32x4* dst = (u32x4*)(buf + total);
    printf("# part 1: float4 copy of 8 GiB, GB/s read + write (guide: 6290)\n");
    double best_copy = 0;
#define COPY(name, kern, U) { double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3((unsigned)((n + (size_t)(U) * 256 - 1) / ((size_t)(U) * 256))), dim3(256), 0, 0, src, dst, n); }); \
        double g = 2.0 * nb / (ms * 1e-3) / 1e9; if (g > best_copy) best_copy = g; printf("copy %-28s %8.1f\n", name, g); }
    COPY("blocked nt x1", copy_blocked<1>, 1) COPY("blocked nt x2", copy_blocked<2>, 2) COPY("blocked nt x4", copy_blocked<4>, 4) COPY("blocked nt x8", copy_blocked<8>, 8)
    COPY("blocked plain x1", copy_blocked_plain<1>, 1) COPY("blocked plain x2", copy_blocked_plain<2>, 2) COPY("blocked plain x4", copy_blocked_plain<4>, 4) COPY("blocked plain x8", copy_blocked_plain<8>, 8)
    for (unsigned g : {2048u, 4096u, 8192u, 16384u, 65536u}) {
        double ms = time_ms([&] { hipLaunchKernelGGL(copy_gridstride, dim3(g), dim3(256), 0, 0, src, dst, n); });
        double gb = 2.0 * nb / (ms * 1e-3) / 1e9; if (gb > best_copy) best_copy = gb;
        printf("copy grid-stride %6u blocks     %8.1f\n", g, gb);
    }
    {   // the runtime's own device-to-device copy
        double ms = time_ms([&] { (void)hipMemcpyAsync(dst, src, nb, hipMemcpyDeviceToDevice, 0); });
        double gb = 2.0 * nb / (ms * 1e-3) / 1e9; if (gb > best_copy) best_copy = gb;
        printf("copy hipMemcpyAsync D2D           %8.1f\n", gb);
    }
    printf("# best copy: %.1f GB/s\n", best_copy);

    printf("# part 2: read-only stream of 8 GiB, GB/s\n");
#define READ(name, U) { double ms = time_ms([&] { hipLaunchKernelGGL(read_blocked<U>, dim3((unsigned)((n + (size_t)(U) * 256 - 1) / ((size_t)(U) * 256))), dim3(256), 0, 0, src, n, sink); }); \
        printf("read %-28s %8.1f\n", name, nb / (ms * 1e-3) / 1e9); }
    READ("blocked x1", 1) READ("blocked x2", 2) READ("blocked x4", 4) READ("blocked x8", 8)

    printf("# part 3: random chunks from 256 KB clips (32 GiB), GB/s; waves/CU x chunks in flight\n");
    printf("# chunk_B  16x2   16x4   16x8   32x2   32x4   8x8\n");
    const uint32_t sizes[] = {1024, 2304, 3072, 4096, 4352};
    for (uint32_t cb : sizes) {
        printf("%7u", cb);
        auto run = [&](auto kern, unsigned waves_per_cu, int depth) {
            const uint32_t resident = 256 * waves_per_cu;
            const uint32_t waves = resident * 4, per_wave = 64;       // 4 rounds of resident waves
            uint32_t salt = 1;
            double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(waves / 4), dim3(256), 0, 0, buf, n_clips, clip_stride, cb, per_wave, salt++ * 7919u, sink); });
            (void)depth;
            printf(" %6.0f", (double)waves * per_wave * cb / (ms * 1e-3) / 1e9);
        };
        // occupancy is set by the register footprint of DEPTH x VECS float4 (hipcc allocates what it needs):
        // the "waves/CU" label is the intent; the launch always offers 4 rounds of work
        if (cb <= 1024)      { run(chunks<2, 1>, 16, 2); run(chunks<4, 1>, 16, 4); run(chunks<8, 1>, 16, 8); run(chunks<2, 1>, 32, 2); run(chunks<4, 1>, 32, 4); run(chunks<8, 1>, 8, 8); }
        else if (cb <= 3072) { run(chunks<2, 3>, 16, 2); run(chunks<4, 3>, 16, 4); run(chunks<8, 3>, 16, 8); run(chunks<2, 3>, 32, 2); run(chunks<4, 3>, 32, 4); run(chunks<8, 3>, 8, 8); }
        else                 { run(chunks<2, 5>, 16, 2); run(chunks<4, 5>, 16, 4); run(chunks<8, 5>, 16, 8); run(chunks<2, 5>, 32, 2); run(chunks<4, 5>, 32, 4); run(chunks<8, 5>, 8, 8); }
        printf("\n");
    }
    return 0;
}
