// Micro-benchmark: what HBM delivers for buffered_write's access pattern and nothing else -- every wave, per "source", reads one
// contiguous chunk (a leaf window: 4.5 KB) from a random place of a 64-GiB region and writes one contiguous chunk (the ring
// stretch: 4 KB) to a random place of a 20-GiB region; `depth` reads are in flight per wave, `waves_per_cu` waves resident.
// No arithmetic, no LDS: the bandwidth ceiling of the pattern.  Prints read + write GB/s.  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(64) void k(const unsigned char* __restrict__ rbase, size_t r_bytes, unsigned char* __restrict__ wbase, size_t w_bytes,
                                       uint32_t read_bytes, uint32_t write_bytes, uint32_t per_wave, uint32_t salt) {
    const uint32_t wave = blockIdx.x, lane = threadIdx.x;
    u32x4 v[DEPTH][5];
    auto place = [&](uint32_t c, size_t span, uint32_t len) {
        uint64_t h = (uint64_t)(wave * per_wave + c + 1) * 0x9E3779B97F4A7C15ull + salt;
        h ^= h >> 29;
        return (size_t)(h % (span - len - 16)) & ~(size_t)15;
    };
    auto load = [&](int q, uint32_t c) {
        const unsigned char* p = rbase + place(c, r_bytes, read_bytes);
#pragma unroll
        for (int j = 0; j < 5; ++j)
            v[q][j] = ((uint32_t)(j * 1024 + lane * 16) < read_bytes) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + j * 1024 + lane * 16)) : u32x4{0, 0, 0, 0};
    };
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) load(q, (uint32_t)q);
    for (uint32_t c = 0; c < per_wave; c += DEPTH) {
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) {
            unsigned char* w = wbase + place(c + q + 0x40000000u, w_bytes, write_bytes);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((uint32_t)(j * 1024 + lane * 16) < write_bytes) *reinterpret_cast<u32x4*>(w + j * 1024 + lane * 16) = v[q][j] ^ v[q][4];
            if (c + DEPTH + q < per_wave) load(q, c + DEPTH + q);
        }
    }
}

int main() {
    const size_t r_bytes = (size_t)64 << 30, w_bytes = (size_t)20 << 30;
    unsigned char *rb, *wb;
    if (hipMalloc(&rb, r_bytes) != hipSuccess || hipMalloc(&wb, w_bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(rb, 1, r_bytes);
    (void)hipMemset(wb, 0, w_bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t read_bytes = 4608, write_bytes = 4096, total_chunks = 262144;
    printf("# waves/CU depth  GB/s(read+write)  ms per 262144 chunk pairs (%u B read + %u B written each)\n", read_bytes, write_bytes);
    for (uint32_t wpc : {4u, 7u, 8u, 12u, 16u}) {
        for (int depth : {1, 2}) {
            const uint32_t waves = 256 * wpc, per_wave = (total_chunks + waves - 1) / waves;
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                (void)hipEventRecord(e0);
                if (depth == 1) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, rb, r_bytes, wb, w_bytes, read_bytes, write_bytes, per_wave, (uint32_t)rep * 7919u);
                else hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, rb, r_bytes, wb, w_bytes, read_bytes, write_bytes, per_wave, (uint32_t)rep * 7919u);
                (void)hipEventRecord(e1);
                (void)hipDeviceSynchronize();
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2 && ms < best) best = ms;
            }
            const double bytes = (double)waves * per_wave * (read_bytes + write_bytes);
            printf("%8u %5d %10.1f %10.4f\n", wpc, depth, bytes / (best * 1e-3) / 1e9, best * (double)total_chunks / ((double)waves * per_wave));
        }
    }
    return 0;
}
