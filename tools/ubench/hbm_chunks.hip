// Micro-benchmark: HBM bandwidth for the mix kernel's access pattern -- every wave reads a short
// contiguous chunk (a source's window) from a different 256 KB-strided clip -- against the streaming
// copy figure.  Each wave reads `per_wave` chunks of `chunk_bytes` (16 B per lane per load, lanes
// beyond the chunk idle), four chunks in flight.  Prints GB/s per chunk size.  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ base, size_t n_clips, size_t clip_stride, uint32_t chunk_bytes,
                                          uint32_t per_wave, uint32_t salt, unsigned int* __restrict__ sink) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t vecs = (chunk_bytes + 1023) / 1024;          // b128 loads per lane per chunk
    unsigned int acc = 0;
    for (uint32_t c = 0; c < per_wave; c += 4) {
        u32x4 v[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // pseudo-random clip and offset inside it, different for every (wave, chunk)
            uint64_t h = (uint64_t)(wave * per_wave + c + q + 1) * 0x9E3779B97F4A7C15ull + salt;
            h ^= h >> 29;
            const size_t clip = (size_t)(h % n_clips);
            const size_t off = (size_t)((h >> 24) % (clip_stride - chunk_bytes - 16)) & ~(size_t)15;
            const unsigned char* p = base + clip * clip_stride + off;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((uint32_t)j < vecs && (uint32_t)(j * 1024 + lane * 16) < chunk_bytes)
                    v[q][j] = *reinterpret_cast<const u32x4*>(p + j * 1024 + lane * 16);
                else
                    v[q][j] = u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += v[q][j].x ^ v[q][j].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void copy_k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
    const size_t clip_stride = 256 * 1024, n_clips = 131072;    // 32 GiB
    unsigned char* buf; unsigned int* sink;
    if (hipMalloc(&buf, n_clips * clip_stride) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 1, n_clips * clip_stride);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("# chunk_bytes GB/s   (4096 waves resident = 16/CU, 4 chunks in flight per wave)\n");
    const uint32_t sizes[] = {512, 1024, 2304, 3072, 4096};
    for (int pass = 0; pass < 3; ++pass)
    for (uint32_t cb : sizes) {
        const uint32_t vecs = (cb + 1023) / 1024;
        if (vecs > 4) { continue; }
        const uint32_t waves = 4096 * 4, per_wave = 64;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(waves / 4), dim3(256), 0, 0, buf, n_clips, clip_stride, cb, per_wave, (uint32_t)rep * 7919u, sink);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2 && pass == 2) printf("%6u %8.1f\n", cb, (double)waves * per_wave * cb / (ms * 1e-3) / 1e9);
        }
    }
    {   // streaming copy of 8 GiB (read + write counted)
        const size_t n = (size_t)8 << 30;
        for (int rep = 0; rep < 12; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(copy_k, dim3(256 * 16), dim3(256), 0, 0, (const u32x4*)buf, (u32x4*)(buf + ((size_t)16 << 30)), n / 16);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 11) printf("# float4 copy: %.1f GB/s (read + write)\n", 2.0 * n / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
