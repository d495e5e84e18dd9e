// Probe: does `buffer_load_dwordx4 ... lds` (the mix kernel's window staging, kernels.h window_desc / window_dma)
// zero-fill exactly the lanes that fall outside the descriptor -- windows that start before the clip (negative
// offsets), that run past its end, and a record count that cuts a 16-byte group in half?  Uses the product's own
// window_desc() / window_dma() / window_wait().  Prints one line per case; exit code 1 on any mismatch.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o dma_probe dma_probe.hip
#include "../../oddio_amd/csrc/kernels.h"
#include <cstdio>
#include <vector>
using namespace oddio_hip;

__global__ __launch_bounds__(64) void probe(const float* clip, int clip_len4, int ws, int nvec, int cut_bytes, float* out) {
    __shared__ __attribute__((aligned(16))) float win[WIN_CAP];
    const int lane = threadIdx.x;
    for (int k = lane; k < WIN_CAP; k += 64) win[k] = -1.0f;          // stale data the DMA must overwrite (also with zeros)
    wave_sync();
    int4 desc = window_desc(clip, clip_len4, ws, nvec);
    if (cut_bytes) desc.z -= cut_bytes;                                // a record count that is not a multiple of 16
    // what make_tile_rec packs into the TileRec: descriptor words 0-2, info = nvec << 8 | negvec << 16
    const int negvec = desc.z > 0 ? ((-desc.w) >> 4) : 0;
    const uint32_t info = ((uint32_t)nvec << 8) | ((uint32_t)negvec << 16);
    window_dma((uint32_t)(uintptr_t)win, (uint32_t)__builtin_amdgcn_readfirstlane(desc.x), (uint32_t)__builtin_amdgcn_readfirstlane(desc.y),
               (uint32_t)__builtin_amdgcn_readfirstlane(desc.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)info), 16 * lane);
    window_wait();
    wave_sync();
    for (int k = lane; k < WIN_CAP; k += 64) out[k] = win[k];
}

int main() {
    const int L = 1000, L4 = (L + 3) & ~3;                             // clip of 1000 samples, padded like Frames::from_slice
    std::vector<float> h(L4, 0.0f);
    for (int i = 0; i < L; ++i) h[i] = (float)(i + 1);
    float *d_clip, *d_out;
    (void)hipMalloc(&d_clip, (L4 + 4096) * sizeof(float));             // the bytes after the clip are poison: reading them is a bug
    std::vector<float> poison(L4 + 4096, 12345.0f);
    (void)hipMemcpy(d_clip, poison.data(), poison.size() * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_clip, h.data(), L4 * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMalloc(&d_out, WIN_CAP * sizeof(float));
    struct Case { int ws, nvec, cut; const char* what; };
    const Case cases[] = {
        {0, 150, 0, "inside the clip, three DMA pieces"},
        {100, 64, 0, "inside the clip, one piece"},
        {-8, 40, 0, "starts 8 samples before the clip"},
        {-300, 152, 0, "starts 300 samples before the clip"},
        {-700, 100, 0, "ends before the clip starts"},
        {900, 100, 0, "runs past the clip's end"},
        {1200, 30, 0, "starts after the clip's end"},
        {96, 32, 8, "record count cut in the middle of a 16-byte group"},
        {-604, 39, 0, "short window far before the clip (fuzz seed 5094)"},
        {-604, 20, 0, "shorter"},
        {-16, 39, 0, "window starting 16 samples before the clip"},
        {-4, 39, 0, "window starting 4 samples before the clip"},
    };
    int bad_total = 0;
    for (const Case& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_clip, L4, c.ws, c.nvec, c.cut, d_out);
        std::vector<float> o(WIN_CAP);
        (void)hipMemcpy(o.data(), d_out, WIN_CAP * sizeof(float), hipMemcpyDeviceToHost);
        int bad = 0;
        const int ws_pos = c.ws > 0 ? c.ws : 0;
        long long rec = (long long)(L4 - ws_pos) * 4, wend = (long long)(c.ws < 0 ? c.ws : 0) * 4 + (long long)c.nvec * 16;
        if (rec > wend) rec = wend;
        if (rec < 0) rec = 0;
        rec -= c.cut;
        for (int k = 0; k < 4 * c.nvec && k < WIN_CAP; ++k) {
            const int s = c.ws + k;                                    // sample index the window slot stands for
            const long long byte = (long long)(s - ws_pos) * 4;       // offset inside the descriptor
            float want = 0.0f;
            if (s >= 0 && byte >= 0 && byte + 4 <= rec) want = s < L ? (float)(s + 1) : 0.0f;
            if (o[k] != want) { if (bad < 4) printf("    slot %d (sample %d): got %g want %g\n", k, s, o[k], want); ++bad; }
        }
        printf("%-55s ws %5d nvec %3d: %s\n", c.what, c.ws, c.nvec, bad ? "MISMATCH" : "ok (zeros exactly outside the descriptor)");
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
