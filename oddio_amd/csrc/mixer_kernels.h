// mixer_kernels.h -- gfx950 kernels for Mixer<[f32;2]> of MonoToStereo<mono source>
// (src/mixer.rs:92-119, src/signal.rs:61-91).  Same exact-cursor scheme as kernels.h, with the
// Mixer's 1024-frame staging chunk (mixer.rs:77) as the unit: one tile == one chunk.
#pragma once
#include "kernels.h"
#include "buffered_kernels.h"
#include "buffered_fast.h"

#ifndef ODDIO_MIXU_POLICY
#define ODDIO_MIXU_POLICY 0      // cache policy of mixer_mix_unit's clip loads (buffer-load aux bits: 2 = nt); measured in round 6
#endif
namespace oddio_hip {

// one thread per slot: stop / finished scan (mixer.rs:100-106) and cursor bookkeeping
__global__ __launch_bounds__(256) void mixer_prepass(uint32_t n_sources, uint32_t n_frames, float interval,
                                                     const MixStatic* __restrict__ st, MixDyn* __restrict__ dyn,
                                                     MixParams* __restrict__ par, uint32_t* __restrict__ stopped_hdr,
                                                     uint32_t stopped_cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sources) return;
    MixDyn d = dyn[i];
    const MixStatic s = st[i];
    MixParams p = {};
    if (d.flags & MIXDYN_STOPPED) { p.flags = EAR_SKIP; par[i] = p; return; }
    bool fin = (d.flags & MIXDYN_STOP_REQUESTED) != 0;
    if (s.kind == KIND_FRAMES) fin = fin || d.t >= (double)(s.clip_len - 1u) / (double)s.clip_rate;   // frames.rs:204-206
    if (fin) {
        d.flags |= MIXDYN_STOPPED;
        const uint32_t k = atomicAdd(&stopped_hdr[0], 1u);
        if (k < stopped_cap) stopped_hdr[1 + k] = d.id;
        p.flags = EAR_SKIP;
        par[i] = p;
        dyn[i] = d;
        return;
    }
    p.t_start = d.t;
    p.phase_start = d.phase;
    for (uint32_t done = 0; done < n_frames; done += 1024u) {
        const uint32_t len = (n_frames - done) < 1024u ? (n_frames - done) : 1024u;
        if (s.kind == KIND_FRAMES) d.t = d.t + (double)interval * (double)len;                          // frames.rs:198
        else if (s.kind == KIND_SINE) d.phase = fmodf(d.phase + (interval * (float)len) * s.freq_or_value, ODDIO_TAU);   // sine.rs:39
    }
    par[i] = p;
    dyn[i] = d;
}

constexpr int MIXER_GROUP = 64;
constexpr int MIXER_TILE = 1024;               // one tile == one Mixer staging chunk (mixer.rs:77)
constexpr int MIXER_WIN_CAP = 1536;           // padded single-copy window (one pad float per 16 samples)
constexpr int MIXER_WIN_PAD = MIXER_WIN_CAP + MIXER_WIN_CAP / 16 + 16;
constexpr int MIXER_CKPT_STRIDE = 65;
constexpr int MIXER_WIN_VECS = (MIXER_WIN_CAP / 4 + 63) / 64;   // 16-byte vectors per lane that cover the largest staged window

constexpr int MIXER_SUB = 16;                  // sources whose cursor checkpoints are in LDS at a time (round 4: 64 -> 16.6 KB of the 24.7 a wave
                                               // took, six waves per CU; with 16 a wave takes 12.3 KB and the registers decide: twelve)
struct MixerLds {
    float ckpt[MIXER_SUB * MIXER_CKPT_STRIDE];   // [source of the batch][64 checkpoints]
    int cinfo[64 * 4];                    // per source: {wrel, frac bits, fast, path}
    int sinfo[64 * 2];                    // per source: {ws, count}
    float win[MIXER_WIN_PAD];
};

// `groups_per_wave`: groups of 64 sources a wave walks; with 2^k in its top byte instead, 2^k waves share ONE group and each renders
// 64 >> k of its sources (small mixers: a wave renders its sources one after the other, ~2.5 us each -- 64 of them were 170 us
// per callback however few sources the mixer held; round 4).
// STORE (ORDERED mode above the serial threshold, round 4): instead of accumulating, every source's contribution is written to its
// rows in the layout ordered_sum reads (kernels.h: [group of 16 sources][ear][16-frame column block][source][16 frames]; MonoToStereo:
// the same row for both ears), and ordered_sum adds them in the reference's order (mixer.rs:100-117, reverse slot order).
// TRACK (ODDIO_HIP_MODE_TRACKED, the scene's mode of pair_kernels.h on a Mixer): 2 = the second pass -- `track_start` holds, in the partial
// tiles' layout, the value the reference's running sum has when its reverse walk (mixer.rs:100-117) reaches this wave's sources
// (mixer_track_prefix over the first pass's partial tiles); the wave restarts there and leaves end - start (the wave the walk starts
// with keeps its start: zero).  The first pass is the plain kernel.
template <bool FULL, bool STORE = false, int TRACK = 0>
__global__ __launch_bounds__(64) void mixer_mix(uint32_t n_sources, uint32_t n_frames, float interval,
                                                const MixStatic* __restrict__ st, const MixParams* __restrict__ par,
                                                float* __restrict__ partials, uint32_t groups_per_wave, uint32_t n_groups,
                                                float* __restrict__ rows, uint32_t rows_ncb, const float* __restrict__ track_start = nullptr) {
    __shared__ MixerLds L;
    const int lane = threadIdx.x;
    const uint32_t wave = blockIdx.x, tile = blockIdx.y;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
    if (TRACK == 2) {   // (the partial tiles hold interleaved stereo, both channels the mono sum: the left one)
        const float4* src = reinterpret_cast<const float4*>(track_start + ((size_t)tile * gridDim.x + wave) * (2 * MIXER_TILE) + 32 * lane);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 v = src[q]; acc[2 * q] = v.x; acc[2 * q + 1] = v.z; }
    }
    const uint32_t frame0 = tile * MIXER_TILE + 16u * (uint32_t)lane;
    const uint32_t split_log2 = groups_per_wave >> 24;
    uint32_t g_lo = wave * (groups_per_wave & 0xffffffu);
    uint32_t g_hi = g_lo + (groups_per_wave & 0xffffffu);
    int j_lo = 0, j_hi = MIXER_GROUP;
    if (split_log2) {
        g_lo = wave >> split_log2;
        g_hi = g_lo + 1u;
        const int per = MIXER_GROUP >> split_log2;
        j_lo = (int)(wave & ((1u << split_log2) - 1u)) * per;
        j_hi = j_lo + per;
    }
    if (g_hi > n_groups) g_hi = n_groups;
    const int len_tile = (int)n_frames - (int)(tile * MIXER_TILE) > MIXER_TILE ? MIXER_TILE : (int)n_frames - (int)(tile * MIXER_TILE);

    for (uint32_t g = g_hi; g-- > g_lo;) {
        // ---- phase A: one lane per source ----
        const uint32_t srcA = g * MIXER_GROUP + (uint32_t)lane;
        const bool validA = srcA < n_sources;
        MixParams mp = {};
        MixStatic ss = {};
        mp.flags = EAR_SKIP;
        if (validA) { mp = par[srcA]; ss = st[srcA]; }
        const bool live = validA && !(mp.flags & EAR_SKIP);
        int path = PATH_SKIP, fast = 0, wbase = 0, ws = 0, count = 0;
        float frac0 = 0.0f, ds = 0.0f, ph = mp.phase_start;
        int generic = 0;
        if (live && ss.kind == KIND_FRAMES) {
            double t_c = mp.t_start;
            for (uint32_t cc = 0; cc < tile; ++cc) t_c = t_c + (double)interval * 1024.0;
            const double s0 = t_c * (double)ss.clip_rate;
            ds = interval * (float)ss.clip_rate;
            const long long base = f64_as_isize(s0);
            frac0 = (float)(s0 - (double)base);
            fast = fabsf(ds - 1.0f) <= FLT_EPSILON;
            if (!(fabs(s0) < 1.0e9) || !(fabsf(ds) < 16384.0f)) generic = 1;
            wbase = (int)base;
        } else if (live && ss.kind == KIND_SINE) {
            for (uint32_t cc = 0; cc < tile; ++cc) ph = fmodf(ph + (interval * 1024.0f) * ss.freq_or_value, ODDIO_TAU);
        }
        if (live) {
            if (ss.kind == KIND_SINE) path = PATH_SINE;
            else if (ss.kind == KIND_CONSTANT) path = PATH_CONST;
            else {
                // the window only has to contain every index the 1024 sequentially rounded `offset += ds` steps can reach: the
                // cursor at frame 1023 is within 1023 half-ulps of its own magnitude (6.1e-5 relative) of the closed form
                int lo, hi;
                if (fast) { lo = wbase; hi = wbase + 1023; }
                else {
                    const float xb = frac0 + 1023.0f * ds;
                    const float m = fabsf(xb) * 1.0e-4f + 1.0e-2f;
                    if (!(fabsf(xb) < 8.0e6f)) generic = 1;
                    lo = wbase + (int)floorf(fminf(frac0, xb - m));
                    hi = wbase + (int)ceilf(fmaxf(frac0, xb + m));
                }
                ws = lo & ~3;
                count = hi + 2 - ws;
                path = (generic || count > MIXER_WIN_CAP) ? PATH_GENERIC : PATH_LDS;
            }
        }
        L.cinfo[lane * 4 + 0] = wbase - ws;
        L.cinfo[lane * 4 + 1] = __float_as_int(frac0);
        L.cinfo[lane * 4 + 2] = fast;
        L.cinfo[lane * 4 + 3] = path;
        L.sinfo[lane * 2 + 0] = ws;
        L.sinfo[lane * 2 + 1] = count;
        wave_sync();

        // ---- phase B: one source at a time, lane l owns frames 16l..16l+15 of the tile ----
        // a staged source's window travels global memory -> `pre` (registers) -> L.win
        float4 pre[MIXER_WIN_VECS];
#define MIXER_FETCH(JN)                                                                                                   \
    {                                                                                                                     \
        const int ws_n = __builtin_amdgcn_readfirstlane(L.sinfo[(JN) * 2 + 0]);                                           \
        const int nvec_n = (__builtin_amdgcn_readfirstlane(L.sinfo[(JN) * 2 + 1]) + 3) >> 2;                              \
        const uint64_t cp_n = ((uint64_t)(uint32_t)rl_i((int)((uint64_t)ss.clip >> 32), (JN)) << 32) |                    \
                              (uint64_t)(uint32_t)rl_i((int)((uint64_t)ss.clip & 0xffffffffu), (JN));                     \
        const float* clip_n = (const float*)cp_n;                                                                         \
        const int len4_n = (int)((rl_i((int)ss.clip_len, (JN)) + 3) & ~3);                                                \
        _Pragma("unroll") for (int k = 0; k < MIXER_WIN_VECS; ++k) {                                                      \
            const int v = lane + 64 * k, idx = ws_n + 4 * v;                                                              \
            pre[k] = (v < nvec_n && idx >= 0 && idx < len4_n) ? *reinterpret_cast<const float4*>(clip_n + idx) : make_float4(0.f, 0.f, 0.f, 0.f); \
        }                                                                                                                 \
    }
#define MIXER_COMMIT(JN)                                                                                                  \
    {                                                                                                                     \
        const int nvec_n = (__builtin_amdgcn_readfirstlane(L.sinfo[(JN) * 2 + 1]) + 3) >> 2;                              \
        _Pragma("unroll") for (int k = 0; k < MIXER_WIN_VECS; ++k) {                                                      \
            const int v = lane + 64 * k;                                                                                  \
            if (v < nvec_n) {                                                                                             \
                const int li = 4 * v, pos = li + (li >> 4);                                                               \
                L.win[pos + 0] = pre[k].x; L.win[pos + 1] = pre[k].y; L.win[pos + 2] = pre[k].z; L.win[pos + 3] = pre[k].w; \
                if ((li & 15) == 0 && li > 0) L.win[pos - 1] = pre[k].x;                                                  \
            }                                                                                                             \
        }                                                                                                                 \
    }
        {
            int first = -1;
            for (int t = j_hi - 1; t >= j_lo; --t)
                if (__builtin_amdgcn_readfirstlane(L.cinfo[t * 4 + 3]) == PATH_LDS) { first = t; break; }
            if (first >= 0) MIXER_FETCH(first)
        }
#pragma unroll 1
        for (int j = j_hi - 1; j >= j_lo; --j) {
            if (j == j_hi - 1 || (j & (MIXER_SUB - 1)) == MIXER_SUB - 1) {
                // a new batch of 16 sources: the exact f32 cursor scan (frames.rs:189-196) of each, by the lane that holds its
                // parameters, a checkpoint every 16 frames
                wave_sync();
                if ((lane / MIXER_SUB) == (j / MIXER_SUB)) {
                    float x = frac0;
                    float* ck = &L.ckpt[(lane & (MIXER_SUB - 1)) * MIXER_CKPT_STRIDE];
#pragma unroll 1
                    for (int b = 0; b < 63; ++b) {
                        ck[b] = x;
#pragma unroll
                        for (int i = 0; i < 16; ++i) x = x + ds;
                    }
                    ck[63] = x;
                }
                wave_sync();
            }
            const int path_j = __builtin_amdgcn_readfirstlane(L.cinfo[j * 4 + 3]);
            if (STORE) {
                // the previous source's row is out; this one starts from zero (a skipped source leaves a row of zeros: x + 0.0 == x
                // for every x the running sum can hold, which is never -0.0)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
            }
            if (path_j != PATH_SKIP) {
            const float fg = rl_f(ss.fixed_gain, j);
            const bool active = FULL || frame0 < n_frames;
            if (path_j == PATH_LDS) {
                // this source's window is in `pre` (fetched while the staged source before it was rendered): into the LDS stage,
                // padded layout; then the next staged source's fetch starts and lands while this one renders (round 4: one
                // source at a time, load then render, a wave spent 2.5 - 5 us per source waiting)
                wave_sync();
                MIXER_COMMIT(j)
                wave_sync();
                {
                    int nxt = -1;
                    for (int t = j - 1; t >= j_lo; --t)
                        if (__builtin_amdgcn_readfirstlane(L.cinfo[t * 4 + 3]) == PATH_LDS) { nxt = t; break; }
                    if (nxt >= 0) MIXER_FETCH(nxt)
                }
                if (active) {
                    const int wrel = rl_i(L.cinfo[j * 4 + 0], 0);
                    const float fracf = __int_as_float(rl_i(L.cinfo[j * 4 + 1], 0));
                    const int fast_j = rl_i(L.cinfo[j * 4 + 2], 0);
                    const float ds_j = rl_f(ds, j);
                    if (fast_j) {
                        const int w0 = wrel + 16 * lane;
                        float a = L.win[w0 + (w0 >> 4)];
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int w1 = w0 + i + 1;
                            const float bb = L.win[w1 + (w1 >> 4)];
                            float v = a + fracf * (bb - a);
                            v = v * fg;
                            if (FULL || frame0 + (uint32_t)i < n_frames) acc[i] = acc[i] + v;
                            a = bb;
                        }
                    } else {
                        float xx = L.ckpt[(j & (MIXER_SUB - 1)) * MIXER_CKPT_STRIDE + lane];
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int tr = (int)xx;
                            const float fr = xx - (float)tr;
                            const int w = wrel + tr;
                            const int pos = w + (w >> 4);
                            const float a = L.win[pos], bb = L.win[pos + 1];
                            float v = a + fr * (bb - a);
                            v = v * fg;
                            if (FULL || frame0 + (uint32_t)i < n_frames) acc[i] = acc[i] + v;
                            xx = xx + ds_j;
                        }
                    }
                }
            } else if (path_j == PATH_GENERIC) {
                const uint64_t cp = ((uint64_t)(uint32_t)rl_i((int)((uint64_t)ss.clip >> 32), j) << 32) |
                                    (uint64_t)(uint32_t)rl_i((int)((uint64_t)ss.clip & 0xffffffffu), j);
                const float* clip = (const float*)cp;
                const uint32_t clen = (uint32_t)rl_i((int)ss.clip_len, j), crate = (uint32_t)rl_i((int)ss.clip_rate, j);
                const long long tb = __double_as_longlong(mp.t_start);
                const double t0 = __longlong_as_double(((long long)rl_i((int)(tb >> 32), j) << 32) | (long long)(uint32_t)rl_i((int)(tb & 0xffffffff), j));
                if (active) {
                    double t_c = t0;
                    for (uint32_t cc = 0; cc < tile; ++cc) t_c = t_c + (double)interval * 1024.0;
                    const double s0 = t_c * (double)crate;
                    const float dsg = interval * (float)crate;
                    const long long base = f64_as_isize(s0);
                    const float f0 = (float)(s0 - (double)base);
                    const bool fastg = fabsf(dsg - 1.0f) <= FLT_EPSILON;
                    float xx = f0;
                    if (!fastg) for (int k = 0; k < 16 * lane; ++k) xx = xx + dsg;
#pragma unroll 1
                    for (int i = 0; i < 16; ++i) {
                        long long idx; float fr;
                        if (fastg) { idx = base + (long long)(16 * lane + i); fr = f0; }
                        else { const long long tr = (long long)xx; idx = base + tr; fr = xx - (float)tr; }
                        const float a = clip_at(clip, clen, idx), bb = clip_at(clip, clen, idx + 1);
                        float v = a + fr * (bb - a);
                        v = v * fg;
                        if (FULL || frame0 + (uint32_t)i < n_frames) acc[i] = acc[i] + v;
                        xx = xx + dsg;
                    }
                }
            } else {
                const float fv = rl_f(ss.freq_or_value, j);
                const float ph_j = rl_f(ph, j);
                if (active) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float v;
                        if (path_j == PATH_SINE) {
                            const float t = interval * (float)(16 * lane + i);   // sine.rs:36
                            v = sinf(t * fv + ph_j);
                        } else {
                            v = fv;
                        }
                        v = v * fg;
                        if (FULL || frame0 + (uint32_t)i < n_frames) acc[i] = acc[i] + v;
                    }
                }
            }
            }   // path_j != PATH_SKIP
            if (STORE) {
                const uint32_t slot = g * MIXER_GROUP + (uint32_t)j;
                if (slot < n_sources && (FULL || frame0 < n_frames)) {
                    // the lane's 16 sums are one 64-byte row: column block tile * 64 + lane of source `slot`, both ears
                    unsigned char* p = reinterpret_cast<unsigned char*>(rows) + (size_t)(slot >> 4) * (2u * (size_t)rows_ncb * 1024u) +
                                       ((size_t)tile * 64u + (size_t)lane) * 1024u + (size_t)(slot & 15u) * 64u;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float4* d4 = reinterpret_cast<float4*>(p + (size_t)e * rows_ncb * 1024u);
#pragma unroll
                        for (int q = 0; q < 4; ++q) d4[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                    }
                }
            }
        }
        wave_sync();
    }
#undef MIXER_FETCH
#undef MIXER_COMMIT
    (void)len_tile;
    if (STORE) return;
    // MonoToStereo: duplicate (signal.rs:73-80); Mixer adds per channel (mixer.rs:114-116)
    if (TRACK == 2 && wave + 1u != gridDim.x) {
        const float4* src = reinterpret_cast<const float4*>(track_start + ((size_t)tile * gridDim.x + wave) * (2 * MIXER_TILE) + 32 * lane);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 v = src[q]; acc[2 * q] = acc[2 * q] - v.x; acc[2 * q + 1] = acc[2 * q + 1] - v.z; }
    }
    float4* dst = reinterpret_cast<float4*>(partials + ((size_t)tile * gridDim.x + wave) * (2 * MIXER_TILE) + 32 * lane);
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[q] = make_float4(acc[2 * q], acc[2 * q], acc[2 * q + 1], acc[2 * q + 1]);
}

// mixer_mix_unit (round 5): the Mixer's FAST-mode kernel when every live source is a plain MonoToStereo<FramesSignal> (mono clip,
// optional FixedGain) whose clip rate is the output rate -- `interval * rate` within EPSILON of 1, FramesSignal's constant-fract branch
// (frames.rs:180-187): out[i] = lerp(pair(base + i), fract) with ONE fract per source and callback.  No cursor to scan, no window to
// stage: the wave reads the source's 1024 samples straight from HBM, 16 bytes per lane and instruction, fully coalesced -- lane l owns
// the frames 4 l + 256 k .. + 3 (k = 0..3) -- once at the pair's first samples and once four bytes further on (the second samples: the
// same cache lines).  Out-of-clip indices read zeros through a bounds-checked buffer descriptor (frames.rs:105-123).  Per source that
// is 8 loads and 32 VALU operations per lane for 16 output frames; the sources of a group of 64 are software-pipelined two deep.
// Arithmetic per contribution exactly the reference's (lerp, FixedGain); the sum is FAST mode's tree over waves (mixer_reduce).
// The host launches it only when its own bookkeeping says every live source has that shape (mixer_host.inc: n_unit == len).
// Algorithmic bytes per callback: S * (4 * N + 64) + 8 * N; mixer_mix (cursor scan, window through LDS) reached 0.46-0.48 of the
// 8 TB/s peak on it, this kernel is bound by HBM alone.
typedef float mixf4 __attribute__((ext_vector_type(4)));
template <int TRACK = 0>      // (TRACK: see mixer_mix)
__global__ __launch_bounds__(64) void mixer_mix_unit(uint32_t n_sources, uint32_t n_frames, float interval,
                                                     const MixStatic* __restrict__ st, const MixParams* __restrict__ par,
                                                     float* __restrict__ partials, uint32_t groups_per_wave, uint32_t n_groups,
                                                     const float* __restrict__ track_start = nullptr) {
    const int lane = threadIdx.x;
    const uint32_t wave = blockIdx.x, tile = blockIdx.y;
    mixf4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = mixf4{0.0f, 0.0f, 0.0f, 0.0f};
    if (TRACK == 2) {
        const float* src = track_start + ((size_t)tile * gridDim.x + wave) * (2 * MIXER_TILE);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4* s4 = reinterpret_cast<const float4*>(src + 2 * (4 * lane + 256 * k));
            const float4 a0 = s4[0], a1 = s4[1];
            acc[k] = mixf4{a0.x, a0.z, a1.x, a1.z};
        }
    }
    const uint32_t split_log2 = groups_per_wave >> 24;
    uint32_t g_lo = wave * (groups_per_wave & 0xffffffu);
    uint32_t g_hi = g_lo + (groups_per_wave & 0xffffffu);
    int j_lo = 0, j_hi = MIXER_GROUP;
    if (split_log2) {
        g_lo = wave >> split_log2;
        g_hi = g_lo + 1u;
        const int per = MIXER_GROUP >> split_log2;
        j_lo = (int)(wave & ((1u << split_log2) - 1u)) * per;
        j_hi = j_lo + per;
    }
    if (g_hi > n_groups) g_hi = n_groups;
    const int voff_lane = 16 * lane;                 // byte offset of this lane's first frame inside a 256-frame block

    for (uint32_t g = g_hi; g-- > g_lo;) {
        // ---- one lane per source: the clock split of frames.rs:177-181 ----
        const uint32_t srcA = g * MIXER_GROUP + (uint32_t)lane;
        MixParams mp = {};
        MixStatic ss = {};
        mp.flags = EAR_SKIP;
        if (srcA < n_sources) { mp = par[srcA]; ss = st[srcA]; }
        const bool live = srcA < n_sources && !(mp.flags & EAR_SKIP) && ss.kind == KIND_FRAMES;
        int base_i = 0;
        float frac0 = 0.0f;
        if (live) {
            double t_c = mp.t_start;
            for (uint32_t cc = 0; cc < tile; ++cc) t_c = t_c + (double)interval * 1024.0;      // frames.rs:198 per earlier staging chunk
            const double s0 = t_c * (double)ss.clip_rate;
            const long long base = f64_as_isize(s0);
            frac0 = (float)(s0 - (double)base);
            // (indices beyond +-(2^29 - 8192) samples lie outside every clip this kernel is handed; clamped there so that the 32-bit byte
            // offsets below -- 4 * base_i plus at most 4 * 1024 + 16 -- neither wrap nor land inside the descriptor's range: a source
            // scheduled hours ahead reads zeros, frames.rs:105-123)
            constexpr long long BASE_LIM = (1ll << 29) - 8192;
            base_i = (int)(base > BASE_LIM ? BASE_LIM : (base < -BASE_LIM ? -BASE_LIM : base));
        }
        const unsigned long long live_mask = __ballot(live);
        const uint32_t clip_lo = (uint32_t)((uint64_t)ss.clip & 0xffffffffu), clip_hi = (uint32_t)((uint64_t)ss.clip >> 32) & 0xffffu;
        // ---- the sources of the group in descending slot order (mixer.rs:100), two in flight ----
        mixf4 a[2][4], b[2][4];
#define MIXU_LOAD(J, SLOT)                                                                                                \
    {                                                                                                                     \
        const __amdgpu_buffer_rsrc_t r_ = __builtin_amdgcn_make_buffer_rsrc(                                              \
            (void*)(((uint64_t)(uint32_t)rl_i((int)clip_hi, (J)) << 32) | (uint64_t)(uint32_t)rl_i((int)clip_lo, (J))), 0,  \
            (int)(4u * (uint32_t)rl_i((int)ss.clip_len, (J))), 0x00020000);                                               \
        const int vo_ = 4 * rl_i(base_i, (J)) + voff_lane;                                                                \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                   \
            a[SLOT][k] = __builtin_bit_cast(mixf4, __builtin_amdgcn_raw_buffer_load_b128(r_, vo_ + 1024 * k, 0, ODDIO_MIXU_POLICY));       \
            b[SLOT][k] = __builtin_bit_cast(mixf4, __builtin_amdgcn_raw_buffer_load_b128(r_, vo_ + 1024 * k + 4, 0, ODDIO_MIXU_POLICY));   \
        }                                                                                                                 \
    }
        unsigned long long todo = live_mask;
        if (j_hi < 64) todo &= (1ull << j_hi) - 1ull;
        todo &= ~((1ull << j_lo) - 1ull);
        int slot = 0;
        int cur = todo ? 63 - __builtin_clzll(todo) : -1;
        if (cur >= 0) MIXU_LOAD(cur, 0)
        while (cur >= 0) {
            todo &= ~(1ull << cur);
            const int nxt = todo ? 63 - __builtin_clzll(todo) : -1;
            if (nxt >= 0) {
                if (slot == 0) MIXU_LOAD(nxt, 1) else MIXU_LOAD(nxt, 0)
            }
            const float fr = rl_f(frac0, cur), fg = rl_f(ss.fixed_gain, cur);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const mixf4 av = slot == 0 ? a[0][k] : a[1][k], bv = slot == 0 ? b[0][k] : b[1][k];
                mixf4 v = av + fr * (bv - av);                      // frame.rs:39-41 (unfused)
                v = v * fg;                                          // gain.rs:32-37 (x * 1.0 == x without FixedGain)
                acc[k] = acc[k] + v;                                 // mixer.rs:114-116
            }
            slot ^= 1;
            cur = nxt;
        }
#undef MIXU_LOAD
    }
    // MonoToStereo: duplicate (signal.rs:73-80); the partial tile is interleaved stereo like mixer_mix's
    float* dst = partials + ((size_t)tile * gridDim.x + wave) * (2 * MIXER_TILE);
    (void)n_frames;
    if (TRACK == 2 && wave + 1u != gridDim.x) {
        const float* src = track_start + ((size_t)tile * gridDim.x + wave) * (2 * MIXER_TILE);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4* s4 = reinterpret_cast<const float4*>(src + 2 * (4 * lane + 256 * k));
            const float4 a0 = s4[0], a1 = s4[1];
            acc[k] = acc[k] - mixf4{a0.x, a0.z, a1.x, a1.z};
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float4* d4 = reinterpret_cast<float4*>(dst + 2 * (4 * lane + 256 * k));
        d4[0] = make_float4(acc[k].x, acc[k].x, acc[k].y, acc[k].y);
        d4[1] = make_float4(acc[k].z, acc[k].z, acc[k].w, acc[k].w);
    }
}

// ODDIO_HIP_MODE_TRACKED on a Mixer, between the passes: prefix[tile][w][o] = the partial tiles of the waves the reverse walk passes before
// wave w (w + 1 .. n_waves - 1), added in that order -- the layout of the partial tiles in and out.  grid = (2 * MIXER_TILE / 64, tiles);
// block = 64 outputs x 16 segments of the wave list (segment sums, their scan, the running values).
__global__ __launch_bounds__(1024) void mixer_track_prefix(const float* __restrict__ partials, float* __restrict__ prefix, uint32_t n_waves) {
    __shared__ float tot[16][64];
    const uint32_t ox = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.y * n_waves * (2 * MIXER_TILE) + (size_t)blockIdx.x * 64 + ox;
    const int per = (int)((n_waves + 15) / 16);
    const int hi = (int)n_waves - 1 - (int)seg * per;
    const int lo = hi - per + 1 > 0 ? hi - per + 1 : 0;
    constexpr int BATCH = 16;
    float sum = 0.0f;
    for (int w = hi; w >= lo; w -= BATCH) {
        float v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) v[k] = (w - k >= lo) ? partials[base + (size_t)(w - k) * (2 * MIXER_TILE)] : 0.0f;
#pragma unroll
        for (int k = 0; k < BATCH; ++k) sum = sum + v[k];
    }
    tot[seg][ox] = sum;
    __syncthreads();
    float run = 0.0f;
    for (uint32_t k = 0; k < seg; ++k) run = run + tot[k][ox];
    for (int w = hi; w >= lo; w -= BATCH) {
        float v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) v[k] = (w - k >= lo) ? partials[base + (size_t)(w - k) * (2 * MIXER_TILE)] : 0.0f;
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
            if (w - k >= lo) { prefix[base + (size_t)(w - k) * (2 * MIXER_TILE)] = run; run = run + v[k]; }
    }
}

// out[o] = sum over waves (fixed order) of interleaved partial tiles, then Reinhard / Tanh.
// Round 6: 16 outputs x 16 segments per block (was 64 x 16 in 32 blocks of 1 024 threads: an eighth of the chip, one dependent load at a
// time -- 40 us for the 32 MB of a 262 144-source mixer, a sixth of its callback), 8 loads in flight per thread; the segments and the order
// of every addition are what they were: the same bits.
constexpr int MIXRED_OUT = 16, MIXRED_SEGS = 16, MIXRED_BATCH = 8;
__global__ __launch_bounds__(MIXRED_OUT * MIXRED_SEGS) void mixer_reduce(const float* __restrict__ partials, float* __restrict__ out,
                                                                         uint32_t n_waves, uint32_t n_frames, int postfx) {
    __shared__ float red[MIXRED_SEGS][MIXRED_OUT];
    const uint32_t ox = threadIdx.x & (MIXRED_OUT - 1), seg = threadIdx.x / MIXRED_OUT;
    const uint32_t o = blockIdx.x * MIXRED_OUT + ox;
    const uint32_t n_out = 2 * n_frames;
    const uint32_t tile = o / (2 * MIXER_TILE), within = o % (2 * MIXER_TILE);
    float s = 0.0f;
    if (o < n_out) {
        const float* p = partials + (size_t)tile * n_waves * (2 * MIXER_TILE) + within;
        bool first = true;
        for (uint32_t w = seg; w < n_waves; w += MIXRED_SEGS * MIXRED_BATCH) {
            float v[MIXRED_BATCH];
#pragma unroll
            for (int k = 0; k < MIXRED_BATCH; ++k) {
                const uint32_t wk = w + (uint32_t)k * MIXRED_SEGS;
                v[k] = wk < n_waves ? p[(size_t)wk * (2 * MIXER_TILE)] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < MIXRED_BATCH; ++k) {
                if (w + (uint32_t)k * MIXRED_SEGS < n_waves) { s = first ? v[k] : s + v[k]; first = false; }
            }
        }
    }
    red[seg][ox] = s;
    __syncthreads();
    if (seg == 0 && o < n_out) {
        float t = red[0][ox];
        const uint32_t nseg = n_waves < (uint32_t)MIXRED_SEGS ? n_waves : (uint32_t)MIXRED_SEGS;
        for (uint32_t k = 1; k < nseg; ++k) t = t + red[k][ox];
        out[o] = postfx_apply(t, postfx);
    }
}

// ORDERED mode of the general path above the serial threshold (round 4): the slabs ([slot][frame][channels of the leaf]) re-laid as
// the rows ordered_sum reads ([group of 16 sources][ear][16-frame column block][source][16 frames]); a mono slab feeds both ears
// (MonoToStereo, signal.rs:73-80), a stopped source leaves rows of zeros.  One wavefront per (group of 16 sources, 8 column blocks):
// lane (j, q) moves the 16-byte piece q of source j's row -- the 16 rows of a (group, ear, column block) leave as one contiguous KiB
// (one thread per (source, column block), 64 bytes at a 1-KiB stride each, took 0.41 ms for 65 536 sources; this 0.15).
__global__ __launch_bounds__(64) void mixer_general_rows(const float* __restrict__ slabs, const uint32_t* __restrict__ skip,
                                                         const BufStatic* __restrict__ st, uint32_t n_sources, uint32_t n_frames,
                                                         float* __restrict__ rows, uint32_t rows_ncb) {
    const uint32_t G = blockIdx.y, lane = threadIdx.x, j = lane >> 2, q = lane & 3u;
    const uint32_t slot = G * 16u + j;
    const bool valid = slot < n_sources;
    const uint32_t C = valid && st[slot].channels == 2u ? 2u : 1u;
    const bool live = valid && skip[slot] == 0u;
    const float* my = slabs + (size_t)slot * 2 * n_frames;
    unsigned char* gbase = reinterpret_cast<unsigned char*>(rows) + (size_t)G * (2u * (size_t)rows_ncb * 1024u) + (size_t)lane * 16u;
    const uint32_t ncb_used = (n_frames + 15u) / 16u;
#pragma unroll 2
    for (uint32_t k = 0; k < 8u; ++k) {
        const uint32_t cb = blockIdx.x * 8u + k;
        if (cb >= ncb_used) break;
        const uint32_t f0 = cb * 16u + 4u * q;                       // this lane's four frames
        float l[4] = {0.f, 0.f, 0.f, 0.f}, r[4] = {0.f, 0.f, 0.f, 0.f};
        if (live) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t f = f0 + (uint32_t)t;
                if (f < n_frames) { l[t] = my[(size_t)f * C]; r[t] = my[(size_t)f * C + (C - 1u)]; }
            }
        }
        *reinterpret_cast<float4*>(gbase + (size_t)cb * 1024u) = make_float4(l[0], l[1], l[2], l[3]);
        *reinterpret_cast<float4*>(gbase + ((size_t)rows_ncb + cb) * 1024u) = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// FAST mode of the general path (round 4): the slab sum as a tree of row sums.  A workgroup adds `per` consecutive rows (slabs: a mono
// slab feeds both channels, a stopped source is skipped; or partial sums of an earlier stage) into one row of `out`; three stages
// of 64 take 262 144 slabs to one row, in a fixed order, every slab read once with whole-row (4 - 8 KiB) accesses.  (The tiled sum
// above reads 256-byte pieces of each slab from 16 x n_out / 64 workgroups: 0.94 ms for 65 536 sources.)
template <bool SLABS>
__global__ __launch_bounds__(64) void mixer_sum_rows(const float* __restrict__ in, size_t in_stride, const uint32_t* __restrict__ skip,
                                                     const BufStatic* __restrict__ st, uint32_t n_rows, uint32_t per, uint32_t n_out,
                                                     float* __restrict__ out, int postfx, int apply_postfx) {
    // grid = (chunks of `per` rows, quads of 64 output columns): one wavefront, four consecutive outputs per lane
    const uint32_t lo = blockIdx.x * per;
    const uint32_t hi = lo + per < n_rows ? lo + per : n_rows;
    const uint32_t o = 256u * blockIdx.y + 4u * threadIdx.x;
    if (o >= n_out) return;                                             // n_out is even: a lane's quad is whole or half
    const bool whole = o + 3u < n_out;
    auto fetch = [&](uint32_t r) -> float4 {
        const float* row = in + (size_t)r * in_stride;
        if (SLABS) {
            if (skip[r]) return make_float4(0.f, 0.f, 0.f, 0.f);          // (x + 0.0 == x: the running sums are never -0.0)
            if (st[r].channels != 2u) {                                   // a mono slab feeds both channels (signal.rs:73-80)
                const float a = row[o >> 1], b = whole ? row[(o >> 1) + 1u] : 0.f;
                return make_float4(a, a, b, b);
            }
        }
        if (whole) { const f4u t = *reinterpret_cast<const f4u*>(row + o); return make_float4(t.x, t.y, t.z, t.w); }   // (rows of an odd frame count are 8-byte aligned only)
        return make_float4(row[o], row[o + 1], 0.f, 0.f);
    };
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t r = hi;
    while (r >= lo + 4u) {                                              // four rows' loads in flight, added in descending row order
        const float4 v0 = fetch(r - 1u), v1 = fetch(r - 2u), v2 = fetch(r - 3u), v3 = fetch(r - 4u);
        acc.x = ((((acc.x + v0.x) + v1.x) + v2.x) + v3.x); acc.y = ((((acc.y + v0.y) + v1.y) + v2.y) + v3.y);
        acc.z = ((((acc.z + v0.z) + v1.z) + v2.z) + v3.z); acc.w = ((((acc.w + v0.w) + v1.w) + v2.w) + v3.w);
        r -= 4u;
    }
    while (r > lo) {
        const float4 v = fetch(--r);
        acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
    }
    float* dst = out + (size_t)blockIdx.x * n_out + o;
    if (apply_postfx) { acc.x = postfx_apply(acc.x, postfx); acc.y = postfx_apply(acc.y, postfx); acc.z = postfx_apply(acc.z, postfx); acc.w = postfx_apply(acc.w, postfx); }
    dst[0] = acc.x; dst[1] = acc.y;
    if (whole) { dst[2] = acc.z; dst[3] = acc.w; }
}

// The first stage of that tree, over the slabs themselves: grid = (chunks of `per` slabs, blocks of 512 frames); a lane takes eight
// consecutive frames -- 2 KiB of a mono slab (4 of a stereo one) per wavefront and row, where the generic stage above reads 512 bytes:
// 65 536 slabs 128 -> 60 us.  Same sums in the same order.
__global__ __launch_bounds__(64) void mixer_sum_slabs(const float* __restrict__ slabs, const uint32_t* __restrict__ skip, const BufStatic* __restrict__ st,
                                                      uint32_t n_rows, uint32_t per, uint32_t n_frames, float* __restrict__ out, int postfx, int apply_postfx) {
    const uint32_t lo = blockIdx.x * per;
    const uint32_t hi = lo + per < n_rows ? lo + per : n_rows;
    const uint32_t f0 = 512u * blockIdx.y + 8u * threadIdx.x;
    if (f0 >= n_frames) return;
    const uint32_t nf = n_frames - f0 < 8u ? n_frames - f0 : 8u;       // frames of this lane
    float l[8], r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { l[k] = 0.f; r[k] = 0.f; }
    for (uint32_t row = hi; row-- > lo;) {
        if (skip[row]) continue;
        const float* my = slabs + (size_t)row * 2 * n_frames;
        if (st[row].channels == 2u) {
            const float* p = my + 2 * (size_t)f0;
            if (nf == 8u) {
                const f4u a = *reinterpret_cast<const f4u*>(p), b = *reinterpret_cast<const f4u*>(p + 4), c = *reinterpret_cast<const f4u*>(p + 8), d = *reinterpret_cast<const f4u*>(p + 12);
                l[0] += a.x; r[0] += a.y; l[1] += a.z; r[1] += a.w; l[2] += b.x; r[2] += b.y; l[3] += b.z; r[3] += b.w;
                l[4] += c.x; r[4] += c.y; l[5] += c.z; r[5] += c.w; l[6] += d.x; r[6] += d.y; l[7] += d.z; r[7] += d.w;
            } else {
                for (uint32_t k = 0; k < nf; ++k) { l[k] += p[2 * k]; r[k] += p[2 * k + 1]; }
            }
        } else {                                                      // a mono slab feeds both channels (signal.rs:73-80)
            const float* p = my + f0;
            if (nf == 8u) {
                const f4u a = *reinterpret_cast<const f4u*>(p), b = *reinterpret_cast<const f4u*>(p + 4);
                l[0] += a.x; r[0] += a.x; l[1] += a.y; r[1] += a.y; l[2] += a.z; r[2] += a.z; l[3] += a.w; r[3] += a.w;
                l[4] += b.x; r[4] += b.x; l[5] += b.y; r[5] += b.y; l[6] += b.z; r[6] += b.z; l[7] += b.w; r[7] += b.w;
            } else {
                for (uint32_t k = 0; k < nf; ++k) { l[k] += p[k]; r[k] += p[k]; }
            }
        }
    }
    float* dst = out + (size_t)blockIdx.x * 2 * n_frames + 2 * (size_t)f0;
    for (uint32_t k = 0; k < 8u; ++k) {
        if (k >= nf) break;
        dst[2 * k] = apply_postfx ? postfx_apply(l[k], postfx) : l[k];
        dst[2 * k + 1] = apply_postfx ? postfx_apply(r[k], postfx) : r[k];
    }
}

// ---- general path: any leaf (FramesSignal mono/stereo, Sine, Constant, Cycle, Stream) inside any chain of
// FixedGain / Gain / Speed filters, optionally inside a Fader.  One wave per source replays Mixer::sample's per-source
// work (mixer.rs:100-117) through the filter chain into the source's own slab (inner_sample_wave / fader_sample_wave,
// buffered_kernels.h); mixer_general_reduce then adds the slabs in reverse slot order (bit-identical sum order).
// One wave per general-path source: every shape the ABI accepts (mono or stereo leaf under any filter chain, Fader).
// Round 4: the CHAIN sources of the general set -- a mono FramesSignal under FixedGain / Gain / Speed filters, no Fader
// (BufStatic::flags & BUF_FAST_OK), in callbacks of 1 .. 1024 frames (one Mixer staging chunk, mixer.rs:77,109-117) -- are rendered
// by buffered_write (buffered_fast.h: 16 sources per wavefront, the exact running sums scanned one stream per lane) instead of one
// wavefront per source: the source's slab plays the ring, written from index RING_MIRROR of a ring that starts RING_MIRROR floats
// before the slab (so that no store is a mirror store, and a full chunk leaves as whole kilobytes).  This kernel is the walk for
// them, one thread per slot: mixer.rs:100-106's stopped / finished scan, then the record (chain_write_rec), the Smoothed::set and
// clock commits.  Every other source, and a chain source buffered_write cannot take this callback (three Gains ramping at once, a
// resample ratio out of range), is marked BW_SLOW and left -- stop logic included -- to mixer_general_sources_wave.
__global__ __launch_bounds__(128) void mixer_chain_walk(uint32_t n_sources, uint32_t n_frames, float interval, const BufStatic* __restrict__ st,
                                                        BufDyn* __restrict__ dyn, WriteRec* __restrict__ wrecs, float* __restrict__ slabs,
                                                        uint32_t* __restrict__ skip, uint32_t* __restrict__ stopped_hdr, uint32_t stopped_cap,
                                                        uint32_t* __restrict__ bounds_err, int acc_mode) {
    // acc_mode: buffered_write<ACC> adds the chain sources up itself: they have no slab, the slab sums skip them
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sources) return;
    const BufStatic s = st[i];
    WriteRec wr = {};                                   // info == BW_SKIP: nothing to render
    const bool clip_leaf = s.kind == KIND_FRAMES;
    if (!(s.flags & BUF_FAST_OK) || s.fader || !(clip_leaf || s.kind == KIND_SINE || s.kind == KIND_CONSTANT || s.kind == KIND_CYCLE) || s.channels != 1u) { wr.info = BW_SLOW; wrecs[i] = wr; return; }
    const BufDyn d = dyn[i];
    if (d.common.flags & MIXDYN_STOPPED) { skip[i] = 1; wrecs[i] = wr; return; }
    bool fin = (d.common.flags & MIXDYN_STOP_REQUESTED) != 0;                                                       // mixer.rs:102
    if (clip_leaf) fin = fin || d.common.t >= (double)(s.clip_len - 1u) / (double)s.clip_rate;                       // frames.rs:204-206 (a Sine / Constant never finishes)
    if (fin) {
        dyn[i].common.flags = d.common.flags | MIXDYN_STOPPED;
        const uint32_t k = atomicAdd(&stopped_hdr[0], 1u);
        if (k < stopped_cap) stopped_hdr[1 + k] = d.common.id;
        skip[i] = 1;
        wrecs[i] = wr;
        return;
    }
    double t_new;
    float sm_prev[MAX_WRAP], sm_next[MAX_WRAP], sm_prog[MAX_WRAP];
    float* ring = slabs + (size_t)i * 2 * n_frames - RING_MIRROR;
    float phase_new;
    const bool fast = chain_write_rec(wr, bounds_err, s, d, d.common, interval, n_frames, 0u, false, (size_t)RING_MIRROR, ring, 1u << 24, i, true,
                                      t_new, sm_prev, sm_next, sm_prog, phase_new);
    if (fast) {
        dyn[i].common.t = t_new;                                                                                     // frames.rs:198
        dyn[i].common.phase = phase_new;                                                                             // sine.rs:39
#pragma unroll
        for (int w = 0; w < MAX_WRAP; ++w) { dyn[i].sm_prev[w] = sm_prev[w]; dyn[i].sm_next[w] = sm_next[w]; dyn[i].sm_progress[w] = sm_prog[w]; }
        skip[i] = acc_mode ? 1u : 0u;
    } else {
        wr = WriteRec{};
        wr.info = BW_SLOW;
    }
    wrecs[i] = wr;
}

// `wrecs` (null: every source): only the sources mixer_chain_walk left to this kernel (BW_SLOW).
__global__ __launch_bounds__(64) void mixer_general_sources_wave(uint32_t n_sources, uint32_t n_frames, float interval,
                                                                 BufStatic* __restrict__ st, BufDyn* __restrict__ dyn,
                                                                 float* __restrict__ slabs, uint32_t* __restrict__ skip,
                                                                 uint32_t* __restrict__ stopped_hdr, uint32_t stopped_cap,
                                                                 FaderRec* __restrict__ faders, float* __restrict__ fader_scratch,
                                                                 const WriteRec* __restrict__ wrecs) {
    __shared__ float ck[8][64];
    const uint32_t i = blockIdx.x;
    const int lane = threadIdx.x;
    if (i >= n_sources) return;
    if (wrecs && (wrecs[i].info & 7u) != BW_SLOW) return;
    BufStatic s = st[i];
    BufDyn d = dyn[i];
    if (d.common.flags & MIXDYN_STOPPED) { if (lane == 0) skip[i] = 1; return; }
    bool fin = (d.common.flags & MIXDYN_STOP_REQUESTED) != 0;                                                       // mixer.rs:102
    if (!s.fader) {   // Fader::is_finished is always false (fader.rs:76-79); a Cycle, a Sine, a Constant never finish
        if (s.kind == KIND_FRAMES) fin = fin || d.common.t >= (double)(s.clip_len - 1u) / (double)s.clip_rate;      // frames.rs:204-206
        if (s.kind == KIND_STREAM) fin = fin || (d.stream_stopping && d.common.phase == (float)d.stream_len);       // stream.rs:88-90
    }
    if (fin) {
        if (lane == 0) {
            d.common.flags |= MIXDYN_STOPPED;
            const uint32_t k = atomicAdd(&stopped_hdr[0], 1u);
            if (k < stopped_cap) stopped_hdr[1 + k] = d.common.id;
            skip[i] = 1;
            dyn[i] = d;
        }
        return;
    }
    if (lane == 0) skip[i] = 0;
    const uint32_t C = s.channels == 2u ? 2u : 1u;
    float* my = slabs + (size_t)i * 2 * n_frames;
    for (uint32_t done = 0; done < n_frames; done += 1024u) {                                                       // mixer.rs:109-117
        const uint32_t len = (n_frames - done) < 1024u ? (n_frames - done) : 1024u;
        bool faded = false;
        if (s.fader) faded = fader_sample_wave(s, d, faders[s.fader - 1u], fader_scratch + (size_t)(s.fader - 1u) * FADER_BUF * 2u, interval, my + (size_t)done * C, len, ck, lane);
        if (!faded) inner_sample_wave(s, d, interval, my + (size_t)done * C, len, ck, lane);
    }
    if (lane == 0) {   // what the samplers advance (a whole-struct store keeps the untouched fields alive in scratch)
        dyn[i].common.t = d.common.t; dyn[i].common.phase = d.common.phase;
        dyn[i].stream_len = d.stream_len; dyn[i].stream_stopping = d.stream_stopping;
#pragma unroll
        for (int w = 0; w < MAX_WRAP; ++w) {
            dyn[i].shared[w] = d.shared[w];   // (a completed fade brings the new signal's targets with it)
            dyn[i].sm_prev[w] = d.sm_prev[w]; dyn[i].sm_next[w] = d.sm_next[w]; dyn[i].sm_progress[w] = d.sm_progress[w];
        }
        if (s.fader) st[i] = s;   // a completed fade swapped the signals
    }
}
__global__ void mixer_general_reduce(const float* __restrict__ slabs, const uint32_t* __restrict__ skip, const BufStatic* __restrict__ st,
                                     uint32_t n_sources, uint32_t n_frames, float* __restrict__ out, int postfx) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= 2 * n_frames) return;
    const uint32_t f = o >> 1, ch = o & 1;
    float s = 0.0f;
    for (uint32_t i = n_sources; i-- > 0;) {
        if (skip[i]) continue;
        const float* my = slabs + (size_t)i * 2 * n_frames;
        const float v = (st[i].channels == 2) ? my[2 * f + ch] : my[f];   // MonoToStereo: duplicate (signal.rs:73-80)
        s = s + v;                                                          // frame::mix, mixer.rs:114-116
    }
    out[o] = postfx_apply(s, postfx);
}

// The same sum for many sources: grid = (ceil(n_out / 64), n_slices), block = 256.  Slice k adds slots
// [k*per, (k+1)*per) in descending order (rows staged through LDS 64 at a time, so that the sequential adds do
// not wait for HBM one by one); mixer_general_reduce_finish adds the slices, last slice first, and applies the
// post filter.  ORDERED mode uses ONE slice: the reference's exact sequence (mixer.rs:100-117).
constexpr int MIXRED_ROWS = 64;
__global__ __launch_bounds__(256) void mixer_general_reduce_tiled(const float* __restrict__ slabs, const uint32_t* __restrict__ skip,
                                                                  const BufStatic* __restrict__ st, uint32_t n_sources, uint32_t n_frames,
                                                                  float* __restrict__ part, uint32_t n_slices) {
    __shared__ float tile[MIXRED_ROWS][65];
    __shared__ uint32_t sk[MIXRED_ROWS];
    const uint32_t n_out = 2 * n_frames;
    const uint32_t ox = threadIdx.x & 63, sy = threadIdx.x >> 6;
    const uint32_t o = blockIdx.x * 64 + ox;
    const uint32_t per = (n_sources + n_slices - 1) / n_slices;
    const uint32_t lo = blockIdx.y * per;
    uint32_t hi = lo + per < n_sources ? lo + per : n_sources;
    float s = 0.0f;
    while (hi > lo) {
        const uint32_t base = hi - lo >= (uint32_t)MIXRED_ROWS ? hi - MIXRED_ROWS : lo;
        const uint32_t cnt = hi - base;
        for (uint32_t r = sy; r < cnt; r += 4) {
            const float* my = slabs + (size_t)(base + r) * n_out;
            tile[r][ox] = o < n_out ? ((st[base + r].channels == 2) ? my[o] : my[o >> 1]) : 0.0f;   // MonoToStereo: duplicate (signal.rs:73-80)
        }
        if (threadIdx.x < cnt) sk[threadIdx.x] = skip[base + threadIdx.x];
        __syncthreads();
        if (sy == 0)
            for (uint32_t r = cnt; r-- > 0;)
                if (!sk[r]) s = s + tile[r][ox];                                                     // frame::mix, mixer.rs:114-116
        __syncthreads();
        hi = base;
    }
    if (sy == 0 && o < n_out) part[(size_t)blockIdx.y * n_out + o] = s;
}
__global__ void mixer_general_reduce_finish(const float* __restrict__ part, uint32_t n_slices, uint32_t n_frames, float* __restrict__ out, int postfx) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_out = 2 * n_frames;
    if (o >= n_out) return;
    float s = part[(size_t)(n_slices - 1) * n_out + o];          // the walk starts at the last slot
    for (uint32_t k = n_slices - 1; k-- > 0;) s = s + part[(size_t)k * n_out + o];
    out[o] = postfx_apply(s, postfx);
}

// fast representation -> general representation (when the first filtered / cycle / stereo source arrives)
__global__ void mixer_convert_to_general(uint32_t n, const MixStatic* __restrict__ ms, const MixDyn* __restrict__ md,
                                         BufStatic* __restrict__ bs, BufDyn* __restrict__ bd) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BufStatic s = {};
    s.clip = ms[i].clip; s.clip_len = ms[i].clip_len; s.clip_rate = ms[i].clip_rate; s.freq_or_value = ms[i].freq_or_value;
    s.kind = ms[i].kind; s.channels = 1;
    s.flags = (ms[i].kind == KIND_FRAMES || ms[i].kind == KIND_SINE || ms[i].kind == KIND_CONSTANT) ? BUF_FAST_OK : 0u;      // (a chain source from now on: mixer_chain_walk)
    if (ms[i].fixed_gain != 1.0f) { s.n_wrap = 1; s.wrap_kind[0] = WRAP_FIXED_GAIN; s.wrap_param[0] = ms[i].fixed_gain; }
    BufDyn d = {};
    d.common.t = md[i].t; d.common.phase = md[i].phase; d.common.flags = md[i].flags; d.common.id = md[i].id;
    for (int w = 0; w < MAX_WRAP; ++w) { d.shared[w] = 1.0f; d.sm_prev[w] = 1.0f; d.sm_next[w] = 1.0f; d.sm_progress[w] = 1.0f; }
    bs[i] = s;
    bd[i] = d;
}

__global__ void mixer_general_request_stop(const uint32_t* __restrict__ slots, uint32_t n, BufDyn* dyn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dyn[slots[i]].common.flags |= MIXDYN_STOP_REQUESTED;
}
__global__ void mixer_general_apply_moves(const BufMove* __restrict__ mv, uint32_t n, BufStatic* st, BufDyn* dyn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st[mv[i].dst] = st[mv[i].src];
    dyn[mv[i].dst] = dyn[mv[i].src];
}

struct MixerMove { uint32_t dst, src; };
__global__ void mixer_apply_moves(const MixerMove* __restrict__ mv, uint32_t n, MixStatic* st, MixDyn* dyn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st[mv[i].dst] = st[mv[i].src];
    dyn[mv[i].dst] = dyn[mv[i].src];
}
__global__ void mixer_request_stop(const uint32_t* __restrict__ slots, uint32_t n, MixDyn* dyn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dyn[slots[i]].flags |= MIXDYN_STOP_REQUESTED;
}

}  // namespace oddio_hip
