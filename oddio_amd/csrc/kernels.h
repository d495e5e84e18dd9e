// kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels of the SpatialScene hot path.
//
// Compiled with -ffp-contract=off and correctly rounded f32 divide/sqrt: every f32/f64 operation
// is a separately rounded IEEE op in the reference's order, so per-source contributions are
// bit-identical to the reference CPU `Signal::sample()` for FramesSignal/Constant sources
// (Sine differs only by the device sinf vs glibc sinf, a few ulp).
//
// Kernels
//   spatial_prepass   1 thread / source   walk_set + EarState + cursor bookkeeping
//                                         (spatial.rs:191-265, :445-469 scalar part, :501-549)
//   spatial_mix       1 wave  / 8 sources per step; the per-sample loop
//                                         (spatial.rs:456-463 + frames.rs:176-201 + sine.rs:34-40)
//   reduce_partials   fixed-order sum of the per-wave stereo partials + Reinhard/Tanh epilogue
//                                         (reinhard.rs:32, tanh.rs:26)
//
// Mix kernel work decomposition (why it is not "one lane = one output frame"):
//   FramesSignal's slow path advances its f32 cursor by a *sequentially rounded* `offset += ds`
//   (frames.rs:189-196) restarted from the f64 clock every <=256-frame chunk (spatial.rs:456).
//   A closed form offset0 + k*ds is not within tolerance, so the running sum must
//   be reproduced exactly.  Phase A: 64 lanes = 8 sources x 2 ears x 4 chunks each run the exact
//   255-step f32 scan once and leave 16 checkpoints (every 16 frames) in LDS.  Phase B: for one
//   source at a time, lane l owns output frames 16l..16l+15 (both ears, 32 register accumulators),
//   restarts from its checkpoint and replays 15 exact adds.  The source's sample window
//   (~N*ds + 32 floats) is staged once, coalesced (16 B/lane), into LDS with one pad float per
//   16 samples (lane stride 17 => conflict-free ds_read2_b32), zero-filled outside the clip so that
//   frames.rs:105-123 `get_pair` needs no branches.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

#include "device_types.h"

namespace oddio_hip {

#define ODDIO_TAU 6.28318530717958647692528676655900577f
#define ODDIO_SPEED_OF_SOUND 343.0f
#define ODDIO_HEAD_RADIUS 0.1075f
#define ODDIO_POSITION_SMOOTHING_PERIOD 0.5f

// ---------------------------------------------------------------------------------------------
// math/mod.rs:33-94 on plain floats (same evaluation order as the reference)
// ---------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
struct Quat { float s, x, y, z; };

__device__ __forceinline__ float v3_norm(V3 a) {
    float s = 0.0f;
    s = s + a.x * a.x;
    s = s + a.y * a.y;
    s = s + a.z * a.z;
    return sqrtf(s);
}
__device__ __forceinline__ V3 v3_scale(V3 a, float f) { return {a.x * f, a.y * f, a.z * f}; }
__device__ __forceinline__ V3 v3_add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 v3_mix(V3 a, V3 b, float r) {
    float ir = 1.0f - r;
    return {ir * a.x + r * b.x, ir * a.y + r * b.y, ir * a.z + r * b.z};
}
__device__ __forceinline__ Quat quat_mul(Quat q, Quat r) {
    Quat o;
    o.s = q.s * r.s - q.x * r.x - q.y * r.y - q.z * r.z;
    o.x = q.s * r.x + q.x * r.s + q.y * r.z - q.z * r.y;
    o.y = q.s * r.y - q.x * r.z + q.y * r.s + q.z * r.x;
    o.z = q.s * r.z + q.x * r.y - q.y * r.x + q.z * r.s;
    return o;
}
__device__ __forceinline__ V3 quat_rotate(Quat rot, V3 p) {
    Quat pq = {0.0f, p.x, p.y, p.z};
    Quat inv = {rot.s, -rot.x, -rot.y, -rot.z};
    Quat o = quat_mul(rot, quat_mul(pq, inv));
    return {o.x, o.y, o.z};
}

// spatial.rs:501-511
__device__ __forceinline__ V3 smoothed_position(V3 prev_position, float state_dt, float dt_arg, V3 npos, V3 nvel) {
    float dt = state_dt + dt_arg;
    V3 change = v3_scale(nvel, dt);
    V3 naive = v3_add(prev_position, change);
    V3 intended = v3_add(npos, change);
    return v3_mix(naive, intended, fminf(dt / ODDIO_POSITION_SMOOTHING_PERIOD, 1.0f));
}

// spatial.rs:531-549 + Ear::pos/dir :573-598
__device__ __forceinline__ void ear_state(V3 p, int ear, float radius, float& offset, float& gain) {
    const float ex = ear == 0 ? -ODDIO_HEAD_RADIUS : ODDIO_HEAD_RADIUS;
    const float sign = ear == 0 ? -1.0f : 1.0f;
    const float dirx = sign * 4.0f / sqrtf(17.0f);
    const float dirz = -1.0f / sqrtf(17.0f);
    V3 v = {p.x - ex, p.y - 0.0f, p.z - 0.0f};
    float distance = v3_norm(v);
    offset = distance * (-1.0f / ODDIO_SPEED_OF_SOUND);
    float distance_gain = radius / fmaxf(distance, radius);
    float stereo;
    if (distance < 1e-3f) {
        stereo = 0.5f + 0.5f;
    } else {
        float k = 0.5f / distance;
        V3 q = v3_scale(p, k);
        float d = 0.0f;
        d = d + dirx * q.x;
        d = d + 0.0f * q.y;
        d = d + dirz * q.z;
        stereo = 0.5f + d;
    }
    gain = stereo * distance_gain;
}

// Rust `f64 as isize` (saturating, NaN -> 0)
__device__ __forceinline__ long long f64_as_isize(double x) {
    if (x != x) return 0;
    if (x >= 9223372036854775807.0) return 0x7fffffffffffffffLL;
    if (x <= -9223372036854775808.0) return (long long)0x8000000000000000ULL;
    return (long long)x;
}

// ---------------------------------------------------------------------------------------------
// prepass: one thread per live slot
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spatial_prepass(SceneParams P, const SrcStatic* __restrict__ st,
                                                       SrcDyn* __restrict__ dyn, SrcPending* __restrict__ pend,
                                                       EarParams* __restrict__ ear, uint32_t* __restrict__ stopped_hdr,
                                                       uint32_t stopped_cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_sources) return;
    SrcDyn d = dyn[i];
    const SrcStatic s = st[i];
    EarParams e0 = {}, e1 = {};
    if (d.flags & DYN_STOPPED) {  // removed earlier, compaction not applied yet: never mixed again
        e0.flags = EAR_SKIP; e1.flags = EAR_SKIP;
        ear[2 * i] = e0; ear[2 * i + 1] = e1;
        return;
    }
    const float elapsed = P.elapsed;
    const float nf = (float)P.n_frames;
    V3 tpos = {d.tgt_pos[0], d.tgt_pos[1], d.tgt_pos[2]};
    V3 tvel = {d.tgt_vel[0], d.tgt_vel[1], d.tgt_vel[2]};
    V3 ppos = {d.prev_pos[0], d.prev_pos[1], d.prev_pos[2]};
    // spatial.rs:216-226 motion.refresh()
    const SrcPending pm = pend[i];
    if (pm.flags & PEND_FRESH) {
        V3 npos = {pm.pos[0], pm.pos[1], pm.pos[2]};
        V3 nvel = {pm.vel[0], pm.vel[1], pm.vel[2]};
        ppos = (pm.flags & PEND_DISCONTINUITY) ? npos : smoothed_position(ppos, d.state_dt, 0.0f, tpos, tvel);
        tpos = npos; tvel = nvel;
        d.state_dt = 0.0f;
        pend[i].flags = 0;
    }
    const Quat prev_rot = {P.prev_rot[0], P.prev_rot[1], P.prev_rot[2], P.prev_rot[3]};
    const Quat rot = {P.rot[0], P.rot[1], P.rot[2], P.rot[3]};
    const V3 p0 = quat_rotate(prev_rot, smoothed_position(ppos, d.state_dt, 0.0f, tpos, tvel));      // :228-231
    const V3 p1 = quat_rotate(rot, smoothed_position(ppos, d.state_dt, elapsed, tpos, tvel));         // :232-235
    d.state_dt = d.state_dt + elapsed;                                                                // :238
    d.tgt_pos[0] = tpos.x; d.tgt_pos[1] = tpos.y; d.tgt_pos[2] = tpos.z;
    d.tgt_vel[0] = tvel.x; d.tgt_vel[1] = tvel.y; d.tgt_vel[2] = tvel.z;
    d.prev_pos[0] = ppos.x; d.prev_pos[1] = ppos.y; d.prev_pos[2] = ppos.z;

    // spatial.rs:243-261 finished bookkeeping (propagation-delay aware)
    const float distance = v3_norm(p0);
    if (d.flags & DYN_HAS_FINISHED_FOR) {
        if (d.finished_for > distance / ODDIO_SPEED_OF_SOUND) d.flags |= DYN_STOPPED;
        else d.finished_for = d.finished_for + elapsed;
    } else {
        bool fin = false;
        if (s.kind == KIND_FRAMES) fin = d.t >= (double)(s.clip_len - 1u) / (double)s.clip_rate;     // frames.rs:204-206
        if (fin) { d.flags |= DYN_HAS_FINISHED_FOR; d.finished_for = elapsed; }
    }
    if (d.flags & DYN_STOPPED) {
        const uint32_t k = atomicAdd(&stopped_hdr[0], 1u);
        if (k < stopped_cap) stopped_hdr[1 + k] = d.id;
        e0.flags = EAR_SKIP; e1.flags = EAR_SKIP;
        ear[2 * i] = e0; ear[2 * i + 1] = e1;
        dyn[i] = d;
        return;
    }

    // spatial.rs:446-468: the scalar part of mix_signal; the sampling itself is the mix kernel's.
    const uint32_t n = P.n_frames;
    for (int e = 0; e < 2; ++e) {
        float off0, g0, off1, g1;
        ear_state(p0, e, s.radius, off0, g0);
        ear_state(p1, e, s.radius, off1, g1);
        const float eff = (elapsed + off1) - off0;      // :451
        const float dt = eff / nf;                      // :452
        const float dg = (g1 - g0) / nf;                // :453
        EarParams ep = {};
        ep.dt = dt; ep.g0 = g0; ep.dg = dg;
        const float back = -eff - off0;                 // :465
        if (s.kind == KIND_FRAMES) {
            d.t = d.t + (double)off0;                   // seek(prev_state.offset), frames.rs:211-213
            ep.t_ear = d.t;
            for (uint32_t done = 0; done < n; done += 256u) {
                const uint32_t len = (n - done) < 256u ? (n - done) : 256u;
                d.t = d.t + (double)dt * (double)len;   // frames.rs:198
            }
            d.t = d.t + (double)back;
        } else if (s.kind == KIND_SINE) {
            const float fr = s.freq_or_value;
            d.phase = fmodf(d.phase + off0 * fr, ODDIO_TAU);   // sine.rs:25-28
            ep.phase_ear = d.phase;
            for (uint32_t done = 0; done < n; done += 256u) {
                const uint32_t len = (n - done) < 256u ? (n - done) : 256u;
                d.phase = fmodf(d.phase + (dt * (float)len) * fr, ODDIO_TAU);   // sine.rs:39
            }
            d.phase = fmodf(d.phase + back * fr, ODDIO_TAU);
        }
        ear[2 * i + e] = ep;
    }
    if (s.kind == KIND_FRAMES) d.t = d.t + (double)elapsed;                                     // :468
    else if (s.kind == KIND_SINE) d.phase = fmodf(d.phase + elapsed * s.freq_or_value, ODDIO_TAU);
    dyn[i] = d;
}

// ---------------------------------------------------------------------------------------------
// mix kernel
// ---------------------------------------------------------------------------------------------
constexpr int MIX_GROUP = 8;                 // sources per phase-A step
constexpr int WIN_CAP = 1536;                // samples staged per source (covers ds <= ~1.46 at N=1024)
constexpr int WIN_PAD = WIN_CAP + WIN_CAP / 16 + 16;
constexpr int CKPT_STRIDE = 17;
constexpr int TILE_FRAMES = 1024;            // frames per wave pass = 4 chunks of 256 (spatial.rs:393)

enum : int { PATH_SKIP = 0, PATH_LDS = 1, PATH_GENERIC = 2, PATH_SINE = 3, PATH_CONST = 4 };

struct MixLds {
    float ckpt[64 * CKPT_STRIDE];            // cursor checkpoints, [phaseA lane][16] (+1 pad)
    int cinfo[64 * 4];                       // per phase-A lane: {wrel, frac/fast bits, fast, len}
    int sinfo[MIX_GROUP * 4];                // per source: {ws, count, path, -}
    float win[WIN_PAD];                      // padded sample window of the current source
};
constexpr size_t MIX_LDS_BYTES = sizeof(MixLds);

__device__ __forceinline__ void wave_sync() {
    // single-wave workgroups: LDS ops of one wave execute in order; this is a compiler fence plus
    // the (free for one wave) barrier.
    __syncthreads();
}

__device__ __forceinline__ float rl_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// One source, LDS path, both ears.  acc[2*i+e] += lerp * gain  (spatial.rs:458-462)
template <bool FULL, bool HAS_FG>
__device__ __forceinline__ void mix_source_lds(const MixLds& L, int j, int lane, float (&acc)[32], float fbase,
                                               uint32_t frame0, uint32_t n_frames, float fixed_gain,
                                               float g0L, float dgL, float dsL, float g0R, float dgR, float dsR) {
    const int c = lane >> 4, b = lane & 15;
    if (!FULL && frame0 >= n_frames) return;   // this lane's 16 frames lie past the end of `out`
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float g0 = e ? g0R : g0L, dg = e ? dgR : dgL, ds = e ? dsR : dsL;
        const int la = j * 8 + e * 4 + c;
        const int wrel = L.cinfo[la * 4 + 0];
        const float fracf = __int_as_float(L.cinfo[la * 4 + 1]);
        const int fast = L.cinfo[la * 4 + 2];
        if (fast) {
            // frames.rs:180-187: constant fract, consecutive pairs
            const int w0 = wrel + 16 * b;
            int pos = w0 + (w0 >> 4);
            float a = L.win[pos];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int w1 = w0 + i + 1;
                const float bb = L.win[w1 + (w1 >> 4)];
                float v = a + fracf * (bb - a);
                if (HAS_FG) v = v * fixed_gain;
                const float gain = g0 + (fbase + (float)i) * dg;
                const float p = v * gain;
                if (FULL || frame0 + (uint32_t)i < n_frames) acc[2 * i + e] = acc[2 * i + e] + p;
                a = bb;
            }
        } else {
            // frames.rs:189-196: x_{16b+i} = x_{16b} (+ ds) i times, exactly as the scan produced it
            float x = L.ckpt[la * CKPT_STRIDE + b];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int tr = (int)x;
                const float fr = x - (float)tr;
                const int w = wrel + tr;
                const int pos = w + (w >> 4);
                const float a = L.win[pos];
                const float bb = L.win[pos + 1];
                float v = a + fr * (bb - a);
                if (HAS_FG) v = v * fixed_gain;
                const float gain = g0 + (fbase + (float)i) * dg;
                const float p = v * gain;
                if (FULL || frame0 + (uint32_t)i < n_frames) acc[2 * i + e] = acc[2 * i + e] + p;
                x = x + ds;
            }
        }
    }
}

// frames.rs:105-123 straight from global memory (sources whose window does not fit the LDS stage,
// absurd cursors, |ds| huge ...).  Correct for every input, slow.
__device__ __forceinline__ float clip_at(const float* clip, uint32_t len, long long i) {
    return (i >= 0 && i < (long long)len) ? clip[i] : 0.0f;
}

template <bool FULL>
__device__ __forceinline__ void mix_source_generic(int lane, float (&acc)[32], float fbase, uint32_t frame0,
                                                uint32_t n_frames, uint32_t tile, const float* clip, uint32_t clip_len,
                                                uint32_t clip_rate, float fixed_gain, double t_earL, float dtL, float g0L,
                                                float dgL, double t_earR, float dtR, float g0R, float dgR) {
    const int c = lane >> 4, b = lane & 15;
    const uint32_t c_abs = tile * 4u + (uint32_t)c;
#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
        const double t_ear = e ? t_earR : t_earL;
        const float dt = e ? dtR : dtL, g0 = e ? g0R : g0L, dg = e ? dgR : dgL;
        double t_c = t_ear;
        for (uint32_t cc = 0; cc < c_abs; ++cc) t_c = t_c + (double)dt * 256.0;
        const double s0 = t_c * (double)clip_rate;
        const float ds = dt * (float)clip_rate;
        const long long base = f64_as_isize(s0);
        const float frac0 = (float)(s0 - (double)base);
        const bool fast = fabsf(ds - 1.0f) <= FLT_EPSILON;
        float x = frac0;
        if (!fast) for (int k = 0; k < 16 * b; ++k) x = x + ds;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            long long idx; float fr;
            if (fast) { idx = base + (long long)(16 * b + i); fr = frac0; }
            else { const long long tr = (long long)x; idx = base + tr; fr = x - (float)tr; }
            const float a = clip_at(clip, clip_len, idx), bb = clip_at(clip, clip_len, idx + 1);
            float v = a + fr * (bb - a);
            v = v * fixed_gain;
            const float gain = g0 + (fbase + (float)i) * dg;
            const float p = v * gain;
            if (FULL || frame0 + (uint32_t)i < n_frames) acc[2 * i + e] = acc[2 * i + e] + p;
            x = x + ds;
        }
    }
}

// sine.rs:34-40 inside the spatial chunk loop; Constant (constant.rs:16-18)
template <bool FULL, bool IS_SINE>
__device__ __forceinline__ void mix_source_analytic(int lane, float (&acc)[32], float fbase, uint32_t frame0,
                                                 uint32_t n_frames, uint32_t tile, float freq_or_value, float fixed_gain,
                                                 float phL, float dtL, float g0L, float dgL, float phR, float dtR,
                                                 float g0R, float dgR) {
    const int c = lane >> 4, b = lane & 15;
    const uint32_t c_abs = tile * 4u + (uint32_t)c;
#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
        const float dt = e ? dtR : dtL, g0 = e ? g0R : g0L, dg = e ? dgR : dgL;
        float ph = e ? phR : phL;
        if (IS_SINE) for (uint32_t cc = 0; cc < c_abs; ++cc) ph = fmodf(ph + (dt * 256.0f) * freq_or_value, ODDIO_TAU);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v;
            if (IS_SINE) {
                const float t = dt * (float)(16 * b + i);
                v = sinf(t * freq_or_value + ph);
            } else {
                v = freq_or_value;
            }
            v = v * fixed_gain;
            const float gain = g0 + (fbase + (float)i) * dg;
            const float p = v * gain;
            if (FULL || frame0 + (uint32_t)i < n_frames) acc[2 * i + e] = acc[2 * i + e] + p;
        }
    }
}

// grid = (n_waves, n_tiles); block = 64 (one wave).  Wave w walks groups [g_lo, g_hi) of 8 slots
// in DESCENDING order (the reference's reverse set walk, spatial.rs:204) and leaves its partial
// stereo tile in partials[(tile * n_waves + w) * 2048 + ...] (interleaved L,R).
template <bool FULL>
__global__ __launch_bounds__(64) void spatial_mix(SceneParams P, const SrcStatic* __restrict__ st,
                                                  const EarParams* __restrict__ ear, float* __restrict__ partials,
                                                  uint32_t groups_per_wave, uint32_t n_groups) {
    __shared__ MixLds L;
    const int lane = threadIdx.x;
    const uint32_t wave = blockIdx.x, tile = blockIdx.y;
    const uint32_t n_frames = P.n_frames;
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0f;
    const uint32_t frame0 = tile * TILE_FRAMES + 16u * (uint32_t)lane;   // this lane's first output frame
    const float fbase = (float)frame0;                                    // `i as f32` base (spatial.rs:459)

    const uint32_t g_lo = wave * groups_per_wave;
    uint32_t g_hi = g_lo + groups_per_wave;
    if (g_hi > n_groups) g_hi = n_groups;

    // phase-A role of this lane
    const int jA = lane >> 3, eA = (lane >> 2) & 1, cA = lane & 3;
    const uint32_t cA_abs = tile * 4u + (uint32_t)cA;
    const int lenA = (int)n_frames - (int)(cA_abs * 256u) < 0 ? 0 : ((int)n_frames - (int)(cA_abs * 256u) > 256 ? 256 : (int)n_frames - (int)(cA_abs * 256u));

    for (uint32_t g = g_hi; g-- > g_lo;) {
        // ------------------------------ phase A ------------------------------
        const uint32_t srcA = g * MIX_GROUP + (uint32_t)jA;
        const bool validA = srcA < P.n_sources;
        EarParams ep = {};
        SrcStatic ss = {};
        ep.flags = EAR_SKIP;
        if (validA) { ep = ear[2 * srcA + eA]; ss = st[srcA]; }
        const bool live = validA && !(ep.flags & EAR_SKIP);
        int lo = 0x7fffffff, hi = (int)0x80000000;
        int generic = 0, fast = 0, wbase = 0;
        float frac0 = 0.0f, ds = 0.0f;
        if (live && ss.kind == KIND_FRAMES) {
            double t_c = ep.t_ear;
            for (uint32_t cc = 0; cc < cA_abs; ++cc) t_c = t_c + (double)ep.dt * 256.0;   // frames.rs:198 per chunk
            const double s0 = t_c * (double)ss.clip_rate;                                 // frames.rs:177
            ds = ep.dt * (float)ss.clip_rate;                                             // :178
            const long long base = f64_as_isize(s0);                                      // :179
            frac0 = (float)(s0 - (double)base);                                           // :181 / :189
            fast = fabsf(ds - 1.0f) <= FLT_EPSILON;                                       // :180
            if (!(fabs(s0) < 1.0e9) || !(fabsf(ds) < 65536.0f)) generic = 1;
            wbase = (int)base;
        }
        // exact f32 cursor scan (frames.rs:189-196); checkpoints every 16 frames
        float x = frac0;
        {
            float* ck = &L.ckpt[lane * CKPT_STRIDE];
#pragma unroll 1
            for (int b = 0; b < 15; ++b) {
                ck[b] = x;
#pragma unroll
                for (int i = 0; i < 16; ++i) x = x + ds;
            }
            ck[15] = x;
#pragma unroll
            for (int i = 0; i < 15; ++i) x = x + ds;   // x == offset at frame 255 of the chunk
        }
        if (live && ss.kind == KIND_FRAMES && lenA > 0 && !generic) {
            int i0, i1;
            if (fast) { i0 = wbase; i1 = wbase + 255; }
            else {
                if (!(fabsf(x) < 8.0e6f)) generic = 1;
                i0 = wbase + (int)frac0;
                i1 = wbase + (int)x;
            }
            lo = i0 < i1 ? i0 : i1;
            hi = i0 < i1 ? i1 : i0;
        }
        // per-source (8 lanes) reduction of window bounds
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            const int olo = __shfl_xor(lo, m), ohi = __shfl_xor(hi, m), og = __shfl_xor(generic, m);
            lo = olo < lo ? olo : lo;
            hi = ohi > hi ? ohi : hi;
            generic |= og;
        }
        const int ws = lo & ~3;
        const int count = hi + 2 - ws;
        int path = PATH_SKIP;
        if (live) {
            if (ss.kind == KIND_SINE) path = PATH_SINE;
            else if (ss.kind == KIND_CONSTANT) path = PATH_CONST;
            else if (generic) path = PATH_GENERIC;
            else if (lo > hi) path = PATH_SKIP;      // no frames in this tile
            else path = (count <= WIN_CAP) ? PATH_LDS : PATH_GENERIC;
        }
        L.cinfo[lane * 4 + 0] = wbase - ws;
        L.cinfo[lane * 4 + 1] = __float_as_int(frac0);
        L.cinfo[lane * 4 + 2] = fast;
        L.cinfo[lane * 4 + 3] = lenA;
        if ((lane & 7) == 0) {
            L.sinfo[jA * 4 + 0] = ws;
            L.sinfo[jA * 4 + 1] = count;
            L.sinfo[jA * 4 + 2] = path;
        }
        wave_sync();

        // ------------------------------ phase B ------------------------------
#pragma unroll 1
        for (int j = MIX_GROUP - 1; j >= 0; --j) {
            const int path_j = __builtin_amdgcn_readfirstlane(L.sinfo[j * 4 + 2]);
            if (path_j == PATH_SKIP) continue;
            const int laL = j * 8, laR = j * 8 + 4;
            const float g0L = rl_f(ep.g0, laL), dgL = rl_f(ep.dg, laL), dtL = rl_f(ep.dt, laL);
            const float g0R = rl_f(ep.g0, laR), dgR = rl_f(ep.dg, laR), dtR = rl_f(ep.dt, laR);
            const float fg = rl_f(ss.fixed_gain, laL);
            if (path_j == PATH_LDS) {
                const int ws_j = __builtin_amdgcn_readfirstlane(L.sinfo[j * 4 + 0]);
                const int count_j = __builtin_amdgcn_readfirstlane(L.sinfo[j * 4 + 1]);
                const uint64_t cp = ((uint64_t)(uint32_t)rl_i((int)((uint64_t)ss.clip >> 32), laL) << 32) |
                                    (uint64_t)(uint32_t)rl_i((int)((uint64_t)ss.clip & 0xffffffffu), laL);
                const float* clip = (const float*)cp;
                const int clip_len4 = (int)((rl_i((int)ss.clip_len, laL) + 3) & ~3);
                const int nvec = (count_j + 3) >> 2;
                wave_sync();   // previous source's readers are done with L.win
                for (int v = lane; v < nvec; v += 64) {
                    const int idx = ws_j + 4 * v;
                    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (idx >= 0 && idx < clip_len4) val = *reinterpret_cast<const float4*>(clip + idx);
                    const int li = 4 * v;
                    const int pos = li + (li >> 4);
                    L.win[pos + 0] = val.x; L.win[pos + 1] = val.y; L.win[pos + 2] = val.z; L.win[pos + 3] = val.w;
                    if ((li & 15) == 0 && li > 0) L.win[pos - 1] = val.x;   // duplicate across the pad
                }
                wave_sync();
                const float dsL = rl_f(ds, laL), dsR = rl_f(ds, laR);
                if (fg != 1.0f)
                    mix_source_lds<FULL, true>(L, j, lane, acc, fbase, frame0, n_frames, fg, g0L, dgL, dsL, g0R, dgR, dsR);
                else
                    mix_source_lds<FULL, false>(L, j, lane, acc, fbase, frame0, n_frames, fg, g0L, dgL, dsL, g0R, dgR, dsR);
            } else if (path_j == PATH_GENERIC) {
                const uint64_t cp = ((uint64_t)(uint32_t)rl_i((int)((uint64_t)ss.clip >> 32), laL) << 32) |
                                    (uint64_t)(uint32_t)rl_i((int)((uint64_t)ss.clip & 0xffffffffu), laL);
                const uint32_t clen = (uint32_t)rl_i((int)ss.clip_len, laL), crate = (uint32_t)rl_i((int)ss.clip_rate, laL);
                const long long tLb = __double_as_longlong(ep.t_ear);
                const double tL = __longlong_as_double(((long long)rl_i((int)(tLb >> 32), laL) << 32) | (long long)(uint32_t)rl_i((int)(tLb & 0xffffffff), laL));
                const double tR = __longlong_as_double(((long long)rl_i((int)(tLb >> 32), laR) << 32) | (long long)(uint32_t)rl_i((int)(tLb & 0xffffffff), laR));
                mix_source_generic<FULL>(lane, acc, fbase, frame0, n_frames, tile, (const float*)cp, clen, crate, fg,
                                         tL, dtL, g0L, dgL, tR, dtR, g0R, dgR);
            } else {
                const float fv = rl_f(ss.freq_or_value, laL);
                const float phL = rl_f(ep.phase_ear, laL), phR = rl_f(ep.phase_ear, laR);
                if (path_j == PATH_SINE)
                    mix_source_analytic<FULL, true>(lane, acc, fbase, frame0, n_frames, tile, fv, fg, phL, dtL, g0L, dgL, phR, dtR, g0R, dgR);
                else
                    mix_source_analytic<FULL, false>(lane, acc, fbase, frame0, n_frames, tile, fv, fg, phL, dtL, g0L, dgL, phR, dtR, g0R, dgR);
            }
        }
        wave_sync();   // before the next group's phase A overwrites ckpt/cinfo
    }

    // this lane's 16 frames x 2 ears are 32 consecutive floats of the interleaved partial tile
    float4* dst = reinterpret_cast<float4*>(partials + ((size_t)tile * gridDim.x + wave) * (2 * TILE_FRAMES) + 32 * lane);
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
}

// ---------------------------------------------------------------------------------------------
// reduce: out[f][e] = sum over waves (fixed order) ; then Reinhard / Tanh
// grid = (ceil(2*n_frames / 64)), block = 1024 = 64 outputs x 16 segments
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float postfx_apply(float x, int postfx) {
    if (postfx == 1) return x / (1.0f + fabsf(x));   // reinhard.rs:32
    if (postfx == 2) return tanhf(x);                // tanh.rs:26
    return x;
}

__global__ __launch_bounds__(1024) void reduce_partials(const float* __restrict__ partials, float* __restrict__ out,
                                                        uint32_t n_waves, uint32_t n_frames, int postfx) {
    __shared__ float red[16][64];
    const uint32_t ox = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const uint32_t o = blockIdx.x * 64 + ox;          // flat interleaved output index
    const uint32_t n_out = 2 * n_frames;
    const uint32_t tile = o / (2 * TILE_FRAMES), within = o % (2 * TILE_FRAMES);
    float s = 0.0f;
    if (o < n_out) {
        const float* p = partials + (size_t)tile * n_waves * (2 * TILE_FRAMES) + within;
        bool first = true;
        for (uint32_t w = seg; w < n_waves; w += 16) {
            const float v = p[(size_t)w * (2 * TILE_FRAMES)];
            s = first ? v : s + v;
            first = false;
        }
    }
    red[seg][ox] = s;
    __syncthreads();
    if (seg == 0 && o < n_out) {
        // fixed-order combine; segments beyond n_waves hold exact zeros
        float t = red[0][ox];
        const uint32_t nseg = n_waves < 16 ? n_waves : 16;
        for (uint32_t k = 1; k < nseg; ++k) t = t + red[k][ox];
        out[o] = postfx_apply(t, postfx);
    }
}

__global__ void postfx_kernel(float* __restrict__ buf, uint32_t n, int postfx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = postfx_apply(buf[i], postfx);
}

__global__ void zero_kernel(float* __restrict__ buf, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// control-plane helpers (device side of set.rs / swap.rs semantics)
// ---------------------------------------------------------------------------------------------
struct MotionUpdate { uint32_t slot; float pos[3]; float vel[3]; uint32_t discontinuity; };

__global__ void apply_motion_updates(const MotionUpdate* __restrict__ up, uint32_t n, SrcPending* __restrict__ pend) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const MotionUpdate u = up[i];
    SrcPending p;
    p.pos[0] = u.pos[0]; p.pos[1] = u.pos[1]; p.pos[2] = u.pos[2];
    p.vel[0] = u.vel[0]; p.vel[1] = u.vel[1]; p.vel[2] = u.vel[2];
    p.flags = PEND_FRESH | (u.discontinuity ? PEND_DISCONTINUITY : 0u);
    p.pad = 0;
    pend[u.slot] = p;
}

// swap_remove moves (set.rs:183-188): dst <- src, all pairs independent (host resolves chains)
struct SlotMove { uint32_t dst, src; };

__global__ void apply_slot_moves(const SlotMove* __restrict__ mv, uint32_t n, SrcStatic* st, SrcDyn* dyn, SrcPending* pend) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SlotMove m = mv[i];
    st[m.dst] = st[m.src];
    dyn[m.dst] = dyn[m.src];
    pend[m.dst] = pend[m.src];
}

__global__ void seek_all_kernel(SrcDyn* __restrict__ dyn, const SrcStatic* __restrict__ st, uint32_t n, float seconds) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (st[i].kind == KIND_FRAMES) dyn[i].t = dyn[i].t + (double)seconds;                              // frames.rs:211-213
    else if (st[i].kind == KIND_SINE) dyn[i].phase = fmodf(dyn[i].phase + seconds * st[i].freq_or_value, ODDIO_TAU);
}

}  // namespace oddio_hip
