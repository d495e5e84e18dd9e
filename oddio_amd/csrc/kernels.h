// kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels of the SpatialScene hot path.
//
// Compiled with -ffp-contract=off and correctly rounded f32 divide/sqrt: every f32/f64 operation
// is a separately rounded IEEE op in the reference's order, so per-source contributions are
// bit-identical to the reference CPU `Signal::sample()` for FramesSignal/Constant sources
// (Sine differs only by the device sinf vs glibc sinf, a few ulp).
//
// Kernels
//   spatial_prepass   1 thread / source (table rows moved through LDS a wavefront at a time)
//                                         walk_set + EarState + cursor bookkeeping + the tile records
//                                         (spatial.rs:191-265, :445-469 scalar part, :501-549; frames.rs:176-201 set-up)
//   tile_records      the tile records of the later passes of a callback longer than REC_TILES tiles
//   cycle_sources     1 wave / Seek-set Cycle source: serial cursor scan + 64-lane render (cycle.rs:26-60)
//   spatial_mix       2-wave workgroups, 16 sources per phase-A group; the per-sample loop
//                                         (spatial.rs:456-463 + frames.rs:176-201 + sine.rs:34-40);
//                                         <.., STORE>: per-source contribution rows for ORDERED mode at scale
//   reduce_partials   fixed-order sum of the workgroup partial tiles + Reinhard/Tanh epilogue
//                                         (reinhard.rs:32, tanh.rs:26); set_kernels.h adds the set compaction
//   ordered_sum       the reference's sequential sum over the contribution rows (spatial.rs:204,460):
//                                         a loader wave (HBM -> LDS ring) and an adder wave (DPP add chain) per column block
//
// Mix kernel work decomposition (why it is not "one lane = one output frame"):
//   FramesSignal's slow path advances its f32 cursor by a *sequentially rounded* `offset += ds`
//   (frames.rs:189-196) restarted from the f64 clock every <=256-frame chunk (spatial.rs:456).
//   A closed form offset0 + k*ds is not within tolerance, so the running sum must
//   be reproduced exactly.  A tile is 512 output frames (two 256-frame chunks).
//   The walk kernel leaves one TileRec per (source, tile): window descriptor, per-ear {ds, g0, dg, wrel}, per-chunk
//   start offsets -- everything of frames.rs:176-201 that needs f64 or does not depend on the output frame.
//   Phase A: 64 lanes = 16 sources x 2 ears x 2 chunks each take their stream's record words (loaded one group ahead),
//   run the exact 255-step f32 scan once and leave 16 checkpoints (every 16 frames) + {4*wrel, g0, dg, ds} in a
//   20-word LDS block per stream.
//   Phase B: for one source at a time, lanes 0-31 are the left ear and 32-63 the right ear; lane l
//   owns 16 consecutive output frames of its ear (16 register accumulators), restarts from its
//   checkpoint and replays 15 exact adds.  The source's sample window (~N*ds + 32 floats) is
//   fetched one source ahead with bounds-checked 16 B buffer loads straight into LDS (out-of-clip
//   reads return 0, so frames.rs:105-123 `get_pair` needs no branches) and read back as
//   ds_read2_b32 pairs.  Windows with |ds-1| < PAD_EPS use a layout with one pad float per 16
//   samples so that the near-unit lane stride does not alias LDS banks.
//   The waves of a workgroup sum their register tiles through LDS in fixed order and write one
//   planar partial tile per workgroup; DESIGN.md section 4 has the byte and cycle accounting.
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

#ifdef ODDIO_HIP_BOUNDS
__device__ __noinline__ void bounds_fail(uint32_t* err, uint32_t code, uint32_t value, uint32_t src) {
    if (err == nullptr) return;
    if (atomicAdd(err, 1u) == 0u) { err[1] = code; err[2] = value; err[3] = src; }
}
#endif

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

// A per-source soft clip around the signal (SrcStatic::fx): Reinhard<T>::sample `x / (1 + |x|)` (reinhard.rs:28-35) or
// Tanh<T>::sample `tanh(x)` (tanh.rs:22-29) applied to every sample the inner signal produced, with the source's FixedGain
// (gain.rs:32-37) inside the clip or, FX_CLIP_FIRST, outside it.  (x * 1.0 == x: a source without FixedGain carries 1.0.)
// (registers only: the four ops packed in one word, the gains in four scalars, every index a literal -- a struct indexed by a loop
// counter would live in scratch, and the mix kernels that call the rare paths have none)
struct FxRegs { uint32_t n, ops; float g0, g1, g2, g3; };
__device__ __forceinline__ FxRegs load_fx_chain(const SrcStatic& s, const SrcStatic* __restrict__ st) {
    FxRegs r = {0u, 0u, 1.0f, 1.0f, 1.0f, 1.0f};
    if (s.fx & FX_CHAIN) {
        const FxChain* c = reinterpret_cast<const StaticHeader*>(st - 1)->fx_chains + __float_as_uint(s.fixed_gain);   // (device_types.h: the header before slot 0)
        const uint2 h = *reinterpret_cast<const uint2*>(c);                 // {n, op[0..3]}
        const float4 g = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(c) + 8);
        r.n = h.x; r.ops = h.y; r.g0 = g.x; r.g1 = g.y; r.g2 = g.z; r.g3 = g.w;
    }
    return r;
}
__device__ __forceinline__ float fx_op(float v, uint32_t op, float g) {
    if (op == FXOP_FIXED_GAIN) return v * g;                                // gain.rs:32-37
    if (op == FXOP_REINHARD) return v / (1.0f + fabsf(v));                  // reinhard.rs:32
    return tanhf(v);                                                        // tanh.rs:26
}
__device__ __forceinline__ float apply_fx(float v, float fixed_gain, int fx, const FxRegs& ch) {
    if (fx & FX_CHAIN) {     // a general nest of Seek wrappers, innermost first (device_types.h)
        if (ch.n > 0u) v = fx_op(v, ch.ops & 255u, ch.g0);
        if (ch.n > 1u) v = fx_op(v, (ch.ops >> 8) & 255u, ch.g1);
        if (ch.n > 2u) v = fx_op(v, (ch.ops >> 16) & 255u, ch.g2);
        if (ch.n > 3u) v = fx_op(v, ch.ops >> 24, ch.g3);
        return v;
    }
    if (!(fx & FX_CLIP_FIRST)) v = v * fixed_gain;
    if (fx & FX_REINHARD) v = v / (1.0f + fabsf(v));
    else if (fx & FX_TANH) v = tanhf(v);
    if (fx & FX_CLIP_FIRST) v = v * fixed_gain;
    return v;
}

// ---------------------------------------------------------------------------------------------
// prepass: one thread per live slot
// ---------------------------------------------------------------------------------------------
// `d_len` is the device-resident set length (set_kernels.h); `len_snap` receives the length this walk saw
// (the mix kernel of the callback reads it; the set is compacted at the end of the callback).
__device__ __forceinline__ void prepass_source(const SceneParams& P, const uint32_t i, SrcDyn& d, const SrcStatic& s,
                                               SrcPending* __restrict__ pend, EarParams& e0, EarParams& e1,
                                               uint32_t* __restrict__ stopped_hdr, uint32_t stopped_cap, int check_pending) {
    e0 = EarParams{}; e1 = EarParams{};
    if (d.flags & DYN_STOPPED) {  // removed earlier, compaction not applied yet: never mixed again
        e0.flags = EAR_SKIP; e1.flags = EAR_SKIP;
        return;
    }
    const float elapsed = P.elapsed;
    const float nf = (float)P.n_frames;
    V3 tpos = {d.tgt_pos[0], d.tgt_pos[1], d.tgt_pos[2]};
    V3 tvel = {d.tgt_vel[0], d.tgt_vel[1], d.tgt_vel[2]};
    V3 ppos = {d.prev_pos[0], d.prev_pos[1], d.prev_pos[2]};
    // spatial.rs:216-226 motion.refresh()
    // (the host passes check_pending == 0 when no set_motion has been flushed since every pending
    // slot was last consumed: saves 32 B/source of reads)
    if (check_pending) {
        const SrcPending pm = pend[i];
        if (pm.flags & PEND_FRESH) {
            V3 npos = {pm.pos[0], pm.pos[1], pm.pos[2]};
            V3 nvel = {pm.vel[0], pm.vel[1], pm.vel[2]};
            ppos = (pm.flags & PEND_DISCONTINUITY) ? npos : smoothed_position(ppos, d.state_dt, 0.0f, tpos, tvel);
            tpos = npos; tvel = nvel;
            d.state_dt = 0.0f;
            pend[i].flags = 0;
        }
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
        if (s.kind == KIND_FRAMES || s.kind == KIND_DOWNMIX) fin = d.t >= (double)(s.clip_len - 1u) / (double)s.clip_rate;     // frames.rs:204-206
        if (fin) { d.flags |= DYN_HAS_FINISHED_FOR; d.finished_for = elapsed; }
    }
    if (d.flags & DYN_STOPPED) {
        const uint32_t k = atomicAdd(&stopped_hdr[0], 1u);
        if (k < stopped_cap) stopped_hdr[1 + k] = d.id;
        e0.flags = EAR_SKIP; e1.flags = EAR_SKIP;
        return;
    }

    // spatial.rs:446-468: the scalar part of mix_signal; the sampling itself is the mix kernel's.
    const uint32_t n = P.n_frames;
#pragma unroll
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
        if (s.kind == KIND_FRAMES || s.kind == KIND_DOWNMIX) {
            d.t = d.t + (double)off0;                   // seek(prev_state.offset), frames.rs:211-213
            ep.t_ear = d.t;
            for (uint32_t done = 0; done < n; done += 256u) {
                // Downmix::sample always renders its whole 256-frame buffer (downmix.rs:24-29), so the
                // inner clock of a Downmix source advances 256 frames even for a short last chunk
                const uint32_t len = (s.kind == KIND_DOWNMIX || (n - done) >= 256u) ? 256u : (n - done);
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
        } else if (s.kind == KIND_CYCLE) {
            ep.phase_ear = off0;                        // cycle_sources replays the seeks around the chunks
            ep.t_ear = (double)eff;
        }
        if (e == 0) e0 = ep; else e1 = ep;
    }
    if (s.kind == KIND_FRAMES || s.kind == KIND_DOWNMIX) d.t = d.t + (double)elapsed;           // :468
    else if (s.kind == KIND_SINE) d.phase = fmodf(d.phase + elapsed * s.freq_or_value, ODDIO_TAU);
}

__device__ __forceinline__ double f64_rem_euclid(double a, double b) {   // core f64::rem_euclid
    const double r = fmod(a, b);
    return r < 0.0 ? r + fabs(b) : r;
}

// ---------------------------------------------------------------------------------------------
// mix kernel
// ---------------------------------------------------------------------------------------------
// Shape (measured on MI355X: tools/ubench/valu_rate.hip, hbm_ceiling.hip and the PMC passes under
// profiles/): one wave issues a VALU op every ~4.6 cycles and two always-ready waves saturate a
// SIMD (~2.4 cycles per wave64 op); v_pk_*_f32 costs two scalar ops; the LDS serves a 32-lane
// group with ~2-way bank conflicts whatever the layout when the lanes' runs are 16*ds samples
// apart; HBM delivers ~5.6 TB/s for 2.3 KB windows at random places.  All three are within ~25 %
// of each other for this loop, so the kernel is built to keep all of them busy at once:
//   * a wave renders a 512-frame tile (2 chunks of 256, spatial.rs:393) of its sources: lanes
//     0-31 are the left ear, 32-63 the right ear; lane (e, c, b) owns the 16 consecutive frames
//     256c+16b.. of ear e => 16 register accumulators;
//   * phase A handles 16 sources at a time: lane (j, e, c) runs the exact f32 cursor scan of
//     source j / ear e / chunk c and leaves 16 checkpoints in LDS;
//   * the source's sample window goes HBM -> LDS directly (buffer_load ... lds, 16 B per lane,
//     bounds-checked descriptor: hardware zero fill == frames.rs:105-123 out-of-range rule) into
//     one of two window buffers, one source ahead of the compute, with no register staging;
//   * the per-sample LDS pair reads are software-pipelined by hand (MIX_DEPTH samples in flight),
//     so a wave computes on sample i while the pairs of samples i+1.. are on their way;
//   * workgroups are MIX_WG_WAVES independent waves (own LDS slice, wave-local ordering only);
//     their accumulators are summed through LDS in fixed order before one partial tile is written.
#ifndef ODDIO_MIX_WG_WAVES
#define ODDIO_MIX_WG_WAVES 2
#endif
#ifndef ODDIO_MIX_WAVES
#define ODDIO_MIX_WAVES 4
#endif
#ifndef ODDIO_MIX_DEPTH
#define ODDIO_MIX_DEPTH 1
#endif
#ifndef ODDIO_WIN_POLICY
#define ODDIO_WIN_POLICY " nt"    // cache policy of the window loads: streaming -- every sample is read once per callback (same box, spatial_mix:
                                 // default policy 0.2343 / 0.2284 ms, nt 0.2253 / 0.2265, sc1 0.2337 / 0.2334; profiles/r03_ab_window_policy.txt)
#endif

constexpr int MIX_WG_WAVES = ODDIO_MIX_WG_WAVES;       // waves per workgroup
#ifndef ODDIO_ORD_POLICY
#define ODDIO_ORD_POLICY ""     // cache policy of ordered_sum's row loads (" nt": streaming); measured in round 6
#endif
#ifndef ODDIO_ROWS_NT
#define ODDIO_ROWS_NT 0      // 1: ORDERED's contribution rows leave as streaming (nt) stores -- measured in round 6
#endif
#ifndef ODDIO_DIAG
#define ODDIO_DIAG 0   // diagnostic builds only (tools/ubench, profiles/r05_exp_*): 1 conflict-free (wrong) LDS addresses, 2 no sample loop, 4 no cursor scan, 8 no window DMA
#endif
constexpr int MIX_WAVES_PER_SIMD = ODDIO_MIX_WAVES;    // register budget: 512 / this
constexpr int MIX_DEPTH = ODDIO_MIX_DEPTH;             // samples whose LDS pair reads are in flight together
constexpr int MIX_WAVES_PER_CU = 16;                   // default grid size (waves per CU)
constexpr int MIX_GROUP = 16;                // sources per phase-A step (16 x 2 ears x 2 chunks = 64 lanes)
constexpr int TILE_FRAMES = 512;             // frames per (wave, tile) pass
constexpr int TILE_CHUNKS = TILE_FRAMES / 256;
// Workgroup partial sums: the callback's frames in blocks of PART_FRAMES; block fb holds every workgroup's [ear][PART_FRAMES] values
// next to each other -- partials[(fb * n_wgs + wg) * PART_BLOCK + ear * PART_FRAMES + k] -- so that the reduce block of those
// frames reads one contiguous stretch of n_wgs * 64 bytes.  (Rounds 1-4 kept one planar [ear][512] tile per workgroup: the
// reduce then took 32 bytes from each of n_wgs rows 4 KB apart, a quarter of every line it touched -- 9.4 us for 1024 tiles,
// 14 us for the 2048 of spatial_mix_pair.)
constexpr int PART_FRAMES = 8;
constexpr int PART_BLOCK = 2 * PART_FRAMES;
constexpr int PART_STRIDE = 2 * TILE_FRAMES;   // floats of partial sums per workgroup and tile (allocation sizes)
constexpr int WIN_CAP = 608;                 // samples staged per source and tile (ds <= ~1.11)
constexpr int WIN_PIECES = (WIN_CAP * 4 + 1023) / 1024;   // 1 KiB DMA pieces covering a window buffer
#ifndef ODDIO_PAD_EPS
#define ODDIO_PAD_EPS 0.004f
#endif
constexpr float PAD_EPS = ODDIO_PAD_EPS;           // |ds - 1| below this: lanes' runs sit 16 samples apart -> padded layout
// Windows larger than the LDS stage (resample ratios above ~1.11: a 96 or 192 kHz clip in a 48 kHz scene, strong Doppler) are
// staged in `npass` sub-windows of WIN_CAP samples, MULTI_STRIDE apart; a lane renders its 16 frames in the pass whose
// sub-window holds its whole run (16 * ds + 2 samples <= the overlap WIN_CAP - MULTI_STRIDE).  TileRec::info bits 24-26: npass.
constexpr int MULTI_STRIDE = 532;
constexpr float MULTI_DS_MAX = 4.5f;
constexpr int MULTI_PASS_MAX = 7;
constexpr float MULTI_DS_MAX_STEREO = 2.2f;      // interleaved stereo windows (Downmix): a lane's run is 2 * (16 * ds + 2) floats
static_assert(WIN_CAP - MULTI_STRIDE >= 2 * (16 * 2.2f + 2) + 1, "a stereo lane's run fits the overlap of two sub-windows");
static_assert(MULTI_STRIDE % 4 == 0 && WIN_CAP - MULTI_STRIDE >= 16 * 4.5f + 3, "a lane's run fits the overlap of two sub-windows");

enum : int { PATH_SKIP = 0, PATH_LDS = 1, PATH_GENERIC = 2, PATH_SINE = 3, PATH_CONST = 4, PATH_ROW = 5, PATH_SINE_INLINE = 6 };

// LDS map of one wave (bytes)
//   WIN0/WIN1 two window buffers.  General sources: plain (sample s at 4*s), pairs read with
//         ds_read2_b32.  Sources whose resample ratio is within PAD_EPS of 1 (and the
//         |ds-1| <= EPSILON fast path) are re-laid in place with one pad float per 16 samples
//         (slot s + s/16; the pad repeats the next sample), otherwise the lanes' 16-frame runs,
//         16 samples apart, would all hit two banks.
//   STREAM 64 blocks of 20 words, one per phase-A stream (source j, ear e, chunk c) = lane 4j + 2e + c:
//         words 0-15 the cursor checkpoints (checkpoint 0 == the chunk's start offset frac0),
//         words 16-19 {4 * wrel, g0, dg, ds}: what a phase-B lane of that stream needs besides its checkpoint.
constexpr int LDS_WIN0 = 0;
constexpr int WIN_BYTES = WIN_CAP * 4;
constexpr int LDS_WIN1 = LDS_WIN0 + WIN_BYTES;
constexpr int LDS_STREAM = LDS_WIN1 + WIN_BYTES;
constexpr int STREAM_WORDS = 20;
constexpr int LDS_TOTAL = LDS_STREAM + 64 * STREAM_WORDS * 4;
constexpr int SFLAG_NEG = 1, SFLAG_FAST_L = 2, SFLAG_FAST_R = 4, SFLAG_PAD = 8, SFLAG_FG = 16;
// RING records (the ring reads of the buffered set, buffered_fast.h): a cursor of the tile may reach the ring's end
// (ring.rs:66-74's rewrite); shares the bit of SFLAG_FG, which a ring record never carries
constexpr int SFLAG_WRAP = 16;
static_assert(WIN_BYTES % 16 == 0 && LDS_TOTAL % 16 == 0, "per-wave LDS slices and window buffers stay 16-byte aligned");
static_assert(LDS_TOTAL <= 10240, "16 waves per CU need <= 10 KB of LDS each");
// accumulators parked for an out-of-line source: slot i of lane l at word i * PARK_STRIDE + l (65, not 64: the exact per-lane path
// walks the frames lane-major -- adjacent lanes on adjacent frames, for its global loads -- and so reaches the slots of one
// lane from 16 lanes at once: 64 apart they would share a bank)
constexpr int PARK_STRIDE = 65;
static_assert(LDS_STREAM >= 16 * PARK_STRIDE * 4, "accumulator parking / cross-wave reduction use ~4 KB at offset 0 and must not reach the stream blocks");
static_assert(WIN_PIECES * 1024 <= 4095 + 1024, "the DMA's 12-bit instruction offset reaches every piece");

__device__ __forceinline__ void wave_sync() {
    // Waves never share window/checkpoint data: the LDS pipeline executes one wave's DS operations
    // in issue order, so a same-wave write -> read hand-off needs no s_barrier, only a compiler
    // fence.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float rl_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// ---------------------------------------------------------------------------------------------
// Cycle in the Seek set (cycle.rs:26-60 inside spatial.rs:446-468), one wave per slot.
// Cycle::sample leaves `cursor = base + offset` with the f32-accumulated offset, and the scene
// seeks it back and forth between the ears, so every chunk's start depends on the rounding of all
// the chunks before it (left ear first).  Lane 0 replays that cursor arithmetic alone (no memory
// reads) and drops a (base, offset) checkpoint every 16 frames; the 64 lanes then restart from
// their checkpoints with the same step function and render 16 frames each.  The wave writes the
// source's finished contribution s * gain (spatial.rs:459-460) to its row; spatial_mix adds the row
// at the source's place in the set walk, so ORDERED mode stays bit-exact.
// ---------------------------------------------------------------------------------------------
// one frame of cycle.rs:30-50: the pair to interpolate and its fraction; advances (base, offset)
// (`offset as usize`: negative and NaN -> 0.  Clips are shorter than 2^30 samples (oddio_hip_scene_play_cycle) and
// offset stays below len + 256 * ds, so base + trunc fits 32 bits: one v_cvt_u32_f32 and 32-bit compares instead of
// 64-bit sequences in the 2048-step serial scan.  The clamp at 2^31 is never reached.)
__device__ __forceinline__ uint32_t f32_as_index(float x) { return (uint32_t)fminf(fmaxf(x, 0.0f), 2147483648.0f); }
__device__ __forceinline__ void cycle_step(uint32_t& base, float& offset, uint32_t len, float ds, uint32_t& ia, uint32_t& ib, float& fract) {
    const uint32_t trunc = f32_as_index(offset);
    fract = offset - (float)trunc;
    const uint32_t x = base + trunc;
    if (x < len - 1u) { ia = x; ib = x + 1u; }
    else if (x < len) { ia = x; ib = 0u; }
    else {
        base = 0u;
        offset = (float)(x % len) + fract;
        const uint32_t x2 = f32_as_index(offset);
        if (x2 < len - 1u) { ia = x2; ib = x2 + 1u; } else { ia = x2; ib = 0u; }
    }
    offset = offset + ds;
}

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // four floats at any 4-byte aligned address

// The Seek-set Cycle sources of a callback, two kernels over the list the walk makes of them (cycle_list: [par] = their number
// this callback -- par = callback parity, the walk zeroes the other counter for the next one -- [2..] their slots):
//   cycle_scan    one LANE per source: the serial cursor arithmetic of both ears and every chunk (no memory reads), 64 sources
//                 per wavefront, a (base, offset) checkpoint every 16 frames; advances the cursor.  (Rounds 2-3 ran this scan in
//                 lane 0 of a wave per source: 45 us of dependent arithmetic per source and wave slot.)
//   cycle_render  one WAVE per source: the 64 lanes restart from their checkpoints with the same step function and render 16
//                 frames each into the source's row.
struct CycleCk { uint32_t base; float offset; };
__device__ __forceinline__ int4 window_desc(const float* clip, int clip_len4, int ws, int nvec);
// `lanes`: lanes of a wavefront that carry a source.  The scan is one serial chain per source, and a block of 16 steps in which
// ANY lane's cursor reaches its clip's end sends the whole wave through the step-by-step branch (5000-sample clips, 64 sources
// per wave: a fifth of all blocks, 44 us per callback); small sets therefore spread over more waves.  (Round 6, 65 536 sources of
// 5000-sample loops: 4096 waves of 16 sources are bound by the SIMDs' issue rate with a quarter of their lanes working -- a Cycle
// callback takes 0.229 ms with 16 sources per wave, 0.208 ms with 32, 0.217 ms with 64, 0.261 ms with 8.)
// Round 4: the scan also knows, per (ear, 256-frame chunk), where the cursor starts and ends -- a tile whose four streams never
// reach the clip's last sample is, bit for bit, a FramesSignal tile (cycle.rs:30-36 == frames.rs:188-196 while x < len - 1): it gets
// a PATH_LDS record and spatial_mix renders it from the staged window like any clip; only tiles that touch the clip's end (and
// callbacks longer than the record tiles) keep the row path, and `rlist` ([par] counter, [2..] slot | pass-0 flag << 31) names the
// sources cycle_render has rows to make for.
__global__ __launch_bounds__(64) void cycle_scan(SceneParams P, const SrcStatic* __restrict__ st, SrcDyn* __restrict__ dyn,
                                                 const EarParams* __restrict__ ear, const uint32_t* __restrict__ list, uint32_t par,
                                                 CycleCk* __restrict__ ck, uint32_t ck_stride, uint32_t lanes, TileRec* __restrict__ recs,
                                                 uint32_t rec_stride, uint32_t n_rec_tiles, uint32_t* __restrict__ rlist) {
    if (threadIdx.x >= lanes) return;
    const uint32_t q = blockIdx.x * lanes + threadIdx.x;
    if (q >= list[par]) return;
    const uint32_t i = list[2u + q];
    const SrcStatic s = st[i];
    const EarParams e0 = ear[2 * i], e1 = ear[2 * i + 1];
    double cursor = dyn[i].t;
    const double rate = (double)s.clip_rate, lenf = (double)s.clip_len;
    const uint32_t len = s.clip_len;
    const uint32_t n = P.n_frames;
    const uint32_t row = __float_as_uint(s.freq_or_value);
    constexpr int RC = REC_TILES * TILE_CHUNKS;          // chunks the tile records cover
    uint32_t sb[2][RC] = {}, shi[2][RC] = {};             // per (ear, chunk): the base the chunk starts from, the last index it reads
    float so[2][RC] = {};                                 //                   the offset it starts from
    uint32_t nonlin = 0u;                                 // bit e * RC + c: the chunk reads the clip's last sample or wraps
    for (int e = 0; e < 2; ++e) {
        const EarParams ep = e ? e1 : e0;
        const float off0 = ep.phase_ear, eff = (float)ep.t_ear;
        const float ds = ep.dt * (float)s.clip_rate;                              // cycle.rs:27
        CycleCk* c = ck + ((size_t)row * 2u + (uint32_t)e) * ck_stride;
        cursor = f64_rem_euclid(cursor + (double)off0 * rate, lenf);              // spatial.rs:449 -> cycle.rs:57-60
        for (uint32_t done = 0; done < n; done += 256u) {                         // spatial.rs:456
            const uint32_t len_c = (n - done) < 256u ? (n - done) : 256u;
            uint32_t base = (uint32_t)f64_as_isize(cursor);                       // cycle.rs:28 (0 <= cursor < len + 256 * ds)
            float offset = (float)(cursor - (double)base);                        // :29
            const uint32_t cidx = done >> 8;
            const uint32_t base0 = base; const float offset0 = offset;
            uint32_t last = base; bool hit = !(ds > 0.0f);
            // one block of up to 16 steps from frame k0 of the chunk, with its checkpoint
            auto block16 = [&](uint32_t k0) {
                CycleCk v; v.base = base; v.offset = offset;
                c[(done + k0) >> 4] = v;
                const uint32_t cnt = (len_c - k0) < 16u ? (len_c - k0) : 16u;
                uint32_t ia, ib; float fract;
                if (cnt == 16u) {
                    // 16 steps at once: unless the cursor reaches the clip's end inside them (checked on the offset the 16th
                    // step reads), the steps are 16 plain adds; the lanes that do wrap redo the block step by step
                    float o = offset, o15 = offset;
#pragma unroll
                    for (int k = 0; k < 16; ++k) { o15 = o; o = o + ds; }
                    const uint32_t xl = base + f32_as_index(o15);
                    const bool plain = ds > 0.0f && xl < len;
                    if (plain) { offset = o; last = xl; hit = hit || xl + 1u >= len; }
                    else {
                        hit = true;
#pragma unroll 1
                        for (int k = 0; k < 16; ++k) cycle_step(base, offset, len, ds, ia, ib, fract);
                    }
                } else {
                    // (the lanes of spatial_mix render whole blocks of 16 frames: the frames past the callback's end are
                    // discarded, but their reads happen -- the window covers them: 15 plain steps from the block's start)
                    float o15 = offset;
#pragma unroll
                    for (int k = 0; k < 15; ++k) o15 = o15 + ds;
                    const uint32_t reach = base + f32_as_index(o15);
                    for (uint32_t k = 0; k < cnt; ++k) {
                        last = base + f32_as_index(offset);
                        hit = hit || last + 1u >= len;
                        cycle_step(base, offset, len, ds, ia, ib, fract);
                    }
                    last = max(last, reach);
                }
            };
            uint32_t k0 = 0;
            // 64 steps at a time while the chunk has them: the chain of dependent adds IS this kernel (2048 per source: ~10 000
            // cycles at best), and the end-of-clip test, the loop and the checkpoint bookkeeping per 16 steps nearly doubled it.
            // Checked on the offset the 64th step reads; a lane that reaches the clip's end redoes the 64 steps in blocks of 16
            // (which rewrite the three checkpoints stored on the way).
            for (; k0 + 64u <= len_c; k0 += 64u) {
                float o = offset, o63 = offset;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    CycleCk v; v.base = base; v.offset = o;
                    c[((done + k0) >> 4) + (uint32_t)q] = v;
#pragma unroll
                    for (int k = 0; k < 16; ++k) { o63 = o; o = o + ds; }
                }
                const uint32_t xl = base + f32_as_index(o63);
                if (ds > 0.0f && xl < len) { offset = o; last = xl; hit = hit || xl + 1u >= len; }
                else {
#pragma unroll 1
                    for (uint32_t q = 0; q < 4u; ++q) block16(k0 + 16u * q);
                }
            }
#pragma unroll 1
            for (; k0 < len_c; k0 += 16u) block16(k0);
#pragma unroll
            for (int ee = 0; ee < 2; ++ee)
#pragma unroll
                for (int cc = 0; cc < RC; ++cc)
                    if (e == ee && cidx == (uint32_t)cc) { sb[ee][cc] = base0; so[ee][cc] = offset0; shi[ee][cc] = last; if (hit) nonlin |= 1u << (ee * RC + cc); }
            cursor = (double)base + (double)offset;                               // cycle.rs:52
        }
        cursor = f64_rem_euclid(cursor + (double)(-eff - off0) * rate, lenf);     // spatial.rs:465
    }
    cursor = f64_rem_euclid(cursor + (double)P.elapsed * rate, lenf);             // spatial.rs:468
    dyn[i].t = cursor;

    // ---- the tile records (the walk left PATH_ROW for every tile of a Cycle source) ----
    const float ds_e[2] = {e0.dt * (float)s.clip_rate, e1.dt * (float)s.clip_rate};
    bool rows_pass0 = false;
    uint32_t row_tiles = 0u;           // bit t: tile t of the record tiles takes the row path (cycle_render renders those tiles only)
#pragma unroll
    for (int t = 0; t < REC_TILES; ++t) {
        if ((uint32_t)t >= n_rec_tiles) break;
        TileRec r = {};
        r.info = PATH_ROW;
        uint32_t lo = 0xffffffffu, hi = 0u; bool lin = true, any = false; int fl = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (fabsf(ds_e[e] - 1.0f) < PAD_EPS) fl |= SFLAG_PAD;
#pragma unroll
            for (int c = 0; c < TILE_CHUNKS; ++c) {
                const int cc = t * TILE_CHUNKS + c;
                r.frac0[e][c] = so[e][cc];
                if ((uint32_t)cc * 256u < n) {
                    any = true;
                    lin = lin && !((nonlin >> (e * RC + cc)) & 1u);
                    lo = min(lo, sb[e][cc]); hi = max(hi, shi[e][cc]);
                }
            }
            const EarParams& ep = e ? e1 : e0;
            r.ear[e].ds = ds_e[e]; r.ear[e].g0 = ep.g0; r.ear[e].dg = ep.dg;
        }
        bool staged = any && lin && lo <= hi && !s.fx;   // (hi may lie past the clip: reads of discarded frames, zero-filled by the descriptor;
                                                         //  a soft-clipped Cycle keeps the row path: cycle_render applies the clip)
        if (staged) {
            const int ws = (int)(lo & ~3u);
            const int count = (int)hi + 2 - ws;
            const int vec_samples = ((count + 3) >> 2) << 2;
            if ((fl & SFLAG_PAD) && vec_samples + (vec_samples >> 4) + 1 > WIN_CAP) fl &= ~SFLAG_PAD;
            int npass = 1;
            if (count > WIN_CAP) {
                npass = (count - WIN_CAP + MULTI_STRIDE - 1) / MULTI_STRIDE + 1;
                staged = ds_e[0] <= MULTI_DS_MAX && ds_e[1] <= MULTI_DS_MAX && npass <= MULTI_PASS_MAX;
                fl &= ~SFLAG_PAD;
            }
            // (both ears' cursors on the same lap of the clip: otherwise the window would span the clip)
            staged = staged && (int)sb[0][t * TILE_CHUNKS] - ws <= 65535 && (int)sb[1][t * TILE_CHUNKS] - ws <= 65535 && (int)sb[0][t * TILE_CHUNKS + 1] - ws <= 65535 &&
                     (int)sb[1][t * TILE_CHUNKS + 1] - ws <= 65535;
            if (staged) {
                if (s.fixed_gain != 1.0f) fl |= SFLAG_FG;
                const int nvec_all = (count + 3) >> 2;
                const int nvec = npass > 1 ? WIN_CAP / 4 : nvec_all;
                const int4 d = window_desc(s.clip, (int)((s.clip_len + 3u) & ~3u), ws, nvec_all);
                r.desc[0] = (uint32_t)d.x; r.desc[1] = (uint32_t)d.y; r.desc[2] = (uint32_t)d.z;
                (void)ODDIO_BOUNDS_CHECK(P.bounds_err, nvec >= 1 && nvec * 4 <= WIN_CAP && d.z >= 0 && d.z <= nvec_all * 16 && d.w == 0, BOUNDS_RECORD, nvec, t);
                r.info = (uint32_t)PATH_LDS | ((uint32_t)fl << 3) | ((uint32_t)nvec << 8) | ((uint32_t)(npass > 1 ? npass : 0) << 24);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int w0 = min(max((int)sb[e][t * TILE_CHUNKS] - ws, 0), 65535), w1 = min(max((int)sb[e][t * TILE_CHUNKS + 1] - ws, 0), 65535);
                    r.ear[e].wrel = (uint32_t)w0 | ((uint32_t)w1 << 16);
                }
            }
        }
        if (!staged) { rows_pass0 = true; row_tiles |= 1u << t; }
        recs[(size_t)t * rec_stride + i] = r;
    }
    if (rows_pass0 || n > (uint32_t)(REC_TILES * TILE_FRAMES)) {
        const uint32_t k = atomicAdd(&rlist[par], 1u);
        rlist[2u + k] = i | (rows_pass0 ? (0x80000000u | (row_tiles << 29)) : 0u);      // (slots below 2^29: oddio_hip_scene_play_cycle)
    }
}

static_assert(REC_TILES * TILE_FRAMES == 1024, "cycle_render's pass 0 is exactly the tiles cycle_scan writes records for");
constexpr int CYCLE_WAVES = 4;   // sources per workgroup (independent waves)
constexpr int CYCLE_WIN_CAP = 1280;   // samples staged per source, ear and 1024-frame pass (ds <= ~1.24)
__global__ __launch_bounds__(64 * CYCLE_WAVES) void cycle_render(SceneParams P, const SrcStatic* __restrict__ st, const EarParams* __restrict__ ear,
                                                                const uint32_t* __restrict__ rlist, uint32_t par,
                                                                const CycleCk* __restrict__ ck, uint32_t ck_stride) {
    __shared__ float win_all[CYCLE_WAVES][CYCLE_WIN_CAP];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* win = win_all[wv];
    const uint32_t q = blockIdx.x * CYCLE_WAVES + (uint32_t)wv;
    if (q >= rlist[par]) return;
    const uint32_t entry = rlist[2u + q];
    const uint32_t i = entry & 0x1fffffffu;
    const uint32_t first_pass = (entry >> 31) ? 0u : (uint32_t)(REC_TILES * TILE_FRAMES);    // (pass 0 = the record tiles: rows only if cycle_scan asked for them)
    // Round 6: ... and only for the tiles that take the row path (bits 29-30).  Most sources of a callback that touch their loop's end do so in
    // ONE tile; the other is staged like a clip's.  The kernel is bound by its bytes (the window in, 4 KB of row per ear and tile out).
    const uint32_t row_tiles = (entry >> 29) & 3u;
    static_assert(REC_TILES == 2 && TILE_FRAMES == 512, "two record tiles of 32 lanes each");
    const SrcStatic s = st[i];
    const FxRegs fxch = load_fx_chain(s, st);
    const uint32_t len = s.clip_len;
    const uint32_t n = P.n_frames;
    const uint32_t rowi = __float_as_uint(s.freq_or_value);
    float* row = P.cycle_rows + (size_t)rowi * 2u * P.cycle_plane;
    for (int e = 0; e < 2; ++e) {
        const EarParams ep = ear[2 * i + e];
        float* plane = row + (size_t)e * P.cycle_plane;
        const CycleCk* c = ck + ((size_t)rowi * 2u + (uint32_t)e) * ck_stride;
        const float ds = ep.dt * (float)s.clip_rate;
        for (uint32_t pass0 = first_pass; pass0 < n; pass0 += 1024u) {
            const uint32_t m = (n - pass0) < 1024u ? (n - pass0) : 1024u;
            // The stretch of the clip this pass reads, staged in LDS with coalesced loads (every lane fetching its own 32
            // samples touched ~35 lines per load instruction: the kernel was bound by that).  It starts at the first frame's
            // index and is linear modulo the clip length; a pass that needs more than the stage, or more than one lap of the
            // clip, reads global memory directly.
            // frames [fa, fb) of the pass are rendered: all of a later pass; the row-path tiles of pass 0
            uint32_t fa = 0u, fb = m;
            if (pass0 == 0u) {
                if (!(row_tiles & 1u)) fa = (uint32_t)TILE_FRAMES < m ? (uint32_t)TILE_FRAMES : m;
                if (!(row_tiles & 2u)) fb = (uint32_t)TILE_FRAMES < m ? (uint32_t)TILE_FRAMES : m;
            }
            const CycleCk v0 = c[(pass0 + fa) >> 4];
            const float span = (float)(fb > fa ? fb - fa : 0u) * ds;
            const bool staged = ds > 0.0f && span * 1.0001f + 8.0f < (float)CYCLE_WIN_CAP && span * 1.0001f + 8.0f < (float)len;
            const uint32_t w_count = staged ? (uint32_t)(span * 1.0001f) + 8u : 0u;
            uint32_t w_start = v0.base + f32_as_index(v0.offset);
            w_start = w_start >= len ? w_start % len : w_start;
            if (staged) {
                for (uint32_t p = (uint32_t)lane; p < w_count; p += 64u) {
                    uint32_t idx = w_start + p;
                    idx = idx >= len ? idx - len : idx;
                    win[p] = s.clip[idx];
                }
            }
            wave_sync();
            const uint32_t f0 = pass0 + 16u * (uint32_t)lane;
            if (f0 < n && f0 - pass0 >= fa && f0 - pass0 < fb) {
                const uint32_t done = f0 & ~255u;
                const uint32_t len_c = (n - done) < 256u ? (n - done) : 256u;
                const uint32_t cnt = (len_c - (f0 - done)) < 16u ? (len_c - (f0 - done)) : 16u;
                const CycleCk v = c[f0 >> 4];
                uint32_t base = v.base;
                float offset = v.offset;
                // the 16 index pairs first (registers only), then all the loads together, then the arithmetic
                uint32_t ia[16], ib[16]; float fract[16], a[16], b[16];
                {
                    // (as in cycle_scan: when the cursor stays inside the clip for the lane's 16 frames, a step is an add, a
                    // truncation and the fraction -- decided per wave, so that the common case runs without the compares)
                    float o15 = offset;
#pragma unroll
                    for (int k = 0; k < 15; ++k) o15 = o15 + ds;
                    const bool plain = ds > 0.0f && base + f32_as_index(o15) < len;
                    if (__all(plain)) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const uint32_t tr = (uint32_t)offset;              // offset >= 0 here
                            fract[k] = offset - (float)tr;
                            ia[k] = base + tr;
                            ib[k] = ia[k] + 1u == len ? 0u : ia[k] + 1u;
                            offset = offset + ds;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 16; ++k) cycle_step(base, offset, len, ds, ia[k], ib[k], fract[k]);   // steps past cnt touch nothing
                    }
                }
                if (staged) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        // the stage is linear modulo the clip length: the pair's second sample (ia + 1, or 0 after len - 1) is the next slot
                        uint32_t pa = ia[k] >= w_start ? ia[k] - w_start : ia[k] + len - w_start;
                        pa = pa < w_count - 1u ? pa : w_count - 2u;     // (frames past cnt, not stored)
                        a[k] = win[pa];
                        b[k] = win[pa + 1u];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const bool on = (uint32_t)k < cnt;
                        a[k] = on ? s.clip[ia[k]] : 0.0f;
                        b[k] = on ? s.clip[ib[k]] : 0.0f;
                    }
                }
                float o[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    float vv = a[k] + fract[k] * (b[k] - a[k]);                   // frame::lerp
                    vv = apply_fx(vv, s.fixed_gain, s.fx, fxch);                  // FixedGain, gain.rs:32-37 (and the source's soft clip)
                    o[k] = vv * (ep.g0 + (float)(f0 + (uint32_t)k) * ep.dg);      // spatial.rs:459-460
                }
                if (cnt == 16u) {
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        f4u w = {o[4 * k4], o[4 * k4 + 1], o[4 * k4 + 2], o[4 * k4 + 3]};
                        *reinterpret_cast<f4u*>(plane + f0 + 4 * k4) = w;
                    }
                } else {
                    for (uint32_t k = 0; k < cnt; ++k) plane[f0 + k] = o[k];
                }
            }
            wave_sync();      // before the next pass / ear refills the stage
        }
    }
}

// ---------------------------------------------------------------------------------------------
// prepass kernel.  The tables are arrays of 32/64-byte structs; a lane reading its own struct with 16-byte loads makes
// every load instruction touch 64 separate lines.  The wave instead moves its 64 consecutive structs as one contiguous
// block (16 B per lane, lane-consecutive) and transposes through LDS (row stride W+1 words: conflict-free).
// ---------------------------------------------------------------------------------------------
template <class T> __device__ __forceinline__ void wave_aos_load(T& out, const T* __restrict__ arr, uint32_t first, uint32_t n_valid, int lane, uint32_t* lds) {
    constexpr int W = sizeof(T) / 4, Q = W / 4;       // words and 16-byte quads per element
    const uint4* src = reinterpret_cast<const uint4*>(arr + first);
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int v = lane + 64 * k;                    // quad index inside the wave's block of 64 elements
        const int el = v / Q, part = v % Q;
        if ((uint32_t)el < n_valid) {
            const uint4 q = src[v];
            uint32_t* dst = lds + el * (W + 1) + part * 4;
            dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
        }
    }
    wave_sync();
    uint32_t o[W];
#pragma unroll
    for (int w = 0; w < W; ++w) o[w] = lds[lane * (W + 1) + w];
    __builtin_memcpy(&out, o, sizeof(T));
    wave_sync();
}
// Two tables at once: both blocks' global loads are in flight together (one after the other, the second waited behind the
// first's LDS round trip: a memory latency of the walk, which at small set sizes is nothing but latencies).
template <class A, class B> __device__ __forceinline__ void wave_aos_load2(A& out_a, const A* __restrict__ arr_a, B& out_b, const B* __restrict__ arr_b,
                                                                           uint32_t first, uint32_t n_valid, int lane, uint32_t* lds) {
    constexpr int WA = sizeof(A) / 4, QA = WA / 4, WB = sizeof(B) / 4, QB = WB / 4;
    const uint4* src_a = reinterpret_cast<const uint4*>(arr_a + first);
    const uint4* src_b = reinterpret_cast<const uint4*>(arr_b + first);
    uint4 qa[QA], qb[QB];
#pragma unroll
    for (int k = 0; k < QA; ++k) { const int v = lane + 64 * k; qa[k] = (uint32_t)(v / QA) < n_valid ? src_a[v] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
    for (int k = 0; k < QB; ++k) { const int v = lane + 64 * k; qb[k] = (uint32_t)(v / QB) < n_valid ? src_b[v] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
    for (int k = 0; k < QA; ++k) {
        const int v = lane + 64 * k;
        uint32_t* dst = lds + (v / QA) * (WA + 1) + (v % QA) * 4;
        dst[0] = qa[k].x; dst[1] = qa[k].y; dst[2] = qa[k].z; dst[3] = qa[k].w;
    }
    wave_sync();
    uint32_t oa[WA];
#pragma unroll
    for (int w = 0; w < WA; ++w) oa[w] = lds[lane * (WA + 1) + w];
    __builtin_memcpy(&out_a, oa, sizeof(A));
    wave_sync();
#pragma unroll
    for (int k = 0; k < QB; ++k) {
        const int v = lane + 64 * k;
        uint32_t* dst = lds + (v / QB) * (WB + 1) + (v % QB) * 4;
        dst[0] = qb[k].x; dst[1] = qb[k].y; dst[2] = qb[k].z; dst[3] = qb[k].w;
    }
    wave_sync();
    uint32_t ob[WB];
#pragma unroll
    for (int w = 0; w < WB; ++w) ob[w] = lds[lane * (WB + 1) + w];
    __builtin_memcpy(&out_b, ob, sizeof(B));
    wave_sync();
}

template <class T> __device__ __forceinline__ void wave_aos_store(const T& in, T* __restrict__ arr, uint32_t first, uint32_t n_valid, int lane, uint32_t* lds) {
    constexpr int W = sizeof(T) / 4, Q = W / 4;
    uint32_t o[W];
    __builtin_memcpy(o, &in, sizeof(T));
#pragma unroll
    for (int w = 0; w < W; ++w) lds[lane * (W + 1) + w] = o[w];
    wave_sync();
    uint4* dstg = reinterpret_cast<uint4*>(arr + first);
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int v = lane + 64 * k;
        const int el = v / Q, part = v % Q;
        if ((uint32_t)el < n_valid) {
            const uint32_t* src = lds + el * (W + 1) + part * 4;
            dstg[v] = make_uint4(src[0], src[1], src[2], src[3]);
        }
    }
    wave_sync();
}
struct alignas(16) EarPair { EarParams e[2]; };

// HBM -> LDS window descriptor: samples [ws, ws + 4*nvec) of a clip through a buffer descriptor clipped to
// [window, clip end): lanes outside it get zeros with no memory traffic (frames.rs:105-123).
// Returns {descriptor word 0, word 1, byte count, byte offset of the window start relative to the descriptor
// base (<= 0)}.
__device__ __forceinline__ int4 window_desc(const float* clip, int clip_len4, int ws, int nvec) {
    const int ws_pos = ws > 0 ? ws : 0;             // first in-clip sample of the window
    const int neg4 = (ws < 0 ? ws : 0) * 4;         // byte offset of the window start relative to it (<= 0)
    long long rec = (long long)(clip_len4 - ws_pos) * 4;          // bytes to the (padded) clip end
    const long long wend = (long long)neg4 + (long long)nvec * 16;   // bytes to the window end
    if (rec > wend) rec = wend;
    if (rec < 0) rec = 0;
    const uint64_t base = (uint64_t)(clip + ws_pos);
    return make_int4((int)(base & 0xffffffffu), (int)((base >> 32) & 0xffffu), (int)rec, neg4);   // stride 0
}

// frames.rs:176-201's per-chunk set-up for one (source, 512-frame tile): what spatial_mix's phase A used to redo per
// tile.  A pure function of the walk's per-ear results (EarParams), the source and the tile index.
//   per (ear, chunk) stream: t_c = the inner clock at the chunk's start (frames.rs:198 applied per earlier chunk),
//   s0 = t_c * rate (f64, :177), base = s0 as isize (:179), frac0 = (s0 - base) as f32 (:181/:189), ds (:178);
//   the window = every sample index the tile's four streams can touch.  Its upper end only has to be an upper bound:
//   the cursor after 255 sequentially rounded `offset += ds` steps is below frac0 + 255 * ds by at most
//   255 half-ulps of its own magnitude (1.6e-5 relative), so the f32 closed form plus a 1e-4 relative margin covers it.
// (device_types.h: a Downmix source over a clip that carries its mono sum is a plain clip source to the fused kernels)
__device__ __forceinline__ SrcStatic downmix_as_mono(const SceneParams& P, const SrcStatic& s) {
    SrcStatic r = s;
    if (s.kind == KIND_DOWNMIX && P.fused && __float_as_uint(s.freq_or_value) == 1u) {
        r.kind = KIND_FRAMES;
        r.clip = s.clip + downmix_presum_offset(s.clip_len);
    }
    return r;
}
__device__ __forceinline__ TileRec make_tile_rec(const SceneParams& P, const SrcStatic& s_in, const EarParams& e0, const EarParams& e1, uint32_t tile) {
    const SrcStatic s = downmix_as_mono(P, s_in);
    TileRec r = {};                                  // info == 0: PATH_SKIP
    if (e0.flags & EAR_SKIP) return r;
    if (s.kind == KIND_SINE) {
        // sine.rs:34-40 inside the spatial chunk loop.  The record carries what a lane needs -- frequency, FixedGain factor, per
        // ear {dt, g0, dg}, per chunk the phase the chunk starts from (the `fmodf` chain of sine.rs:39 over the chunks before) --
        // so that spatial_mix renders the source inline, accumulators in registers, unless the argument of `sin` can leave the
        // range its argument reduction is exact for (then: the out-of-line path with the library's sinf).
        r.info = PATH_SINE;
        r.desc[0] = __float_as_uint(s.freq_or_value); r.desc[1] = __float_as_uint(s.fixed_gain);
        bool small = true;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const EarParams& ep = e ? e1 : e0;
            r.ear[e].ds = ep.dt; r.ear[e].g0 = ep.g0; r.ear[e].dg = ep.dg; r.ear[e].wrel = 0u;
            float ph = ep.phase_ear;
            for (uint32_t cc = 0; cc < tile * TILE_CHUNKS; ++cc) ph = fmodf(ph + (ep.dt * 256.0f) * s.freq_or_value, ODDIO_TAU);
#pragma unroll
            for (int c = 0; c < TILE_CHUNKS; ++c) {
                r.frac0[e][c] = ph;
                ph = fmodf(ph + (ep.dt * 256.0f) * s.freq_or_value, ODDIO_TAU);
            }
            if (!(fabsf((ep.dt * 255.0f) * s.freq_or_value) < 12000.0f)) small = false;     // (also false for NaN)
        }
        if (small && !s.fx) r.info = PATH_SINE_INLINE;      // (a soft-clipped Sine: out of line, apply_fx)
        return r;
    }
    if (s.kind == KIND_CONSTANT) { r.info = PATH_CONST; return r; }
    if (s.kind == KIND_CYCLE) { r.info = PATH_ROW; return r; }
    if (s.kind != KIND_FRAMES && s.kind != KIND_DOWNMIX) { r.info = PATH_GENERIC; return r; }
    // a per-source soft clip: Reinhard is rendered inline by the staged loops of the kernels that compile it in (P.dmx: the
    // Downmix-capable instantiations; info bits 28-30 = SrcStatic::fx), Tanh only by the fused (FAST-mode) ones, else by the exact per-lane path
    if (s.fx && (!P.dmx || ((s.fx & FX_TANH) && !P.fused) || (s.fx & FX_CHAIN))) { r.info = PATH_GENERIC; return r; }   // (FX_CHAIN: always the exact per-lane path)
    // Downmix<FramesSignal<[f32;2]>> (downmix.rs:24-29 over frames.rs:176-201): the same cursor, the window holds interleaved
    // stereo frames -- `mul` floats per frame; always variant 2 of spatial_mix (sub-windows when the window is larger than the stage)
    const bool stereo = s.kind == KIND_DOWNMIX;
    if (stereo && !P.dmx) { r.info = PATH_GENERIC; return r; }   // (the scene was launched without its Downmix-capable kernels)
    const int mul = stereo ? 2 : 1;
    int lo = 0x7fffffff, hi = (int)0x80000000, generic = 0, fl = 0;
    int wbase[2][2];
    const double rate = (double)s.clip_rate;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const EarParams& ep = e ? e1 : e0;
        const float ds = ep.dt * (float)s.clip_rate;                                      // frames.rs:178
        const float dev = fabsf(ds - 1.0f);
        const bool fast = dev <= FLT_EPSILON;                                            // :180
        if (fast) fl |= e ? SFLAG_FAST_R : SFLAG_FAST_L;
        if (dev < PAD_EPS) fl |= SFLAG_PAD;
        // the staged path needs a forward-running, sane cursor; everything else is exact but slow
        if (!(ds > 0.0f) || !(ds < 4096.0f)) generic = 1;
        double t_c = ep.t_ear;
        for (uint32_t cc = 0; cc < tile * TILE_CHUNKS; ++cc) t_c = t_c + (double)ep.dt * 256.0;   // frames.rs:198 per chunk
#pragma unroll
        for (int c = 0; c < TILE_CHUNKS; ++c) {
            const uint32_t c_abs = tile * TILE_CHUNKS + (uint32_t)c;
            const int rem = (int)P.n_frames - (int)(c_abs * 256u);
            const int len = rem < 0 ? 0 : (rem > 256 ? 256 : rem);
            const double s0 = t_c * rate;                                                 // :177
            const long long base = f64_as_isize(s0);                                      // :179
            const float frac0 = (float)(s0 - (double)base);                               // :181 / :189
            if (!(fabs(s0) < 1.0e9)) generic = 1;
            if (frac0 < 0.0f) fl |= SFLAG_NEG;
            wbase[e][c] = (int)base;
            r.frac0[e][c] = frac0;
            if (len > 0 && !generic) {
                int i0, i1;
                if (fast) { i0 = (int)base; i1 = (int)base + 255; }
                else {
                    const float xb = frac0 + 255.0f * ds;
                    const float xu = xb + fabsf(xb) * 1.0e-4f + 1.0e-2f;                  // >= the exactly rounded running sum (f32 is enough for a bound)
                    if (!(xu < 8.0e6f)) generic = 1;
                    i0 = (int)base + (int)frac0;
                    i1 = (int)base + (int)xu;
                }
                lo = min(lo, min(i0, i1));
                hi = max(hi, max(i0, i1));
            }
            t_c = t_c + (double)ep.dt * 256.0;
        }
        r.ear[e].ds = ds; r.ear[e].g0 = ep.g0; r.ear[e].dg = ep.dg;
    }
    const int ws = lo & ~3;
    const int count = (hi + 2 - ws) * mul;           // floats
    if (stereo) fl &= ~SFLAG_PAD;                    // (the padded layout is the mono kernels')
    {   // the padded layout (one extra slot per 16 samples, written a whole 16-byte vector at a time) must fit the
        // window buffer too.  Near-unit windows are ~550 samples, but a listener rotation inside the callback can pull
        // one ear's ratio to 1 while the other's window grows to the full 608: re-laid, that window would run 12 floats
        // into the next buffer.  Such a source keeps the plain layout (bank conflicts only), or, if it needs the
        // constant-fract branch that only the padded variant implements, the exact per-lane path.
        const int vec_samples = ((count + 3) >> 2) << 2;
        if ((fl & SFLAG_PAD) && vec_samples + (vec_samples >> 4) + 1 > WIN_CAP) {
            if (fl & (SFLAG_FAST_L | SFLAG_FAST_R)) generic = 1;
            else fl &= ~SFLAG_PAD;
        }
    }
    int path, npass = 1;
    if (generic) path = PATH_GENERIC;
    else if (lo > hi) path = PATH_SKIP;              // no frames in this tile
    else if (count <= WIN_CAP) path = PATH_LDS;
    else {
        // larger than the stage: sub-windows, if both ears step forward slowly enough for a lane's run to fit their overlap
        // (the constant-fract branch exists for the padded single window only)
        npass = (count - WIN_CAP + MULTI_STRIDE - 1) / MULTI_STRIDE + 1;
        const float ds_max = stereo ? MULTI_DS_MAX_STEREO : MULTI_DS_MAX;
        const bool ok = r.ear[0].ds <= ds_max && r.ear[1].ds <= ds_max && (stereo || !(fl & (SFLAG_FAST_L | SFLAG_FAST_R))) && npass <= MULTI_PASS_MAX;
        path = ok ? PATH_LDS : PATH_GENERIC;
        fl &= ~SFLAG_PAD;
    }
    if (path != PATH_LDS) { r.info = (uint32_t)path; return r; }
    if (s.fixed_gain != 1.0f || s.fx) fl |= SFLAG_FG;
    const int nvec_all = (count + 3) >> 2;
    const int nvec = npass > 1 ? WIN_CAP / 4 : nvec_all;      // per DMA: a whole sub-window (the descriptor clips the last one)
    const int4 d = window_desc(s.clip, (int)((s.clip_len * (uint32_t)mul + 3u) & ~3u), ws * mul, nvec_all);
    r.desc[0] = (uint32_t)d.x; r.desc[1] = (uint32_t)d.y; r.desc[2] = (uint32_t)d.z;
    // a window that starts before the clip: the descriptor base is the clip start and the first -ws/4 vectors are out
    // of range (zeros); one that lies entirely before it has a zero-byte descriptor, any offset reads zeros
    const int negvec = (d.z > 0) ? ((-d.w) >> 4) : 0;
    if (negvec > 255) { r.info = (uint32_t)PATH_GENERIC; return r; }   // (a large window that starts far before its clip)
    (void)ODDIO_BOUNDS_CHECK(P.bounds_err, nvec >= 1 && nvec * 4 <= WIN_CAP && negvec >= 0 && negvec <= 255 && d.z >= 0 && d.z <= nvec_all * 16 &&
                             wbase[0][0] - ws >= 0 && (wbase[1][1] - ws) * mul <= 65535, BOUNDS_RECORD, nvec, tile);
    r.info = (uint32_t)path | ((uint32_t)fl << 3) | ((uint32_t)nvec << 8) | ((uint32_t)negvec << 16) | ((uint32_t)(npass > 1 ? npass : 0) << 24) |
             ((uint32_t)(stereo ? 1 : 0) << 27) | ((uint32_t)s.fx << 28);
#pragma unroll
    for (int e = 0; e < 2; ++e) {      // a chunk's base in floats from the window start
        const int w0 = min(max((wbase[e][0] - ws) * mul, 0), 65535), w1 = min(max((wbase[e][1] - ws) * mul, 0), 65535);
        r.ear[e].wrel = (uint32_t)w0 | ((uint32_t)w1 << 16);
    }
    return r;
}

// The record of spatial_mix_pair (pair_kernels.h): make_tile_rec's set-up for the four chunks of a callback of up to 1024
// frames at once, with ONE window -- every sample index the eight streams (2 ears x 4 chunks) can touch.  Sources the pair
// kernel has no staged variant for (windows larger than its stage, stereo clips, absurd cursors) take its exact per-lane path.
constexpr int PAIR_WIN_CAP = 1184;               // samples staged per source (ds <= ~1.11 over 1024 frames)
constexpr int PAIR_CHUNKS = 4;
__device__ __forceinline__ PairRec make_pair_rec(const SceneParams& P, const SrcStatic& s_in, const EarParams& e0, const EarParams& e1) {
    const SrcStatic s = downmix_as_mono(P, s_in);
    PairRec r = {};                                  // info == 0: PATH_SKIP
    if (e0.flags & EAR_SKIP) return r;
    if (s.kind == KIND_SINE) {                       // (see make_tile_rec: the record carries what the inline Sine needs)
        r.info = PATH_SINE;
        r.desc[0] = __float_as_uint(s.freq_or_value); r.desc[1] = __float_as_uint(s.fixed_gain);
        bool small = true;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const EarParams& ep = e ? e1 : e0;
            r.ear[e].ds = ep.dt; r.ear[e].g0 = ep.g0; r.ear[e].dg = ep.dg;
            float ph = ep.phase_ear;
#pragma unroll
            for (int c = 0; c < PAIR_CHUNKS; ++c) {
                r.ear[e].frac0[c] = ph;
                ph = fmodf(ph + (ep.dt * 256.0f) * s.freq_or_value, ODDIO_TAU);      // sine.rs:39 per chunk
            }
            if (!(fabsf((ep.dt * 255.0f) * s.freq_or_value) < 12000.0f)) small = false;
        }
        if (small && !s.fx) r.info = PATH_SINE_INLINE;
        return r;
    }
    if (s.kind == KIND_CONSTANT) { r.info = PATH_CONST; return r; }
    if (s.kind == KIND_CYCLE) { r.info = PATH_ROW; return r; }
    if (s.kind != KIND_FRAMES) { r.info = PATH_GENERIC; return r; }     // (Downmix: the exact per-lane path)
    if (((s.fx & FX_TANH) && !P.fused) || (s.fx & FX_CHAIN)) { r.info = PATH_GENERIC; return r; }   // (the soft clips are rendered inline: info bits 28-30 = SrcStatic::fx; Tanh by the fused kernel only)
    int lo = 0x7fffffff, hi = (int)0x80000000, generic = 0, fl = 0;
    int wbase[2][PAIR_CHUNKS];
    const double rate = (double)s.clip_rate;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const EarParams& ep = e ? e1 : e0;
        const float ds = ep.dt * (float)s.clip_rate;                                      // frames.rs:178
        const float dev = fabsf(ds - 1.0f);
        const bool fast = dev <= FLT_EPSILON;                                            // :180
        if (fast) fl |= e ? SFLAG_FAST_R : SFLAG_FAST_L;
        if (dev < PAD_EPS) fl |= SFLAG_PAD;
        if (!(ds > 0.0f) || !(ds < 4096.0f)) generic = 1;
        double t_c = ep.t_ear;
#pragma unroll
        for (int c = 0; c < PAIR_CHUNKS; ++c) {
            const int rem = (int)P.n_frames - 256 * c;
            const int len = rem < 0 ? 0 : (rem > 256 ? 256 : rem);
            const double s0 = t_c * rate;                                                 // :177
            const long long base = f64_as_isize(s0);                                      // :179
            const float frac0 = (float)(s0 - (double)base);                               // :181 / :189
            if (!(fabs(s0) < 1.0e9)) generic = 1;
            if (frac0 < 0.0f) fl |= SFLAG_NEG;
            wbase[e][c] = (int)base;
            r.ear[e].frac0[c] = frac0;
            if (len > 0 && !generic) {
                int i0, i1;
                if (fast) { i0 = (int)base; i1 = (int)base + 255; }
                else {
                    const float xb = frac0 + 255.0f * ds;
                    const float xu = xb + fabsf(xb) * 1.0e-4f + 1.0e-2f;                  // >= the exactly rounded running sum
                    if (!(xu < 8.0e6f)) generic = 1;
                    i0 = (int)base + (int)frac0;
                    i1 = (int)base + (int)xu;
                }
                lo = min(lo, min(i0, i1));
                hi = max(hi, max(i0, i1));
            }
            t_c = t_c + (double)ep.dt * 256.0;                                            // frames.rs:198 per chunk
        }
        r.ear[e].ds = ds; r.ear[e].g0 = ep.g0; r.ear[e].dg = ep.dg;
    }
    const int ws = lo & ~3;
    const int count = hi + 2 - ws;                   // floats
    {   // the padded layout must fit the window buffer too (see make_tile_rec)
        const int vec_samples = ((count + 3) >> 2) << 2;
        if ((fl & SFLAG_PAD) && vec_samples + (vec_samples >> 4) + 1 > PAIR_WIN_CAP) {
            if (fl & (SFLAG_FAST_L | SFLAG_FAST_R)) generic = 1;
            else fl &= ~SFLAG_PAD;
        }
    }
    int path;
    if (generic) path = PATH_GENERIC;
    else if (lo > hi) path = PATH_SKIP;
    else if (count <= PAIR_WIN_CAP) path = PATH_LDS;
    else path = PATH_GENERIC;                        // larger than the stage (resample ratios above ~1.11)
    if (path != PATH_LDS) { r.info = (uint32_t)path; return r; }
    if (s.fixed_gain != 1.0f || s.fx) fl |= SFLAG_FG;
    const int nvec = (count + 3) >> 2;
    const int4 d = window_desc(s.clip, (int)((s.clip_len + 3u) & ~3u), ws, nvec);
    r.desc[0] = (uint32_t)d.x; r.desc[1] = (uint32_t)d.y; r.desc[2] = (uint32_t)d.z;
    const int negvec = (d.z > 0) ? ((-d.w) >> 4) : 0;
    if (negvec > 511) { r.info = (uint32_t)PATH_GENERIC; return r; }
    (void)ODDIO_BOUNDS_CHECK(P.bounds_err, nvec >= 1 && nvec * 4 <= PAIR_WIN_CAP && negvec >= 0 && d.z >= 0 && d.z <= nvec * 16 &&
                             wbase[0][0] - ws >= 0 && wbase[1][PAIR_CHUNKS - 1] - ws <= 65535, BOUNDS_RECORD, nvec, 0);
    r.info = (uint32_t)path | ((uint32_t)fl << 3) | ((uint32_t)nvec << 8) | ((uint32_t)negvec << 17) | ((uint32_t)s.fx << 28);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int c = 0; c < PAIR_CHUNKS; ++c) r.ear[e].wrel[c] = (uint16_t)min(max(wbase[e][c] - ws, 0), 65535);
    return r;
}

// `d_len` is the device-resident set length (set_kernels.h); `len_snap` receives the length this walk saw
// (the mix kernel of the callback reads it; the set is compacted at the end of the callback).
// `recs`: the tile records of the callback's first `n_rec_tiles` (<= REC_TILES) tiles, [tile][rec_stride].
// EarParams are written only where something reads them: the out-of-line paths of spatial_mix / cycle_sources and
// tile_records (callbacks longer than REC_TILES tiles).
__global__ __launch_bounds__(256) void spatial_prepass(SceneParams P, const SrcStatic* __restrict__ st,
                                                       SrcDyn* __restrict__ dyn, SrcPending* __restrict__ pend,
                                                       EarParams* __restrict__ ear, uint32_t* __restrict__ stopped_hdr,
                                                       uint32_t stopped_cap, int check_pending, const uint32_t* __restrict__ d_len,
                                                       uint32_t* __restrict__ len_snap, TileRec* __restrict__ recs, uint32_t rec_stride,
                                                       uint32_t n_rec_tiles, int ear_always, uint32_t* __restrict__ cycle_list,
                                                       uint32_t* __restrict__ cycle_rlist, uint32_t cycle_par) {
    __shared__ uint32_t stage[4][64 * (sizeof(PairRec) / 4 + 1)];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t* lds = stage[threadIdx.x >> 6];
    const uint32_t len = d_len[0];
    if (i == 0) { *len_snap = len; if (cycle_list) { cycle_list[cycle_par ^ 1u] = 0u; cycle_rlist[cycle_par ^ 1u] = 0u; } }   // (the lists of the callback after this one)
    const uint32_t first = i - (uint32_t)lane;
    if (first >= len) return;                                       // whole wave past the end
    const uint32_t n_valid = (len - first) < 64u ? (len - first) : 64u;
    SrcDyn d = {};
    SrcStatic s = {};
    wave_aos_load2(d, dyn, s, st, first, n_valid, lane, lds);
    EarPair ep = {};
    ep.e[0].flags = EAR_SKIP; ep.e[1].flags = EAR_SKIP;
    if (i < len) prepass_source(P, i, d, s, pend, ep.e[0], ep.e[1], stopped_hdr, stopped_cap, check_pending);
    if (cycle_list && i < len && s.kind == KIND_CYCLE && !(ep.e[0].flags & EAR_SKIP)) {   // the Cycle sources rendered this callback (cycle_scan / cycle_render)
        const uint32_t k = atomicAdd(&cycle_list[cycle_par], 1u);
        cycle_list[2u + k] = i;
    }
    bool needs_ear = false;
    if (P.pair) {   // spatial_mix_pair's record: one per source for the whole callback
        const PairRec r = make_pair_rec(P, s, ep.e[0], ep.e[1]);
        const uint32_t path = r.info & 7u;
        needs_ear = path != PATH_LDS && path != PATH_SKIP;
        wave_aos_store(r, reinterpret_cast<PairRec*>(recs), first, n_valid, lane, lds);
    } else
    for (uint32_t t = 0; t < n_rec_tiles; ++t) {
        const TileRec r = make_tile_rec(P, s, ep.e[0], ep.e[1], t);
        const uint32_t path = r.info & 7u;
        needs_ear = needs_ear || (path != PATH_LDS && path != PATH_SKIP);
        wave_aos_store(r, recs + (size_t)t * rec_stride, first, n_valid, lane, lds);
    }
    needs_ear = needs_ear || (i < len && s.kind == KIND_CYCLE);    // cycle_sources reads the flags of a stopped Cycle too
    if (ear_always || __any(needs_ear)) wave_aos_store(ep, reinterpret_cast<EarPair*>(ear), first, n_valid, lane, lds);
    wave_aos_store(d, dyn, first, n_valid, lane, lds);
}

// Callbacks longer than REC_TILES tiles run the mix in passes; this writes the records of the pass's tiles
// [tile0, tile0 + n) from the EarParams the walk left.
__global__ __launch_bounds__(256) void tile_records(SceneParams P, const SrcStatic* __restrict__ st, const EarParams* __restrict__ ear,
                                                    const uint32_t* __restrict__ len_snap, TileRec* __restrict__ recs, uint32_t rec_stride,
                                                    uint32_t tile0, uint32_t n) {
    __shared__ uint32_t stage[4][64 * 17];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t* lds = stage[threadIdx.x >> 6];
    const uint32_t len = *len_snap;
    const uint32_t first = i - (uint32_t)lane;
    if (first >= len) return;
    const uint32_t n_valid = (len - first) < 64u ? (len - first) : 64u;
    SrcStatic s = {};
    EarPair ep = {};
    wave_aos_load(s, st, first, n_valid, lane, lds);
    wave_aos_load(ep, reinterpret_cast<const EarPair*>(ear), first, n_valid, lane, lds);
    if (i >= len) { ep.e[0].flags = EAR_SKIP; ep.e[1].flags = EAR_SKIP; }
    for (uint32_t t = 0; t < n; ++t) {
        const TileRec r = make_tile_rec(P, s, ep.e[0], ep.e[1], tile0 + t);
        wave_aos_store(r, recs + (size_t)t * rec_stride, first, n_valid, lane, lds);
    }
}



typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// HBM -> LDS: the window of one (source, tile) into the window buffer at LDS byte address `lds_dst`, 16 B per
// lane per 1 KiB piece, through the bounds-checked descriptor the walk kernel made (window_desc): lanes outside
// it get zeros with no memory traffic (frames.rs:105-123).  The loads are issued from inline asm on purpose: hipcc
// would make every later ds_read wait for ALL outstanding LDS-DMA (it cannot tell the two window buffers apart),
// which would serialise the prefetch of the next source with the reads of the current one.  Completion is awaited
// with window_wait().  d0..d2, info: wave-uniform words of the source's TileRec; lane16 = 16 * lane.
constexpr int WIN_LAST_LANES = (WIN_BYTES - 2048) / 16;   // lanes of the third piece that stay inside the window buffer
__device__ __forceinline__ void window_dma(uint32_t lds_dst, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t info, int lane16, int extra = 0) {
    u32x4 rsrc;
    rsrc.x = d0; rsrc.y = d1; rsrc.z = d2;
    rsrc.w = 0x00020000u;
    const int nvec = (int)((info >> 8) & 255u);
    const int neg = -16 * (int)((info >> 16) & 255u);
#ifdef ODDIO_HIP_BOUNDS
    if (nvec * 16 > WIN_BYTES || (!(info >> 24) && (int)d2 > nvec * 16 + neg + 16)) asm volatile("s_trap 2");   // a window larger than its buffer would overwrite a neighbour's LDS
#endif
    const int voff = neg + lane16 + extra;   // negative offsets wrap to huge unsigned values: out of range -> 0  (extra: the sub-window's byte offset)
    uint32_t keep;
    // pieces 0 and 1 from every lane: lanes past the window write zeros inside the buffer (harmless, no traffic)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
                 "buffer_load_dwordx4 %1, %2, 0 offen" ODDIO_WIN_POLICY " lds\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:1024" ODDIO_WIN_POLICY " lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
    if (nvec > 128) {
        if (lane16 < 16 * WIN_LAST_LANES)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:2048" ODDIO_WIN_POLICY " lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
    }
    static_assert(WIN_PIECES == 3 && WIN_LAST_LANES > 0 && WIN_LAST_LANES <= 64, "three pieces cover a window buffer");
}
__device__ __forceinline__ void window_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Near-unit sources: plain -> padded layout in place (slot(s) = s + s/16; the pad slot repeats the
// following sample so that a pair (w, w+1) is always two adjacent dwords).  One wave: the DS
// operations execute in issue order, so every read below precedes every write.
__device__ __forceinline__ void window_repack_padded(unsigned char* win_bytes, int nvec, int lane, uint32_t* err) {
    // (the lane index is made opaque here: hipcc otherwise hoists this function's address arithmetic out of every
    // loop of the kernel and parks it in ~10 VGPRs that the common path then lacks)
    asm volatile("" : "+v"(lane));
    u32x4 v[WIN_PIECES];
#pragma unroll
    for (int k = 0; k < WIN_PIECES; ++k)
        v[k] = (lane + 64 * k < nvec) ? *reinterpret_cast<const u32x4*>(win_bytes + 16 * lane + 1024 * k) : u32x4{0u, 0u, 0u, 0u};
    wave_sync();
    unsigned int* win = reinterpret_cast<unsigned int*>(win_bytes);
#pragma unroll
    for (int k = 0; k < WIN_PIECES; ++k) {
        const int q = lane + 64 * k;
        if (q < nvec) {
            const int li = 4 * q;
            const int pos = li + (li >> 4);
            if (!ODDIO_BOUNDS_CHECK(err, pos + 3 < WIN_CAP, BOUNDS_REPACK, pos, nvec)) continue;
            win[pos + 0] = v[k].x; win[pos + 1] = v[k].y; win[pos + 2] = v[k].z; win[pos + 3] = v[k].w;
            if ((li & 15) == 0 && li > 0) win[pos - 1] = v[k].x;
        }
    }
    wave_sync();
}

// tanh for the FAST-mode kernels (contract: 1e-5 of the output's peak; the exact kernels keep the library's tanhf on the per-lane path):
// |x| < 0.3: the odd Taylor polynomial to x^9 (next term 5e-8 relative); otherwise 1 - 2 / (exp(2|x|) + 1) from v_exp_f32 and
// v_rcp_f32 (<= 4e-7 relative).  tanh.rs:22-29.
__device__ __forceinline__ float tanh_fast(float x) {
    const float a = fminf(fabsf(x), 10.0f);
    const float x2 = a * a;
    float p = 2.1869488536155203e-2f;                         // 62/2835
    p = __builtin_fmaf(p, x2, -5.3968253968253968e-2f);       // -17/315
    p = __builtin_fmaf(p, x2, 1.3333333333333333e-1f);        // 2/15
    p = __builtin_fmaf(p, x2, -3.3333333333333333e-1f);       // -1/3
    const float small = __builtin_fmaf(a * x2, p, a);
    const float e = __builtin_amdgcn_exp2f(a * 2.8853900817779268f);
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
    return __builtin_copysignf(a < 0.3f ? small : big, x);
}

// A per-source soft clip (apply_fx) in the staged-window loops: FX instantiations (the Downmix-capable spatial_mix kernels, spatial_mix_pair) render Reinhard
// sources inline -- `fxk` is wave-uniform -- (Tanh sources: tanh_fast in the fused kernels; the exact kernels send them down the per-lane path with the library's tanhf)
// FAST (the fused FAST-mode kernels, whose contract is the 1e-5 tolerance): the quotient from v_rcp_f32 and one Newton step -- within
// an ulp of the correctly rounded divide the exact kernels keep, a third of its instructions.
template <bool HAS_FG, bool FX, bool FAST = false>
__device__ __forceinline__ float gain_or_fx(float v, float fixed_gain, int fxk) {
    if (FX && fxk) {
        if (!(fxk & FX_CLIP_FIRST)) v = v * fixed_gain;
        if (FAST && (fxk & FX_TANH)) {                        // (only the fused kernels are handed Tanh sources: SceneParams::fused)
            v = tanh_fast(v);
            if (fxk & FX_CLIP_FIRST) v = v * fixed_gain;
            return v;
        }
        const float d = 1.0f + fabsf(v);
        if (FAST) {
            const float r = __builtin_amdgcn_rcpf(d);
            const float q = v * r;
            v = __builtin_fmaf(__builtin_fmaf(-d, q, v), r, q);
        } else {
            v = v / d;                                        // reinhard.rs:32
        }
        if (fxk & FX_CLIP_FIRST) v = v * fixed_gain;
        return v;
    }
    if (HAS_FG) v = v * fixed_gain;                           // gain.rs:32-37
    return v;
}

// One source, staged-window path.  acc[i] += lerp * gain for this lane's ear (spatial.rs:458-462).
// NONNEG: every cursor value of the source is >= 0, so fract(x) == x - trunc(x) (one v_fract_f32).
// PAD: padded window layout (see above); also serves frames.rs:180-187's constant-fract path.
// `x` is the lane's checkpoint (the cursor at its first frame), `wrel4` 4 x its chunk's base index
// relative to the window start.
// The accumulate is an in-place v_add_f32 (inline asm with a tied operand) in the FULL kernels: hipcc otherwise
// gives the sums of each inlined variant fresh registers and copies all 16 back at every join.
// FUSED (FAST mode when a callback is rendered by more than one wave, i.e. when the sum is a tree anyway): acc = fma(v, g, acc),
// one rounding where spatial.rs:460's `o += s * gain` has two -- one VALU op less per sample (0.233 -> 0.227 ms per launch
// at the headline size, same box) and closer to the exact sum; the lerp and the gain ramp of those kernels are fused the same
// way (mix_source_lds).  ORDERED mode, the contribution rows and single-wave
// callbacks keep the reference's two roundings: the product, then the sum.
template <bool FULL, bool FUSED>
__device__ __forceinline__ void acc_add(float& acc, float v, float g, bool on) {
    if (FUSED) {
        if (FULL) asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v), "v"(g));
        else if (on) acc = __builtin_fmaf(v, g, acc);
    } else {
        const float p = v * g;                                // spatial.rs:459-460
        if (FULL) asm("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(p));
        else if (on) acc = acc + p;
    }
}
// ST (variant 2 only): `stereo` (wave-uniform) marks a window of interleaved stereo frames -- Downmix<FramesSignal<[f32;2]>>,
// downmix.rs:24-29: each channel interpolated (frames.rs:180-196, incl. the constant-fract branch `fast`), then channels().sum().
// RAMP1 (spatial_mix_pair's fused instantiations): the gain of frame i is fma(i, dg, fma(fi[0], dg, g0)) -- `i` a literal of the
// unrolled loop -- instead of fma(fi[i], dg, g0): within an ulp of it, and the kernel keeps one `frame as f32` register, not 16.
template <bool FULL, bool HAS_FG, bool NONNEG, bool PAD, bool FUSED, bool WRAP = false, bool ST = false, int CAP = WIN_CAP, bool FX = false, bool RAMP1 = false>
__device__ __forceinline__ void mix_source_lds(const unsigned char* win_bytes, int wrel4, float x, int b, int fast, float frac0, float (&acc)[16],
                                               const float (&fi)[16], uint32_t frame0, uint32_t n_frames, float fixed_gain, float g0, float dg,
                                               float ds, int win_samples, uint32_t* err, int ring_len = 0, int stereo = 0, int fxk = 0) {
    if (!FULL && frame0 >= n_frames) return;   // this lane's 16 frames lie past the end of `out`
    const float* win = reinterpret_cast<const float*>(win_bytes);
    static_assert(!RAMP1 || FUSED, "the two-step gain ramp is a FAST-mode form");
    const float gbase = RAMP1 ? __builtin_fmaf(fi[0], dg, g0) : 0.0f;
#define ODDIO_GAIN_AT(I) (RAMP1 ? __builtin_fmaf((float)(I), dg, gbase) : (FUSED ? __builtin_fmaf(fi[I], dg, g0) : g0 + fi[I] * dg))
    if (WRAP) {
        // Ring::sample (ring.rs:59-78) for a lane whose cursor may pass the ring's end: `x >= len` rewrites the cursor to
        // (x % len) + fract before the step.  The staged window is linear across the ring's end (the ring carries a mirror
        // of its first samples behind its last one), so the LDS position of index x - len is that of x: only the cursor
        // value changes.  Not pipelined: a few percent of the sources of a callback take this variant.
        int wrel = wrel4 >> 2;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int tr = (int)x;
            const float fr = x - (float)tr;                                    // ring.rs:61 (from the cursor before the rewrite)
            if (tr >= ring_len) { tr -= ring_len; x = (float)tr + fr; wrel += ring_len; }   // :66-68 (len <= x < 2 len)
            const int w = wrel + tr;
            float a = 0.0f, bb = 0.0f;
            if (ODDIO_BOUNDS_CHECK(err, w >= 0 && w + 1 < win_samples, BOUNDS_WINDOW_INDEX, w, win_samples)) { a = win[w]; bb = win[w + 1]; }
            x = x + ds;                                                        // :77
            asm volatile("" : "+v"(dg));
            const float v = FUSED ? __builtin_fmaf(fr, bb - a, a) : a + fr * (bb - a);
            acc_add<FULL, FUSED>(acc[i], v, ODDIO_GAIN_AT(i), frame0 + (uint32_t)i < n_frames);
        }
        return;
    }
    if (PAD && fast) {
        // frames.rs:180-187 (|ds - 1| <= EPSILON): constant fract, consecutive pairs
        const int w0 = (wrel4 >> 2) + 16 * b;
        if (!ODDIO_BOUNDS_CHECK(err, w0 >= 0 && w0 + 16 < win_samples && (w0 + 16) + ((w0 + 16) >> 4) < CAP, BOUNDS_PAD_INDEX, w0, win_samples)) return;
        float a = win[w0 + (w0 >> 4)];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int w1 = w0 + i + 1;
            const float bb = win[w1 + (w1 >> 4)];
            float v = FUSED ? __builtin_fmaf(frac0, bb - a, a) : a + frac0 * (bb - a);
            v = gain_or_fx<HAS_FG, FX, FUSED>(v, fixed_gain, fxk);
            acc_add<FULL, FUSED>(acc[i], v, ODDIO_GAIN_AT(i), frame0 + (uint32_t)i < n_frames);
            a = bb;
        }
        return;
    }
    // frames.rs:189-196: x_{16b+i} = x_{16b} (+ ds) i times, exactly as the scan produced it.
    // Software pipelined by hand: the pair reads of samples i+1 .. i+MIX_DEPTH are issued before
    // sample i is consumed; sched_barrier keeps hipcc from sinking them back next to their use.
    const float* wbase = reinterpret_cast<const float*>(win_bytes + wrel4);
    const int wrel = wrel4 >> 2;
    if (ST && stereo) {
        // Interleaved stereo frames (not pipelined by hand: a handful of sources): frame index w -> floats 2 w .. 2 w + 3
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int tr = (int)x;
            float f = x - (float)tr;                                           // frames.rs:192
            if (fast) { tr = 16 * b + i; f = frac0; }                          // frames.rs:180-187
            x = x + ds;
            float l0 = 0.0f, r0 = 0.0f, l1 = 0.0f, r1 = 0.0f;
            if (ODDIO_BOUNDS_CHECK(err, wrel + 2 * tr >= 0 && wrel + 2 * tr + 3 < win_samples, BOUNDS_WINDOW_INDEX, wrel + 2 * tr, win_samples)) {
                const float* q_ = wbase + 2 * tr;
                l0 = q_[0]; r0 = q_[1]; l1 = q_[2]; r1 = q_[3];
            }
            asm volatile("" : "+v"(dg));
            const float vl = FUSED ? __builtin_fmaf(f, l1 - l0, l0) : l0 + f * (l1 - l0);   // frame.rs:39-41 per channel
            const float vr = FUSED ? __builtin_fmaf(f, r1 - r0, r0) : r0 + f * (r1 - r0);
            float v = (0.0f + vl) + vr;                                        // downmix.rs:27-29: channels().sum() from 0.0
            v = gain_or_fx<HAS_FG, FX, FUSED>(v, fixed_gain, fxk);
            acc_add<FULL, FUSED>(acc[i], v, ODDIO_GAIN_AT(i), frame0 + (uint32_t)i < n_frames);
            __builtin_amdgcn_sched_barrier(0);     // one frame at a time: the kernel has no registers to spare for reads hoisted across frames
        }
        return;
    }
    float a[16], bb[16], fr[16];
#define ODDIO_ISSUE(I)                                                                  \
    {                                                                                   \
        const int tr = (int)x;                                  /* v_cvt_i32_f32 (toward zero) */ \
        fr[I] = NONNEG ? __builtin_amdgcn_fractf(x) : x - (float)tr;   /* frames.rs:192 */          \
        int w = tr;                                                                     \
        const bool in_ = ODDIO_BOUNDS_CHECK(err, wrel + tr >= 0 && wrel + tr + 1 < win_samples, BOUNDS_WINDOW_INDEX, wrel + tr, win_samples); \
        if (!in_) { a[I] = 0.0f; bb[I] = 0.0f; }                                        \
        else if (PAD) { w = wrel + tr; w = w + (w >> 4); (void)ODDIO_BOUNDS_CHECK(err, w + 1 < CAP, BOUNDS_PAD_INDEX, w, win_samples); a[I] = win[w]; bb[I] = win[w + 1]; } \
        else if (ODDIO_DIAG & 1) { const float* p_ = win + ((tr & 0x800) + (int)(threadIdx.x & 31)); a[I] = p_[0]; bb[I] = p_[1]; } /* diagnostic build only: wrong samples, no bank conflicts */ \
        else { a[I] = wbase[w]; bb[I] = wbase[w + 1]; }         /* one ds_read2_b32 */   \
        x = x + ds;                                             /* frames.rs:194 */      \
    }
#pragma unroll
    for (int i = 0; i < MIX_DEPTH; ++i) ODDIO_ISSUE(i)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i + MIX_DEPTH < 16) ODDIO_ISSUE(i + MIX_DEPTH)
        __builtin_amdgcn_sched_barrier(0);
        // keeps hipcc from hoisting 16 gain values out of the loop (16 VGPRs -> scratch); placed before the lerp so that
        // the instruction after it does not read dg (hipcc pads an asm statement whose output is read next with s_nop)
        asm volatile("" : "+v"(dg));
        float v = FUSED ? __builtin_fmaf(fr[i], bb[i] - a[i], a[i]) : a[i] + fr[i] * (bb[i] - a[i]);   // frame.rs:39-41 lerp (unfused in the exact kernels)
        v = gain_or_fx<HAS_FG, FX, FUSED>(v, fixed_gain, fxk);                       // gain.rs:32-37
        acc_add<FULL, FUSED>(acc[i], v, ODDIO_GAIN_AT(i), frame0 + (uint32_t)i < n_frames);   // spatial.rs:459-460
        __builtin_amdgcn_sched_barrier(0);
    }
#undef ODDIO_ISSUE
#undef ODDIO_GAIN_AT
}

// sin(x) for |x| < ~12 600: three-constant Cody-Waite reduction by 2 pi (exact products for |k| <= 2^11), a fold to
// [-pi/2, pi/2] and the odd Taylor polynomial to x^11 (5.7e-8 at the interval's ends).  ~1e-7 absolute, against the
// reference's libm sinf: inside the Sine tolerance of the tests (1e-5 of max|ref|), like the library's device sinf.
__device__ __forceinline__ float sin_small(float x) {
    const float k = __builtin_rintf(x * 0.15915494309189535f);
    float r = __builtin_fmaf(-k, 6.28125f, x);                               // 2 pi = 6.28125 + 1.9350051879882812e-3 + 3.0199159819e-7
    r = __builtin_fmaf(-k, 1.9350051879882812e-3f, r);
    r = __builtin_fmaf(-k, 3.0199159819e-7f, r);
    const float pi = 3.14159265358979323846f;
    const float f = __builtin_copysignf(pi, r) - r;                          // sin(r) == sin(pi - r) (r > 0), sin(-pi - r) (r < 0)
    r = fabsf(r) > 1.57079632679489661923f ? f : r;
    const float r2 = r * r;
    float p = -2.5052108385441720e-8f;
    p = __builtin_fmaf(p, r2, 2.7557319223985893e-6f);
    p = __builtin_fmaf(p, r2, -1.9841269841269841e-4f);
    p = __builtin_fmaf(p, r2, 8.3333333333333332e-3f);
    p = __builtin_fmaf(p, r2, -1.6666666666666666e-1f);
    return __builtin_fmaf(r * r2, p, r);
}

// One Sine source on accumulators parked in LDS (slot i of lane l at acc_lds[i * PARK_STRIDE + l], like mix_source_rare -- inline, with the
// accumulators in registers, the sine's temporaries push the hot kernel into scratch memory):
// acc[i] += (sin((dt * i) * freq + phase) * fixed_gain) * gain for this lane's 16 frames of its chunk (sine.rs:34-38,
// gain.rs:32-37, spatial.rs:459-460).  The argument is the reference's, operation for operation; every parameter comes from the
// source's tile record (no global loads here).
__device__ __noinline__ void mix_source_sine(float* acc_lds, int lane, uint32_t frame0, uint32_t n_frames, float phase, float dt, float freq,
                                            float fixed_gain, float g0, float dg) {
    const float ib = (float)(16 * (lane & 15));
    const float fbase = (float)frame0;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const float t = dt * (ib + (float)i);                                // interval * (i as f32): exact integer in f32
        float v = sin_small(t * freq + phase);
        v = v * fixed_gain;
        const float p = v * (g0 + (fbase + (float)i) * dg);
        if (frame0 + (uint32_t)i < n_frames) acc_lds[i * PARK_STRIDE + lane] = acc_lds[i * PARK_STRIDE + lane] + p;
    }
}

// frames.rs:105-123 straight from global memory
__device__ __forceinline__ float clip_at(const float* clip, uint32_t len, long long i) {
    return (i >= 0 && i < (long long)len) ? clip[i] : 0.0f;
}

constexpr int GEN_BATCH = 2;   // frames whose loads are in flight together in the exact per-lane path (its registers count against the kernels that call it)
// ---- rare paths -------------------------------------------------------------------------------
// Sources that do not take the staged-window path (windows larger than the LDS stage, backwards
// or absurd cursors, Sine / Constant sources).  Kept out of line and working on accumulators
// parked in LDS (slot i of lane l at acc_lds[i * PARK_STRIDE + l]) so that their register needs (sinf range
// reduction, 64-bit indices) do not inflate the hot kernel's allocation.  They fetch the source's
// parameters from global memory themselves.
__device__ __forceinline__ void mix_source_rare_body(float* acc_lds, int lane, uint32_t frame0, uint32_t n_frames, uint32_t c_abs, int path,
                                                     const SrcStatic* __restrict__ st, const EarParams* __restrict__ ear, uint32_t src,
                                                     const float* cycle_rows, uint32_t cycle_plane, const int eB) {
    const int b = lane & 15;
    const SrcStatic s = st[src];
    const FxRegs fxch = load_fx_chain(s, st);
    const EarParams ep = ear[2 * src + eB];
    const float fbase = (float)frame0;
    const float g0 = ep.g0, dg = ep.dg, dt = ep.dt, fixed_gain = s.fixed_gain;
    if (path == PATH_ROW) {
        // Seek-set Cycle: the contribution was rendered by cycle_sources; add it at this source's position
        const float* plane = cycle_rows + ((size_t)__float_as_uint(s.freq_or_value) * 2u + (uint32_t)eB) * cycle_plane;
        if (frame0 + 16u <= n_frames) {   // all 16 loads in flight together (one at a time, a row cost 16 memory latencies)
            f4u r[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) r[k4] = *reinterpret_cast<const f4u*>(plane + frame0 + 4 * k4);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_lds[i * PARK_STRIDE + lane] = acc_lds[i * PARK_STRIDE + lane] + r[i >> 2][i & 3];
            return;
        }
#pragma unroll 1
        for (int i = 0; i < 16; ++i)
            if (frame0 + (uint32_t)i < n_frames) acc_lds[i * PARK_STRIDE + lane] = acc_lds[i * PARK_STRIDE + lane] + plane[frame0 + (uint32_t)i];
        return;
    }
    if (path == PATH_SINE || path == PATH_CONST) {
        // sine.rs:34-40 inside the spatial chunk loop; Constant (constant.rs:16-18)
        const bool is_sine = path == PATH_SINE;
        float ph = ep.phase_ear;
        if (is_sine) for (uint32_t cc = 0; cc < c_abs; ++cc) ph = fmodf(ph + (dt * 256.0f) * s.freq_or_value, ODDIO_TAU);
#pragma unroll 1
        for (int i = 0; i < 16; ++i) {
            float v;
            if (is_sine) {
                const float t = dt * (float)(16 * b + i);
                v = sinf(t * s.freq_or_value + ph);
            } else {
                v = s.freq_or_value;
            }
            v = apply_fx(v, fixed_gain, s.fx, fxch);
            const float p = v * (g0 + (fbase + (float)i) * dg);
            if (frame0 + (uint32_t)i < n_frames) acc_lds[i * PARK_STRIDE + lane] = acc_lds[i * PARK_STRIDE + lane] + p;
        }
        return;
    }
    // PATH_GENERIC: frames.rs:176-201 per lane from global memory (also Downmix, downmix.rs:27-29)
    const bool downmix = s.kind == KIND_DOWNMIX;
    const float* clip = s.clip;
    const uint32_t clip_len = s.clip_len;
    double t_c = ep.t_ear;
    for (uint32_t cc = 0; cc < c_abs; ++cc) t_c = t_c + (double)dt * 256.0;
    const double s0 = t_c * (double)s.clip_rate;
    const float ds = dt * (float)s.clip_rate;
    const long long base = f64_as_isize(s0);
    const float frac0 = (float)(s0 - (double)base);
    const bool fast = fabsf(ds - 1.0f) <= FLT_EPSILON;
    // Lane-major over the chunk: in step i the lane renders frame 16 * i + b of its chunk -- the 16 lanes of a chunk read 16
    // adjacent frames, a few cache lines per load instruction (block-major, each lane on its own 16 frames 16 * ds samples from
    // its neighbour's, every load instruction touched 64 lines: Downmix sources cost 6x a staged FramesSignal) -- and adds it to
    // slot b of the lane that owns the frame, (lane & ~15) | i.  The cursor is the same serial chain: b adds to the lane's
    // first frame, 16 more per step.
    const uint32_t chunk0 = frame0 - 16u * (uint32_t)b;          // first frame of the lane's chunk (callback-relative)
    float x = frac0;
    if (!fast) for (int k = 0; k < b; ++k) x = x + ds;
    // GEN_BATCH frames at a time: the indices first (registers only), then all the loads together, then the arithmetic
#pragma unroll 1
    for (int i0 = 0; i0 < 16; i0 += GEN_BATCH) {
        long long idx[GEN_BATCH]; float fr[GEN_BATCH];
#pragma unroll
        for (int k = 0; k < GEN_BATCH; ++k) {
            if (fast) { idx[k] = base + (long long)(16 * (i0 + k) + b); fr[k] = frac0; }
            else {
                const long long tr = (long long)x; idx[k] = base + tr; fr[k] = x - (float)tr;
#pragma unroll
                for (int q = 0; q < 16; ++q) x = x + ds;
            }
        }
        float v[GEN_BATCH];
        if (downmix) {   // per-channel lerp of the stereo frame, then channels().sum()
            float2 fa[GEN_BATCH], fb[GEN_BATCH];
#pragma unroll
            for (int k = 0; k < GEN_BATCH; ++k) {
                const bool in_a = idx[k] >= 0 && idx[k] < (long long)clip_len, in_b = idx[k] + 1 >= 0 && idx[k] + 1 < (long long)clip_len;
                fa[k] = in_a ? reinterpret_cast<const float2*>(clip)[idx[k]] : make_float2(0.0f, 0.0f);
                fb[k] = in_b ? reinterpret_cast<const float2*>(clip)[idx[k] + 1] : make_float2(0.0f, 0.0f);
            }
#pragma unroll
            for (int k = 0; k < GEN_BATCH; ++k) {
                const float l = fa[k].x + fr[k] * (fb[k].x - fa[k].x), r = fa[k].y + fr[k] * (fb[k].y - fa[k].y);
                float sum = 0.0f;
                sum = sum + l;
                sum = sum + r;
                v[k] = sum;
            }
        } else {
            float a[GEN_BATCH], bb[GEN_BATCH];
#pragma unroll
            for (int k = 0; k < GEN_BATCH; ++k) { a[k] = clip_at(clip, clip_len, idx[k]); bb[k] = clip_at(clip, clip_len, idx[k] + 1); }
#pragma unroll
            for (int k = 0; k < GEN_BATCH; ++k) v[k] = a[k] + fr[k] * (bb[k] - a[k]);
        }
#pragma unroll
        for (int k = 0; k < GEN_BATCH; ++k) {
            const int i = i0 + k;
            const uint32_t f = chunk0 + 16u * (uint32_t)i + (uint32_t)b;
            const float vg = apply_fx(v[k], fixed_gain, s.fx, fxch);
            const float p = vg * (g0 + (float)f * dg);
            float* slot = acc_lds + b * PARK_STRIDE + ((lane & ~15) | i);
            if (f < n_frames) *slot = *slot + p;
        }
    }
}

// spatial_mix: lanes 0-31 are the left ear, 32-63 the right ear
__device__ __noinline__ void mix_source_rare(float* acc_lds, int lane, uint32_t frame0, uint32_t n_frames, uint32_t c_abs, int path,
                                             const SrcStatic* __restrict__ st, const EarParams* __restrict__ ear, uint32_t src,
                                             const float* cycle_rows, uint32_t cycle_plane) {
    mix_source_rare_body(acc_lds, lane, frame0, n_frames, c_abs, path, st, ear, src, cycle_rows, cycle_plane, lane >> 5);
}
// spatial_mix_pair: the whole wavefront renders ear `eB`
__device__ __noinline__ void mix_source_rare_ear(float* acc_lds, int lane, uint32_t frame0, uint32_t n_frames, uint32_t c_abs, int path,
                                                 const SrcStatic* __restrict__ st, const EarParams* __restrict__ ear, uint32_t src,
                                                 const float* cycle_rows, uint32_t cycle_plane, int eB) {
    mix_source_rare_body(acc_lds, lane, frame0, n_frames, c_abs, path, st, ear, src, cycle_rows, cycle_plane, eB);
}

// RING: a buffered source that the general kernel rendered (buffered_sources_wave: every shape the fast path does not
// take) left `s * gain` for both ears in its slab row [frame][ear]; it is added at the source's place in the walk.
__device__ __noinline__ void mix_source_slab(float* acc_lds, int lane, uint32_t frame0, uint32_t n_frames, const float* __restrict__ row) {
    const int eB = lane >> 5;
#pragma unroll 1
    for (int i = 0; i < 16; ++i)
        if (frame0 + (uint32_t)i < n_frames) acc_lds[i * PARK_STRIDE + lane] = acc_lds[i * PARK_STRIDE + lane] + row[2 * (frame0 + (uint32_t)i) + eB];
}

// v_mov_b32_dpp quad_perm:[K,K,K,K]: every lane reads lane K of its group of four
template <int K> __device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), K * 0x55, 0xf, 0xf, true));
}

// STORE (ORDERED mode at scale): one source's 16 accumulators of every lane -> its rows in `contrib`.
// A lane's 16 values are one 64-byte row; written as they are, every store instruction would touch 64 separate
// lines with 16 bytes each.  The four lanes of a quad (four adjacent column blocks) instead exchange 16-byte pieces
// (DPP, registers only), so that store k writes the whole row of the quad's lane k: 64 contiguous bytes per quad, and
// the quad's four rows of one source sit next to each other (256 bytes = two full lines from four consecutive stores).
__device__ __forceinline__ void rows_transpose(const float (&acc)[16], float4 (&o)[4], int lane) {
    const int i = lane & 3;
#define ODDIO_PICK(K, M) \
    { const float c0 = quad_bcast<K>(acc[M]), c1 = quad_bcast<K>(acc[4 + M]), c2 = quad_bcast<K>(acc[8 + M]), c3 = quad_bcast<K>(acc[12 + M]); \
      const float v = i == 0 ? c0 : (i == 1 ? c1 : (i == 2 ? c2 : c3)); \
      if (M == 0) o[K].x = v; else if (M == 1) o[K].y = v; else if (M == 2) o[K].z = v; else o[K].w = v; }
#define ODDIO_PICK4(K) ODDIO_PICK(K, 0) ODDIO_PICK(K, 1) ODDIO_PICK(K, 2) ODDIO_PICK(K, 3)
    ODDIO_PICK4(0) ODDIO_PICK4(1) ODDIO_PICK4(2) ODDIO_PICK4(3)
#undef ODDIO_PICK4
#undef ODDIO_PICK
}
// rows of sources J (o) and J + 1 (o1) for the four column blocks of the quad: column block k is the k-th 1-KiB block
__device__ __forceinline__ void rows_store(const float4 (&o)[4], const float4 (&o1)[4], unsigned char* p) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#if ODDIO_ROWS_NT
        typedef float v4f_ __attribute__((ext_vector_type(4)));
        v4f_ a_ = {o[k].x, o[k].y, o[k].z, o[k].w}, b_ = {o1[k].x, o1[k].y, o1[k].z, o1[k].w};
        __builtin_nontemporal_store(a_, reinterpret_cast<v4f_*>(p + k * (MIX_GROUP * 64)));
        __builtin_nontemporal_store(b_, reinterpret_cast<v4f_*>(p + k * (MIX_GROUP * 64) + 64));
#else
        *reinterpret_cast<float4*>(p + k * (MIX_GROUP * 64)) = o[k];
        *reinterpret_cast<float4*>(p + k * (MIX_GROUP * 64) + 64) = o1[k];
#endif
    }
}

// grid = (n_workgroups, tiles of this pass); block = 64 * MIX_WG_WAVES.  Wave w walks groups [g_lo, g_hi) of
// 16 slots in DESCENDING order (the reference's reverse set walk, spatial.rs:204).  A workgroup
// leaves ONE partial tile: partials[(tile * n_wgs + wg) * PART_STRIDE + e * 512 + f] (planar L | R).
// `recs`: the tile records of this pass, [blockIdx.y][rec_stride], made by the walk kernel (make_tile_rec);
// `tile0`: the callback tile that blockIdx.y == 0 renders.
// STORE (ORDERED mode at scale): instead of accumulating, every source's contribution `s * gain` (spatial.rs:459-460)
// is written out -- layout contrib[group of 16 sources][ear][quad of column blocks][source in group][column block of
// the quad][16 frames], i.e. a lane's 16 accumulators are one 64-byte row, the four rows a quad of lanes holds for one
// source are 256 contiguous bytes (two whole 128-byte lines per source, written back to back) and everything a wave
// writes while it walks a group lies within 2 * ncb KiB -- and ordered_sum then adds the rows in the reference's order.  `partials` / `init` are unused there.  contrib_ncb = column blocks per ear.
// RING: the records describe Ring::sample reads of the buffered set (buffered_fast.h): frac0 is the cursor's absolute
// position in the ring, desc[2] the ring length (the window's byte count is 16 * nvec: a ring carries a mirror of its
// first samples behind its end, so a window never needs clipping), SFLAG_WRAP marks sources whose cursors may pass the
// ring's end, and an out-of-line source adds the slab row the general kernel rendered (P.cycle_rows = the slabs,
// [slot][frame][ear]).  The non-RING instantiations compile to what they were before the flag existed.
// DMX: the instantiations a scene with Downmix sources runs (variant 2 can render interleaved stereo windows); kept apart because
// the plain kernels have no register to spare (127 of 128: with the stereo loop compiled in they spill 48-80 bytes per lane).
// TRACK (ODDIO_HIP_MODE_TRACKED for callbacks of up to 512 frames; pair_kernels.h has the mode's description and the kernel for longer
// ones): every WAVE is a block of the tracked sum -- its sources are a contiguous stretch of the reference's walk -- and leaves its own
// partial tile (index: the wave, gridDim.x * MIX_WG_WAVES of them per tile; no cross-wave sum).  1: first pass, sums from zero.
// 2: second pass, `init` holds every wave's start value in the partial tiles' layout (track_prefix); the wave leaves end - start
// (the wave the reference's walk starts with keeps its start: the buffered set's sum).
template <bool FULL, bool STORE = false, bool FUSED = false, bool RING = false, bool DMX = false, int TRACK = 0>
__global__ __launch_bounds__(64 * MIX_WG_WAVES, MIX_WAVES_PER_SIMD) void spatial_mix(SceneParams P, const SrcStatic* __restrict__ st,
                                                                                     const EarParams* __restrict__ ear,
                                                                                     const TileRec* __restrict__ recs, uint32_t rec_stride, uint32_t tile0,
                                                                                     float* __restrict__ partials, const float* __restrict__ init,
                                                                                     uint32_t groups_per_wave, uint32_t n_groups,
                                                                                     const uint32_t* __restrict__ n_sources_ptr,
                                                                                     float* __restrict__ contrib, uint32_t contrib_ncb) {
    __shared__ __attribute__((aligned(16))) unsigned char smem_all[LDS_TOTAL * MIX_WG_WAVES];
    const uint32_t n_sources = *n_sources_ptr;   // the set length this callback's walk saw (n_groups is the host's upper bound)
    const int wv = threadIdx.x >> 6;
    unsigned char* smem = smem_all + LDS_TOTAL * wv;
    // LDS byte address of this wave's slice (what the DMA's M0 wants): low half of the flat address
    const uint32_t lds_slice = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)smem);
    const int lane = threadIdx.x & 63;
    const int lane16 = 16 * lane;
    const uint32_t wave = blockIdx.x * MIX_WG_WAVES + wv, tile = tile0 + blockIdx.y;
    const TileRec* __restrict__ trecs = recs + (size_t)blockIdx.y * rec_stride;
    const uint32_t n_frames = P.n_frames;
    float acc[16], fi[16];
    float4 held_[4] = {};        // STORE: the rows of the odd source of a pair (rows_store)
    // phase-B role: ear e, chunk c (of the tile), block b -> 16 consecutive frames
    const int eB = lane >> 5, cB = (lane >> 4) & 1, bB = lane & 15;
    const uint32_t frame0 = tile * TILE_FRAMES + 16u * (uint32_t)(lane & 31);   // this lane's first output frame
    const float fbase = (float)frame0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { acc[k] = 0.0f; fi[k] = fbase + (float)k; }   // `i as f32` (spatial.rs:459)
    static_assert(TRACK == 0 || !STORE, "TRACK accumulates"); static_assert(TRACK != 2 || !FUSED, "the tracked sums carry the reference's roundings (the first pass only places their start values: fused or not)");
    if (TRACK == 2) {
        if (frame0 < n_frames) {
            const uint32_t n_part = gridDim.x * MIX_WG_WAVES;
            const float* src = init + (((size_t)tile * (TILE_FRAMES / PART_FRAMES) + 2u * (uint32_t)(lane & 31)) * n_part + wave) * PART_BLOCK + (size_t)eB * PART_FRAMES;
            const size_t step = (size_t)n_part * PART_BLOCK;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 v = reinterpret_cast<const float4*>(src + (q4 >> 1) * step)[q4 & 1];
                acc[4 * q4] = v.x; acc[4 * q4 + 1] = v.y; acc[4 * q4 + 2] = v.z; acc[4 * q4 + 3] = v.w;
            }
        }
    } else if (!STORE && TRACK == 0 && init != nullptr && wave == 0) {
        // the buffered set is walked before the seekable one (spatial.rs:395-438): its sum is the
        // value the first source of this walk is added to
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (frame0 + (uint32_t)k < n_frames) acc[k] = init[2 * (frame0 + (uint32_t)k) + eB];
    }
    const uint32_t cB_abs = tile * TILE_CHUNKS + (uint32_t)cB;

    // Small scenes (fewer groups than resident waves): 2^split waves share a group, each rendering 16 >> split of its
    // sources -- the callback's latency is a wave's walk, and a quarter of a group is a quarter of the time.  `groups_per_wave`
    // carries split in its top byte (then one group per 2^split waves).
    const uint32_t split_log2 = groups_per_wave >> 24;
    uint32_t g_lo = wave * (groups_per_wave & 0xffffffu);
    uint32_t g_hi = g_lo + (groups_per_wave & 0xffffffu);
    unsigned part_mask = 0xffffu;
    if (split_log2) {
        g_lo = wave >> split_log2;
        g_hi = g_lo + 1u;
        const uint32_t per = (uint32_t)MIX_GROUP >> split_log2;
        part_mask = ((1u << per) - 1u) << ((wave & ((1u << split_log2) - 1u)) * per);
    }
    part_mask = (unsigned)__builtin_amdgcn_readfirstlane((int)part_mask);   // (wave-uniform; the compiler cannot see that: `wave` comes from threadIdx)
    if (g_hi > n_groups) g_hi = n_groups;

    // this lane's stream block in phase B belongs to stream 4j + 2e + c: byte offset of source 0's, then 4 blocks per source
    unsigned char* const blkB0 = smem + LDS_STREAM + (eB * 2 + cB) * (STREAM_WORDS * 4);
    constexpr int BLK_SRC = 4 * STREAM_WORDS * 4;            // bytes of stream blocks per source
    // STORE: this lane's byte offset inside a group's rows for source 0 (store_rows)
    const uint32_t row_off0 = ((uint32_t)eB * contrib_ncb + ((frame0 >> 4) & ~3u)) * (MIX_GROUP * 64u) + 16u * (uint32_t)(lane & 3);   // 1-KiB block of (ear, first column block of the quad) + piece

    // The records of a group -- lanes 0-15: {descriptor words, info} of source `lane`; lane (j, e, c): the stream's
    // {ds, g0, dg, wrel} and frac0 -- are fetched one group ahead: the loads are issued during the
    // first source of the group before and have landed long before the group boundary, and that group's last source
    // starts the first window of this one, so a boundary costs the cursor scan and nothing else.
#define ODDIO_LOAD_GROUP(GG, V, Q, F)                                                                                     \
    {                                                                                                                     \
        /* phase-A role: source j of the group, ear e, chunk c (from an opaque copy of the lane index, so that this */   \
        /* once-per-group address arithmetic is redone here instead of living in VGPRs through phase B) */               \
        int la_ = lane;                                                                                                   \
        asm volatile("" : "+v"(la_));                                                                                     \
        const TileRec* __restrict__ grec_ = trecs + (size_t)(GG) * MIX_GROUP;                                             \
        V = make_uint4(0u, 0u, 0u, 0u); Q = make_float4(0.0f, 0.0f, 0.0f, 0.0f); F = 0.0f;                                \
        if (la_ < MIX_GROUP && (GG) * MIX_GROUP + (uint32_t)la_ < n_sources) V = *reinterpret_cast<const uint4*>(grec_ + la_); \
        if ((GG) * MIX_GROUP + (uint32_t)(la_ >> 2) < n_sources) {                                                        \
            const TileRec* r_ = grec_ + (la_ >> 2);                                                                       \
            Q = *reinterpret_cast<const float4*>(&r_->ear[(la_ >> 1) & 1]);     /* {ds, g0, dg, wrel} */                  \
            F = r_->frac0[(la_ >> 1) & 1][la_ & 1];                                                                       \
        }                                                                                                                 \
    }
    uint4 pv = make_uint4(0u, 0u, 0u, 0u);
    float4 pq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float pf = 0.0f;
    if (g_hi > g_lo) ODDIO_LOAD_GROUP(g_hi - 1u, pv, pq, pf)
    int buf = 0;
    bool pre_issued = false;     // the last source of the previous group already started this group's first window
    for (uint32_t g = g_hi; g-- > g_lo;) {
        // ------------------------------ phase A ------------------------------
        // lanes 0-15 keep {descriptor words, info} of source `lane` for the whole group (read per source with
        // v_readlane: wave-uniform values without a trip through memory)
        uint4 vdesc;
        float4 q;
        float frac0;
        vdesc = pv; q = pq; frac0 = pf;
        bool need_prefetch = g > g_lo;
        int laneA = lane;
        asm volatile("" : "+v"(laneA));
        const int cA = laneA & 1;
        // bit j of lds_mask: source j of the group takes the staged-window path; rare_mask: an out-of-line path
        const int pj = (int)(vdesc.w & 7u);
        const unsigned lds_mask = (unsigned)__ballot(pj == PATH_LDS) & part_mask;
        const unsigned rare_mask = (unsigned)__ballot(pj != PATH_LDS && pj != PATH_SKIP) & part_mask;
        // `cur`: the next staged source of the walk; its window is in flight to / sits in WIN[buf]
        int cur = lds_mask ? 31 - __builtin_clz(lds_mask) : -1;
        uint32_t cur_info = 0;
#define ODDIO_ISSUE_WINDOW_OF(VD, JN, BUF)                                                                                \
    window_dma(lds_slice + (uint32_t)((BUF) ? LDS_WIN1 : LDS_WIN0), (uint32_t)__builtin_amdgcn_readlane((int)(VD).x, (JN)),  \
               (uint32_t)__builtin_amdgcn_readlane((int)(VD).y, (JN)),                                                      \
               RING ? 16u * (((uint32_t)__builtin_amdgcn_readlane((int)(VD).w, (JN)) >> 8) & 255u) : (uint32_t)__builtin_amdgcn_readlane((int)(VD).z, (JN)), \
               (uint32_t)__builtin_amdgcn_readlane((int)(VD).w, (JN)), lane16);
#define ODDIO_ISSUE_WINDOW(JN, BUF) ODDIO_ISSUE_WINDOW_OF(vdesc, JN, BUF)
        if (cur >= 0) {   // the first window is on its way while the cursors are scanned
            cur_info = (uint32_t)__builtin_amdgcn_readlane((int)vdesc.w, cur);
            if (!pre_issued) ODDIO_ISSUE_WINDOW(cur, buf)
        }
        pre_issued = false;
        {
            // exact f32 cursor scan (frames.rs:189-196) of stream (j, e, c): checkpoints every 16 frames
            float* blk = reinterpret_cast<float*>(smem + LDS_STREAM + laneA * (STREAM_WORDS * 4));
            const float ds = q.x;
            float x = frac0;
            if (RING && __ballot(laneA < MIX_GROUP && (((vdesc.w >> 3) & SFLAG_WRAP) != 0u))) {
                // a source of the group may pass its ring's end: Ring::sample's rewrite (ring.rs:66-68) is part of the
                // running sum.  Blocks of 16 steps that no stream can wrap in run the plain adds.
                const int rlen = __shfl((int)vdesc.z, laneA >> 2);
                const float lenf = (float)rlen;                     // exact: fast-path rings are shorter than 2^24 samples
#pragma unroll 1
                for (int b = 0; b < 15; ++b) {
                    blk[b] = x;
                    if (__any(x + 17.0f * ds >= lenf)) {
#pragma unroll 1
                        for (int i = 0; i < 16; ++i) {
                            const int tr = (int)x;
                            if (tr >= rlen) x = (float)(tr - rlen) + (x - (float)tr);
                            x = x + ds;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) x = x + ds;
                    }
                }
            } else {
#pragma unroll 1
                for (int b = 0; b < 15; ++b) {
                    blk[b] = x;
#pragma unroll
                    for (int i = 0; i < 16; ++i) x = x + ds;
                }
            }
            blk[15] = x;
            uint32_t wr = (__float_as_uint(q.w) >> (16 * cA)) & 0xffffu;
            if (RING) wr -= (uint32_t)(int)frac0;   // the stream's window position belongs to index trunc(frac0), not to index 0
            *reinterpret_cast<float4*>(blk + 16) = make_float4(__uint_as_float(4u * wr), q.y, q.z, q.x);
        }
        wave_sync();

        // ------------------------------ phase B ------------------------------
        // lane data of the staged source about to be mixed: its checkpoint and {4 * wrel, g0, dg, ds}
        float cx0 = 0.0f;
        float4 ct = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#define ODDIO_LANE_DATA(J, X0, T)                                                                                         \
    {                                                                                                                     \
        const unsigned char* blk_ = blkB0 + (J) * BLK_SRC;                                                                \
        X0 = reinterpret_cast<const float*>(blk_)[bB];                                                                    \
        T = *reinterpret_cast<const float4*>(blk_ + 64);                                                                  \
    }
        if (cur >= 0) ODDIO_LANE_DATA(cur, cx0, ct)
        // one staged source (== cur): wait for its window, start the next one's, mix
        // VAR: 0 the common source (no FixedGain, non-negative cursor), 1 padded layout (resample ratio within PAD_EPS
        // of 1), 2 FixedGain and/or a cursor that starts negative; -1: decided here (wave-uniform branches)
#define ODDIO_VARIANT(INFO) (RING ? ((((INFO) >> 3) & SFLAG_WRAP) ? 3 : ((((INFO) >> 3) & SFLAG_PAD) ? 1 : 0)) \
                                  : ((((INFO) >> 24) & 15u) ? 2 : ((((INFO) >> 3) & SFLAG_PAD) ? 1 : ((((INFO) >> 3) & (SFLAG_FG | SFLAG_NEG)) ? 2 : 0))))
        // A source whose window is larger than the stage (info bits 24-26 = its number of sub-windows) takes variant 2's loop
        // once per sub-window: `mpass` counts them; a pass starts the next sub-window instead of the next source's window, and
        // only the lanes whose runs lie in its sub-window render.
        int mpass = 0;
#define ODDIO_STAGED_SOURCE(VAR, PRE)                                                                                        \
    {                                                                                                                     \
        const int flags_j = (int)((cur_info >> 3) & 31u);                                                                 \
        const int var_j = (VAR) >= 0 ? (VAR) : ODDIO_VARIANT(cur_info);                                                   \
        unsigned char* win_bytes = smem + (buf ? LDS_WIN1 : LDS_WIN0);                                                    \
        window_wait();                                    /* this source's window has landed */                          \
        /* (so have the next group's records if an earlier source of this group fetched them: the empty asm makes */     \
        /* hipcc put its own wait for those loads here, where it is free, instead of in front of the next window) */     \
        asm volatile("" : "+v"(pv.x), "+v"(pv.y), "+v"(pv.z), "+v"(pv.w), "+v"(pq.x), "+v"(pq.y), "+v"(pq.z), "+v"(pq.w), "+v"(pf)); \
        const bool fetched_before = !need_prefetch;                                                                       \
        if (need_prefetch) { ODDIO_LOAD_GROUP(g - 1u, pv, pq, pf) need_prefetch = false; }                                \
        const unsigned below = lds_mask & ((1u << cur) - 1u);                                                             \
        const int nxt = below ? 31 - __builtin_clz(below) : -1;                                                           \
        uint32_t nxt_info = 0;                                                                                            \
        float nx0 = 0.0f;                                                                                                 \
        float4 nt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                                                                  \
        const int npass_ = (!RING && (var_j == 2)) ? (int)((cur_info >> 24) & 7u) : 0;                                    \
        const bool more_ = mpass + 1 < npass_;                                                                            \
        if (more_) {      /* the source's next sub-window */                                                              \
            window_dma(lds_slice + (uint32_t)((buf ^ 1) ? LDS_WIN1 : LDS_WIN0), (uint32_t)__builtin_amdgcn_readlane((int)vdesc.x, cur), \
                       (uint32_t)__builtin_amdgcn_readlane((int)vdesc.y, cur), (uint32_t)__builtin_amdgcn_readlane((int)vdesc.z, cur), cur_info, lane16, \
                       4 * MULTI_STRIDE * (mpass + 1));                                                                   \
        } else if (nxt >= 0) {   /* start the next staged source of this group; lands while we compute */                \
            nxt_info = (uint32_t)__builtin_amdgcn_readlane((int)vdesc.w, nxt);                                            \
            ODDIO_ISSUE_WINDOW(nxt, buf ^ 1)                                                                              \
            ODDIO_LANE_DATA(nxt, nx0, nt)                                                                                 \
        } else if ((PRE) && g > g_lo && fetched_before) {                                           \
            /* last staged source of the group: the other window buffer is free for the next group's first window */     \
            const unsigned nm_ = (unsigned)__ballot((int)(pv.w & 7u) == PATH_LDS);                                        \
            if (nm_) { ODDIO_ISSUE_WINDOW_OF(pv, 31 - __builtin_clz(nm_), buf ^ 1) pre_issued = true; }                   \
        }                                                                                                                 \
        const int wrel4 = __float_as_int(ct.x);                                                                           \
        if (RING && var_j == 3) {                                                                                  \
            /* a lane whose checkpoint lies behind its stream's start has already been rewritten: its window position is */ \
            /* one ring length further on */                                                                              \
            const int rlen_ = __builtin_amdgcn_readlane((int)vdesc.z, cur);                                               \
            const float start_ = reinterpret_cast<const float*>(blkB0 + cur * BLK_SRC)[0];   /* checkpoint 0 */           \
            const int wr4_ = wrel4 + (((int)cx0 < (int)start_) ? 4 * rlen_ : 0);                                          \
            mix_source_lds<FULL, false, true, false, FUSED, true>(win_bytes, wr4_, cx0, bB, 0, 0.0f, acc, fi, frame0, n_frames, 1.0f, ct.y, ct.z, ct.w, 4 * (int)((cur_info >> 8) & 255u), P.bounds_err, rlen_); \
        } else if (var_j == 1) {                                                                                          \
            const float fg = (!RING && (flags_j & SFLAG_FG)) ? st[g * MIX_GROUP + (uint32_t)cur].fixed_gain : 1.0f;   /* v * 1.0 == v */ \
            const float frac0_ = reinterpret_cast<const float*>(blkB0 + cur * BLK_SRC)[0];   /* checkpoint 0 */           \
            const int fast_e = eB ? (flags_j & SFLAG_FAST_R) : (flags_j & SFLAG_FAST_L);                                  \
            window_repack_padded(win_bytes, (int)((cur_info >> 8) & 255u), lane, P.bounds_err);                                         \
            mix_source_lds<FULL, true, false, true, FUSED, false, false, WIN_CAP, DMX>(win_bytes, wrel4, cx0, bB, fast_e, frac0_, acc, fi, frame0, n_frames, fg, ct.y, ct.z, ct.w, 4 * (int)((cur_info >> 8) & 255u), P.bounds_err, 0, 0, \
                                                                                         (DMX && !RING) ? (int)((cur_info >> 28) & 7u) : 0); \
        } else if (var_j == 0) {                                                                                          \
            mix_source_lds<FULL, false, true, false, FUSED>(win_bytes, wrel4, cx0, bB, 0, 0.0f, acc, fi, frame0, n_frames, 1.0f, ct.y, ct.z, ct.w, 4 * (int)((cur_info >> 8) & 255u), P.bounds_err); \
        } else {                                                                                                          \
            const float fg = (!RING && (flags_j & SFLAG_FG)) ? st[g * MIX_GROUP + (uint32_t)cur].fixed_gain : 1.0f;       \
            bool on_ = true;                                                                                              \
            int w4_ = wrel4;                                                                                              \
            const int stereo_ = DMX ? (int)((cur_info >> 27) & 1u) : 0;       /* interleaved stereo frames (Downmix) */      \
            const int fast_s = stereo_ ? (eB ? (flags_j & SFLAG_FAST_R) : (flags_j & SFLAG_FAST_L)) : 0;                  \
            const float frac0_s = fast_s ? reinterpret_cast<const float*>(blkB0 + cur * BLK_SRC)[0] : 0.0f;   /* checkpoint 0 */ \
            if (npass_) {   /* the lanes whose runs lie in this sub-window (the last one takes what lies beyond it too) */ \
                int lp_ = ((wrel4 >> 2) + ((int)cx0 << stereo_)) / MULTI_STRIDE;                                          \
                lp_ = lp_ < 0 ? 0 : (lp_ > npass_ - 1 ? npass_ - 1 : lp_);                                                \
                on_ = lp_ == mpass;                                                                                       \
                w4_ = wrel4 - 4 * MULTI_STRIDE * mpass;                                                                   \
            }                                                                                                             \
            if (on_) mix_source_lds<FULL, true, false, false, FUSED, false, DMX, WIN_CAP, DMX>(win_bytes, w4_, cx0, bB, fast_s, frac0_s, acc, fi, frame0, n_frames, fg, ct.y, ct.z, ct.w, \
                                                                                   4 * (int)((cur_info >> 8) & 255u), P.bounds_err, 0, stereo_, (DMX && !RING) ? (int)((cur_info >> 28) & 7u) : 0); \
        }                                                                                                                 \
        buf ^= 1;                                                                                                         \
        if (more_) ++mpass;                                                                                               \
        else { mpass = 0; cur = nxt; cur_info = nxt_info; cx0 = nx0; ct = nt; }                                           \
    }
        // STORE: the accumulators hold exactly one source's contribution (0 + p); write the rows, start the next from zero
        // (a skipped source -- stopped, or no frames in this tile -- leaves a row of zeros: x + 0.0 == x for every x the
        // running sum can hold, which is never -0.0).  (Issuing the stores one source late, after the next window wait,
        // so that nothing waits for them, changes nothing: the kernel moves 3.3 GB in 0.65 ms, it is HBM bound.)
#define ODDIO_EMIT(J)                                                                                                     \
    if (STORE) {                                                                                                          \
        /* rows leave in pairs: source J + 1 (odd) waits in registers for source J, and the two 64-byte rows of each */  \
        /* column block are written back to back -- whole 128-byte lines */                                              \
        if ((J) & 1) rows_transpose(acc, held_, lane);                                                                    \
        else {                                                                                                            \
            float4 o_[4];                                                                                                 \
            rows_transpose(acc, o_, lane);                                                                                \
            rows_store(o_, held_, reinterpret_cast<unsigned char*>(contrib) + (size_t)g * 2u * contrib_ncb * (MIX_GROUP * 64u) + (row_off0 + 64u * (uint32_t)(J))); \
        }                                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < 16; ++k) acc[k] = 0.0f;                                                     \
    }
        // rare path: park the accumulators in LDS (over the window buffers: a window in flight is
        // awaited first and fetched again afterwards), run out of line, fetch them back
#define ODDIO_RARE_SOURCE(J)                                                                                              \
    {                                                                                                                     \
        const int path_j = __builtin_amdgcn_readlane((int)vdesc.w, (J)) & 7;                                              \
        float* park = reinterpret_cast<float*>(smem);                                                                     \
        /* (a Sine's parameters: its stream block holds the chunk's phase and {., g0, dg, dt}; read before the park area is written) */ \
        float ph_ = 0.0f;                                                                                                 \
        float4 t_ = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                                                                  \
        if (!RING && path_j == PATH_SINE_INLINE) { ODDIO_LANE_DATA((J), ph_, t_) ph_ = reinterpret_cast<const float*>(blkB0 + (J) * BLK_SRC)[0]; } \
        window_wait();                                                                                                    \
        wave_sync();                                                                                                      \
        _Pragma("unroll") for (int k = 0; k < 16; ++k) park[k * PARK_STRIDE + lane] = acc[k];                                      \
        wave_sync();                                                                                                      \
        if (!RING && path_j == PATH_SINE_INLINE)                                                                          \
            mix_source_sine(park, lane, frame0, n_frames, ph_, t_.w, __int_as_float(__builtin_amdgcn_readlane((int)vdesc.x, (J))), \
                            __int_as_float(__builtin_amdgcn_readlane((int)vdesc.y, (J))), t_.y, t_.z);                    \
        else if (RING) mix_source_slab(park, lane, frame0, n_frames, P.cycle_rows + (size_t)(g * MIX_GROUP + (uint32_t)(J)) * P.cycle_plane);    \
        else mix_source_rare(park, lane, frame0, n_frames, cB_abs, path_j, st, ear, g * MIX_GROUP + (uint32_t)(J), P.cycle_rows, P.cycle_plane); \
        wave_sync();                                                                                                      \
        _Pragma("unroll") for (int k = 0; k < 16; ++k) acc[k] = park[k * PARK_STRIDE + lane];                                      \
        wave_sync();                                                                                                      \
        if (cur >= 0) ODDIO_ISSUE_WINDOW(cur, buf)                                                                        \
    }
        if (!STORE) {
            // The walk of a group, in descending slot order: runs of staged sources -- one loop per variant, each with
            // nothing but its own body (with the three bodies as arms of one loop hipcc keeps two copies of the 16
            // accumulators and moves them around every source) -- interrupted by the rare out-of-line sources.
            unsigned rm = rare_mask;
            while (cur >= 0 || rm) {
                const int rj = rm ? 31 - __builtin_clz(rm) : -1;
                if (rj > cur) {
                    rm &= ~(1u << rj);
                    ODDIO_RARE_SOURCE(rj)
                    continue;
                }
#pragma unroll 1
                while (cur > rj && ODDIO_VARIANT(cur_info) == 0) ODDIO_STAGED_SOURCE(0, rm == 0)
#pragma unroll 1
                while (cur > rj && ODDIO_VARIANT(cur_info) == 1) ODDIO_STAGED_SOURCE(1, rm == 0)
#pragma unroll 1
                while (cur > rj && ODDIO_VARIANT(cur_info) == 2) ODDIO_STAGED_SOURCE(2, rm == 0)
                if (RING) {
#pragma unroll 1
                    while (cur > rj && ODDIO_VARIANT(cur_info) == 3) ODDIO_STAGED_SOURCE(3, rm == 0)
                }
            }
        } else {
#pragma unroll 1
            for (int j = MIX_GROUP - 1; j >= 0; --j) {
                if (j == cur) {
                    do ODDIO_STAGED_SOURCE(-1, false) while (!RING && mpass != 0);   // (every sub-window of a large window)
                } else if ((rare_mask >> j) & 1u) {
                    ODDIO_RARE_SOURCE(j)
                }
                ODDIO_EMIT(j)
            }
        }
#undef ODDIO_RARE_SOURCE
#undef ODDIO_EMIT
#undef ODDIO_STAGED_SOURCE
#undef ODDIO_VARIANT
#undef ODDIO_LANE_DATA
#undef ODDIO_ISSUE_WINDOW
#undef ODDIO_ISSUE_WINDOW_OF
        if (need_prefetch) ODDIO_LOAD_GROUP(g - 1u, pv, pq, pf)   // a group without a staged source
        wave_sync();   // before the next group's phase A overwrites the stream blocks
    }
#undef ODDIO_LOAD_GROUP

    if (STORE) return;
    // ---- cross-wave reduction through LDS, fixed order (wave 0 + wave 1 + ...), then one store ----
    // partial tile is planar: [ear][512 frames]; this lane owns frames 16*(lane&31).. of ear lane>>5
    // (addresses derived from an opaque copy of the lane index: computed here, not carried through the walk in VGPRs)
    int le = threadIdx.x & 63;
    asm volatile("" : "+v"(le));
    // the lane's 16 frames are two blocks of PART_FRAMES frames: partials[(block * n_wgs + wg) * PART_BLOCK + ear * PART_FRAMES + k]
    float* dst = partials + (((size_t)tile * (TILE_FRAMES / PART_FRAMES) + 2u * (uint32_t)(le & 31)) * gridDim.x + blockIdx.x) * PART_BLOCK + (size_t)(le >> 5) * PART_FRAMES;
    const size_t dst_step = (size_t)gridDim.x * PART_BLOCK;      // to the lane's second block
    if (TRACK) {   // one partial tile per wave
        const uint32_t n_part = gridDim.x * MIX_WG_WAVES;
        const size_t off = (((size_t)tile * (TILE_FRAMES / PART_FRAMES) + 2u * (uint32_t)(le & 31)) * n_part + wave) * PART_BLOCK + (size_t)(le >> 5) * PART_FRAMES;
        const size_t step = (size_t)n_part * PART_BLOCK;
        if (16u * (uint32_t)(le & 31) + tile * TILE_FRAMES >= n_frames && !FULL) return;
        if (TRACK == 2 && (P.track_all || wave + 1u != n_part)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = reinterpret_cast<const float4*>(init + off + (q >> 1) * step)[q & 1];
                acc[4 * q] = acc[4 * q] - v.x; acc[4 * q + 1] = acc[4 * q + 1] - v.y; acc[4 * q + 2] = acc[4 * q + 2] - v.z; acc[4 * q + 3] = acc[4 * q + 3] - v.w;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(partials + off + (q >> 1) * step)[q & 1] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        return;
    }
    if (MIX_WG_WAVES == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(dst + (q >> 1) * dst_step)[q & 1] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        return;
    }
    __syncthreads();   // every wave is done with its slice
    if (wv != 0) {
        float* mine = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int k = 0; k < 16; ++k) mine[k * 64 + le] = acc[k];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll 1
        for (int w = 1; w < MIX_WG_WAVES; ++w) {
            const float* other = reinterpret_cast<const float*>(smem_all + LDS_TOTAL * w);
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = acc[k] + other[k * 64 + le];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(dst + (q >> 1) * dst_step)[q & 1] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
}

// ---------------------------------------------------------------------------------------------
// reduce: out[f][e] = sum over workgroup partials (fixed order) ; then Reinhard / Tanh
// partials are planar tiles [tile][wg][e][512]; out is interleaved stereo.
// grid = (ceil(n_frames / 32)), block = 1024: thread = (segment s of 16, ear e, frame-in-32)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float postfx_apply(float x, int postfx) {
    if (postfx == 1) return x / (1.0f + fabsf(x));   // reinhard.rs:32
    if (postfx == 2) return tanhf(x);                // tanh.rs:26
    return x;
}

constexpr int RED_FRAMES = 8;    // output frames per reduce block: 256 threads = (8 frames x 2 ears) x 16 segments (measured: 4 -> 10.9 us, 8 -> 6.8, 16 -> 7.7, 32 -> 13.6)
constexpr int RED_SEGS = 16;     // strided segments of the workgroup-partial list
constexpr int RED_BATCH = 32;    // loads in flight per thread

// One block sums ALL workgroup partials of its 16 outputs in a fixed order that depends only on n_wgs
// (segment s adds workgroups s, s+16, ... in ascending order; the 16 segment sums are then added in
// ascending order), applies Reinhard / Tanh and writes the interleaved stereo frames.  With n_wgs == 1
// (ORDERED mode) the output is that workgroup's value unchanged.
// SEGS: 16 (256 threads), or 32 (512 threads) for the 2048 partial tiles per tile that spatial_mix_pair leaves -- twice the
// data with the same ~64 loads per thread.
template <int SEGS = RED_SEGS>
__device__ __forceinline__ void reduce_partials_body(const float* __restrict__ partials, float* __restrict__ out, uint32_t n_wgs,
                                                     uint32_t n_frames, int postfx, uint32_t block) {
    constexpr int RED_SEGS = SEGS;     // (shadows the default for the code below)
    __shared__ float red[RED_SEGS][2 * RED_FRAMES];
    const uint32_t ox = threadIdx.x & (2 * RED_FRAMES - 1), seg = threadIdx.x / (2 * RED_FRAMES);
    const uint32_t e = ox / RED_FRAMES;
    const uint32_t f = block * RED_FRAMES + (ox % RED_FRAMES);      // output frame
    static_assert(RED_FRAMES == PART_FRAMES, "a reduce block sums one block of the partial sums");
    float s = 0.0f;
    if (f < n_frames) {
        const float* p = partials + (size_t)block * n_wgs * PART_BLOCK + ox;      // (ox == e * PART_FRAMES + frame in block)
        // the first addend is taken as is (0.0f + x would turn a -0.0 into +0.0); the rest in batches whose loads
        // are all in flight together (the partials were just written by other XCDs: every load is a ~1 us miss)
        bool first = true;
        for (uint32_t w = seg; w < n_wgs; w += RED_BATCH * RED_SEGS) {
            float v[RED_BATCH];
#pragma unroll
            for (int k = 0; k < RED_BATCH; ++k) {
                const uint32_t wk = w + (uint32_t)k * RED_SEGS;
                v[k] = wk < n_wgs ? p[(size_t)wk * PART_BLOCK] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < RED_BATCH; ++k)
                if (w + (uint32_t)k * RED_SEGS < n_wgs) { s = first ? v[k] : s + v[k]; first = false; }
        }
    }
    red[seg][ox] = s;
    __syncthreads();
    if (seg == 0 && f < n_frames) {
        float t = red[0][ox];
        const uint32_t nseg = n_wgs < (uint32_t)RED_SEGS ? n_wgs : (uint32_t)RED_SEGS;
        for (uint32_t k = 1; k < nseg; ++k) t = t + red[k][ox];
        out[2 * f + e] = postfx_apply(n_wgs ? t : 0.0f, postfx);
    }
}

// ORDERED mode at scale, second half: out[f][e] = ((init + row[len-1]) + row[len-2]) + ... + row[0] -- the
// reference's sequential f32 sum in its reverse set walk (spatial.rs:204,460).  The chain of `len` dependent adds per
// output IS the cost (one wave issues a dependent VALU op every ~4.6 cycles: ~0.5 ms at 262 144 sources), so the
// wave that walks it issues as little else as possible: one wavefront per (ear, column block of 16 frames); lane
// 4k + m holds frame k; a tile is 8 source groups = 8 chunks of 1 KiB (16 rows of 16 frames), streamed HBM -> LDS
// through a ring of ORD_RING tiles (buffer_load ... lds, several tiles ahead of the adds, no registers) by a second
// wave of the workgroup.  One
// ds_read2st64_b32 fetches rows 4i + m and 4i + 4 + m for the quad's lane m, and the adds take their operand from
// the quad's lanes in turn (v_add_f32_dpp quad_perm:[m,m,m,m]): 8 adds per LDS instruction, every lane of a quad
// carrying the same running sum.  Rows of sources >= len (stale chunks of an earlier, longer set) are never added.
// grid = (column blocks, 2 ears), block = 128.
constexpr int ORD_GROUPS = 8;                 // source groups per tile
constexpr int ORD_ROWS = ORD_GROUPS * MIX_GROUP;   // 128 sources: 8 KiB = 8 DMA instructions
#ifndef ODDIO_ORD_LOADERS
#define ODDIO_ORD_LOADERS 1
#endif
constexpr int ORD_LOADERS = ODDIO_ORD_LOADERS;     // loader waves per workgroup (tile kk is fetched by loader kk % ORD_LOADERS)
constexpr int ORD_RING = 8 * ORD_LOADERS;     // tiles in the LDS ring (64 KiB per loader); each loader keeps up to 7 in flight (56 of its 63 VMEM slots)
constexpr int ORD_Q = 8;                      // quad steps (4 rows each) per register batch: 4 batches per tile (chain_add8)

// s = (((s + v[lane 3 of the quad]) + v[lane 2]) + v[lane 1]) + v[lane 0] for eight registers in a row: the DPP operand
// is src0, the running sum stays in place.  One asm statement for the whole batch: hipcc pads every inline-asm
// statement with an s_nop (it cannot see that these adds need none: a DPP source written by an LDS read, the
// VALU-written sum read as a plain operand), which would double the instructions on the chain.
#define ODDIO_DPP4(V) \
    "v_add_f32_dpp %0, " V ", %0 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %0, " V ", %0 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %0, " V ", %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %0, " V ", %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void chain_add8(float& s, const float (&v)[8]) {
    asm(ODDIO_DPP4("%1") ODDIO_DPP4("%2") ODDIO_DPP4("%3") ODDIO_DPP4("%4") ODDIO_DPP4("%5") ODDIO_DPP4("%6") ODDIO_DPP4("%7") ODDIO_DPP4("%8")
        : "+v"(s) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
}
#undef ODDIO_DPP4

__global__ __launch_bounds__(64 * (1 + ORD_LOADERS)) void ordered_sum(const float* __restrict__ contrib, uint32_t contrib_ncb,
                                                   const uint32_t* __restrict__ n_sources_ptr, uint32_t n_frames,
                                                   const float* __restrict__ init, float* __restrict__ out, int postfx) {
    // Two waves: wave 1 streams the rows HBM -> LDS ring (every LDS-DMA instruction costs its issuer ~60 cycles, which
    // would otherwise sit on the chain), wave 0 does nothing but the adds.  They meet in two LDS counters:
    // `landed` = tiles (counted from the top) whose rows are in the ring, `consumed` = tiles whose rows are in the
    // adder's registers.
    __shared__ __attribute__((aligned(16))) float ring[ORD_RING][ORD_ROWS][16];
    __shared__ uint32_t landed[ORD_LOADERS], consumed;   // landed[w]: tiles of loader w (its kk / ORD_LOADERS) in the ring
    const int lane = threadIdx.x & 63;
    const bool loader = threadIdx.x >= 64;
    const uint32_t cb = blockIdx.x, e = blockIdx.y;
    const uint32_t n = *n_sources_ptr;
    const uint32_t n_tiles = (n + ORD_ROWS - 1) / ORD_ROWS;
    if (threadIdx.x < (unsigned)ORD_LOADERS) landed[threadIdx.x] = 0u;
    if (threadIdx.x == 0) consumed = 0u;
    __syncthreads();
    if (loader) {
        const uint32_t group_stride = 2u * contrib_ncb * 1024u;      // bytes from a group's chunk to the next group's
        // group 0: the 4-KiB chunk of this column block's quad, this column block's 64-byte rows 256 bytes apart
        const unsigned char* chunk0 = reinterpret_cast<const unsigned char*>(contrib) + ((size_t)e * contrib_ncb + cb) * 1024u;
        const uint32_t lds0 = (uint32_t)(uintptr_t)&ring[0][0][0];
        // the instruction offset (i * 1024) advances both the LDS and the memory address: the scalar offset adds the rest of i * group_stride
        const uint32_t so = group_stride - 1024u;
        const int voff = lane * 16;                                  // lane = (row of the group, 16-byte piece): a group's 16 rows are 1 KiB in a row
        constexpr int IN_FLIGHT = 6;                                 // this loader's tiles in flight after each wait (one more right after an issue)
        static_assert(8 * (IN_FLIGHT + 1) <= 63, "the tiles in flight fit the 6-bit VMEM counter");
        static_assert((IN_FLIGHT + 2) * ORD_LOADERS <= ORD_RING, "ring slots for everything in flight");
        const uint32_t me = (threadIdx.x >> 6) - 1u;                 // loader index
        uint32_t mine = 0;                                           // tiles this loader has issued
        // kk counts tiles from the top: tile index t = n_tiles - 1 - kk; all inside the allocation (sized in whole tiles)
        for (uint32_t kk = me; kk < n_tiles; kk += (uint32_t)ORD_LOADERS, ++mine) {
            const uint32_t t = n_tiles - 1u - kk;
            // the slot of tile kk was read by tile kk - ORD_RING
            while (kk >= (uint32_t)ORD_RING && __hip_atomic_load(&consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + (uint32_t)ORD_RING <= kk)
                __builtin_amdgcn_s_sleep(1);
            wave_sync();
            const uint64_t base = (uint64_t)(chunk0 + (size_t)t * ORD_GROUPS * group_stride);
            u32x4 rsrc;
            rsrc.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)(base & 0xffffffffu));
            rsrc.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffffu));
            rsrc.z = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ORD_GROUPS * group_stride));
            rsrc.w = 0x00020000u;
            const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (kk % ORD_RING) * (ORD_ROWS * 64)));
            const uint32_t so0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)so);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %1, %2, 0 offen" ODDIO_ORD_POLICY " lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %4 offen offset:1024" ODDIO_ORD_POLICY " lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %5 offen offset:2048" ODDIO_ORD_POLICY " lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %6 offen offset:3072" ODDIO_ORD_POLICY " lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(dst), "s"(so0), "s"(2u * so0), "s"(3u * so0) : "memory");
            const uint32_t dst2 = dst + 4096u;
            const int voff2 = voff + (int)(4u * group_stride);      // chunk 4 (8 * 128 KiB fits the 32-bit offset)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %1, %2, 0 offen" ODDIO_ORD_POLICY " lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %4 offen offset:1024" ODDIO_ORD_POLICY " lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %5 offen offset:2048" ODDIO_ORD_POLICY " lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %6 offen offset:3072" ODDIO_ORD_POLICY " lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff2), "s"(rsrc), "s"(dst2), "s"(so0), "s"(2u * so0), "s"(3u * so0) : "memory");
            if (mine >= (uint32_t)IN_FLIGHT) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * IN_FLIGHT) : "memory");      // everything but the newest IN_FLIGHT tiles has landed
                wave_sync();
                if (lane == 0) __hip_atomic_store(&landed[me], mine + 1u - (uint32_t)IN_FLIGHT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
        if (lane == 0) __hip_atomic_store(&landed[me], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    static_assert(ORD_ROWS * 64 == 8192, "a tile is 8 DMA instructions of 1 KiB");
    // ---- the adder ----
    const int fk = lane >> 2, m = lane & 3;
    const uint32_t f = cb * 16 + (uint32_t)fk;
    float s = (init != nullptr && f < n_frames) ? init[2 * f + e] : 0.0f;   // the buffered set's sum (walked first, spatial.rs:395-438)
#define ODDIO_ORD_WAIT(KK)                                                                                                \
    {   /* tile KK (counted from the top) is in its ring slot */                                                          \
        while (__hip_atomic_load(&landed[(KK) % ORD_LOADERS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= (KK) / ORD_LOADERS) __builtin_amdgcn_s_sleep(1); \
        wave_sync();                                                                                                      \
    }
#define ODDIO_ORD_DONE(KK)                                                                                                \
    {   /* every row of tile KK is in registers: its slot may be refilled */                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
        if (lane == 0) __hip_atomic_store(&consumed, (KK) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);          \
    }
    uint32_t kk = 0;
    if (n_tiles > 0 && (n % ORD_ROWS) != 0u) {
        // top tile of a set whose length is not a whole number of tiles: rows of sources >= n are not part of the sum
        ODDIO_ORD_WAIT(0u)
        const float* tp = &ring[0][0][fk];
        for (int r = (int)(n - (n_tiles - 1u) * ORD_ROWS) - 1; r >= 0; --r) s = s + tp[r * 16];
        ODDIO_ORD_DONE(0u)
        kk = 1;
    }
    // whole tiles, software pipelined over batches of ORD_Q quad steps: the LDS reads of batch b + 1 (of the next
    // tile across a tile boundary, once that tile has landed) are in flight while batch b is added
    float v[2][ORD_Q];
    const uint32_t lane_off = (uint32_t)m * 16u + (uint32_t)fk;          // floats: row m, frame fk
    // batch B of a tile = quad steps i = 31 - 8B .. 24 - 8B (rows 4i + m), read in descending order
    // (sched_barrier: the reads of the next batch are issued BEFORE the adds that cover their latency; hipcc otherwise
    // sinks them below the adds and waits for them right after issuing them)
#define ODDIO_ORD_LOAD(SET, TP, B)                                                                                        \
    _Pragma("unroll") for (int q = 0; q < ORD_Q; ++q) v[SET][q] = (TP)[(31 - ORD_Q * (B) - q) * 64];                      \
    __builtin_amdgcn_sched_barrier(0);
#define ODDIO_ORD_ADD(SET) chain_add8(s, v[SET]); __builtin_amdgcn_sched_barrier(0);
    if (kk < n_tiles) {
        ODDIO_ORD_WAIT(kk)
        const float* tp = &ring[kk % ORD_RING][0][0] + lane_off;
        ODDIO_ORD_LOAD(0, tp, 0)
        for (; kk < n_tiles; ++kk) {
            ODDIO_ORD_LOAD(1, tp, 1)
            ODDIO_ORD_ADD(0)
            ODDIO_ORD_LOAD(0, tp, 2)
            ODDIO_ORD_ADD(1)
            ODDIO_ORD_LOAD(1, tp, 3)
            const uint32_t* next_landed = &landed[(kk + 1u) % ORD_LOADERS];
            uint32_t seen = __hip_atomic_load(next_landed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // read under the adds below
            ODDIO_ORD_ADD(0)
            ODDIO_ORD_DONE(kk)
            if (kk + 1 < n_tiles) {
                while (seen <= (kk + 1u) / ORD_LOADERS) { __builtin_amdgcn_s_sleep(1); seen = __hip_atomic_load(next_landed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                wave_sync();
                tp = &ring[(kk + 1u) % ORD_RING][0][0] + lane_off;
                ODDIO_ORD_LOAD(0, tp, 0)
            }
            ODDIO_ORD_ADD(1)
        }
    }
#undef ODDIO_ORD_ADD
#undef ODDIO_ORD_LOAD
#undef ODDIO_ORD_DONE
#undef ODDIO_ORD_WAIT
    if (m == 0 && f < n_frames) out[2 * f + e] = postfx_apply(s, postfx);
}

__global__ void postfx_kernel(float* __restrict__ buf, uint32_t n, int postfx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = postfx_apply(buf[i], postfx);
}

// ---------------------------------------------------------------------------------------------
// Adapt (adapt.rs:63-87) as an epilogue over the stereo sum, followed by the Reinhard/Tanh filter:
//   avg_squared = sample^2 * alpha + avg_squared * (1 - alpha)   -- a rounding-exact recurrence over
// the output frames, so one lane walks it; everything around it (the channel sum, sample^2 * alpha,
// sqrt, the gain law, the scaling and the post filter) is done by the whole block.
// ---------------------------------------------------------------------------------------------
struct AdaptParams { float alpha, one_minus_alpha, max_gain, low, high; };
constexpr int ADAPT_CHUNK = 2048;

// CHANNELS: the frame type of the signal Adapt wraps -- [f32; 2] (scenes, stereo mixers) or f32 (Mixer<f32>)
template <int CHANNELS>
__global__ __launch_bounds__(256) void adapt_kernel(float* __restrict__ buf, uint32_t n_frames, AdaptParams A,
                                                    float* __restrict__ avg_squared, int postfx) {
    __shared__ __attribute__((aligned(16))) float drive[ADAPT_CHUNK];
    const uint32_t tid = threadIdx.x;
    float avg = *avg_squared;
    for (uint32_t base = 0; base < n_frames; base += ADAPT_CHUNK) {
        const uint32_t cnt = (n_frames - base) < (uint32_t)ADAPT_CHUNK ? (n_frames - base) : (uint32_t)ADAPT_CHUNK;
        for (uint32_t i = tid; i < cnt; i += 256) {
            float sample = 0.0f;                              // x.channels().iter().sum::<f32>()
#pragma unroll
            for (int c = 0; c < CHANNELS; ++c) sample = sample + buf[(size_t)(base + i) * CHANNELS + c];
            drive[i] = sample * sample * A.alpha;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t i = 0;
            for (; i + 8 <= cnt; i += 8) {                    // loads up front, then the dependent chain
                float d[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] = drive[i + k];
#pragma unroll
                for (int k = 0; k < 8; ++k) { avg = d[k] + avg * A.one_minus_alpha; d[k] = avg; }
#pragma unroll
                for (int k = 0; k < 8; ++k) drive[i + k] = d[k];
            }
            for (; i < cnt; ++i) { avg = drive[i] + avg * A.one_minus_alpha; drive[i] = avg; }
        }
        __syncthreads();
        for (uint32_t i = tid; i < cnt; i += 256) {
            const float avg_peak = sqrtf(drive[i]) * sqrtf(2.0f);
            float gain = 1.0f;
            if (avg_peak < A.low) gain = fminf(A.low / avg_peak, A.max_gain);   // f32::min: NaN-ignoring
            else if (avg_peak > A.high) gain = A.high / avg_peak;
#pragma unroll
            for (int c = 0; c < CHANNELS; ++c) {
                float* x = buf + (size_t)(base + i) * CHANNELS + c;
                *x = postfx_apply(*x * gain, postfx);
            }
        }
        __syncthreads();
    }
    if (tid == 0) *avg_squared = avg;
}

// Mixer<f32>: the mono mix is the left channel of the stereo pipeline's (a mono source duplicated by the implicit
// MonoToStereo adds the same value to both channels, signal.rs:73-80)
__global__ void take_left_kernel(const float* __restrict__ stereo, float* __restrict__ mono, uint32_t n_frames) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_frames) mono[i] = stereo[2 * i];
}

__global__ void zero_kernel(float* __restrict__ buf, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = 0.0f;
}

}  // namespace oddio_hip
