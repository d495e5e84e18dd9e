// buffered_kernels.h -- the buffered variant of the spatial path:
// `SpatialSceneControl::play_buffered` (src/spatial.rs:314-340), `Ring` (src/ring.rs:4-80) and the
// filters that are not `Seek` and therefore only enter a scene this way: `Gain` (+ `Smoothed`,
// src/gain.rs:58-127, src/smooth.rs) and `Speed` (src/speed.rs:26-40), besides `FixedGain`.
//
// Every source writes its contribution to its own [N][2] slab (ring write through the filter chain, then per ear /
// per 256-frame chunk ring reads with the f32 cursor and its wrap rule); `buffered_reduce` then adds the slabs in the
// reference's reverse-slot order, so the result is bit-identical to the reference's `o[ear] += s * gain` sequence for
// FramesSignal/Constant leaves.  `buffered_sources_wave` renders every shape one WAVE per source (scanner lanes
// replay the exact f32 running sums, 64 lanes expand them).
#pragma once
#include "kernels.h"

namespace oddio_hip {

enum : uint32_t { WRAP_FIXED_GAIN = 1, WRAP_GAIN = 2, WRAP_SPEED = 3, WRAP_REINHARD = 4, WRAP_TANH = 5 };   // == ODDIO_HIP_FILTER_*
constexpr int MAX_WRAP = 4;

struct alignas(16) BufStatic {
    const float* clip;      // leaf FramesSignal
    uint32_t clip_len;
    uint32_t clip_rate;
    float freq_or_value;    // leaf Sine (rad/s) / Constant
    uint32_t kind;          // KIND_*
    float* ring;            // Ring::buffer (ring.rs:5), device memory owned by the scene
    uint32_t ring_len;
    uint32_t rate;          // SpatialSignalBuffered::rate (spatial.rs:19)
    float max_delay;        // :20
    float radius;
    uint32_t n_wrap;        // filters, innermost first
    uint32_t wrap_kind[MAX_WRAP];
    float wrap_param[MAX_WRAP];   // FixedGain: linear gain
    uint32_t channels;      // 1 (mono) or 2 (interleaved stereo clip; Mixer general path only)
    uint32_t fader;         // Mixer general path: 1 + index of this source's FaderRec (0: not a Fader)
    uint32_t flags;         // BUF_FAST_OK (buffered_fast.h): set at play_buffered for the shapes buffered_write renders
};
static_assert(sizeof(BufStatic) == 96, "BufStatic layout");

struct alignas(16) BufDyn {
    SrcDyn common;          // clock / phase, Motion, State, finished_for, flags, id
    float ring_write;       // Ring::write (ring.rs:6)
    uint32_t stream_len;    // KIND_STREAM: spsc::Receiver::len (spsc.rs:121); Stream::t is common.phase
    uint32_t stream_stopping;   // Stream::stopping (stream.rs:12)
    uint32_t pad0;
    float shared[MAX_WRAP]; // Gain: atomically shared target (gain.rs:59) / Speed: factor (speed.rs:9)
    float sm_prev[MAX_WRAP];     // Smoothed<f32> of each Gain (smooth.rs:26-30)
    float sm_next[MAX_WRAP];
    float sm_progress[MAX_WRAP];
};
static_assert(sizeof(BufDyn) == 64 + 16 + 64, "BufDyn layout");

struct ControlUpdate { uint32_t slot; uint32_t index; float value; uint32_t pad; };

__device__ __forceinline__ size_t f32_as_usize(float x) {   // Rust `f32 as usize`
    if (!(x > 0.0f)) return 0;
    if (x >= 1.8446744e19f) return ~(size_t)0;
    return (size_t)x;
}
__device__ __forceinline__ float f32_rem_euclid(float a, float b) {   // core f32::rem_euclid
    const float r = fmodf(a, b);
    return r < 0.0f ? r + fabsf(b) : r;
}

// ---- the inner signal: leaf + filter chain, Signal::sample(interval, out[0..n]) -----------------
// `out` holds n frames of C = s.channels interleaved floats.
__device__ __forceinline__ float clip_ch(const float* clip, uint32_t len, uint32_t C, uint32_t ch, long long i) {
    return (i >= 0 && i < (long long)len) ? clip[(size_t)i * C + ch] : 0.0f;   // frames.rs:105-123
}

// ---- Fader (fader.rs:10-93) around a leaf + filter chain -----
// `next` is what swap::Receiver::received() holds after a refresh (the signal being faded to, or the
// retired one after a completed fade); `pend` is the control's flushed, not yet refreshed Command.
// Only the SIGNAL part of a BufStatic / BufDyn pair takes part (leaf, filters, clocks, smoothers);
// the slot's own state (ring, motion, flags, ids) stays where it is.
struct alignas(16) FaderPending {
    BufStatic st; BufDyn dyn;
    float duration;
    uint32_t fresh;         // swap.rs FRESH_BIT
    uint32_t gen;           // host-side generation of this command (the source's original signal is generation 0)
    uint32_t pad;
};
struct alignas(16) FaderRec {
    BufStatic next_st; BufDyn next_dyn;
    float progress;         // Fader::progress, 1.0 when no fade is running (fader.rs:21)
    float duration;         // Command::duration of `next`
    uint32_t cur_gen;       // generation of the signal playing as `inner`: every older command's signal is dead
    uint32_t next_gen;      // generation of `next`
    FaderPending pend;
};
constexpr uint32_t FADER_BUF = 1024;   // fader.rs:51

__device__ __forceinline__ void signal_assign(BufStatic& st, BufDyn& dyn, const BufStatic& from_st, const BufDyn& from_dyn) {
    st.clip = from_st.clip; st.clip_len = from_st.clip_len; st.clip_rate = from_st.clip_rate;
    st.freq_or_value = from_st.freq_or_value; st.kind = from_st.kind; st.channels = from_st.channels;
    st.n_wrap = from_st.n_wrap;
#pragma unroll
    for (int w = 0; w < MAX_WRAP; ++w) {
        st.wrap_kind[w] = from_st.wrap_kind[w]; st.wrap_param[w] = from_st.wrap_param[w];
        dyn.shared[w] = from_dyn.shared[w]; dyn.sm_prev[w] = from_dyn.sm_prev[w];
        dyn.sm_next[w] = from_dyn.sm_next[w]; dyn.sm_progress[w] = from_dyn.sm_progress[w];
    }
    dyn.common.t = from_dyn.common.t; dyn.common.phase = from_dyn.common.phase;
    dyn.stream_len = from_dyn.stream_len; dyn.stream_stopping = from_dyn.stream_stopping;
}

// ---- wave-per-source formulation of the buffered set and of the Mixer's general path ------------------
// play_buffered(filters(leaf)) with any FixedGain / Gain / Speed chain (what Gain and Speed sources, the reason
// the buffered path exists, look like), optionally inside a Fader.  The reference's loops are sequential only in
// their f32 running sums -- the leaf's cursor `offset += ds` (frames.rs:189-196), each Gain's
// `progress = min(progress + step, 1)` (gain.rs:114-120, smooth.rs:47-49) and Ring::sample's cursor with its
// wrap rewrite (ring.rs:59-78).  One lane per running sum replays it exactly and drops a checkpoint every 16
// steps in LDS; then all 64 lanes restart from their checkpoints and produce 16 frames each, like the mix
// kernel's phase A / phase B.  Stream, Sine and Constant leaves and the FramesSignal fast path have closed-form
// cursors and need no scan at all.  (Rounds 1-2 rendered these one thread per source: same bits, ~100x the critical path.)
__device__ __forceinline__ void wg_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// inner.sample(interval, out[0..n]) for leaf + filter chain; every lane holds the same `s` / `d`.
// Leaves: FramesSignal and Cycle (mono or interleaved stereo clip), Stream, Sine, Constant.  `out` holds n frames of
// C = s.channels interleaved floats.  ck rows: 0 leaf cursor, 1-4 Gain progress, 5 Cycle base (6: the Fader's progress).
__device__ __forceinline__ void inner_sample_wave(const BufStatic& s, BufDyn& d, float interval, float* out, uint32_t n, float (*ck)[64], int lane) {
    // (every loop over the filter chain is unrolled over MAX_WRAP with compile-time indices: arrays indexed by a runtime
    // filter number live in scratch memory, 256 bytes per lane before)
    const uint32_t C = s.channels == 2u ? 2u : 1u;
    float level_interval[MAX_WRAP];
    float cur = interval;
#pragma unroll
    for (int w = MAX_WRAP - 1; w >= 0; --w) {
        level_interval[w] = cur;
        if ((uint32_t)w < s.n_wrap && s.wrap_kind[w] == WRAP_SPEED) cur = cur * d.shared[w];   // speed.rs:32-35
    }
    const bool frm = s.kind == KIND_FRAMES, cyc = s.kind == KIND_CYCLE, strm = s.kind == KIND_STREAM, sine = s.kind == KIND_SINE;
    // leaf FramesSignal: frames.rs:176-201, Cycle: cycle.rs:26-53 (d.common.t is then the cursor in samples; ck row 5 holds `base`)
    const double s0 = d.common.t * (double)s.clip_rate;
    const float ds = cur * (float)s.clip_rate;
    const long long base = f64_as_isize(s0);
    const bool fast = frm && fabsf(ds - 1.0f) <= FLT_EPSILON;
    const float frac0 = (float)(s0 - (double)base);
    uint32_t cbase = cyc ? (uint32_t)f64_as_isize(d.common.t) : 0u;        // cycle.rs:28
    float coff = cyc ? (float)(d.common.t - (double)cbase) : 0.0f;          // :29
    // leaf Stream (stream.rs:69-85): the spsc ring in pinned host memory.  The header words are taken from lane 0:
    // its loads follow its own release store of `read` in an earlier call, and every lane must see the same producer state.
    StreamHeader* hdr = nullptr;
    uint32_t q_size = 0, q_read = 0, q_len = 0;
    const float phase0 = d.common.phase;                                     // Stream::t / Sine::phase at the start of the call
    if (strm) {
        hdr = reinterpret_cast<StreamHeader*>(const_cast<float*>(s.clip)) - 1;
        q_size = s.clip_len;                                                 // capacity + 1 (spsc.rs:12)
        q_read = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr->read);   // only lane 0 ever stores it
        const uint32_t write = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&hdr->write, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM));
        q_len = write >= q_read ? write - q_read : write + q_size - q_read; // update(): readable_len, spsc.rs:218-225
        if (q_len < d.stream_len) q_len = d.stream_len;                      // never shrinks (debug_assert, spsc.rs:131)
        if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&hdr->closed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))) d.stream_stopping = 1u;   // :71-73
    }
    // Gain: Smoothed::set when the shared target moved (gain.rs:106-109), then ramp or constant
    bool ramp[MAX_WRAP];
    float gconst[MAX_WRAP], step[MAX_WRAP];
    bool any_ramp = false;
#pragma unroll
    for (uint32_t w = 0; w < MAX_WRAP; ++w) {
        ramp[w] = false; gconst[w] = 1.0f; step[w] = 0.0f;
        if (w < s.n_wrap && s.wrap_kind[w] == WRAP_GAIN) {
            const float shared = d.shared[w];
            if (d.sm_next[w] != shared) {
                d.sm_prev[w] = d.sm_prev[w] + d.sm_progress[w] * (d.sm_next[w] - d.sm_prev[w]);
                d.sm_next[w] = shared;
                d.sm_progress[w] = 0.0f;
            }
            ramp[w] = d.sm_progress[w] != 1.0f;
            any_ramp = any_ramp || ramp[w];
            gconst[w] = d.sm_prev[w] + d.sm_progress[w] * (d.sm_next[w] - d.sm_prev[w]);
            step[w] = level_interval[w] / 0.1f;                       // SMOOTHING_PERIOD, gain.rs:163
        }
    }
    // running sums: lane 0 the leaf cursor (FramesSignal slow path, Cycle), lane 1 + w the progress of Gain w; a chain
    // with no running sum at all (fast path, Stream, Sine, Constant, settled gains) skips the scan
    const bool cursor = (frm && !fast) || cyc;
    const bool any_scan = cursor || any_ramp;                                // wave-uniform
    const bool is_gain = lane >= 1 && lane <= MAX_WRAP;
    float scan = 0.0f, inc = 0.0f;
    if (lane == 0) { scan = frac0; inc = ds; }
#pragma unroll
    for (int w = 0; w < MAX_WRAP; ++w) if (lane == 1 + w) { scan = d.sm_progress[w]; inc = step[w]; }
    for (uint32_t p0 = 0; p0 < n; p0 += 1024u) {
        const uint32_t m = (n - p0) < 1024u ? (n - p0) : 1024u;
        const uint32_t nb = (m + 15u) / 16u;
        if (any_scan) {
            if (lane == 0 && cyc) {
                for (uint32_t b = 0; b < nb; ++b) {
                    ck[0][b] = coff;
                    ck[5][b] = __uint_as_float(cbase);
                    const uint32_t cnt = (m - 16u * b) < 16u ? (m - 16u * b) : 16u;
                    for (uint32_t i = 0; i < cnt; ++i) { uint32_t ia, ib; float fr; cycle_step(cbase, coff, s.clip_len, ds, ia, ib, fr); }
                }
            } else if (lane <= MAX_WRAP) {
                for (uint32_t b = 0; b < nb; ++b) {
                    ck[lane][b] = scan;
                    const uint32_t cnt = (m - 16u * b) < 16u ? (m - 16u * b) : 16u;
                    for (uint32_t i = 0; i < cnt; ++i) {
                        const float v = scan + inc;
                        scan = is_gain ? fminf(v, 1.0f) : v;
                    }
                }
            }
            wg_sync();
        }
        if (16u * (uint32_t)lane < m) {
            const uint32_t f0 = p0 + 16u * (uint32_t)lane;
            const uint32_t cnt = (m - 16u * (uint32_t)lane) < 16u ? (m - 16u * (uint32_t)lane) : 16u;
            float off = cursor ? ck[0][lane] : 0.0f;
            uint32_t cb_ = cyc ? __float_as_uint(ck[5][lane]) : 0u;
            float pr[MAX_WRAP];
#pragma unroll
            for (int w = 0; w < MAX_WRAP; ++w) pr[w] = ramp[w] ? ck[1 + w][lane] : 1.0f;
            for (uint32_t k = 0; k < cnt; ++k) {
                float v0, v1 = 0.0f;
                if (cyc) {                                                                         // cycle.rs:30-50
                    uint32_t ia, ib;
                    float fr;
                    cycle_step(cb_, off, s.clip_len, ds, ia, ib, fr);
                    const float a = s.clip[(size_t)ia * C], b = s.clip[(size_t)ib * C];
                    v0 = a + fr * (b - a);
                    if (C == 2u) { const float a1 = s.clip[(size_t)ia * 2 + 1], b1 = s.clip[(size_t)ib * 2 + 1]; v1 = a1 + fr * (b1 - a1); }
                } else if (frm) {
                    long long idx;
                    float fr;
                    if (fast) { idx = base + (long long)(f0 + k); fr = frac0; }                   // frames.rs:180-187
                    else { const long long tr = (long long)off; idx = base + tr; fr = off - (float)tr; off = off + ds; }   // :189-196
                    const float a = clip_ch(s.clip, s.clip_len, C, 0u, idx), b = clip_ch(s.clip, s.clip_len, C, 0u, idx + 1);
                    v0 = a + fr * (b - a);
                    if (C == 2u) { const float a1 = clip_ch(s.clip, s.clip_len, 2u, 1u, idx), b1 = clip_ch(s.clip, s.clip_len, 2u, 1u, idx + 1); v1 = a1 + fr * (b1 - a1); }
                } else if (strm) {                                                                 // stream.rs:76-84
                    const float sv = phase0 + ds * (float)(f0 + k);                                // :78
                    const float x0f = truncf(sv);
                    const long long x0 = f64_as_isize((double)x0f);                                // s.trunc() as isize
                    const float fr = sv - x0f;                                                     // f32::fract
                    const bool ina = x0 >= 0 && x0 < (long long)q_len, inb = x0 + 1 >= 0 && x0 + 1 < (long long)q_len;   // Stream::get: zero outside [0, len)
                    const size_t xa = (size_t)((q_read + (uint32_t)x0) % q_size), xb = (size_t)((q_read + (uint32_t)(x0 + 1)) % q_size);
                    const float a = ina ? s.clip[xa * C] : 0.0f, b = inb ? s.clip[xb * C] : 0.0f;
                    v0 = a + fr * (b - a);
                    if (C == 2u) { const float a1 = ina ? s.clip[xa * 2 + 1] : 0.0f, b1 = inb ? s.clip[xb * 2 + 1] : 0.0f; v1 = a1 + fr * (b1 - a1); }
                } else if (sine) {                                                                 // sine.rs:34-38
                    const float t = cur * (float)(f0 + k);
                    v0 = sinf(t * s.freq_or_value + phase0);
                } else {                                                                           // constant.rs:16-18
                    v0 = s.freq_or_value; v1 = s.freq_or_value;
                }
#pragma unroll
                for (uint32_t w = 0; w < MAX_WRAP; ++w) {
                    if (w >= s.n_wrap) continue;
                    if (s.wrap_kind[w] == WRAP_FIXED_GAIN) { v0 = v0 * s.wrap_param[w]; v1 = v1 * s.wrap_param[w]; }   // gain.rs:32-37
                    else if (s.wrap_kind[w] == WRAP_REINHARD) { v0 = v0 / (1.0f + fabsf(v0)); v1 = v1 / (1.0f + fabsf(v1)); }   // reinhard.rs:28-35 (per channel)
                    else if (s.wrap_kind[w] == WRAP_TANH) { v0 = tanhf(v0); v1 = tanhf(v1); }                                // tanh.rs:22-29
                    else if (s.wrap_kind[w] == WRAP_GAIN) {                                       // gain.rs:110-121
                        if (ramp[w]) {
                            const float g = d.sm_prev[w] + pr[w] * (d.sm_next[w] - d.sm_prev[w]);
                            v0 = v0 * g; v1 = v1 * g;
                            pr[w] = fminf(pr[w] + step[w], 1.0f);
                        } else if (gconst[w] != 1.0f) { v0 = v0 * gconst[w]; v1 = v1 * gconst[w]; }
                    }
                }
                if (C == 2u) { out[(size_t)(f0 + k) * 2] = v0; out[(size_t)(f0 + k) * 2 + 1] = v1; }
                else out[f0 + k] = v0;
            }
        }
        wg_sync();   // before the next pass overwrites the checkpoints; `out` is read back by other lanes (ring reads, Fader)
    }
#pragma unroll
    for (int w = 0; w < MAX_WRAP; ++w) {
        const float fin = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(scan), 1 + w));
        if (ramp[w]) d.sm_progress[w] = fin;
    }
    if (cyc) {   // cycle.rs:52: the cursor the scanner lane ended on, for every lane
        const uint32_t fb = (uint32_t)__builtin_amdgcn_readlane((int)cbase, 0);
        const float fo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coff), 0));
        d.common.t = (double)fb + (double)fo;
    } else if (frm) {
        d.common.t = d.common.t + (double)cur * (double)n;                                        // frames.rs:198
    } else if (strm) {   // advance(interval * out.len() as f32), stream.rs:59-64
        const float next = phase0 + (cur * (float)n) * (float)s.clip_rate;
        const float t = fminf(next, (float)q_len);
        uint32_t rel = (uint32_t)f32_as_usize(t);
        if (rel > q_len) rel = q_len;
        if (rel && lane == 0) __hip_atomic_store(&hdr->read, (q_read + rel) % q_size, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // release(), spsc.rs:227-235
        d.stream_len = q_len - rel;
        d.common.phase = t - truncf(t);
    } else if (sine) {
        d.common.phase = fmodf(phase0 + (cur * (float)n) * s.freq_or_value, ODDIO_TAU);           // sine.rs:39
    }
}

// Fader::sample (fader.rs:36-73), one wave.  Returns false when no fade is running and none is waiting: the caller
// renders the plain signal (the fast path, fader.rs:37-45).  `scratch` holds FADER_BUF frames of the source's channel
// count.  The fade's progress is a running f32 sum like a Gain's: lane 0 replays it and drops a checkpoint every 16
// frames (ck row 6), the lanes then cross-fade the 16 frames they rendered themselves.
__device__ __forceinline__ bool fader_sample_wave(BufStatic& st, BufDyn& dyn, FaderRec& F, float* scratch, float interval, float* out, uint32_t n,
                                                  float (*ck)[64], int lane) {
    const uint32_t C = st.channels == 2u ? 2u : 1u;
    float progress = F.progress;
    const bool refresh = progress >= 1.0f;
    if (refresh && !F.pend.fresh) return false;
    BufStatic nst;                                        // private copies, written back once
    BufDyn ndyn;
    float duration;
    uint32_t next_gen;
    if (refresh) { nst = F.pend.st; ndyn = F.pend.dyn; duration = F.pend.duration; next_gen = F.pend.gen; progress = 0.0f; }   // self.next.refresh()
    else { nst = F.next_st; ndyn = F.next_dyn; duration = F.duration; next_gen = F.next_gen; }
    wg_sync();                                            // every lane has read the record before lane 0 rewrites it
    if (refresh && lane == 0) {
        // received = pending, record to record in 16-byte pieces (assigned from the register copies, hipcc stages the
        // structs through scratch memory)
        const uint4* from_st = reinterpret_cast<const uint4*>(&F.pend.st);
        const uint4* from_dyn = reinterpret_cast<const uint4*>(&F.pend.dyn);
        uint4* to_st = reinterpret_cast<uint4*>(&F.next_st);
        uint4* to_dyn = reinterpret_cast<uint4*>(&F.next_dyn);
        for (uint32_t q = 0; q < sizeof(BufStatic) / 16; ++q) to_st[q] = from_st[q];
        for (uint32_t q = 0; q < sizeof(BufDyn) / 16; ++q) to_dyn[q] = from_dyn[q];
        F.duration = duration; F.next_gen = next_gen; F.pend.fresh = 0u;
    }
    const float increment = interval / duration;
    uint32_t off = 0;
    while (off < n) {
        const uint32_t rem = n - off;
        const uint32_t m = rem < FADER_BUF ? rem : FADER_BUF;
        inner_sample_wave(st, dyn, interval, scratch, FADER_BUF, ck, lane);                    // the whole buffer, fader.rs:53
        inner_sample_wave(nst, ndyn, interval, out + (size_t)off * C, rem, ck, lane);          // all that is left, :54
        float pscan = progress;
        if (lane == 0) {
            for (uint32_t b = 0; 16u * b < m; ++b) {
                ck[6][b] = pscan;
                const uint32_t cnt = (m - 16u * b) < 16u ? (m - 16u * b) : 16u;
                for (uint32_t i = 0; i < cnt; ++i) pscan = fminf(pscan + increment, 1.0f);
            }
        }
        wg_sync();
        if (16u * (uint32_t)lane < m) {
            const uint32_t cnt = (m - 16u * (uint32_t)lane) < 16u ? (m - 16u * (uint32_t)lane) : 16u;
            float p = ck[6][lane];
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t f = 16u * (uint32_t)lane + k;
                const float fade_out = sqrtf(1.0f - p);
                const float fade_in = sqrtf(p);
                for (uint32_t ch = 0; ch < C; ++ch) {
                    float* o = out + (size_t)(off + f) * C + ch;
                    *o = scratch[(size_t)f * C + ch] * fade_out + *o * fade_in;                    // frame::mix(scale(x), scale(o))
                }
                p = fminf(p + increment, 1.0f);
            }
        }
        progress = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pscan)));
        wg_sync();                                        // the next pass renders over what this one mixed
        off += m;
    }
    if (lane == 0) F.progress = progress;
    if (progress >= 1.0f) {   // mem::swap(&mut self.inner, &mut next.fade_to)
        const BufStatic old_st = st;
        const BufDyn old_dyn = dyn;
        signal_assign(st, dyn, nst, ndyn);
        if (lane == 0) {
            signal_assign(F.next_st, F.next_dyn, old_st, old_dyn);
            const uint32_t g = F.cur_gen; F.cur_gen = next_gen; F.next_gen = g;
        }
    } else if (lane == 0) {
        // next.fade_to keeps what its sampler advanced (the record already holds every other field; a whole-struct store
        // keeps those alive in scratch memory)
        F.next_dyn.common.t = ndyn.common.t; F.next_dyn.common.phase = ndyn.common.phase;
        F.next_dyn.stream_len = ndyn.stream_len; F.next_dyn.stream_stopping = ndyn.stream_stopping;
#pragma unroll
        for (int w = 0; w < MAX_WRAP; ++w) {
            F.next_dyn.sm_prev[w] = ndyn.sm_prev[w]; F.next_dyn.sm_next[w] = ndyn.sm_next[w]; F.next_dyn.sm_progress[w] = ndyn.sm_progress[w];
        }
    }
    wg_sync();   // a second call in the same callback (the ring write wraps) reads the record lane 0 has just written
    return true;
}

// The rendering of one buffered source after its walk, one wave: Ring::write through the filter chain (ring.rs:18-41),
// then Ring::sample per ear and 256-frame chunk (ring.rs:51-79, spatial.rs:409-431) into the source's slab row `my`
// ([frame][ear], `s * gain`).  Every lane holds the same `s` / `d`; per ear: prev_offset, dt, g0, d_gain (spatial.rs:412-421).
// Also keeps the mirror behind the ring's end (buffered_fast.h: the first RING_MIRROR samples repeated) up to date.
constexpr uint32_t RING_MIRROR = 640;            // >= WIN_CAP + 4: a tile's window of spatial_mix<.., RING>, started anywhere before the ring's end
__device__ __forceinline__ void buffered_render_wave(const SceneParams& P, BufStatic& s, BufDyn& d, float po0, float dt0, float g00, float dg0,
                                                     float po1, float dt1, float g01, float dg1, float* __restrict__ my,
                                                     FaderRec* __restrict__ faders, float* __restrict__ fader_scratch, float (*ck)[64], int lane) {
    const float elapsed = P.elapsed;
    const uint32_t n = P.n_frames;
    float* ring = s.ring;
    const uint32_t len = s.ring_len;
    const float prev_offset[2] = {po0, po1}, dts[2] = {dt0, dt1}, g0s[2] = {g00, g01}, dgs[2] = {dg0, dg1};
    {   // Ring::write (ring.rs:18-41): extend the delay queue with new data
        const float end = fmodf(d.ring_write + elapsed * (float)s.rate, (float)len);
        const size_t start_idx = f32_as_usize(ceilf(d.ring_write));
        const size_t end_idx = f32_as_usize(ceilf(end));
        const float interval = 1.0f / (float)s.rate;
        // one or (when the write wraps) two stretches of the ring; one call site, so that the sampler is inlined once
        const int n_seg = end_idx > start_idx ? 1 : 2;
        for (int sg = 0; sg < n_seg; ++sg) {
            float* o = sg == 0 ? ring + start_idx : ring;
            const uint32_t cnt = n_seg == 1 ? (uint32_t)(end_idx - start_idx) : (sg == 0 ? (uint32_t)(len - start_idx) : (uint32_t)end_idx);
            bool faded = false;
            if (s.fader) faded = fader_sample_wave(s, d, faders[s.fader - 1u], fader_scratch + (size_t)(s.fader - 1u) * FADER_BUF, interval, o, cnt, ck, lane);
            if (!faded) inner_sample_wave(s, d, interval, o, cnt, ck, lane);
        }
        d.ring_write = end;
        if (n_seg == 2 || start_idx < RING_MIRROR) {   // the write touched the ring's first samples: repeat them behind its end
            wg_sync();
            for (uint32_t q = (uint32_t)lane; q < RING_MIRROR && q < len; q += 64u) ring[len + q] = ring[q];
        }
    }
    wg_sync();   // the ring samples written above are read back by other lanes below

    // Ring::sample (ring.rs:51-79) per ear and 256-frame chunk (spatial.rs:424): 4 chunks x 2 ears per pass
    auto ring_step = [&](float& offset, float ds_, float& a, float& b, float& fract) {
        size_t x = (size_t)offset;
        fract = offset - (float)x;
        if (x < (size_t)len - 1) { a = ring[x]; b = ring[x + 1]; }
        else if (x < (size_t)len) { a = ring[x]; b = ring[0]; }
        else {
            x = x % len;
            offset = (float)x + fract;
            if (x < (size_t)len - 1) { a = ring[x]; b = ring[x + 1]; }
            else { a = ring[x]; b = ring[0]; }
        }
        offset = offset + ds_;
    };
    for (uint32_t pass0 = 0; pass0 < n; pass0 += 1024u) {
        if (lane < 8) {
            const int e = lane >> 2;
            const uint32_t done = pass0 + 256u * (uint32_t)(lane & 3);
            if (done < n) {
                const uint32_t clen = (n - done) < 256u ? (n - done) : 256u;
                const float t = prev_offset[e] + (float)done * dts[e];                 // idx == done at the chunk's start (spatial.rs:424)
                float offset = f32_rem_euclid(d.ring_write + t * (float)s.rate, (float)len);
                const float ds_ = dts[e] * (float)s.rate;
                for (uint32_t b = 0; b * 16u < clen; ++b) {
                    ck[lane][b] = offset;
                    const uint32_t cnt = (clen - 16u * b) < 16u ? (clen - 16u * b) : 16u;
                    for (uint32_t k = 0; k < cnt; ++k) { float a, bb, fr; ring_step(offset, ds_, a, bb, fr); }
                }
            }
        }
        wg_sync();
        for (int e = 0; e < 2; ++e) {
            const int cq = lane >> 4, b = lane & 15;
            const uint32_t done = pass0 + 256u * (uint32_t)cq;
            const uint32_t f0 = done + 16u * (uint32_t)b;
            if (f0 < n) {
                const uint32_t clen = (n - done) < 256u ? (n - done) : 256u;
                const uint32_t cnt = (clen - 16u * (uint32_t)b) < 16u ? (clen - 16u * (uint32_t)b) : 16u;
                float offset = ck[e * 4 + cq][b];
                const float ds_ = dts[e] * (float)s.rate;
                for (uint32_t k = 0; k < cnt; ++k) {
                    float a, bb, fr;
                    ring_step(offset, ds_, a, bb, fr);
                    const float v = a + fr * (bb - a);
                    const float gain = g0s[e] + (float)(f0 + k) * dgs[e];              // spatial.rs:426
                    my[2 * (f0 + k) + e] = v * gain;
                }
            }
        }
        wg_sync();
    }
}

// One wave per buffered slot, every shape the ABI accepts (leaf FramesSignal / Cycle / Stream / Sine / Constant under any
// FixedGain / Gain / Speed chain, optionally inside a Fader).
__global__ __launch_bounds__(64) void buffered_sources_wave(SceneParams P, const uint32_t* __restrict__ d_len_b, BufStatic* __restrict__ st,
                                                            BufDyn* __restrict__ dyn, SrcPending* __restrict__ pend,
                                                            float* __restrict__ contrib, uint32_t* __restrict__ skip,
                                                            uint32_t* __restrict__ stopped_hdr, uint32_t stopped_cap,
                                                            FaderRec* __restrict__ faders, float* __restrict__ fader_scratch) {
    __shared__ float ck[8][64];
    const uint32_t i = blockIdx.x;
    const int lane = threadIdx.x;
    if (i >= d_len_b[0]) return;
    BufStatic s = st[i];
    BufDyn d = dyn[i];
    SrcDyn& c = d.common;
    if (c.flags & DYN_STOPPED) { if (lane == 0) skip[i] = 1; return; }
    const float elapsed = P.elapsed;
    const uint32_t n = P.n_frames;
    const float nf = (float)n;
    V3 tpos = {c.tgt_pos[0], c.tgt_pos[1], c.tgt_pos[2]};
    V3 tvel = {c.tgt_vel[0], c.tgt_vel[1], c.tgt_vel[2]};
    V3 ppos = {c.prev_pos[0], c.prev_pos[1], c.prev_pos[2]};
    const SrcPending pm = pend[i];
    if (pm.flags & PEND_FRESH) {   // spatial.rs:216-226
        V3 npos = {pm.pos[0], pm.pos[1], pm.pos[2]};
        V3 nvel = {pm.vel[0], pm.vel[1], pm.vel[2]};
        ppos = (pm.flags & PEND_DISCONTINUITY) ? npos : smoothed_position(ppos, c.state_dt, 0.0f, tpos, tvel);
        tpos = npos; tvel = nvel;
        c.state_dt = 0.0f;
    }
    const Quat prev_rot = {P.prev_rot[0], P.prev_rot[1], P.prev_rot[2], P.prev_rot[3]};
    const Quat rot = {P.rot[0], P.rot[1], P.rot[2], P.rot[3]};
    const V3 p0 = quat_rotate(prev_rot, smoothed_position(ppos, c.state_dt, 0.0f, tpos, tvel));
    const V3 p1 = quat_rotate(rot, smoothed_position(ppos, c.state_dt, elapsed, tpos, tvel));
    c.state_dt = c.state_dt + elapsed;
    c.tgt_pos[0] = tpos.x; c.tgt_pos[1] = tpos.y; c.tgt_pos[2] = tpos.z;
    c.tgt_vel[0] = tvel.x; c.tgt_vel[1] = tvel.y; c.tgt_vel[2] = tvel.z;
    c.prev_pos[0] = ppos.x; c.prev_pos[1] = ppos.y; c.prev_pos[2] = ppos.z;
    // spatial.rs:243-261
    const float distance = v3_norm(p0);
    if (c.flags & DYN_HAS_FINISHED_FOR) {
        if (c.finished_for > distance / ODDIO_SPEED_OF_SOUND) c.flags |= DYN_STOPPED;
        else c.finished_for = c.finished_for + elapsed;
    } else {
        bool fin = false;
        if (!s.fader) {   // Fader::is_finished is always false (fader.rs:76-79); is_finished passes through the filters; a Cycle never finishes
            if (s.kind == KIND_FRAMES) fin = c.t >= (double)(s.clip_len - 1u) / (double)s.clip_rate;
            if (s.kind == KIND_STREAM) fin = d.stream_stopping && c.phase == (float)d.stream_len;       // stream.rs:88-90
        }
        if (fin) { c.flags |= DYN_HAS_FINISHED_FOR; c.finished_for = elapsed; }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0 && (pm.flags & PEND_FRESH)) pend[i].flags = 0;
    if (c.flags & DYN_STOPPED) {
        if (lane == 0) {
            const uint32_t k = atomicAdd(&stopped_hdr[0], 1u);
            if (k < stopped_cap) stopped_hdr[1 + k] = c.id;
            skip[i] = 1;
            dyn[i] = d;
        }
        return;
    }
    if (lane == 0) skip[i] = 0;
    float po[2], dtv[2], g0v[2], dgv[2];
    for (int e = 0; e < 2; ++e) {   // spatial.rs:409-423
        float off0, g0, off1, g1;
        ear_state(p0, e, s.radius, off0, g0);
        ear_state(p1, e, s.radius, off1, g1);
        po[e] = fmaxf(off0 - elapsed, -s.max_delay);
        const float next_offset = fmaxf(off1, -s.max_delay);
        dtv[e] = (next_offset - po[e]) / nf;
        dgv[e] = (g1 - g0) / nf;
        g0v[e] = g0;
    }
    buffered_render_wave(P, s, d, po[0], dtv[0], g0v[0], dgv[0], po[1], dtv[1], g0v[1], dgv[1], contrib + (size_t)i * 2 * n, faders, fader_scratch, ck, lane);
    if (s.fader && lane == 0) st[i] = s;   // a completed fade swapped the signals
    if (lane == 0) dyn[i] = d;
}

// out_b[o] = ((0 + contrib[last]) + ... + contrib[0]) : the reference's reverse walk (spatial.rs:204).
// grid = (ceil(n_out / 64), n_slices), block = 256.  Slice k sums slots [k*per, (k+1)*per) in descending
// order into part[k][o]; buffered_reduce_finish adds the slices, last slice first.  ORDERED mode uses ONE
// slice (the reference's exact sequence); FAST mode 32 (a deterministic tree, like the seekable set's).
// Rows are staged through LDS 64 at a time so that the sequential adds do not wait for HBM one by one.
constexpr int BUFRED_ROWS = 64;
__global__ __launch_bounds__(256) void buffered_reduce(const float* __restrict__ contrib, const uint32_t* __restrict__ skip,
                                                       const uint32_t* __restrict__ d_len_b, uint32_t n_frames, float* __restrict__ part,
                                                       uint32_t n_slices) {
    __shared__ float tile[BUFRED_ROWS][65];
    __shared__ uint32_t sk[BUFRED_ROWS];
    const uint32_t n_out = 2 * n_frames;
    const uint32_t ox = threadIdx.x & 63, sy = threadIdx.x >> 6;
    const uint32_t o = blockIdx.x * 64 + ox;
    const uint32_t len = d_len_b[0];
    const uint32_t per = (len + n_slices - 1) / n_slices;
    const uint32_t lo = blockIdx.y * per;
    uint32_t hi = lo + per < len ? lo + per : len;
    float s = 0.0f;
    while (hi > lo) {
        const uint32_t base = hi - lo >= (uint32_t)BUFRED_ROWS ? hi - BUFRED_ROWS : lo;
        const uint32_t cnt = hi - base;
        for (uint32_t r = sy; r < cnt; r += 4) tile[r][ox] = o < n_out ? contrib[(size_t)(base + r) * n_out + o] : 0.0f;
        if (threadIdx.x < cnt) sk[threadIdx.x] = skip[base + threadIdx.x];
        __syncthreads();
        if (sy == 0)
            for (uint32_t r = cnt; r-- > 0;)
                if (!sk[r]) s = s + tile[r][ox];
        __syncthreads();
        hi = base;
    }
    if (sy == 0 && o < n_out) part[(size_t)blockIdx.y * n_out + o] = s;
}
__global__ void buffered_reduce_finish(const float* __restrict__ part, uint32_t n_slices, uint32_t n_frames, float* __restrict__ out_b) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_out = 2 * n_frames;
    if (o >= n_out) return;
    float s = part[(size_t)(n_slices - 1) * n_out + o];          // the walk starts at the last slot
    for (uint32_t k = n_slices - 1; k-- > 0;) s = s + part[(size_t)k * n_out + o];
    out_b[o] = s;
}

// GainControl / SpeedControl stores by slot: one thread per update.  The host keeps only the LAST value per (slot, filter)
// of a callback (relaxed "latest value" stores, gain.rs:158-160), so the updates are independent.
__global__ void apply_control_updates(const ControlUpdate* __restrict__ up, uint32_t n, BufDyn* __restrict__ dyn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dyn[up[i].slot].shared[up[i].index & (MAX_WRAP - 1)] = up[i].value;
}

struct BufMove { uint32_t dst, src; };

// The generation each Fader is playing, for the control thread (pinned host memory): clips handed over by
// older fade_to commands can be released (the reference drops a retired signal at the next fade_to).
__global__ void publish_fader_gens(const FaderRec* __restrict__ faders, uint32_t n, uint32_t* __restrict__ host_gen) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) host_gen[i] = faders[i].cur_gen;
}

// out[o] = postfx(in[o])  (scenes that hold only buffered sources)
__global__ void copy_postfx_kernel(const float* __restrict__ in, float* __restrict__ out, uint32_t n, int postfx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = postfx_apply(in[i], postfx);
}

}  // namespace oddio_hip
