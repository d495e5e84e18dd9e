// buffered_fast.h -- the buffered set (play_buffered, src/spatial.rs:314-340,395-433) at scale.
//
// The general kernel (buffered_kernels.h: one wave per source, a slab row per source, a separate slab sum) renders
// every shape the ABI accepts; this file is the path for the shapes Gain and Speed sources actually have -- a mono
// FramesSignal under any FixedGain / Gain / Speed chain -- built like the Seek set's:
//
//   buffered_walk    1 thread / slot: walk_set (spatial.rs:191-265), both EarStates, and everything of the callback that
//                    is scalar: Ring::write's stretch of the ring (ring.rs:18-41), the leaf's cursor set-up per call
//                    (frames.rs:176-181), Smoothed::set (gain.rs:106-109), and the records of the two kernels below.
//                    Sources it cannot describe (other leaves, Fader, out-of-range cursors) go on a list for the
//                    general kernel, whose slab row the mix adds at the source's place.
//   buffered_write   Ring::write through the filter chain.  A wave takes 16 sources at a time: the exact f32 running
//                    sums -- the leaf cursor `offset += ds` (frames.rs:189-196, 1024 steps) and each ramping Gain's
//                    `progress = min(progress + step, 1)` (smooth.rs:47-49) -- are replayed one stream per lane with a
//                    checkpoint every 16 frames; then, per source, the leaf window goes HBM -> LDS directly
//                    (buffer_load ... lds), the 64 lanes render 16 frames each from their checkpoints and store them
//                    to the ring.
//   spatial_mix<.., RING>  (kernels.h) Ring::sample per ear and 256-frame chunk (ring.rs:51-79, spatial.rs:409-431):
//                    the same kernel as the Seek set's, fed with tile records whose window is a stretch of the ring.
//                    FAST mode accumulates in registers across sources (no slab, no slab sum); ORDERED mode writes
//                    contribution rows that ordered_sum adds in the reference's order.
//
// A ring is allocated with RING_MIRROR extra floats behind its last sample that repeat its first ones (kept up to
// date by whoever writes indices < RING_MIRROR): a read window that passes the ring's end is then one linear stretch
// of memory, and `(b[len-1], b[0])` (ring.rs:63-65) is an ordinary adjacent pair.
#pragma once
#include "buffered_kernels.h"

namespace oddio_hip {

static_assert(RING_MIRROR >= WIN_CAP + 4 && RING_MIRROR % 4 == 0, "a mix window fits the mirror");
constexpr uint32_t BUF_FAST_OK = 1u;             // BufStatic::flags: the shape is one buffered_write renders
constexpr uint32_t RING_FAST_MIN = 2048, RING_FAST_MAX = 1u << 24;   // ring lengths the fast path takes ((float)len exact; one wrap per chunk at most)
constexpr uint32_t BW_FRAMES = 1024;             // frames per Ring::write the fast path takes (64 lanes x 16)
#ifndef ODDIO_BW_NT_STORE
#define ODDIO_BW_NT_STORE 1      // the ring stores are streaming (nt) stores: a callback writes 1 GB of ring that nothing reads before the next one;
                                 // left to the L2's write-back they disturb the walk and the ring reads too (round 6: write 0.60 -> 0.52 ms with 9 waves per CU,
                                 // walk 0.047 -> 0.039, reads 0.241 -> 0.225; 0: plain stores)
#endif
#ifndef ODDIO_BW_GROUP_LOG2
#define ODDIO_BW_GROUP_LOG2 4
#endif
constexpr int BW_GROUP_LOG2 = ODDIO_BW_GROUP_LOG2;
constexpr int BW_GROUP = 1 << BW_GROUP_LOG2;    // sources per scan
constexpr int BW_SLOTS = 3;                      // running sums per source: leaf cursor, two ramping Gains
constexpr int BW_STREAMS = BW_SLOTS * BW_GROUP;
constexpr int BW_CK_STRIDE = BW_STREAMS + 1;     // odd: lane b's reads (row b) and the scanners' writes (one row) are both conflict-free
constexpr int BW_WIN_CAP = 1216;                 // leaf samples staged per source (ds <= ~1.18 for 1024 frames)
constexpr int BW_WIN_BYTES = BW_WIN_CAP * 4;
constexpr int BW_WIN_PIECES = (BW_WIN_BYTES + 1023) / 1024;
constexpr int BW_LDS_WIN0 = 0, BW_LDS_WIN1 = BW_WIN_BYTES, BW_LDS_CK = 2 * BW_WIN_BYTES;
// Checkpoints of the running sums every BW_CK_FRAMES frames (round 6: 32, was 16): a lane whose 16 frames start in the middle of a
// checkpoint interval replays the 16 steps before them with the scan's own operations -- 16 issue slots per stream -- and the block
// shrinks from 12.5 to 6.3 KB: 10 resident waves per CU instead of 7, in a kernel whose time follows its occupancy
// (ODDIO_HIP_WRITE_WAVES_PER_CU 4 / 5 / 6 / 7: 0.80 / 0.68 / 0.60 / 0.57 ms, profiles/r06_ab_buffered_write.txt).
#ifndef ODDIO_BW_CK_SHIFT
#define ODDIO_BW_CK_SHIFT 1      // 0: a checkpoint per 16-frame block (one per lane); 1: per 32 frames
#endif
constexpr int BW_CK_SHIFT = ODDIO_BW_CK_SHIFT;
constexpr int BW_CK_ROWS = 64 >> BW_CK_SHIFT;
static_assert(BW_CK_SHIFT == 0 || BW_CK_SHIFT == 1, "a lane replays at most one 16-frame block");
constexpr int BW_LDS_TOTAL = BW_LDS_CK + BW_CK_ROWS * BW_CK_STRIDE * 4;
static_assert(BW_WIN_BYTES % 16 == 0 && BW_WIN_PIECES == 5, "leaf window buffer: five 1-KiB DMA pieces, the last one partial");

enum : uint32_t { BW_SKIP = 0, BW_FAST = 1, BW_SLOW = 2 };
// BWF_SYNTH (round 5): the leaf has no clip -- a Constant (the record's nvec field = 0; constant.rs:16-18) or a Sine (nvec = 1;
// sine.rs:34-40: frac0[s] = the phase the s-th inner.sample call starts from, ds = the interval at the leaf) -- and is computed
// where a FramesSignal's window would be read: the same chain, ring stores and sums as a clip source (before, such sources took the
// general kernel, one wavefront and a slab each: 12x the time at scale).
// BWF_CYCLE (round 5): the leaf is a Cycle (cycle.rs:26-53) over a mono clip at least as long as the callback's window: its cursor is
// the pair (base, offset) -- the scan replays cycle.rs:37-42's rewrite at the clip's end (and :28-29's split at the second
// inner.sample call of a wrapping Ring::write), `base` travels in the checkpoint rows of ramp slot B (such a chain has at most one
// ramping Gain) -- and its window, linear modulo the clip length, is staged by plain loads; the wave leaves the cursor in BufDyn.
enum : uint32_t { BWF_LEAF_FAST = 8u, BWF_SEG2 = 16u, BWF_SPECIAL = 32u, BWF_PAD = 64u, BWF_SYNTH = 128u, BWF_CYCLE = 1u << 28 };

// What the walk leaves for buffered_write per slot.
//   desc     buffer descriptor words 0-2 of the leaf window (window_desc: clipped to the clip, zeros outside it)
//   info     path (bits 0-2) | BWF_* | nvec << 8 (12 bits) | negvec << 20 (8 bits)
//   frac0[s] start offset of Ring::write's s-th inner.sample call (frames.rs:181/189); wrel[s] its base index relative to
//            the window start; cnt1 frames belong to the first call, cnt to both
//   ops      n_wrap | per filter w, 2 bits at 4 + 2w: 0 multiply by c[w] (FixedGain, a settled Gain; 1.0 for Speed), 1 / 2 the
//            ramping Gain in slot A / B | the filter index of ramp A at 12, of ramp B at 14
struct alignas(16) WriteRec {
    uint32_t desc[3];
    uint32_t info;
    float* ring;
    uint32_t ring_len;
    uint32_t start_idx;       // Ring::write's first index (ring.rs:20)
    float frac0[2];
    float ds;
    uint32_t wrel;            // wrel[0] | wrel[1] << 16
    uint32_t cnt;             // cnt1 | cnt << 16
    uint32_t ops;
    float c[MAX_WRAP];
    float rprev[2], rnext[2], rp0[2], rstep[2];
    float leaf_a;             // a synthesised leaf (BWF_SYNTH): Sine's freq in rad/s (sine.rs:21) or the Constant's value
    uint32_t cyc_base;        // a Cycle leaf (BWF_CYCLE): `base` of cycle.rs:28 at the callback's start (frac0[0] = its `offset`, :29)
    uint32_t pad[4];
};
static_assert(sizeof(WriteRec) == 128, "WriteRec layout");
// (buffered_write reads the record as 32-bit words through v_readlane: word indices below)
static_assert(offsetof(WriteRec, info) == 12 && offsetof(WriteRec, ring) == 16 && offsetof(WriteRec, ring_len) == 24 && offsetof(WriteRec, start_idx) == 28 &&
              offsetof(WriteRec, frac0) == 32 && offsetof(WriteRec, ds) == 40 && offsetof(WriteRec, wrel) == 44 && offsetof(WriteRec, cnt) == 48 &&
              offsetof(WriteRec, ops) == 52 && offsetof(WriteRec, c) == 56 && offsetof(WriteRec, rprev) == 72 && offsetof(WriteRec, rnext) == 80 &&
              offsetof(WriteRec, rp0) == 88 && offsetof(WriteRec, rstep) == 96 && offsetof(WriteRec, leaf_a) == 104 && offsetof(WriteRec, cyc_base) == 108, "WriteRec word indices");

// the per-ear scalars of spatial.rs:409-423 for a source the general kernel renders after buffered_walk
struct alignas(16) BufEar { float prev_offset, dt, g0, dg; };
struct alignas(16) BufEarPair { BufEar e[2]; };

// One (source, 512-frame tile) of Ring::sample reads as a tile record of spatial_mix<.., RING>.
//   per (ear, chunk) stream: t = prev_offset + i * dt at the chunk's first frame (spatial.rs:424), cursor =
//   rem_euclid(write + t * rate, len) (ring.rs:57), step = dt * rate (:58).
// The window is the stretch of the ring (indices taken modulo len, linear in memory thanks to the mirror) that the
// tile's four streams can touch.  Returns false when the tile cannot be staged (step out of range, window too large).
__device__ __forceinline__ bool ring_tile_rec(TileRec& r, const SceneParams& P, float* ring, uint32_t len, uint32_t rate, float write,
                                              const BufEar (&ear)[2], uint32_t tile) {
    r = TileRec{};
    const float lenf = (float)len, ratef = (float)rate;
    int start[2][2], last[2][2];
    bool live[2][2];
    bool any = false, wrap = false;
    uint32_t fl = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float ds = ear[e].dt * ratef;                                              // ring.rs:58
        if (!(ds > 0.0f) || !(ds < 2.0f)) return false;
        if (fabsf(ds - 1.0f) < PAD_EPS) fl |= SFLAG_PAD;
#pragma unroll
        for (int c = 0; c < TILE_CHUNKS; ++c) {
            const uint32_t done = (tile * TILE_CHUNKS + (uint32_t)c) * 256u;
            live[e][c] = done < P.n_frames;
            start[e][c] = 0; last[e][c] = 0;
            if (!live[e][c]) continue;
            const float t = ear[e].prev_offset + (float)done * ear[e].dt;               // spatial.rs:424
            const float off = f32_rem_euclid(write + t * ratef, lenf);                   // ring.rs:57
            if (!(off >= 0.0f) || !(off <= lenf)) return false;
            r.frac0[e][c] = off;
            const float xb = off + 255.0f * ds;                  // (a lane renders its 16 frames even when the callback ends inside them)
            const float xu = xb + xb * 1.0e-4f + 1.0e-2f;        // >= the sequentially rounded cursor (before or after a rewrite)
            start[e][c] = (int)off;
            last[e][c] = (int)xu + 1;                            // the pair's second sample
            if (last[e][c] >= (int)len) wrap = true;             // the cursor may reach len: ring.rs:66-74
            any = true;
        }
        r.ear[e].ds = ds; r.ear[e].g0 = ear[e].g0; r.ear[e].dg = ear[e].dg;
    }
    if (!any) return true;                                       // PATH_SKIP: no frames in this tile
    // the streams lie within a few hundred samples of each other on the circle: positions relative to the first live one
    int ref = 0; bool have = false;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int c = 0; c < TILE_CHUNKS; ++c) if (live[e][c] && !have) { ref = start[e][c]; have = true; }
    const int ilen = (int)len, half = ilen >> 1;
    int lo = 0x7fffffff;
    int rel[2][2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int c = 0; c < TILE_CHUNKS; ++c) {
            rel[e][c] = 0;
            if (!live[e][c]) continue;
            int d = start[e][c] - ref;                           // in (-len, len]
            if (d > half) d -= ilen; else if (d < -half) d += ilen;
            rel[e][c] = d;
            lo = min(lo, d);
        }
    int ws = ref + lo;                                           // the window's first index, on the circle
    if (ws < 0) ws += ilen; else if (ws >= ilen) ws -= ilen;
    const int al = ws & 3;
    ws -= al;                                                    // 16-byte aligned (rings are)
    int count = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        uint32_t w01 = 0;
#pragma unroll
        for (int c = 0; c < TILE_CHUNKS; ++c) {
            if (!live[e][c]) continue;
            const int p = rel[e][c] - lo + al;                   // window position of the stream's first index
            count = max(count, p + (last[e][c] - start[e][c]) + 1);
            w01 |= (uint32_t)p << (16 * c);
        }
        r.ear[e].wrel = w01;
    }
    const int vec_samples = ((count + 3) >> 2) << 2;
    if (vec_samples > WIN_CAP) return false;
    if (wrap) { fl &= ~(uint32_t)SFLAG_PAD; fl |= SFLAG_WRAP; }
    if ((fl & SFLAG_PAD) && vec_samples + (vec_samples >> 4) + 1 > WIN_CAP) fl &= ~(uint32_t)SFLAG_PAD;
    const int nvec = (count + 3) >> 2;
    const uint64_t base = (uint64_t)(ring + ws);
    r.desc[0] = (uint32_t)(base & 0xffffffffu); r.desc[1] = (uint32_t)((base >> 32) & 0xffffu); r.desc[2] = len;
    r.info = (uint32_t)PATH_LDS | (fl << 3) | ((uint32_t)nvec << 8);
    (void)ODDIO_BOUNDS_CHECK(P.bounds_err, nvec >= 1 && nvec * 4 <= WIN_CAP && ws >= 0 && (uint32_t)(ws + nvec * 4) <= len + RING_MIRROR, BOUNDS_RECORD, nvec, tile);
    return true;
}

// The record of one Ring::write-shaped render through the filter chain (what buffered_write consumes): `cnt1` (+ `cnt2` after the
// ring's end when `seg2`) frames from `start_idx` on, each segment one inner.sample call at `interval` seconds per frame at the top
// of the chain.  Works on copies of the Smoothed state (`sm_*`) and of the leaf clock (`t_new`): the caller commits them if the
// result is true, and leaves the source to the general kernel otherwise.  Shared by the scene's buffered set (buffered_walk) and the
// Mixer's chain sources (mixer_chain_walk, mixer_kernels.h), whose "ring" is the source's slab.
__device__ __forceinline__ bool chain_write_rec(WriteRec& wr, uint32_t* bounds_err, const BufStatic& s, const BufDyn& d, const SrcDyn& c, float interval,
                                                uint32_t cnt1, uint32_t cnt2, bool seg2, size_t start_idx, float* ring, uint32_t rlen, uint32_t src_index,
                                                bool fast, double& t_new, float (&sm_prev)[MAX_WRAP], float (&sm_next)[MAX_WRAP], float (&sm_prog)[MAX_WRAP],
                                                float& phase_new) {
    // the chain: interval per level (speed.rs:32-35), Smoothed::set (gain.rs:106-109) -- on copies; committed only if fast
    uint32_t ops = s.n_wrap & 7u;
    t_new = c.t;
    phase_new = c.phase;
    if (fast) {
        float level_interval[MAX_WRAP];
        float cur = interval;
#pragma unroll
        for (int w = MAX_WRAP - 1; w >= 0; --w) {
            level_interval[w] = cur;
            if ((uint32_t)w < s.n_wrap && s.wrap_kind[w] == WRAP_SPEED) cur = cur * d.shared[w];
        }
        int n_ramp = 0;
#pragma unroll
        for (uint32_t w = 0; w < MAX_WRAP; ++w) {
            sm_prev[w] = d.sm_prev[w]; sm_next[w] = d.sm_next[w]; sm_prog[w] = d.sm_progress[w];
            wr.c[w] = 1.0f;
            if (w >= s.n_wrap) continue;
            if (s.wrap_kind[w] == WRAP_FIXED_GAIN) wr.c[w] = s.wrap_param[w];                  // gain.rs:32-37
            else if (s.wrap_kind[w] == WRAP_GAIN) {
                const float shared = d.shared[w];
                if (sm_next[w] != shared) {
                    sm_prev[w] = sm_prev[w] + sm_prog[w] * (sm_next[w] - sm_prev[w]);
                    sm_next[w] = shared;
                    sm_prog[w] = 0.0f;
                }
                if (sm_prog[w] != 1.0f) {                                                       // a running ramp (gain.rs:114-120)
                    const float step = level_interval[w] / 0.1f;                                 // SMOOTHING_PERIOD, gain.rs:163
                    if (n_ramp == 0) {   // (no array indexed by n_ramp: that would put the record in scratch memory)
                        wr.rprev[0] = sm_prev[w]; wr.rnext[0] = sm_next[w]; wr.rp0[0] = sm_prog[w]; wr.rstep[0] = step;
                        ops |= 1u << (4 + 2 * w);
                        ops |= w << 12;
                    } else if (n_ramp == 1) {
                        wr.rprev[1] = sm_prev[w]; wr.rnext[1] = sm_next[w]; wr.rp0[1] = sm_prog[w]; wr.rstep[1] = step;
                        ops |= 2u << (4 + 2 * w);
                        ops |= w << 14;
                    }
                    n_ramp++;
                } else {
                    wr.c[w] = sm_prev[w] + sm_prog[w] * (sm_next[w] - sm_prev[w]);               // gain.rs:110-113 (x * 1.0 == x: no need to skip)
                }
            }
        }
        if (n_ramp > 2) fast = false;
        if (s.kind == KIND_CONSTANT || s.kind == KIND_SINE) {
            // a leaf without a clip: nothing to stage, no cursor to scan
            const bool sine = s.kind == KIND_SINE;
            uint32_t fl = BWF_SYNTH | BWF_LEAF_FAST;
            if (seg2) fl |= BWF_SEG2;
            if (seg2 || start_idx < RING_MIRROR || cnt1 + cnt2 != BW_FRAMES) fl |= BWF_SPECIAL;
            if (sine) {
                // sin_small's argument reduction is exact for |x| < ~12 600 (kernels.h): the phase is in (-TAU, TAU), t * freq below:
                if (!(fabsf((cur * (float)BW_FRAMES) * s.freq_or_value) < 12000.0f)) fast = false;
                wr.frac0[0] = c.phase;
                const float ph1 = fmodf(c.phase + (cur * (float)cnt1) * s.freq_or_value, ODDIO_TAU);     // sine.rs:39 after the first call
                wr.frac0[1] = ph1;
                phase_new = seg2 ? fmodf(ph1 + (cur * (float)cnt2) * s.freq_or_value, ODDIO_TAU) : ph1;
            }
            if (fast) {
                wr.desc[0] = 0u; wr.desc[1] = 0u; wr.desc[2] = 0u;              // (an empty descriptor: the window DMA moves nothing)
                wr.info = BW_FAST | fl | ((sine ? 1u : 0u) << 8);
                wr.ring = ring; wr.ring_len = rlen; wr.start_idx = (uint32_t)start_idx;
                wr.ds = cur;
                wr.leaf_a = s.freq_or_value;
                wr.wrel = 0u;
                wr.cnt = cnt1 | ((cnt1 + cnt2) << 16);
                wr.ops = ops;
            }
            return fast;
        }
        if (s.kind == KIND_CYCLE) {
            const float dsc = cur * (float)s.clip_rate;                                           // cycle.rs:27
            const uint32_t cnt = cnt1 + cnt2;
            if (n_ramp > 1 || !(dsc > 0.0f) || !(dsc < 8.0f) || s.clip_len < 2u || !(c.t >= 0.0) || !(c.t < 1073741824.0)) fast = false;
            const uint32_t w_count = fast ? (uint32_t)((float)cnt * dsc * 1.0001f) + 8u : 0u;      // every index the callback can read, as one stretch
            if (w_count > (uint32_t)BW_WIN_CAP || w_count > s.clip_len) fast = false;              // (shorter loops lap inside a callback: general kernel)
            if (fast) {
                const uint32_t base0 = (uint32_t)c.t;                                              // `cursor as usize` (:28)
                uint32_t fl = BWF_CYCLE;
                if (seg2) fl |= BWF_SEG2;
                if (seg2 || start_idx < RING_MIRROR || cnt != BW_FRAMES) fl |= BWF_SPECIAL;
                const uint64_t cp = (uint64_t)s.clip;
                wr.desc[0] = (uint32_t)(cp & 0xffffffffu); wr.desc[1] = (uint32_t)(cp >> 32) & 0xffffu; wr.desc[2] = 4u * s.clip_len;
                wr.info = BW_FAST | fl;
                wr.ring = ring; wr.ring_len = rlen; wr.start_idx = (uint32_t)start_idx;
                wr.frac0[0] = (float)(c.t - (double)base0);                                        // :29
                wr.ds = dsc;
                wr.cyc_base = base0;
                wr.wrel = 0u;
                wr.cnt = cnt1 | (cnt << 16);
                wr.ops = ops;
            }
            return fast;                               // (the cursor is advanced by buffered_write: it is the scan's result)
        }
        // leaf: frames.rs:176-181 per inner.sample call
        const float ds = cur * (float)s.clip_rate;
        if (!(ds > 0.0f) || !(ds < 64.0f)) fast = false;
        const bool leaf_fast = fabsf(ds - 1.0f) <= FLT_EPSILON;
        int lo = 0x7fffffff, hi = (int)0x80000000;
        int base_s[2] = {0, 0};
        const uint32_t cnts[2] = {cnt1, cnt2};
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            if (sg == 1 && !seg2) break;
            const double s0 = t_new * (double)s.clip_rate;
            const long long base = f64_as_isize(s0);
            const float frac0 = (float)(s0 - (double)base);
            if (!(fabs(s0) < 1.0e9)) fast = false;
            base_s[sg] = (int)base;
            wr.frac0[sg] = frac0;
            if (cnts[sg] > 0 && fast) {
                int i0, i1;
                if (leaf_fast) { i0 = (int)base; i1 = (int)base + (int)cnts[sg] - 1; }
                else {
                    const float xb = frac0 + (float)(cnts[sg] - 1u) * ds;
                    const float xu = xb + fabsf(xb) * 1.0e-4f + 1.0e-2f;
                    if (!(xu < 8.0e6f)) fast = false;
                    i0 = (int)base + (int)frac0;
                    i1 = (int)base + (int)xu;
                }
                lo = min(lo, i0); hi = max(hi, i1);
            }
            t_new = t_new + (double)cur * (double)cnts[sg];                                      // frames.rs:198
        }
        if (fast) {
            const int ws = lo & ~3;
            const int count = hi + 2 - ws;
            const int nvec = (count + 3) >> 2;
            uint32_t fl = 0;
            if (leaf_fast) fl |= BWF_LEAF_FAST;
            if (seg2) fl |= BWF_SEG2;
            if (seg2 || start_idx < RING_MIRROR || cnt1 + cnt2 != BW_FRAMES) fl |= BWF_SPECIAL;
            if (fabsf(ds - 1.0f) < PAD_EPS) {
                if (nvec * 4 + (nvec * 4 >> 4) + 1 <= BW_WIN_CAP) fl |= BWF_PAD;
                else if (leaf_fast) fast = false;         // the constant-fract loop exists for the padded layout only
            }
            if (count < 2 || nvec * 4 > BW_WIN_CAP) fast = false;
            if (fast) {
                const int4 dd = window_desc(s.clip, (int)((s.clip_len + 3u) & ~3u), ws, nvec);
                const int negvec = (dd.z > 0) ? ((-dd.w) >> 4) : 0;
                if (negvec > 255) fast = false;
                wr.desc[0] = (uint32_t)dd.x; wr.desc[1] = (uint32_t)dd.y; wr.desc[2] = (uint32_t)dd.z;
                wr.info = BW_FAST | fl | ((uint32_t)nvec << 8) | ((uint32_t)negvec << 20);
                wr.ring = ring; wr.ring_len = rlen; wr.start_idx = (uint32_t)start_idx;
                wr.ds = ds;
                (void)ODDIO_BOUNDS_CHECK(bounds_err, nvec >= 1 && nvec * 4 <= BW_WIN_CAP && base_s[0] - ws >= 0 && base_s[0] - ws < count &&
                                         (!seg2 || (base_s[1] - ws >= 0 && base_s[1] - ws < count)) && dd.z >= 0 && dd.z <= nvec * 16, BOUNDS_RECORD, nvec, src_index);
                wr.wrel = (uint32_t)(base_s[0] - ws) | ((uint32_t)((seg2 ? base_s[1] : base_s[0]) - ws) << 16);
                wr.cnt = cnt1 | ((cnt1 + cnt2) << 16);
                wr.ops = ops;
            }
        }
    }
    return fast;
}

// ---------------------------------------------------------------------------------------------------------------
// buffered_walk: one thread per slot of the buffered set
// ---------------------------------------------------------------------------------------------------------------
// d_len_b: the device-resident set length; len_snap: what this walk saw (read by the later kernels of the callback).
// slow_hdr: [par] = number of sources left to the general kernel this callback (par = callback parity; this walk zeroes the
// other one for the next callback: no memset on the stream), [2..] their slots.
__global__ __launch_bounds__(128) void buffered_walk(SceneParams P, const BufStatic* __restrict__ st, BufDyn* __restrict__ dyn, SrcPending* __restrict__ pend,
                                                     int check_pending, const uint32_t* __restrict__ d_len_b, uint32_t* __restrict__ len_snap,
                                                     WriteRec* __restrict__ wrecs, TileRec* __restrict__ trecs, uint32_t rec_stride,
                                                     BufEarPair* __restrict__ bear, uint32_t* __restrict__ slow_hdr, uint32_t par,
                                                     uint32_t* __restrict__ stopped_hdr, uint32_t stopped_cap) {
    __shared__ uint32_t stage[2][64 * 37];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t* lds = stage[threadIdx.x >> 6];
    const uint32_t len = d_len_b[0];
    if (i == 0) { *len_snap = len; slow_hdr[par ^ 1u] = 0u; }
    const uint32_t first = i - (uint32_t)lane;
    if (first >= len) return;
    const uint32_t n_valid = (len - first) < 64u ? (len - first) : 64u;
    BufStatic s = {};
    BufDyn d = {};
    wave_aos_load2(s, st, d, dyn, first, n_valid, lane, lds);
    WriteRec wr = {};
    TileRec tr[REC_TILES];
#pragma unroll
    for (int t = 0; t < REC_TILES; ++t) tr[t] = TileRec{};
    BufEarPair be = {};
    const uint32_t n = P.n_frames;
    const uint32_t n_tiles = (n + TILE_FRAMES - 1) / TILE_FRAMES;
    bool write_dyn = false;
    if (i < len && !(d.common.flags & DYN_STOPPED)) {
        SrcDyn& c = d.common;
        write_dyn = true;
        const float elapsed = P.elapsed;
        const float nf = (float)n;
        V3 tpos = {c.tgt_pos[0], c.tgt_pos[1], c.tgt_pos[2]};
        V3 tvel = {c.tgt_vel[0], c.tgt_vel[1], c.tgt_vel[2]};
        V3 ppos = {c.prev_pos[0], c.prev_pos[1], c.prev_pos[2]};
        if (check_pending) {   // spatial.rs:216-226
            const SrcPending pm = pend[i];
            if (pm.flags & PEND_FRESH) {
                V3 npos = {pm.pos[0], pm.pos[1], pm.pos[2]};
                V3 nvel = {pm.vel[0], pm.vel[1], pm.vel[2]};
                ppos = (pm.flags & PEND_DISCONTINUITY) ? npos : smoothed_position(ppos, c.state_dt, 0.0f, tpos, tvel);
                tpos = npos; tvel = nvel;
                c.state_dt = 0.0f;
                pend[i].flags = 0;
            }
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
        if (c.flags & DYN_STOPPED) {
            const uint32_t k = atomicAdd(&stopped_hdr[0], 1u);
            if (k < stopped_cap) stopped_hdr[1 + k] = c.id;
        } else {
            BufEar ear[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {   // spatial.rs:409-423
                float off0, g0, off1, g1;
                ear_state(p0, e, s.radius, off0, g0);
                ear_state(p1, e, s.radius, off1, g1);
                ear[e].prev_offset = fmaxf(off0 - elapsed, -s.max_delay);
                const float next_offset = fmaxf(off1, -s.max_delay);
                ear[e].dt = (next_offset - ear[e].prev_offset) / nf;
                ear[e].dg = (g1 - g0) / nf;
                ear[e].g0 = g0;
            }
            be.e[0] = ear[0]; be.e[1] = ear[1];
            // ---- can buffered_write + spatial_mix<RING> render this source this callback? ----
            bool fast = (s.flags & BUF_FAST_OK) != 0u && n >= 1u && n <= (uint32_t)(REC_TILES * TILE_FRAMES);
            const uint32_t rlen = s.ring_len;
            const float lenf = (float)rlen;
            // Ring::write (ring.rs:18-41)
            const float end = fmodf(d.ring_write + elapsed * (float)s.rate, lenf);
            const size_t start_idx = f32_as_usize(ceilf(d.ring_write));
            const size_t end_idx = f32_as_usize(ceilf(end));
            uint32_t cnt1 = 0, cnt2 = 0;
            bool seg2 = false;
            if (fast) {
                if (end_idx > start_idx) { cnt1 = (uint32_t)(end_idx - start_idx); }
                else if (start_idx <= rlen) { seg2 = true; cnt1 = (uint32_t)(rlen - start_idx); cnt2 = (uint32_t)end_idx; }
                else fast = false;
                if (cnt1 + cnt2 < 1u || cnt1 + cnt2 > BW_FRAMES || cnt1 > BW_FRAMES) fast = false;
            }
            float sm_prev[MAX_WRAP], sm_next[MAX_WRAP], sm_prog[MAX_WRAP];
            double t_new = c.t;
            float phase_new = c.phase;
            fast = chain_write_rec(wr, P.bounds_err, s, d, c, 1.0f / (float)s.rate, cnt1, cnt2, seg2, start_idx, s.ring, rlen, i, fast, t_new, sm_prev, sm_next, sm_prog, phase_new);
            // the ring reads of this callback start from the cursor Ring::write leaves (ring.rs:40)
            if (fast) {
#pragma unroll
                for (int t = 0; t < REC_TILES; ++t)
                    if ((uint32_t)t < n_tiles && fast) fast = ring_tile_rec(tr[t], P, s.ring, rlen, s.rate, end, ear, (uint32_t)t);
            }
            if (fast) {
                c.t = t_new;
                c.phase = phase_new;
                d.ring_write = end;
#pragma unroll
                for (int w = 0; w < MAX_WRAP; ++w) { d.sm_prev[w] = sm_prev[w]; d.sm_next[w] = sm_next[w]; d.sm_progress[w] = sm_prog[w]; }
            } else {
                wr = WriteRec{};
                wr.info = BW_SLOW;
                const uint32_t k = atomicAdd(&slow_hdr[par], 1u);
                slow_hdr[2 + k] = i;
#pragma unroll
                for (int t = 0; t < REC_TILES; ++t) { tr[t] = TileRec{}; if ((uint32_t)t < n_tiles) tr[t].info = PATH_ROW; }
            }
        }
    }
    wave_aos_store(wr, wrecs, first, n_valid, lane, lds);
#pragma unroll
    for (int t = 0; t < REC_TILES; ++t)
        if ((uint32_t)t < n_tiles) wave_aos_store(tr[t], trecs + (size_t)t * rec_stride, first, n_valid, lane, lds);
    if (__any((wr.info & 7u) == BW_SLOW)) wave_aos_store(be, bear, first, n_valid, lane, lds);
    if (__any(write_dyn)) wave_aos_store(d, dyn, first, n_valid, lane, lds);
}

// ---------------------------------------------------------------------------------------------------------------
// buffered_write: Ring::write (ring.rs:18-41) through FixedGain / Gain / Speed over a FramesSignal
// ---------------------------------------------------------------------------------------------------------------
// HBM -> LDS: `nvec` 16-byte vectors of the leaf window, 1 KiB per instruction (see window_dma in kernels.h).
__device__ __forceinline__ void leaf_window_dma(uint32_t lds_dst, uint32_t d0, uint32_t d1, uint32_t d2, int nvec, int negvec, int lane16) {
    u32x4 rsrc;
    rsrc.x = d0; rsrc.y = d1; rsrc.z = d2;
    rsrc.w = 0x00020000u;
    const int voff = -16 * negvec + lane16;      // negative offsets wrap to huge unsigned values: out of range -> 0
    uint32_t keep;
    // pieces 0-3 from every lane (lanes past the window write zeros inside the buffer), piece 4 from the lanes that stay inside it
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
                 "buffer_load_dwordx4 %1, %2, 0 offen nt lds\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:1024 nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
    if (nvec > 128)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
                     "buffer_load_dwordx4 %1, %2, 0 offen offset:2048 nt lds\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:3072 nt lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
    if (nvec > 256) {
        if (lane16 < BW_WIN_BYTES - 4096)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
                         "buffer_load_dwordx4 %1, %2, 0 offen nt lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff + 4096), "s"(rsrc), "s"(lds_dst + 4096u) : "memory");
    }
}

// plain -> padded layout in place (slot(s) = s + s/16; the pad slot repeats the following sample): see window_repack_padded
__device__ __forceinline__ void leaf_repack_padded(unsigned char* win_bytes, int nvec, int lane) {
    u32x4 v[BW_WIN_PIECES];
#pragma unroll
    for (int k = 0; k < BW_WIN_PIECES; ++k)
        v[k] = (lane + 64 * k < nvec) ? *reinterpret_cast<const u32x4*>(win_bytes + 16 * lane + 1024 * k) : u32x4{0u, 0u, 0u, 0u};
    wave_sync();
    unsigned int* win = reinterpret_cast<unsigned int*>(win_bytes);
#pragma unroll
    for (int k = 0; k < BW_WIN_PIECES; ++k) {
        const int q = lane + 64 * k;
        if (q < nvec) {
            const int li = 4 * q;
            const int pos = li + (li >> 4);
            win[pos + 0] = v[k].x; win[pos + 1] = v[k].y; win[pos + 2] = v[k].z; win[pos + 3] = v[k].w;
            if ((li & 15) == 0 && li > 0) win[pos - 1] = v[k].x;
        }
    }
    wave_sync();
}


// grid = waves (one 64-thread workgroup each); wave w renders groups [w * groups_per_wave, ...) of 16 slots.
// (bounds_err: the bounds-checked build's violation record -- device_types.h; null otherwise)
// ACC (the Mixer's chain sources in FAST mode, mixer_kernels.h): nothing is stored per source -- the wave adds what it renders to 16
// running sums per lane and leaves ONE row, frames duplicated into both channels (MonoToStereo, signal.rs:73-80), at
// acc_rows[blockIdx.x * acc_stride ..]: the sum is a tree over waves anyway, and a slab per source that a sum kernel re-reads is
// twice the chain's traffic.
template <bool ACC = false>
__global__ __launch_bounds__(64) void buffered_write(const WriteRec* __restrict__ wrecs, const uint32_t* __restrict__ len_snap,
                                                     BufDyn* __restrict__ dyn, uint32_t groups_per_wave, uint32_t* __restrict__ bounds_err,
                                                     float* __restrict__ acc_rows, uint32_t acc_stride, uint32_t acc_frames) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[BW_LDS_TOTAL];
    const int lane = threadIdx.x;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
    const int lane16 = 16 * lane;
    const uint32_t n_sources = *len_snap;
    const uint32_t n_groups = (n_sources + BW_GROUP - 1) / BW_GROUP;
    // `groups_per_wave`: groups of 16 sources per wave; with 2^k in its top byte instead, 2^k waves share ONE group and each renders
    // 16 >> k of its sources (small sets: a wave renders its sources one after the other, ~3 us each)
    const uint32_t split_log2 = groups_per_wave >> 24;
    uint32_t g_lo = blockIdx.x * (groups_per_wave & 0xffffffu);
    uint32_t g_hi = g_lo + (groups_per_wave & 0xffffffu);
    unsigned long long part_mask = 0xffffull;
    if (split_log2) {
        g_lo = blockIdx.x >> split_log2;
        g_hi = g_lo + 1u;
        const uint32_t per = (uint32_t)BW_GROUP >> split_log2;
        part_mask = ((1ull << per) - 1ull) << ((blockIdx.x & ((1u << split_log2) - 1u)) * per);
    }
    if (g_hi > n_groups) g_hi = n_groups;
    const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)smem);
    float* ck = reinterpret_cast<float*>(smem + BW_LDS_CK);
    const int slot = lane >> BW_GROUP_LOG2, jA = lane & (BW_GROUP - 1);
    int buf = 0;
    for (uint32_t g = g_lo; g < g_hi; ++g) {
        // ------------------------------ phase A: the running sums of the group ------------------------------
        const uint32_t srcA = g * BW_GROUP + (uint32_t)jA;
        uint32_t infoA = 0;
        float x = 0.0f, inc = 0.0f, cap = __builtin_inff(), x2 = 0.0f;
        uint32_t n1 = 0xffffu;
        bool is_ramp = false, is_cursor = false, is_cyc = false;
        uint32_t cb = 0u, clen = 0u;                   // a Cycle's `base` and clip length (slot-0 lane of the source)
        if (srcA < n_sources && slot < BW_SLOTS) {
            const WriteRec* r = wrecs + srcA;
            infoA = r->info;
            if ((infoA & 7u) == BW_FAST) {
                if (slot == 0) {
                    if (!(infoA & BWF_LEAF_FAST)) {
                        is_cursor = true;
                        x = r->frac0[0]; inc = r->ds;
                        if (infoA & BWF_SEG2) { n1 = r->cnt & 0xffffu; x2 = r->frac0[1]; }
                        if (infoA & BWF_CYCLE) { is_cyc = true; cb = r->cyc_base; clen = r->desc[2] >> 2; }
                    }
                } else {
                    const uint32_t ops = r->ops;
                    bool present = false;
#pragma unroll
                    for (int w = 0; w < MAX_WRAP; ++w) present = present || (((ops >> (4 + 2 * w)) & 3u) == (uint32_t)slot);
                    if (present) { is_ramp = true; x = r->rp0[slot - 1]; inc = r->rstep[slot - 1]; cap = 1.0f; }
                }
            }
        }
        const unsigned long long fast_mask = __ballot(slot == 0 && (infoA & 7u) == BW_FAST) & part_mask;   // bit j: source j of the group is rendered here
        if (fast_mask == 0ull) continue;
        // lanes 0-15 keep the record of source `lane` in registers for the whole group: phase B takes a source's (wave-uniform)
        // words from there with v_readlane -- fetched from memory at the point of use, every source paid a scalar-load latency
        uint32_t recw[28];
        {
            const uint4* rp = reinterpret_cast<const uint4*>(wrecs + g * BW_GROUP + (uint32_t)(lane & (BW_GROUP - 1)));
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (lane < BW_GROUP && g * BW_GROUP + (uint32_t)lane < n_sources) v = rp[q];
                recw[4 * q] = v.x; recw[4 * q + 1] = v.y; recw[4 * q + 2] = v.z; recw[4 * q + 3] = v.w;
            }
        }
#define ODDIO_RW(K, J) ((uint32_t)__builtin_amdgcn_readlane((int)recw[(K)], (J)))
#define ODDIO_RF(K, J) __int_as_float(__builtin_amdgcn_readlane((int)recw[(K)], (J)))
        // the first window is on its way while the sums are scanned
        int cur = __builtin_ctzll(fast_mask);
#define ODDIO_BW_ISSUE(J, BUF)                                                                                            \
    {                                                                                                                     \
        const uint32_t i_ = ODDIO_RW(3, (J));                                                                             \
        if (!(i_ & BWF_CYCLE))     /* (a Cycle's window is staged by plain loads at its turn) */                           \
        leaf_window_dma(lds_base + (uint32_t)((BUF) ? BW_LDS_WIN1 : BW_LDS_WIN0), ODDIO_RW(0, (J)), ODDIO_RW(1, (J)), ODDIO_RW(2, (J)),  \
                        (int)((i_ >> 8) & 0xfffu), (int)((i_ >> 20) & 0xffu), lane16);                                    \
    }
        ODDIO_BW_ISSUE(cur, buf)
        const bool any_scan = __any(is_cursor || is_ramp);
        if (any_scan) {
            // blocks of 16 frames in which some cursor restarts (the second inner.sample call of a Ring::write that wraps)
            const unsigned long long restart_lanes = __ballot(n1 < 0xffffu);
            unsigned long long restart_blocks = 0ull;
            for (unsigned long long m = restart_lanes; m; m &= m - 1ull) {
                const uint32_t n1j = (uint32_t)__builtin_amdgcn_readlane((int)n1, __builtin_ctzll(m));
                if (n1j < BW_FRAMES) restart_blocks |= 1ull << (n1j >> 4);
            }
            const bool any_ramp = __any(is_ramp);
            const bool any_cyc = __any(is_cyc);
            const bool own_row = lane < BW_STREAMS && !(slot == 2 && (infoA & BWF_CYCLE));   // (slot B's rows of a Cycle source hold its `base`)
            float* row = ck + lane;
#pragma unroll 1
            for (int b = 0; b < 64; ++b) {
                if ((b & ((1 << BW_CK_SHIFT) - 1)) == 0) {
                    if (own_row) row[(b >> BW_CK_SHIFT) * BW_CK_STRIDE] = x;
                    if (is_cyc) row[(b >> BW_CK_SHIFT) * BW_CK_STRIDE + 2 * BW_GROUP] = __uint_as_float(cb);
                }
                // a block in which a Cycle's cursor may reach its clip's end, or restarts: cycle.rs:37-42 / :28-29 step by step
                if (any_cyc && __any(is_cyc && (cb + f32_as_index(x + 17.0f * inc) + 2u >= clen || ((uint32_t)b == (n1 >> 4) && n1 < BW_FRAMES)))) {
#pragma unroll 1
                    for (int k = 0; k < 16; ++k) {
                        if ((uint32_t)(16 * b + k) == n1) {
                            if (is_cyc) { const double cur_ = (double)cb + (double)x; cb = (uint32_t)cur_; x = (float)(cur_ - (double)cb); }   // the second call's :28-29
                            else x = x2;
                        }
                        if (is_cyc) {
                            const uint32_t tr = f32_as_index(x);
                            if (cb + tr >= clen) { const float fr = x - (float)tr; x = (float)((cb + tr) % clen) + fr; cb = 0u; }            // :39-41
                        }
                        x = fminf(x + inc, cap);
                    }
                } else
                if ((restart_blocks >> b) & 1ull) {
#pragma unroll 1
                    for (int k = 0; k < 16; ++k) {
                        if ((uint32_t)(16 * b + k) == n1) x = x2;                  // frames.rs:176-181 of the second call
                        x = fminf(x + inc, cap);
                    }
                } else if (any_ramp) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) x = fminf(x + inc, cap);          // smooth.rs:47-49 / frames.rs:194
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k) x = x + inc;
                }
            }
            wave_sync();
        }
        // ------------------------------ phase B: one source at a time ------------------------------
        unsigned long long todo = fast_mask;
        int light = 0;            // VMEM instructions issued after the window that is waited for next, when their number is known (4-6); else 0
        while (todo) {
            const int j = cur;
            todo &= todo - 1ull;
            const uint32_t src = g * BW_GROUP + (uint32_t)j;
            const uint32_t info = ODDIO_RW(3, j);
            unsigned char* win_bytes = smem + (buf ? BW_LDS_WIN1 : BW_LDS_WIN0);
            // this source's window has landed.  (vmcnt counts loads and stores in issue order: everything issued after the window
            // -- the ring stores of the source before, `light` of them -- may stay in flight; waiting for those stores too, a
            // full write latency per source, made the kernel 2x slower)
            // The counts assume that hipcc emits exactly one VMEM instruction per ring store / progress store of the source before
            // (fewer -- merged or skipped stores -- and the wait could return before the window has landed).  -DODDIO_BW_STRICT_WAIT
            // (the bounds-checked build, libodd_hip_debug.so) waits for everything instead: tests/test_hip_bounds_build.py runs the
            // bit-exact fuzz seeds against both builds, so a drift of the product build's count shows up as a parity failure there.
#ifdef ODDIO_BW_STRICT_WAIT
            (void)light;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
            if (light == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (light == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (light == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            if (todo) { cur = __builtin_ctzll(todo); ODDIO_BW_ISSUE(cur, buf ^ 1) }
            float* const ring = reinterpret_cast<float*>(((uint64_t)ODDIO_RW(5, j) << 32) | (uint64_t)ODDIO_RW(4, j));
            const uint32_t rlen = ODDIO_RW(6, j);
            const uint32_t start_idx = ODDIO_RW(7, j);
            const float fr0 = ODDIO_RF(8, j), fr1 = ODDIO_RF(9, j), ds = ODDIO_RF(10, j);
            const uint32_t wrelw = ODDIO_RW(11, j);
            const int wrel0 = (int)(wrelw & 0xffffu), wrel1 = (int)(wrelw >> 16);
            const uint32_t cntw = ODDIO_RW(12, j);
            const uint32_t cnt1 = cntw & 0xffffu, cnt = cntw >> 16;
            const uint32_t ops = ODDIO_RW(13, j);
            const int nvec = (int)((info >> 8) & 0xfffu);
            const bool pad = (info & BWF_PAD) != 0u, leaf_fast = (info & BWF_LEAF_FAST) != 0u, seg2 = (info & BWF_SEG2) != 0u;
            if (pad) leaf_repack_padded(win_bytes, nvec, lane);
            const float* win = reinterpret_cast<const float*>(win_bytes);
            const uint32_t f0 = 16u * (uint32_t)lane;
            float out[16];
            const int win_slots = pad ? 4 * nvec + (4 * nvec >> 4) + 1 : 4 * nvec;    // (debug build: what a stored frame may read)
            (void)win_slots;
            const float* ckl = ck + (lane >> BW_CK_SHIFT) * BW_CK_STRIDE;
            const bool ck_replay = BW_CK_SHIFT && (lane & 1);     // this lane's frames start 16 steps behind its checkpoint
            // ---- the leaf: FramesSignal::sample (frames.rs:176-201) for this lane's 16 frames ----
            bool cyc_store = false;                       // (a Cycle: this lane holds the cursor after the callback's last frame)
            uint32_t cyc_cb = 0u; float cyc_off = 0.0f;
            if (info & BWF_CYCLE) {
                // Cycle::sample (cycle.rs:26-53).  The window -- every index the callback reads, from the first one on, linear modulo
                // the clip length (the pair's second sample, s[x + 1] or s[0] behind s[len - 1], is always the next slot) -- by plain loads:
                const uint32_t clen_ = ODDIO_RW(2, j) >> 2;
                const float* clip_ = reinterpret_cast<const float*>(((uint64_t)ODDIO_RW(1, j) << 32) | (uint64_t)ODDIO_RW(0, j));
                const uint32_t w_start = (ODDIO_RW(27, j) + f32_as_index(fr0)) % clen_;
                const uint32_t w_count = (uint32_t)((float)cnt * ds * 1.0001f) + 8u;             // (<= BW_WIN_CAP, <= clen_: chain_write_rec)
                float* wst = reinterpret_cast<float*>(win_bytes);
                for (uint32_t p_ = (uint32_t)lane; p_ < w_count; p_ += 64u) {
                    uint32_t idx = w_start + p_;
                    idx = idx >= clen_ ? idx - clen_ : idx;
                    wst[p_] = clip_[idx];
                }
                wave_sync();
                float off = ckl[j];
                uint32_t cbv = __float_as_uint(ckl[2 * BW_GROUP + j]);
                if (ck_replay) {      // the scan's steps of the 16 frames before this lane's (cycle.rs:28-29, :37-42, :50: no sample is read)
#pragma unroll 1
                    for (int k = 0; k < 16; ++k) {
                        const uint32_t f = f0 - 16u + (uint32_t)k;
                        if (seg2 && f == cnt1) { const double cur_ = (double)cbv + (double)off; cbv = (uint32_t)cur_; off = (float)(cur_ - (double)cbv); }
                        const uint32_t tr = f32_as_index(off);
                        if (cbv + tr >= clen_) { const float fr = off - (float)tr; off = (float)((cbv + tr) % clen_) + fr; cbv = 0u; }
                        off = off + ds;
                    }
                }
#pragma unroll 1
                for (int k = 0; k < 16; ++k) {
                    const uint32_t f = f0 + (uint32_t)k;
                    if (seg2 && f == cnt1) { const double cur_ = (double)cbv + (double)off; cbv = (uint32_t)cur_; off = (float)(cur_ - (double)cbv); }   // :28-29 of the second call
                    const uint32_t tr = f32_as_index(off);
                    const float fr = off - (float)tr;                                             // :31-32
                    uint32_t xi = cbv + tr;
                    if (xi >= clen_) { cbv = 0u; off = (float)(xi % clen_) + fr; xi = f32_as_index(off); }   // :39-41
                    uint32_t p_ = xi >= w_start ? xi - w_start : xi + clen_ - w_start;
                    (void)ODDIO_BOUNDS_CHECK(bounds_err, f >= cnt || p_ + 1u < w_count, BOUNDS_WINDOW_INDEX, p_, src);
                    p_ = p_ < w_count - 1u ? p_ : w_count - 2u;                                   // (frames past cnt, not stored)
                    const float a = wst[p_], bb = wst[p_ + 1u];
                    out[k] = a + fr * (bb - a);
                    off = off + ds;                                                               // :50
                    if (f + 1u == cnt) { cyc_store = true; cyc_cb = cbv; cyc_off = off; }
                }
                wave_sync();          // (the stores below reuse the window buffer)
            } else if (info & BWF_SYNTH) {
                const float la = ODDIO_RF(26, j);
                if (nvec == 0) {          // Constant::sample (constant.rs:16-18)
#pragma unroll
                    for (int k = 0; k < 16; ++k) out[k] = la;
                } else {                  // Sine::sample (sine.rs:34-38): sin((interval * i as f32) * freq + phase), i counted from the call's first frame
#pragma unroll 4
                    for (int k = 0; k < 16; ++k) {
                        const uint32_t f = f0 + (uint32_t)k;
                        const bool second = seg2 && f >= cnt1;
                        const float t = ds * (float)(second ? f - cnt1 : f);
                        out[k] = sin_small(t * la + (second ? fr1 : fr0));
                    }
                }
            } else if (!seg2) {
                if (leaf_fast) {          // :180-187 constant fract, consecutive pairs (padded layout)
                    const int w0 = wrel0 + (int)f0;
                    {   // (the frames this lane stores: the last of them reads the pair (w0 + kv - 1, w0 + kv))
                        const int kv = f0 >= cnt ? 0 : (int)((cnt - f0) < 16u ? (cnt - f0) : 16u);
                        (void)kv;
                        (void)ODDIO_BOUNDS_CHECK(bounds_err, kv == 0 || (w0 >= 0 && (w0 + kv) + ((w0 + kv) >> 4) < win_slots), BOUNDS_PAD_INDEX, w0, src);
                    }
                    float a = win[w0 + (w0 >> 4)];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int w1 = w0 + k + 1;
                        const float bb = win[w1 + (w1 >> 4)];
                        out[k] = a + fr0 * (bb - a);
                        a = bb;
                    }
                } else {
                    float xx = ckl[j];
                    if (ck_replay) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) xx = xx + ds;          // frames.rs:194, the scan's own adds
                    }
                    if (pad) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const int tr = (int)xx;
                            const float fr = xx - (float)tr;
                            int w = wrel0 + tr; w = w + (w >> 4);
                            (void)ODDIO_BOUNDS_CHECK(bounds_err, f0 + (uint32_t)k >= cnt || (w >= 0 && w + 1 < win_slots), BOUNDS_PAD_INDEX, w, src);
                            const float a = win[w], bb = win[w + 1];
                            out[k] = a + fr * (bb - a);
                            xx = xx + ds;
                        }
                    } else {
                        const float* wb = win + wrel0;
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const int tr = (int)xx;
                            const float fr = xx - (float)tr;
                            (void)ODDIO_BOUNDS_CHECK(bounds_err, f0 + (uint32_t)k >= cnt || (wrel0 + tr >= 0 && wrel0 + tr + 1 < win_slots), BOUNDS_WINDOW_INDEX, wrel0 + tr, src);
                            const float a = wb[tr], bb = wb[tr + 1];
                            out[k] = a + fr * (bb - a);
                            xx = xx + ds;
                        }
                    }
                }
            } else {
                // the Ring::write wraps: frames >= cnt1 belong to a second inner.sample call, which restarts the cursor
                // from the f64 clock (frames.rs:176-181).  A few percent of the sources of a callback.
                float xx = leaf_fast ? 0.0f : ckl[j];
                if (ck_replay && !leaf_fast) {
#pragma unroll 1
                    for (int k = 0; k < 16; ++k) {
                        if (f0 - 16u + (uint32_t)k == cnt1) xx = fr1;       // the second call's restart (frames.rs:176-181), as in the scan
                        xx = xx + ds;
                    }
                }
#pragma unroll 1
                for (int k = 0; k < 16; ++k) {
                    const uint32_t f = f0 + (uint32_t)k;
                    if (f == cnt1) xx = fr1;
                    const bool second = f >= cnt1;
                    const int wrel = second ? wrel1 : wrel0;
                    int tr; float fr;
                    if (leaf_fast) { tr = (int)(second ? f - cnt1 : f); fr = second ? fr1 : fr0; }
                    else { tr = (int)xx; fr = xx - (float)tr; xx = xx + ds; }
                    int w = wrel + tr;
                    if (pad) w = w + (w >> 4);
                    (void)ODDIO_BOUNDS_CHECK(bounds_err, f >= cnt || (w >= 0 && w + 1 < win_slots), BOUNDS_WINDOW_INDEX, w, src);
                    w = min(max(w, 0), BW_WIN_CAP - 2);          // frames past cnt (not stored) may run off the window
                    const float a = win[w], bb = win[w + 1];
                    out[k] = a + fr * (bb - a);
                }
            }
            // ---- the filters, innermost first (gain.rs:32-37, :110-121) ----
            const uint32_t n_wrap = ops & 7u;
            const uint32_t last_lane = (cnt - 1u) >> 4, last_k = (cnt - 1u) & 15u;
#pragma unroll
            for (uint32_t w = 0; w < MAX_WRAP; ++w) {
                if (w >= n_wrap) break;
                const uint32_t kind = (ops >> (4 + 2 * w)) & 3u;
                if (kind == 0u) {
                    const float cw = ODDIO_RF(14 + w, j);
                    if (cw != 1.0f) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) out[k] = out[k] * cw;
                    }
                } else {
                    const int ri = (int)kind - 1;
                    const float prev = ri ? ODDIO_RF(19, j) : ODDIO_RF(18, j);
                    const float next = ri ? ODDIO_RF(21, j) : ODDIO_RF(20, j);
                    const float step = ri ? ODDIO_RF(25, j) : ODDIO_RF(24, j);
                    float p = ckl[BW_GROUP * (int)kind + j];
                    if (ck_replay) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) p = fminf(p + step, 1.0f);   // smooth.rs:47-49, the scan's own steps
                    }
                    float pfin = p;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float gq = prev + p * (next - prev);        // Smoothed::get (smooth.rs:51-53)
                        out[k] = out[k] * gq;
                        p = fminf(p + step, 1.0f);                        // advance (smooth.rs:47-49)
                        if ((uint32_t)k == last_k) pfin = p;
                    }
                    // the Smoothed the next callback starts from: the progress after the last frame written
                    if ((uint32_t)lane == last_lane) dyn[src].sm_progress[(ops >> (12 + 2 * ri)) & 3u] = pfin;
                }
            }
            if (cyc_store) dyn[src].common.t = (double)cyc_cb + (double)cyc_off;                  // cycle.rs:52
            if (ACC) {
#pragma unroll
                for (int k = 0; k < 16; ++k) if (f0 + (uint32_t)k < cnt) acc[k] = acc[k] + out[k];
                light = 0;        // (only the progress stores above were issued after the next window: wait for everything)
                wave_sync();
                buf ^= 1;
                continue;
            }
            // ---- Ring::write's stores (ring.rs:33-38) ----
            // A lane holds 16 consecutive frames; stored from there, every instruction would touch 64 lines with 16 bytes each
            // (8 partial writes per 128-byte line: the kernel was bound by the L2's request rate, 104 M requests per launch).
            // The frames go through the window buffer just consumed instead (frame f at byte 4 f + 16 (f >> 6): the 16-byte
            // pad per 64 frames keeps the 64-byte lane rows off each other's banks) and leave as whole kilobytes.
            {
                float4* stg = reinterpret_cast<float4*>(win_bytes + 64 * lane + 16 * (lane >> 2));
#pragma unroll
                for (int q = 0; q < 4; ++q) stg[q] = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
            }
            wave_sync();
            if (!(info & BWF_SPECIAL)) {
                float* dst = ring + start_idx + 4u * (uint32_t)lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(win_bytes + 1024 * q + 16 * lane + 16 * (4 * q + (lane >> 4)));
                    f4u v4 = {v.x, v.y, v.z, v.w};
#if ODDIO_BW_NT_STORE
                    __builtin_nontemporal_store(v4, reinterpret_cast<f4u*>(dst + 256 * q));
#else
                    *reinterpret_cast<f4u*>(dst + 256 * q) = v4;
#endif
                }
            } else {
#pragma unroll 1
                for (int k = 0; k < 16; ++k) {
                    const uint32_t f = 64u * (uint32_t)k + (uint32_t)lane;
                    const float v = *reinterpret_cast<const float*>(win_bytes + 272 * k + 4 * lane);
                    if (f < cnt) {
                        const uint32_t idx = f < cnt1 ? start_idx + f : f - cnt1;
                        ring[idx] = v;
                        if (idx < RING_MIRROR) ring[rlen + idx] = v;     // the mirror behind the ring's end
                    }
                }
            }
            // stores of this source: 4 vector stores + one per ramping Gain (its progress), issued after the next window's DMA
            light = (info & (BWF_SPECIAL | BWF_CYCLE)) ? 0 : 4 + (int)(((ops >> 4) & 3u) != 0u) + (int)(((ops >> 6) & 3u) != 0u) + (int)(((ops >> 8) & 3u) != 0u) + (int)(((ops >> 10) & 3u) != 0u);
            wave_sync();      // every lane is done with this window buffer before it is refilled two sources on
            buf ^= 1;
        }
#undef ODDIO_BW_ISSUE
#undef ODDIO_RW
#undef ODDIO_RF
    }
    if (ACC) {
        float* dst = acc_rows + (size_t)blockIdx.x * acc_stride;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t f = 16u * (uint32_t)lane + (uint32_t)k;
            if (f < acc_frames) { dst[2u * f] = acc[k]; dst[2u * f + 1u] = acc[k]; }
        }
    }
}

// The general kernel's rendering of the sources buffered_walk left on the slow list (every shape the ABI accepts), after
// the walk: Ring::write through inner_sample_wave / fader_sample_wave, the per-ear Ring::sample reads into the source's
// slab row, and the mirror behind the ring's end.  grid = any; workgroup w takes list entries w, w + gridDim.x, ...
__global__ __launch_bounds__(64) void buffered_sources_slow(SceneParams P, const uint32_t* __restrict__ slow_hdr, uint32_t par, BufStatic* __restrict__ st,
                                                            BufDyn* __restrict__ dyn, const BufEarPair* __restrict__ bear,
                                                            float* __restrict__ contrib, FaderRec* __restrict__ faders, float* __restrict__ fader_scratch) {
    __shared__ float ck[8][64];
    const int lane = threadIdx.x;
    const uint32_t n_slow = slow_hdr[par];
    for (uint32_t q = blockIdx.x; q < n_slow; q += gridDim.x) {
        const uint32_t i = slow_hdr[2 + q];
        BufStatic s = st[i];
        BufDyn d = dyn[i];
        const BufEarPair be = bear[i];
        buffered_render_wave(P, s, d, be.e[0].prev_offset, be.e[0].dt, be.e[0].g0, be.e[0].dg, be.e[1].prev_offset, be.e[1].dt, be.e[1].g0, be.e[1].dg,
                             contrib + (size_t)i * 2 * P.n_frames, faders, fader_scratch, ck, lane);
        if (s.fader && lane == 0) st[i] = s;   // a completed fade swapped the signals
        if (lane == 0) dyn[i] = d;
        wg_sync();
    }
}

}  // namespace oddio_hip
