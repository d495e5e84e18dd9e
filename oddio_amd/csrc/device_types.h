// device_types.h -- source-table layout shared by the host scene and the gfx950 kernels.
//
// One SpatialScene (src/spatial.rs:160-189) is a structure of arrays in HBM, indexed by *slot*
// (the position in the reference's `Set`, src/set.rs:141-188).  Per slot:
//   SrcStatic  (32 B)  what was passed to play(): clip, rate, FixedGain, radius, kind
//   SrcDyn     (64 B)  what the audio thread mutates: cursor t / phase, Motion, State, finished_for
//   SrcPending (32 B)  the control thread's latest Motion (swap.rs "pending" slot)
//   EarParams  (2x32 B) written by the prepass each callback, read by the mix kernel
#pragma once
#include <stdint.h>

namespace oddio_hip {

// KIND_CYCLE: Mixer general path and the buffered set sample it one wave per source; in the Seek set it is
// rendered serially by `cycle_sources` into a contribution row that the mix kernel adds in set order
// (SrcStatic::freq_or_value holds the row index as raw bits, SrcDyn::t the cursor in samples).
// KIND_DOWNMIX: Downmix<FramesSignal<[f32;2]>> (downmix.rs) in the Seek set: interleaved stereo clip, each
// channel interpolated and the two summed; rendered by the per-lane global-memory path of spatial_mix.
// KIND_STREAM: Stream<T> (stream.rs): the SPSC ring lives in pinned, GPU-visible host memory (StreamHeader
// followed by the samples); general paths (Mixer, buffered set) only.
enum : uint32_t { KIND_FRAMES = 0, KIND_SINE = 1, KIND_CONSTANT = 2, KIND_CYCLE = 3, KIND_DOWNMIX = 4, KIND_STREAM = 5 };

// spsc.rs Header (:246-249) plus the "sender dropped" flag that Arc::strong_count provides there (:165-167).
// `write` is stored by the producer (host, release), `read` by the consumer (device, system scope).
struct StreamHeader { uint32_t read, write, closed, pad; };
enum : uint32_t { DYN_HAS_FINISHED_FOR = 1u, DYN_STOPPED = 2u };
enum : uint32_t { PEND_FRESH = 1u, PEND_DISCONTINUITY = 2u };
enum : uint32_t { EAR_SKIP = 1u };

struct alignas(16) SrcStatic {
    const float* clip;      // device pointer, 16-byte aligned, tail padded to a multiple of 4 floats
    uint32_t clip_len;      // samples (>= 1)
    uint32_t clip_rate;     // Hz (Frames::rate, frames.rs:20 holds it as f64)
    float fixed_gain;       // FixedGain linear factor (gain.rs:20); 1.0 when there is no wrapper
    float radius;           // SpatialOptions::radius
    float freq_or_value;    // Sine: rad/s (sine.rs:21); Constant: the value
    uint16_t kind;          // KIND_*
    uint16_t fx;            // FX_*: a per-source soft clip around the signal (reinhard.rs:22-50, tanh.rs:16-44 are Signal + Seek wrappers)
};
static_assert(sizeof(SrcStatic) == 32, "SrcStatic layout");
// Downmix<FramesSignal<[f32;2]>> in FAST mode (round 6): a stereo clip the library owns (oddio_hip_frames_from_slice_stereo) carries,
// behind its interleaved frames, the mono clip M[i] = L[i] + R[i] (one f32 add per frame, made when the clip is uploaded).  The
// fused (FAST-mode, 1e-5 tolerance) kernels render a Downmix source as a plain clip over M: lerp(M) is lerp(L) + lerp(R)
// (downmix.rs:27-29 over frame.rs:39-41) to a rounding or two -- ~1e-7 of |L| + |R| -- at the cost of a mono source instead of 3.2 x.
// The exact kernels (ORDERED, single-wave callbacks, FAST_UNFUSED, the contribution rows) keep the two interpolations and the sum.
// SrcStatic::freq_or_value == bits 1: the clip has the mono sum (KIND_DOWNMIX sources only).
__host__ __device__ inline uint32_t downmix_presum_offset(uint32_t stereo_frames) { return ((2u * stereo_frames + 3u) & ~3u) + 4u; }   // floats from the clip's start, 16-byte aligned
// `Reinhard<T>` / `Tanh<T>` wrapped around a Seek-set source, with the source's optional FixedGain inside the clip
// (Reinhard(FixedGain(x)), the default) or outside it (FX_CLIP_FIRST: FixedGain(Reinhard(x)))
enum : uint16_t { FX_REINHARD = 1, FX_TANH = 2, FX_CLIP_FIRST = 4, FX_CHAIN = 8 };
// FX_CHAIN (round 6): any other nest of the Seek wrappers -- FixedGain (gain.rs:39-51), Reinhard (reinhard.rs:42-50), Tanh
// (tanh.rs:36-44) are `impl<T: Seek> Seek`, so Reinhard(Tanh(x)), FixedGain(FixedGain(x)), Reinhard(FixedGain(Tanh(FixedGain(x)))) are
// legal play()s.  Up to 4 wrappers, innermost first, each applied to every sample with its own rounding; the chain lives in a table
// indexed by the source's handle id (StaticHeader::fx_chains, below), and SrcStatic::fixed_gain holds that id as raw bits.  Such sources are
// rendered by the exact per-lane paths (mix_source_rare_body, cycle_render), never by the staged loops.
enum : uint8_t { FXOP_FIXED_GAIN = 1, FXOP_REINHARD = 2, FXOP_TANH = 3 };
struct alignas(8) FxChain {   // (8 + 16 bytes: the header and the gains are one load each)
    uint32_t n;             // wrappers (1..4)
    uint8_t op[4];          // FXOP_*, innermost first
    float gain[4];          // FXOP_FIXED_GAIN: the linear factor (gain.rs:20)
};
static_assert(sizeof(FxChain) == 24, "FxChain layout");
// The chain table is reached from the SrcStatic table itself: the scene allocates one 32-byte header IN FRONT of slot 0 (st[-1]) that
// holds the table's pointer.  The kernels that can meet an FX_CHAIN source have the SrcStatic pointer anyway; a kernel argument of its
// own stays live in two SGPRs through the mix kernels' hot loops, which run at the SGPR limit (measured: +1.4 % on spatial_mix_pair).
struct alignas(16) StaticHeader { const FxChain* fx_chains; uint64_t pad[3]; };
static_assert(sizeof(StaticHeader) == 32, "one SrcStatic slot");

struct alignas(16) SrcDyn {
    double t;               // FramesSignal::t seconds (frames.rs:145)
    float phase;            // Sine::phase (sine.rs:7)
    float state_dt;         // State::dt (spatial.rs:491)
    float tgt_pos[3];       // Motion received (spatial.rs:480-485)
    float tgt_vel[3];
    float prev_pos[3];      // State::prev_position (spatial.rs:489)
    float finished_for;     // Common::finished_for payload (spatial.rs:89)
    uint32_t flags;         // DYN_*
    uint32_t id;            // handle id (the returned `Spatial`), reported when the source stops
};
static_assert(sizeof(SrcDyn) == 64, "SrcDyn layout");

struct alignas(16) SrcPending {
    float pos[3];
    float vel[3];
    uint32_t flags;         // PEND_*
    uint32_t pad;
};
static_assert(sizeof(SrcPending) == 32, "SrcPending layout");

struct alignas(16) EarParams {
    double t_ear;           // inner cursor after seek(prev_state.offset) (spatial.rs:449)
    float dt;               // effective_elapsed / N (spatial.rs:452)
    float g0;               // prev_state.gain
    float dg;               // d_gain (spatial.rs:453)
    float phase_ear;        // Sine phase after seek(prev_state.offset)
                            // (KIND_CYCLE: phase_ear = prev_state.offset, t_ear = effective_elapsed; the
                            //  cursor itself is advanced by cycle_sources)
    uint32_t flags;         // EAR_SKIP: source stopped / not mixed this callback
    uint32_t pad;
};
static_assert(sizeof(EarParams) == 32, "EarParams layout");

// What the walk kernel leaves for the mix kernel per (source, 512-frame tile): everything of frames.rs:176-201's
// set-up that does not depend on the output frame -- the f64 cursor split per 256-frame chunk (`frac0`), the
// resample step, the gain ramp, and the clip window the tile touches as a ready-made buffer descriptor -- so that
// spatial_mix starts each source from 64 bytes instead of redoing the f64 arithmetic per tile.
//   desc     buffer descriptor words 0-2 of the window, clipped to the clip (kernels.h window_desc)
//   info     path (bits 0-2) | SFLAG_* (bits 3-7) | nvec (bits 8-15: 16-byte vectors of the window) | negvec (bits
//            16-23: vectors of the window that lie before the clip start; the descriptor base is the clip start then)
//   ear[e]   {ds, g0, dg, wrel chunk 0 | chunk 1 << 16}: wrel = the chunk's base index relative to the window start
//   frac0    [ear][chunk] the chunk's start offset (frames.rs:181/189)
struct alignas(16) TileRec {
    uint32_t desc[3];
    uint32_t info;
    struct { float ds, g0, dg; uint32_t wrel; } ear[2];
    float frac0[2][2];
};
static_assert(sizeof(TileRec) == 64, "TileRec layout");
constexpr int REC_TILES = 2;   // tiles whose records the walk kernel writes itself; longer callbacks run in passes of this many tiles

// The same for spatial_mix_pair (pair_kernels.h): ONE record per source for a callback of up to 1024 frames (four 256-frame
// chunks), whose window -- every sample both ears touch in the whole callback -- is staged once per source by a workgroup
// of two wavefronts, one per ear.
//   desc     buffer descriptor words 0-2 of the window (kernels.h window_desc)
//   info     path (bits 0-2) | SFLAG_* (bits 3-7) | nvec (bits 8-16: 16-byte vectors of the window) | negvec (bits 17-25)
//   ear[e]   {ds, g0, dg}, the four chunks' start offsets (frames.rs:181/189) and their base indices relative to the window start
struct alignas(8) PairEar {
    float ds, g0, dg;
    uint32_t pad;
    float frac0[4];
    uint16_t wrel[4];
};
struct alignas(16) PairRec {
    uint32_t desc[3];
    uint32_t info;
    PairEar ear[2];
};
static_assert(sizeof(PairEar) == 40 && sizeof(PairRec) == 96, "PairRec layout");
static_assert(sizeof(PairRec) <= REC_TILES * sizeof(TileRec), "a PairRec fits the room of a source's tile records");

struct SceneParams {
    float prev_rot[4];      // (s,x,y,z) listener rotation used for the callback's start
    float rot[4];           //           ... and for its end (spatial.rs:382-386)
    float interval;
    float elapsed;          // interval * N as f32 (spatial.rs:394)
    uint32_t n_frames;
    uint32_t n_sources;     // live slots
    float* cycle_rows;      // Seek-set Cycle contribution rows: [row][ear][cycle_plane] (null: none played)
    uint32_t cycle_plane;   // floats per ear plane (= max_frames)
    uint32_t dmx;           // the callback's mix kernels are the Downmix-capable ones (spatial_mix<.., DMX>): the walk may stage stereo windows
    uint32_t pair;          // the callback's mix kernel is spatial_mix_pair: the walk writes PairRecs instead of tile records
    uint32_t fused;         // the callback's mix kernel is a FAST-mode (fused) instantiation: its staged loops render Tanh sources too (tanh_fast)
    uint32_t track_all;     // TRACK second pass of a sharded scene: every block leaves end - start (the start of the walk's first block is the ranks' prefix, not this rank's to keep)
    uint32_t* bounds_err;   // debug build (-DODDIO_HIP_BOUNDS): {count, first code, first value, first source}; null otherwise
};

// The bounds-checked build (`make debug` -> libodd_hip_debug.so): every index into a staged window, every padded
// re-layout position and every record the walk kernel packs is checked; a violation is counted, its first instance
// recorded, and the access skipped -- the next sample call returns ODDIO_HIP_EBOUNDS.  (The reference runs miri in CI
// and debug_assert!s its ring indices, ring.rs:24-27,52-56; this is the device path's counterpart.)
enum : uint32_t { BOUNDS_WINDOW_INDEX = 1, BOUNDS_PAD_INDEX = 2, BOUNDS_REPACK = 3, BOUNDS_RECORD = 4, BOUNDS_WINDOW_BYTES = 5 };
#ifdef ODDIO_HIP_BOUNDS
#define ODDIO_BOUNDS_CHECK(ERR, OK, CODE, VALUE, SRC)                                                         \
    ((OK) ? true : (oddio_hip::bounds_fail((ERR), (CODE), (uint32_t)(VALUE), (uint32_t)(SRC)), false))
#else
#define ODDIO_BOUNDS_CHECK(ERR, OK, CODE, VALUE, SRC) true
#endif

// Mixer<[f32;2]> of MonoToStereo<mono source> (mixer.rs, signal.rs:61-91)
struct alignas(16) MixStatic {
    const float* clip;
    uint32_t clip_len;
    uint32_t clip_rate;
    float fixed_gain;
    float freq_or_value;
    uint32_t kind;
    uint32_t pad;
};
static_assert(sizeof(MixStatic) == 32, "MixStatic layout");

enum : uint32_t { MIXDYN_STOPPED = 2u, MIXDYN_STOP_REQUESTED = 4u };

struct alignas(16) MixDyn {
    double t;               // FramesSignal::t
    float phase;            // Sine::phase
    uint32_t flags;         // MIXDYN_*
    uint32_t id;            // handle id (the returned `Mixed`)
    uint32_t pad[3];
};
static_assert(sizeof(MixDyn) == 32, "MixDyn layout");

struct alignas(16) MixParams {   // written by mixer_prepass each callback
    double t_start;         // cursor at the start of the callback
    float phase_start;
    uint32_t flags;         // EAR_SKIP
};
static_assert(sizeof(MixParams) == 16, "MixParams layout");

}  // namespace oddio_hip
