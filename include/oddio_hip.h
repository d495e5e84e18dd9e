/*
 * oddio_hip.h -- C ABI of the MI355X-native oddio hot path (libodd_hip.so).
 *
 * This is the drop-in boundary for oddio's SpatialScene / Mixer sample loop.  The reference crate
 * (oddio 0.7.4) has no FFI layer of its own: its operator API is the Rust trait pair
 * `Signal` / `Seek` (src/signal.rs:14-58), the driver `oddio::run` (src/lib.rs:90-93) and the
 * scene / mixer controls (src/spatial.rs:289-349, src/mixer.rs:18-44).  Each entry point below
 * names the reference item it replaces; INTEGRATION.md shows the Rust `extern "C"` shim a
 * maintainer would add so that `HipSpatialScene: Signal<Frame = [f32; 2]>` forwards here.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative ODDIO_HIP_E* / a
 *     positive hipError_t on failure; oddio_hip_last_error() gives a thread-local message.
 *   - vectors are float[3] (x,y,z); quaternions are float[4] = (s, x, y, z) like mint::Quaternion.
 *   - `out` buffers are interleaved stereo [L0,R0,L1,R1,...] == oddio::frame_stereo's view
 *     (src/lib.rs:98-100), caller-owned and fully overwritten (src/spatial.rs:389-391).
 *   - threading mirrors the reference: exactly one thread calls the sample / run entry points of a scene/mixer;
 *     control calls (play / set_motion / set_listener_rotation / stop) may come from another
 *     thread and take effect at the top of the next *_sample (src/set.rs:141-168, src/swap.rs).
 *   - no device allocation happens inside *_sample; the GPU submission makes it soft real time.
 */
#ifndef ODDIO_HIP_H
#define ODDIO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ODDIO_HIP_ABI_VERSION 1

enum {
    ODDIO_HIP_OK = 0,
    ODDIO_HIP_EINVAL = -1,    /* bad argument */
    ODDIO_HIP_ENOMEM = -2,    /* capacity (max_sources / max_frames) exceeded or allocation failed */
    ODDIO_HIP_ENODEV = -3,    /* no usable HIP device */
    ODDIO_HIP_ESTATE = -4,    /* unknown / released handle */
    ODDIO_HIP_EBOUNDS = -5,   /* libodd_hip_debug.so only: a device-side bounds check failed during the previous callback */
};

enum { /* oddio_hip_scene_set_postfx */
    ODDIO_HIP_POSTFX_NONE = 0,
    ODDIO_HIP_POSTFX_REINHARD = 1, /* oddio::Reinhard::new(scene), src/reinhard.rs:28-35 */
    ODDIO_HIP_POSTFX_TANH = 2,     /* oddio::Tanh::new(scene),     src/tanh.rs:22-29 */
};

enum { /* oddio_hip_scene_set_mode */
    ODDIO_HIP_MODE_FAST = 0,    /* sources spread over the whole chip; deterministic tree sum.  When a callback is
                                   rendered by more than one wavefront (more than 16 sources) the lerp, the gain
                                   ramp and the running sums use fused multiply-adds (one rounding where
                                   src/frame.rs:39-41 and src/spatial.rs:459-460 have two): within 1e-5 of the
                                   reference, not its bits -- use ORDERED for those */
    ODDIO_HIP_MODE_ORDERED = 1, /* the contributions are added in the reference's reverse-index order:
                                   bit-comparable with the sequential f32 sum of src/spatial.rs:204,460.
                                   Up to 32 sources one wavefront walks the set; above that every source's
                                   contribution is rendered on the whole chip and a second kernel adds the
                                   rows in order (about 5.5x the FAST callback at 262 144 sources, 1.5x at
                                   1024; set_mode allocates 2 x 8 KiB -- mixers: 1 x -- per source slot and
                                   1024 frames for it, on the calling thread: back-to-back
                                   oddio_hip_scene_sample_device calls on the scene's own stream overlap the
                                   render of one callback with the sum of the one before). */
    ODDIO_HIP_MODE_FAST_UNFUSED = 2, /* scenes: FAST's tree sum with the reference's arithmetic in every contribution
                                   (a + t*(b-a), prev_gain + i*d_gain, o += s*gain: each operation rounded by itself,
                                   src/frame.rs:39-41, src/spatial.rs:459-460) -- what FAST was before round 3; ~3-6 %
                                   slower per callback at 262 144 sources.  Mixers treat it as FAST. */
    ODDIO_HIP_MODE_TRACKED = 3, /* scenes: the reference's sequential sum (src/spatial.rs:204,460) to ~1e-6 of the output's peak
                                   at a fraction of ORDERED's cost, for the scenes where the tree sum of FAST is farther than 1e-5
                                   from it (at 262 144 sources the reference's f32 sum is itself 1-2e-5 from the exact sum, and
                                   so is every other order).  The callback is mixed twice: a first pass leaves every
                                   workgroup's partial sums; the second restarts each workgroup's running sums at the value
                                   the reference's sum has when its walk reaches that workgroup's sources -- the prefix of
                                   the partial sums -- and adds them in the reference's order with its roundings.  A
                                   sequential f32 sum rounds each addend to the ulp of the running sum's binade, so the
                                   restarted sums repeat the reference's rounding errors, not just its exact terms.
                                   Applies to callbacks of up to 1024 frames over >= 32 768 live sources (the TRACK
                                   instantiations of spatial_mix_pair and, up to 512 frames, of spatial_mix: about 2.2x
                                   a FAST callback), and to the ring reads of a buffered set of that size; every other
                                   callback (or set) is an ORDERED one (set_mode allocates as for ORDERED,
                                   plus one buffer of start values).  A scene in a reduce group (sharded): the same mode on
                                   every rank -- between the passes the ranks exchange their totals (ncclAllGather / the p2p
                                   slab) and each starts at the sum of the ranks above it, the reference's walk order
                                   over contiguous index shards.  Mixers: the fast path (plain MonoToStereo / mono sources) of
                                   mixers of >= 8 192 sources is mixed twice in the same way (mixer_kernels.h TRACK);
                                   smaller mixers and the general path (Gain / Speed chains, ...) are ORDERED. */
};

typedef struct oddio_hip_frames oddio_hip_frames; /* == Arc<Frames<f32>>, src/frames.rs:19-22 */
typedef struct oddio_hip_scene oddio_hip_scene;   /* == SpatialScene + SpatialSceneControl */
typedef struct oddio_hip_mixer oddio_hip_mixer;   /* == Mixer<[f32;2]> or Mixer<f32>, + MixerControl */

/* ---- library ---- */
int oddio_hip_abi_version(void);
/* 1 for the bounds-checked build (libodd_hip_debug.so, `make -C oddio_amd/csrc debug`), 0 for the product library. */
int oddio_hip_bounds_checked(void);
const char* oddio_hip_last_error(void);
int oddio_hip_device_count(int* count);

/* ---- Frames<f32> (src/frames.rs:26-47 `Frames::from_slice`) ----
 * Uploads `len` mono samples once; the handle is reference counted like the Arc.  Empty clips are
 * rejected (the reference panics on them in get_pair, src/frames.rs:111). */
int oddio_hip_frames_from_slice(int device, uint32_t rate, const float* samples, size_t len,
                                oddio_hip_frames** out);
/* Frames<[f32; 2]> (interleaved stereo, e.g. examples/wav.rs:45-70): playable in a Mixer, or in a scene under Downmix
 * (oddio_hip_scene_play_frames_downmix).  The upload also stores the mono clip L[i] + R[i] behind the frames (+ 50 % device memory):
 * what FAST-mode scenes render a Downmix source from -- lerp(L + R) for the reference's lerp(L) + lerp(R) (src/downmix.rs:27-29),
 * ~1e-7 of |L| + |R| apart, inside FAST's 1e-5 contract; ORDERED / FAST_UNFUSED interpolate both channels and stay bit-exact.
 * ODDIO_HIP_DOWNMIX_PRESUM=0 in the environment keeps the stereo windows in every mode. */
int oddio_hip_frames_from_slice_stereo(int device, uint32_t rate, const float* interleaved,
                                       size_t n_frames, oddio_hip_frames** out);
/* Same, but `dev_samples` is already a device pointer on `device`.  copy != 0: D2D copy;
 * copy == 0: borrow (caller keeps the memory alive; `len` must be a multiple of 4 floats and the
 * pointer 16-byte aligned). */
int oddio_hip_frames_from_device(int device, uint32_t rate, const float* dev_samples, size_t len,
                                 int copy, oddio_hip_frames** out);
int oddio_hip_frames_retain(oddio_hip_frames* f);
int oddio_hip_frames_release(oddio_hip_frames* f);
int oddio_hip_frames_info(const oddio_hip_frames* f, uint32_t* rate, size_t* len);
/* Arc::strong_count: how many owners (caller handles, playing sources, pending fades) hold the clip. */
int oddio_hip_frames_refcount(const oddio_hip_frames* f, int* count);

/* ---- SpatialScene (src/spatial.rs:160-189 `SpatialScene::new`) ---- */
int oddio_hip_scene_create(int device, uint32_t max_sources, uint32_t max_frames,
                           oddio_hip_scene** out);
int oddio_hip_scene_destroy(oddio_hip_scene* scene);

/* SpatialSceneControl::play (src/spatial.rs:289-302) of a seekable mono signal with
 * SpatialOptions{position, velocity, radius} (src/spatial.rs:354-371).  The signal is one of:
 *   FramesSignal::new(frames, start_seconds)  (src/frames.rs:156-169), optionally wrapped in
 *   FixedGain::new(.., gain_db) (src/gain.rs:18-23; pass NAN for "no FixedGain wrapper");
 *   Sine::new(phase, frequency_hz)            (src/sine.rs:18-23);
 *   Constant::new(value)                      (src/constant.rs);
 *   Cycle::new(frames)                        (src/cycle.rs:17-23; Seek: src/cycle.rs:56-61), optionally
 *   inside FixedGain.  A Cycle's cursor is a serial rounding chain through both ears
 *   (src/cycle.rs:52 inside src/spatial.rs:446-468): one wavefront per Cycle replays it; at most
 *   1024 per scene (environment ODDIO_HIP_MAX_CYCLE), ODDIO_HIP_ENOMEM beyond that.
 * `*source_id` is the handle (== the returned `Spatial`). */
int oddio_hip_scene_play_frames(oddio_hip_scene* scene, oddio_hip_frames* frames,
                                double start_seconds, float fixed_gain_db, const float position[3],
                                const float velocity[3], float radius, uint32_t* source_id);
int oddio_hip_scene_play_sine(oddio_hip_scene* scene, float phase, float frequency_hz,
                              float fixed_gain_db, const float position[3],
                              const float velocity[3], float radius, uint32_t* source_id);
int oddio_hip_scene_play_constant(oddio_hip_scene* scene, float value, const float position[3],
                                  const float velocity[3], float radius, uint32_t* source_id);
/* play(Downmix::new(FramesSignal::new(stereo_frames, start_seconds)), options) (src/downmix.rs:8-47),
 * optionally inside FixedGain: both channels are interpolated and summed.  `frames` must come from
 * oddio_hip_frames_from_slice_stereo.  Like the reference's Downmix::sample (src/downmix.rs:24-29) the
 * inner clock advances a whole 256-frame buffer per chunk, also for a short last chunk. */
int oddio_hip_scene_play_frames_downmix(oddio_hip_scene* scene, oddio_hip_frames* frames,
                                        double start_seconds, float fixed_gain_db,
                                        const float position[3], const float velocity[3], float radius,
                                        uint32_t* source_id);
/* play(Cycle::new(frames)) (src/cycle.rs:17-23; optionally inside FixedGain).  Mono clips of fewer than 2^30 samples
 * (ODDIO_HIP_EINVAL otherwise: the device replays the cursor with 32-bit index arithmetic). */
int oddio_hip_scene_play_cycle(oddio_hip_scene* scene, oddio_hip_frames* frames, float fixed_gain_db,
                               const float position[3], const float velocity[3], float radius,
                               uint32_t* source_id);
/* Bulk form of play_frames for large scenes: n sources, arrays of length n (positions and
 * velocities are [n][3]); ids receives n handles (may be NULL). */
int oddio_hip_scene_play_frames_batch(oddio_hip_scene* scene, size_t n,
                                      oddio_hip_frames* const* frames, const double* start_seconds,
                                      const float* fixed_gain_db /* NULL: none */,
                                      const float* positions, const float* velocities,
                                      const float* radii, uint32_t* ids);

/* ---- buffered sources: SpatialSceneControl::play_buffered (src/spatial.rs:314-340) ----
 * For signals that are not `Seek`.  The inner signal is a leaf (ODDIO_HIP_LEAF_*: FramesSignal /
 * Sine / Constant, arguments as in the play_* calls above) wrapped in up to 4 filters, innermost
 * first: FixedGain(db) (src/gain.rs:9-51), Gain (src/gain.rs:58-127, param = initial amplitude
 * ratio, normally 1) and Speed (src/speed.rs, param = initial factor, normally 1).  The scene owns a
 * `Ring` (src/ring.rs) of ceil((max_distance / 343 + buffer_duration) * rate) + 1 samples per source
 * in HBM, allocated here (control thread) and sampled at `rate`. */
enum { ODDIO_HIP_LEAF_FRAMES = 0, ODDIO_HIP_LEAF_SINE = 1, ODDIO_HIP_LEAF_CONSTANT = 2,
       ODDIO_HIP_LEAF_CYCLE = 3 /* Cycle::new(frames), src/cycle.rs (buffered sources and Mixer chains) */,
       ODDIO_HIP_LEAF_DOWNMIX = 4 /* Downmix::new(FramesSignal(stereo frames)), src/downmix.rs:18-47: oddio_hip_scene_play_filtered only */ };
enum { ODDIO_HIP_FILTER_FIXED_GAIN = 1, ODDIO_HIP_FILTER_GAIN = 2, ODDIO_HIP_FILTER_SPEED = 3,
       /* per-source soft clips, `Reinhard<T>` (src/reinhard.rs:22-50) and `Tanh<T>` (src/tanh.rs:16-44): Signal + Seek
        * wrappers over any signal, param unused.  In buffered and Mixer chains anywhere among the filters; in
        * oddio_hip_scene_play_filtered see there. */
       ODDIO_HIP_FILTER_REINHARD = 4, ODDIO_HIP_FILTER_TANH = 5 };
typedef struct oddio_hip_filter { int kind; float param; } oddio_hip_filter;
/* play(filters(leaf), options) for the filters that keep a signal `Seek` -- FixedGain(db) (src/gain.rs:9-51,
 * :39-51 Seek), Reinhard (src/reinhard.rs:42-50) and Tanh (src/tanh.rs:36-44), each `impl<T: Seek> Seek` -- innermost
 * first, up to 4 of them in any order and multiplicity: `Reinhard::new(FixedGain::new(x, db))`,
 * `Reinhard::new(Tanh::new(x))`, `FixedGain::new(FixedGain::new(x, a), b)` (two roundings, like the reference),
 * `Reinhard::new(Downmix::new(x))`.  Every wrapper is applied to every sample of the source before the distance gain
 * and the sum (src/spatial.rs:457-462 samples the wrapped signal).  At most one FixedGain and one soft clip: rendered
 * inline by the staged kernels; longer nests: the exact per-lane path (correct and ORDERED-bit-exact, several times
 * slower per source).  ODDIO_HIP_EINVAL for Gain / Speed (not Seek: oddio_hip_scene_play_buffered).
 * leaf_kind: ODDIO_HIP_LEAF_FRAMES (mono clip) / _SINE / _CONSTANT / _CYCLE / _DOWNMIX (stereo clip), arguments as in
 * the play_* calls above. */
int oddio_hip_scene_play_filtered(oddio_hip_scene* scene, int leaf_kind, oddio_hip_frames* frames,
                                  double start_seconds, float phase, float frequency_hz_or_value,
                                  const oddio_hip_filter* filters, int n_filters,
                                  const float position[3], const float velocity[3], float radius,
                                  uint32_t* source_id);
int oddio_hip_scene_reserve_buffered(oddio_hip_scene* scene, uint32_t max_buffered);
int oddio_hip_scene_play_buffered(oddio_hip_scene* scene, int leaf_kind, oddio_hip_frames* frames,
                                  double start_seconds, float phase, float frequency_hz_or_value,
                                  const oddio_hip_filter* filters, int n_filters,
                                  const float position[3], const float velocity[3], float radius,
                                  float max_distance, uint32_t rate, float buffer_duration,
                                  uint32_t* source_id);
/* GainControl::set_amplitude_ratio / set_gain (src/gain.rs:141-160), SpeedControl::set_speed
 * (src/speed.rs:52-54) of filter `filter_index` (position in the `filters` array) of a buffered
 * source; like the reference's relaxed atomics they take effect at the next sample call. */
int oddio_hip_source_set_gain(oddio_hip_scene* scene, uint32_t source_id, int filter_index, float amplitude_ratio);
int oddio_hip_source_set_gain_db(oddio_hip_scene* scene, uint32_t source_id, int filter_index, float db);
int oddio_hip_source_set_speed(oddio_hip_scene* scene, uint32_t source_id, int filter_index, float factor);
/* GainControl::amplitude_ratio / gain (src/gain.rs:133-150: gain = 20 * log10(amplitude_ratio)) and
 * SpeedControl::speed (src/speed.rs:47-49): the value the control last stored, or the one the filter
 * was built with -- what the reference's relaxed load returns on the thread that stores. */
int oddio_hip_source_get_amplitude_ratio(oddio_hip_scene* scene, uint32_t source_id, int filter_index, float* amplitude_ratio);
int oddio_hip_source_get_gain_db(oddio_hip_scene* scene, uint32_t source_id, int filter_index, float* db);
int oddio_hip_source_get_speed(oddio_hip_scene* scene, uint32_t source_id, int filter_index, float* factor);
/* n x SpatialSceneControl::play_buffered(filters(FramesSignal::new(frames[i], start_seconds[i])), ..)
 * (src/spatial.rs:314-340) under one lock, the rings from one stretch of device memory: how a scene
 * with 10^5..10^6 Gain / Speed sources is populated.  The chain's filter kinds are shared;
 * filter_params holds n_filters values per source ([n][n_filters], meaning as oddio_hip_filter.param). */
int oddio_hip_scene_play_buffered_batch(oddio_hip_scene* scene, size_t n, oddio_hip_frames* const* frames,
                                        const double* start_seconds, const int* filter_kinds, int n_filters,
                                        const float* filter_params, const float* positions,
                                        const float* velocities, const float* radii, float max_distance,
                                        uint32_t rate, float buffer_duration, uint32_t* ids);
/* n GainControl::set_amplitude_ratio / SpeedControl::set_speed stores (src/gain.rs:158-160,
 * src/speed.rs:52-54) to filter `filter_index` of n buffered sources, under one lock. */
int oddio_hip_scene_set_control_batch(oddio_hip_scene* scene, size_t n, const uint32_t* source_ids,
                                      int filter_index, const float* values);
/* The same n stores, and n Spatial::set_motion calls (src/spatial.rs:137-149; positions / velocities
 * [n][3]), with the handle ids and the values in DEVICE memory (readable on the scene's device): one
 * message on the control queue, applied by a kernel in message order at the next sample call -- for
 * hosts that compute gains or trajectories on the GPU.  The arrays must stay valid until the sample call
 * that consumes the message has executed (oddio_hip_scene_device_updates_pending below tells when).  Ids of sources that have left the
 * scene are skipped.  The get_* calls above do not see device-side stores. */
int oddio_hip_scene_set_control_device(oddio_hip_scene* scene, size_t n, const uint32_t* d_source_ids,
                                       int filter_index, const float* d_values);
int oddio_hip_scene_set_motion_device(oddio_hip_scene* scene, size_t n, const uint32_t* d_source_ids,
                                      const float* d_positions, const float* d_velocities, int discontinuity);
/* How long the arrays of the two calls above must live: a message is consumed by the sample call that drains it from the
 * control queue -- normally the next one, a later one when that call's staging was still busy -- and read by a kernel
 * that call enqueues.  *n_pending = the device-array messages no sample call has handed to the stream yet; once it is 0,
 * oddio_hip_scene_synchronize (or any wait on the scene's stream) makes every array passed so far free to release. */
int oddio_hip_scene_device_updates_pending(oddio_hip_scene* scene, size_t* n_pending);
/* Which kernels render the buffered set (results are identical): 1 (default) = the batched path for
 * FramesSignal leaves under FixedGain / Gain / Speed chains (ring write 16 sources per wavefront, ring
 * reads in the Seek set's mix kernel) with the general kernel for every other shape; 0 = the general
 * one-wavefront-per-source kernel for everything (tests; env ODDIO_HIP_BUFFERED_FAST=0 at the first
 * play_buffered does the same). */
int oddio_hip_scene_set_buffered_fast(oddio_hip_scene* scene, int enable);
/* (tests) the number of buffered sources the last callback of up to 1024 frames left to the general
 * kernel (0 when every source took the batched path); waits for the scene's stream. */
int oddio_hip_debug_buffered_slow(oddio_hip_scene* scene, uint32_t* n_slow);
/* (bench) FramesSignal::t = seconds for every buffered source with a FramesSignal leaf, in stream
 * order.  Not part of the reference's interface (a buffered signal is not Seek): bench.py keeps its
 * sources inside their clips with it, as it does with oddio_hip_scene_seek_all for the Seek set. */
int oddio_hip_debug_reset_buffered_clock(oddio_hip_scene* scene, double seconds);
/* `scene.recv_buffered.len()` after the last sample call */
int oddio_hip_scene_len_buffered(oddio_hip_scene* scene, size_t* len);

/* Spatial::set_motion (src/spatial.rs:137-149) */
int oddio_hip_source_set_motion(oddio_hip_scene* scene, uint32_t source_id, const float position[3],
                                const float velocity[3], int discontinuity);
/* Bulk form of set_motion: n handles, positions/velocities [n][3], one discontinuity flag. */
int oddio_hip_scene_set_motion_batch(oddio_hip_scene* scene, size_t n, const uint32_t* source_ids,
                                     const float* positions, const float* velocities,
                                     int discontinuity);
/* Spatial::is_finished (src/spatial.rs:154-156): true once the source ended AND its propagation
 * delay has elapsed (src/spatial.rs:243-261). */
int oddio_hip_source_is_finished(oddio_hip_scene* scene, uint32_t source_id, int* finished);
/* Dropping the `Spatial` handle: the id may be reused once the source has been removed. */
int oddio_hip_source_release(oddio_hip_scene* scene, uint32_t source_id);
/* FramesSignalControl::playback_position (src/frames.rs:238-240), as of the last sample call. */
int oddio_hip_source_playback_position(oddio_hip_scene* scene, uint32_t source_id, double* seconds);

/* Adapt::new(scene, initial_rms, AdaptOptions{tau, max_gain, low, high}) (src/adapt.rs:25-31, :36-61;
 * the filter itself :63-87): adaptive gain that keeps the running RMS of the (channel-summed) output
 * inside [low, high].  Applied on the device to the scene's stereo sum BEFORE the Reinhard/Tanh post
 * filter, i.e. the composition is Reinhard::new(Adapt::new(scene, ..)) as src/adapt.rs:7-9 recommends.
 * Calling it (re)starts the filter state at initial_rms^2; enable = 0 removes the filter.  Call it
 * while no `sample` is in flight (it is a constructor in the reference). */
int oddio_hip_scene_set_adapt(oddio_hip_scene* scene, int enable, float initial_rms, float tau,
                              float max_gain, float low, float high);
/* SpatialSceneControl::set_listener_rotation (src/spatial.rs:345-349); stores the inverse. */
int oddio_hip_scene_set_listener_rotation(oddio_hip_scene* scene, const float rotation_sxyz[4]);

int oddio_hip_scene_set_postfx(oddio_hip_scene* scene, int postfx);
int oddio_hip_scene_set_mode(oddio_hip_scene* scene, int mode);
/* What a *_sample* call does when all of its staging slots for control updates are still unread by the device,
 * i.e. the caller has enqueued several callbacks with control traffic (oddio_hip_scene_sample_device) without
 * waiting for the device:
 *   0 (default) never wait: the updates stay queued and take effect one callback later (src/signal.rs:11-13, "the
 *     audio thread waits for nobody") -- beyond that run-ahead the output is no longer callback-for-callback the
 *     reference's;
 *   1 wait for the oldest slot (offline rendering into device buffers: every update takes effect in the callback
 *     it was sent before, like the reference's, at the price of a host wait). */
int oddio_hip_scene_set_exact_updates(oddio_hip_scene* scene, int wait_for_staging);
/* Number of live sources in the set after the last sample call (== `scene.recv.len()` in the
 * reference's own test, src/spatial.rs:643). */
int oddio_hip_scene_len(oddio_hip_scene* scene, size_t* len);

/* Signal::sample for SpatialScene (src/spatial.rs:376-471): renders n_frames stereo frames
 * separated by `interval` seconds into host memory `out` (2*n_frames floats).  n_frames may be 0
 * (the walk -- motion ingest, finished bookkeeping -- still runs, as in the reference). */
int oddio_hip_scene_sample(oddio_hip_scene* scene, float interval, float* out, size_t n_frames);
/* oddio::run (src/lib.rs:90-93): interval = 1.0 / sample_rate as f32. */
int oddio_hip_scene_run(oddio_hip_scene* scene, uint32_t sample_rate, float* out, size_t n_frames);
/* Same as _sample, but `dev_out` is device memory on the scene's device and the call only
 * enqueues work on the scene's stream (no host synchronisation); source removal bookkeeping is
 * picked up one call later.  Used for multi-GPU reduction and benchmarking. */
int oddio_hip_scene_sample_device(oddio_hip_scene* scene, float interval, float* dev_out,
                                  size_t n_frames);
/* Renders the post-mix filter only (Reinhard/Tanh) over a device buffer of 2*n_frames floats;
 * used after the cross-GPU sum of partial stereo buffers (sharded scenes). */
int oddio_hip_postfx_device(int device, int postfx, float* dev_buf, size_t n_frames, void* hip_stream);
/* ---- one logical scene sharded over the GPUs of a node (BASELINE configs[4]) ----
 * The reference has no multi-device code; its per-frame sum over sources (src/spatial.rs:456-463) is
 * what gets split: every rank owns a contiguous index shard of the sources in its own scene, and after
 * oddio_hip_scene_reduce_init every *_sample* call of that scene sums the ranks' partial stereo
 * buffers with ONE RCCL all-reduce (2*n_frames f32, sum) enqueued on the scene's stream between the
 * shard's own reduction and the post-mix filter (Reinhard/Tanh/Adapt wrap the *scene*, so they run
 * after the sum, on every rank).  All ranks must call *_sample* with the same n_frames AND THE SAME MODE: a TRACKED
 * callback of a grouped scene issues one collective more than the other modes (the ranks' first-pass totals: ncclAllGather /
 * the slab's second block of rows), so oddio_hip_scene_set_mode must take effect between the same two callbacks on every rank.
 * Which collectives a callback issues depends on the mode and the group alone -- the exchange buffer is allocated by reduce_init
 * for every group (ODDIO_HIP_ENOMEM instead of joining without it), and a TRACKED scene needs a librccl that exports
 * ncclAllGather (reduce_init and set_mode return ODDIO_HIP_ENODEV otherwise) -- never on what one rank's allocations or live
 * source count happen to be.  Ranks that disagree anyway: the peer-to-peer group times out (ODDIO_HIP_ESTATE, below); RCCL hangs.
 *   rank 0:    oddio_hip_reduce_unique_id(id)   (ncclGetUniqueId) -> hand `id` to the other ranks
 *   all ranks: oddio_hip_scene_reduce_init(scene, rank, world, id, ODDIO_HIP_UNIQUE_ID_BYTES)
 * RCCL (librccl.so.1) is loaded on first use; a process that never shards does not need it. */
#define ODDIO_HIP_UNIQUE_ID_BYTES 128
int oddio_hip_reduce_unique_id(void* unique_id, size_t unique_id_bytes);
int oddio_hip_scene_reduce_init(oddio_hip_scene* scene, int rank, int world, const void* unique_id,
                                size_t unique_id_bytes);
int oddio_hip_scene_reduce_destroy(oddio_hip_scene* scene);
/* The reduce group as the library sees it: kind 0 none / 1 RCCL all-reduce / 2 peer-to-peer slab; world =
 * ncclCommCount of the communicator (or the slab's rank count); the librccl the functions were resolved
 * from and its ncclGetVersion.  For reports: bench.py prints them next to its multi-GPU numbers. */
int oddio_hip_scene_reduce_info(oddio_hip_scene* scene, int* kind, int* world, int* rccl_version,
                                char* lib_path, size_t lib_path_bytes);
/* The same reduction WITHOUT RCCL and with a fixed summation order: every peer writes its
 * 2*n_frames-float partial into a slab in rank 0's device memory (hipIpc: peer-to-peer over xGMI, or the same HBM
 * when ranks share a GPU), rank 0 adds  ((p0 + p1) + p2) + ...  and every rank copies the mix back -- run-to-run and
 * rank-count deterministic, and usable with several ranks on ONE device.  One process per rank.
 *   rank 0:      oddio_hip_scene_reduce_init_p2p(scene, 0, world, handle, ODDIO_HIP_P2P_HANDLE_BYTES)   fills `handle`
 *   other ranks: oddio_hip_scene_reduce_init_p2p(scene, rank, world, handle, ...)                        with rank 0's bytes
 * Replaces src/spatial.rs:456-463's sum across the shards like the RCCL variant; a rank that does not show up within
 * ~2 s makes the next *_sample* call return ODDIO_HIP_ESTATE instead of hanging the device. */
#define ODDIO_HIP_P2P_HANDLE_BYTES 128
int oddio_hip_scene_reduce_init_p2p(oddio_hip_scene* scene, int rank, int world, void* handle, size_t handle_bytes);
/* Block until everything enqueued on the scene's stream has finished. */
int oddio_hip_scene_synchronize(oddio_hip_scene* scene);
/* Make the scene enqueue on a caller-owned hipStream_t (e.g. the stream a framework's collectives
 * run on) instead of its private stream. */
int oddio_hip_scene_set_stream(oddio_hip_scene* scene, void* hip_stream);
/* The scene's hipStream_t (for callers that order their own work after sample_device -- or BEFORE it: the scene's private stream is
 * hipStreamNonBlocking and does not wait for the NULL stream, so device memory handed to the scene (borrowed clips, update arrays,
 * the dev_out of a *_sample_device call) must be complete, or ordered by an event this stream waits for, before the call that uses it;
 * hipMemset and device-to-device hipMemcpy return before they have run.  See INTEGRATION.md, "Streams"). */
int oddio_hip_scene_stream(oddio_hip_scene* scene, void** hip_stream);
/* Seek::seek applied to every live source (src/signal.rs:48-51): t += seconds. */
int oddio_hip_scene_seek_all(oddio_hip_scene* scene, float seconds);

/* Per-kernel timing of the most recent *_sample* call, measured with hipEvents on the scene's
 * stream (milliseconds): [0] prepass, [1] mix, [2] reduce+postfx.  Blocks until that call's work
 * has finished.  Enabled by oddio_hip_scene_set_profiling(scene, 1); with (scene, 2) only the mix kernel is
 * bracketed (two events per callback instead of four; [0] and [2] read 0); with (scene, 1 + k), k >= 2, only the mix
 * kernel of every k-th call is (an event pair costs the stream ~10 us of gaps around the kernel it brackets: a
 * throughput measurement samples the kernel instead of bracketing every launch). */
int oddio_hip_scene_set_profiling(oddio_hip_scene* scene, int enable);
int oddio_hip_scene_last_kernel_ms(oddio_hip_scene* scene, float ms[3]);
/* The same for up to `max_calls` most recent profiled calls (oldest first; ms is [n][3]); the
 * library keeps the last 512.  Lets a benchmark time the mix kernel over its whole timed region
 * without synchronising inside it. */
int oddio_hip_scene_kernel_ms_history(oddio_hip_scene* scene, float* ms, size_t max_calls,
                                      size_t* n_calls);
/* The buffered set's stages of the last profiled calls (set_profiling(1); callbacks of up to 1024 frames):
 * ms[3 * i + {0, 1, 2}] = walk, ring write through the filter chains (+ the general kernel), ring reads + their sum. */
int oddio_hip_scene_buffered_ms_history(oddio_hip_scene* scene, float* ms, size_t max_calls, size_t* n_calls);

/* Debug: the runtime's view of the mix kernel's residency (64-thread blocks per CU) and its
 * register / LDS footprint. */
int oddio_hip_debug_mix_occupancy(int device, int* blocks_per_cu, int* num_cus, int* vgprs,
                                  int* lds_bytes);

/* SpatialSceneControl::play_buffered(Fader::new(chain).1, ..) (src/fader.rs:16-28) and
 * FaderControl::fade_to (:83-93) for that source: see oddio_hip_mixer_play_fader for the semantics.
 * Arguments as in oddio_hip_scene_play_buffered; at most 256 Fader sources per scene. */
int oddio_hip_scene_play_buffered_fader(oddio_hip_scene* scene, int leaf_kind, oddio_hip_frames* frames,
                                        double start_seconds, float phase, float frequency_hz_or_value,
                                        const oddio_hip_filter* filters, int n_filters,
                                        const float position[3], const float velocity[3], float radius,
                                        float max_distance, uint32_t rate, float buffer_duration,
                                        uint32_t* source_id);
int oddio_hip_source_fade_to(oddio_hip_scene* scene, uint32_t source_id, int leaf_kind,
                             oddio_hip_frames* frames, double start_seconds, float phase,
                             float frequency_hz_or_value, const oddio_hip_filter* filters,
                             int n_filters, float duration);

/* ---- Stream (src/stream.rs) ------------------------------------------------------------------
 * Stream::new(rate, size) -> (StreamControl, Stream) (src/stream.rs:24-34): dynamic audio pushed by
 * another thread.  The SPSC ring (src/spsc.rs) lives in pinned, GPU-visible host memory: `write` is a
 * memcpy plus one release store (no lock, no GPU call), the kernels read the ring directly and publish
 * the read index back.  `channels` is 1 (spatial scenes, mixers) or 2 (mixers).  The handle is the
 * StreamControl; the Stream itself is moved into a scene / mixer by exactly one play call. */
typedef struct oddio_hip_stream oddio_hip_stream;
int oddio_hip_stream_create(int device, uint32_t rate, size_t size_frames, uint32_t channels,
                            oddio_hip_stream** out);
/* StreamControl::write (src/stream.rs:107-113): appends a prefix of `samples` (interleaved when
 * stereo); *consumed = frames taken (the rest should be passed again later). */
int oddio_hip_stream_write(oddio_hip_stream* stream, const float* samples, size_t n_frames,
                           size_t* consumed);
/* StreamControl::free (src/stream.rs:101-103): lower bound of what the next write will take. */
int oddio_hip_stream_free(oddio_hip_stream* stream, size_t* n_frames);
/* drop(StreamControl): no more data will come; the playing Stream reports is_finished once it has
 * been drained (src/stream.rs:71-73, :88-90) and is then removed like any finished source.  Invalidates
 * the handle; the ring is freed when the scene / mixer has dropped the Stream too. */
int oddio_hip_stream_drop(oddio_hip_stream* stream);
/* SpatialSceneControl::play_buffered(filters(Stream), ..) (src/spatial.rs:314-340): the only way a
 * Stream (not Seek) enters a spatial scene. */
int oddio_hip_scene_play_buffered_stream(oddio_hip_scene* scene, oddio_hip_stream* stream,
                                         const oddio_hip_filter* filters, int n_filters,
                                         const float position[3], const float velocity[3],
                                         float radius, float max_distance, uint32_t rate,
                                         float buffer_duration, uint32_t* source_id);

/* ---- Mixer<[f32;2]> (src/mixer.rs:70-81 `Mixer::new`) ---- */
int oddio_hip_mixer_create(int device, uint32_t max_sources, uint32_t max_frames,
                           oddio_hip_mixer** out);
/* Mixer<f32> (src/mixer.rs:46-81 is generic over `T: Frame`; `impl Frame for f32`, src/frame.rs:53-61): plays mono
 * signals only (a clip / stream with two channels is ODDIO_HIP_EINVAL: the reference's type system rules it out), and
 * *_sample / *_run write n_frames floats.  Everything else as for the stereo mixer. */
int oddio_hip_mixer_create_mono(int device, uint32_t max_sources, uint32_t max_frames,
                                oddio_hip_mixer** out);
int oddio_hip_mixer_destroy(oddio_hip_mixer* mixer);
/* MixerControl::play (src/mixer.rs:18-26) of MonoToStereo::new(inner) (src/signal.rs:61-91) with
 * inner = Sine / FramesSignal (optionally FixedGain-wrapped) / Constant, as above. */
int oddio_hip_mixer_play_sine(oddio_hip_mixer* mixer, float phase, float frequency_hz,
                              float fixed_gain_db, uint32_t* source_id);
int oddio_hip_mixer_play_frames(oddio_hip_mixer* mixer, oddio_hip_frames* frames,
                                double start_seconds, float fixed_gain_db, uint32_t* source_id);
int oddio_hip_mixer_play_constant(oddio_hip_mixer* mixer, float value, uint32_t* source_id);
/* MixerControl::play of any supported signal: leaf (ODDIO_HIP_LEAF_*, arguments as above; mono
 * leaves are implicitly MonoToStereo'd, stereo clips play as is) inside up to 4 filters, innermost
 * first (FixedGain / Gain / Speed; see oddio_hip_filter).  A mixer that has ever been given a Gain,
 * Speed, Cycle, a stereo clip or more than one filter renders through the general path from then on: chains over a mono
 * FramesSignal 16 sources per wavefront, every other shape one wavefront per source, a slab per source (8 bytes per source and
 * frame of max_frames) summed afterwards (65 536 Gain<MonoToStereo<FramesSignal>> sources: 0.18 ms per 1024-frame callback,
 * 0.64 in ORDERED mode). */
int oddio_hip_mixer_play_chain(oddio_hip_mixer* mixer, int leaf_kind, oddio_hip_frames* frames,
                               double start_seconds, float phase, float frequency_hz_or_value,
                               const oddio_hip_filter* filters, int n_filters, uint32_t* source_id);
/* GainControl / SpeedControl of filter `filter_index` of a mixer source */
/* MixerControl::play(Fader::new(chain).1) (src/fader.rs:16-28) and FaderControl::fade_to (:83-93): the
 * source cross-fades (constant power, src/fader.rs:57-60) from its current signal to the given chain
 * over `duration` seconds; a fade in progress completes first, a waiting command is replaced.  Both
 * signals must have the same channel count.  Fader::is_finished is always false: the source leaves the
 * mixer only through oddio_hip_mixer_stop.  At most 256 Fader sources per mixer.  Like the reference,
 * the outgoing signal renders a whole 1024-frame buffer per call while a fade runs (src/fader.rs:51-53).
 * Arguments as in oddio_hip_mixer_play_chain. */
int oddio_hip_mixer_play_fader(oddio_hip_mixer* mixer, int leaf_kind, oddio_hip_frames* frames,
                               double start_seconds, float phase, float frequency_hz_or_value,
                               const oddio_hip_filter* filters, int n_filters, uint32_t* source_id);
int oddio_hip_mixer_fade_to(oddio_hip_mixer* mixer, uint32_t source_id, int leaf_kind,
                            oddio_hip_frames* frames, double start_seconds, float phase,
                            float frequency_hz_or_value, const oddio_hip_filter* filters,
                            int n_filters, float duration);
/* MixerControl::play(filters(Stream)) (a mono stream is wrapped in MonoToStereo like other mono leaves) */
int oddio_hip_mixer_play_stream(oddio_hip_mixer* mixer, oddio_hip_stream* stream,
                                const oddio_hip_filter* filters, int n_filters, uint32_t* source_id);
int oddio_hip_mixer_set_gain(oddio_hip_mixer* mixer, uint32_t source_id, int filter_index, float amplitude_ratio);
int oddio_hip_mixer_set_gain_db(oddio_hip_mixer* mixer, uint32_t source_id, int filter_index, float db);
int oddio_hip_mixer_set_speed(oddio_hip_mixer* mixer, uint32_t source_id, int filter_index, float factor);
/* GainControl::amplitude_ratio / gain (src/gain.rs:133-150), SpeedControl::speed (src/speed.rs:47-49) */
int oddio_hip_mixer_get_amplitude_ratio(oddio_hip_mixer* mixer, uint32_t source_id, int filter_index, float* amplitude_ratio);
int oddio_hip_mixer_get_gain_db(oddio_hip_mixer* mixer, uint32_t source_id, int filter_index, float* db);
int oddio_hip_mixer_get_speed(oddio_hip_mixer* mixer, uint32_t source_id, int filter_index, float* factor);
/* Mixed::stop / Mixed::is_stopped (src/mixer.rs:34-43) */
int oddio_hip_mixer_stop(oddio_hip_mixer* mixer, uint32_t source_id);
int oddio_hip_mixer_is_stopped(oddio_hip_mixer* mixer, uint32_t source_id, int* stopped);
/* drop(Mixed): the handle is not used any more; its id is recycled once the source has left the mixer. */
int oddio_hip_mixer_source_release(oddio_hip_mixer* mixer, uint32_t source_id);
int oddio_hip_mixer_len(oddio_hip_mixer* mixer, size_t* len);
int oddio_hip_mixer_set_postfx(oddio_hip_mixer* mixer, int postfx);
/* Adapt::new(mixer, ..): same as oddio_hip_scene_set_adapt for a Mixer (examples/adapt.rs:6-16). */
int oddio_hip_mixer_set_adapt(oddio_hip_mixer* mixer, int enable, float initial_rms, float tau,
                              float max_gain, float low, float high);
/* The sum modes of oddio_hip_scene_set_mode for a Mixer (src/mixer.rs:100-117 walks the set in reverse slot order).  ORDERED on a
 * mixer created for more than 32 sources allocates the contribution rows of the two-kernel path (8 bytes per source and frame
 * of max_frames, on the calling -- control -- thread); without them (out of memory) large mixers keep the one-wavefront walk. */
int oddio_hip_mixer_set_mode(oddio_hip_mixer* mixer, int mode);
/* Signal::sample for Mixer (src/mixer.rs:92-119) / oddio::run */
int oddio_hip_mixer_sample(oddio_hip_mixer* mixer, float interval, float* out, size_t n_frames);
int oddio_hip_mixer_run(oddio_hip_mixer* mixer, uint32_t sample_rate, float* out, size_t n_frames);
/* Mixer::sample (src/mixer.rs:92-119) with the frames left in DEVICE memory (`dev_out`: channels * n_frames floats on the
 * mixer's device) and no wait: the counterpart of oddio_hip_scene_sample_device for hosts that consume the mix on the GPU
 * or enqueue several callbacks.  The work runs on the mixer's stream in call order.  Sources the callback stopped are
 * skipped by later callbacks at once; the handles (oddio_hip_mixer_is_stopped) and the slot bookkeeping learn of them when
 * a later call settles the callback's pinned snapshot -- every oddio_hip_mixer_sample call and, in ORDERED mode (where the
 * sum order is the set order), every call does so before it starts. */
int oddio_hip_mixer_sample_device(oddio_hip_mixer* mixer, float interval, float* dev_out, size_t n_frames);
/* Waits for everything enqueued on the mixer's stream (device-output callbacks). */
int oddio_hip_mixer_synchronize(oddio_hip_mixer* mixer);

#ifdef __cplusplus
}
#endif
#endif /* ODDIO_HIP_H */
