/*
 * oddio_oracle.h -- CPU restatement of the oddio SpatialScene / Mixer hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call it -- and there only as the checker /
 * the reported CPU baseline.  The product path (oddio_amd/) never calls into it.
 *
 * It restates, op for op (same f32/f64 types, same evaluation order, same 256/1024-frame
 * chunking, same reverse set-index walk, no FMA contraction), the reference files named on each
 * function (paths relative to the reference crate root, oddio 0.7.4).
 *
 * Pinning status: the reference is Rust and cannot be compiled in the build image, so this oracle
 * is pinned by (1) every known-answer test the reference's own test modules hold for the path
 * (tests/test_oracle_kats.py transcribes them), (2) bit-equality with an independent numpy
 * restatement (oracle/oracle_np.py) on seeded random scenes, (3) analytic spot checks.
 * SpatialScene's numeric output itself is NOT pinned by any reference test ("parity unpinned" for
 * that function).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (see oracle/Makefile).
 */
#ifndef ODDIO_ORACLE_H
#define ODDIO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oo_frames oo_frames;
typedef struct oo_signal oo_signal;

/* ---- Frames<T> (src/frames.rs:19-47) ---- */
oo_frames* oo_frames_from_slice(uint32_t rate, const float* samples, size_t len, int channels);
oo_frames* oo_frames_borrow(uint32_t rate, const float* samples, size_t len, int channels); /* harness: no copy */
void oo_frames_retain(oo_frames* f);
void oo_frames_release(oo_frames* f);

/* ---- signal constructors (each takes ownership of `inner`) ---- */
oo_signal* oo_frames_signal_new(oo_frames* data, double start_seconds); /* frames.rs:156-169 */
oo_signal* oo_sine_new(float phase, float frequency_hz);                 /* sine.rs:18-23 */
oo_signal* oo_constant_new(float l, float r, int channels);              /* constant.rs */
oo_signal* oo_cycle_new(oo_frames* data);                                /* cycle.rs:17-23 */
oo_signal* oo_fixed_gain_new(oo_signal* inner, float db);                /* gain.rs:18-23 */
oo_signal* oo_gain_new(oo_signal* inner);                                /* gain.rs:66-74 */
oo_signal* oo_speed_new(oo_signal* inner);                               /* speed.rs:16-24 */
oo_signal* oo_mono_to_stereo_new(oo_signal* inner);                      /* signal.rs:61-68 */
oo_signal* oo_adapt_new(oo_signal* inner, float initial_rms, float tau, float max_gain,
                       float low, float high);                          /* adapt.rs:25-31, :36-61 */
void oo_constant_set(oo_signal* s, float v0, float v1);                  /* test fixture: adapt.rs:127 */
oo_signal* oo_stream_new(uint32_t rate, size_t size, int channels);      /* stream.rs:24-34 */
size_t oo_stream_write(oo_signal* s, const float* data, size_t n_frames); /* stream.rs:107-113 */
size_t oo_stream_free(const oo_signal* s);                               /* stream.rs:101-103 */
void oo_stream_close(oo_signal* s);                                      /* drop(StreamControl) */
oo_signal* oo_fader_new(oo_signal* inner);                               /* fader.rs:18-28 */
void oo_fader_fade_to(oo_signal* fader, oo_signal* signal, float duration); /* fader.rs:86-92 (takes ownership) */
oo_signal* oo_downmix_new(oo_signal* inner);                             /* downmix.rs:10-15 */
oo_signal* oo_reinhard_new(oo_signal* inner);                            /* reinhard.rs:16-20 */
oo_signal* oo_tanh_new(oo_signal* inner);                                /* tanh.rs:10-14 */
oo_signal* oo_mixer_new(int channels);                                   /* mixer.rs:70-81 */
oo_signal* oo_scene_new(void);                                           /* spatial.rs:170-188 */
/* reference test fixtures */
oo_signal* oo_counting_new(uint32_t start);                              /* signal.rs:97-108 */
oo_signal* oo_time_new(float start);                                     /* ring.rs:86-97 */
oo_signal* oo_finished_new(void);                                        /* spatial.rs:611-627 */
void oo_signal_free(oo_signal* s);

/* ---- Signal / Seek (src/signal.rs:14-58) ---- */
int  oo_channels(const oo_signal* s);
void oo_sample(oo_signal* s, float interval, float* out, size_t n_frames);
void oo_seek(oo_signal* s, float seconds);
int  oo_is_finished(const oo_signal* s);
int  oo_is_seek(const oo_signal* s);
/* lib.rs:90-93 */
void oo_run(oo_signal* s, uint32_t sample_rate, float* out, size_t n_frames);

/* ---- controls ---- */
void   oo_gain_set_amplitude_ratio(oo_signal* gain, float factor);  /* GainControl, gain.rs:158-160 */
void   oo_gain_set_gain_db(oo_signal* gain, float db);              /* gain.rs:141-143 */
void   oo_gain_init_amplitude_ratio(oo_signal* gain, float factor); /* Gain::set_amplitude_ratio, gain.rs:90-93 */
void   oo_speed_set(oo_signal* speed, float factor);                /* speed.rs:52-54 */
double oo_frames_signal_t(const oo_signal* s);
double oo_frames_playback_position(const oo_signal* s);             /* frames.rs:238-240 */
int    oo_frames_control_is_finished(const oo_signal* s);           /* frames.rs:244-247 */
float  oo_sine_phase(const oo_signal* s);

/* Mixer (mixer.rs:18-44).  Returns a handle index valid for the mixer's lifetime. */
int  oo_mixer_play(oo_signal* mixer, oo_signal* signal);
void oo_mixer_stop(oo_signal* mixer, int handle);
int  oo_mixer_is_stopped(const oo_signal* mixer, int handle);
size_t oo_mixer_len(const oo_signal* mixer);

/* SpatialScene (spatial.rs:289-349).  Returns a handle index valid for the scene's lifetime. */
void oo_scene_play_frames_bulk(oo_signal* scene, size_t n, uint32_t rate, const float* clips, size_t clip_len, size_t clip_stride,
                               const uint32_t* clip_of /* NULL: source i plays clip i */, const double* start_seconds,
                               const float* positions, const float* velocities, const float* radii); /* harness: n x play(FramesSignal::new(..)), clips borrowed */
int  oo_scene_play(oo_signal* scene, oo_signal* signal, const float pos[3], const float vel[3],
                   float radius);
int  oo_scene_play_buffered(oo_signal* scene, oo_signal* signal, const float pos[3],
                            const float vel[3], float radius, float max_distance, uint32_t rate,
                            float buffer_duration);
void oo_scene_set_motion(oo_signal* scene, int handle, const float pos[3], const float vel[3],
                         int discontinuity);
int  oo_scene_is_finished(const oo_signal* scene, int handle);
void oo_scene_set_listener_rotation(oo_signal* scene, const float q_sxyz[4]);
size_t oo_scene_len(const oo_signal* scene);          /* seek set */
size_t oo_scene_len_buffered(const oo_signal* scene); /* buffered set */
/* Accumulate in f64 instead of f32 (NOT reference behaviour; used only as the "exact sum"
 * yardstick for large scenes).  out64 has 2*n doubles. */
void oo_scene_sample_f64acc(oo_signal* scene, float interval, double* out64, size_t n_frames);

/* ---- exposed helpers for known-answer tests ---- */
void  oo_rotate(const float q_sxyz[4], const float p[3], float out[3]); /* math/mod.rs:81-94 */
void  oo_ear_state(const float pos[3], int ear, float radius, float* offset, float* gain);
                                                                         /* spatial.rs:531-549 */
/* Smoothed<f32> (smooth.rs:26-91) */
typedef struct { float prev, next, progress; } oo_smoothed;
oo_smoothed oo_smoothed_new(float x);
void  oo_smoothed_advance(oo_smoothed* s, float proportion);
void  oo_smoothed_set(oo_smoothed* s, float v);
float oo_smoothed_get(const oo_smoothed* s);
/* Ring (ring.rs:4-80) */
typedef struct oo_ring oo_ring;
oo_ring* oo_ring_new(size_t capacity);
void  oo_ring_free(oo_ring* r);
void  oo_ring_write(oo_ring* r, oo_signal* s, uint32_t rate, float dt);
void  oo_ring_delay(oo_ring* r, uint32_t rate, float dt);
void  oo_ring_sample(const oo_ring* r, uint32_t rate, float t, float interval, float* out, size_t n);
float oo_ring_write_cursor(const oo_ring* r);
const float* oo_ring_buffer(const oo_ring* r, size_t* len);

/* Bench harness (bench.py's cpu_baseline leg): n_threads pthreads, each running n_callbacks callbacks of its own scene;
 * returns the wall time from the common start to the last thread's end. */
double oo_bench_scenes(size_t n_threads, oo_signal** scenes, uint32_t rate, size_t n_frames, size_t n_callbacks, float* outs, double* per_thread);

#ifdef __cplusplus
}
#endif
#endif
