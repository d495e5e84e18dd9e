/*
 * oddio_oracle.c -- CPU restatement of the oddio SpatialScene / Mixer hot path (see header).
 * TEST INFRASTRUCTURE ONLY.  Every function cites the reference file:line it follows
 * (paths relative to the reference crate root).
 *
 * Compile with: -O2 -ffp-contract=off -fno-fast-math   (Rust never contracts a*b+c, never
 * reassociates; x86-64 SSE2 gives plain IEEE f32/f64 per operation).
 */
#include "oddio_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */

#define OO_TAU 6.28318530717958647692528676655900577f /* core::f32::consts::TAU */
#define SPEED_OF_SOUND 343.0f                          /* spatial.rs:602 */
#define HEAD_RADIUS 0.1075f                            /* spatial.rs:605 */
#define POSITION_SMOOTHING_PERIOD 0.5f                 /* spatial.rs:520 */
#define GAIN_SMOOTHING_PERIOD 0.1f                     /* gain.rs:163 */
#define SPATIAL_CHUNK 256                              /* spatial.rs:393 */
#define MIXER_CHUNK 1024                               /* mixer.rs:77 */

/* Rust `f64 as isize`: truncate toward zero, saturate, NaN -> 0. */
static int64_t f64_as_isize(double x) {
    if (x != x) return 0;
    if (x >= 9223372036854775807.0) return INT64_MAX;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)x;
}
/* Rust `f32 as usize`: truncate toward zero, saturate at 0 / max, NaN -> 0. */
static size_t f32_as_usize(float x) {
    if (x != x || x <= 0.0f) return 0;
    if (x >= 18446744073709551615.0f) return SIZE_MAX;
    return (size_t)x;
}
/* f32::rem_euclid (core): r = self % rhs; if r < 0 { r + |rhs| } else { r } */
static float f32_rem_euclid(float a, float b) {
    float r = fmodf(a, b);
    return r < 0.0f ? r + fabsf(b) : r;
}
static float f32_min(float a, float b) { /* f32::min: NaN-ignoring */
    if (a != a) return b;
    if (b != b) return a;
    return a < b ? a : b;
}
static float f32_max(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    return a > b ? a : b;
}

/* frame.rs:39-41  lerp(a,b,t) = a + t*(b-a) */
static inline float lerp1(float a, float b, float t) {
    float d = b - a;
    float m = t * d;
    return a + m;
}

/* ------------------------------------------------------------------------------------------ */
/* math/mod.rs                                                                                */
/* ------------------------------------------------------------------------------------------ */

typedef struct { float x, y, z; } v3;
typedef struct { float s; v3 v; } quat;

static float v3_norm(v3 a) { /* math/mod.rs:33-35: map(powi(2)).sum().sqrt() */
    float s = 0.0f;
    s = s + a.x * a.x;
    s = s + a.y * a.y;
    s = s + a.z * a.z;
    return sqrtf(s);
}
static float v3_dot(v3 a, v3 b) { /* :37-43 */
    float s = 0.0f;
    s = s + a.x * b.x;
    s = s + a.y * b.y;
    s = s + a.z * b.z;
    return s;
}
static v3 v3_scale(v3 a, float f) { v3 r = {a.x * f, a.y * f, a.z * f}; return r; }      /* :45-47 */
static v3 v3_sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }     /* :49-51 */
static v3 v3_add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }     /* :53-55 */
static v3 v3_mix(v3 a, v3 b, float r) {                                                   /* :57-60 */
    float ir = 1.0f - r;
    v3 o = {ir * a.x + r * b.x, ir * a.y + r * b.y, ir * a.z + r * b.z};
    return o;
}
static quat quat_invert(quat q) { /* :62-67 */
    quat r = {q.s, {-q.v.x, -q.v.y, -q.v.z}};
    return r;
}
static quat quat_mul(quat q, quat r) { /* :69-79, left-associative as written */
    quat o;
    o.s = q.s * r.s - q.v.x * r.v.x - q.v.y * r.v.y - q.v.z * r.v.z;
    o.v.x = q.s * r.v.x + q.v.x * r.s + q.v.y * r.v.z - q.v.z * r.v.y;
    o.v.y = q.s * r.v.y - q.v.x * r.v.z + q.v.y * r.s + q.v.z * r.v.x;
    o.v.z = q.s * r.v.z + q.v.x * r.v.y - q.v.y * r.v.x + q.v.z * r.s;
    return o;
}
static v3 quat_rotate(quat rot, v3 p) { /* :81-94 */
    quat pq = {0.0f, p};
    return quat_mul(rot, quat_mul(pq, quat_invert(rot))).v;
}

void oo_rotate(const float q[4], const float p[3], float out[3]) {
    quat qq = {q[0], {q[1], q[2], q[3]}};
    v3 pp = {p[0], p[1], p[2]};
    v3 r = quat_rotate(qq, pp);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ------------------------------------------------------------------------------------------ */
/* smooth.rs                                                                                  */
/* ------------------------------------------------------------------------------------------ */

oo_smoothed oo_smoothed_new(float x) { oo_smoothed s = {x, x, 1.0f}; return s; }        /* :33-44 */
void oo_smoothed_advance(oo_smoothed* s, float p) {                                      /* :47-49 */
    s->progress = f32_min(s->progress + p, 1.0f);
}
float oo_smoothed_get(const oo_smoothed* s) {                                            /* :67-72, :86-91 */
    float diff = s->next - s->prev;
    return s->prev + s->progress * diff;
}
void oo_smoothed_set(oo_smoothed* s, float v) {                                          /* :57-64 */
    s->prev = oo_smoothed_get(s);
    s->next = v;
    s->progress = 0.0f;
}

/* ------------------------------------------------------------------------------------------ */
/* frames.rs: Frames<T>                                                                       */
/* ------------------------------------------------------------------------------------------ */

struct oo_frames {
    double rate;    /* frames.rs:20 (stored as f64, from u32) */
    size_t len;     /* frames */
    int channels;   /* 1: Frames<f32>, 2: Frames<[f32;2]> */
    int refcount;   /* Arc */
    int borrowed;   /* samples belong to the caller (oo_frames_borrow): not freed */
    float* samples; /* len*channels */
};

oo_frames* oo_frames_from_slice(uint32_t rate, const float* samples, size_t len, int channels) {
    oo_frames* f = (oo_frames*)calloc(1, sizeof(*f));
    f->rate = (double)rate;
    f->len = len;
    f->channels = channels;
    f->refcount = 1;
    f->samples = (float*)malloc(sizeof(float) * (len * channels + 1));
    if (len) memcpy(f->samples, samples, sizeof(float) * len * channels);
    return f;
}
/* Harness helper for large scenes: the clip memory stays with the caller (which must outlive the
 * frames); get_pair never reads past samples[len * channels - 1]. */
oo_frames* oo_frames_borrow(uint32_t rate, const float* samples, size_t len, int channels) {
    oo_frames* f = (oo_frames*)calloc(1, sizeof(*f));
    f->rate = (double)rate;
    f->len = len;
    f->channels = channels;
    f->refcount = 1;
    f->borrowed = 1;
    f->samples = (float*)samples;
    return f;
}
void oo_frames_retain(oo_frames* f) { f->refcount++; }
void oo_frames_release(oo_frames* f) {
    if (--f->refcount == 0) { if (!f->borrowed) free(f->samples); free(f); }
}

/* frames.rs:105-123 get_pair, one channel `ch`.  Note `len - 1` is usize arithmetic: for an empty
 * clip it wraps to usize::MAX and the first branch would index out of bounds (a panic in the
 * reference); empty clips are rejected at the boundary instead. */
static inline void frames_get_pair(const oo_frames* f, int64_t sample, int ch, float* a, float* b) {
    const int C = f->channels;
    if (sample >= 0) {
        uint64_t s = (uint64_t)sample;
        if (s < (uint64_t)f->len - 1) { *a = f->samples[s * C + ch]; *b = f->samples[(s + 1) * C + ch]; }
        else if (s < (uint64_t)f->len) { *a = f->samples[s * C + ch]; *b = 0.0f; }
        else { *a = 0.0f; *b = 0.0f; }
    } else if (sample < -1) { *a = 0.0f; *b = 0.0f; }
    else { *a = 0.0f; *b = f->samples[ch]; }
}

/* ------------------------------------------------------------------------------------------ */
/* ring.rs                                                                                    */
/* ------------------------------------------------------------------------------------------ */

struct oo_ring { float* buffer; size_t len; float write; };

oo_ring* oo_ring_new(size_t capacity) { /* :10-15 */
    oo_ring* r = (oo_ring*)calloc(1, sizeof(*r));
    r->buffer = (float*)calloc(capacity ? capacity : 1, sizeof(float));
    r->len = capacity;
    r->write = 0.0f;
    return r;
}
void oo_ring_free(oo_ring* r) { if (r) { free(r->buffer); free(r); } }
float oo_ring_write_cursor(const oo_ring* r) { return r->write; }
const float* oo_ring_buffer(const oo_ring* r, size_t* len) { *len = r->len; return r->buffer; }

void oo_ring_write(oo_ring* r, oo_signal* s, uint32_t rate, float dt) { /* :18-41 */
    float end = fmodf(r->write + dt * (float)rate, (float)r->len);
    size_t start_idx = f32_as_usize(ceilf(r->write));
    size_t end_idx = f32_as_usize(ceilf(end));
    float interval = 1.0f / (float)rate;
    if (end_idx > start_idx) {
        oo_sample(s, interval, r->buffer + start_idx, end_idx - start_idx);
    } else {
        oo_sample(s, interval, r->buffer + start_idx, r->len - start_idx);
        oo_sample(s, interval, r->buffer, end_idx);
    }
    r->write = end;
}
void oo_ring_delay(oo_ring* r, uint32_t rate, float dt) { /* :45-47 */
    r->write = fmodf(r->write + (float)rate * dt, (float)r->len);
}
void oo_ring_sample(const oo_ring* r, uint32_t rate, float t, float interval, float* out, size_t n) {
    /* :51-79 */
    float offset = f32_rem_euclid(r->write + t * (float)rate, (float)r->len);
    float ds = interval * (float)rate;
    for (size_t i = 0; i < n; i++) {
        size_t trunc = (size_t)offset; /* to_int_unchecked::<usize> */
        float fract = offset - (float)trunc;
        size_t x = trunc;
        float a, b;
        if (x < r->len - 1) { a = r->buffer[x]; b = r->buffer[x + 1]; }
        else if (x < r->len) { a = r->buffer[x]; b = r->buffer[0]; }
        else {
            x = x % r->len;
            offset = (float)x + fract;
            if (x < r->len - 1) { a = r->buffer[x]; b = r->buffer[x + 1]; }
            else { a = r->buffer[x]; b = r->buffer[0]; }
        }
        out[i] = lerp1(a, b, fract);
        offset += ds;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* signals                                                                                    */
/* ------------------------------------------------------------------------------------------ */

enum {
    K_FRAMES, K_SINE, K_CONSTANT, K_CYCLE, K_FIXED_GAIN, K_GAIN, K_SPEED, K_MONO_TO_STEREO,
    K_REINHARD, K_TANH, K_MIXER, K_SCENE, K_COUNTING, K_TIME, K_FINISHED, K_ADAPT, K_DOWNMIX, K_STREAM, K_FADER
};

typedef struct { int stop; oo_signal* inner; } mixed_entry; /* mixer.rs:46-49 */

typedef struct { v3 position, velocity; int discontinuity; } motion; /* spatial.rs:480-485 */

typedef struct spatial_entry {
    /* Common, spatial.rs:83-117 */
    float radius;
    motion received;   /* swap::Receiver current value */
    motion pending;    /* value flushed by the control, not yet refreshed */
    int fresh;
    v3 prev_position;  /* State, spatial.rs:487-492 */
    float state_dt;
    int has_finished_for;
    float finished_for;
    int stopped;
    oo_signal* inner;
    /* buffered variant, spatial.rs:18-29 */
    int buffered;
    uint32_t rate;
    float max_delay;
    oo_ring* queue;
} spatial_entry;

typedef struct { void** items; size_t len, cap; } ptrvec;
static void pv_push(ptrvec* v, void* p) {
    if (v->len == v->cap) { v->cap = v->cap ? v->cap * 2 : 16; v->items = realloc(v->items, v->cap * sizeof(void*)); }
    v->items[v->len++] = p;
}
static void* pv_swap_remove(ptrvec* v, size_t i) { /* Vec::swap_remove, set.rs:183-188 */
    void* p = v->items[i];
    v->items[i] = v->items[v->len - 1];
    v->len--;
    return p;
}

struct oo_signal {
    int kind;
    int channels;
    oo_signal* inner;
    /* K_FRAMES / K_CYCLE */
    oo_frames* data;
    double t;          /* frames.rs:145 ; cycle.rs cursor */
    int64_t sample_t;  /* frames.rs:149 */
    /* K_SINE */
    float phase, frequency;
    /* K_CONSTANT */
    float cval[2];
    /* K_FIXED_GAIN / K_GAIN / K_SPEED */
    float gain;        /* FixedGain.gain */
    float shared;      /* Gain.shared / Speed.speed (atomics in the reference) */
    oo_smoothed smooth;
    /* K_STREAM, stream.rs:6-13 over spsc.rs (ring of capacity + 1 slots, read/write indices) */
    float* sbuf; size_t ssize, sread, swrite, slen; int sclosed, sstopping; uint32_t srate; float st;
    /* K_FADER, fader.rs:10-14: inner is `inner`; the swap channel's received / pending Commands */
    float fprogress; oo_signal* fnext; float fduration; oo_signal* fpend; float fpend_duration; int ffresh;
    /* K_ADAPT, adapt.rs:14-18 + AdaptOptions :36-50 */
    float avg_squared, tau, max_gain, low, high;
    /* fixtures */
    uint32_t counter;
    float time;
    /* K_MIXER */
    ptrvec set;        /* live entries (mixed_entry* / spatial_entry*) */
    ptrvec pending;    /* inserted but not yet `update()`d (set.rs:141-168) */
    ptrvec handles;    /* every entry ever created, by handle index */
    float* staging;
    /* K_SCENE */
    ptrvec bset, bpending; /* buffered set */
    quat rot_received, rot_pending;
    int rot_fresh;
};

static oo_signal* sig_new(int kind, int channels) {
    oo_signal* s = (oo_signal*)calloc(1, sizeof(*s));
    s->kind = kind;
    s->channels = channels;
    return s;
}

int oo_channels(const oo_signal* s) { return s->channels; }

oo_signal* oo_frames_signal_new(oo_frames* data, double start_seconds) { /* frames.rs:156-169 */
    oo_signal* s = sig_new(K_FRAMES, data->channels);
    oo_frames_retain(data);
    s->data = data;
    s->t = start_seconds;
    s->sample_t = f64_as_isize(start_seconds * data->rate);
    return s;
}
oo_signal* oo_sine_new(float phase, float frequency_hz) { /* sine.rs:18-23 */
    oo_signal* s = sig_new(K_SINE, 1);
    s->phase = phase;
    s->frequency = frequency_hz * OO_TAU;
    return s;
}
oo_signal* oo_constant_new(float l, float r, int channels) {
    oo_signal* s = sig_new(K_CONSTANT, channels);
    s->cval[0] = l; s->cval[1] = r;
    return s;
}
oo_signal* oo_cycle_new(oo_frames* data) { /* cycle.rs:17-23 */
    oo_signal* s = sig_new(K_CYCLE, data->channels);
    oo_frames_retain(data);
    s->data = data;
    s->t = 0.0;
    return s;
}
oo_signal* oo_fixed_gain_new(oo_signal* inner, float db) { /* gain.rs:18-23 */
    oo_signal* s = sig_new(K_FIXED_GAIN, inner->channels);
    s->inner = inner;
    s->gain = powf(10.0f, db / 20.0f);
    return s;
}
oo_signal* oo_gain_new(oo_signal* inner) { /* gain.rs:66-74 */
    oo_signal* s = sig_new(K_GAIN, inner->channels);
    s->inner = inner;
    s->shared = 1.0f;
    s->smooth = oo_smoothed_new(1.0f);
    return s;
}
oo_signal* oo_speed_new(oo_signal* inner) { /* speed.rs:16-24 */
    oo_signal* s = sig_new(K_SPEED, inner->channels);
    s->inner = inner;
    s->shared = 1.0f;
    return s;
}
oo_signal* oo_mono_to_stereo_new(oo_signal* inner) { oo_signal* s = sig_new(K_MONO_TO_STEREO, 2); s->inner = inner; return s; }
oo_signal* oo_adapt_new(oo_signal* inner, float initial_rms, float tau, float max_gain, float low, float high) { /* adapt.rs:25-31 */
    oo_signal* s = sig_new(K_ADAPT, inner->channels);
    s->inner = inner;
    s->avg_squared = initial_rms * initial_rms;
    s->tau = tau; s->max_gain = max_gain; s->low = low; s->high = high;
    return s;
}
void oo_constant_set(oo_signal* s, float v0, float v1) { s->cval[0] = v0; s->cval[1] = v1; } /* adapt.rs:127 `adapt.inner.0 = ..` */
oo_signal* oo_stream_new(uint32_t rate, size_t size, int channels) { /* stream.rs:24-34, spsc.rs:11-19 */
    oo_signal* s = sig_new(K_STREAM, channels);
    s->ssize = size + 1;
    s->sbuf = calloc(s->ssize * (size_t)channels, sizeof(float));
    s->srate = rate;
    return s;
}
size_t oo_stream_free(const oo_signal* s) { /* spsc.rs:75-86 */
    if (s->swrite < s->sread) return s->sread - s->swrite - 1;
    if (s->sread >= 1) return s->ssize - s->swrite + (s->sread - 1);
    return s->ssize - s->swrite - 1;
}
size_t oo_stream_write(oo_signal* s, const float* data, size_t n) { /* StreamControl::write -> spsc.rs:27-67 */
    const size_t C = (size_t)s->channels;
    size_t cap1, cap2;
    if (s->swrite < s->sread) { cap1 = s->sread - s->swrite - 1; cap2 = 0; }
    else if (s->sread >= 1) { cap1 = s->ssize - s->swrite; cap2 = s->sread - 1; }
    else { cap1 = s->ssize - s->swrite - 1; cap2 = 0; }
    const size_t n1 = cap1 < n ? cap1 : n;
    memcpy(s->sbuf + s->swrite * C, data, n1 * C * sizeof(float));
    const size_t rest = n - n1;
    const size_t n2 = cap2 < rest ? cap2 : rest;
    memcpy(s->sbuf, data + n1 * C, n2 * C * sizeof(float));
    s->swrite = (s->swrite + n1 + n2) % s->ssize;
    return n1 + n2;
}
void oo_stream_close(oo_signal* s) { s->sclosed = 1; } /* drop(StreamControl) */
static float stream_get(const oo_signal* s, int64_t sample, int ch) { /* stream.rs:37-49 */
    if (sample < 0) return 0.0f;
    if ((uint64_t)sample >= s->slen) return 0.0f;
    return s->sbuf[((s->sread + (size_t)sample) % s->ssize) * (size_t)s->channels + (size_t)ch];
}
static void stream_sample(oo_signal* s, float interval, float* out, size_t n) { /* stream.rs:69-85 */
    const int C = s->channels;
    s->slen = s->swrite >= s->sread ? s->swrite - s->sread : s->swrite + s->ssize - s->sread; /* update(), spsc.rs:128-132 */
    if (s->sclosed) s->sstopping = 1;
    const float s0 = s->st;
    const float ds = interval * (float)s->srate;
    for (size_t i = 0; i < n; i++) {
        const float sv = s0 + ds * (float)i;
        const float x0f = truncf(sv);
        const int64_t x0 = f64_as_isize((double)x0f);
        const float fract = sv - x0f; /* f32::fract */
        for (int c = 0; c < C; c++) out[i * C + c] = lerp1(stream_get(s, x0, c), stream_get(s, x0 + 1, c), fract);
    }
    { /* advance(interval * out.len() as f32), stream.rs:59-64 */
        const float dt = interval * (float)n;
        const float next = s->st + dt * (float)s->srate;
        const float t = f32_min(next, (float)s->slen);
        size_t rel = f32_as_usize(t);
        if (rel > s->slen) rel = s->slen;
        s->sread = (s->sread + rel) % s->ssize; /* release(), spsc.rs:135-142 */
        s->slen -= rel;
        s->st = t - truncf(t);
    }
}
oo_signal* oo_fader_new(oo_signal* inner) { /* fader.rs:18-28 */
    oo_signal* s = sig_new(K_FADER, inner->channels);
    s->inner = inner;
    s->fprogress = 1.0f;
    return s;
}
void oo_fader_fade_to(oo_signal* s, oo_signal* signal, float duration) { /* fader.rs:86-92: a waiting command is replaced */
    if (s->ffresh && s->fpend) oo_signal_free(s->fpend);
    s->fpend = signal; s->fpend_duration = duration; s->ffresh = 1;
}
static void fader_sample(oo_signal* s, float interval, float* out, size_t n) { /* fader.rs:36-73 */
    const int C = s->channels;
    if (s->fprogress >= 1.0f) {
        if (s->ffresh) { /* next.refresh(): the retired signal held by `received` is dropped by the control later */
            if (s->fnext) oo_signal_free(s->fnext);
            s->fnext = s->fpend; s->fduration = s->fpend_duration;
            s->fpend = NULL; s->ffresh = 0;
            s->fprogress = 0.0f;
        } else {
            oo_sample(s->inner, interval, out, n);
            return;
        }
    }
    const float increment = interval / s->fduration;
    float buffer[1024 * 2];
    size_t off = 0;
    while (off < n) {
        const size_t rem = n - off;
        const size_t m = rem < 1024 ? rem : 1024;
        oo_sample(s->inner, interval, buffer, 1024);          /* the whole buffer, :53 */
        oo_sample(s->fnext, interval, out + off * C, rem);    /* everything that is left, :54 */
        for (size_t k = 0; k < m; k++) {
            const float fade_out = sqrtf(1.0f - s->fprogress);
            const float fade_in = sqrtf(s->fprogress);
            for (int c = 0; c < C; c++) {
                float* o = out + (off + k) * C + c;
                *o = buffer[k * C + c] * fade_out + *o * fade_in;
            }
            s->fprogress = f32_min(s->fprogress + increment, 1.0f);
        }
        off += m;
    }
    if (s->fprogress >= 1.0f) { oo_signal* t = s->inner; s->inner = s->fnext; s->fnext = t; } /* mem::swap, :70 */
}
oo_signal* oo_downmix_new(oo_signal* inner) { oo_signal* s = sig_new(K_DOWNMIX, 1); s->inner = inner; return s; } /* downmix.rs:10-15 */
oo_signal* oo_reinhard_new(oo_signal* inner) { oo_signal* s = sig_new(K_REINHARD, inner->channels); s->inner = inner; return s; }
oo_signal* oo_tanh_new(oo_signal* inner) { oo_signal* s = sig_new(K_TANH, inner->channels); s->inner = inner; return s; }
oo_signal* oo_mixer_new(int channels) { /* mixer.rs:70-81 */
    oo_signal* s = sig_new(K_MIXER, channels);
    s->staging = (float*)calloc(MIXER_CHUNK * channels, sizeof(float));
    return s;
}
oo_signal* oo_scene_new(void) { /* spatial.rs:170-188 */
    oo_signal* s = sig_new(K_SCENE, 2);
    quat id = {1.0f, {0.0f, 0.0f, 0.0f}};
    s->rot_received = id;
    s->rot_pending = id;
    return s;
}
oo_signal* oo_counting_new(uint32_t start) { oo_signal* s = sig_new(K_COUNTING, 1); s->counter = start; return s; }
oo_signal* oo_time_new(float start) { oo_signal* s = sig_new(K_TIME, 1); s->time = start; return s; }
oo_signal* oo_finished_new(void) { return sig_new(K_FINISHED, 1); }

static void spatial_entry_free(spatial_entry* e) {
    oo_signal_free(e->inner);
    oo_ring_free(e->queue);
    free(e);
}
void oo_signal_free(oo_signal* s) {
    if (!s) return;
    if (s->kind == K_MIXER) {
        for (size_t i = 0; i < s->handles.len; i++) {
            mixed_entry* e = (mixed_entry*)s->handles.items[i];
            oo_signal_free(e->inner);
            free(e);
        }
    } else if (s->kind == K_SCENE) {
        for (size_t i = 0; i < s->handles.len; i++) spatial_entry_free((spatial_entry*)s->handles.items[i]);
    }
    free(s->set.items); free(s->pending.items); free(s->handles.items);
    free(s->bset.items); free(s->bpending.items);
    free(s->staging);
    free(s->sbuf);
    oo_signal_free(s->fnext);
    oo_signal_free(s->fpend);
    if (s->data) oo_frames_release(s->data);
    oo_signal_free(s->inner);
    free(s);
}

int oo_is_seek(const oo_signal* s) {
    switch (s->kind) {
    case K_FRAMES: case K_SINE: case K_CONSTANT: case K_CYCLE: case K_FINISHED: return 1;
    case K_FIXED_GAIN: case K_MONO_TO_STEREO: case K_REINHARD: case K_TANH: case K_DOWNMIX: return oo_is_seek(s->inner);
    default: return 0; /* Gain, Speed, Mixer, SpatialScene do not implement Seek */
    }
}

/* ---- controls ---- */
void oo_gain_set_amplitude_ratio(oo_signal* g, float factor) { g->shared = factor; }          /* gain.rs:158-160 */
void oo_gain_set_gain_db(oo_signal* g, float db) { g->shared = powf(10.0f, db / 20.0f); }      /* gain.rs:141-143 */
void oo_gain_init_amplitude_ratio(oo_signal* g, float factor) {                                /* gain.rs:90-93 */
    g->shared = factor;
    g->smooth = oo_smoothed_new(factor);
}
void oo_speed_set(oo_signal* s, float factor) { s->shared = factor; }                          /* speed.rs:52-54 */
double oo_frames_signal_t(const oo_signal* s) { return s->t; }
double oo_frames_playback_position(const oo_signal* s) { return (double)s->sample_t / s->data->rate; } /* frames.rs:238-240 */
int oo_frames_control_is_finished(const oo_signal* s) {                                        /* frames.rs:244-247 */
    return s->sample_t >= 0 && (uint64_t)s->sample_t >= (uint64_t)s->data->len;
}
float oo_sine_phase(const oo_signal* s) { return s->phase; }

/* ------------------------------------------------------------------------------------------ */
/* Signal::sample for the leaf sources and filters                                            */
/* ------------------------------------------------------------------------------------------ */

static void frames_sample(oo_signal* s, float interval, float* out, size_t n) { /* frames.rs:176-201 */
    const oo_frames* f = s->data;
    const int C = f->channels;
    double s0 = s->t * f->rate;
    float ds = interval * (float)f->rate;
    int64_t base = f64_as_isize(s0);
    if (fabsf(ds - 1.0f) <= FLT_EPSILON) {
        float fract = (float)(s0 - (double)base);
        for (size_t i = 0; i < n; i++) {
            for (int ch = 0; ch < C; ch++) {
                float a, b;
                frames_get_pair(f, base + (int64_t)i, ch, &a, &b);
                out[i * C + ch] = lerp1(a, b, fract);
            }
        }
    } else {
        float offset = (float)(s0 - (double)base);
        for (size_t i = 0; i < n; i++) {
            int64_t trunc = (int64_t)offset; /* to_int_unchecked::<isize>: toward zero */
            float fract = offset - (float)trunc;
            for (int ch = 0; ch < C; ch++) {
                float a, b;
                frames_get_pair(f, base + trunc, ch, &a, &b);
                out[i * C + ch] = lerp1(a, b, fract);
            }
            offset += ds;
        }
    }
    s->t += (double)interval * (double)n;
    s->sample_t = f64_as_isize(s->t * f->rate);
}

static void sine_seek_to(oo_signal* s, float t) { /* sine.rs:25-28 */
    s->phase = fmodf(s->phase + t * s->frequency, OO_TAU);
}
static void sine_sample(oo_signal* s, float interval, float* out, size_t n) { /* sine.rs:34-40 */
    for (size_t i = 0; i < n; i++) {
        float t = interval * (float)i;
        out[i] = sinf(t * s->frequency + s->phase);
    }
    sine_seek_to(s, interval * (float)n);
}

static void cycle_sample(oo_signal* s, float interval, float* out, size_t n) { /* cycle.rs:26-53 */
    const oo_frames* f = s->data;
    const int C = f->channels;
    float ds = interval * (float)(uint32_t)f->rate; /* self.frames.rate() as f32 */
    size_t base = (size_t)f64_as_isize(s->t);
    float offset = (float)(s->t - (double)base);
    for (size_t i = 0; i < n; i++) {
        size_t trunc = (size_t)offset;
        float fract = offset - (float)trunc;
        size_t x = base + trunc;
        size_t ia, ib;
        if (x < f->len - 1) { ia = x; ib = x + 1; }
        else if (x < f->len) { ia = x; ib = 0; }
        else {
            base = 0;
            offset = (float)(x % f->len) + fract;
            size_t x2 = (size_t)offset;
            if (x2 < f->len - 1) { ia = x2; ib = x2 + 1; } else { ia = x2; ib = 0; }
        }
        for (int ch = 0; ch < C; ch++) out[i * C + ch] = lerp1(f->samples[ia * C + ch], f->samples[ib * C + ch], fract);
        offset += ds;
    }
    s->t = (double)base + (double)offset;
}

static void gain_sample(oo_signal* s, float interval, float* out, size_t n) { /* gain.rs:103-122 */
    const int C = s->channels;
    oo_sample(s->inner, interval, out, n);
    float shared = s->shared;
    if (s->smooth.next != shared) oo_smoothed_set(&s->smooth, shared);
    if (s->smooth.progress == 1.0f) {
        float g = oo_smoothed_get(&s->smooth);
        if (g != 1.0f) for (size_t i = 0; i < n * C; i++) out[i] = out[i] * g;
        return;
    }
    for (size_t i = 0; i < n; i++) {
        float g = oo_smoothed_get(&s->smooth);
        for (int ch = 0; ch < C; ch++) out[i * C + ch] = out[i * C + ch] * g;
        oo_smoothed_advance(&s->smooth, interval / GAIN_SMOOTHING_PERIOD);
    }
}

static void mixer_update(oo_signal* m) { /* set.rs:141-168: Insert -> push, in send order */
    for (size_t i = 0; i < m->pending.len; i++) pv_push(&m->set, m->pending.items[i]);
    m->pending.len = 0;
}

static void mixer_sample(oo_signal* m, float interval, float* out, size_t n) { /* mixer.rs:92-119 */
    const int C = m->channels;
    mixer_update(m);
    for (size_t i = 0; i < n * C; i++) out[i] = 0.0f;
    for (size_t i = m->set.len; i-- > 0;) {
        mixed_entry* e = (mixed_entry*)m->set.items[i];
        if (e->stop || oo_is_finished(e->inner)) {
            e->stop = 1;
            pv_swap_remove(&m->set, i);
            continue;
        }
        size_t done = 0;
        while (done < n) {
            size_t k = n - done < MIXER_CHUNK ? n - done : MIXER_CHUNK;
            oo_sample(e->inner, interval, m->staging, k);
            for (size_t j = 0; j < k * C; j++) out[done * C + j] = out[done * C + j] + m->staging[j];
            done += k;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* spatial.rs                                                                                 */
/* ------------------------------------------------------------------------------------------ */

static v3 smoothed_position(const spatial_entry* e, float dt_arg, const motion* next) { /* :501-511 */
    float dt = e->state_dt + dt_arg;
    v3 position_change = v3_scale(next->velocity, dt);
    v3 naive_position = v3_add(e->prev_position, position_change);
    v3 intended_position = v3_add(next->position, position_change);
    return v3_mix(naive_position, intended_position, f32_min(dt / POSITION_SMOOTHING_PERIOD, 1.0f));
}

static void ear_state(v3 p, int ear, float radius, float* offset, float* gain) { /* :531-549, :573-598 */
    v3 ear_pos = {ear == 0 ? -HEAD_RADIUS : HEAD_RADIUS, 0.0f, 0.0f};
    float sign = ear == 0 ? -1.0f : 1.0f;
    v3 ear_dir = {sign * 4.0f / sqrtf(17.0f), 0.0f, -1.0f / sqrtf(17.0f)};
    float distance = v3_norm(v3_sub(p, ear_pos));
    *offset = distance * (-1.0f / SPEED_OF_SOUND);
    float distance_gain = radius / f32_max(distance, radius);
    float stereo_gain;
    if (distance < 1e-3f) stereo_gain = 0.5f + 0.5f;
    else stereo_gain = 0.5f + v3_dot(ear_dir, v3_scale(p, 0.5f / distance));
    *gain = stereo_gain * distance_gain;
}
void oo_ear_state(const float pos[3], int ear, float radius, float* offset, float* gain) {
    v3 p = {pos[0], pos[1], pos[2]};
    ear_state(p, ear, radius, offset, gain);
}

/* accumulate s*gain into out (f32, reference) or out64 (f64 yardstick) */
static inline void acc(float* out, double* out64, size_t frame, int ear, float v) {
    if (out64) out64[frame * 2 + ear] += (double)v;
    else out[frame * 2 + ear] = out[frame * 2 + ear] + v;
}

static void mix_seek(spatial_entry* e, v3 prev_position, v3 next_position, float elapsed,
                     float* out, double* out64, size_t n) { /* spatial.rs:445-469 */
    float buf[SPATIAL_CHUNK];
    for (int ear = 0; ear < 2; ear++) {
        float off0, g0, off1, g1;
        ear_state(prev_position, ear, e->radius, &off0, &g0);
        ear_state(next_position, ear, e->radius, &off1, &g1);
        oo_seek(e->inner, off0);
        float effective_elapsed = (elapsed + off1) - off0;
        float dt = effective_elapsed / (float)n;
        float d_gain = (g1 - g0) / (float)n;
        size_t i = 0;
        for (size_t done = 0; done < n; done += SPATIAL_CHUNK) {
            size_t len = n - done < SPATIAL_CHUNK ? n - done : SPATIAL_CHUNK;
            oo_sample(e->inner, dt, buf, len);
            for (size_t k = 0; k < len; k++) {
                float gain = g0 + (float)i * d_gain;
                acc(out, out64, done + k, ear, buf[k] * gain);
                i += 1;
            }
        }
        oo_seek(e->inner, -effective_elapsed - off0);
    }
    oo_seek(e->inner, elapsed);
}

static void mix_buffered(spatial_entry* e, v3 prev_position, v3 next_position, float elapsed,
                         float* out, double* out64, size_t n) { /* spatial.rs:402-431 */
    float buf[SPATIAL_CHUNK];
    oo_ring_write(e->queue, e->inner, e->rate, elapsed);
    for (int ear = 0; ear < 2; ear++) {
        float off0, g0, off1, g1;
        ear_state(prev_position, ear, e->radius, &off0, &g0);
        ear_state(next_position, ear, e->radius, &off1, &g1);
        float prev_offset = f32_max(off0 - elapsed, -e->max_delay);
        float next_offset = f32_max(off1, -e->max_delay);
        float dt = (next_offset - prev_offset) / (float)n;
        float d_gain = (g1 - g0) / (float)n;
        size_t i = 0;
        for (size_t done = 0; done < n; done += SPATIAL_CHUNK) {
            size_t len = n - done < SPATIAL_CHUNK ? n - done : SPATIAL_CHUNK;
            float t = prev_offset + (float)i * dt;
            oo_ring_sample(e->queue, e->rate, t, dt, buf, len);
            for (size_t k = 0; k < len; k++) {
                float gain = g0 + (float)i * d_gain;
                acc(out, out64, done + k, ear, buf[k] * gain);
                i += 1;
            }
        }
    }
}

static void walk_set(ptrvec* pending, ptrvec* set, quat prev_rot, quat rot, float elapsed,
                     float* out, double* out64, size_t n) { /* spatial.rs:191-265 */
    for (size_t i = 0; i < pending->len; i++) pv_push(set, pending->items[i]); /* set.update() */
    pending->len = 0;
    for (size_t i = set->len; i-- > 0;) {
        spatial_entry* e = (spatial_entry*)set->items[i];
        v3 prev_position, next_position;
        {
            motion orig_next = e->received;
            if (e->fresh) { /* motion.refresh() */
                e->received = e->pending;
                e->fresh = 0;
                if (e->received.discontinuity) e->prev_position = e->received.position;
                else e->prev_position = smoothed_position(e, 0.0f, &orig_next);
                e->state_dt = 0.0f;
            }
            prev_position = quat_rotate(prev_rot, smoothed_position(e, 0.0f, &e->received));
            next_position = quat_rotate(rot, smoothed_position(e, elapsed, &e->received));
            e->state_dt += elapsed;
        }
        float distance = v3_norm(prev_position);
        if (e->has_finished_for) {
            if (e->finished_for > distance / SPEED_OF_SOUND) e->stopped = 1;
            else e->finished_for = e->finished_for + elapsed;
        } else if (oo_is_finished(e->inner)) {
            e->has_finished_for = 1;
            e->finished_for = elapsed;
        }
        if (e->stopped) { pv_swap_remove(set, i); continue; }
        if (e->buffered) mix_buffered(e, prev_position, next_position, elapsed, out, out64, n);
        else mix_seek(e, prev_position, next_position, elapsed, out, out64, n);
    }
}

static void scene_sample_impl(oo_signal* s, float interval, float* out, double* out64, size_t n) {
    /* spatial.rs:376-471 */
    quat prev_rot = s->rot_received;
    if (s->rot_fresh) { s->rot_received = s->rot_pending; s->rot_fresh = 0; }
    quat rot = s->rot_received;
    if (out64) for (size_t i = 0; i < 2 * n; i++) out64[i] = 0.0;
    else for (size_t i = 0; i < 2 * n; i++) out[i] = 0.0f;
    float elapsed = interval * (float)n;
    walk_set(&s->bpending, &s->bset, prev_rot, rot, elapsed, out, out64, n);
    walk_set(&s->pending, &s->set, prev_rot, rot, elapsed, out, out64, n);
}
void oo_scene_sample_f64acc(oo_signal* s, float interval, double* out64, size_t n) {
    scene_sample_impl(s, interval, NULL, out64, n);
}

static spatial_entry* spatial_entry_new(oo_signal* scene, oo_signal* inner, const float pos[3],
                                        const float vel[3], float radius) { /* spatial.rs:94-116 */
    spatial_entry* e = (spatial_entry*)calloc(1, sizeof(*e));
    v3 p = {pos[0], pos[1], pos[2]}, v = {vel[0], vel[1], vel[2]};
    e->radius = radius;
    e->received.position = p; e->received.velocity = v; e->received.discontinuity = 0;
    e->pending = e->received;
    e->prev_position = p;
    e->state_dt = 0.0f;
    e->inner = inner;
    pv_push(&scene->handles, e);
    return e;
}
int oo_scene_play(oo_signal* scene, oo_signal* signal, const float pos[3], const float vel[3], float radius) {
    spatial_entry* e = spatial_entry_new(scene, signal, pos, vel, radius);
    pv_push(&scene->pending, e);
    return (int)scene->handles.len - 1;
}
/* Harness helper: n x `scene.play(FramesSignal::new(frames_i, start_i), SpatialOptions{..})` with
 * frames_i = the mono clip at clips + (clip_of ? clip_of[i] : i) * clip_stride (borrowed, see oo_frames_borrow).  Same calls
 * as the one-at-a-time path, made from C so that 10^5..10^6-source scenes set up in seconds. */
void oo_scene_play_frames_bulk(oo_signal* scene, size_t n, uint32_t rate, const float* clips, size_t clip_len, size_t clip_stride,
                               const uint32_t* clip_of, const double* start_seconds, const float* positions, const float* velocities,
                               const float* radii) {
    for (size_t i = 0; i < n; ++i) {
        oo_frames* f = oo_frames_borrow(rate, clips + (clip_of ? (size_t)clip_of[i] : i) * clip_stride, clip_len, 1);
        oo_signal* sig = oo_frames_signal_new(f, start_seconds[i]);
        oo_frames_release(f);   /* the signal holds the Arc now */
        (void)oo_scene_play(scene, sig, positions + 3 * i, velocities + 3 * i, radii[i]);
    }
}
int oo_scene_play_buffered(oo_signal* scene, oo_signal* signal, const float pos[3], const float vel[3],
                           float radius, float max_distance, uint32_t rate, float buffer_duration) {
    /* spatial.rs:314-340, :31-56 */
    spatial_entry* e = spatial_entry_new(scene, signal, pos, vel, radius);
    float max_delay = max_distance / SPEED_OF_SOUND + buffer_duration;
    e->buffered = 1;
    e->rate = rate;
    e->max_delay = max_delay;
    e->queue = oo_ring_new(f32_as_usize(ceilf(max_delay * (float)rate)) + 1);
    oo_ring_delay(e->queue, rate, f32_min(v3_norm(e->received.position) / SPEED_OF_SOUND, max_delay));
    pv_push(&scene->bpending, e);
    return (int)scene->handles.len - 1;
}
void oo_scene_set_motion(oo_signal* scene, int h, const float pos[3], const float vel[3], int disc) {
    spatial_entry* e = (spatial_entry*)scene->handles.items[h]; /* spatial.rs:137-149 */
    v3 p = {pos[0], pos[1], pos[2]}, v = {vel[0], vel[1], vel[2]};
    e->pending.position = p; e->pending.velocity = v; e->pending.discontinuity = disc;
    e->fresh = 1;
}
int oo_scene_is_finished(const oo_signal* scene, int h) { return ((spatial_entry*)scene->handles.items[h])->stopped; }
void oo_scene_set_listener_rotation(oo_signal* scene, const float q[4]) { /* spatial.rs:345-349 */
    quat r = {q[0], {q[1], q[2], q[3]}};
    scene->rot_pending = quat_invert(r);
    scene->rot_fresh = 1;
}
size_t oo_scene_len(const oo_signal* scene) { return scene->set.len; }
size_t oo_scene_len_buffered(const oo_signal* scene) { return scene->bset.len; }

int oo_mixer_play(oo_signal* m, oo_signal* signal) { /* mixer.rs:18-26 */
    mixed_entry* e = (mixed_entry*)calloc(1, sizeof(*e));
    e->inner = signal;
    pv_push(&m->handles, e);
    pv_push(&m->pending, e);
    return (int)m->handles.len - 1;
}
void oo_mixer_stop(oo_signal* m, int h) { ((mixed_entry*)m->handles.items[h])->stop = 1; }
int oo_mixer_is_stopped(const oo_signal* m, int h) { return ((mixed_entry*)m->handles.items[h])->stop; }
size_t oo_mixer_len(const oo_signal* m) { return m->set.len; }

/* ------------------------------------------------------------------------------------------ */
/* dispatch                                                                                   */
/* ------------------------------------------------------------------------------------------ */

void oo_sample(oo_signal* s, float interval, float* out, size_t n) {
    const int C = s->channels;
    switch (s->kind) {
    case K_FRAMES: frames_sample(s, interval, out, n); break;
    case K_SINE: sine_sample(s, interval, out, n); break;
    case K_CONSTANT: /* constant.rs:16-18 */
        for (size_t i = 0; i < n; i++) for (int ch = 0; ch < C; ch++) out[i * C + ch] = s->cval[ch];
        break;
    case K_CYCLE: cycle_sample(s, interval, out, n); break;
    case K_FIXED_GAIN: /* gain.rs:32-37 */
        oo_sample(s->inner, interval, out, n);
        for (size_t i = 0; i < n * C; i++) out[i] = out[i] * s->gain;
        break;
    case K_GAIN: gain_sample(s, interval, out, n); break;
    case K_SPEED: /* speed.rs:32-35 */
        oo_sample(s->inner, interval * s->shared, out, n);
        break;
    case K_MONO_TO_STEREO: /* signal.rs:73-80 */
        oo_sample(s->inner, interval, out, n);
        for (size_t i = 2 * n; i-- > 0;) out[i] = out[i / 2];
        break;
    case K_REINHARD: /* reinhard.rs:28-35 */
        oo_sample(s->inner, interval, out, n);
        for (size_t i = 0; i < n * C; i++) out[i] = out[i] / (1.0f + fabsf(out[i]));
        break;
    case K_TANH: /* tanh.rs:22-29 */
        oo_sample(s->inner, interval, out, n);
        for (size_t i = 0; i < n * C; i++) out[i] = tanhf(out[i]);
        break;
    case K_STREAM: stream_sample(s, interval, out, n); break;
    case K_FADER: fader_sample(s, interval, out, n); break;
    case K_DOWNMIX: { /* downmix.rs:23-33: the inner signal always renders the whole 256-frame buffer */
        const int IC = s->inner->channels;
        float buf[256 * 2];
        for (size_t done = 0; done < n; done += 256) {
            const size_t len = n - done < 256 ? n - done : 256;
            oo_sample(s->inner, interval, buf, 256);
            for (size_t i = 0; i < len; i++) {
                float acc = 0.0f; /* Iterator::sum::<f32>() */
                for (int c = 0; c < IC; c++) acc = acc + buf[i * IC + c];
                out[done + i] = acc;
            }
        }
        break;
    }
    case K_ADAPT: { /* adapt.rs:69-87 */
        const float alpha = 1.0f - expf(-interval / s->tau);
        oo_sample(s->inner, interval, out, n);
        for (size_t i = 0; i < n; i++) {
            float sample = 0.0f; /* iter().sum::<f32>(): the zero's sign cannot survive the squaring below */
            for (int c = 0; c < C; c++) sample = sample + out[i * C + c];
            s->avg_squared = sample * sample * alpha + s->avg_squared * (1.0f - alpha);
            const float avg_peak = sqrtf(s->avg_squared) * sqrtf(2.0f);
            float gain;
            if (avg_peak < s->low) { gain = s->low / avg_peak; gain = f32_min(gain, s->max_gain); }
            else if (avg_peak > s->high) gain = s->high / avg_peak;
            else gain = 1.0f;
            for (int c = 0; c < C; c++) out[i * C + c] = out[i * C + c] * gain;
        }
        break;
    }
    case K_MIXER: mixer_sample(s, interval, out, n); break;
    case K_SCENE: scene_sample_impl(s, interval, out, NULL, n); break;
    case K_COUNTING: /* signal.rs:101-107 */
        for (size_t i = 0; i < n; i++) { out[i] = (float)s->counter; s->counter += 1; }
        break;
    case K_TIME: /* ring.rs:90-96 */
        for (size_t i = 0; i < n; i++) { out[i] = s->time; s->time = s->time + interval; }
        break;
    case K_FINISHED: /* spatial.rs:616-618 */
        for (size_t i = 0; i < n; i++) out[i] = 0.0f;
        break;
    }
}

void oo_seek(oo_signal* s, float seconds) {
    switch (s->kind) {
    case K_FRAMES: s->t += (double)seconds; break; /* frames.rs:211-213 */
    case K_SINE: sine_seek_to(s, seconds); break;  /* sine.rs:43-47 */
    case K_CYCLE: { /* cycle.rs:57-60 */
        double len = (double)s->data->len;
        double x = s->t + (double)seconds * (double)(uint32_t)s->data->rate;
        double r = fmod(x, len);
        s->t = r < 0.0 ? r + fabs(len) : r;
        break;
    }
    case K_FIXED_GAIN: case K_MONO_TO_STEREO: case K_REINHARD: case K_TANH: case K_DOWNMIX: oo_seek(s->inner, seconds); break;
    default: break; /* Constant / FinishedSignal: no-op */
    }
}

int oo_is_finished(const oo_signal* s) {
    switch (s->kind) {
    case K_FRAMES: /* frames.rs:204-206; (len - 1) is usize arithmetic */
        return s->t >= (double)(uint64_t)((uint64_t)s->data->len - 1) / s->data->rate;
    case K_FINISHED: return 1;
    case K_STREAM: return s->sstopping && s->st == (float)s->slen; /* stream.rs:88-90 */
    case K_FIXED_GAIN: case K_GAIN: case K_SPEED: case K_MONO_TO_STEREO: case K_REINHARD: case K_TANH: case K_ADAPT: case K_DOWNMIX:
        return oo_is_finished(s->inner);
    default: return 0;
    }
}

void oo_run(oo_signal* s, uint32_t sample_rate, float* out, size_t n) { /* lib.rs:90-93 */
    float interval = 1.0f / (float)sample_rate;
    oo_sample(s, interval, out, n);
}


/* ---- bench harness (not part of the restatement): T threads, each rendering its own scene -------------------------
 * What a user of the reference would do to use every host core: T independent partial scenes, one audio thread each
 * (the partial buffers summed afterwards).  Returns the wall time of the slowest thread's `n_callbacks` callbacks,
 * all threads released together; each thread's seconds land in `per_thread` (may be NULL). */
#include <pthread.h>
#include <time.h>
typedef struct {
    oo_signal* scene; uint32_t rate; size_t n_frames, n_callbacks; float* out;
    pthread_barrier_t* start; double seconds;
} oo_bench_arg;
static double oo_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static void* oo_bench_thread(void* p) {
    oo_bench_arg* a = (oo_bench_arg*)p;
    pthread_barrier_wait(a->start);
    const double t0 = oo_now();
    for (size_t k = 0; k < a->n_callbacks; k++) oo_run(a->scene, a->rate, a->out, a->n_frames);
    a->seconds = oo_now() - t0;
    return NULL;
}
double oo_bench_scenes(size_t n_threads, oo_signal** scenes, uint32_t rate, size_t n_frames, size_t n_callbacks, float* outs, double* per_thread) {
    pthread_t* th = (pthread_t*)calloc(n_threads, sizeof(*th));
    oo_bench_arg* args = (oo_bench_arg*)calloc(n_threads, sizeof(*args));
    pthread_barrier_t start;
    pthread_barrier_init(&start, NULL, (unsigned)n_threads + 1u);
    for (size_t t = 0; t < n_threads; t++) {
        args[t].scene = scenes[t]; args[t].rate = rate; args[t].n_frames = n_frames; args[t].n_callbacks = n_callbacks;
        args[t].out = outs + t * n_frames * 2; args[t].start = &start;
        pthread_create(&th[t], NULL, oo_bench_thread, &args[t]);
    }
    pthread_barrier_wait(&start);
    const double t0 = oo_now();
    for (size_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    const double wall = oo_now() - t0;
    for (size_t t = 0; t < n_threads; t++) if (per_thread) per_thread[t] = args[t].seconds;
    pthread_barrier_destroy(&start);
    free(th); free(args);
    return wall;
}
