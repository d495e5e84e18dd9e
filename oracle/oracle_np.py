"""Independent numpy-float32 restatement of the oddio hot path (second implementation).

TEST INFRASTRUCTURE ONLY.  Written from the operation-order specification (SURVEY.md Appendix A,
which cites src/spatial.rs:376-471, src/frames.rs:176-213, src/sine.rs:18-47, src/mixer.rs:92-119,
src/math/mod.rs:33-94; round 4: src/ring.rs:4-80, src/gain.rs:18-37,58-127, src/smooth.rs:26-91, src/speed.rs:10-40,
src/spatial.rs:18-57,314-340,395-433 -- the buffered path and the filters), deliberately NOT from oracle/oddio_oracle.c: different structure
(array-at-a-time, functional state), so that bit-equality between the two on seeded scenes is
evidence that both follow the specification (tests/test_oracle_cross.py).

Every arithmetic op is a separately rounded IEEE f32 op (numpy never fuses), f64 where the
reference uses f64.  Transcendentals go through glibc (ctypes -> libm sinf/tanhf/powf/fmodf),
which is what Rust's std f32::sin/tanh/powf/% resolve to on Linux.
"""
from __future__ import annotations

import ctypes
import ctypes.util

import numpy as np

f32 = np.float32
f64 = np.float64

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("sinf", "tanhf", "expf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
for _n in ("powf", "fmodf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float, ctypes.c_float]


def sinf(x):
    x = np.asarray(x, dtype=f32)
    return np.array([_libm.sinf(float(v)) for v in x.ravel()], dtype=f32).reshape(x.shape)


def tanhf(x):
    x = np.asarray(x, dtype=f32)
    return np.array([_libm.tanhf(float(v)) for v in x.ravel()], dtype=f32).reshape(x.shape)


def expf(x):
    return f32(_libm.expf(float(f32(x))))


def powf(a, b):
    return f32(_libm.powf(float(f32(a)), float(f32(b))))


def fmodf(a, b):
    return f32(_libm.fmodf(float(f32(a)), float(f32(b))))


TAU = f32(6.2831855)
EPS = f32(1.1920929e-7)
C_SOUND = f32(343.0)
INV_NEG_C = f32(-1.0) / f32(343.0)
HEAD = f32(0.1075)
SMOOTH = f32(0.5)
SQRT17 = np.sqrt(f32(17.0))
DZ = f32(-1.0) / SQRT17
ONE, HALF, ZERO = f32(1.0), f32(0.5), f32(0.0)


def _trunc_i64(x64):
    """Rust `f64 as isize`."""
    x64 = float(x64)
    if x64 != x64:
        return 0
    return int(max(min(x64, 9.2e18), -9.2e18))


# ---------------------------------------------------------------------------------------------
# quaternion / vector helpers on float32 arrays of shape [3] / [4]=(s,x,y,z)
# ---------------------------------------------------------------------------------------------

def quat_mul(q, r):
    qs, qx, qy, qz = (f32(v) for v in q)
    rs, rx, ry, rz = (f32(v) for v in r)
    return np.array([
        qs * rs - qx * rx - qy * ry - qz * rz,
        qs * rx + qx * rs + qy * rz - qz * ry,
        qs * ry - qx * rz + qy * rs + qz * rx,
        qs * rz + qx * ry - qy * rx + qz * rs,
    ], dtype=f32)


def conj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]], dtype=f32)


def rotate(q, p):
    pq = np.array([0.0, p[0], p[1], p[2]], dtype=f32)
    return quat_mul(q, quat_mul(pq, conj(q)))[1:]


def norm(v):
    s = f32(0.0)
    for c in v:
        s = s + f32(c) * f32(c)
    return np.sqrt(s)


def ear_state(p, ear, radius):
    p = np.asarray(p, dtype=f32)
    ex = -HEAD if ear == 0 else HEAD
    v = np.array([p[0] - ex, p[1] - ZERO, p[2] - ZERO], dtype=f32)
    distance = norm(v)
    offset = distance * INV_NEG_C
    distance_gain = f32(radius) / max(distance, f32(radius))
    if distance < f32(1e-3):
        stereo = HALF + HALF
    else:
        k = HALF / distance
        q = p * k
        dx = (f32(-1.0) if ear == 0 else f32(1.0)) * f32(4.0) / SQRT17
        d = ZERO + dx * q[0]
        d = d + ZERO * q[1]
        d = d + DZ * q[2]
        stereo = HALF + d
    return f32(offset), f32(stereo * distance_gain)


# ---------------------------------------------------------------------------------------------
# sources: plain dicts, functional sample()
# ---------------------------------------------------------------------------------------------

def frames_source(rate, samples, start_seconds=0.0, fixed_gain_db=None):
    s = {"kind": "frames", "rate": int(rate), "samples": np.asarray(samples, dtype=f32), "t": f64(start_seconds)}
    if fixed_gain_db is not None:
        s["fixed_gain"] = powf(10.0, f32(fixed_gain_db) / f32(20.0))
    return s


def sine_source(phase, hz, fixed_gain_db=None):
    s = {"kind": "sine", "phase": f32(phase), "freq": f32(hz) * TAU}
    if fixed_gain_db is not None:
        s["fixed_gain"] = powf(10.0, f32(fixed_gain_db) / f32(20.0))
    return s


def constant_source(value):
    return {"kind": "constant", "value": f32(value)}


def downmix_source(rate, stereo_samples, start_seconds=0.0, fixed_gain_db=None):
    """Downmix::new(FramesSignal::new(Frames<[f32;2]>, start)) (src/downmix.rs, src/frames.rs)."""
    x = np.asarray(stereo_samples, dtype=f32)
    assert x.ndim == 2 and x.shape[1] == 2
    s = {"kind": "downmix", "rate": int(rate), "samples": x, "t": f64(start_seconds)}
    if fixed_gain_db is not None:
        s["fixed_gain"] = powf(10.0, f32(fixed_gain_db) / f32(20.0))
    return s


def cycle_source(rate, samples, fixed_gain_db=None):
    """Cycle::new (src/cycle.rs:17-23): cursor (f64, in samples) starts at 0."""
    s = {"kind": "cycle", "rate": int(rate), "samples": np.asarray(samples, dtype=f32), "cursor": f64(0.0)}
    if fixed_gain_db is not None:
        s["fixed_gain"] = powf(10.0, f32(fixed_gain_db) / f32(20.0))
    return s


# ---- filters around a source (round 4): FixedGain / Gain (with its Smoothed ramp) / Speed ----------------------------

SMOOTHING_PERIOD = f32(0.1)       # src/gain.rs:163


def fixed_gain_filter(inner, db):
    """FixedGain::new(inner, db) (src/gain.rs:18-23): gain = 10^(db / 20)."""
    return {"kind": "fixed", "inner": inner, "gain": powf(10.0, f32(db) / f32(20.0))}


def gain_filter(inner, initial_ratio=None):
    """Gain::new(inner) (src/gain.rs:66-76): shared = 1.0, Smoothed::new(1.0) (src/smooth.rs:34-44).  `initial_ratio`:
    Gain::set_amplitude_ratio before playing (gain.rs:90-93): the store AND a fresh Smoothed."""
    g = {"kind": "gain", "inner": inner, "shared": ONE, "prev": ONE, "next": ONE, "progress": ONE}
    if initial_ratio is not None:
        g["shared"] = f32(initial_ratio)
        g["prev"] = g["next"] = f32(initial_ratio)
    return g


def gain_control_set(g, ratio):
    """GainControl::set_amplitude_ratio (src/gain.rs:157-160): a relaxed store; the filter notices at its next sample call."""
    g["shared"] = f32(ratio)


def speed_filter(inner, speed=1.0):
    """Speed::new(inner) (src/speed.rs:16-24) + SpeedControl::set_speed (:52-54)."""
    return {"kind": "speed", "inner": inner, "speed": f32(speed)}


def speed_control_set(sp, factor):
    sp["speed"] = f32(factor)


def reinhard_filter(inner):
    """Reinhard::new(inner) (src/reinhard.rs:16-20) around a source: `x / (1 + |x|)` on every sample (:28-35)."""
    return {"kind": "reinhard", "inner": inner}


def tanh_filter(inner):
    """Tanh::new(inner) (src/tanh.rs:10-14) around a source: `tanh(x)` on every sample (:22-29)."""
    return {"kind": "tanh", "inner": inner}


def _smoothed_get(g):
    """Smoothed::get (src/smooth.rs:67-72) with f32::interpolate (:84-89): prev + progress * (next - prev)."""
    diff = g["next"] - g["prev"]
    return f32(g["prev"] + g["progress"] * diff)


def _gather_pair(samples, idx):
    """pair(i) = (S(i), S(i+1)) with S(i) = samples[i] inside the clip, 0 outside."""
    n = samples.shape[0]
    def S(i):
        ok = (i >= 0) & (i < n)
        return np.where(ok, samples[np.clip(i, 0, max(n - 1, 0))], ZERO).astype(f32)
    return S(idx), S(idx + 1)


def src_sample(src, interval, n):
    interval = f32(interval)
    kind = src["kind"]
    if kind == "fixed":                                   # FixedGain::sample, src/gain.rs:30-37
        return (src_sample(src["inner"], interval, n) * src["gain"]).astype(f32)
    if kind == "speed":                                   # Speed::sample, src/speed.rs:32-35
        return src_sample(src["inner"], interval * src["speed"], n)
    if kind == "reinhard":                                # Reinhard::sample, src/reinhard.rs:28-35
        return reinhard(src_sample(src["inner"], interval, n))
    if kind == "tanh":                                    # Tanh::sample, src/tanh.rs:22-29
        return tanh_clip(src_sample(src["inner"], interval, n))
    if kind == "gain":                                    # Gain::sample, src/gain.rs:103-122
        out = src_sample(src["inner"], interval, n)
        if src["next"] != src["shared"]:                  # :106-108 -> Smoothed::set, src/smooth.rs:57-64
            src["prev"] = _smoothed_get(src)
            src["next"] = src["shared"]
            src["progress"] = ZERO
        if src["progress"] == ONE:                        # :109-117
            g = _smoothed_get(src)
            return (out * g).astype(f32) if g != ONE else out
        # :118-121: per frame `x * gain.get()` then `advance(interval / SMOOTHING_PERIOD)` = min(progress + step, 1)
        # (src/smooth.rs:47-49).  The running progress is a strictly sequential f32 sum; once it is clamped it stays 1.
        step = interval / SMOOTHING_PERIOD
        assert step > ZERO
        seq = np.full(n + 1, step, dtype=f32)
        seq[0] = src["progress"]
        prog = np.minimum(np.cumsum(seq, dtype=f32), ONE)  # prog[k] = progress before frame k, prog[n] after the call
        diff = src["next"] - src["prev"]
        gains = (src["prev"] + prog[:n] * diff).astype(f32)
        src["progress"] = f32(prog[n])
        return (out * gains).astype(f32)
    if kind == "frames":
        rate64 = f64(src["rate"])
        s0 = src["t"] * rate64
        ds = interval * f32(src["rate"])
        base = _trunc_i64(s0)
        frac0 = f32(s0 - f64(base))
        if abs(ds - ONE) <= EPS:
            idx = base + np.arange(n, dtype=np.int64)
            a, b = _gather_pair(src["samples"], idx)
            out = a + frac0 * (b - a)
        else:
            steps = np.full(n, ds, dtype=f32)
            if n:
                steps[0] = frac0
            offs = np.cumsum(steps, dtype=f32)  # strictly sequential f32 accumulation
            tr = offs.astype(np.int64)           # toward zero
            fr = offs - tr.astype(f32)
            a, b = _gather_pair(src["samples"], base + tr)
            out = a + fr * (b - a)
        src["t"] = src["t"] + f64(interval) * f64(n)
        out = out.astype(f32)
    elif kind == "sine":
        i = np.arange(n, dtype=f32)
        out = sinf((interval * i) * src["freq"] + src["phase"])
        src["phase"] = fmodf(src["phase"] + (interval * f32(n)) * src["freq"], TAU)
    elif kind == "constant":
        out = np.full(n, src["value"], dtype=f32)
    elif kind == "downmix":
        # src/downmix.rs:23-33: chunks of 256; the inner FramesSignal always renders the whole buffer,
        # so its clock moves 256 frames per chunk even when fewer are used.  Each channel is a mono
        # FramesSignal over the same clock (frame::lerp is per channel, src/frame.rs:39-41).
        out = np.zeros(n, dtype=f32)
        for done in range(0, n, 256):
            ln = min(256, n - done)
            chans = []
            for c in range(2):
                mono = {"kind": "frames", "rate": src["rate"], "samples": np.ascontiguousarray(src["samples"][:, c]), "t": src["t"]}
                chans.append(src_sample(mono, interval, 256))
                t_after = mono["t"]
            acc = np.zeros(256, dtype=f32)
            acc = acc + chans[0]
            acc = acc + chans[1]
            out[done:done + ln] = acc[:ln]
            src["t"] = t_after
    elif kind == "cycle":
        # src/cycle.rs:26-53.  The f32 offset restarts whenever the read position passes the end of
        # the clip, so the loop is written frame by frame (numpy scalars: every op rounds to f32).
        smp = src["samples"]
        length = len(smp)
        ds = interval * f32(src["rate"])
        base = int(src["cursor"])                       # `as usize`: toward zero, cursor >= 0
        offset = f32(src["cursor"] - f64(base))
        out = np.zeros(n, dtype=f32)
        for i in range(n):
            tr = int(offset)
            fract = offset - f32(tr)
            x = base + tr
            if x >= length:
                base = 0
                offset = f32(x % length) + fract
                x = int(offset)
            a = smp[x]
            b = smp[x + 1] if x < length - 1 else smp[0]
            out[i] = a + fract * (b - a)
            offset = offset + ds
        src["cursor"] = f64(base) + f64(offset)
    else:
        raise ValueError(kind)
    if "fixed_gain" in src:
        out = out * src["fixed_gain"]
    return out


def src_seek(src, seconds):
    seconds = f32(seconds)
    if src["kind"] in ("fixed", "reinhard", "tanh"):      # Seek passes through (gain.rs:44-50, reinhard.rs:46-49, tanh.rs:40-43)
        return src_seek(src["inner"], seconds)
    if src["kind"] in ("frames", "downmix"):
        src["t"] = src["t"] + f64(seconds)
    elif src["kind"] == "sine":
        src["phase"] = fmodf(src["phase"] + seconds * src["freq"], TAU)
    elif src["kind"] == "cycle":                        # src/cycle.rs:57-60, f64::rem_euclid
        length = f64(len(src["samples"]))
        r = np.fmod(src["cursor"] + f64(seconds) * f64(src["rate"]), length)
        src["cursor"] = r + length if r < 0 else r


def src_is_finished(src):
    if src["kind"] in ("fixed", "gain", "speed", "reinhard", "tanh"):   # is_finished passes through the filters (gain.rs:39-41,124-126, speed.rs:37-39, reinhard.rs:37-39, tanh.rs:31-33)
        return src_is_finished(src["inner"])
    if src["kind"] in ("frames", "downmix"):
        return bool(src["t"] >= f64(len(src["samples"]) - 1) / f64(src["rate"]))
    return False


# ---------------------------------------------------------------------------------------------
# SpatialScene (seekable sources)
# ---------------------------------------------------------------------------------------------

def _rem_euclid_f32(a, b):
    """f32::rem_euclid: r = a % b (fmod); r < 0 -> r + |b|."""
    r = fmodf(f32(a), f32(b))
    return f32(r + abs(f32(b))) if r < ZERO else f32(r)


class Ring:
    """src/ring.rs:4-80 -- the delay queue of a buffered spatial source."""

    def __init__(self, capacity):
        self.buffer = np.zeros(int(capacity), dtype=f32)
        self.write_pos = ZERO

    def delay(self, rate, dt):                            # :45-47
        self.write_pos = fmodf(self.write_pos + f32(rate) * f32(dt), f32(len(self.buffer)))

    def write(self, src, rate, dt):                       # :18-41
        n = len(self.buffer)
        end = fmodf(self.write_pos + f32(dt) * f32(rate), f32(n))
        start_idx = int(np.ceil(self.write_pos))
        end_idx = int(np.ceil(end))
        interval = ONE / f32(rate)
        if end_idx > start_idx:
            self.buffer[start_idx:end_idx] = src_sample(src, interval, end_idx - start_idx)
        else:
            self.buffer[start_idx:] = src_sample(src, interval, n - start_idx)
            self.buffer[:end_idx] = src_sample(src, interval, end_idx)
        self.write_pos = end

    def sample(self, rate, t, interval, n_out):           # :51-79
        buf, n = self.buffer, len(self.buffer)
        offset = _rem_euclid_f32(self.write_pos + f32(t) * f32(rate), f32(n))
        ds = f32(interval) * f32(rate)
        out = np.zeros(n_out, dtype=f32)
        for i in range(n_out):
            trunc = int(offset)                           # to_int_unchecked: offset >= 0 here
            fract = f32(offset - f32(trunc))
            x = trunc
            if x < n - 1:
                a, b = buf[x], buf[x + 1]
            elif x < n:
                a, b = buf[x], buf[0]
            else:
                x = x % n
                offset = f32(f32(x) + fract)
                a, b = (buf[x], buf[x + 1]) if x < n - 1 else (buf[x], buf[0])
            out[i] = f32(a + fract * f32(b - a))          # frame::lerp
            offset = f32(offset + ds)
        return out


class Scene:
    def __init__(self):
        self.set = []
        self.pending = []
        self.bset = []                                    # the buffered set (play_buffered): walked first (spatial.rs:395-433)
        self.bpending = []
        self.all = []
        self.rot = np.array([1, 0, 0, 0], dtype=f32)
        self.rot_pending = None

    def play(self, src, position, velocity=(0, 0, 0), radius=0.1):
        e = {
            "src": src, "radius": f32(radius),
            "pos": np.asarray(position, dtype=f32).copy(), "vel": np.asarray(velocity, dtype=f32).copy(),
            "pending": None, "prev_position": np.asarray(position, dtype=f32).copy(), "dt": f32(0.0),
            "finished_for": None, "stopped": False,
        }
        self.pending.append(e)
        self.all.append(e)
        return len(self.all) - 1

    def play_buffered(self, src, position, velocity=(0, 0, 0), radius=0.1, max_distance=100.0, rate=48000, buffer_duration=0.1):
        """SpatialSceneControl::play_buffered (src/spatial.rs:314-340) -> SpatialSignalBuffered::new (:31-56)."""
        max_delay = f32(max_distance) / C_SOUND + f32(buffer_duration)
        queue = Ring(int(np.ceil(max_delay * f32(rate))) + 1)
        pos = np.asarray(position, dtype=f32).copy()
        queue.delay(rate, min(norm(pos) / C_SOUND, max_delay))
        e = {
            "src": src, "radius": f32(radius), "pos": pos, "vel": np.asarray(velocity, dtype=f32).copy(),
            "pending": None, "prev_position": pos.copy(), "dt": f32(0.0), "finished_for": None, "stopped": False,
            "rate": int(rate), "max_delay": max_delay, "queue": queue,
        }
        self.bpending.append(e)
        self.all.append(e)
        return len(self.all) - 1

    def set_motion(self, h, position, velocity, discontinuity):
        self.all[h]["pending"] = (np.asarray(position, dtype=f32).copy(), np.asarray(velocity, dtype=f32).copy(), bool(discontinuity))

    def set_listener_rotation(self, q):
        self.rot_pending = conj(np.asarray(q, dtype=f32))

    def is_finished(self, h):
        return self.all[h]["stopped"]

    @staticmethod
    def _smoothed(e, dt_arg, pos, vel):
        dt = e["dt"] + f32(dt_arg)
        c = vel * dt
        naive = e["prev_position"] + c
        intended = pos + c
        r = min(dt / SMOOTH, ONE)
        ir = ONE - r
        return (ir * naive + r * intended).astype(f32)

    def _walk(self, the_set, prev_rot, rot, elapsed, render):
        """walk_set (src/spatial.rs:191-265): reverse index order, motion hand-off, the finished / propagation-delay rule,
        swap_remove; `render(e, p0, p1)` for every source that stays."""
        i = len(the_set) - 1
        while i >= 0:
            e = the_set[i]
            if e["pending"] is not None:
                old_pos, old_vel = e["pos"], e["vel"]
                npos, nvel, disc = e["pending"]
                e["pending"] = None
                e["prev_position"] = npos.copy() if disc else self._smoothed(e, 0.0, old_pos, old_vel)
                e["pos"], e["vel"] = npos, nvel
                e["dt"] = f32(0.0)
            p0 = rotate(prev_rot, self._smoothed(e, 0.0, e["pos"], e["vel"]))
            p1 = rotate(rot, self._smoothed(e, elapsed, e["pos"], e["vel"]))
            e["dt"] = e["dt"] + elapsed
            distance = norm(p0)
            if e["finished_for"] is not None:
                if e["finished_for"] > distance / C_SOUND:
                    e["stopped"] = True
                else:
                    e["finished_for"] = e["finished_for"] + elapsed
            elif src_is_finished(e["src"]):
                e["finished_for"] = elapsed
            if e["stopped"]:
                the_set[i] = the_set[-1]
                the_set.pop()
                i -= 1
                continue
            render(e, p0, p1)
            i -= 1

    def sample(self, interval, n, acc_dtype=np.float32):
        interval = f32(interval)
        prev_rot = self.rot
        if self.rot_pending is not None:
            self.rot = self.rot_pending
            self.rot_pending = None
        rot = self.rot
        out = np.zeros((n, 2), dtype=acc_dtype)
        elapsed = interval * f32(n)
        nf = f32(n)

        def render_buffered(e, p0, p1):                   # src/spatial.rs:403-432
            e["queue"].write(e["src"], e["rate"], elapsed)
            for ear in (0, 1):
                off0, g0 = ear_state(p0, ear, e["radius"])
                off1, g1 = ear_state(p1, ear, e["radius"])
                prev_offset = max(f32(off0 - elapsed), f32(-e["max_delay"]))
                next_offset = max(off1, f32(-e["max_delay"]))
                dt = f32(next_offset - prev_offset) / nf
                d_gain = f32(g1 - g0) / nf
                for c0 in range(0, n, 256):
                    ln = min(256, n - c0)
                    t = f32(prev_offset + f32(c0) * dt)
                    buf = e["queue"].sample(e["rate"], t, dt, ln)
                    gain = g0 + np.arange(c0, c0 + ln, dtype=f32) * d_gain
                    contrib = (buf * gain).astype(f32)
                    out[c0:c0 + ln, ear] = out[c0:c0 + ln, ear] + contrib.astype(acc_dtype)

        def render_seek(e, p0, p1):                       # src/spatial.rs:446-468
            src = e["src"]
            for ear in (0, 1):
                off0, g0 = ear_state(p0, ear, e["radius"])
                off1, g1 = ear_state(p1, ear, e["radius"])
                src_seek(src, off0)
                eff = (elapsed + off1) - off0
                dt = eff / nf
                d_gain = (g1 - g0) / nf
                for c0 in range(0, n, 256):
                    ln = min(256, n - c0)
                    buf = src_sample(src, dt, ln)
                    gain = g0 + np.arange(c0, c0 + ln, dtype=f32) * d_gain
                    contrib = (buf * gain).astype(f32)
                    out[c0:c0 + ln, ear] = out[c0:c0 + ln, ear] + contrib.astype(acc_dtype)
                src_seek(src, (-eff) - off0)
            src_seek(src, elapsed)

        self.bset.extend(self.bpending)                   # set.update() then the walk, buffered set first (:395-433)
        self.bpending = []
        self._walk(self.bset, prev_rot, rot, elapsed, render_buffered)
        self.set.extend(self.pending)
        self.pending = []
        self._walk(self.set, prev_rot, rot, elapsed, render_seek)
        return out


# ---------------------------------------------------------------------------------------------
# Mixer (stereo, MonoToStereo<mono source>) and post filters
# ---------------------------------------------------------------------------------------------

class Mixer:
    """Mixer<[f32;2]> of MonoToStereo<src> (mixer.rs:92-119, signal.rs:73-80) or Mixer<f32>."""

    def __init__(self, channels=2):
        self.channels = channels
        self.set = []
        self.pending = []

    def play(self, src):
        e = {"src": src, "stop": False}
        self.pending.append(e)
        return e

    def sample(self, interval, n):
        self.set.extend(self.pending)
        self.pending = []
        out = np.zeros((n, self.channels), dtype=f32)
        i = len(self.set) - 1
        while i >= 0:
            e = self.set[i]
            if e["stop"] or src_is_finished(e["src"]):
                e["stop"] = True
                self.set[i] = self.set[-1]
                self.set.pop()
                i -= 1
                continue
            for c0 in range(0, n, 1024):
                ln = min(1024, n - c0)
                mono = src_sample(e["src"], interval, ln)
                out[c0:c0 + ln, :] = out[c0:c0 + ln, :] + mono[:, None]
            i -= 1
        return out if self.channels == 2 else out


def reinhard(x):
    x = np.asarray(x, dtype=f32)
    return (x / (ONE + np.abs(x))).astype(f32)


class Stream:
    """src/stream.rs + src/spsc.rs, second transcription: the receiver's window is a Python list
    (released items are popped), so none of the ring index arithmetic of the C oracle is shared."""

    def __init__(self, rate, size, channels=1):
        self.rate, self.capacity, self.channels = int(rate), int(size), channels
        self.sent = []          # written, not yet seen by update()
        self.window = []        # Receiver: items [0, len)
        self.t = f32(0.0)
        self.closed = self.stopping = False

    def free(self):
        return self.capacity - len(self.sent) - len(self.window)

    def write(self, samples):
        x = np.asarray(samples, dtype=f32).reshape(len(samples), -1)
        n = min(self.free(), len(x))
        self.sent.extend(x[:n])
        return n

    def close(self):
        self.closed = True

    def sample(self, interval, n):
        interval = f32(interval)
        self.window.extend(self.sent)      # update()
        self.sent = []
        if self.closed:
            self.stopping = True
        zero = np.zeros(self.channels, dtype=f32)
        get = lambda k: self.window[k] if 0 <= k < len(self.window) else zero
        ds = interval * f32(self.rate)
        out = np.zeros((n, self.channels), dtype=f32)
        for i in range(n):
            sv = self.t + ds * f32(i)
            x0 = int(np.trunc(sv))
            fract = sv - np.trunc(sv)
            a, b = get(x0), get(x0 + 1)
            out[i] = a + fract * (b - a)
        nxt = self.t + (interval * f32(n)) * f32(self.rate)
        t = min(nxt, f32(len(self.window)))
        rel = int(t)
        del self.window[:rel]
        self.t = f32(t - np.trunc(t))
        return out[:, 0] if self.channels == 1 else out

    def is_finished(self):
        return bool(self.stopping and self.t == f32(len(self.window)))


class Fader:
    """src/fader.rs:10-93, second transcription.  Signals are objects with `sample(interval, n)` that
    return float32 [n] or [n, C] and advance their own state (e.g. `SrcSignal` below)."""

    def __init__(self, inner):
        self.inner, self.progress = inner, ONE
        self.received = None          # (signal, duration) held by the swap receiver
        self.pending = None           # flushed by fade_to, not yet refreshed

    def fade_to(self, signal, duration):
        self.pending = (signal, f32(duration))

    def sample(self, interval, n):
        interval = f32(interval)
        if self.progress >= ONE:
            if self.pending is not None:
                self.received, self.pending = self.pending, None
                self.progress = ZERO
            else:
                return self.inner.sample(interval, n)
        nxt, duration = self.received
        increment = interval / duration
        pieces = []
        remaining = n
        out_tail = None
        while remaining > 0:
            m = min(1024, remaining)
            buffer = self.inner.sample(interval, 1024)            # the whole buffer every time
            out_tail = np.array(nxt.sample(interval, remaining))  # everything that is left, again
            for k in range(m):
                fade_out = np.sqrt(ONE - self.progress)
                fade_in = np.sqrt(self.progress)
                out_tail[k] = buffer[k] * fade_out + out_tail[k] * fade_in
                self.progress = min(self.progress + increment, ONE)
            pieces.append(out_tail[:m])
            remaining -= m
        if self.progress >= ONE:
            self.inner, self.received = nxt, (self.inner, duration)
        return np.concatenate(pieces).astype(f32)


class SrcSignal:
    """Adapter: a source dict of this module (frames / sine / constant / cycle / downmix) as an object."""

    def __init__(self, src):
        self.src = src

    def sample(self, interval, n):
        return src_sample(self.src, interval, n)


class Adapt:
    """src/adapt.rs:14-87 as a filter over already-rendered frames x[n] or x[n, C] (second,
    independent transcription: the recurrence runs on numpy scalars, the gain law is vectorised)."""

    def __init__(self, initial_rms, tau=0.1, max_gain=np.inf, low=None, high=None):
        r2 = np.sqrt(f32(2.0))
        self.avg_squared = f32(initial_rms) * f32(initial_rms)
        self.tau, self.max_gain = f32(tau), f32(max_gain)
        self.low = f32(0.1) / r2 if low is None else f32(low)
        self.high = f32(0.5) / r2 if high is None else f32(high)

    def __call__(self, interval, x):
        x = np.asarray(x, dtype=f32)
        frames = x.reshape(len(x), -1)
        alpha = ONE - expf(-f32(interval) / self.tau)
        oma = ONE - alpha
        sample = np.zeros(len(x), dtype=f32)
        for c in range(frames.shape[1]):
            sample = sample + frames[:, c]
        drive = (sample * sample) * alpha
        avg = np.empty(len(x), dtype=f32)
        a = self.avg_squared
        for i in range(len(x)):
            a = drive[i] + a * oma
            avg[i] = a
        self.avg_squared = a
        with np.errstate(divide="ignore", invalid="ignore"):
            peak = np.sqrt(avg) * np.sqrt(f32(2.0))
            up = np.where(np.isnan(self.low / peak), self.max_gain, np.minimum(self.low / peak, self.max_gain))   # f32::min ignores NaN
            gain = np.where(peak < self.low, up, np.where(peak > self.high, self.high / peak, ONE)).astype(f32)
            return (frames * gain[:, None]).astype(f32).reshape(x.shape)


def tanh_clip(x):
    return tanhf(x)
