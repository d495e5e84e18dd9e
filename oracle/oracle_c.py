"""ctypes binding of the C restatement (oracle/oddio_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package (oddio_amd/).

Class and method names mirror the reference crate (oddio 0.7.4) so the known-answer tests read
like the reference's own `#[test]` functions.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboddio_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "oddio_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, f32, f64, u32, sz, i32 = C.c_void_p, C.c_float, C.c_double, C.c_uint32, C.c_size_t, C.c_int
        fp = C.POINTER(C.c_float)
        sig = {
            "oo_frames_from_slice": (vp, [u32, fp, sz, i32]),
            "oo_frames_retain": (None, [vp]),
            "oo_frames_release": (None, [vp]),
            "oo_frames_signal_new": (vp, [vp, f64]),
            "oo_sine_new": (vp, [f32, f32]),
            "oo_constant_new": (vp, [f32, f32, i32]),
            "oo_cycle_new": (vp, [vp]),
            "oo_fixed_gain_new": (vp, [vp, f32]),
            "oo_gain_new": (vp, [vp]),
            "oo_speed_new": (vp, [vp]),
            "oo_mono_to_stereo_new": (vp, [vp]),
            "oo_reinhard_new": (vp, [vp]),
            "oo_downmix_new": (vp, [vp]),
            "oo_fader_new": (vp, [vp]),
            "oo_fader_fade_to": (None, [vp, vp, f32]),
            "oo_stream_new": (vp, [C.c_uint32, C.c_size_t, C.c_int]),
            "oo_stream_write": (C.c_size_t, [vp, fp, C.c_size_t]),
            "oo_stream_free": (C.c_size_t, [vp]),
            "oo_stream_close": (None, [vp]),
            "oo_adapt_new": (vp, [vp, f32, f32, f32, f32, f32]),
            "oo_constant_set": (None, [vp, f32, f32]),
            "oo_tanh_new": (vp, [vp]),
            "oo_mixer_new": (vp, [i32]),
            "oo_scene_new": (vp, []),
            "oo_counting_new": (vp, [u32]),
            "oo_time_new": (vp, [f32]),
            "oo_finished_new": (vp, []),
            "oo_signal_free": (None, [vp]),
            "oo_channels": (i32, [vp]),
            "oo_sample": (None, [vp, f32, fp, sz]),
            "oo_seek": (None, [vp, f32]),
            "oo_is_finished": (i32, [vp]),
            "oo_is_seek": (i32, [vp]),
            "oo_run": (None, [vp, u32, fp, sz]),
            "oo_gain_set_amplitude_ratio": (None, [vp, f32]),
            "oo_gain_set_gain_db": (None, [vp, f32]),
            "oo_gain_init_amplitude_ratio": (None, [vp, f32]),
            "oo_speed_set": (None, [vp, f32]),
            "oo_frames_signal_t": (f64, [vp]),
            "oo_frames_playback_position": (f64, [vp]),
            "oo_frames_control_is_finished": (i32, [vp]),
            "oo_sine_phase": (f32, [vp]),
            "oo_mixer_play": (i32, [vp, vp]),
            "oo_mixer_stop": (None, [vp, i32]),
            "oo_mixer_is_stopped": (i32, [vp, i32]),
            "oo_mixer_len": (sz, [vp]),
            "oo_scene_play": (i32, [vp, vp, fp, fp, f32]),
            "oo_scene_play_frames_bulk": (None, [vp, sz, u32, fp, sz, sz, C.POINTER(C.c_uint32), C.POINTER(C.c_double), fp, fp, fp]),
            "oo_scene_play_buffered": (i32, [vp, vp, fp, fp, f32, f32, u32, f32]),
            "oo_scene_set_motion": (None, [vp, i32, fp, fp, i32]),
            "oo_scene_is_finished": (i32, [vp, i32]),
            "oo_scene_set_listener_rotation": (None, [vp, fp]),
            "oo_scene_len": (sz, [vp]),
            "oo_scene_len_buffered": (sz, [vp]),
            "oo_scene_sample_f64acc": (None, [vp, f32, C.POINTER(C.c_double), sz]),
            "oo_rotate": (None, [fp, fp, fp]),
            "oo_ear_state": (None, [fp, i32, f32, fp, fp]),
            "oo_ring_new": (vp, [sz]),
            "oo_ring_free": (None, [vp]),
            "oo_ring_write": (None, [vp, vp, u32, f32]),
            "oo_ring_delay": (None, [vp, u32, f32]),
            "oo_ring_sample": (None, [vp, u32, f32, f32, fp, sz]),
            "oo_ring_write_cursor": (f32, [vp]),
            "oo_ring_buffer": (fp, [vp, C.POINTER(sz)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _vec3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(3))


class Frames:
    """oddio::Frames<f32> / Frames<[f32;2]> (src/frames.rs:19-47)."""

    def __init__(self, rate: int, samples):
        a = np.ascontiguousarray(np.asarray(samples, dtype=np.float32))
        self.channels = 1 if a.ndim == 1 else a.shape[1]
        self.len = a.shape[0]
        self.rate = int(rate)
        self._h = lib().oo_frames_from_slice(self.rate, _fp(a), self.len, self.channels)

    @classmethod
    def from_slice(cls, rate, samples):
        return cls(rate, samples)

    def __del__(self):
        try:
            lib().oo_frames_release(self._h)
        except Exception:
            pass


class Signal:
    """Base wrapper; owns its handle unless moved into a parent signal."""

    def __init__(self, handle, inner=None):
        self._h = handle
        self._owned = True
        self._keep = inner
        if inner is not None:
            inner._owned = False

    @property
    def channels(self):
        return lib().oo_channels(self._h)

    def sample(self, interval, out):
        """Signal::sample (src/signal.rs:19).  `out`: float32 array [n] or [n, 2], filled in place."""
        assert out.dtype == np.float32 and out.flags.c_contiguous
        n = out.shape[0]
        lib().oo_sample(self._h, np.float32(interval), _fp(out), n)
        return out

    def sample_n(self, interval, n):
        shape = (n,) if self.channels == 1 else (n, self.channels)
        return self.sample(interval, np.zeros(shape, dtype=np.float32))

    def seek(self, seconds):
        assert lib().oo_is_seek(self._h), "signal does not implement Seek"
        lib().oo_seek(self._h, np.float32(seconds))

    def is_finished(self):
        return bool(lib().oo_is_finished(self._h))

    def __del__(self):
        try:
            if self._owned and self._h:
                lib().oo_signal_free(self._h)
        except Exception:
            pass


def run(signal: Signal, sample_rate: int, out):
    """oddio::run (src/lib.rs:90-93)."""
    lib().oo_run(signal._h, int(sample_rate), _fp(out), out.shape[0])
    return out


class FramesSignal(Signal):
    def __init__(self, frames: Frames, start_seconds: float = 0.0):
        super().__init__(lib().oo_frames_signal_new(frames._h, float(start_seconds)))
        self._frames = frames

    @property
    def t(self):
        return lib().oo_frames_signal_t(self._h)

    def playback_position(self):
        return lib().oo_frames_playback_position(self._h)

    def control_is_finished(self):
        return bool(lib().oo_frames_control_is_finished(self._h))


class Sine(Signal):
    def __init__(self, phase, frequency_hz):
        super().__init__(lib().oo_sine_new(np.float32(phase), np.float32(frequency_hz)))

    @property
    def phase(self):
        return lib().oo_sine_phase(self._h)


class Constant(Signal):
    def __init__(self, value):
        if np.ndim(value) == 0:
            h = lib().oo_constant_new(np.float32(value), np.float32(0), 1)
        else:
            h = lib().oo_constant_new(np.float32(value[0]), np.float32(value[1]), 2)
        super().__init__(h)

    def set(self, value):
        """`constant.0 = value` (the tests mutate the fixture in place, src/adapt.rs:127)."""
        v = np.atleast_1d(np.asarray(value, np.float32))
        lib().oo_constant_set(self._h, v[0], v[-1])


class Cycle(Signal):
    def __init__(self, frames: Frames):
        super().__init__(lib().oo_cycle_new(frames._h))
        self._frames = frames


class FixedGain(Signal):
    def __init__(self, inner: Signal, db):
        super().__init__(lib().oo_fixed_gain_new(inner._h, np.float32(db)), inner)


class Gain(Signal):
    def __init__(self, inner: Signal):
        super().__init__(lib().oo_gain_new(inner._h), inner)

    def set_amplitude_ratio(self, factor):  # GainControl::set_amplitude_ratio
        lib().oo_gain_set_amplitude_ratio(self._h, np.float32(factor))

    def set_gain(self, db):
        lib().oo_gain_set_gain_db(self._h, np.float32(db))

    def init_amplitude_ratio(self, factor):  # Gain::set_amplitude_ratio
        lib().oo_gain_init_amplitude_ratio(self._h, np.float32(factor))


class Speed(Signal):
    def __init__(self, inner: Signal):
        super().__init__(lib().oo_speed_new(inner._h), inner)

    def set_speed(self, factor):
        lib().oo_speed_set(self._h, np.float32(factor))


class MonoToStereo(Signal):
    def __init__(self, inner: Signal):
        super().__init__(lib().oo_mono_to_stereo_new(inner._h), inner)


class Stream(Signal):
    """Stream::new(rate, size) (src/stream.rs:24-34); the object is both halves: the signal and its
    StreamControl (`write`, `free`; `close` = dropping the control)."""

    def __init__(self, rate, size, channels=1):
        super().__init__(lib().oo_stream_new(int(rate), int(size), int(channels)))
        self._channels = channels

    def write(self, samples):
        a = np.ascontiguousarray(np.asarray(samples, dtype=np.float32))
        return int(lib().oo_stream_write(self._h, _fp(a), a.shape[0]))

    def free(self):
        return int(lib().oo_stream_free(self._h))

    def close(self):
        lib().oo_stream_close(self._h)


class Fader(Signal):
    """Fader::new(inner) (src/fader.rs:16-28); the object is also its FaderControl (`fade_to`)."""

    def __init__(self, inner: Signal):
        super().__init__(lib().oo_fader_new(inner._h), inner)
        self._faded = []

    def fade_to(self, signal: Signal, duration):
        signal._owned = False
        self._faded.append(signal)
        lib().oo_fader_fade_to(self._h, signal._h, np.float32(duration))


class Downmix(Signal):
    """Downmix::new(signal) (src/downmix.rs:8-16)."""

    def __init__(self, inner: Signal):
        super().__init__(lib().oo_downmix_new(inner._h), inner)


class Reinhard(Signal):
    def __init__(self, inner: Signal):
        super().__init__(lib().oo_reinhard_new(inner._h), inner)


class Tanh(Signal):
    def __init__(self, inner: Signal):
        super().__init__(lib().oo_tanh_new(inner._h), inner)


class AdaptOptions:
    """src/adapt.rs:36-61 (defaults :52-61)."""

    def __init__(self, tau=0.1, max_gain=np.inf, low=None, high=None):
        r2 = np.sqrt(np.float32(2.0))
        self.tau, self.max_gain = np.float32(tau), np.float32(max_gain)
        self.low = np.float32(0.1) / r2 if low is None else np.float32(low)
        self.high = np.float32(0.5) / r2 if high is None else np.float32(high)


class Adapt(Signal):
    """Adapt::new(signal, initial_rms, options) (src/adapt.rs:25-31)."""

    def __init__(self, inner: Signal, initial_rms, options: AdaptOptions = None):
        o = options or AdaptOptions()
        super().__init__(lib().oo_adapt_new(inner._h, np.float32(initial_rms), o.tau, o.max_gain, o.low, o.high), inner)


class CountingSignal(Signal):
    def __init__(self, start=0):
        super().__init__(lib().oo_counting_new(int(start)))


class TimeSignal(Signal):
    def __init__(self, start):
        super().__init__(lib().oo_time_new(np.float32(start)))


class FinishedSignal(Signal):
    def __init__(self):
        super().__init__(lib().oo_finished_new())


class Mixed:
    def __init__(self, mixer, h):
        self._m, self._i = mixer, h

    def stop(self):
        lib().oo_mixer_stop(self._m._h, self._i)

    def is_stopped(self):
        return bool(lib().oo_mixer_is_stopped(self._m._h, self._i))


class Mixer(Signal):
    """oddio::Mixer<T> (src/mixer.rs); `channels` selects T = f32 or [f32;2]."""

    def __init__(self, channels=2):
        super().__init__(lib().oo_mixer_new(channels))
        self._kids = []

    def play(self, signal: Signal) -> Mixed:
        assert signal.channels == self.channels
        signal._owned = False
        self._kids.append(signal)
        return Mixed(self, lib().oo_mixer_play(self._h, signal._h))

    def __len__(self):
        return lib().oo_mixer_len(self._h)


class SpatialOptions:
    def __init__(self, position=(0.0, 0.0, 0.0), velocity=(0.0, 0.0, 0.0), radius=0.1):
        self.position, self.velocity, self.radius = position, velocity, radius


class Spatial:
    def __init__(self, scene, h):
        self._s, self._i = scene, h

    def set_motion(self, position, velocity, discontinuity):
        lib().oo_scene_set_motion(self._s._h, self._i, _fp(_vec3(position)), _fp(_vec3(velocity)), int(bool(discontinuity)))

    def is_finished(self):
        return bool(lib().oo_scene_is_finished(self._s._h, self._i))


class SpatialScene(Signal):
    """oddio::SpatialScene + SpatialSceneControl (src/spatial.rs)."""

    def __init__(self):
        super().__init__(lib().oo_scene_new())
        self._kids = []

    def play(self, signal: Signal, options: SpatialOptions) -> Spatial:
        assert signal.channels == 1 and lib().oo_is_seek(signal._h)
        signal._owned = False
        self._kids.append(signal)
        h = lib().oo_scene_play(self._h, signal._h, _fp(_vec3(options.position)), _fp(_vec3(options.velocity)), np.float32(options.radius))
        return Spatial(self, h)

    def play_buffered(self, signal: Signal, options: SpatialOptions, max_distance, rate, buffer_duration) -> Spatial:
        assert signal.channels == 1
        signal._owned = False
        self._kids.append(signal)
        h = lib().oo_scene_play_buffered(self._h, signal._h, _fp(_vec3(options.position)), _fp(_vec3(options.velocity)),
                                         np.float32(options.radius), np.float32(max_distance), int(rate), np.float32(buffer_duration))
        return Spatial(self, h)

    def play_frames_bulk(self, rate, clips, start_seconds, positions, velocities, radii, clip_of=None):
        """n x `play(FramesSignal::new(Frames(rate, clips[k]), start[i]), SpatialOptions{..})` from C, k = i or
        clip_of[i].  `clips` is a [*, clip_len] float32 array that is BORROWED (kept alive by this scene)."""
        clips = np.ascontiguousarray(clips, dtype=np.float32)
        clip_len = clips.shape[1]
        n = clips.shape[0] if clip_of is None else len(clip_of)
        cof = None if clip_of is None else np.ascontiguousarray(np.asarray(clip_of, dtype=np.uint32))
        st = np.ascontiguousarray(np.broadcast_to(np.asarray(start_seconds, dtype=np.float64), (n,)))
        pos = np.ascontiguousarray(np.asarray(positions, dtype=np.float32).reshape(n, 3))
        vel = np.ascontiguousarray(np.asarray(velocities, dtype=np.float32).reshape(n, 3))
        rad = np.ascontiguousarray(np.broadcast_to(np.asarray(radii, dtype=np.float32), (n,)))
        self._kids.append(clips)
        lib().oo_scene_play_frames_bulk(self._h, n, int(rate), _fp(clips), clip_len, clip_len,
                                        None if cof is None else cof.ctypes.data_as(C.POINTER(C.c_uint32)),
                                        st.ctypes.data_as(C.POINTER(C.c_double)), _fp(pos), _fp(vel), _fp(rad))

    def set_listener_rotation(self, q_sxyz):
        q = np.ascontiguousarray(np.asarray(q_sxyz, dtype=np.float32).reshape(4))
        lib().oo_scene_set_listener_rotation(self._h, _fp(q))

    def sample_f64acc(self, interval, n):
        out = np.zeros((n, 2), dtype=np.float64)
        lib().oo_scene_sample_f64acc(self._h, np.float32(interval), out.ctypes.data_as(C.POINTER(C.c_double)), n)
        return out

    def __len__(self):
        return lib().oo_scene_len(self._h)

    def len_buffered(self):
        return lib().oo_scene_len_buffered(self._h)


class Ring:
    """oddio::ring::Ring (src/ring.rs:4-80)."""

    def __init__(self, capacity):
        self._h = lib().oo_ring_new(capacity)

    def write(self, signal: Signal, rate, dt):
        lib().oo_ring_write(self._h, signal._h, int(rate), np.float32(dt))

    def delay(self, rate, dt):
        lib().oo_ring_delay(self._h, int(rate), np.float32(dt))

    def sample(self, rate, t, interval, n):
        out = np.zeros(n, dtype=np.float32)
        lib().oo_ring_sample(self._h, int(rate), np.float32(t), np.float32(interval), _fp(out), n)
        return out

    @property
    def write_cursor(self):
        return lib().oo_ring_write_cursor(self._h)

    @property
    def buffer(self):
        n = C.c_size_t()
        p = lib().oo_ring_buffer(self._h, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def __del__(self):
        try:
            lib().oo_ring_free(self._h)
        except Exception:
            pass


class Smoothed:
    """oddio::Smoothed<f32> (src/smooth.rs:26-91)."""

    class _S(C.Structure):
        _fields_ = [("prev", C.c_float), ("next", C.c_float), ("progress", C.c_float)]

    def __init__(self, x):
        L = lib()
        L.oo_smoothed_new.restype = Smoothed._S
        L.oo_smoothed_new.argtypes = [C.c_float]
        L.oo_smoothed_advance.argtypes = [C.POINTER(Smoothed._S), C.c_float]
        L.oo_smoothed_set.argtypes = [C.POINTER(Smoothed._S), C.c_float]
        L.oo_smoothed_get.argtypes = [C.POINTER(Smoothed._S)]
        L.oo_smoothed_get.restype = C.c_float
        self._s = L.oo_smoothed_new(np.float32(x))

    def advance(self, p):
        lib().oo_smoothed_advance(C.byref(self._s), np.float32(p))

    def set(self, v):
        lib().oo_smoothed_set(C.byref(self._s), np.float32(v))

    def get(self):
        return lib().oo_smoothed_get(C.byref(self._s))


def rotate(q_sxyz, p):
    q = np.ascontiguousarray(np.asarray(q_sxyz, dtype=np.float32))
    pp = _vec3(p)
    out = np.zeros(3, dtype=np.float32)
    lib().oo_rotate(_fp(q), _fp(pp), _fp(out))
    return out


def ear_state(pos, ear, radius):
    off, gain = C.c_float(), C.c_float()
    lib().oo_ear_state(_fp(_vec3(pos)), int(ear), np.float32(radius), C.byref(off), C.byref(gain))
    return np.float32(off.value), np.float32(gain.value)
