"""Host-side mirror of the reference's operator interface for the hot path, over the C ABI.

Same names, argument meaning and error behaviour as oddio 0.7.4 (paths relative to the reference
crate root):

    Frames.from_slice / FramesSignal(frames, start)   src/frames.rs:26-47, :156-169
    Sine(phase, hz), Constant(x), FixedGain(sig, db)   src/sine.rs:18-23, src/constant.rs, src/gain.rs:18-23
    SpatialScene() -> (control, scene)                 src/spatial.rs:170-188
    SpatialSceneControl.play / set_listener_rotation   src/spatial.rs:289-302, :345-349
    Spatial.set_motion / is_finished                   src/spatial.rs:137-156
    Mixer() -> (control, mixer), MixerControl.play     src/mixer.rs:70-81, :18-26
    Mixed.stop / is_stopped                            src/mixer.rs:34-43
    MonoToStereo, Reinhard, Tanh                       src/signal.rs:61-91, src/reinhard.rs, src/tanh.rs
    run(signal, sample_rate, out), frame_stereo        src/lib.rs:90-104

Signals here are *descriptions* until they are moved into a scene or mixer (as in the reference,
`play` takes ownership); the per-sample work happens on the GPU inside `sample`.  Filters that the
device path does not implement raise TypeError at `play` -- nothing silently falls back to a CPU.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib

POSTFX_NONE, POSTFX_REINHARD, POSTFX_TANH = 0, 1, 2
MODE_FAST, MODE_ORDERED, MODE_FAST_UNFUSED, MODE_TRACKED = 0, 1, 2, 3


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _vec3(v):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(3))
    return a


class Frames:
    """oddio::Frames<f32> held in HBM (src/frames.rs:19-47).  Reference counted like the Arc."""

    channels = 1

    def __init__(self, handle, rate, length, device, keepalive=None):
        self._h, self.rate, self.len, self.device, self._keepalive = handle, rate, length, device, keepalive

    @classmethod
    def from_slice(cls, rate: int, samples, device: int = 0) -> "Frames":
        a = np.ascontiguousarray(np.asarray(samples, dtype=np.float32))
        h = C.c_void_p()
        if a.ndim == 2 and a.shape[1] == 2:      # Frames<[f32; 2]>: playable in a Mixer only
            _lib.check(_lib.lib().oddio_hip_frames_from_slice_stereo(device, int(rate), _fp(a), a.shape[0], C.byref(h)))
            f = cls(h, int(rate), a.shape[0], device)
            f.channels = 2
            return f
        if a.ndim != 1:
            raise TypeError("clips are mono [n] or interleaved stereo [n, 2]")
        _lib.check(_lib.lib().oddio_hip_frames_from_slice(device, int(rate), _fp(a), a.shape[0], C.byref(h)))
        return cls(h, int(rate), a.shape[0], device)

    @classmethod
    def from_device_ptr(cls, rate: int, dev_ptr: int, length: int, device: int = 0, copy: bool = False, keepalive=None) -> "Frames":
        h = C.c_void_p()
        _lib.check(_lib.lib().oddio_hip_frames_from_device(device, int(rate), C.c_void_p(dev_ptr), int(length), int(copy), C.byref(h)))
        return cls(h, int(rate), int(length), device, keepalive)

    def runtime(self) -> float:  # frames.rs:85-87
        return self.len / float(self.rate)

    def __len__(self):
        return self.len

    def __del__(self):
        try:
            if self._h:
                _lib.lib().oddio_hip_frames_release(self._h)
                self._h = None
        except Exception:
            pass


class Signal:
    channels = 1
    seekable = True


class FramesSignal(Signal):
    def __init__(self, frames: Frames, start_seconds: float = 0.0):
        self.frames, self.start_seconds = frames, float(start_seconds)
        self.channels = frames.channels

    @classmethod
    def new(cls, frames, start_seconds):
        return cls(frames, start_seconds)


class Cycle(Signal):
    """Cycle::new(frames) (src/cycle.rs:17-23): loops a clip end to end.  Seek (src/cycle.rs:56-61):
    playable in a spatial scene through `play` and `play_buffered`, and in Mixer chains."""

    def __init__(self, frames: Frames):
        self.frames = frames
        self.channels = frames.channels


class Sine(Signal):
    def __init__(self, phase: float, frequency_hz: float):
        self.phase, self.frequency_hz = np.float32(phase), np.float32(frequency_hz)


class Constant(Signal):
    def __init__(self, value: float):
        self.value = np.float32(value)


class StreamControl:
    """src/stream.rs:96-114: the producer's half.  `write` is a memcpy into pinned memory plus a
    release store; it never calls into the GPU runtime."""

    def __init__(self, handle, channels):
        self._h, self._channels = handle, channels

    def write(self, samples) -> int:
        a = np.ascontiguousarray(np.asarray(samples, dtype=np.float32))
        if a.ndim != (1 if self._channels == 1 else 2) or (a.ndim == 2 and a.shape[1] != 2):
            raise TypeError("samples must be [n] (mono stream) or [n, 2] (stereo stream)")
        n = C.c_size_t()
        _lib.check(_lib.lib().oddio_hip_stream_write(self._h, _fp(a), a.shape[0], C.byref(n)))
        return int(n.value)

    def free(self) -> int:
        n = C.c_size_t()
        _lib.check(_lib.lib().oddio_hip_stream_free(self._h, C.byref(n)))
        return int(n.value)

    def drop(self):
        """drop(StreamControl): the Stream finishes once it has been drained (src/stream.rs:88-90)."""
        if self._h is not None:
            _lib.check(_lib.lib().oddio_hip_stream_drop(self._h))
            self._h = None

    close = drop


class Stream(Signal):
    """Stream::new(rate, size) -> (StreamControl, Stream)  (src/stream.rs:24-34).  Not `Seek`: enters a
    spatial scene through play_buffered, a Mixer through play."""
    seekable = False

    def __init__(self, control: StreamControl, rate, channels):
        self.control, self.rate, self.channels = control, int(rate), channels

    @classmethod
    def new(cls, rate: int, size: int, channels: int = 1, device: int = 0):
        h = C.c_void_p()
        _lib.check(_lib.lib().oddio_hip_stream_create(device, int(rate), int(size), int(channels), C.byref(h)))
        control = StreamControl(h, channels)
        return control, cls(control, rate, channels)


class FixedGain(Signal):
    def __init__(self, inner: Signal, db: float):
        self.inner, self.db = inner, np.float32(db)
        self.seekable = inner.seekable


class GainControl:
    """src/gain.rs:129-161.  Bound to its source once the signal has been played."""

    def __init__(self):
        self._target = None      # (scene or None, source id, filter index)
        self._ratio = np.float32(1.0)

    def _bind(self, scene, sid, index, handle=None):
        # `handle`: the Mixed / Spatial object of the source -- kept alive by its controls, so that the handle id is
        # released (and may change owner) only once the handle AND every control of the signal are gone, like the Arc
        # a GainControl holds on its signal's shared state in the reference
        self._target = (scene, sid, index)
        self._handle = handle

    def set_amplitude_ratio(self, factor):
        self._ratio = np.float32(factor)
        if self._target is not None:
            owner, sid, index = self._target
            fn = _lib.lib().oddio_hip_mixer_set_gain if isinstance(owner, _MixerSignal) else _lib.lib().oddio_hip_source_set_gain
            _lib.check(fn(owner._h, sid, index, np.float32(factor)))

    def set_gain(self, db):
        self.set_amplitude_ratio(np.float32(_powf10(db)))

    def amplitude_ratio(self):
        """GainControl::amplitude_ratio (src/gain.rs:147-150); through the C ABI once the signal has been played."""
        if self._target is not None:
            owner, sid, index = self._target
            fn = _lib.lib().oddio_hip_mixer_get_amplitude_ratio if isinstance(owner, _MixerSignal) else _lib.lib().oddio_hip_source_get_amplitude_ratio
            v = C.c_float()
            _lib.check(fn(owner._h, sid, index, C.byref(v)))
            return float(v.value)
        return float(self._ratio)

    def gain(self):
        """GainControl::gain (src/gain.rs:133-135): 20 * log10(amplitude_ratio), in decibels."""
        if self._target is not None:
            owner, sid, index = self._target
            fn = _lib.lib().oddio_hip_mixer_get_gain_db if isinstance(owner, _MixerSignal) else _lib.lib().oddio_hip_source_get_gain_db
            v = C.c_float()
            _lib.check(fn(owner._h, sid, index, C.byref(v)))
            return float(v.value)
        return float(np.float32(20.0) * np.log10(self._ratio, dtype=np.float32))


class SpeedControl:
    """src/speed.rs:42-55"""

    def __init__(self):
        self._target = None
        self._speed = np.float32(1.0)

    def _bind(self, scene, sid, index, handle=None):
        self._target = (scene, sid, index)
        self._handle = handle          # see GainControl._bind

    def set_speed(self, factor):
        self._speed = np.float32(factor)
        if self._target is not None:
            owner, sid, index = self._target
            fn = _lib.lib().oddio_hip_mixer_set_speed if isinstance(owner, _MixerSignal) else _lib.lib().oddio_hip_source_set_speed
            _lib.check(fn(owner._h, sid, index, np.float32(factor)))

    def speed(self):
        """SpeedControl::speed (src/speed.rs:47-49); through the C ABI once the signal has been played."""
        if self._target is not None:
            owner, sid, index = self._target
            fn = _lib.lib().oddio_hip_mixer_get_speed if isinstance(owner, _MixerSignal) else _lib.lib().oddio_hip_source_get_speed
            v = C.c_float()
            _lib.check(fn(owner._h, sid, index, C.byref(v)))
            return float(v.value)
        return float(self._speed)


def _powf10(db):
    """10.0f32.powf(db / 20.0) through libm's powf, as Rust's std does (src/gain.rs:143)."""
    import ctypes.util
    libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    libm.powf.restype = C.c_float
    libm.powf.argtypes = [C.c_float, C.c_float]
    return libm.powf(10.0, float(np.float32(db) / np.float32(20.0)))


class Gain(Signal):
    """Gain::new(signal) -> (GainControl, Gain)  (src/gain.rs:66-74).  Not `Seek`: play_buffered only."""
    seekable = False

    def __init__(self, inner: Signal):
        self.inner = inner
        self.control = GainControl()

    @classmethod
    def new(cls, inner):
        g = cls(inner)
        return g.control, g


class Speed(Signal):
    """Speed::new(signal) -> (SpeedControl, Speed)  (src/speed.rs:16-24).  Not `Seek`: play_buffered only."""
    seekable = False

    def __init__(self, inner: Signal):
        self.inner = inner
        self.control = SpeedControl()

    @classmethod
    def new(cls, inner):
        sp = cls(inner)
        return sp.control, sp


FILTER_FIXED_GAIN, FILTER_GAIN, FILTER_SPEED, FILTER_REINHARD, FILTER_TANH = 1, 2, 3, 4, 5


def _leaf_args(leaf, keep):
    """-> (leaf_kind, frames handle, start_seconds, phase, frequency_hz_or_value)"""
    if isinstance(leaf, Downmix):            # (oddio_hip_scene_play_filtered only)
        keep.append(leaf.inner.frames)
        return (4, leaf.inner.frames._h, leaf.inner.start_seconds, 0.0, 0.0)
    if isinstance(leaf, FramesSignal):
        keep.append(leaf.frames)
        return (0, leaf.frames._h, leaf.start_seconds, 0.0, 0.0)
    if isinstance(leaf, Cycle):
        keep.append(leaf.frames)
        return (3, leaf.frames._h, 0.0, 0.0, 0.0)
    if isinstance(leaf, Sine):
        return (1, None, 0.0, float(leaf.phase), float(leaf.frequency_hz))
    return (2, None, 0.0, 0.0, float(leaf.value))


class _Filter(C.Structure):
    _fields_ = [("kind", C.c_int), ("param", C.c_float)]


def _unwrap_chain(signal):
    """-> (leaf, [(kind, param, control or None)] innermost first)"""
    chain = []
    while isinstance(signal, (FixedGain, Gain, Speed, Reinhard)) and not _is_postfx(signal):
        if isinstance(signal, FixedGain):
            chain.append((FILTER_FIXED_GAIN, float(signal.db), None))
        elif isinstance(signal, Reinhard):       # (Tanh is a subclass)
            chain.append((signal._FILTER, 0.0, None))
        elif isinstance(signal, Gain):
            chain.append((FILTER_GAIN, float(signal.control._ratio), signal.control))
        else:
            chain.append((FILTER_SPEED, float(signal.control._speed), signal.control))
        signal = signal.inner
    if not isinstance(signal, (FramesSignal, Sine, Constant, Cycle, Stream)):
        raise TypeError(f"{type(signal).__name__} is not implemented on the device path")
    if len(chain) > 4:
        raise TypeError("at most 4 filters around a buffered source")
    return signal, chain[::-1]


class Downmix(Signal):
    """Downmix::new(signal) (src/downmix.rs:8-16): sums the channels.  Device support: a stereo
    FramesSignal played in a spatial scene (`play(Downmix(FramesSignal(stereo_frames)), ..)`)."""
    channels = 1

    def __init__(self, inner: Signal):
        if not isinstance(inner, FramesSignal) or inner.frames.channels != 2:
            raise TypeError("the device Downmix wraps a FramesSignal over a stereo clip")
        self.inner = inner
        self.seekable = True


class MonoToStereo(Signal):
    channels = 2

    def __init__(self, inner: Signal):
        if inner.channels != 1:
            raise TypeError("MonoToStereo takes a mono signal (src/signal.rs:70)")
        self.inner = inner


def _is_postfx(signal):
    """Reinhard / Tanh around a whole SpatialScene / Mixer (the reduce kernel's epilogue), as opposed to around a source."""
    return isinstance(signal, Reinhard) and signal._postfx


def _unwrap(signal):
    """-> (leaf, fixed_gain_db or NaN, [(filter kind, param)] innermost first when the nest needs play_filtered, else None)

    A Seek chain: FixedGain / Reinhard / Tanh in any order and multiplicity, up to 4 (src/gain.rs:39-51, src/reinhard.rs:42-50,
    src/tanh.rs:36-44 are `impl<T: Seek> Seek` for the wrappers)."""
    db = math.nan
    chain = []
    while isinstance(signal, (FixedGain, Reinhard)) and not _is_postfx(signal):
        if isinstance(signal, FixedGain):
            chain.append((FILTER_FIXED_GAIN, float(signal.db)))
        else:
            chain.append((signal._FILTER, 0.0))
        signal = signal.inner
    chain = chain[::-1]
    kinds = [k for k, _ in chain]
    if isinstance(signal, (Gain, Speed)):
        raise TypeError("Gain / Speed are not Seek (src/gain.rs:53-57, src/speed.rs): use play_buffered")
    if len(chain) > 4:
        raise TypeError("at most 4 wrappers around a played source on the device path")
    if not isinstance(signal, (FramesSignal, Sine, Constant, Cycle, Downmix)):
        raise TypeError(f"{type(signal).__name__} is not implemented on the device path "
                        "(supported: FramesSignal, Downmix of a stereo FramesSignal, Cycle, Sine, Constant; FixedGain, Reinhard, Tanh around them)")
    if kinds in ([], [FILTER_FIXED_GAIN]) and not (kinds and isinstance(signal, Constant)):
        if kinds:
            db = chain[0][1]
        return signal, db, None
    # (FixedGain around a Constant goes through play_filtered as well: oddio_hip_scene_play_constant has no gain argument)
    return signal, db, chain


class SpatialOptions:
    """src/spatial.rs:354-371 (default radius 0.1)."""

    def __init__(self, position=(0.0, 0.0, 0.0), velocity=(0.0, 0.0, 0.0), radius=0.1):
        self.position, self.velocity, self.radius = position, velocity, radius


class Spatial:
    """Handle returned by SpatialSceneControl.play (src/spatial.rs:119-157)."""

    def __init__(self, scene, sid):
        self._scene, self.id = scene, sid

    def set_motion(self, position, velocity, discontinuity: bool):
        _lib.check(_lib.lib().oddio_hip_source_set_motion(self._scene._h, self.id, _fp(_vec3(position)), _fp(_vec3(velocity)), int(bool(discontinuity))))

    def is_finished(self) -> bool:
        out = C.c_int()
        _lib.check(_lib.lib().oddio_hip_source_is_finished(self._scene._h, self.id, C.byref(out)))
        return bool(out.value)

    def playback_position(self) -> float:
        out = C.c_double()
        _lib.check(_lib.lib().oddio_hip_source_playback_position(self._scene._h, self.id, C.byref(out)))
        return out.value

    def release(self):
        """Dropping the `Spatial` handle: the source keeps playing; its id may be reused once it is removed."""
        _lib.check(_lib.lib().oddio_hip_source_release(self._scene._h, self.id))


class _SceneSignal(Signal):
    """The `SpatialScene` half: implements Signal<Frame = [f32; 2]> (src/spatial.rs:373-477)."""
    channels = 2
    seekable = False

    def __init__(self, device, max_sources, max_frames):
        self._h = C.c_void_p()
        self.device = device
        self.max_frames = max_frames
        _lib.check(_lib.lib().oddio_hip_scene_create(device, int(max_sources), int(max_frames), C.byref(self._h)))
        self._keep = []

    def sample(self, interval, out: np.ndarray):
        assert out.dtype == np.float32 and out.flags.c_contiguous and (out.ndim == 2 and out.shape[1] == 2 or out.shape[0] == 0)
        _lib.check(_lib.lib().oddio_hip_scene_sample(self._h, np.float32(interval), _fp(out), out.shape[0]))
        return out

    def sample_n(self, interval, n):
        return self.sample(interval, np.zeros((n, 2), dtype=np.float32))

    def sample_device(self, interval, dev_ptr: int, n_frames: int):
        _lib.check(_lib.lib().oddio_hip_scene_sample_device(self._h, np.float32(interval), C.c_void_p(dev_ptr), int(n_frames)))

    def synchronize(self):
        _lib.check(_lib.lib().oddio_hip_scene_synchronize(self._h))

    def stream(self) -> int:
        p = C.c_void_p()
        _lib.check(_lib.lib().oddio_hip_scene_stream(self._h, C.byref(p)))
        return p.value or 0

    def set_stream(self, hip_stream: int):
        _lib.check(_lib.lib().oddio_hip_scene_set_stream(self._h, C.c_void_p(hip_stream)))

    def seek_all(self, seconds):
        _lib.check(_lib.lib().oddio_hip_scene_seek_all(self._h, np.float32(seconds)))

    def reduce_init(self, rank: int, world: int, unique_id: bytes):
        """Join the stereo-buffer all-reduce of a scene sharded over `world` GPUs: from now on every
        sample call sums the ranks' partial buffers over RCCL before the post-mix filter."""
        buf = C.create_string_buffer(bytes(unique_id), len(unique_id))
        _lib.check(_lib.lib().oddio_hip_scene_reduce_init(self._h, int(rank), int(world), buf, len(unique_id)))

    def reduce_init_p2p(self, rank: int, world: int, handle: bytes | None = None) -> bytes:
        """Join the deterministic peer-to-peer reduce (include/oddio_hip.h): rank 0 passes no handle and gets the one
        to hand to the other ranks; they pass rank 0's.  One process per rank."""
        buf = C.create_string_buffer(P2P_HANDLE_BYTES)
        if rank != 0:
            if handle is None or len(handle) != P2P_HANDLE_BYTES:
                raise ValueError("ranks > 0 need rank 0's handle")
            buf.raw = handle
        _lib.check(_lib.lib().oddio_hip_scene_reduce_init_p2p(self._h, int(rank), int(world), buf, P2P_HANDLE_BYTES))
        return bytes(buf.raw)

    def reduce_destroy(self):
        _lib.check(_lib.lib().oddio_hip_scene_reduce_destroy(self._h))

    def is_finished(self):
        return False  # spatial.rs:473-476

    def reserve_buffered(self, max_buffered: int):
        """Capacity of the buffered set (default 256); call before the first play_buffered."""
        _lib.check(_lib.lib().oddio_hip_scene_reserve_buffered(self._h, int(max_buffered)))

    def debug_buffered_slow(self) -> int:
        """(tests) buffered sources the last callback left to the general kernel."""
        n = C.c_uint32()
        _lib.check(_lib.lib().oddio_hip_debug_buffered_slow(self._h, C.byref(n)))
        return n.value

    def debug_reset_buffered_clock(self, seconds: float):
        """(bench) FramesSignal::t = seconds for every buffered FramesSignal leaf, in stream order."""
        _lib.check(_lib.lib().oddio_hip_debug_reset_buffered_clock(self._h, float(seconds)))

    def reduce_info(self) -> dict:
        """The reduce group as the library sees it: {"kind": "none" | "rccl" | "p2p", "world", "rccl_version", "rccl_lib"}."""
        kind, world, ver = C.c_int(), C.c_int(), C.c_int()
        buf = C.create_string_buffer(512)
        _lib.check(_lib.lib().oddio_hip_scene_reduce_info(self._h, C.byref(kind), C.byref(world), C.byref(ver), buf, 512))
        return {"kind": ("none", "rccl", "p2p")[kind.value], "world": world.value, "rccl_version": ver.value, "rccl_lib": buf.value.decode() or None}

    def set_buffered_fast(self, enable: bool):
        """Which kernels render the buffered set (identical results): the batched path (default) or the general kernel for everything."""
        _lib.check(_lib.lib().oddio_hip_scene_set_buffered_fast(self._h, int(bool(enable))))

    def set_postfx(self, kind):
        _lib.check(_lib.lib().oddio_hip_scene_set_postfx(self._h, int(kind)))

    def set_adapt(self, enable, initial_rms=0.0, options=None):
        o = options or AdaptOptions()
        _lib.check(_lib.lib().oddio_hip_scene_set_adapt(self._h, int(bool(enable)), np.float32(initial_rms), o.tau, o.max_gain, o.low, o.high))

    def set_mode(self, mode):
        _lib.check(_lib.lib().oddio_hip_scene_set_mode(self._h, int(mode)))

    def set_exact_updates(self, wait_for_staging: bool):
        """Un-synchronised `sample_device` callers: wait for a staging slot rather than let updates slip a callback."""
        _lib.check(_lib.lib().oddio_hip_scene_set_exact_updates(self._h, int(bool(wait_for_staging))))

    def set_profiling(self, on):
        """False/0 off, True/1 events around every stage, 2 events around the mix kernel only, 1 + k (k >= 2): around the
        mix kernel of every k-th call."""
        _lib.check(_lib.lib().oddio_hip_scene_set_profiling(self._h, int(on) if int(on) >= 2 else int(bool(on))))

    def last_kernel_ms(self):
        ms = (C.c_float * 3)()
        _lib.check(_lib.lib().oddio_hip_scene_last_kernel_ms(self._h, ms))
        return [ms[0], ms[1], ms[2]]

    def kernel_ms_history(self, max_calls=512):
        """[n, 3] milliseconds (prepass, mix, reduce+postfx) of the most recent profiled calls."""
        buf = np.zeros((max_calls, 3), dtype=np.float32)
        n = C.c_size_t()
        _lib.check(_lib.lib().oddio_hip_scene_kernel_ms_history(self._h, _fp(buf), max_calls, C.byref(n)))
        return buf[:n.value].copy()

    def buffered_ms_history(self, max_calls=512):
        """[n, 3] milliseconds (walk, ring write, ring reads + sum) of the buffered set's stages in the most recent profiled calls."""
        buf = np.zeros((max_calls, 3), dtype=np.float32)
        n = C.c_size_t()
        _lib.check(_lib.lib().oddio_hip_scene_buffered_ms_history(self._h, _fp(buf), max_calls, C.byref(n)))
        return buf[:n.value].copy()

    def __len__(self):
        n = C.c_size_t()
        _lib.check(_lib.lib().oddio_hip_scene_len(self._h, C.byref(n)))
        return n.value

    def len_buffered(self):
        n = C.c_size_t()
        _lib.check(_lib.lib().oddio_hip_scene_len_buffered(self._h, C.byref(n)))
        return n.value

    def close(self):
        if self._h:
            _lib.lib().oddio_hip_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SpatialSceneControl:
    """src/spatial.rs:267-350"""

    def __init__(self, scene: _SceneSignal):
        self._scene = scene

    def play(self, signal: Signal, options: SpatialOptions) -> Spatial:
        if signal.channels != 1:
            raise TypeError("signals in a spatial scene must be single-channel (src/spatial.rs:278-279)")
        leaf, db, chain = _unwrap(signal)
        L, s = _lib.lib(), self._scene
        sid = C.c_uint32()
        pos, vel = _vec3(options.position), _vec3(options.velocity)
        if chain is not None:      # a per-source Reinhard / Tanh (with or without a FixedGain on either side of it)
            if isinstance(leaf, Cycle) and leaf.frames.channels != 1:
                raise TypeError("signals in a spatial scene must be single-channel (src/spatial.rs:278-279)")
            filt = (_Filter * len(chain))()
            for i, (kind, param) in enumerate(chain):
                filt[i].kind, filt[i].param = kind, param
            args = _leaf_args(leaf, s._keep)
            _lib.check(L.oddio_hip_scene_play_filtered(s._h, args[0], args[1], args[2], args[3], args[4], C.cast(filt, C.c_void_p), len(chain),
                                                       _fp(pos), _fp(vel), np.float32(options.radius), C.byref(sid)))
        elif isinstance(leaf, Downmix):
            s._keep.append(leaf.inner.frames)
            _lib.check(L.oddio_hip_scene_play_frames_downmix(s._h, leaf.inner.frames._h, leaf.inner.start_seconds, db, _fp(pos), _fp(vel),
                                                             np.float32(options.radius), C.byref(sid)))
        elif isinstance(leaf, FramesSignal):
            s._keep.append(leaf.frames)
            _lib.check(L.oddio_hip_scene_play_frames(s._h, leaf.frames._h, leaf.start_seconds, db, _fp(pos), _fp(vel), np.float32(options.radius), C.byref(sid)))
        elif isinstance(leaf, Sine):
            _lib.check(L.oddio_hip_scene_play_sine(s._h, leaf.phase, leaf.frequency_hz, db, _fp(pos), _fp(vel), np.float32(options.radius), C.byref(sid)))
        elif isinstance(leaf, Cycle):
            if leaf.frames.channels != 1:
                raise TypeError("signals in a spatial scene must be single-channel (src/spatial.rs:278-279)")
            s._keep.append(leaf.frames)
            _lib.check(L.oddio_hip_scene_play_cycle(s._h, leaf.frames._h, db, _fp(pos), _fp(vel), np.float32(options.radius), C.byref(sid)))
        else:
            _lib.check(L.oddio_hip_scene_play_constant(s._h, leaf.value, _fp(pos), _fp(vel), np.float32(options.radius), C.byref(sid)))
        return Spatial(s, sid.value)

    def play_buffered(self, signal: Signal, options: SpatialOptions, max_distance: float, rate: int, buffer_duration: float) -> Spatial:
        """SpatialSceneControl::play_buffered (src/spatial.rs:314-340)."""
        if signal.channels != 1:
            raise TypeError("signals in a spatial scene must be single-channel")
        fader = signal if isinstance(signal, Fader) else None
        leaf, chain = _unwrap_chain(signal.inner if fader else signal)
        L, s = _lib.lib(), self._scene
        sid = C.c_uint32()
        pos, vel = _vec3(options.position), _vec3(options.velocity)
        filt = (_Filter * max(len(chain), 1))()
        for i, (kind, param, _) in enumerate(chain):
            filt[i].kind, filt[i].param = kind, param
        if fader is not None:
            if isinstance(leaf, Stream):
                raise TypeError("Fader<..Stream..> is not implemented on the device path")
            args = _leaf_args(leaf, s._keep)
            _lib.check(L.oddio_hip_scene_play_buffered_fader(s._h, args[0], args[1], args[2], args[3], args[4], C.cast(filt, C.c_void_p), len(chain),
                                                             _fp(pos), _fp(vel), np.float32(options.radius), np.float32(max_distance), int(rate),
                                                             np.float32(buffer_duration), C.byref(sid)))
            fader.control._bind_scene(s, sid.value)
        elif isinstance(leaf, Stream):
            if leaf.control._h is None:
                raise ValueError("the StreamControl has already been dropped")
            _lib.check(L.oddio_hip_scene_play_buffered_stream(s._h, leaf.control._h, C.cast(filt, C.c_void_p), len(chain), _fp(pos), _fp(vel),
                                                              np.float32(options.radius), np.float32(max_distance), int(rate),
                                                              np.float32(buffer_duration), C.byref(sid)))
        else:
            args = _leaf_args(leaf, s._keep)
            _lib.check(L.oddio_hip_scene_play_buffered(s._h, args[0], args[1], args[2], args[3], args[4], C.cast(filt, C.c_void_p), len(chain),
                                                       _fp(pos), _fp(vel), np.float32(options.radius), np.float32(max_distance), int(rate),
                                                       np.float32(buffer_duration), C.byref(sid)))
        spatial = Spatial(s, sid.value)
        for i, (_, _, control) in enumerate(chain):
            if control is not None:
                control._bind(s, sid.value, i, spatial)
        return spatial

    def play_frames_batch(self, frames_list, start_seconds, positions, velocities, radii, fixed_gain_db=None):
        """Bulk `play(FramesSignal::new(frames[i], start[i]), SpatialOptions{..})` for large scenes."""
        n = len(frames_list)
        s = self._scene
        s._keep.extend(frames_list)
        arr = (C.c_void_p * n)(*[f._h.value for f in frames_list])
        st = np.ascontiguousarray(np.asarray(start_seconds, dtype=np.float64).reshape(n))
        pos = np.ascontiguousarray(np.asarray(positions, dtype=np.float32).reshape(n, 3))
        vel = np.ascontiguousarray(np.asarray(velocities, dtype=np.float32).reshape(n, 3))
        rad = np.ascontiguousarray(np.asarray(radii, dtype=np.float32).reshape(n))
        ids = np.zeros(n, dtype=np.uint32)
        fg = None
        if fixed_gain_db is not None:
            fg = np.ascontiguousarray(np.asarray(fixed_gain_db, dtype=np.float32).reshape(n))
        _lib.check(_lib.lib().oddio_hip_scene_play_frames_batch(
            s._h, n, arr, st.ctypes.data_as(C.POINTER(C.c_double)), _fp(fg) if fg is not None else None,
            _fp(pos), _fp(vel), _fp(rad), ids.ctypes.data_as(C.POINTER(C.c_uint32))))
        return [Spatial(s, int(i)) for i in ids]

    def play_buffered_frames_batch(self, frames_list, start_seconds, filter_kinds, filter_params, positions, velocities, radii,
                                   max_distance: float, rate: int, buffer_duration: float):
        """Bulk `play_buffered(filters(FramesSignal::new(frames[i], start[i])), ..)` (src/spatial.rs:314-340) for large scenes:
        `filter_kinds` (innermost first: FILTER_FIXED_GAIN / FILTER_GAIN / FILTER_SPEED) is shared, `filter_params` is [n][len(kinds)].
        Returns the handle ids (np.uint32); controls are addressed with set_control_batch(ids, filter_index, values)."""
        n = len(frames_list)
        s = self._scene
        s._keep.extend(frames_list)
        arr = (C.c_void_p * n)(*[f._h.value for f in frames_list])
        st = np.ascontiguousarray(np.asarray(start_seconds, dtype=np.float64).reshape(n))
        kinds = np.ascontiguousarray(np.asarray(filter_kinds, dtype=np.int32).reshape(-1))
        nf = len(kinds)
        par = np.ascontiguousarray(np.asarray(filter_params, dtype=np.float32).reshape(n, nf)) if nf else np.zeros((n, 1), np.float32)
        pos = np.ascontiguousarray(np.asarray(positions, dtype=np.float32).reshape(n, 3))
        vel = np.ascontiguousarray(np.asarray(velocities, dtype=np.float32).reshape(n, 3))
        rad = np.ascontiguousarray(np.asarray(radii, dtype=np.float32).reshape(n))
        ids = np.zeros(n, dtype=np.uint32)
        _lib.check(_lib.lib().oddio_hip_scene_play_buffered_batch(
            s._h, n, arr, st.ctypes.data_as(C.POINTER(C.c_double)), kinds.ctypes.data_as(C.POINTER(C.c_int32)), nf, _fp(par),
            _fp(pos), _fp(vel), _fp(rad), np.float32(max_distance), int(rate), np.float32(buffer_duration),
            ids.ctypes.data_as(C.POINTER(C.c_uint32))))
        return ids

    def set_control_batch(self, ids, filter_index: int, values):
        """n GainControl::set_amplitude_ratio / SpeedControl::set_speed stores under one lock."""
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.uint32))
        vals = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(len(ids)))
        _lib.check(_lib.lib().oddio_hip_scene_set_control_batch(self._scene._h, len(ids), ids.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                                 int(filter_index), _fp(vals)))

    def set_control_device(self, n: int, d_ids_ptr: int, filter_index: int, d_values_ptr: int):
        """n gain / speed stores with ids (uint32) and values (float32) in device memory; the arrays must outlive the next sample call's execution."""
        _lib.check(_lib.lib().oddio_hip_scene_set_control_device(self._scene._h, int(n), C.c_void_p(d_ids_ptr), int(filter_index), C.c_void_p(d_values_ptr)))

    def set_motion_device(self, n: int, d_ids_ptr: int, d_positions_ptr: int, d_velocities_ptr: int, discontinuity: bool):
        """n set_motion calls with ids, positions [n][3], velocities [n][3] in device memory; same lifetime rule."""
        _lib.check(_lib.lib().oddio_hip_scene_set_motion_device(self._scene._h, int(n), C.c_void_p(d_ids_ptr), C.c_void_p(d_positions_ptr),
                                                                C.c_void_p(d_velocities_ptr), int(bool(discontinuity))))

    def set_motion_batch(self, handles, positions, velocities, discontinuity: bool):
        if isinstance(handles, np.ndarray):
            ids = np.ascontiguousarray(handles.astype(np.uint32, copy=False))
        else:
            ids = np.ascontiguousarray(np.array([h.id for h in handles], dtype=np.uint32))
        n = len(ids)
        pos = np.ascontiguousarray(np.asarray(positions, dtype=np.float32).reshape(n, 3))
        vel = np.ascontiguousarray(np.asarray(velocities, dtype=np.float32).reshape(n, 3))
        _lib.check(_lib.lib().oddio_hip_scene_set_motion_batch(
            self._scene._h, n, ids.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(pos), _fp(vel), int(bool(discontinuity))))

    def set_listener_rotation(self, rotation_sxyz):
        q = np.ascontiguousarray(np.asarray(rotation_sxyz, dtype=np.float32).reshape(4))
        _lib.check(_lib.lib().oddio_hip_scene_set_listener_rotation(self._scene._h, _fp(q)))


UNIQUE_ID_BYTES = 128
P2P_HANDLE_BYTES = 128


def reduce_unique_id() -> bytes:
    """`ncclGetUniqueId` for a sharded scene's reduce group: rank 0 makes it, every rank passes it to
    `scene.reduce_init` (hand it over with whatever the host program uses: a file, MPI, torch.distributed)."""
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    _lib.check(_lib.lib().oddio_hip_reduce_unique_id(buf, UNIQUE_ID_BYTES))
    return buf.raw


def SpatialScene(device: int = 0, max_sources: int = 4096, max_frames: int = 4096):
    """SpatialScene::new() -> (SpatialSceneControl, SpatialScene)  (src/spatial.rs:170-188)."""
    scene = _SceneSignal(device, max_sources, max_frames)
    return SpatialSceneControl(scene), scene


class AdaptOptions:
    """src/adapt.rs:36-61 (defaults :52-61)."""

    def __init__(self, tau=0.1, max_gain=np.inf, low=None, high=None):
        r2 = np.sqrt(np.float32(2.0))
        self.tau, self.max_gain = np.float32(tau), np.float32(max_gain)
        self.low = np.float32(0.1) / r2 if low is None else np.float32(low)
        self.high = np.float32(0.5) / r2 if high is None else np.float32(high)


class Adapt(Signal):
    """Adapt::new(scene_or_mixer, initial_rms, options) (src/adapt.rs:25-31): device epilogue over
    the stereo sum (one lane walks the avg_squared recurrence, the block does the rest)."""
    channels = 2
    seekable = False

    def __init__(self, inner, initial_rms, options: AdaptOptions = None):
        if not isinstance(inner, (_SceneSignal, _MixerSignal)):
            raise TypeError("the device Adapt wraps a SpatialScene or a Mixer directly (Reinhard/Tanh go outside it)")
        o = options or AdaptOptions()
        self.inner = inner
        inner.set_adapt(True, initial_rms, o)

    def sample(self, interval, out):
        return self.inner.sample(interval, out)

    def sample_n(self, interval, n):
        return self.inner.sample_n(interval, n)

    def is_finished(self):
        return self.inner.is_finished()


class Reinhard(Signal):
    """Reinhard::new(signal) (src/reinhard.rs:22-50).  Around a SpatialScene, a Mixer or an Adapt of one: fused into
    the reduce kernel's epilogue (or into the Adapt epilogue).  Around a source signal: a per-source soft clip --
    `x / (1 + |x|)` on every sample of that source before it is mixed -- playable wherever the wrapped signal is (it is
    Seek when the inner signal is, src/reinhard.rs:42-50): `play` for [FixedGain] [Reinhard] chains over a leaf,
    `play_buffered` and Mixer chains anywhere among the filters."""
    channels = 2
    seekable = False
    _KIND = POSTFX_REINHARD
    _FILTER = FILTER_REINHARD

    def __init__(self, inner):
        target = inner.inner if isinstance(inner, Adapt) else inner
        self._postfx = isinstance(target, (_SceneSignal, _MixerSignal))
        self.inner = inner
        if self._postfx:
            target.set_postfx(self._KIND)
        elif isinstance(inner, Signal):
            self.channels, self.seekable = inner.channels, inner.seekable
        else:
            raise TypeError("Reinhard / Tanh wrap a SpatialScene, a Mixer, an Adapt around one, or a source signal")

    def sample(self, interval, out):
        return self.inner.sample(interval, out)

    def sample_n(self, interval, n):
        return self.inner.sample_n(interval, n)

    def is_finished(self):
        return self.inner.is_finished()


class Tanh(Reinhard):
    """Tanh::new(signal) (src/tanh.rs:16-44): like Reinhard, for a whole scene / mixer or per source."""
    _KIND = POSTFX_TANH
    _FILTER = FILTER_TANH


class Mixed:
    """src/mixer.rs:30-44"""

    def __init__(self, mixer, sid):
        self._m, self.id = mixer, sid

    def stop(self):
        _lib.check(_lib.lib().oddio_hip_mixer_stop(self._m._h, self.id))

    def is_stopped(self) -> bool:
        out = C.c_int()
        _lib.check(_lib.lib().oddio_hip_mixer_is_stopped(self._m._h, self.id, C.byref(out)))
        return bool(out.value)

    def __del__(self):      # drop(Mixed): the library may recycle the handle id once the source is gone
        try:
            if self._m._h:
                _lib.lib().oddio_hip_mixer_source_release(self._m._h, self.id)
        except Exception:
            pass


class _MixerSignal(Signal):
    seekable = False

    def __init__(self, device, max_sources, max_frames, channels=2):
        assert channels in (1, 2), "Mixer<f32> or Mixer<[f32; 2]>"
        self.channels = channels
        self._h = C.c_void_p()
        create = _lib.lib().oddio_hip_mixer_create if channels == 2 else _lib.lib().oddio_hip_mixer_create_mono
        _lib.check(create(device, int(max_sources), int(max_frames), C.byref(self._h)))
        self._keep = []

    def sample(self, interval, out):
        assert out.dtype == np.float32 and out.flags.c_contiguous
        _lib.check(_lib.lib().oddio_hip_mixer_sample(self._h, np.float32(interval), _fp(out), out.shape[0]))
        return out

    def sample_n(self, interval, n):
        return self.sample(interval, np.zeros((n, 2) if self.channels == 2 else (n,), dtype=np.float32))

    def sample_device(self, interval, dev_ptr: int, n_frames: int):
        """Mixer::sample with the frames left in device memory (channels * n_frames floats at `dev_ptr`) and no wait
        (oddio_hip_mixer_sample_device); `synchronize()` waits for the mixer's stream."""
        _lib.check(_lib.lib().oddio_hip_mixer_sample_device(self._h, np.float32(interval), C.c_void_p(dev_ptr), int(n_frames)))

    def synchronize(self):
        _lib.check(_lib.lib().oddio_hip_mixer_synchronize(self._h))

    def is_finished(self):
        return False

    def set_postfx(self, kind):
        _lib.check(_lib.lib().oddio_hip_mixer_set_postfx(self._h, int(kind)))

    def set_adapt(self, enable, initial_rms=0.0, options=None):
        o = options or AdaptOptions()
        _lib.check(_lib.lib().oddio_hip_mixer_set_adapt(self._h, int(bool(enable)), np.float32(initial_rms), o.tau, o.max_gain, o.low, o.high))

    def set_mode(self, mode):
        _lib.check(_lib.lib().oddio_hip_mixer_set_mode(self._h, int(mode)))

    def __len__(self):
        n = C.c_size_t()
        _lib.check(_lib.lib().oddio_hip_mixer_len(self._h, C.byref(n)))
        return n.value

    def close(self):
        if self._h:
            _lib.lib().oddio_hip_mixer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MixerControl:
    """src/mixer.rs:7-27 for Mixer<[f32;2]> and Mixer<f32>"""

    def __init__(self, mixer):
        self._m = mixer

    def _parse(self, signal):
        """-> (leaf, [(kind, param, control)] innermost first) for a nest whose output is the mixer's frame type"""
        chain, stereo_seen, sig = [], False, signal
        while isinstance(sig, (FixedGain, Gain, Speed, MonoToStereo, Reinhard)) and not _is_postfx(sig):
            if isinstance(sig, MonoToStereo):
                if stereo_seen:
                    raise TypeError("MonoToStereo appears twice")
                stereo_seen = True
            elif isinstance(sig, FixedGain):
                chain.append((FILTER_FIXED_GAIN, float(sig.db), None))
            elif isinstance(sig, Reinhard):      # per-source soft clip (per channel: reinhard.rs:28-35, tanh.rs:22-29)
                chain.append((sig._FILTER, 0.0, None))
            elif isinstance(sig, Gain):
                chain.append((FILTER_GAIN, float(sig.control._ratio), sig.control))
            else:
                chain.append((FILTER_SPEED, float(sig.control._speed), sig.control))
            sig = sig.inner
        if not isinstance(sig, (FramesSignal, Sine, Constant, Cycle, Stream)):
            raise TypeError(f"{type(sig).__name__} is not implemented on the device path")
        leaf_channels = getattr(sig, "channels", 1)
        if self._m.channels == 1:
            if stereo_seen or leaf_channels != 1:
                raise TypeError("this is a Mixer<f32>: it plays Signal<Frame = f32> (no MonoToStereo, no stereo clips)")
        elif (leaf_channels == 1) != stereo_seen:
            raise TypeError("the device Mixer is Mixer<[f32;2]>: mono signals need MonoToStereo::new, stereo clips must not have it")
        chain = chain[::-1]
        if len(chain) > 4:
            raise TypeError("at most 4 filters")
        filt = (_Filter * max(len(chain), 1))()
        for i, (kind, param, _) in enumerate(chain):
            filt[i].kind, filt[i].param = kind, param
        return sig, chain, filt

    def play(self, signal: Signal) -> Mixed:
        """MixerControl::play for Mixer<[f32;2]>: any nest of FixedGain / Gain / Speed / MonoToStereo
        around FramesSignal (mono or stereo clip), Cycle, Sine, Constant or Stream whose output is
        stereo; or a Fader around such a nest."""
        L, m = _lib.lib(), self._m
        sid = C.c_uint32()
        if isinstance(signal, Fader):
            sig, chain, filt = self._parse(signal.inner)
            if isinstance(sig, Stream):
                raise TypeError("Fader<..Stream..> is not implemented on the device path")
            args = _leaf_args(sig, m._keep)
            _lib.check(L.oddio_hip_mixer_play_fader(m._h, args[0], args[1], args[2], args[3], args[4], C.cast(filt, C.c_void_p), len(chain), C.byref(sid)))
            signal.control._bind(self, m, sid.value)
        else:
            sig, chain, filt = self._parse(signal)
            if isinstance(sig, Stream):
                if sig.control._h is None:
                    raise ValueError("the StreamControl has already been dropped")
                _lib.check(L.oddio_hip_mixer_play_stream(m._h, sig.control._h, C.cast(filt, C.c_void_p), len(chain), C.byref(sid)))
            else:
                args = _leaf_args(sig, m._keep)
                _lib.check(L.oddio_hip_mixer_play_chain(m._h, args[0], args[1], args[2], args[3], args[4], C.cast(filt, C.c_void_p), len(chain), C.byref(sid)))
        mixed = Mixed(m, sid.value)
        for i, (_, _, control) in enumerate(chain):
            if control is not None:
                control._bind(m, sid.value, i, mixed)
        return mixed


class FaderControl:
    """src/fader.rs:82-93"""

    def __init__(self):
        self._target = None

    def _bind(self, mixer_control, mixer, sid):
        self._target = (mixer_control, mixer, sid)

    def _bind_scene(self, scene, sid):
        self._target = (None, scene, sid)

    def fade_to(self, signal: Signal, duration: float):
        if self._target is None:
            raise ValueError("the Fader has not been played yet")
        mc, m, sid = self._target
        if mc is None:      # a buffered source of a spatial scene
            leaf, chain = _unwrap_chain(signal)
            if isinstance(leaf, Stream):
                raise TypeError("fading to a Stream is not implemented on the device path")
            filt = (_Filter * max(len(chain), 1))()
            for i, (kind, param, _) in enumerate(chain):
                filt[i].kind, filt[i].param = kind, param
            args = _leaf_args(leaf, m._keep)
            _lib.check(_lib.lib().oddio_hip_source_fade_to(m._h, sid, args[0], args[1], args[2], args[3], args[4], C.cast(filt, C.c_void_p),
                                                           len(chain), np.float32(duration)))
            return
        sig, chain, filt = mc._parse(signal)
        if isinstance(sig, Stream):
            raise TypeError("fading to a Stream is not implemented on the device path")
        args = _leaf_args(sig, m._keep)
        _lib.check(_lib.lib().oddio_hip_mixer_fade_to(m._h, sid, args[0], args[1], args[2], args[3], args[4], C.cast(filt, C.c_void_p), len(chain),
                                                      np.float32(duration)))


class Fader(Signal):
    """Fader::new(inner) -> (FaderControl, Fader)  (src/fader.rs:16-28).  Device support: played in a
    Mixer, or as a buffered source of a spatial scene."""
    seekable = False

    def __init__(self, inner: Signal):
        self.inner = inner
        self.channels = inner.channels
        self.control = FaderControl()

    @classmethod
    def new(cls, inner):
        f = cls(inner)
        return f.control, f


def Mixer(device: int = 0, max_sources: int = 4096, max_frames: int = 4096, channels: int = 2):
    """Mixer::new() -> (MixerControl, Mixer)  (src/mixer.rs:70-81); `channels` selects the frame type T = f32 or [f32; 2]
    (a Mixer<f32> plays mono signals and fills `out` with n_frames floats)."""
    m = _MixerSignal(device, max_sources, max_frames, channels)
    return MixerControl(m), m


def run(signal, sample_rate: int, out: np.ndarray):
    """oddio::run (src/lib.rs:90-93): interval = 1.0 / sample_rate as f32."""
    interval = np.float32(1.0) / np.float32(sample_rate)
    return signal.sample(interval, out)


def frame_stereo(xs: np.ndarray) -> np.ndarray:
    """oddio::frame_stereo (src/lib.rs:98-100): view interleaved [2n] floats as [n, 2] frames."""
    return xs.reshape(-1, 2)
