"""ctypes loader for libodd_hip.so (the C ABI of include/oddio_hip.h).

The HIP extension is the product: if the shared library is missing this module raises -- there is
no CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("ODDIO_HIP_LIB") or os.path.join(_HERE, "libodd_hip.so")   # env override: kernel A/B builds
HEADER = os.path.join(os.path.dirname(_HERE), "include", "oddio_hip.h")


def build(force: bool = False) -> str:
    """Compile libodd_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    newest = max(os.path.getmtime(os.path.join(src_dir, f)) for f in os.listdir(src_dir))
    newest = max(newest, os.path.getmtime(HEADER))
    if force or not os.path.exists(SO_PATH) or os.path.getmtime(SO_PATH) < newest:
        subprocess.check_call(["make", "-C", src_dir, "-s"])
    return SO_PATH


def mix_kernel_source_hash() -> str:
    """sha256 (first 16 hex digits) of the sources the scene's mix kernels are compiled from -- what profiles/pmc_latest.json is
    stamped with by tools/make_pmc_json.py, so that bench.py can tell when the counters it quotes were measured on other kernels."""
    import hashlib
    h = hashlib.sha256()
    for name in ("device_types.h", "kernels.h", "pair_kernels.h"):
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


DEBUG_SO_PATH = os.path.join(_HERE, "libodd_hip_debug.so")


def build_debug(force: bool = False) -> str:
    """The bounds-checked build (`make debug`): same sources with -DODDIO_HIP_BOUNDS; used by tests/test_hip_bounds_build.py
    through ODDIO_HIP_LIB, never by default."""
    src_dir = os.path.join(_HERE, "csrc")
    newest = max(os.path.getmtime(os.path.join(src_dir, f)) for f in os.listdir(src_dir))
    newest = max(newest, os.path.getmtime(HEADER))
    if force or not os.path.exists(DEBUG_SO_PATH) or os.path.getmtime(DEBUG_SO_PATH) < newest:
        subprocess.check_call(["make", "-C", src_dir, "-s", "debug"])
    return DEBUG_SO_PATH


def declared_symbols() -> list[str]:
    """Every function include/oddio_hip.h declares."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oddio_hip_[a-z0-9_]+)\s*\(", text)))


class OddioHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"oddio_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    import sys
    if "torch" in sys.modules:
        # when PyTorch shares the process, let it bring up its HIP runtime first so that both bind
        # the same libamdhip64 (the reverse order leaves torch without a device)
        torch = sys.modules["torch"]
        if torch.cuda.is_available():
            torch.cuda.init()
    L = C.CDLL(SO_PATH)
    vp, f32, f64, u32, sz, i32 = C.c_void_p, C.c_float, C.c_double, C.c_uint32, C.c_size_t, C.c_int
    fp, u32p, vpp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p)
    sig = {
        "oddio_hip_abi_version": (i32, []),
        "oddio_hip_bounds_checked": (i32, []),
        "oddio_hip_last_error": (C.c_char_p, []),
        "oddio_hip_device_count": (i32, [C.POINTER(i32)]),
        "oddio_hip_frames_from_slice": (i32, [i32, u32, fp, sz, vpp]),
        "oddio_hip_frames_from_slice_stereo": (i32, [i32, u32, fp, sz, vpp]),
        "oddio_hip_frames_from_device": (i32, [i32, u32, vp, sz, i32, vpp]),
        "oddio_hip_frames_retain": (i32, [vp]),
        "oddio_hip_frames_release": (i32, [vp]),
        "oddio_hip_frames_info": (i32, [vp, u32p, C.POINTER(sz)]),
        "oddio_hip_frames_refcount": (i32, [vp, C.POINTER(i32)]),
        "oddio_hip_scene_create": (i32, [i32, u32, u32, vpp]),
        "oddio_hip_scene_destroy": (i32, [vp]),
        "oddio_hip_scene_play_frames": (i32, [vp, vp, f64, f32, fp, fp, f32, u32p]),
        "oddio_hip_scene_play_sine": (i32, [vp, f32, f32, f32, fp, fp, f32, u32p]),
        "oddio_hip_scene_play_constant": (i32, [vp, f32, fp, fp, f32, u32p]),
        "oddio_hip_scene_set_adapt": (i32, [vp, i32, f32, f32, f32, f32, f32]),
        "oddio_hip_mixer_set_adapt": (i32, [vp, i32, f32, f32, f32, f32, f32]),
        "oddio_hip_scene_play_frames_downmix": (i32, [vp, vp, f64, f32, fp, fp, f32, u32p]),
        "oddio_hip_stream_create": (i32, [i32, u32, C.c_size_t, u32, vpp]),
        "oddio_hip_stream_write": (i32, [vp, fp, C.c_size_t, C.POINTER(C.c_size_t)]),
        "oddio_hip_stream_free": (i32, [vp, C.POINTER(C.c_size_t)]),
        "oddio_hip_stream_drop": (i32, [vp]),
        "oddio_hip_scene_play_buffered_stream": (i32, [vp, vp, vp, i32, fp, fp, f32, f32, u32, f32, u32p]),
        "oddio_hip_scene_play_buffered_fader": (i32, [vp, i32, vp, f64, f32, f32, vp, i32, fp, fp, f32, f32, u32, f32, u32p]),
        "oddio_hip_source_fade_to": (i32, [vp, u32, i32, vp, f64, f32, f32, vp, i32, f32]),
        "oddio_hip_mixer_play_fader": (i32, [vp, i32, vp, f64, f32, f32, vp, i32, u32p]),
        "oddio_hip_mixer_fade_to": (i32, [vp, u32, i32, vp, f64, f32, f32, vp, i32, f32]),
        "oddio_hip_mixer_play_stream": (i32, [vp, vp, vp, i32, u32p]),
        "oddio_hip_scene_play_cycle": (i32, [vp, vp, f32, fp, fp, f32, u32p]),
        "oddio_hip_scene_play_frames_batch": (i32, [vp, sz, vpp, C.POINTER(f64), fp, fp, fp, fp, u32p]),
        "oddio_hip_source_set_motion": (i32, [vp, u32, fp, fp, i32]),
        "oddio_hip_scene_reserve_buffered": (i32, [vp, u32]),
        "oddio_hip_scene_play_buffered": (i32, [vp, i32, vp, f64, f32, f32, vp, i32, fp, fp, f32, f32, u32, f32, u32p]),
        "oddio_hip_scene_device_updates_pending": (i32, [vp, C.POINTER(sz)]),
        "oddio_hip_scene_play_filtered": (i32, [vp, i32, vp, f64, f32, f32, vp, i32, fp, fp, f32, u32p]),
        "oddio_hip_source_set_gain": (i32, [vp, u32, i32, f32]),
        "oddio_hip_source_set_gain_db": (i32, [vp, u32, i32, f32]),
        "oddio_hip_source_set_speed": (i32, [vp, u32, i32, f32]),
        "oddio_hip_scene_len_buffered": (i32, [vp, C.POINTER(sz)]),
        "oddio_hip_source_get_amplitude_ratio": (i32, [vp, u32, i32, fp]),
        "oddio_hip_source_get_gain_db": (i32, [vp, u32, i32, fp]),
        "oddio_hip_source_get_speed": (i32, [vp, u32, i32, fp]),
        "oddio_hip_mixer_get_amplitude_ratio": (i32, [vp, u32, i32, fp]),
        "oddio_hip_mixer_get_gain_db": (i32, [vp, u32, i32, fp]),
        "oddio_hip_mixer_get_speed": (i32, [vp, u32, i32, fp]),
        "oddio_hip_scene_play_buffered_batch": (i32, [vp, sz, vpp, C.POINTER(f64), C.POINTER(i32), i32, fp, fp, fp, fp, f32, u32, f32, u32p]),
        "oddio_hip_scene_set_control_batch": (i32, [vp, sz, u32p, i32, fp]),
        "oddio_hip_scene_set_control_device": (i32, [vp, sz, vp, i32, vp]),
        "oddio_hip_scene_set_motion_device": (i32, [vp, sz, vp, vp, vp, i32]),
        "oddio_hip_scene_set_buffered_fast": (i32, [vp, i32]),
        "oddio_hip_debug_buffered_slow": (i32, [vp, u32p]),
        "oddio_hip_debug_reset_buffered_clock": (i32, [vp, f64]),
        "oddio_hip_source_is_finished": (i32, [vp, u32, C.POINTER(i32)]),
        "oddio_hip_source_release": (i32, [vp, u32]),
        "oddio_hip_source_playback_position": (i32, [vp, u32, C.POINTER(f64)]),
        "oddio_hip_scene_set_listener_rotation": (i32, [vp, fp]),
        "oddio_hip_scene_set_postfx": (i32, [vp, i32]),
        "oddio_hip_scene_set_mode": (i32, [vp, i32]),
        "oddio_hip_scene_set_exact_updates": (i32, [vp, i32]),
        "oddio_hip_scene_len": (i32, [vp, C.POINTER(sz)]),
        "oddio_hip_scene_sample": (i32, [vp, f32, fp, sz]),
        "oddio_hip_scene_run": (i32, [vp, u32, fp, sz]),
        "oddio_hip_scene_sample_device": (i32, [vp, f32, vp, sz]),
        "oddio_hip_postfx_device": (i32, [i32, i32, vp, sz, vp]),
        "oddio_hip_scene_synchronize": (i32, [vp]),
        "oddio_hip_reduce_unique_id": (i32, [vp, sz]),
        "oddio_hip_scene_reduce_init": (i32, [vp, i32, i32, vp, sz]),
        "oddio_hip_scene_reduce_destroy": (i32, [vp]),
        "oddio_hip_scene_reduce_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.c_char_p, sz]),
        "oddio_hip_scene_reduce_init_p2p": (i32, [vp, i32, i32, vp, sz]),
        "oddio_hip_scene_stream": (i32, [vp, vpp]),
        "oddio_hip_scene_set_stream": (i32, [vp, vp]),
        "oddio_hip_scene_seek_all": (i32, [vp, f32]),
        "oddio_hip_scene_set_profiling": (i32, [vp, i32]),
        "oddio_hip_scene_last_kernel_ms": (i32, [vp, fp]),
        "oddio_hip_scene_kernel_ms_history": (i32, [vp, fp, sz, C.POINTER(sz)]),
        "oddio_hip_scene_buffered_ms_history": (i32, [vp, fp, sz, C.POINTER(sz)]),
        "oddio_hip_scene_set_motion_batch": (i32, [vp, sz, u32p, fp, fp, i32]),
        "oddio_hip_debug_mix_occupancy": (i32, [i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "oddio_hip_mixer_create": (i32, [i32, u32, u32, vpp]),
        "oddio_hip_mixer_create_mono": (i32, [i32, u32, u32, vpp]),
        "oddio_hip_mixer_destroy": (i32, [vp]),
        "oddio_hip_mixer_play_sine": (i32, [vp, f32, f32, f32, u32p]),
        "oddio_hip_mixer_play_frames": (i32, [vp, vp, f64, f32, u32p]),
        "oddio_hip_mixer_play_constant": (i32, [vp, f32, u32p]),
        "oddio_hip_mixer_play_chain": (i32, [vp, i32, vp, f64, f32, f32, vp, i32, u32p]),
        "oddio_hip_mixer_set_gain": (i32, [vp, u32, i32, f32]),
        "oddio_hip_mixer_set_gain_db": (i32, [vp, u32, i32, f32]),
        "oddio_hip_mixer_set_speed": (i32, [vp, u32, i32, f32]),
        "oddio_hip_mixer_stop": (i32, [vp, u32]),
        "oddio_hip_mixer_source_release": (i32, [vp, u32]),
        "oddio_hip_mixer_is_stopped": (i32, [vp, u32, C.POINTER(i32)]),
        "oddio_hip_mixer_len": (i32, [vp, C.POINTER(sz)]),
        "oddio_hip_mixer_set_postfx": (i32, [vp, i32]),
        "oddio_hip_mixer_set_mode": (i32, [vp, i32]),
        "oddio_hip_mixer_sample": (i32, [vp, f32, fp, sz]),
        "oddio_hip_mixer_sample_device": (i32, [vp, f32, vp, sz]),
        "oddio_hip_mixer_synchronize": (i32, [vp]),
        "oddio_hip_mixer_run": (i32, [vp, u32, fp, sz]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    L._signatures = sig
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise OddioHipError(rc, lib().oddio_hip_last_error().decode("utf-8", "replace"))
