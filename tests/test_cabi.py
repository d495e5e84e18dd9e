"""The C-ABI library loads and exports every symbol include/oddio_hip.h declares (CPU; no compute)."""
import ctypes as C

from oddio_amd import _lib


def test_library_exports_every_declared_symbol():
    _lib.build()
    L = _lib.lib()
    declared = _lib.declared_symbols()
    assert len(declared) >= 40
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert set(declared) == set(L._signatures), set(declared) ^ set(L._signatures)
    assert L.oddio_hip_abi_version() == 1


def test_argument_validation_without_gpu():
    L = _lib.lib()
    assert L.oddio_hip_scene_set_postfx(None, 1) == -1           # ODDIO_HIP_EINVAL
    assert b"NULL" in L.oddio_hip_last_error() or b"bad" in L.oddio_hip_last_error()
    h = C.c_void_p()
    assert L.oddio_hip_scene_create(0, 0, 16, C.byref(h)) == -1  # max_sources must be > 0
    assert L.oddio_hip_frames_from_slice(0, 48000, None, 0, C.byref(h)) != 0   # empty clip rejected
