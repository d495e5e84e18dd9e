"""Host-side WAV conversions (oddio_amd/wav.py) against the formulas in the reference's examples
(examples/wav.rs:30-42, examples/offline.rs:38)."""
import struct

import numpy as np
import pytest

from oddio_amd import wav


def riff(tag, channels, rate, bits, payload):
    block = channels * bits // 8
    fmt = struct.pack("<HHIIHH", tag, channels, rate, rate * block, block, bits)
    return (b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(payload)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
            + b"LIST" + struct.pack("<I", 4) + b"abcd" + b"data" + struct.pack("<I", len(payload)) + payload)


def test_int_pcm_scaling_matches_wav_rs():
    ints = np.array([0, 1, -1, 32767, -32768, 12345, -4321, 7], dtype="<i2")
    rate, x = wav.read_wav(riff(1, 2, 8000, 16, ints.tobytes()))
    assert rate == 8000 and x.shape == (4, 2)
    np.testing.assert_array_equal(x.ravel(), ints.astype(np.float32) / np.float32(32767))   # max_value = 2^15 - 1
    assert x.min() < -1.0                                                                    # -32768/32767, as in the reference
    rate, x8 = wav.read_wav(riff(1, 1, 11025, 8, bytes([0, 128, 255, 1])))
    np.testing.assert_array_equal(x8, np.array([-128, 0, 127, -127], np.float32) / np.float32(127))
    b24 = b"".join(int(v).to_bytes(3, "little", signed=True) for v in (8388607, -8388608, 5, -5))
    _, x24 = wav.read_wav(riff(1, 1, 48000, 24, b24))
    np.testing.assert_array_equal(x24, np.array([8388607, -8388608, 5, -5], np.float32) / np.float32(8388607))
    _, xf = wav.read_wav(riff(3, 1, 48000, 32, np.array([0.25, -1.5], "<f4").tobytes()))
    np.testing.assert_array_equal(xf, np.array([0.25, -1.5], np.float32))


def test_f32_to_i16_is_rust_as_cast():
    x = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 1.5, -1.5, 0.99999, -3.05e-5, np.nan], np.float32)
    got = wav.to_i16(x)
    want = [0, 16383, -16383, 32767, -32767, 32767, -32768, 32766, 0, 0]    # toward zero, saturating, NaN -> 0
    assert got.tolist() == want


def test_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=(1000, 2)).astype(np.float32)
    p = tmp_path / "a.wav"
    wav.write_wav(p, 22050, x)
    rate, y = wav.read_wav(p)
    assert rate == 22050 and y.shape == x.shape
    np.testing.assert_array_equal(y, wav.to_i16(x).astype(np.float32) / np.float32(32767))
    wav.write_wav(p, 48000, x[:, 0], float32=True)
    rate, z = wav.read_wav(p)
    np.testing.assert_array_equal(z, x[:, 0])
    with pytest.raises(ValueError):
        wav.read_wav(b"RIFFxxxxWAVE")
