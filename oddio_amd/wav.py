"""WAV <-> float frames, with the conversions the reference's examples use (host side only).

examples/wav.rs:30-42   integer PCM -> f32:  sample as f32 / (2^(bits-1) - 1) as f32; float PCM as is;
                        channels stay interleaved and are viewed as stereo frames (`frame_stereo`).
examples/offline.rs:38, examples/adapt.rs:32-36   f32 -> 16-bit PCM: (sample * i16::MAX as f32) as i16
                        (Rust `as`: toward zero, saturating, NaN -> 0).

The reference reads/writes through the `hound` crate; this is a small RIFF reader/writer with the
same sample conventions (8-bit WAV data is unsigned on disk and signed after decoding, like hound).
"""
from __future__ import annotations

import struct

import numpy as np

_PCM, _FLOAT, _EXTENSIBLE = 1, 3, 0xFFFE


def read_wav(path_or_bytes):
    """-> (sample_rate, frames) with frames float32 [n] (mono) or [n, channels]."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None or len(fmt) < 16:
        raise ValueError("missing fmt/data chunk")
    tag, channels, rate, _, _, bits = struct.unpack_from("<HHIIHH", fmt, 0)
    if tag == _EXTENSIBLE and len(fmt) >= 26:
        tag = struct.unpack_from("<H", fmt, 24)[0]
    if channels == 0:
        raise ValueError("zero channels")
    if tag == _FLOAT and bits == 32:
        x = np.frombuffer(payload[:len(payload) // 4 * 4], dtype="<f4").astype(np.float32)
    elif tag == _PCM and bits in (8, 16, 24, 32):
        if bits == 8:
            ints = np.frombuffer(payload, dtype=np.uint8).astype(np.int32) - 128
        elif bits == 16:
            ints = np.frombuffer(payload[:len(payload) // 2 * 2], dtype="<i2").astype(np.int32)
        elif bits == 24:
            b = np.frombuffer(payload[:len(payload) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            ints = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            ints = np.where(ints >= 1 << 23, ints - (1 << 24), ints)
        else:
            ints = np.frombuffer(payload[:len(payload) // 4 * 4], dtype="<i4").astype(np.int64)
        max_value = np.float32(2 ** (bits - 1) - 1)             # wav.rs:33 (u32 -> f32)
        x = ints.astype(np.float32) / max_value                  # wav.rs:36
    else:
        raise ValueError(f"unsupported WAV encoding (format tag {tag}, {bits} bits)")
    n = len(x) // channels
    x = np.ascontiguousarray(x[:n * channels])
    return int(rate), (x if channels == 1 else x.reshape(n, channels))


def to_i16(frames: np.ndarray) -> np.ndarray:
    """(sample * i16::MAX as f32) as i16 -- offline.rs:38 / adapt.rs:34."""
    y = np.asarray(frames, dtype=np.float32) * np.float32(32767.0)
    y = np.where(np.isnan(y), np.float32(0.0), y)
    return np.clip(np.trunc(y), -32768.0, 32767.0).astype(np.int16)


def write_wav(path, sample_rate: int, frames: np.ndarray, float32: bool = False):
    """16-bit PCM (default, the examples' WavSpec) or 32-bit float WAV from [n] / [n, channels] frames."""
    frames = np.asarray(frames, dtype=np.float32)
    channels = 1 if frames.ndim == 1 else frames.shape[1]
    if float32:
        tag, bits, body = _FLOAT, 32, np.ascontiguousarray(frames, dtype="<f4").tobytes()
    else:
        tag, bits, body = _PCM, 16, np.ascontiguousarray(to_i16(frames), dtype="<i2").tobytes()
    block = channels * bits // 8
    fmt = struct.pack("<HHIIHH", tag, channels, int(sample_rate), int(sample_rate) * block, block, bits)
    out = b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(body) + (len(body) & 1)) + b"WAVE"
    out += b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")
    with open(path, "wb") as f:
        f.write(out)
