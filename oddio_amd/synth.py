"""Deterministic synthetic scenes: the same generator feeds the HIP path,
the CPU oracle, the golden fixtures and bench.py, so all of them consume identical inputs.

PRNG: SplitMix64, one stream per source: state0 = seed ^ (idx * 0x9E3779B97F4A7C15),
u01 = (x >> 40) * 2^-24.  numpy only; no torch, no oracle imports.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


class SplitMixStreams:
    """`n` independent SplitMix64 streams advanced in lock step (vectorised)."""

    def __init__(self, seed: int, n: int, first_index: int = 0):
        idx = np.arange(first_index, first_index + n, dtype=np.uint64)
        with np.errstate(over="ignore"):
            self.state = np.uint64(seed) ^ (idx * _GOLDEN)

    def next_u64(self) -> np.ndarray:
        with np.errstate(over="ignore"):
            self.state = self.state + _GOLDEN
            z = self.state.copy()
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
        return z

    def next_u01(self) -> np.ndarray:
        """float32 in [0, 1) with 24 random bits."""
        return ((self.next_u64() >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)

    def uniform(self, lo: float, hi: float) -> np.ndarray:
        return (np.float32(lo) + self.next_u01() * np.float32(hi - lo)).astype(np.float32)


def make_scene(seed: int, n_sources: int, first_index: int = 0, *, cube: float = 50.0, vmax: float = 20.0,
               fmin: float = 100.0, fmax: float = 2000.0, radius: float = 0.1) -> dict:
    """Random moving sources: position uniform in [-cube, cube]^3 (re-drawn while |p| < 1),
    velocity uniform in [-vmax, vmax]^3, tone frequency uniform in [fmin, fmax] Hz."""
    st = SplitMixStreams(seed, n_sources, first_index)
    pos = np.stack([st.uniform(-cube, cube) for _ in range(3)], axis=1)
    for _ in range(64):
        bad = np.linalg.norm(pos.astype(np.float64), axis=1) < 1.0
        if not bad.any():
            break
        redraw = np.stack([st.uniform(-cube, cube) for _ in range(3)], axis=1)
        pos[bad] = redraw[bad]
    vel = np.stack([st.uniform(-vmax, vmax) for _ in range(3)], axis=1)
    freq = st.uniform(fmin, fmax)
    phase = (st.next_u01() * np.float32(2.0 * np.pi)).astype(np.float32)
    return {
        "seed": seed, "n": n_sources, "first_index": first_index,
        "position": pos.astype(np.float32), "velocity": vel.astype(np.float32),
        "freq_hz": freq.astype(np.float32), "phase": phase,
        "radius": np.full(n_sources, radius, dtype=np.float32),
    }


def sine_clip(freq_hz: float, length: int, rate: int = 48000, amplitude: float = 1.0) -> np.ndarray:
    """A*sin(2*pi*f*n/rate), evaluated in f64 and rounded once to f32."""
    n = np.arange(length, dtype=np.float64)
    return (amplitude * np.sin(2.0 * np.pi * float(freq_hz) * n / float(rate))).astype(np.float32)


def noise_clip(seed: int, index: int, length: int) -> np.ndarray:
    """White noise in [-1, 1): the harsher interpolation test."""
    st = SplitMixStreams(seed ^ 0x5EED, 1, index)
    out = np.empty(length, dtype=np.float32)
    # one stream, `length` draws: vectorise by jumping the state (SplitMix64 state is a counter)
    with np.errstate(over="ignore"):
        states = st.state[0] + _GOLDEN * np.arange(1, length + 1, dtype=np.uint64)
        z = states
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    out[:] = (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24) * np.float32(2.0) - np.float32(1.0)
    return out


def adversarial_scene(kind: str, n_sources: int, seed: int = 7, clip_len: int = 8192) -> dict:
    """Scenes built to stress a summation order rather than the resampler (tests, bench.py's parity block): the sources share a
    small clip bank (`bank` [n_clips, clip_len], source i plays clip `clip_of[i]`), positions / velocities as in make_scene.

    "cancelling": coherent sources that cancel in pairs -- 4 low sine tones and their negations, sources 2k / 2k + 1 on the same
        tone with opposite signs, all inside a 10 cm cluster 3.6 m from the listener moving at <= 1 m/s: every addend is ~1e2 x
        the residue it leaves, the reference's running sum (src/spatial.rs:459-460) hovers around zero.
    "parked": the sources the reference's reverse walk (spatial.rs:204) meets first -- the highest slots -- are static DC clips that
        lift the running sum of both ears to just under 16.0; the other ~260 000 are quiet noise sources whose random walk then
        crosses that power of two (where the ulp of the sum, and with it every rounding error, doubles) again and again."""
    st = SplitMixStreams(seed ^ 0xADFE, n_sources)
    if kind == "cancelling":
        tones = np.stack([sine_clip(110.0 * (k + 1), clip_len) for k in range(4)])
        bank = np.concatenate([tones, -tones]).astype(np.float32)
        i = np.arange(n_sources)
        clip_of = ((i // 2) % 4 + 4 * (i % 2)).astype(np.uint32)
        centre = np.array([3.0, 0.5, -2.0], dtype=np.float32)
        pos = centre[None, :] + np.stack([st.uniform(-0.05, 0.05) for _ in range(3)], axis=1)
        vel = np.stack([st.uniform(-1.0, 1.0) for _ in range(3)], axis=1)
    elif kind == "parked":
        n_noise_clips = 64
        bank = np.concatenate([np.ones((1, clip_len), dtype=np.float32),
                               np.stack([noise_clip(seed, k, clip_len) for k in range(n_noise_clips)])]).astype(np.float32)
        # one DC source 1 m in front of the listener adds radius / |p - ear| * (0.5 + 0.5 / sqrt(17)) to each ear (spatial.rs:531-549)
        g_dc = 0.1 / np.sqrt(1.0 + 0.1075 ** 2) * (0.5 + 0.5 / np.sqrt(17.0))
        n_dc = min(int(np.floor(16.0 / g_dc)), n_sources // 2)
        n_noise = n_sources - n_dc
        pos = np.stack([st.uniform(-30.0, 30.0) for _ in range(3)], axis=1)
        near = np.linalg.norm(pos.astype(np.float64), axis=1) < 4.0
        pos[near] = pos[near] + np.float32(8.0)
        vel = np.stack([st.uniform(-10.0, 10.0) for _ in range(3)], axis=1)
        pos[n_noise:] = np.array([0.0, 0.0, -1.0], dtype=np.float32)
        vel[n_noise:] = 0.0
        clip_of = (1 + (np.arange(n_sources, dtype=np.uint64) * np.uint64(2654435761) >> np.uint64(7)) % np.uint64(n_noise_clips)).astype(np.uint32)
        clip_of[n_noise:] = 0
    else:
        raise ValueError(kind)
    return {"kind": kind, "n": n_sources, "bank": bank, "clip_of": clip_of, "position": pos.astype(np.float32),
            "velocity": vel.astype(np.float32), "radius": np.full(n_sources, 0.1, dtype=np.float32)}
