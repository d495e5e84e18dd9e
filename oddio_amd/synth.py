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
