#!/usr/bin/env python
"""Golden fixtures for the buffered path and the filters (round 4): scenes whose sources are filter chains (FixedGain / Gain / Speed,
innermost first) around a FramesSignal, played with play_buffered (rings) or play, and a Mixer of such chains -- with GainControl /
SpeedControl stores, motion updates and a listener rotation on the way.

As in gen_golden.py the expected outputs come from the C restatement (oracle/oddio_oracle.c) after it has agreed, bit for bit, with
the independent numpy restatement (oracle/oracle_np.py) on the same scenario; a fixture is data only.

    python tests/golden/gen_golden_chains.py        # rewrites tests/golden/chains_*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import chains  # noqa: E402
from oddio_amd import synth  # noqa: E402


def make(name, spec):
    ref_c = chains.run(chains.CBackend(spec), spec)
    ref_np = chains.run(chains.NumpyBackend(spec), spec)
    if not np.array_equal(ref_c, ref_np):
        raise SystemExit(f"{name}: C oracle and numpy restatement disagree -- not writing a fixture")
    assert np.abs(ref_c).max() > 0
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **chains.pack(spec, ref_c))
    print(f"{name}: {len(spec['sources'])} sources, {spec['n_callbacks']}x{spec['n_frames']} frames, {os.path.getsize(path) / 1024:.1f} KiB")


def scene_spec(seed, n_src, n_frames, n_callbacks):
    sc = synth.make_scene(seed, n_src, cube=30.0)
    rng = np.random.default_rng(seed)
    shapes = ([(chains.GAIN, np.nan)], [(chains.SPEED, 0.94), (chains.GAIN, 0.5)], [(chains.SPEED, 1.0), (chains.FIXED, -4.5)],
              [(chains.FIXED, 2.0), (chains.GAIN, np.nan), (chains.GAIN, np.nan)], [])
    sources = []
    for i in range(n_src):
        buffered = i % 4 != 3
        sources.append({"clip": synth.noise_clip(seed, i, 7300 + 97 * i), "rate": (48000, 44100)[i % 2], "start": 0.0 if buffered else 0.1,
                        "pos": sc["position"][i], "vel": sc["velocity"][i], "radius": float(sc["radius"][i]),
                        "chain": shapes[i % 5] if buffered else [], "buffered": buffered,
                        "ring_rate": (48000, 44100, 32000)[i % 3], "max_distance": (100.0, 40.0)[i % 2], "buffer_duration": (0.1, 0.06)[i % 2]})
    ctl, motion = [], []
    for cb in (1, 2, 3):
        for i, s in enumerate(sources):
            for w, (kind, _) in enumerate(s["chain"]):
                if kind == chains.GAIN and cb in (1, 2):
                    ctl.append((cb, i, w, 0.2 + 0.13 * ((i + cb + w) % 6)))
                if kind == chains.SPEED and cb == 3:
                    ctl.append((cb, i, w, 1.04 + 0.005 * (i % 7)))
    for cb in (2, 4):
        for j in (0, 5, 9):
            if j < n_src:
                p = (sc["position"][j] + rng.normal(size=3).astype(np.float32) * (25.0 if cb == 4 else 1.0)).astype(np.float32)
                motion.append((cb, j, p, sc["velocity"][j], cb == 4))
    return {"mixer": False, "sources": sources, "ctl": ctl, "motion": motion, "rotation": [(3, np.array([np.cos(0.3), 0.0, np.sin(0.3), 0.0], np.float32))],
            "n_frames": n_frames, "n_callbacks": n_callbacks, "interval": np.float32(1.0) / np.float32(48000)}


def mixer_spec(seed, n_src, n_frames, n_callbacks):
    spec = scene_spec(seed, n_src, n_frames, n_callbacks)
    spec["mixer"] = True
    spec["motion"], spec["rotation"] = [], []
    for i, s in enumerate(spec["sources"]):
        s["buffered"] = False
        s["start"] = 0.002 * (i % 4)
    return spec


def clip_scene_spec(seed, n_src, n_frames, n_callbacks):
    """Round 5: per-source Reinhard (reinhard.rs:22-50) -- Seek sources played with `play` ([FixedGain] [Reinhard] in either order,
    moving, static (resample ratio exactly 1) and with a cursor that starts before the clip) and buffered chains with the clip
    anywhere among the filters."""
    spec = scene_spec(seed, n_src, n_frames, n_callbacks)
    R = (chains.REINHARD, 0.0)
    seek_shapes = ([R], [(chains.FIXED, -3.0), R], [R, (chains.FIXED, 2.5)], [])
    buf_shapes = ([(chains.GAIN, np.nan), R], [R, (chains.SPEED, 0.97)], [(chains.FIXED, 1.5), R, (chains.GAIN, 0.7)])
    for i, s in enumerate(spec["sources"]):
        s["buffered"] = i % 3 == 2
        s["chain"] = list(buf_shapes[(i // 3) % 3] if s["buffered"] else seek_shapes[i % 4])
        s["start"] = 0.0 if s["buffered"] else (0.1, 0.1, -0.003)[i % 3]
        if i % 5 == 1:
            s["vel"] = np.zeros(3, np.float32)
    spec["ctl"] = [(cb, i, w, v) for cb, i, w, v in spec["ctl"] if False]
    for cb in (1, 3):
        for i, s in enumerate(spec["sources"]):
            for w, (kind, _) in enumerate(s["chain"]):
                if kind == chains.GAIN:
                    spec["ctl"].append((cb, i, w, 0.3 + 0.2 * ((i + cb) % 4)))
                if kind == chains.SPEED and cb == 3:
                    spec["ctl"].append((cb, i, w, 1.03))
    return spec


def clip_mixer_spec(seed, n_src, n_frames, n_callbacks):
    spec = mixer_spec(seed, n_src, n_frames, n_callbacks)
    R = (chains.REINHARD, 0.0)
    shapes = ([R], [(chains.FIXED, 2.0), R], [R, (chains.GAIN, np.nan)], [(chains.SPEED, 1.05), R, (chains.FIXED, -2.0)], [])
    for i, s in enumerate(spec["sources"]):
        s["chain"] = list(shapes[i % 5])
    spec["ctl"] = []
    for cb in (1, 2):
        for i, s in enumerate(spec["sources"]):
            for w, (kind, _) in enumerate(s["chain"]):
                if kind == chains.GAIN:
                    spec["ctl"].append((cb, i, w, 0.4 + 0.15 * ((i + cb) % 3)))
    return spec


def main():
    make("chains_buffered_scene", scene_spec(2001, 10, 1024, 6))
    make("chains_buffered_scene_ragged", scene_spec(2002, 7, 700, 5))
    make("chains_mixer", mixer_spec(2003, 10, 1024, 6))
    make("chains_source_clip_scene", clip_scene_spec(2004, 12, 1024, 5))
    make("chains_source_clip_mixer", clip_mixer_spec(2005, 10, 1024, 4))


if __name__ == "__main__":
    main()
