#!/usr/bin/env python
"""The reference's examples/offline.rs scenario rendered through the HIP path.

A 500 Hz tone (amplitude 80, 3 s at 44.1 kHz) flies past the listener at 50 m/s, 10 m to the
side; the scene is rendered in 512-frame blocks with `oddio::run` semantics and written as a
16-bit stereo WAV (`(sample * i16::MAX as f32) as i16`, examples/offline.rs:33-43).
`render(mod, scene_factory)` is backend-agnostic: tests/test_hip_examples.py runs it on the CPU oracle
too and compares.

    python examples/offline.py [--out offline.wav]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DURATION_SECS, RATE, BLOCK_SIZE, SPEED = 3, 44100, 512, 50.0


def boop():
    t = np.arange(RATE * DURATION_SECS, dtype=np.float32) / np.float32(RATE)
    return (np.sin(t * np.float32(500.0) * np.float32(2.0) * np.float32(np.pi)) * np.float32(80.0)).astype(np.float32)


def render(mod, scene_factory):
    control, scene = scene_factory()
    frames = mod.Frames.from_slice(RATE, boop())
    control.play(mod.FramesSignal(frames, 0.0), mod.SpatialOptions(position=[-SPEED, 10.0, 0.0], velocity=[SPEED, 0.0, 0.0], radius=0.1))
    blocks = []
    for _ in range(RATE * DURATION_SECS // BLOCK_SIZE):
        blocks.append(mod.run(scene, RATE, np.zeros((BLOCK_SIZE, 2), dtype=np.float32)).copy())
    return np.concatenate(blocks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="offline.wav")
    args = ap.parse_args()
    import oddio_amd as oa
    out = render(oa, lambda: oa.SpatialScene(max_sources=8, max_frames=BLOCK_SIZE))
    from oddio_amd import wav
    wav.write_wav(args.out, RATE, out)                     # (sample * i16::MAX as f32) as i16, offline.rs:38
    print(f"wrote {args.out}: {len(out)} frames, peak {np.abs(out).max():.4f}")


if __name__ == "__main__":
    main()
