#!/usr/bin/env python
"""Timing of the paths beside the headline one: buffered spatial sources, the Mixer's general path, Seek-set Cycle
sources (one wave per source since round 2).  Prints ms per 1024-frame callback through the host-output entry point
(stream sync + 8 KiB D2H included) for a few set sizes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oddio_amd as oa  # noqa: E402
from oddio_amd import synth  # noqa: E402

INTERVAL = np.float32(1.0) / np.float32(48000)


def time_calls(sig, n=1024, reps=20):
    for _ in range(3):
        sig.sample_n(INTERVAL, n)
    t0 = time.perf_counter()
    for _ in range(reps):
        sig.sample_n(INTERVAL, n)
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    clip = oa.Frames.from_slice(48000, synth.noise_clip(1, 0, 480000))
    for n_src in (1, 64, 1024, 4096):
        control, scene = oa.SpatialScene(max_sources=8192, max_frames=1024)
        sc = synth.make_scene(5, n_src)
        scene.reserve_buffered(n_src)
        for i in range(n_src):
            gc, g = oa.Gain.new(oa.FramesSignal(clip, 0.0))
            control.play_buffered(g, oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1), 100.0, 48000, 0.1)
        print(f"buffered spatial sources (Gain<FramesSignal>): {n_src:5d} -> {time_calls(scene):8.3f} ms / 1024-frame callback")
        scene.close()
    for n_src in (1, 64, 1024):
        control, mixer = oa.Mixer(max_sources=1024, max_frames=1024)
        for i in range(n_src):
            gc, g = oa.Gain.new(oa.MonoToStereo(oa.FramesSignal(clip, 0.0)))
            control.play(g)
        print(f"mixer general path (Gain<MonoToStereo<FramesSignal>>): {n_src:5d} -> {time_calls(mixer):8.3f} ms / 1024-frame callback")
        mixer.close()
    for n_src in (64, 4096):
        control, scene = oa.SpatialScene(max_sources=8192, max_frames=1024)
        sc = synth.make_scene(5, n_src)
        cyc = oa.Frames.from_slice(48000, synth.noise_clip(2, 0, 5000))
        for i in range(min(n_src, 1024)):
            control.play(oa.Cycle(cyc), oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
        print(f"Seek-set Cycle sources: {min(n_src, 1024):5d} -> {time_calls(scene):8.3f} ms / 1024-frame callback")
        scene.close()

    for n_src in (64, 1024):
        control, scene = oa.SpatialScene(max_sources=8192, max_frames=1024)
        sc = synth.make_scene(6, n_src)
        scene.reserve_buffered(n_src)
        cyc = oa.Frames.from_slice(48000, synth.noise_clip(3, 0, 5000))
        for i in range(n_src):
            gc, g = oa.Gain.new(oa.Cycle(cyc))
            control.play_buffered(g, oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1), 100.0, 48000, 0.1)
        print(f"buffered spatial sources (Gain<Cycle>): {n_src:5d} -> {time_calls(scene):8.3f} ms / 1024-frame callback")
        scene.close()


if __name__ == "__main__":
    main()
