#!/usr/bin/env python
"""Mixer<[f32;2]> of MonoToStereo<FramesSignal> sources at scale: ms per 1024-frame callback (host-output calls: the Mixer has no
device-output entry point), FAST and ORDERED, against a SpatialScene of as many sources.
    python tools/bench_mixer_scale.py [--sources 4096,65536]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import oddio_amd as oa  # noqa: E402
from oddio_amd import synth  # noqa: E402

INTERVAL = np.float32(1.0) / np.float32(48000)


def timed(sig, reps=24, warm=24):
    for _ in range(warm):
        sig.sample_n(INTERVAL, 1024)
    t0 = time.perf_counter()
    for _ in range(reps):
        sig.sample_n(INTERVAL, 1024)
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sources", default="4096,65536")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n_clips, length = 4096, 48000 * 3
    clips = (torch.rand((n_clips, length), device=dev, dtype=torch.float32) * 2.0 - 1.0).contiguous()
    frames = [oa.Frames.from_device_ptr(48000, clips.data_ptr() + 4 * length * i, length, device=0, copy=False) for i in range(n_clips)]
    for S in [int(x) for x in args.sources.split(",")]:
        pick = ((np.arange(S, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(20)) % np.uint64(n_clips)
        control, mixer = oa.Mixer(max_sources=S, max_frames=1024)
        for i in range(S):
            control.play(oa.MonoToStereo(oa.FramesSignal(frames[int(pick[i])], 0.25)))
        fast = timed(mixer)
        mixer.set_mode(oa.MODE_TRACKED)
        tracked = timed(mixer, reps=6, warm=2)
        mixer.set_mode(oa.MODE_ORDERED)
        ordered = timed(mixer, reps=6, warm=2)
        mixer.close()
        # the general path: a Gain around every source (GainControl per sound: what a game's mixer holds)
        control, mixer = oa.Mixer(max_sources=S, max_frames=1024)
        for i in range(S):
            gc, g = oa.Gain.new(oa.MonoToStereo(oa.FramesSignal(frames[int(pick[i])], 0.25)))
            control.play(g)
        gen_fast = timed(mixer, reps=8, warm=4)
        mixer.set_mode(oa.MODE_ORDERED)
        gen_ord = timed(mixer, reps=4, warm=2)
        mixer.close()
        print(f"{S:7d} sources: Mixer of Gain<MonoToStereo<FramesSignal>> (general path) FAST {gen_fast:8.4f} ms  ORDERED {gen_ord:8.4f} ms", flush=True)
        sc = synth.make_scene(3, S)
        scontrol, scene = oa.SpatialScene(max_sources=S, max_frames=1024)
        scontrol.play_frames_batch([frames[int(k)] for k in pick], np.full(S, 0.25), sc["position"], sc["velocity"], sc["radius"])
        sp = timed(scene)
        scene.set_mode(oa.MODE_ORDERED)
        sp_ord = timed(scene, reps=6, warm=2)
        scene.close()
        print(f"{S:7d} sources: Mixer FAST {fast:8.4f} ms  TRACKED {tracked:8.4f} ms  ORDERED {ordered:8.4f} ms | SpatialScene FAST {sp:8.4f} ms  ORDERED {sp_ord:8.4f} ms (host-output callbacks)", flush=True)


if __name__ == "__main__":
    main()
