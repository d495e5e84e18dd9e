#!/usr/bin/env python
"""Churn test: 30 000 callbacks of a scene whose sources are continuously played, finish, are removed and
have their handles released (seekable and buffered); prints free HBM and host RSS every 5 000 callbacks.
Both must stay flat (GPU box)."""
import sys, os, resource
sys.path.insert(0, ".")
import numpy as np, ctypes
import oddio_amd as oa
from oddio_amd import synth
hip = ctypes.CDLL("libamdhip64.so")
def gpu_free():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value
control, scene = oa.SpatialScene(max_sources=256, max_frames=512)
scene.reserve_buffered(64)
clips = [oa.Frames.from_slice(48000, synth.noise_clip(1, i, 3000)) for i in range(4)]
rng = np.random.default_rng(0)
interval = np.float32(1/48000)
live = []
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
marks = []
for cb in range(30000):
    if len(live) < 100:
        p, v = rng.uniform(-5, 5, 3).astype(np.float32), rng.uniform(-10, 10, 3).astype(np.float32)
        if cb % 7 == 0:
            gc, g = oa.Gain.new(oa.FramesSignal(clips[cb % 4], 0.0))
            live.append(control.play_buffered(g, oa.SpatialOptions(p, v, 0.1), 30.0, 48000, 0.05))
        else:
            live.append(control.play(oa.FramesSignal(clips[cb % 4], 0.0), oa.SpatialOptions(p, v, 0.1)))
    scene.sample_n(interval, 512)
    if cb % 16 == 0:
        keep = []
        for h in live:
            if h.is_finished(): h.release()
            else: keep.append(h)
        live = keep
    if cb % 5000 == 0:
        marks.append((cb, gpu_free() >> 10, rss(), len(scene), scene.len_buffered()))
        print(marks[-1], flush=True)
print("gpu free delta KiB (first->last after warmup):", marks[1][1] - marks[-1][1], "rss delta KiB:", marks[-1][2] - marks[1][2])
