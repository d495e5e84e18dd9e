#!/usr/bin/env python
"""A Mixer<[f32;2]> of 262 144 MonoToStereo<FramesSignal> sources, each on its own clip at the output rate (mixer_mix_unit), device
output, callbacks enqueued back to back: ms per 1024-frame callback and the fraction of the 8 TB/s peak for S (4 N + 128) + 8 N bytes.
    [ODDIO_HIP_LIB=variant.so] python tools/bench_mixer_unit.py [--sources 262144]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import oddio_amd as oa  # noqa: E402

N, RATE = 1024, 48000
ap = argparse.ArgumentParser()
ap.add_argument("--sources", type=int, default=262144)
ap.add_argument("--mode", default="fast")
args = ap.parse_args()
S, L = args.sources, 32768
dev = torch.device("cuda", 0)
clips = torch.empty((S, L), device=dev, dtype=torch.float32)
for s0 in range(0, S, 16384):
    clips[s0:s0 + 16384] = torch.rand((min(16384, S - s0), L), device=dev) * 2.0 - 1.0
frames = [oa.Frames.from_device_ptr(RATE, clips.data_ptr() + 4 * L * i, L, device=0, copy=False) for i in range(S)]
control, mixer = oa.Mixer(max_sources=S, max_frames=N)
mixer.set_mode({"fast": oa.MODE_FAST, "tracked": oa.MODE_TRACKED}[args.mode])
for i in range(S):
    control.play(oa.MonoToStereo(oa.FramesSignal(frames[i], 0.0)))
out = torch.zeros((N, 2), device=dev, dtype=torch.float32)
interval = np.float32(1.0) / np.float32(RATE)
for _ in range(8):
    mixer.sample_device(interval, out.data_ptr(), N)
mixer.synchronize()
t0 = time.perf_counter()
K = 20
for _ in range(K):
    mixer.sample_device(interval, out.data_ptr(), N)
mixer.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
b = S * (4 * N + 128) + 8 * N
print(f"{os.environ.get('ODDIO_HIP_LIB', 'libodd_hip.so').split('/')[-1]:24s} {args.mode}: {S} sources  {ms:.4f} ms / callback  {b / (ms * 1e-3) / 1e9 / 8000:.3f} of the peak", flush=True)
assert len(mixer) == S and bool(torch.isfinite(out).all())
