"""What do the per-kernel hipEvents of `set_profiling(True)` cost a callback?  (bench.py keeps them on inside
its timed region because the roofline figure must come from the timed callbacks themselves.)

    python tools/event_overhead.py [--sources 262144] [--steps 40]
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sources", type=int, default=262144)
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    import torch

    import bench

    g = bench.build_gpu_scene(0, args.sources, 65536, 2024, 1.0)
    scene, control = g["scene"], g["control"]
    out = torch.zeros((1024, 2), dtype=torch.float32, device="cuda")
    interval = np.float32(1.0) / np.float32(48000)

    span = 9                       # callbacks a 65 536-sample clip lasts from 1.0 s in, with margin (as in bench.py)
    step = [0]

    def one():
        if step[0] and step[0] % span == 0:
            scene.seek_all(-float(span * 1024) / 48000)
        scene.sample_device(interval, out.data_ptr(), 1024)
        step[0] += 1

    def run(n, prof):
        control.set_motion_batch(g["ids"], g["spec"]["position"], g["spec"]["velocity"], True)
        scene.set_profiling(prof)
        for _ in range(4):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            one()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        scene.set_profiling(False)
        return ms

    scratch = torch.rand((1 << 28,), dtype=torch.float32, device="cuda")
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.2:
        torch.sin(scratch)
    torch.cuda.synchronize()
    for rep in range(4):
        a = run(args.steps, 1)
        m = run(args.steps, 2)
        b = run(args.steps, 0)
        print(f"rep {rep}: ms/callback with events around every stage {a:.4f} | around the mix kernel only {m:.4f} | none {b:.4f}", flush=True)


if __name__ == "__main__":
    main()
