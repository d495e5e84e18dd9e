"""How long does spatial_mix take launch by launch, and what does the GPU have to have been doing
before for it to start at its steady duration?  ("clock ramp".)

    python tools/ramp_probe.py [--sources 262144]

Builds the bench.py scene, then runs sequences of back-to-back callbacks after different kinds of
preceding GPU activity and prints the per-callback mix-kernel duration (hipEvents) compactly.
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
    ap.add_argument("--n", type=int, default=160)
    args = ap.parse_args()
    import torch

    import bench

    g = bench.build_gpu_scene(0, args.sources, 65536, 2024, 1.0)
    scene = g["scene"]
    out = torch.zeros((1024, 2), dtype=torch.float32, device="cuda")
    interval = np.float32(1.0) / np.float32(48000)
    span = 9
    step = [0]

    def run(n, label):
        scene.set_profiling(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if step[0] and step[0] % span == 0:
                scene.seek_all(-float(span * 1024) / 48000)
            scene.sample_device(interval, out.data_ptr(), 1024)
            step[0] += 1
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        h = scene.kernel_ms_history(n)
        mix = h[:, 1] * 1e3
        pts = " ".join(f"{mix[i:i + 10].mean():.0f}" for i in range(0, n, 10))
        print(f"{label:44s} wall/cb {wall:.3f} ms | mix us per 10 launches: {pts}", flush=True)
        scene.set_profiling(False)

    def pre(label, fn, seconds=0.25):
        torch.cuda.synchronize()
        time.sleep(1.0)
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < seconds:
            fn()
            k += 1
            if k % 4 == 0:
                torch.cuda.synchronize()
        run(40, f"{label} (+{k})")

    run(args.n, "A right after scene set-up (GPU synth)")
    time.sleep(1.0)
    run(60, "B after 1 s idle")
    big = torch.empty((1 << 28,), dtype=torch.float32, device="cuda")   # 1 GiB
    big2 = torch.empty_like(big)
    a16 = torch.randn((8192, 8192), device="cuda", dtype=torch.bfloat16)
    a32 = torch.randn((8192, 8192), device="cuda", dtype=torch.float32)
    d64 = torch.rand((1 << 24,), device="cuda", dtype=torch.float64)
    pre("C 0.25 s of 1 GiB copies", lambda: big2.copy_(big))
    pre("D 0.25 s of bf16 matmuls", lambda: a16 @ a16)
    pre("E 0.25 s of f32 matmuls", lambda: a32 @ a32)
    pre("F 0.25 s of f64 sin", lambda: torch.sin(d64))
    pre("G 0.25 s of f32 sin on 1 GiB", lambda: torch.sin(big))
    pre("H 0.05 s of bf16 matmuls", lambda: a16 @ a16, 0.05)
    pre("I 1.0 s of bf16 matmuls", lambda: a16 @ a16, 1.0)
    return

    run(args.n, "A right after scene set-up (GPU synth)")
    time.sleep(1.0)
    run(args.n, "B after 1 s idle")
    big = torch.empty((1 << 30,), dtype=torch.float32, device="cuda")   # 4 GiB
    big2 = torch.empty_like(big)
    torch.cuda.synchronize()
    time.sleep(1.0)
    for ms_target in (20, 60, 200):
        t0 = time.perf_counter()
        n_copies = 0
        while (time.perf_counter() - t0) * 1e3 < ms_target:
            big2.copy_(big)
            n_copies += 1
            if n_copies % 8 == 0:
                torch.cuda.synchronize()
        run(60, f"C after ~{ms_target} ms of 4 GiB copies (+{n_copies})")
        time.sleep(1.0)
    a = torch.randn((8192, 8192), device="cuda", dtype=torch.bfloat16)
    torch.cuda.synchronize()
    time.sleep(1.0)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.2:
        (a @ a)
    run(60, "D after ~200 ms of bf16 matmuls")
    time.sleep(1.0)
    # E: the same callbacks, but issued in bursts with host-side gaps (a real-time engine's pattern)
    scene.set_profiling(True)
    for _ in range(40):
        scene.sample_device(interval, out.data_ptr(), 1024)
        step[0] += 1
        torch.cuda.synchronize()
        time.sleep(0.005)
    h = scene.kernel_ms_history(40)
    print("E one callback per 5 ms (synchronised)      mix us:", " ".join(f"{v * 1e3:.0f}" for v in h[::4, 1]), flush=True)


if __name__ == "__main__":
    main()
