#!/usr/bin/env python
"""ms per callback of the headline scene in its three sum modes (FAST, TRACKED, ORDERED) by callback length (what each mode promises
about the output is tested in tests/test_hip_large_scene.py, not here).  `python tools/modes_by_callback.py [sources [n,n,..]]`."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch

    import oddio_amd as oa
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    g = bench.build_gpu_scene(0, S, 65536, 2024, 1.0)
    scene = g["scene"]
    interval = np.float32(1.0) / np.float32(bench.RATE)
    out = torch.zeros((1024, 2), dtype=torch.float32, device="cuda")
    modes = ((oa.MODE_FAST, "FAST"), (oa.MODE_TRACKED, "TRACKED"), (oa.MODE_ORDERED, "ORDERED"))
    keep = os.environ.get("MODES")          # e.g. MODES=FAST,TRACKED
    if keep:
        modes = tuple(m for m in modes if m[1] in keep.split(","))
    reps = int(os.environ.get("REPS", "8"))     # (callbacks per timing; the clips last 64 callbacks of 1024 frames)
    print(f"{S} sources; ms per callback ({reps} callbacks enqueued back to back after 3 untimed)")
    sizes = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (128, 256, 512, 768, 1024)
    for n in sizes:
        rewind = lambda k: scene.seek_all(-float(k * n) / bench.RATE)
        ms = {}
        for mode, name in modes:
            scene.set_mode(mode)
            for _ in range(3):
                scene.sample_device(interval, out.data_ptr(), n)
            rewind(3)
            scene.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                scene.sample_device(interval, out.data_ptr(), n)
            scene.synchronize()
            ms[name] = (time.perf_counter() - t0) / reps * 1e3
            rewind(reps)
        print(f"{n:5d} frames: " + "  ".join(f"{name} {v:.4f}" for name, v in ms.items()) + " ms", flush=True)
        assert len(scene) == S


if __name__ == "__main__":
    main()
