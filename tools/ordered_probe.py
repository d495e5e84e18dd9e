#!/usr/bin/env python
"""Stage times of the bit-exact (ORDERED) mode at the headline size: prepass / spatial_mix<STORE> / ordered_sum,
from the scene's own hipEvents.  `ODDIO_HIP_LIB=variant.so python tools/ordered_probe.py [sources]`."""
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
    out = torch.zeros((bench.N_FRAMES, 2), dtype=torch.float32, device="cuda")
    interval = np.float32(1.0) / np.float32(bench.RATE)
    for mode, name in ((oa.MODE_FAST, "fast"), (oa.MODE_ORDERED, "ordered")):
        scene.set_mode(mode)
        rewind = lambda k: scene.seek_all(-float(k * bench.N_FRAMES) / bench.RATE)   # the clips last 17 callbacks from their start offset
        for _ in range(3):
            scene.sample_device(interval, out.data_ptr(), bench.N_FRAMES)
        rewind(3)
        scene.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            scene.sample_device(interval, out.data_ptr(), bench.N_FRAMES)
        scene.synchronize()
        wall_plain = (time.perf_counter() - t0) / 8 * 1e3
        rewind(8)
        scene.set_profiling(1)
        scene.synchronize()
        t0 = time.perf_counter()
        n = 8
        for _ in range(n):
            scene.sample_device(interval, out.data_ptr(), bench.N_FRAMES)
        scene.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        st = scene.kernel_ms_history(n)
        scene.set_profiling(False)
        rewind(8)
        assert len(scene) == S, "sources finished inside the probe"
        print(f"{os.path.basename(os.environ.get('ODDIO_HIP_LIB', 'libodd_hip.so'))} {name}: wall {wall_plain:.4f} ms/callback without events, {wall:.4f} with; "
              f"prepass {st[:, 0].mean():.4f}  mix {st[:, 1].mean():.4f}  sum/reduce {st[:, 2].mean():.4f}", flush=True)


if __name__ == "__main__":
    main()
