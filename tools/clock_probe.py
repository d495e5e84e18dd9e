#!/usr/bin/env python
"""Clocks and power while the headline workload runs back to back (GPU box): is spatial_mix held back by power management?
Samples `rocm-smi` while one process enqueues callbacks for ~6 s (the sources are put back every 16 callbacks, so the workload
stays BASELINE's)."""
import os
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showperflevel"], capture_output=True, text=True, timeout=20)
        keep = [ln.strip() for ln in r.stdout.splitlines() if any(k in ln for k in ("sclk", "mclk", "fclk", "Power", "Temperature (Sensor junction)", "Performance Level"))]
        return " | ".join(keep)
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi failed: {e}"


def main():
    import torch
    print("idle:", smi(), flush=True)
    g = bench.build_gpu_scene(0, 262144, 65536, 2024, 1.0)
    scene = g["scene"]
    out = torch.zeros((bench.N_FRAMES, 2), dtype=torch.float32, device="cuda")
    interval = np.float32(1.0) / np.float32(bench.RATE)
    stop = False
    samples = []

    def sampler():
        while not stop:
            samples.append((time.perf_counter(), smi()))
            time.sleep(0.7)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    control = g["control"]
    while time.perf_counter() - t0 < 6.0:
        for _ in range(16):
            scene.sample_device(interval, out.data_ptr(), bench.N_FRAMES)
            n += 1
        # hold BASELINE's workload: every source back to where it starts (cursor, position, velocity), in stream order
        control.set_motion_batch(g["ids"], g["spec"]["position"], g["spec"]["velocity"], True)
        scene.seek_all(-16.0 * bench.N_FRAMES / bench.RATE)
        scene.sample_device(interval, out.data_ptr(), 0)
        if n % 64 == 0:
            scene.synchronize()
    scene.synchronize()
    el = time.perf_counter() - t0
    stop = True
    th.join()
    print(f"{n} callbacks in {el:.2f} s = {el / n * 1e3:.4f} ms/callback")
    for t, s_ in samples:
        print(f"t={t - t0:5.2f}s  {s_}")


if __name__ == "__main__":
    main()
