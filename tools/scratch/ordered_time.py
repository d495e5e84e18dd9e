import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, oddio_amd as oa
for S in (4096, 65536, 262144):
    g = bench.build_gpu_scene(0, S, 65536, 2024, 1.0)
    scene = g["scene"]
    scene.set_mode(oa.MODE_ORDERED)
    out = torch.zeros((1024, 2), dtype=torch.float32, device="cuda")
    interval = np.float32(1.0) / np.float32(48000)
    scene.sample_device(interval, out.data_ptr(), 1024); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        scene.sample_device(interval, out.data_ptr(), 1024)
    torch.cuda.synchronize()
    print(S, "ORDERED ms/callback", (time.perf_counter() - t0) / 5 * 1e3, flush=True)
    scene.close(); del g
