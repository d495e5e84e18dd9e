"""Each source of tests/test_hip_pair_kernel.py's scene ALONE through spatial_mix_pair (MODE_FAST_UNFUSED) against the oracle: which kinds differ?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ODDIO_HIP_PAIR_MIN_GROUPS"] = "1"
os.environ["ODDIO_HIP_PAIR"] = os.environ.get("ODDIO_HIP_PAIR", "1")
import test_hip_pair_kernel as T
import oddio_amd as oa
from oracle import oracle_c as oc
srcs = T._sources(400, 100, with_sine=False)
for j, s in enumerate(srcs):
    control, scene = oa.SpatialScene(max_sources=8, max_frames=1024)
    scene.set_mode(oa.MODE_FAST_UNFUSED)
    control.play(T._hip_signal(oa, s), oa.SpatialOptions(s["pos"], s["vel"], s["radius"]))
    ref = oc.SpatialScene()
    ref.play(T._oracle_signal(s), oc.SpatialOptions(s["pos"], s["vel"], s["radius"]))
    bad = []
    for cb in range(2):
        g = scene.sample_n(T.INTERVAL, 1024); w = ref.sample_n(T.INTERVAL, 1024)
        if not np.array_equal(g, w):
            bad.append((cb, float(np.abs(g - w).max()), float(np.abs(w).max())))
    scene.close()
    if bad:
        print(j, j % 12, s["kind"], "reinhard" if s.get("reinhard") else "", s["gain_db"], bad)
print("done")
