"""Host-side logic of bench.py and the profiling helpers that needs no GPU: the algorithmic-byte formulas the roofline figures are
computed from (the buffered path's: DESIGN.md 4.4), and the PMC -> JSON conversion."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_seek_algorithmic_bytes_is_the_surveys_formula():
    import bench
    # B = S * (4 * (N * r + 32) + P) + 8 * N with r = 1, P = 128: 4 352 B per source and callback at N = 1024
    assert bench.algorithmic_bytes(1, 1024) == 4352 + 8192
    assert bench.algorithmic_bytes(262144, 1024) == 262144 * 4352 + 8192


def test_buffered_algorithmic_bytes():
    import bench
    speeds = np.array([1.0, 0.9, 1.1, 1.0], dtype=np.float32)
    b = bench.buffered_algorithmic_bytes(speeds, 1024)
    s = float(speeds.astype(np.float64).sum())
    assert b["leaf_read"] == 4.0 * 1024 * s
    assert b["ring_write"] == 4 * 1024 * 4 and b["ring_read"] == 4 * (1024 + 32) * 4 and b["parameters"] == 128 * 4
    assert b["total"] == b["leaf_read"] + b["ring_write"] + b["ring_read"] + b["parameters"] + 8 * 1024
    assert abs(b["write_kernel"] + b["read_kernel"] - b["total"]) < 1e-6


def test_make_pmc_json_both_workloads(tmp_path):
    kernels = {
        "spatial_mix<true, false, true, false>": {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 10.0, "_dispatches": 20},
        "spatial_mix<true, true, false, false>": {"FETCH_SIZE": 1100.0, "WRITE_SIZE": 2000.0, "_dispatches": 5},
        "ordered_sum": {"FETCH_SIZE": 1050.0, "WRITE_SIZE": 1.0, "_dispatches": 5},
        "buffered_walk": {"FETCH_SIZE": 50.0, "WRITE_SIZE": 60.0, "_dispatches": 8},
        "buffered_write": {"FETCH_SIZE": 500.0, "WRITE_SIZE": 1000.0, "_dispatches": 8},
        "spatial_mix<true, false, true, true>": {"FETCH_SIZE": 520.0, "WRITE_SIZE": 8.0, "_dispatches": 8},
    }
    summary = tmp_path / "summary.json"
    summary.write_text(json.dumps(kernels))
    out = tmp_path / "seek.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_pmc_json.py"), str(summary), "262144", str(out)])
    j = json.load(open(out))
    assert j["kernel"] == "spatial_mix<true, false, true, false>"                       # the FAST instantiation, not the RING one
    assert j["hbm_bytes_per_launch"] == (2 * 1000.0 + 10.0) * 1024                      # read side x2 (gfx950 FETCH_SIZE correction)
    assert j["ordered_hbm_bytes_per_callback"] == (2 * 1100.0 + 2000.0 + 2 * 1050.0 + 1.0) * 1024
    outb = tmp_path / "buffered.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_pmc_json.py"), str(summary), "262144", str(outb), "buffered"])
    jb = json.load(open(outb))
    assert set(jb["kernels"]) == {"walk", "write", "reads"}
    assert jb["kernels"]["reads"]["kernel"] == "spatial_mix<true, false, true, true>"
    assert jb["hbm_bytes_per_callback"] == ((2 * 50 + 60) + (2 * 500 + 1000) + (2 * 520 + 8)) * 1024.0
