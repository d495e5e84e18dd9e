"""Golden fixtures (tests/golden/*.npz, made by tests/golden/gen_golden.py): inputs + expected
outputs of the hot path.  CPU: both restatements must reproduce them bit for bit.  GPU (-m gpu):
the HIP path, through the C ABI, must too (ORDERED mode: bit-exact for FramesSignal/Constant
sources; scenes with Sine sources within 1e-5 relative -- device sinf vs glibc sinf)."""
import glob
import os

import numpy as np
import pytest

import scenario

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(p).startswith("chains_"))      # (chains_*.npz: tests/test_golden_chains.py)
KINDS = {0: "frames", 1: "sine", 2: "constant", 3: "cycle", 4: "downmix"}


def load(path):
    z = np.load(path)
    n = len(z["kind"])
    srcs = []
    for i in range(n):
        kind = KINDS[int(z["kind"][i])]
        s = {"kind": kind, "pos": z["pos"][i], "vel": z["vel"][i], "radius": float(z["radius"][i]),
             "gain_db": None if np.isnan(z["gain_db"][i]) else float(z["gain_db"][i])}
        if kind == "frames":
            s["clip"] = z["clip_data"][z["clip_offsets"][i]:z["clip_offsets"][i + 1]].copy()
            s["rate"] = int(z["rate"][i])
            s["start"] = float(z["start"][i])
        elif kind == "cycle":
            s["clip"] = z["clip_data"][z["clip_offsets"][i]:z["clip_offsets"][i + 1]].copy()
            s["rate"] = int(z["rate"][i])
        elif kind == "downmix":        # interleaved stereo frames
            s["clip"] = z["clip_data"][z["clip_offsets"][i]:z["clip_offsets"][i + 1]].copy().reshape(-1, 2)
            s["rate"] = int(z["rate"][i])
            s["start"] = float(z["start"][i])
        elif kind == "sine":
            s["phase"], s["hz"] = float(z["phase"][i]), float(z["hz"][i])
        else:
            s["value"] = float(z["value"][i])
        srcs.append(s)
    events = {}
    for row in z["ev_motion"]:
        events.setdefault(int(row[0]), []).append(("motion", int(row[1]), row[2:5].astype(np.float32), row[5:8].astype(np.float32), bool(row[8])))
    for row in z["ev_rotation"]:
        events.setdefault(int(row[0]), []).append(("rotation", row[1:5].astype(np.float32)))
    return {"sources": srcs}, events, int(z["n_frames"]), int(z["n_callbacks"]), np.float32(z["interval"]), int(z["postfx"]), z["expected"]


def test_fixtures_exist():
    assert len(GOLDEN) >= 5


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    spec, events, n_frames, n_cb, interval, postfx, expected = load(path)
    ob = scenario.play_all(scenario.OracleBackend(), spec)
    if postfx:
        ob.set_postfx(postfx)
    got = scenario.run_events([ob], spec, n_frames, n_cb, interval=interval, events=events)["oracle"]
    np.testing.assert_array_equal(got, expected)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_reproduces_golden(path):
    spec, events, n_frames, n_cb, interval, postfx, expected = load(path)
    hb = scenario.play_all(scenario.HipBackend(max_sources=16, max_frames=max(n_frames, 1), mode=1), spec)
    if postfx:
        hb.set_postfx(postfx)
    got = scenario.run_events([hb], spec, n_frames, n_cb, interval=interval, events=events)["hip"]
    hb.close()
    if any(s["kind"] == "sine" for s in spec["sources"]):
        assert np.abs(got - expected).max() <= 1e-5 * np.abs(expected).max()
    else:
        np.testing.assert_array_equal(got, expected)
