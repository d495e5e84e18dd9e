"""Golden fixtures of the buffered path and the filters (tests/golden/chains_*.npz, made by tests/golden/gen_golden_chains.py):
filter chains (FixedGain / Gain / Speed) around FramesSignals in a SpatialScene -- played with play_buffered (rings) or play -- and
in a Mixer, with control stores, motion updates and a rotation on the way.  CPU: both restatements reproduce them bit for bit.
GPU (-m gpu): the HIP path, through the C ABI, does too in ORDERED mode, and stays within 1e-5 of them in FAST mode."""
import glob
import os

import numpy as np
import pytest

import chains

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "chains_*.npz")))
IDS = [os.path.basename(p)[:-4] for p in GOLDEN]


def test_fixtures_exist():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
@pytest.mark.parametrize("backend", ["c", "numpy"])
def test_restatements_reproduce_golden(path, backend):
    spec, expected = chains.load(path)
    b = chains.CBackend(spec) if backend == "c" else chains.NumpyBackend(spec)
    np.testing.assert_array_equal(chains.run(b, spec), expected)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_hip_reproduces_golden_ordered(path):
    spec, expected = chains.load(path)
    hb = chains.HipBackend(spec, mode=1)
    got = chains.run(hb, spec)
    hb.close()
    np.testing.assert_array_equal(got, expected)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_hip_fast_mode_within_tolerance_of_golden(path):
    spec, expected = chains.load(path)
    hb = chains.HipBackend(spec, mode=0)
    got = chains.run(hb, spec)
    hb.close()
    assert np.abs(got - expected).max() <= 1e-5 * np.abs(expected).max()
