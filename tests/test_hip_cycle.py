"""Cycle sources in the Seek set of a SpatialScene (`SpatialSceneControl::play(Cycle::new(..))`,
src/cycle.rs:26-61 driven by src/spatial.rs:446-468) on the HIP path vs the CPU oracle.  GPU only.

A Cycle's cursor is a rounding chain through ears and chunks; the device renders it serially and
adds the row in set order, so ORDERED mode must be bit-exact like FramesSignal sources.
"""
import os

import numpy as np
import pytest

import scenario
from test_hip_parity import rel_err, run_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cycle_len,n_frames", [(1000, 1024), (37, 1024), (3, 300), (1, 256), (48000, 1300), (5000, 1)])
def test_cycle_alone_bit_exact(cycle_len, n_frames):
    spec = scenario.random_spec(300 + cycle_len, 3, kinds=("cycle",), cycle_len=cycle_len)
    ref, got, ob, hb = run_pair(spec, n_frames, 5, mode=1)
    assert cycle_len == 1 or np.abs(ref).max() > 0
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_cycle_mixed_with_frames_ordered_bit_exact():
    spec = scenario.random_spec(41, 40, kinds=("frames", "cycle", "constant", "frames", "cycle"), gain_db=(None, -6.0, None, 3.0),
                                cycle_len=777)
    rng = np.random.default_rng(1)
    events = {}
    for cb in (1, 3):
        evs = []
        for j in (1, 4, 6, 11):
            p = (spec["sources"][j]["pos"] + rng.normal(size=3).astype(np.float32)).astype(np.float32)
            evs.append(("motion", j, p, spec["sources"][j]["vel"], cb == 3 and j == 4))
        events[cb] = evs
    events.setdefault(2, []).append(("rotation", [np.cos(0.2), 0.0, np.sin(0.2), 0.0]))
    ref, got, ob, hb = run_pair(spec, 1024, 6, mode=1, events=events)
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_cycle_survives_swap_remove_of_neighbours():
    # short FramesSignal clips finish and are swap_removed (set.rs:170-188): Cycle slots move, their
    # contribution rows must follow them
    spec = scenario.random_spec(43, 24, kinds=("frames", "cycle", "frames"), clip_len=2000, start=0.0, cube=6.0, cycle_len=300)
    ref, got, ob, hb = run_pair(spec, 1024, 8, mode=1)
    assert len(ob) == len(hb) == 8          # only the Cycles are left
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_cycle_fast_mode_tolerance_and_late_play():
    spec = scenario.random_spec(44, 300, kinds=("frames", "cycle", "sine"), cycle_len=4096)
    extra = scenario.random_spec(45, 2, kinds=("cycle",), cycle_len=100)["sources"]
    events = {2: [("play", extra[0])], 3: [("play", extra[1])]}
    ref, got, ob, hb = run_pair(spec, 1024, 5, mode=0, events=events, max_sources=512)
    assert rel_err(got, ref) <= 1e-5
    hb.close()


def test_cycle_row_capacity_error():
    import oddio_amd as oa
    from oddio_amd._lib import OddioHipError
    os.environ["ODDIO_HIP_MAX_CYCLE"] = "2"
    try:
        control, scene = oa.SpatialScene(max_sources=16, max_frames=256)
        clip = oa.Frames.from_slice(48000, np.arange(8, dtype=np.float32))
        for _ in range(2):
            control.play(oa.Cycle(clip), oa.SpatialOptions((1.0, 0.0, 0.0)))
        with pytest.raises(OddioHipError) as ei:
            control.play(oa.Cycle(clip), oa.SpatialOptions((1.0, 0.0, 0.0)))
        assert ei.value.code == -2               # ODDIO_HIP_ENOMEM
        out = scene.sample_n(np.float32(1.0) / np.float32(48000), 256)
        assert np.isfinite(out).all()
        scene.close()
    finally:
        del os.environ["ODDIO_HIP_MAX_CYCLE"]
