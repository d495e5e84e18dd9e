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


def _cycle_spec(seed, n_src, lens, rates=(48000,), gains=(None,)):
    spec = scenario.random_spec(seed, n_src, kinds=("cycle",), cycle_len=8)
    for i, src in enumerate(spec["sources"]):
        src["clip"] = scenario.synth.noise_clip(seed, i, lens[i % len(lens)])
        src["rate"] = rates[i % len(rates)]
        src["gain_db"] = gains[i % len(gains)]
    return spec


@pytest.mark.parametrize("n_frames", [1024, 700, 1536, 40])
def test_cycle_tiles_from_the_staged_window_bit_exact(n_frames):
    """Round 4: a tile in which no cursor reaches the clip's last sample is rendered by spatial_mix from the staged window
    (cycle_scan writes its record); tiles that touch the clip's end keep the row path.  Clips of many lengths -- far longer than
    a tile (almost every tile staged), a few tiles long (wraps in some tiles, sometimes between the ears), shorter than a tile
    (rows always) -- in one set, 200 sources (more than one wavefront of the scan at every packing)."""
    spec = _cycle_spec(71, 200, lens=(48000, 5000, 1500, 2049, 600, 96000, 513, 24001), gains=(None, -4.5, None))
    ref, got, ob, hb = run_pair(spec, n_frames, 9, mode=1, max_sources=256)
    assert np.abs(ref).max() > 0
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_cycle_resampled_clips_sub_windows_and_motion_bit_exact():
    """96 / 192 kHz loops in a 48 kHz scene (windows of 2 and 4 stage sub-windows), 44.1 kHz, sources that jump (a
    discontinuity moves the delay, hence the cursor, by thousands of samples: both ears on different laps)."""
    spec = _cycle_spec(72, 96, lens=(96000, 30000, 200000, 44100, 7000), rates=(96000, 48000, 192000, 44100))
    rng = np.random.default_rng(5)
    events = {}
    for cb in (1, 2, 4):
        events[cb] = [("motion", j, (spec["sources"][j]["pos"] + rng.normal(size=3).astype(np.float32) * 30).astype(np.float32), spec["sources"][j]["vel"],
                       cb == 2) for j in range(0, 96, 7)]
    events.setdefault(3, []).append(("rotation", [np.cos(0.4), 0.0, np.sin(0.4), 0.0]))
    ref, got, ob, hb = run_pair(spec, 1024, 7, mode=1, events=events, max_sources=128)
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_cycle_staged_tiles_fast_mode_and_large_set():
    """FAST mode (the fused arithmetic of the staged path applies to Cycle tiles now) and a set large enough for the 16-lane
    packing of the scan (> 8192 Cycle sources), ORDERED rows path above the serial threshold."""
    os.environ["ODDIO_HIP_MAX_CYCLE"] = "10000"
    try:
        spec = _cycle_spec(73, 9000, lens=(48000, 5000, 20000))
        clips = {}
        for src in spec["sources"]:                       # (9000 sources share 3 clips: the oracle side stays fast)
            src["clip"] = clips.setdefault(len(src["clip"]), src["clip"])
        ref, got, ob, hb = run_pair(spec, 1024, 3, mode=1, max_sources=10000)
        np.testing.assert_array_equal(got, ref)
        hb.close()
        ref, got, ob, hb = run_pair(spec, 1024, 3, mode=0, max_sources=10000)
        assert rel_err(got, ref) <= 1e-5
        hb.close()
    finally:
        del os.environ["ODDIO_HIP_MAX_CYCLE"]


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
