"""Downmix<FramesSignal<[f32;2]>> (src/downmix.rs:18-47) played in a SpatialScene on the HIP path vs
the CPU oracle.  GPU only.  Bit-exact in ORDERED mode, including ragged callbacks, where the
reference's Downmix::sample advances the inner clip a whole 256-frame buffer per chunk."""
import numpy as np
import pytest

import scenario
from test_hip_parity import rel_err, run_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_frames", [1024, 300, 1, 700, 1300])
def test_downmix_alone_bit_exact(n_frames):
    spec = scenario.random_spec(500 + n_frames, 3, kinds=("downmix",), clip_len=20000, start=0.05, cube=8.0)
    ref, got, ob, hb = run_pair(spec, n_frames, 5, mode=1)
    assert np.abs(ref).max() > 0
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_downmix_mixed_removal_and_fixed_gain():
    spec = scenario.random_spec(51, 21, kinds=("frames", "downmix", "cycle"), gain_db=(None, -6.0, 2.0, None), clip_len=2500, start=0.0,
                                cube=5.0, cycle_len=200)
    ref, got, ob, hb = run_pair(spec, 1024, 7, mode=1)
    assert len(ob) == len(hb) == 7           # FramesSignal and Downmix sources ran out; the Cycles stay
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_downmix_fast_mode_and_resample_ratio():
    spec = scenario.random_spec(52, 90, kinds=("downmix", "frames"), clip_len=30000, rate=44100, start=0.3)
    ref, got, ob, hb = run_pair(spec, 1024, 3, mode=0, max_sources=128)
    assert rel_err(got, ref) <= 1e-5
    hb.close()


@pytest.mark.parametrize("rate,n_frames", [(48000, 1024), (96000, 1024), (44100, 700), (22050, 1536), (192000, 1024)])
def test_downmix_staged_stereo_windows_bit_exact(rate, n_frames):
    """Round 4: Downmix sources take spatial_mix's staged-window path (interleaved stereo windows, sub-windows of the stage;
    spatial_mix<.., DMX>).  Resample ratios up to 2 stay staged, 192 kHz (ratio 4) falls to the exact per-lane path; clips that
    start and end inside the run (windows clipped at both clip edges), a FixedGain on some, FramesSignal sources beside them."""
    spec = scenario.random_spec(530 + n_frames + rate // 1000, 70, kinds=("downmix", "downmix", "frames"), gain_db=(None, -3.0, None, None, 4.0),
                                clip_len=9000 * rate // 48000, rate=rate, start=-0.02, cube=12.0)
    ref, got, ob, hb = run_pair(spec, n_frames, 9, mode=1, max_sources=128)
    assert np.abs(ref).max() > 0 and len(ob) == len(hb)
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_downmix_stationary_sources_take_the_constant_fract_branch():
    """A source at rest relative to the listener has a resample ratio within EPSILON of 1 (frames.rs:180-187: constant fract): the
    stereo window path implements that branch too.  Some sources move, a motion update stops others mid-run."""
    spec = scenario.random_spec(540, 40, kinds=("downmix",), clip_len=30000, start=0.1, cube=10.0)
    for i, src in enumerate(spec["sources"]):
        if i % 3:
            src["vel"] = np.zeros(3, np.float32)
    events = {2: [("motion", j, spec["sources"][j]["pos"], np.zeros(3, np.float32), False) for j in range(0, 40, 3)]}
    ref, got, ob, hb = run_pair(spec, 1024, 6, mode=1, events=events, max_sources=64)
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_downmix_large_set_ordered_rows_and_fast():
    """Above the serial threshold of ORDERED mode (contribution rows + ordered_sum) and in FAST mode, 3000 Downmix sources."""
    spec = scenario.random_spec(550, 3000, kinds=("downmix", "frames", "downmix"), clip_len=12000, start=0.02)
    clips = {}
    for src in spec["sources"]:                           # (a few shared clips keep the host side small)
        src["clip"] = clips.setdefault(src["kind"], src["clip"])
    ref, got, ob, hb = run_pair(spec, 1024, 3, mode=1, max_sources=3072)
    np.testing.assert_array_equal(got, ref)
    hb.close()
    ref, got, ob, hb = run_pair(spec, 1024, 3, mode=0, max_sources=3072)
    assert rel_err(got, ref) <= 1e-5
    hb.close()


def test_downmix_rejects_mono_clip():
    import oddio_amd as oa
    clip = oa.Frames.from_slice(48000, np.zeros(16, np.float32))
    with pytest.raises(TypeError):
        oa.Downmix(oa.FramesSignal(clip, 0.0))
