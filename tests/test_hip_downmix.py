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


def test_downmix_fast_mode_renders_the_mono_sum_of_the_clip():
    """Round 6: in FAST mode (fused kernels, 1e-5 contract) a Downmix source over a library-owned stereo clip is rendered as a plain
    clip source over L + R, which the clip carries behind its frames (device_types.h downmix_presum_offset): lerp(L + R) against the
    reference's lerp(L) + lerp(R).  Anti-correlated channels (R = -0.9 L + ..: the sum is a sixth of either channel) are the hard
    case for that identity; 300 sources, three rates, clips that begin and end inside the run; ORDERED stays the oracle's bits."""
    import oddio_amd as oa
    from oracle import oracle_c as oc
    from oddio_amd import synth
    rng = np.random.default_rng(7)
    sc = synth.make_scene(77, 300, cube=10.0)
    interval = np.float32(1.0) / np.float32(48000)
    scenes = {}
    for name, mode in (("fast", oa.MODE_FAST), ("ordered", oa.MODE_ORDERED)):
        control, scene = oa.SpatialScene(max_sources=512, max_frames=1024)
        scene.set_mode(mode)
        scenes[name] = (control, scene)
    ref = oc.SpatialScene()
    for i in range(300):
        n = 6000 + 37 * i
        left = synth.noise_clip(5, i, n)
        right = (-0.9 * left + 0.1 * synth.noise_clip(6, i, n)).astype(np.float32) if i % 3 else synth.noise_clip(8, i, n)
        st = np.stack([left, right], axis=1)
        rate = (48000, 44100, 32000)[i % 3]
        start = -0.01 if i % 5 == 0 else 0.04
        for control, _ in scenes.values():
            sig = oa.Downmix(oa.FramesSignal(oa.Frames.from_slice(rate, st), start))
            control.play(oa.FixedGain(sig, -2.0) if i % 4 == 0 else sig, oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
        sig = oc.Downmix(oc.FramesSignal(oc.Frames.from_slice(rate, st), start))
        ref.play(oc.FixedGain(sig, -2.0) if i % 4 == 0 else sig, oc.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
    worst = 0.0
    for cb in range(8):
        n = 1024 if cb != 5 else 700
        want = ref.sample_n(interval, n)
        np.testing.assert_array_equal(scenes["ordered"][1].sample_n(interval, n), want)
        got = scenes["fast"][1].sample_n(interval, n)
        worst = max(worst, float(np.abs(got - want).max() / np.abs(want).max()))
    assert len(scenes["fast"][1]) == len(scenes["ordered"][1]) == len(ref)
    print("Downmix through the mono sum, FAST: worst |gpu - reference| / max|reference| =", worst)
    assert worst <= 1e-5
    for _, scene in scenes.values():
        scene.close()
