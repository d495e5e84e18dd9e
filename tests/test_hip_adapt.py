"""Adapt (src/adapt.rs:63-87) as a device epilogue of SpatialScene / Mixer vs the CPU oracle.
GPU only.  The filter is a rounding-exact recurrence: in ORDERED mode with FramesSignal/Constant
sources the whole chain scene -> Adapt (-> Reinhard) must be bit-exact; `alpha` comes from the
host's expf like the reference's."""
import numpy as np
import pytest

import scenario
from test_hip_parity import rel_err

pytestmark = pytest.mark.gpu

INTERVAL = np.float32(1.0) / np.float32(48000)


def pair(spec, mode=1, max_frames=4096):
    ob = scenario.play_all(scenario.OracleBackend(), spec)
    hb = scenario.play_all(scenario.HipBackend(max_sources=len(spec["sources"]) + 8, max_frames=max_frames, mode=mode), spec)
    return ob, hb


@pytest.mark.parametrize("postfx", [0, 1, 2])
def test_scene_adapt_bit_exact(postfx):
    import oddio_amd as oa
    from oracle import oracle_c as oc
    spec = scenario.random_spec(70, 24, kinds=("frames", "frames", "constant"), gain_db=(None, 12.0, -20.0), clip_len=9000, cube=6.0, start=0.0)
    ob, hb = pair(spec)
    rms0 = 1e-3 / np.sqrt(np.float32(2.0))
    top_o = oc.Adapt(ob.scene, rms0, oc.AdaptOptions(max_gain=1e6))
    top_h = oa.Adapt(hb.scene, rms0, oa.AdaptOptions(max_gain=1e6))
    if postfx:
        top_o = (oc.Reinhard, oc.Tanh)[postfx - 1](top_o)
        top_h = (oa.Reinhard, oa.Tanh)[postfx - 1](top_h)
    for n in (1024, 1024, 300, 1, 2500, 1024, 1024, 1024):     # clips end midway: level drops, gain recovers
        ref = top_o.sample_n(INTERVAL, n)
        got = top_h.sample_n(INTERVAL, n)
        assert np.isfinite(ref).all()
        if postfx == 2:
            assert rel_err(got, ref) <= 1e-5            # device tanhf vs glibc
        else:
            np.testing.assert_array_equal(got, ref)
    hb.close()


def test_scene_adapt_default_options_silence_is_nan_like_reference():
    # max_gain = inf (the default) on exact silence: 0 * inf = NaN in the reference too (adapt.rs:75-81)
    import oddio_amd as oa
    from oracle import oracle_c as oc
    control, scene = oa.SpatialScene(max_sources=8, max_frames=256)
    top_h = oa.Adapt(scene, 0.0)
    top_o = oc.Adapt(oc.SpatialScene(), 0.0)
    np.testing.assert_array_equal(top_h.sample_n(INTERVAL, 256), top_o.sample_n(INTERVAL, 256))
    scene.close()


def test_adapt_can_be_removed_and_restarted():
    import oddio_amd as oa
    from oracle import oracle_c as oc
    spec = scenario.random_spec(71, 6, kinds=("frames",), clip_len=30000, cube=4.0, start=0.0)
    ob, hb = pair(spec)
    plain_o = ob.scene.sample_n(INTERVAL, 512)
    hb.scene.set_adapt(True, 0.3)
    hb.scene.set_adapt(False)
    np.testing.assert_array_equal(hb.scene.sample_n(INTERVAL, 512), plain_o)
    top_o = oc.Adapt(ob.scene, 0.05, oc.AdaptOptions(tau=0.02, max_gain=50.0, low=0.2, high=0.4))
    top_h = oa.Adapt(hb.scene, 0.05, oa.AdaptOptions(tau=0.02, max_gain=50.0, low=0.2, high=0.4))
    for _ in range(3):
        np.testing.assert_array_equal(top_h.sample_n(INTERVAL, 1024), top_o.sample_n(INTERVAL, 1024))
    hb.close()


def test_mixer_adapt_example_chain():
    # examples/adapt.rs: Adapt::new(mixer, 1e-3/sqrt2, {tau 0.1, max_gain 1e6, ..}) over quiet then loud sines
    import oddio_amd as oa
    from oracle import oracle_c as oc
    rate, block = 44100, 512
    interval = np.float32(1.0) / np.float32(rate)
    mc_o = oc.Mixer(2)
    mc_h, m_h = oa.Mixer(max_sources=8, max_frames=block)
    m_h.set_mode(oa.MODE_ORDERED)
    rms0 = 1e-3 / np.sqrt(np.float32(2.0))
    top_o = oc.Adapt(mc_o, rms0, oc.AdaptOptions(max_gain=1e6))
    top_h = oa.Adapt(m_h, rms0, oa.AdaptOptions(max_gain=1e6))
    worst = 0.0
    for stage, (hz, db) in enumerate(((5e2, -60.0), (4e2, -2.0))):
        mc_o.play(oc.MonoToStereo(oc.FixedGain(oc.Sine(0.0, hz), db)))
        mc_h.play(oa.MonoToStereo(oa.FixedGain(oa.Sine(0.0, hz), db)))
        for _ in range(20):
            ref = top_o.sample_n(interval, block)
            got = top_h.sample_n(interval, block)
            worst = max(worst, rel_err(got, ref))
    assert worst <= 2e-5          # device sinf feeds a long-memory filter; see test_sine_sources_tolerance
    m_h.close()
