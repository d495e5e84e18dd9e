"""Mixer<[f32;2]> on the HIP path vs the oracle (src/mixer.rs, src/signal.rs:61-91).  GPU only."""
import numpy as np
import pytest

from oddio_amd import synth
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu


def arr(*xs):
    return np.array(xs, dtype=np.float32)


def test_frames_sample_kat_through_mixer():
    # src/frames.rs:269-275 driven through MonoToStereo in a Mixer (rate 1 Hz clip)
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=4, max_frames=16)
    control.play(oa.MonoToStereo(oa.FramesSignal(oa.Frames.from_slice(1, [1.0, 2.0, 3.0, 4.0]), -2.0)))
    for interval, expected in ((0.25, arr(0, 0, 0, 0)), (0.5, arr(0, 0.5, 1.0)), (1.0, arr(1.5, 2.5, 3.5, 2.0, 0.0))):
        out = mixer.sample_n(interval, len(expected))
        np.testing.assert_array_equal(out[:, 0], expected)
        np.testing.assert_array_equal(out[:, 1], expected)
    mixer.close()


def test_mixer_is_stopped_kat():
    # src/mixer.rs:130-147
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=4, max_frames=16)
    handle = control.play(oa.MonoToStereo(oa.FramesSignal(oa.Frames.from_slice(1, [0.0, 0.0]), 0.0)))
    assert not handle.is_stopped()
    mixer.sample_n(0.6, 1)
    assert not handle.is_stopped()
    mixer.sample_n(0.6, 1)
    assert not handle.is_stopped()   # finished, but not noticed until the next scan
    mixer.sample_n(0.0, 1)
    assert handle.is_stopped()
    assert len(mixer) == 0
    mixer.close()


def test_config1_64_sines():
    # BASELINE config 1: 64 MonoToStereo<Sine> in one Mixer<[f32;2]>, 48 kHz, 1024-frame callbacks
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=64, max_frames=1024)
    cm = oc.Mixer(2)
    st = synth.SplitMixStreams(7, 64)
    phase = (st.next_u01() * np.float32(2 * np.pi)).astype(np.float32)
    for k in range(64):
        hz = np.float32(110.0 * 2.0 ** (k / 12.0))
        control.play(oa.MonoToStereo(oa.Sine(phase[k], hz)))
        cm.play(oc.MonoToStereo(oc.Sine(phase[k], hz)))
    for cb in range(3):
        got = oa.run(mixer, 48000, np.zeros((1024, 2), np.float32))
        ref = oc.run(cm, 48000, np.zeros((1024, 2), np.float32))
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    mixer.close()


@pytest.mark.parametrize("mode,n_frames", [(1, 1024), (1, 700), (1, 2500), (0, 1024)])
def test_mixer_frames_sources(mode, n_frames):
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=256, max_frames=4096)
    mixer.set_mode(mode)
    cm = oc.Mixer(2)
    rates = (48000, 44100, 48000, 22050)
    hs, hc = [], []
    for i in range(150):
        clip = synth.noise_clip(9, i, 6000 + 37 * i)
        rate = rates[i % 4]
        db = None if i % 3 else -4.0
        sig = oa.FramesSignal(oa.Frames.from_slice(rate, clip), -0.001 * (i % 5))
        osig = oc.FramesSignal(oc.Frames(rate, clip), -0.001 * (i % 5))
        if db is not None:
            sig, osig = oa.FixedGain(sig, db), oc.FixedGain(osig, db)
        hs.append(control.play(oa.MonoToStereo(sig)))
        hc.append(cm.play(oc.MonoToStereo(osig)))
    for cb in range(5):
        if cb == 2:
            for j in (3, 77):
                hs[j].stop()
                hc[j].stop()
        got = oa.run(mixer, 48000, np.zeros((n_frames, 2), np.float32))
        ref = oc.run(cm, 48000, np.zeros((n_frames, 2), np.float32))
        if mode == 1:
            np.testing.assert_array_equal(got, ref)
        else:
            assert np.abs(got - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30)
        assert len(mixer) == len(cm)
        assert [h.is_stopped() for h in hs] == [h.is_stopped() for h in hc]
    mixer.close()
