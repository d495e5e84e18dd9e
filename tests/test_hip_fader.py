"""Fader (src/fader.rs) played in a Mixer on the HIP path vs the CPU oracle.  GPU only.  Bit-exact
with FramesSignal / Cycle / Constant signals (sqrt is IEEE, every op unfused)."""
import numpy as np
import pytest

from oddio_amd import synth
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

INTERVAL = np.float32(1.0) / np.float32(48000)


def test_fader_smoke_kat_through_mixer():
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=4, max_frames=16)
    fc, f = oa.Fader.new(oa.MonoToStereo(oa.Constant(1.0)))
    control.play(f)
    np.testing.assert_array_equal(mixer.sample_n(np.float32(0.1), 12)[:, 0], np.full(12, 1.0, np.float32))
    fc.fade_to(oa.MonoToStereo(oa.Constant(0.0)), 1.0)
    buf = mixer.sample_n(np.float32(0.1), 12)
    np.testing.assert_array_equal(buf[:, 0], buf[:, 1])
    assert buf[0, 0] == 1.0 and buf[11, 0] == 0.0
    assert abs(buf[5, 0] - np.sqrt(np.float32(0.5))) < 1e-6
    mixer.close()


@pytest.mark.parametrize("stereo", [False, True])
def test_fader_crossfades_match_oracle(stereo):
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=16, max_frames=4096)
    mixer.set_mode(oa.MODE_ORDERED)
    cm = oc.Mixer(2)

    def clip(i, n=60000):
        c = synth.noise_clip(80, i, n)
        return np.stack([c, synth.noise_clip(81, i, n)], axis=1) if stereo else c

    def sig(mod, i, rate=48000, start=0.0, db=None, cycle=False):
        fr = mod.Frames.from_slice(rate, clip(i, 900 if cycle else 60000)) if mod is not oc else oc.Frames(rate, clip(i, 900 if cycle else 60000))
        s = mod.Cycle(fr) if cycle else mod.FramesSignal(fr, start)
        if db is not None:
            s = mod.FixedGain(s, db)
        return s if stereo else mod.MonoToStereo(s)

    fc, f_h = oa.Fader.new(sig(oa, 0))
    f_o = oc.Fader(sig(oc, 0))
    h_h, h_o = control.play(f_h), cm.play(f_o)
    # a plain neighbour, so the fader is not alone in the reduce
    control.play(sig(oa, 9)); cm.play(sig(oc, 9))
    plan = {1: (1, 44100, 0.0, None, False, 0.04),          # starts a fade
            2: (2, 48000, 0.0, -6.0, False, 0.5),           # waits for the running fade ...
            3: (3, 48000, 0.0, None, True, 0.02),           # ... and is replaced by this one before it is used
            9: (4, 22050, 0.01, 3.0, False, 0.1)}
    for cb in range(14):
        if cb in plan:
            i, rate, start, db, cyc, dur = plan[cb]
            fc.fade_to(sig(oa, i, rate, start, db, cyc), dur)
            f_o.fade_to(sig(oc, i, rate, start, db, cyc), dur)
        n = (1024, 1024, 2500, 300, 1024, 4096, 1024, 700, 1024, 1024, 3000, 1024, 1, 1024)[cb]
        a = cm.sample_n(INTERVAL, n)
        b = mixer.sample_n(INTERVAL, n)
        np.testing.assert_array_equal(b, a, err_msg=f"callback {cb}")
        assert len(mixer) == len(cm) == 2                   # Fader::is_finished is always false
    h_h.stop(); h_o.stop()
    np.testing.assert_array_equal(mixer.sample_n(INTERVAL, 512), cm.sample_n(INTERVAL, 512))
    assert len(mixer) == len(cm) == 1
    mixer.close()


def test_fader_errors():
    import oddio_amd as oa
    from oddio_amd._lib import OddioHipError
    control, mixer = oa.Mixer(max_sources=4, max_frames=64)
    fc, f = oa.Fader.new(oa.MonoToStereo(oa.Constant(1.0)))
    with pytest.raises(ValueError):
        fc.fade_to(oa.MonoToStereo(oa.Constant(0.0)), 1.0)          # not played yet
    control.play(f)
    stereo = oa.Frames.from_slice(48000, np.zeros((8, 2), np.float32))
    with pytest.raises(OddioHipError):
        fc.fade_to(oa.FramesSignal(stereo, 0.0), 1.0)               # Fader<T>: same channel layout only
    with pytest.raises(OddioHipError):
        fc.fade_to(oa.MonoToStereo(oa.Constant(0.0)), 0.0)
    assert np.isfinite(mixer.sample_n(INTERVAL, 64)).all()
    mixer.close()


@pytest.mark.parametrize("max_distance", [120.0, 9.0])
def test_fader_as_buffered_spatial_source(max_distance):
    # play_buffered(Fader<FixedGain<FramesSignal>>) with fades while the source moves; a Gain-wrapped
    # target, a Cycle target, a queued command that is replaced; plus a plain seekable neighbour.
    # max_distance 9 m: a ring of 6 060 samples, so the ring write wraps (two Fader::sample calls in one callback,
    # ring.rs:33-38) while fades are running
    import oddio_amd as oa
    control, scene = oa.SpatialScene(max_sources=16, max_frames=2048)
    scene.set_mode(oa.MODE_ORDERED)
    ref = oc.SpatialScene()
    clips = [synth.noise_clip(90, i, 70000) for i in range(5)]

    def chain(mod, i, rate=48000, start=0.0, db=None, cycle=False, gain=False):
        data = clips[i][:700] if cycle else clips[i]
        fr = mod.Frames.from_slice(rate, data) if mod is not oc else oc.Frames(rate, data)
        s = mod.Cycle(fr) if cycle else mod.FramesSignal(fr, start)
        if db is not None:
            s = mod.FixedGain(s, db)
        if gain:
            s = mod.Gain.new(s)[1] if mod is not oc else oc.Gain(s)
        return s

    fc, f_h = oa.Fader.new(chain(oa, 0, db=-2.0))
    f_o = oc.Fader(chain(oc, 0, db=-2.0))
    pos, vel = np.float32([6.0, 1.0, -3.0]), np.float32([-9.0, 0.0, 2.0])
    h_h = control.play_buffered(f_h, oa.SpatialOptions(pos, vel, 0.1), max_distance, 48000, 0.1)
    h_o = ref.play_buffered(f_o, oc.SpatialOptions(pos, vel, 0.1), max_distance, 48000, 0.1)
    control.play(chain(oa, 4), oa.SpatialOptions([2.0, 2.0, 2.0], [1.0, 0.0, 0.0]))
    ref.play(chain(oc, 4), oc.SpatialOptions([2.0, 2.0, 2.0], [1.0, 0.0, 0.0]))
    plan = {1: dict(i=1, rate=44100, dur=0.05),
            2: dict(i=2, gain=True, dur=0.4),            # waits, then is replaced by ...
            3: dict(i=3, cycle=True, db=3.0, dur=0.03),   # ... this one
            8: dict(i=1, start=0.2, dur=0.08),
            10: dict(i=2, rate=22050, dur=0.2)}            # still fading when the small ring wraps again
    for cb in range(13):
        if cb in plan:
            a = dict(plan[cb]); dur = a.pop("dur")
            fc.fade_to(chain(oa, **a), dur)
            f_o.fade_to(chain(oc, **a), dur)
        if cb == 5:
            p2, v2 = np.float32([-3.0, 2.0, 4.0]), np.float32([5.0, 1.0, -1.0])
            h_h.set_motion(p2, v2, False); h_o.set_motion(p2, v2, False)
        n = (1024, 1024, 2048, 300, 1024, 1024, 700, 1024, 1024, 2048, 1024, 1, 1024)[cb]
        ra = ref.sample_n(INTERVAL, n)
        rb = scene.sample_n(INTERVAL, n)
        np.testing.assert_array_equal(rb, ra, err_msg=f"callback {cb}")
        assert scene.len_buffered() == ref.len_buffered() == 1
    scene.close()
