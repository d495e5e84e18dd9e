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


def _big_mixer(n_src, max_frames, seed=21):
    """n_src MonoToStereo sources of every fast-path kind (FramesSignal at four rates, some under FixedGain, some that run out
    of clip; Sine; Constant) in a HIP Mixer and the oracle's; a few clips shared (the host side stays small)."""
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=n_src + 8, max_frames=max_frames)
    cm = oc.Mixer(2)
    rates = (48000, 44100, 96000, 22050)
    clips = [synth.noise_clip(seed, k, 3000 + 997 * k) for k in range(12)]
    frames = {}
    hs, hc = [], []
    for i in range(n_src):
        kind = i % 7
        if kind == 5:
            sig, osig = oa.Sine(0.1 * i, 100.0 + 0.37 * i), oc.Sine(0.1 * i, 100.0 + 0.37 * i)
        elif kind == 6:
            sig, osig = oa.Constant(0.001 * (i % 50) - 0.02), oc.Constant(0.001 * (i % 50) - 0.02)
        else:
            k, rate = i % 12, rates[i % 4]
            if (k, rate) not in frames:
                frames[(k, rate)] = (oa.Frames.from_slice(rate, clips[k]), oc.Frames(rate, clips[k]))
            start = -0.002 * (i % 4) + 0.01 * (i % 3)
            sig, osig = oa.FramesSignal(frames[(k, rate)][0], start), oc.FramesSignal(frames[(k, rate)][1], start)
        if i % 5 == 0 and kind != 6:
            sig, osig = oa.FixedGain(sig, -3.0 - (i % 4)), oc.FixedGain(osig, -3.0 - (i % 4))
        hs.append(control.play(oa.MonoToStereo(sig)))
        hc.append(cm.play(oc.MonoToStereo(osig)))
    return mixer, cm, hs, hc


@pytest.mark.parametrize("n_src,n_frames", [(3000, 1024), (1500, 700), (1100, 2500)])
def test_large_mixer_ordered_rows_path_bit_exact(n_src, n_frames):
    """Round 4: above the serial threshold (1024 sources) ORDERED mode writes every source's contribution to its rows
    (mixer_mix<.., STORE>) and ordered_sum adds them in the reference's order (mixer.rs:100-117): bit-exact, with sources that
    finish or are stopped on the way (rows of zeros, then removal) and the post filter."""
    mixer, cm, hs, hc = _big_mixer(n_src, 4096)
    mixer.set_mode(1)
    mixer.set_postfx(1)
    ref_top = oc.Reinhard(cm)
    for cb in range(6):
        if cb == 2:
            for j in range(5, n_src, 97):
                hs[j].stop()
                hc[j].stop()
        got = mixer.sample_n(np.float32(1.0) / np.float32(48000), n_frames)
        ref = ref_top.sample_n(np.float32(1.0) / np.float32(48000), n_frames)
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()          # (Sine sources: device sinf)
        assert len(mixer) == len(cm)
    mixer.close()


@pytest.mark.parametrize("n_src,n_frames", [(2000, 1024), (1300, 1536)])
def test_large_mixer_ordered_rows_path_exact_kinds_bit_exact(n_src, n_frames):
    """The same without Sine sources (the one kind whose samples differ from glibc's by an ulp): every bit."""
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=n_src, max_frames=2048)
    cm = oc.Mixer(2)
    mixer.set_mode(1)
    clips = [synth.noise_clip(5, k, 2500 + 613 * k) for k in range(9)]
    fr = [(oa.Frames.from_slice(r, c), oc.Frames(r, c)) for c in clips for r in (48000, 44100)]
    hs, hc = [], []
    for i in range(n_src):
        if i % 6 == 5:
            sig, osig = oa.Constant(0.01 * (i % 7)), oc.Constant(0.01 * (i % 7))
        else:
            f = fr[i % len(fr)]
            sig, osig = oa.FramesSignal(f[0], 0.004 * (i % 5)), oc.FramesSignal(f[1], 0.004 * (i % 5))
            if i % 4 == 1:
                sig, osig = oa.FixedGain(sig, -2.5), oc.FixedGain(osig, -2.5)
        hs.append(control.play(oa.MonoToStereo(sig)))
        hc.append(cm.play(oc.MonoToStereo(osig)))
    for cb in range(7):
        if cb == 3:
            for j in range(0, n_src, 41):
                hs[j].stop()
                hc[j].stop()
        got = mixer.sample_n(np.float32(1.0) / np.float32(48000), n_frames)
        ref = cm.sample_n(np.float32(1.0) / np.float32(48000), n_frames)
        np.testing.assert_array_equal(got, ref)
        assert len(mixer) == len(cm)
    assert len(mixer) < n_src            # clips ran out on the way
    mixer.close()


@pytest.mark.parametrize("mode", [1, 0])
def test_thousands_of_mixer_sources_finish_in_one_callback(mode):
    """3000 sources on one 1500-sample clip run out in the same callback (more ids than the stopped list's first copy holds),
    500 more live on: set sizes, the order of the survivors (ORDERED: every bit) and the handles' flags follow the reference."""
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=3600, max_frames=1024)
    cm = oc.Mixer(2)
    mixer.set_mode(mode)
    short, long_ = synth.noise_clip(2, 0, 1500), synth.noise_clip(2, 1, 9000)
    fs, fl = (oa.Frames.from_slice(48000, short), oc.Frames(48000, short)), (oa.Frames.from_slice(48000, long_), oc.Frames(48000, long_))
    hs, hc = [], []
    for i in range(3500):
        f = fl if i % 7 == 3 else fs
        hs.append(control.play(oa.MonoToStereo(oa.FramesSignal(f[0], 0.0))))
        hc.append(cm.play(oc.MonoToStereo(oc.FramesSignal(f[1], 0.0))))
    for cb in range(5):
        got = mixer.sample_n(np.float32(1.0) / np.float32(48000), 1024)
        ref = cm.sample_n(np.float32(1.0) / np.float32(48000), 1024)
        if mode == 1:
            np.testing.assert_array_equal(got, ref)
        else:
            # (3000 copies of one signal add up coherently: the reference's sequential f32 sum drifts ~4e-5 from the exact sum here,
            # the tree sum does not -- the same effect as at 262 144 incoherent sources)
            assert np.abs(got - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-30)
        assert len(mixer) == len(cm)
        assert [h.is_stopped() for h in hs[::13]] == [h.is_stopped() for h in hc[::13]]
    assert len(mixer) == 500
    mixer.close()


def test_general_path_beyond_1024_sources_bit_exact():
    """The general path (Gain / Speed chains) held at most 1024 sources until round 4: 2500 Gain<MonoToStereo<FramesSignal>> and
    Speed sources, gain and speed stores on the way, ORDERED bit-exact and FAST within tolerance."""
    import oddio_amd as oa
    for mode in (1, 0):
        control, mixer = oa.Mixer(max_sources=2600, max_frames=1024)
        cm = oc.Mixer(2)
        mixer.set_mode(mode)
        clips = [synth.noise_clip(8, k, 5000 + 701 * k) for k in range(6)]
        fr = [(oa.Frames.from_slice(48000, c), oc.Frames(48000, c)) for c in clips]
        gh, gc = [], []
        for i in range(2500):
            f = fr[i % 6]
            if i % 3 == 2:
                hsc, hs_ = oa.Speed.new(oa.FramesSignal(f[0], 0.002 * (i % 5)))
                cs_ = oc.Speed(oc.FramesSignal(f[1], 0.002 * (i % 5)))
                hsc.set_speed(0.9 + 0.001 * (i % 200))
                cs_.set_speed(0.9 + 0.001 * (i % 200))
                control.play(oa.MonoToStereo(hs_))
                cm.play(oc.MonoToStereo(cs_))
            else:
                hgc, hg = oa.Gain.new(oa.MonoToStereo(oa.FramesSignal(f[0], 0.001 * (i % 7))))
                cg = oc.Gain(oc.MonoToStereo(oc.FramesSignal(f[1], 0.001 * (i % 7))))
                control.play(hg)
                cm.play(cg)
                gh.append(hgc)
                gc.append(cg)
        for cb in range(4):
            if cb == 1:
                for k in range(0, len(gh), 3):
                    gh[k].set_amplitude_ratio(0.25 + 0.001 * (k % 300))
                    gc[k].set_amplitude_ratio(0.25 + 0.001 * (k % 300))
            got = mixer.sample_n(np.float32(1.0) / np.float32(48000), 1024)
            ref = cm.sample_n(np.float32(1.0) / np.float32(48000), 1024)
            if mode == 1:
                np.testing.assert_array_equal(got, ref)
            else:
                assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
            assert len(mixer) == len(cm)
        mixer.close()


@pytest.mark.parametrize("mode,n_frames", [(0, 1024), (0, 700), (1, 700)])
def test_general_path_chain_and_other_shapes_together(mode, n_frames):
    """600 sources of three kinds in one mixer: chains over a mono clip (buffered_write; in FAST mode added up by the kernel itself),
    Gain over a STEREO clip and Gain<MonoToStereo<Cycle>> (one wavefront per source, a slab each) -- the row-sum tree takes both kinds
    of rows; sources stop on the way."""
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=640, max_frames=1024)
    cm = oc.Mixer(2)
    mixer.set_mode(mode)
    mono = [synth.noise_clip(12, k, 6000 + 400 * k) for k in range(5)]
    st2 = [np.stack([synth.noise_clip(13, k, 5000), synth.noise_clip(14, k, 5000)], axis=1) for k in range(3)]
    fm = [(oa.Frames.from_slice(48000, c), oc.Frames(48000, c)) for c in mono]
    fs = [(oa.Frames.from_slice(44100, c), oc.Frames(44100, c)) for c in st2]
    fc = [(oa.Frames.from_slice(48000, c[:700]), oc.Frames(48000, c[:700])) for c in mono[:2]]
    hs, hc, gh, gc = [], [], [], []
    for i in range(600):
        if i % 3 == 0:
            h_, c_ = oa.MonoToStereo(oa.FramesSignal(fm[i % 5][0], 0.001 * (i % 9))), oc.MonoToStereo(oc.FramesSignal(fm[i % 5][1], 0.001 * (i % 9)))
        elif i % 3 == 1:
            h_, c_ = oa.FramesSignal(fs[i % 3][0], 0.0), oc.FramesSignal(fs[i % 3][1], 0.0)
        else:
            h_, c_ = oa.MonoToStereo(oa.Cycle(fc[i % 2][0])), oc.MonoToStereo(oc.Cycle(fc[i % 2][1]))
        hgc, hg = oa.Gain.new(h_)
        cg = oc.Gain(c_)
        hs.append(control.play(hg))
        hc.append(cm.play(cg))
        gh.append(hgc)
        gc.append(cg)
    iv = np.float32(1.0) / np.float32(48000)
    for cb in range(6):
        if cb == 1:
            for k in range(0, 600, 4):
                gh[k].set_amplitude_ratio(0.3 + 0.002 * (k % 200))
                gc[k].set_amplitude_ratio(0.3 + 0.002 * (k % 200))
        if cb == 3:
            for k in range(5, 600, 31):
                hs[k].stop()
                hc[k].stop()
        got = mixer.sample_n(iv, n_frames)
        ref = cm.sample_n(iv, n_frames)
        if mode == 1:
            np.testing.assert_array_equal(got, ref)
        else:
            assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
        assert len(mixer) == len(cm)
    mixer.close()


@pytest.mark.parametrize("n_src", [5, 40, 64, 100, 700, 3000])
def test_mixer_fast_mode_wave_split_tolerance(n_src):
    """FAST mode at sizes where 2 .. 16 waves share a group of 64 sources (round 4) and beyond: within 1e-5 of the reference."""
    mixer, cm, hs, hc = _big_mixer(n_src, 1024, seed=33)
    for cb in range(3):
        got = mixer.sample_n(np.float32(1.0) / np.float32(48000), 1024)
        ref = cm.sample_n(np.float32(1.0) / np.float32(48000), 1024)
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
        assert len(mixer) == len(cm)
    mixer.close()


# ---- general path: Cycle, Gain / Speed chains, stereo clips (mixer.rs + cycle.rs + gain.rs + speed.rs) ----

CYCLE_FRAMES = [1.0, 2.0, 3.0]


def _cycle_mixer(frames=CYCLE_FRAMES):
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=8, max_frames=64)
    control.play(oa.MonoToStereo(oa.Cycle(oa.Frames.from_slice(1, frames))))
    return mixer


def test_cycle_kats_through_mixer():
    # src/cycle.rs:69-122 (seek-free cases), L == R == the mono cycle
    m = _cycle_mixer()
    np.testing.assert_array_equal(m.sample_n(1.0, 5)[:, 0], arr(1.0, 2.0, 3.0, 1.0, 2.0))            # wrap_single
    m.close()
    m = _cycle_mixer()
    buf = np.concatenate([m.sample_n(1.0, 2), m.sample_n(1.0, 3)])[:, 1]                             # wrap_multi
    np.testing.assert_array_equal(buf, arr(1.0, 2.0, 3.0, 1.0, 2.0))
    m.close()
    m = _cycle_mixer()
    buf = np.concatenate([m.sample_n(0.5, 2), m.sample_n(0.5, 6)])[:, 0]                             # wrap_fract
    np.testing.assert_array_equal(buf, arr(1.0, 1.5, 2.0, 2.5, 3.0, 2.0, 1.0, 1.5))
    m.close()
    m = _cycle_mixer()
    buf = np.concatenate([m.sample_n(10.0, 2), m.sample_n(10.0, 1)])[:, 0]                           # wrap_large_interval
    np.testing.assert_array_equal(buf, arr(1.0, 2.0, 3.0))
    m.close()


def test_gain_smoothing_kat_through_mixer():
    # src/gain.rs:171-179
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=4, max_frames=16)
    gc, g = oa.Gain.new(oa.Constant(1.0))
    control.play(oa.MonoToStereo(g))
    gc.set_amplitude_ratio(5.0)
    np.testing.assert_array_equal(mixer.sample_n(0.025, 6)[:, 0], arr(1.0, 2.0, 3.0, 4.0, 5.0, 5.0))
    np.testing.assert_array_equal(mixer.sample_n(0.025, 6)[:, 1], arr(5.0, 5.0, 5.0, 5.0, 5.0, 5.0))
    mixer.close()


def test_mixer_general_vs_oracle():
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=64, max_frames=4096)
    mixer.set_mode(oa.MODE_ORDERED)
    cm = oc.Mixer(2)
    hs, hc, ctl = [], [], []
    # plain sources first (fast representation), then the ones that force the general path
    for i in range(6):
        clip = synth.noise_clip(13, i, 5000 + 300 * i)
        hs.append(control.play(oa.MonoToStereo(oa.FramesSignal(oa.Frames.from_slice(48000, clip), 0.0))))
        hc.append(cm.play(oc.MonoToStereo(oc.FramesSignal(oc.Frames(48000, clip), 0.0))))
    ref0 = oc.run(cm, 48000, np.zeros((1024, 2), np.float32))
    np.testing.assert_array_equal(oa.run(mixer, 48000, np.zeros((1024, 2), np.float32)), ref0)   # still the fast kernels
    stereo = np.stack([synth.noise_clip(14, 0, 7000), synth.noise_clip(14, 1, 7000)], axis=1)
    gcs, g_h = oa.Gain.new(oa.FramesSignal(oa.Frames.from_slice(44100, stereo), 0.0))
    og = oc.Gain(oc.FramesSignal(oc.Frames(44100, stereo), 0.0))
    hs.append(control.play(g_h)); hc.append(cm.play(og))
    cyc = synth.noise_clip(15, 0, 777)
    scs, s_h = oa.Speed.new(oa.MonoToStereo(oa.FixedGain(oa.Cycle(oa.Frames.from_slice(32000, cyc)), -4.0)))
    osd = oc.Speed(oc.MonoToStereo(oc.FixedGain(oc.Cycle(oc.Frames(32000, cyc)), -4.0)))
    hs.append(control.play(s_h)); hc.append(cm.play(osd))
    gc2, g2 = oa.Gain.new(oa.MonoToStereo(oa.Constant(0.25)))
    og2 = oc.Gain(oc.MonoToStereo(oc.Constant(0.25)))
    hs.append(control.play(g2)); hc.append(cm.play(og2))
    for cb in range(7):
        n = (1024, 700, 2500, 1024, 1024, 1024, 512)[cb]
        if cb == 1:
            gcs.set_gain(-9.0); og.set_gain(-9.0)
            scs.set_speed(1.25); osd.set_speed(1.25)
        if cb == 3:
            gc2.set_amplitude_ratio(3.0); og2.set_amplitude_ratio(3.0)
            hs[2].stop(); hc[2].stop()
        got = oa.run(mixer, 48000, np.zeros((n, 2), np.float32))
        ref = oc.run(cm, 48000, np.zeros((n, 2), np.float32))
        np.testing.assert_array_equal(got, ref)
        assert len(mixer) == len(cm)
        assert [h.is_stopped() for h in hs] == [h.is_stopped() for h in hc]
    mixer.close()


@pytest.mark.parametrize("mode", ["ordered", "fast"])
def test_mixer_general_many_sources(mode):
    """300 filtered sources (wave-per-source kernel + tiled slab sum) beside a few shapes the thread-per-source kernel
    renders; clips of different lengths finish inside the run.  ORDERED: bit-exact; FAST: only the sum order differs."""
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=512, max_frames=1024)
    mixer.set_mode(oa.MODE_ORDERED if mode == "ordered" else oa.MODE_FAST)
    cm = oc.Mixer(2)
    rng = np.random.default_rng(5)
    ctl = []
    for i in range(300):
        clip = synth.noise_clip(21, i, 2500 + 37 * i)
        rate = (48000, 44100, 32000)[i % 3]
        t0 = float(np.float32(rng.uniform(0.0, 0.01)))
        if i % 4 == 0:
            ch, sig = oa.Speed.new(oa.FramesSignal(oa.Frames.from_slice(rate, clip), t0))
            osig = oc.Speed(oc.FramesSignal(oc.Frames(rate, clip), t0))
            gh, sig = oa.Gain.new(oa.MonoToStereo(sig)); og = oc.Gain(oc.MonoToStereo(osig))
            ctl.append((ch, osig, "speed")); ctl.append((gh, og, "gain"))
            control.play(sig); cm.play(og)
        else:
            gh, sig = oa.Gain.new(oa.MonoToStereo(oa.FixedGain(oa.FramesSignal(oa.Frames.from_slice(rate, clip), t0), -3.0)))
            og = oc.Gain(oc.MonoToStereo(oc.FixedGain(oc.FramesSignal(oc.Frames(rate, clip), t0), -3.0)))
            ctl.append((gh, og, "gain"))
            control.play(sig); cm.play(og)
    # shapes that were rendered one thread per source until round 3: Constant leaf, stereo clip, Cycle
    gh, sig = oa.Gain.new(oa.MonoToStereo(oa.Constant(0.125))); og = oc.Gain(oc.MonoToStereo(oc.Constant(0.125)))
    ctl.append((gh, og, "gain")); control.play(sig); cm.play(og)
    stereo = np.stack([synth.noise_clip(22, 0, 9000), synth.noise_clip(22, 1, 9000)], axis=1)
    gh, sig = oa.Gain.new(oa.FramesSignal(oa.Frames.from_slice(44100, stereo), 0.0)); og = oc.Gain(oc.FramesSignal(oc.Frames(44100, stereo), 0.0))
    ctl.append((gh, og, "gain")); control.play(sig); cm.play(og)
    cyc = synth.noise_clip(23, 0, 555)
    sh, sig = oa.Speed.new(oa.MonoToStereo(oa.Cycle(oa.Frames.from_slice(32000, cyc)))); osd = oc.Speed(oc.MonoToStereo(oc.Cycle(oc.Frames(32000, cyc))))
    ctl.append((sh, osd, "speed")); control.play(sig); cm.play(osd)
    interval = np.float32(1.0) / np.float32(48000)
    for cb in range(6):
        if cb in (1, 3):
            for j in range(0, len(ctl), 7):
                h, o, kind = ctl[j]
                if kind == "gain":
                    v = float(np.float32(rng.uniform(0.2, 1.5)))
                    h.set_amplitude_ratio(v); o.set_amplitude_ratio(v)
                else:
                    v = float(np.float32(rng.uniform(0.8, 1.3)))
                    h.set_speed(v); o.set_speed(v)
        n = (1024, 1024, 600, 1024, 1024, 1024)[cb]
        got = mixer.sample_n(interval, n)
        ref = cm.sample_n(interval, n)
        if mode == "ordered":
            np.testing.assert_array_equal(got, ref, err_msg=f"callback {cb}")
        else:
            np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4, err_msg=f"callback {cb}")   # ~300 terms of |x| < 1.5
        assert len(mixer) == len(cm)
    assert len(cm) < 303      # some clips ended inside the run
    mixer.close()


@pytest.mark.parametrize("mode,n_frames", [(1, 1024), (1, 700), (0, 1024)])
def test_mono_mixer_is_mixer_f32(mode, n_frames):
    """Mixer<f32> (mixer.rs:46-81 with `impl Frame for f32`): mono signals, n_frames floats out; Reinhard after it.
    ORDERED: bit-exact vs the oracle's Mixer(channels=1); FAST: only the sum order differs."""
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=256, max_frames=2048, channels=1)
    mixer.set_mode(mode)
    mixer.set_postfx(oa.POSTFX_REINHARD)
    cm = oc.Mixer(1)
    top = oc.Reinhard(cm)
    hs, hc = [], []
    for i in range(120):
        clip = synth.noise_clip(31, i, 3000 + 41 * i)
        rate = (48000, 44100, 22050)[i % 3]
        if i % 5 == 0:
            hs.append(control.play(oa.FixedGain(oa.FramesSignal(oa.Frames.from_slice(rate, clip), 0.0), -4.0)))
            hc.append(cm.play(oc.FixedGain(oc.FramesSignal(oc.Frames(rate, clip), 0.0), -4.0)))
        else:
            hs.append(control.play(oa.FramesSignal(oa.Frames.from_slice(rate, clip), 0.0)))
            hc.append(cm.play(oc.FramesSignal(oc.Frames(rate, clip), 0.0)))
    control.play(oa.Constant(0.25)); cm.play(oc.Constant(0.25))
    interval = np.float32(1.0) / np.float32(48000)
    for cb in range(5):
        if cb == 2:
            hs[7].stop(); hc[7].stop()
        got = mixer.sample_n(interval, n_frames)
        ref = top.sample_n(interval, n_frames)
        assert got.shape == (n_frames,) and ref.shape == (n_frames,)
        if mode == 1:
            np.testing.assert_array_equal(got, ref, err_msg=f"callback {cb}")
        else:
            assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
        assert len(mixer) == len(cm)
    mixer.close()


def test_mono_mixer_with_filters_and_adapt():
    """Gain / Speed chains and Adapt's channel sum over ONE channel (adapt.rs:71) in a Mixer<f32>."""
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=64, max_frames=1024, channels=1)
    mixer.set_mode(oa.MODE_ORDERED)
    mixer.set_adapt(True, initial_rms=0.1, options=oa.AdaptOptions(tau=0.05, max_gain=10.0, low=0.1, high=0.5))
    cm = oc.Mixer(1)
    top = oc.Adapt(cm, 0.1, oc.AdaptOptions(tau=0.05, max_gain=10.0, low=0.1, high=0.5))
    ctl = []
    for i in range(20):
        clip = synth.noise_clip(33, i, 9000)
        gh, sig = oa.Gain.new(oa.FramesSignal(oa.Frames.from_slice(48000, clip), 0.0))
        og = oc.Gain(oc.FramesSignal(oc.Frames(48000, clip), 0.0))
        ctl.append((gh, og))
        control.play(sig); cm.play(og)
    interval = np.float32(1.0) / np.float32(48000)
    for cb in range(4):
        if cb == 1:
            for gh, og in ctl[::3]:
                gh.set_amplitude_ratio(0.4); og.set_amplitude_ratio(0.4)
        np.testing.assert_array_equal(mixer.sample_n(interval, 1024), top.sample_n(interval, 1024), err_msg=f"callback {cb}")
    mixer.close()


def test_mono_mixer_refuses_stereo_signals():
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=4, max_frames=64, channels=1)
    stereo = np.stack([synth.noise_clip(1, 0, 100), synth.noise_clip(1, 1, 100)], axis=1)
    with pytest.raises(TypeError):
        control.play(oa.FramesSignal(oa.Frames.from_slice(48000, stereo), 0.0))
    with pytest.raises(TypeError):
        control.play(oa.MonoToStereo(oa.Sine(0.0, 440.0)))
    mixer.close()


def test_stop_of_a_finished_source_does_not_reach_the_next_owner_of_its_id():
    """Advisor finding: play, finish, stop (a second, idempotent stop), drop the handle, play again -- the new source is
    handed the recycled id and must NOT inherit the stop.  Same for a value sent to a Gain of the old source."""
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=8, max_frames=64)
    cm = oc.Mixer(2)
    interval = np.float32(1.0) / np.float32(48000)
    gh, sig = oa.Gain.new(oa.MonoToStereo(oa.FramesSignal(oa.Frames.from_slice(48000, synth.noise_clip(2, 0, 40)), 0.0)))
    old = control.play(sig)
    oold = cm.play(oc.Gain(oc.MonoToStereo(oc.FramesSignal(oc.Frames(48000, synth.noise_clip(2, 0, 40)), 0.0))))
    for _ in range(3):                       # 40 samples: finished and removed within three 32-frame callbacks
        np.testing.assert_array_equal(mixer.sample_n(interval, 32), cm.sample_n(interval, 32))
    assert old.is_stopped() and oold.is_stopped() and len(mixer) == 0
    old.stop()                               # harmless in the reference (mixer.rs:34-37)
    gh.set_amplitude_ratio(0.0)              # a GainControl that outlived its signal
    old_id = old.id
    del old, gh, sig                         # the handle id is released once the Mixed AND its controls are gone
    import gc
    gc.collect()
    gh2, sig2 = oa.Gain.new(oa.MonoToStereo(oa.FramesSignal(oa.Frames.from_slice(48000, synth.noise_clip(2, 1, 4000)), 0.0)))
    new = control.play(sig2)
    og2 = oc.Gain(oc.MonoToStereo(oc.FramesSignal(oc.Frames(48000, synth.noise_clip(2, 1, 4000)), 0.0)))
    cm.play(og2)
    assert new.id == old_id                  # the id was recycled: the case under test
    for _ in range(4):
        got, ref = mixer.sample_n(interval, 64), cm.sample_n(interval, 64)
        np.testing.assert_array_equal(got, ref)
        assert np.abs(ref).max() > 0
    assert not new.is_stopped() and len(mixer) == 1
    mixer.close()


def test_mixer_control_calls_race_with_sample():
    """Control calls from another thread while the audio thread samples (signal.rs:11-13): plays, stops and gain values
    arrive through the SPSC ring; nothing is lost, nothing deadlocks, every played source ends up stopped."""
    import threading

    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=512, max_frames=256)
    interval = np.float32(1.0) / np.float32(48000)
    handles, errors = [], []
    done = threading.Event()

    def control_thread():
        try:
            rng = np.random.default_rng(3)
            frames = oa.Frames.from_slice(48000, synth.noise_clip(5, 0, 48000))
            for k in range(400):
                gh, sig = oa.Gain.new(oa.MonoToStereo(oa.FramesSignal(frames, float(rng.uniform(0.0, 0.1)))))
                h = control.play(sig)
                handles.append(h)
                gh.set_amplitude_ratio(float(rng.uniform(0.1, 1.0)))
                if k % 3 == 0 and handles:
                    handles[int(rng.integers(0, len(handles)))].stop()
        except Exception as e:          # noqa: BLE001
            errors.append(e)
        finally:
            done.set()
    t = threading.Thread(target=control_thread)
    t.start()
    n_calls = 0
    while not done.is_set() or n_calls < 20:
        out = mixer.sample_n(interval, 256)
        assert np.isfinite(out).all()
        n_calls += 1
        if done.is_set():
            n_calls += 0
    t.join()
    assert not errors, errors
    for h in handles:
        h.stop()
    mixer.sample_n(interval, 256)
    mixer.sample_n(interval, 256)
    assert len(mixer) == 0 and all(h.is_stopped() for h in handles)
    mixer.close()
