"""C oracle == numpy restatement, bit for bit, on seeded random scenes.

The two restatements were written independently from the operation-order spec; equality of every
output bit across motion updates, listener rotation, partial chunks and clip edges is the pin we
have in place of running the (Rust, unbuildable here) reference.
"""
import numpy as np
import pytest

from oddio_amd import synth
from oracle import oracle_c as oc
from oracle import oracle_np as on


def build_pair(seed, n_src, kinds, clip_len=22000, rate=48000, start=0.3, gain_db=None, cycle_len=500):
    sc = synth.make_scene(seed, n_src)
    cs, ns = oc.SpatialScene(), on.Scene()
    hc, hn = [], []
    for i in range(n_src):
        kind = kinds[i % len(kinds)]
        pos, vel, rad = sc["position"][i], sc["velocity"][i], sc["radius"][i]
        db = None if gain_db is None else gain_db[i % len(gain_db)]
        if kind == "frames":
            clip = synth.noise_clip(seed, i, clip_len)
            sig = oc.FramesSignal(oc.Frames(rate, clip), start)
            nsrc = on.frames_source(rate, clip, start, fixed_gain_db=db)
        elif kind == "downmix":
            clip = np.stack([synth.noise_clip(seed, i, clip_len), synth.noise_clip(seed + 999, i, clip_len)], axis=1)
            sig = oc.Downmix(oc.FramesSignal(oc.Frames(rate, clip), start))
            nsrc = on.downmix_source(rate, clip, start, fixed_gain_db=db)
        elif kind == "cycle":
            clip = synth.noise_clip(seed, i, cycle_len)
            sig = oc.Cycle(oc.Frames(rate, clip))
            nsrc = on.cycle_source(rate, clip, fixed_gain_db=db)
        elif kind == "sine":
            sig = oc.Sine(sc["phase"][i], sc["freq_hz"][i])
            nsrc = on.sine_source(sc["phase"][i], sc["freq_hz"][i], fixed_gain_db=db)
        else:
            sig = oc.Constant(0.75)
            nsrc = on.constant_source(0.75)
            db = None
        if db is not None:
            sig = oc.FixedGain(sig, db)
        hc.append(cs.play(sig, oc.SpatialOptions(pos, vel, rad)))
        hn.append(ns.play(nsrc, pos, vel, rad))
    return sc, cs, ns, hc, hn


@pytest.mark.parametrize("seed,n_src,n_frames", [(1, 7, 1024), (2, 5, 512), (3, 4, 300), (4, 3, 1), (5, 6, 1300)])
def test_scene_frames_bit_equal(seed, n_src, n_frames):
    sc, cs, ns, hc, hn = build_pair(seed, n_src, ["frames"])
    interval = np.float32(1.0) / np.float32(48000)
    for cb in range(4):
        a = cs.sample_n(interval, n_frames)
        b = ns.sample(interval, n_frames)
        np.testing.assert_array_equal(a, b)
        assert np.abs(a).max() > 0 or cb > 2


@pytest.mark.parametrize("cycle_len,n_frames", [(500, 1024), (3, 300), (1, 64), (40000, 700)])
def test_scene_cycle_bit_equal(cycle_len, n_frames):
    # Cycle in the Seek set (cycle.rs:26-61 under spatial.rs:446-468): two transcriptions must agree
    sc, cs, ns, hc, hn = build_pair(20 + cycle_len, 4, ["cycle", "frames", "cycle"], gain_db=[None, None, -5.0], cycle_len=cycle_len)
    interval = np.float32(1.0) / np.float32(48000)
    for cb in range(3):
        a = cs.sample_n(interval, n_frames)
        b = ns.sample(interval, n_frames)
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("n_frames", [1024, 300, 1, 700])
def test_scene_downmix_bit_equal(n_frames):
    # Downmix<FramesSignal<[f32;2]>> in the Seek set, incl. ragged callbacks (whole-buffer clock advance)
    sc, cs, ns, hc, hn = build_pair(31, 4, ["downmix", "frames", "downmix"], gain_db=[None, None, -5.0], clip_len=9000)
    interval = np.float32(1.0) / np.float32(48000)
    for cb in range(4):
        a = cs.sample_n(interval, n_frames)
        b = ns.sample(interval, n_frames)
        np.testing.assert_array_equal(a, b)


def test_scene_mixed_kinds_motion_rotation():
    sc, cs, ns, hc, hn = build_pair(11, 9, ["frames", "sine", "constant"], gain_db=[None, -6.0, 3.0])
    interval = np.float32(1.0) / np.float32(48000)
    rng = np.random.default_rng(0)
    for cb in range(6):
        if cb in (1, 3):
            for j in (0, 4, 7):
                p = (sc["position"][j] + rng.normal(size=3).astype(np.float32)).astype(np.float32)
                v = sc["velocity"][j]
                hc[j].set_motion(p, v, cb == 3 and j == 4)
                ns.set_motion(hn[j], p, v, cb == 3 and j == 4)
        if cb in (2, 4):
            ang = 0.3 * cb
            q = np.array([np.cos(ang / 2), 0.0, np.sin(ang / 2), 0.0], dtype=np.float32)
            cs.set_listener_rotation(q)
            ns.set_listener_rotation(q)
        a = cs.sample_n(interval, 1024)
        b = ns.sample(interval, 1024)
        np.testing.assert_array_equal(a, b)


def test_scene_clip_edges_and_removal():
    # clips short enough that sources start before 0 (negative cursor), run off the end, finish,
    # and are swap_removed after their propagation delay.
    seed, n_src = 21, 6
    sc = synth.make_scene(seed, n_src, cube=8.0)
    cs, ns = oc.SpatialScene(), on.Scene()
    hc, hn = [], []
    for i in range(n_src):
        clip = synth.noise_clip(seed, i, 700 + 450 * i)
        start = -0.004 * i
        hc.append(cs.play(oc.FramesSignal(oc.Frames(48000, clip), start), oc.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1)))
        hn.append(ns.play(on.frames_source(48000, clip, start), sc["position"][i], sc["velocity"][i], 0.1))
    interval = np.float32(1.0) / np.float32(48000)
    lens = []
    for cb in range(10):
        a = cs.sample_n(interval, 1024)
        b = ns.sample(interval, 1024)
        np.testing.assert_array_equal(a, b)
        assert len(cs) == len(ns.set)
        lens.append(len(cs))
        assert [h.is_finished() for h in hc] == [ns.is_finished(h) for h in hn]
    assert lens[0] == n_src and lens[-1] == 0


def test_resample_ratio_not_one():
    # 44.1 kHz and 22.05 kHz clips into a 48 kHz scene: ds far from 1, many binades of the f32 cursor
    cs, ns = oc.SpatialScene(), on.Scene()
    for i, rate in enumerate((44100, 22050, 96000)):
        clip = synth.noise_clip(5, i, 9000)
        pos, vel = np.array([3.0 + i, 1.0, -2.0], np.float32), np.array([-30.0, 5.0, 12.0], np.float32)
        cs.play(oc.FramesSignal(oc.Frames(rate, clip), 0.05), oc.SpatialOptions(pos, vel, 0.1))
        ns.play(on.frames_source(rate, clip, 0.05), pos, vel, 0.1)
    interval = np.float32(1.0) / np.float32(48000)
    for cb in range(3):
        np.testing.assert_array_equal(cs.sample_n(interval, 1024), ns.sample(interval, 1024))


def test_mixer_config1_bit_equal():
    # BASELINE config 1: 64 MonoToStereo<Sine> sources in one Mixer<[f32;2]>, 1024-frame callbacks
    cm, nm = oc.Mixer(2), on.Mixer(2)
    st = synth.SplitMixStreams(7, 64)
    phase = (st.next_u01() * np.float32(2 * np.pi)).astype(np.float32)
    for k in range(64):
        hz = np.float32(110.0 * 2.0 ** (k / 12.0))
        cm.play(oc.MonoToStereo(oc.Sine(phase[k], hz)))
        nm.play(on.sine_source(phase[k], hz))
    for cb in range(3):
        a = oc.run(cm, 48000, np.zeros((1024, 2), np.float32))
        b = nm.sample(np.float32(1.0) / np.float32(48000), 1024)
        np.testing.assert_array_equal(a, b)


def test_postfx_bit_equal():
    sc, cs, ns, hc, hn = build_pair(31, 4, ["frames"])
    r = oc.Reinhard(cs)
    interval = np.float32(1.0) / np.float32(48000)
    a = r.sample_n(interval, 512)
    b = on.reinhard(ns.sample(interval, 512))
    np.testing.assert_array_equal(a, b)
    sc, cs, ns, hc, hn = build_pair(32, 4, ["frames"])
    t = oc.Tanh(cs)
    np.testing.assert_array_equal(t.sample_n(interval, 512), on.tanh_clip(ns.sample(interval, 512)))
