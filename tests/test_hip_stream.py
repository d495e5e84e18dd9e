"""Stream (src/stream.rs over src/spsc.rs) on the HIP path vs the CPU oracle: the SPSC ring is pinned
host memory read by the kernels.  GPU only.  Writes happen between callbacks, so both sides see the
same data at each `update()`; everything is then bit-exact (the stream cursor is a closed form)."""
import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

INTERVAL = np.float32(1.0) / np.float32(48000)


def test_stream_kats_through_mixer():
    # stream.rs `smoke` / `cleanup`, rate 1, mono stream lifted by MonoToStereo: both channels carry it
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=4, max_frames=16)
    c, s = oa.Stream.new(1, 3)
    control.play(oa.MonoToStereo(s))

    def out(n):
        o = mixer.sample_n(np.float32(1.0), n)
        np.testing.assert_array_equal(o[:, 0], o[:, 1])
        return o[:, 0].tolist()
    assert c.write([1.0, 2.0]) == 2
    assert c.write([3.0, 4.0]) == 1
    assert out(5) == [1.0, 2.0, 3.0, 0.0, 0.0]
    assert c.write([5.0, 6.0, 7.0, 8.0]) == 3
    assert out(1) == [5.0]
    assert out(4) == [6.0, 7.0, 0.0, 0.0]
    assert out(2) == [0.0, 0.0]
    assert c.free() == 3
    # cleanup: dropped control -> finished once drained -> removed by the mixer
    assert c.write([1.0, 2.0]) == 2
    c.drop()
    assert len(mixer) == 1
    out(1); out(1)
    assert len(mixer) == 1          # is_finished is checked before sampling (mixer.rs:102)
    out(1)
    assert len(mixer) == 0
    mixer.close()


@pytest.mark.parametrize("channels", [1, 2])
def test_stream_in_mixer_with_filters_matches_oracle(channels):
    import oddio_amd as oa
    rng = np.random.default_rng(40 + channels)
    control, mixer = oa.Mixer(max_sources=8, max_frames=2048)
    mixer.set_mode(oa.MODE_ORDERED)
    cm = oc.Mixer(2)
    c_h, s_h = oa.Stream.new(16000, 5000, channels)
    s_o = oc.Stream(16000, 5000, channels)
    gc_h, g_h = oa.Gain.new(s_h if channels == 2 else oa.MonoToStereo(s_h))
    g_o = oc.Gain(s_o if channels == 2 else oc.MonoToStereo(s_o))
    sc_h, sp_h = oa.Speed.new(oa.FixedGain(g_h, -3.0))
    sp_o = oc.Speed(oc.FixedGain(g_o, -3.0))
    control.play(sp_h)
    cm.play(sp_o)
    for cb in range(30):
        n_new = int(rng.integers(0, 900))                  # sometimes underflows, sometimes overflows
        x = rng.uniform(-1, 1, size=(n_new, 2) if channels == 2 else n_new).astype(np.float32)
        if cb <= 24:
            assert c_h.write(x) == s_o.write(x)
            assert c_h.free() == s_o.free()
        if cb == 8:
            gc_h.set_amplitude_ratio(0.3); g_o.set_amplitude_ratio(0.3)
        if cb == 15:
            sc_h.set_speed(1.4); sp_o.set_speed(1.4)
        if cb == 24:
            c_h.drop(); s_o.close()
        n = int(rng.choice([1024, 512, 2048, 300]))
        a = cm.sample_n(INTERVAL, n)
        b = mixer.sample_n(INTERVAL, n)
        np.testing.assert_array_equal(b, a, err_msg=f"callback {cb}")
        assert len(mixer) == len(cm)
    assert len(mixer) == 0
    mixer.close()


def test_stream_play_buffered_in_scene_matches_oracle():
    import oddio_amd as oa
    rng = np.random.default_rng(77)
    control, scene = oa.SpatialScene(max_sources=16, max_frames=1024)
    scene.set_mode(oa.MODE_ORDERED)
    ref = oc.SpatialScene()
    pairs = []
    for k in range(3):
        c_h, s_h = oa.Stream.new(24000, 4000)
        s_o = oc.Stream(24000, 4000)
        pos, vel = np.float32([3.0 + 2 * k, -1.0, 2.0 - k]), np.float32([-4.0, 1.0 * k, 0.5])
        if k == 1:
            gc, sig_h = oa.Gain.new(s_h)
            sig_o = oc.Gain(s_o)
        else:
            sig_h, sig_o = s_h, s_o
        h = control.play_buffered(sig_h, oa.SpatialOptions(pos, vel, 0.1), 80.0, 48000, 0.1)
        r = ref.play_buffered(sig_o, oc.SpatialOptions(pos, vel, 0.1), 80.0, 48000, 0.1)
        pairs.append([c_h, s_o, h, r])
    # a seekable neighbour so that both sets are live
    clip = rng.uniform(-1, 1, 30000).astype(np.float32)
    control.play(oa.FramesSignal(oa.Frames.from_slice(48000, clip), 0.0), oa.SpatialOptions([1.0, 2.0, 3.0], [0.0, 0.0, 0.0]))
    ref.play(oc.FramesSignal(oc.Frames(48000, clip), 0.0), oc.SpatialOptions([1.0, 2.0, 3.0], [0.0, 0.0, 0.0]))
    for cb in range(40):
        for k, p in enumerate(pairs):
            if p[0] is None:
                continue
            x = rng.uniform(-1, 1, int(rng.integers(200, 800))).astype(np.float32)
            assert p[0].write(x) == p[1].write(x)
            if cb == 10 + 6 * k:
                p[0].drop(); p[1].close()
                p[0] = None
        a = ref.sample_n(INTERVAL, 1024)
        b = scene.sample_n(INTERVAL, 1024)
        np.testing.assert_array_equal(b, a, err_msg=f"callback {cb}")
        assert scene.len_buffered() == ref.len_buffered()
        assert [p[2].is_finished() for p in pairs] == [p[3].is_finished() for p in pairs]
    assert scene.len_buffered() == 0        # drained streams were removed after their propagation delay
    scene.close()


def test_stream_errors():
    import oddio_amd as oa
    from oddio_amd._lib import OddioHipError
    control, mixer = oa.Mixer(max_sources=4, max_frames=16)
    c, s = oa.Stream.new(8000, 10)
    control.play(oa.MonoToStereo(s))
    with pytest.raises(OddioHipError):
        control.play(oa.MonoToStereo(s))               # a Stream is moved into exactly one parent
    sc_control, scene = oa.SpatialScene(max_sources=4, max_frames=16)
    c2, s2 = oa.Stream.new(8000, 10, channels=2)
    with pytest.raises(TypeError):
        sc_control.play_buffered(s2, oa.SpatialOptions(), 10.0, 8000, 0.1)   # stereo stream in a spatial scene
    with pytest.raises(TypeError):
        c.write(np.zeros((4, 2), np.float32))
    c.drop(); c2.drop()
    mixer.close(); scene.close()


def test_stream_concurrent_producer_keeps_order():
    # a producer thread pushes the ramp 1, 2, 3, ... while the audio thread renders; with ds == 1 and
    # t == 0 every rendered sample is a ring item or an underflow zero, so the non-zero outputs must be
    # exactly the ramp, in order, no gaps, no repeats (spsc.rs's acquire/release protocol across PCIe)
    import threading
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=4, max_frames=512)
    c, s = oa.Stream.new(48000, 4096)
    control.play(oa.MonoToStereo(s))
    total = 200_000
    done = threading.Event()

    def producer():
        sent = 0
        while sent < total:
            chunk = np.arange(sent + 1, min(sent + 1 + 1500, total + 1), dtype=np.float32)
            sent += c.write(chunk)
        done.set()
    th = threading.Thread(target=producer)
    th.start()
    got = []
    idle = 0
    while idle < 50:
        o = mixer.sample_n(INTERVAL, 512)[:, 0]
        nz = o[o != 0.0]
        got.append(nz)
        idle = idle + 1 if (done.is_set() and len(nz) == 0) else 0
    th.join()
    seq = np.concatenate(got)
    np.testing.assert_array_equal(seq, np.arange(1, total + 1, dtype=np.float32))
    c.drop()
    mixer.close()
