"""Parity of the HIP path (through the C ABI) against the CPU oracle.  GPU only (-m gpu).

Tolerance policy (north_star "1e-5 f32 relative"):
  * per-source contribution and ORDERED mode with FramesSignal/Constant sources: bit-exact
    (assert_array_equal) -- every op is the reference's IEEE op in the reference's order;
  * Sine sources: device sinf vs glibc sinf -> |err| <= 1e-5 * max|ref|;
  * FAST mode (tree sum over waves): max|gpu - ref| <= 1e-5 * max|ref|, and the GPU result must be
    no further from the f64-accumulated sum than the reference's own sequential f32 sum is (x4).
"""
import numpy as np
import pytest

import scenario
from oddio_amd import synth

pytestmark = pytest.mark.gpu

INTERVAL = np.float32(1.0) / np.float32(48000)


def rel_err(got, ref):
    return float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() / max(np.abs(ref).max(), 1e-30))


def run_pair(spec, n_frames, n_callbacks, mode, events=None, postfx=0, max_sources=None, interval=None):
    ob = scenario.play_all(scenario.OracleBackend(), spec)
    hb = scenario.play_all(scenario.HipBackend(max_sources=max_sources or max(8, len(spec["sources"]) + 8),
                                               max_frames=max(n_frames, 1), mode=mode), spec)
    if postfx:
        ob.set_postfx(postfx)
        hb.set_postfx(postfx)
    outs = scenario.run_events([ob, hb], spec, n_frames, n_callbacks, interval=interval, events=events)
    return outs["oracle"], outs["hip"], ob, hb


@pytest.mark.parametrize("seed", range(6))
def test_single_source_contribution_bit_exact(seed):
    # each source alone in a scene == its pre-sum contribution (H2 (i))
    spec = scenario.random_spec(100 + seed, 1, clip_len=24000)
    ref, got, ob, hb = run_pair(spec, 1024, 3, mode=0)
    assert np.abs(ref).max() > 0
    np.testing.assert_array_equal(got, ref)
    hb.close()


@pytest.mark.parametrize("n_src,n_frames", [(37, 1024), (64, 512), (9, 300), (5, 1), (17, 1300), (8, 2048), (130, 1024)])
def test_ordered_mode_bit_exact(n_src, n_frames):
    spec = scenario.random_spec(7 + n_src, n_src, clip_len=30000)
    ref, got, ob, hb = run_pair(spec, n_frames, 3, mode=1)
    assert np.abs(ref).max() > 0
    np.testing.assert_array_equal(got, ref)
    hb.close()


@pytest.mark.parametrize("n_src,n_frames,postfx", [(1025, 1024, 0), (1100, 700, 1), (2049, 1536, 0), (3000, 1, 0), (5003, 1024, 1)])
def test_ordered_mode_large_sets_bit_exact(n_src, n_frames, postfx):
    """Above 1024 sources ORDERED mode renders every source's contribution on the whole chip and adds the rows in the
    reference's order (spatial_mix<.., STORE> + ordered_sum): set sizes that leave partial groups / partial 128-source
    tiles, callback lengths that are not whole 512-frame tiles, every kind of Seek source, a post filter."""
    spec = scenario.random_spec(300 + n_src, n_src, kinds=("frames", "frames", "frames", "constant", "frames", "cycle", "downmix"),
                                clip_len=6000, start=0.05, cube=8.0, gain_db=(None, -4.0, None), cycle_len=333)
    ref, got, ob, hb = run_pair(spec, n_frames, 3, mode=1, postfx=postfx, max_sources=n_src + 7)
    assert np.abs(ref).max() > 0
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_ordered_motion_rotation_constant_fixedgain():
    spec = scenario.random_spec(11, 24, kinds=("frames", "frames", "constant"), gain_db=(None, -6.0, 3.0, None))
    rng = np.random.default_rng(0)
    events = {}
    for cb in (1, 3):
        evs = []
        for j in (0, 4, 7, 13):
            p = (spec["sources"][j]["pos"] + rng.normal(size=3).astype(np.float32)).astype(np.float32)
            evs.append(("motion", j, p, spec["sources"][j]["vel"], cb == 3 and j == 4))
        events[cb] = evs
    for cb in (2, 4):
        ang = 0.3 * cb
        events.setdefault(cb, []).append(("rotation", [np.cos(ang / 2), 0.0, np.sin(ang / 2), 0.0]))
    ref, got, ob, hb = run_pair(spec, 1024, 6, mode=1, events=events)
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_sine_sources_tolerance():
    spec = scenario.random_spec(12, 12, kinds=("sine", "frames", "sine"), gain_db=(None, -3.0))
    ref, got, ob, hb = run_pair(spec, 1024, 4, mode=1)
    assert rel_err(got, ref) <= 1e-5
    hb.close()


def test_clip_edges_and_removal():
    seed, n_src = 21, 6
    sc = synth.make_scene(seed, n_src, cube=8.0)
    sources = []
    for i in range(n_src):
        sources.append({"kind": "frames", "clip": synth.noise_clip(seed, i, 700 + 450 * i), "rate": 48000, "start": -0.004 * i,
                        "pos": sc["position"][i], "vel": sc["velocity"][i], "radius": 0.1, "gain_db": None})
    spec = {"sources": sources}
    ob = scenario.play_all(scenario.OracleBackend(), spec)
    hb = scenario.play_all(scenario.HipBackend(max_sources=16, max_frames=1024, mode=1), spec)
    lens = []
    for cb in range(10):
        a = ob.sample(INTERVAL, 1024)
        b = hb.sample(INTERVAL, 1024)
        np.testing.assert_array_equal(b, a)
        assert len(hb) == len(ob)
        assert [h.is_finished() for h in hb.handles] == [h.is_finished() for h in ob.handles]
        lens.append(len(hb))
    assert lens[0] == n_src and lens[-1] == 0
    hb.close()


def test_spatial_signal_finished_kat():
    # src/spatial.rs:630-665 with FinishedSignal == a one-sample clip (finished at t >= 0, frames.rs:204-206)
    import oddio_amd as oa
    control, scene = oa.SpatialScene(max_sources=8, max_frames=16)
    control.play(oa.FramesSignal(oa.Frames.from_slice(1, [0.0]), 0.0), oa.SpatialOptions(position=[343.0, 0.0, 0.0]))
    scene.sample_n(0.0, 0)
    assert len(scene) == 1, "signal remains after no time has passed"
    scene.sample_n(0.6, 1)
    assert len(scene) == 1, "signal remains partway through propagation"
    scene.sample_n(0.6, 1)
    assert len(scene) == 1, "signal remains immediately after propagation delay expires"
    scene.sample_n(0.0, 0)
    assert len(scene) == 0, "signal dropped on first past after propagation delay expires"
    scene.close()


def test_resample_ratios_and_generic_path():
    # 44.1 / 22.05 / 96 kHz clips in a 48 kHz scene (ds far from 1) + a 1 Hz clip sampled at
    # interval 0.25 s (huge windows -> the direct-from-HBM path)
    sources = []
    for i, rate in enumerate((44100, 22050, 96000, 192000)):
        sources.append({"kind": "frames", "clip": synth.noise_clip(5, i, 40000), "rate": rate, "start": 0.05,
                        "pos": np.array([3.0 + i, 1.0, -2.0], np.float32), "vel": np.array([-30.0, 5.0, 12.0], np.float32),
                        "radius": 0.1, "gain_db": None})
    ref, got, ob, hb = run_pair({"sources": sources}, 1024, 3, mode=1)
    assert np.abs(ref).max() > 0
    np.testing.assert_array_equal(got, ref)
    hb.close()
    # supersonic approach: effective_elapsed < 0 -> the cursor runs backwards (ds < 0)
    src = {"kind": "frames", "clip": synth.noise_clip(6, 0, 60000), "rate": 48000, "start": 0.6,
           "pos": np.array([60.0, 2.0, -1.0], np.float32), "vel": np.array([-500.0, 0.0, 0.0], np.float32), "radius": 0.1, "gain_db": None}
    ref, got, ob, hb = run_pair({"sources": [src]}, 1024, 3, mode=1)
    assert np.abs(ref).max() > 0
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_fast_mode_4096_sources_tolerance():
    # BASELINE config 2 shape: 4096 moving FramesSignal sources, own clip each
    spec = scenario.random_spec(42, 4096, clip_len=20480, noise=False)
    ob = scenario.play_all(scenario.OracleBackend(), spec)
    ob64 = scenario.play_all(scenario.OracleBackend(), spec)
    hb = scenario.play_all(scenario.HipBackend(max_sources=4096, max_frames=1024, mode=0), spec)
    for cb in range(2):
        ref = ob.sample(INTERVAL, 1024)
        ref64 = ob64.sample_f64(INTERVAL, 1024)
        got = hb.sample(INTERVAL, 1024)
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 1e-5 * scale
        err_gpu = np.abs(got.astype(np.float64) - ref64).max()
        err_ref = np.abs(ref.astype(np.float64) - ref64).max()
        assert err_gpu <= 4 * err_ref + 1e-7 * scale, (err_gpu, err_ref)
    hb.close()


@pytest.mark.parametrize("n_src,n_frames", [(1, 1024), (15, 1024), (17, 700), (33, 1024), (63, 512), (65, 1300), (257, 1024), (1000, 1024),
                                            (3001, 1), (5000, 1536)])
def test_fast_mode_ragged_sizes(n_src, n_frames):
    """FAST mode over set sizes that leave partial source groups, partial wavefronts in the walk, one to many
    workgroup partial tiles in the reduce, and callback lengths that are not whole 512-frame tiles."""
    spec = scenario.random_spec(1000 + n_src, n_src, clip_len=8192, start=0.05, cube=8.0, noise=True)   # delay <= 0.04 s: reads stay inside the clip
    ob = scenario.play_all(scenario.OracleBackend(), spec)
    ob64 = scenario.play_all(scenario.OracleBackend(), spec)
    hb = scenario.play_all(scenario.HipBackend(max_sources=n_src, max_frames=n_frames, mode=0), spec)
    for cb in range(2):
        ref = ob.sample(INTERVAL, n_frames)
        ref64 = ob64.sample_f64(INTERVAL, n_frames)
        got = hb.sample(INTERVAL, n_frames)
        scale = max(float(np.abs(ref).max()), 1e-30)
        err_gpu = float(np.abs(got.astype(np.float64) - ref64).max())
        err_ref = float(np.abs(ref.astype(np.float64) - ref64).max())
        assert err_gpu <= 4 * err_ref + 1e-6 * scale, (cb, err_gpu / scale, err_ref / scale)
        assert float(np.abs(got - ref).max()) <= 1e-5 * scale                     # north_star tolerance, as stated (<= 5000 sources)
    hb.close()


def test_fast_mode_deterministic():
    spec = scenario.random_spec(43, 600, clip_len=20480)
    outs = []
    for _ in range(2):
        hb = scenario.play_all(scenario.HipBackend(max_sources=1024, max_frames=1024, mode=0), spec)
        outs.append(np.stack([hb.sample(INTERVAL, 1024).copy() for _ in range(2)]))
        hb.close()
    np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize("postfx", [1, 2])
def test_postfx(postfx):
    spec = scenario.random_spec(31 + postfx, 16, clip_len=24000, cube=3.0)
    ref, got, ob, hb = run_pair(spec, 512, 2, mode=1, postfx=postfx)
    if postfx == 1:
        np.testing.assert_array_equal(got, ref)       # x / (1 + |x|): IEEE ops only
    else:
        assert rel_err(got, ref) <= 1e-5              # device tanhf vs glibc tanhf
    hb.close()


def test_play_during_run_and_empty_scene():
    import oddio_amd as oa
    control, scene = oa.SpatialScene(max_sources=8, max_frames=1024)
    out = scene.sample_n(INTERVAL, 256)
    assert out.shape == (256, 2) and not out.any()      # zeroed output, spatial.rs:389-391
    scene.close()
    spec = scenario.random_spec(77, 6, clip_len=30000)
    late = scenario.random_spec(78, 2, clip_len=30000)["sources"]
    events = {2: [("play", late[0])], 3: [("play", late[1])]}
    ref, got, ob, hb = run_pair(spec, 1024, 5, mode=1, events=events, max_sources=16)
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_long_run_drift_fast_mode():
    # 60 callbacks (1.28 s of audio): cursor bookkeeping (f64 clock + the rounded f32 seeks of
    # spatial.rs:449,465,468) must not drift from the reference; motion updates every 7th callback
    spec = scenario.random_spec(91, 512, clip_len=80000, noise=False)
    ob = scenario.play_all(scenario.OracleBackend(), spec)
    hb = scenario.play_all(scenario.HipBackend(max_sources=512, max_frames=1024, mode=0), spec)
    rng = np.random.default_rng(5)
    for cb in range(60):
        if cb % 7 == 3:
            for j in rng.integers(0, 512, size=40):
                p = (spec["sources"][j]["pos"] + rng.normal(size=3).astype(np.float32) * np.float32(0.3)).astype(np.float32)
                ob.handles[j].set_motion(p, spec["sources"][j]["vel"], False)
                hb.handles[j].set_motion(p, spec["sources"][j]["vel"], False)
        ref = ob.sample(INTERVAL, 1024)
        got = hb.sample(INTERVAL, 1024)
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max(), cb
    hb.close()


def test_capacity_and_error_codes():
    import oddio_amd as oa
    from oddio_amd._lib import OddioHipError
    control, scene = oa.SpatialScene(max_sources=2, max_frames=64)
    fr = oa.Frames.from_slice(48000, np.ones(100, np.float32))
    control.play(oa.FramesSignal(fr, 0.0), oa.SpatialOptions())
    control.play(oa.FramesSignal(fr, 0.0), oa.SpatialOptions())
    with pytest.raises(OddioHipError) as e:
        control.play(oa.FramesSignal(fr, 0.0), oa.SpatialOptions())
    assert e.value.code == -2                      # ODDIO_HIP_ENOMEM: scene is full
    with pytest.raises(OddioHipError) as e:
        scene.sample_n(INTERVAL, 65)
    assert e.value.code == -2                      # n_frames > max_frames
    with pytest.raises(OddioHipError):
        oa.Frames.from_slice(48000, np.zeros(0, np.float32))   # empty clip rejected
    with pytest.raises(TypeError):
        control.play(oa.Gain(oa.FramesSignal(fr, 0.0)), oa.SpatialOptions())   # Gain is not Seek
    out = scene.sample_n(INTERVAL, 64)
    assert np.isfinite(out).all()
    scene.close()


def test_source_at_listener_and_huge_distance():
    # distance < 1e-3 branch of EarState::new (spatial.rs:537-538) and a source 100 km away
    srcs = []
    for pos in ([0.0, 0.0, 0.0], [1e-4, 0.0, 0.0], [1.0e5, 0.0, 0.0], [0.1075, 0.0, 0.0]):
        srcs.append({"kind": "frames", "clip": synth.noise_clip(61, len(srcs), 30000), "rate": 48000, "start": 0.3,
                     "pos": np.array(pos, np.float32), "vel": np.zeros(3, np.float32), "radius": 0.1, "gain_db": None})
    ref, got, ob, hb = run_pair({"sources": srcs}, 1024, 3, mode=1)
    np.testing.assert_array_equal(got, ref)
    hb.close()
