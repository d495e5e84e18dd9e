"""Per-source soft clips: `Reinhard<T>` (src/reinhard.rs:22-50) and `Tanh<T>` (src/tanh.rs:16-44) are Signal + Seek wrappers over
any signal, so `scene.play(Reinhard::new(FramesSignal))`, `play_buffered(Reinhard::new(Gain::new(..)))` and
`mixer.play(MonoToStereo::new(Tanh::new(..)))` are all legal in the reference: every sample of that source is clipped before the
distance gain and the sum.  (The whole-scene `Reinhard::new(scene)` is the reduce kernel's epilogue; tests elsewhere.)

Reinhard is one correctly rounded divide per sample: ORDERED mode reproduces the oracle bit for bit, through every Seek-set
path -- staged windows (plain, padded near-unit layout, constant-fract branch), the exact per-lane path (96 kHz clips), Constant and
Cycle leaves -- and through buffered and Mixer chains (bit-exact golden fixtures: tests/golden/chains_source_clip_*.npz).
Tanh goes through the device's tanhf, a few ulp from glibc's: held to the north_star's 1e-5."""
import numpy as np
import pytest

import scenario  # noqa: F401
from oddio_amd import synth
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

RATE = 48000
INTERVAL = np.float32(1.0) / np.float32(RATE)
N = 1024


def _build(mod, leaf_kind, i, seed, clip_fn, wrap):
    """wrap: sequence of ("fixed", db) / ("reinhard",) / ("tanh",) innermost first"""
    if leaf_kind == "frames":
        sig = mod.FramesSignal(clip_fn(mod, (48000, 44100, 96000)[i % 3], synth.noise_clip(seed, i, 9000)), 0.05 if i % 4 else -0.002)
    elif leaf_kind == "cycle":
        sig = mod.Cycle(clip_fn(mod, 48000, synth.noise_clip(seed, i, 700 + 13 * i)))
    elif leaf_kind == "sine":
        sig = mod.Sine(0.3 * i, 220.0 + 31.0 * i)
    else:
        sig = mod.Constant(0.9 - 0.1 * (i % 7))
    for w in wrap:
        if w[0] == "fixed":
            sig = mod.FixedGain(sig, w[1])
        elif w[0] == "reinhard":
            sig = mod.Reinhard(sig)
        else:
            sig = mod.Tanh(sig)
    return sig


def _clip_oracle(mod, rate, samples):
    return mod.Frames(rate, samples)


def _clip_hip(mod, rate, samples):
    return mod.Frames.from_slice(rate, samples)


WRAPS = ([("reinhard",)], [("fixed", -2.5), ("reinhard",)], [("reinhard",), ("fixed", 4.0)], [], [("fixed", 1.5)])


def _scene_pair(seed, n_src, leaves, clip_kind, mode):
    import oddio_amd as oa
    sc = synth.make_scene(seed, n_src, cube=12.0, vmax=22.0)
    control, scene = oa.SpatialScene(max_sources=n_src + 8, max_frames=N)
    scene.set_mode(mode)
    ref = oc.SpatialScene()
    for i in range(n_src):
        wrap = [(clip_kind,) if w[0] == "reinhard" else w for w in WRAPS[i % len(WRAPS)]]
        leaf = leaves[i % len(leaves)]
        vel = np.zeros(3, np.float32) if i % 6 == 5 else sc["velocity"][i]      # static: resample ratio exactly 1 (frames.rs:180-187)
        control.play(_build(oa, leaf, i, seed, _clip_hip, wrap), oa.SpatialOptions(sc["position"][i], vel, 0.1))
        ref.play(_build(oc, leaf, i, seed, _clip_oracle, wrap), oc.SpatialOptions(sc["position"][i], vel, 0.1))
    return control, scene, ref


@pytest.mark.parametrize("n_src", [10, 40, 150])
def test_reinhard_sources_ordered_bit_exact(n_src):
    import oddio_amd as oa
    control, scene, ref = _scene_pair(700 + n_src, n_src, ("frames", "frames", "constant", "cycle"), "reinhard", oa.MODE_ORDERED)
    for cb in range(5):
        if cb == 2:
            q = np.array([0.9800666, 0.0, 0.1986693, 0.0], np.float32)
            control.set_listener_rotation(q); ref.set_listener_rotation(q)
        got = scene.sample_n(INTERVAL, N if cb != 3 else 600)
        want = ref.sample_n(INTERVAL, N if cb != 3 else 600)
        np.testing.assert_array_equal(got, want, err_msg=f"callback {cb}")
    scene.close()


@pytest.mark.parametrize("clip_kind,mode_name", [("reinhard", "FAST"), ("tanh", "FAST"), ("tanh", "ORDERED")])
def test_clipped_sources_within_tolerance(clip_kind, mode_name):
    import oddio_amd as oa
    control, scene, ref = _scene_pair(811, 60, ("frames", "sine", "frames", "constant"), clip_kind, getattr(oa, "MODE_" + mode_name))
    for cb in range(4):
        got = scene.sample_n(INTERVAL, N)
        want = ref.sample_n(INTERVAL, N)
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), cb
    scene.close()


def test_clip_really_clips():
    """A loud source under Reinhard stays below 1 / (1 + 1/|x|): the wrapper is applied per source, before the sum."""
    import oddio_amd as oa
    control, scene = oa.SpatialScene(max_sources=4, max_frames=N)
    scene.set_mode(oa.MODE_ORDERED)
    loud = (synth.noise_clip(5, 0, 6000) * np.float32(50.0)).astype(np.float32)
    pos = np.array([0.0, 0.0, -0.05], np.float32)          # inside the radius: distance gain 1
    control.play(oa.Reinhard(oa.FramesSignal(oa.Frames.from_slice(RATE, loud), 0.01)), oa.SpatialOptions(pos, np.zeros(3, np.float32), 0.5))
    out = scene.sample_n(INTERVAL, N)
    assert 0.5 < np.abs(out).max() < 1.0
    scene.close()


def test_seek_chain_rules():
    """One FixedGain and one soft clip per Seek chain (anything longer is a buffered chain); Gain / Speed are not Seek."""
    import oddio_amd as oa
    control, scene = oa.SpatialScene(max_sources=4, max_frames=N)
    f = oa.Frames.from_slice(RATE, synth.noise_clip(1, 0, 2000))
    opts = oa.SpatialOptions(np.array([1.0, 0, 0], np.float32), np.zeros(3, np.float32), 0.1)
    with pytest.raises(TypeError):
        control.play(oa.Reinhard(oa.Tanh(oa.FramesSignal(f, 0.0))), opts)
    with pytest.raises(TypeError):
        control.play(oa.Reinhard(oa.Gain.new(oa.FramesSignal(f, 0.0))[1]), opts)
    control.play(oa.FixedGain(oa.Tanh(oa.FramesSignal(f, 0.0)), -1.0), opts)
    control.play_buffered(oa.Reinhard(oa.Tanh(oa.Gain.new(oa.FramesSignal(f, 0.0))[1])), opts, 50.0, RATE, 0.1)
    out = scene.sample_n(INTERVAL, N)
    assert np.isfinite(out).all() and np.abs(out).max() > 0
    scene.close()
