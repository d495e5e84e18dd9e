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
    """Gain / Speed are not Seek (play_buffered takes them); FixedGain / Reinhard / Tanh nest freely (round 6)."""
    import oddio_amd as oa
    control, scene = oa.SpatialScene(max_sources=8, max_frames=N)
    f = oa.Frames.from_slice(RATE, synth.noise_clip(1, 0, 2000))
    opts = oa.SpatialOptions(np.array([1.0, 0, 0], np.float32), np.zeros(3, np.float32), 0.1)
    with pytest.raises(TypeError):
        control.play(oa.Reinhard(oa.Gain.new(oa.FramesSignal(f, 0.0))[1]), opts)
    with pytest.raises(TypeError):       # five wrappers
        control.play(oa.Reinhard(oa.Tanh(oa.Reinhard(oa.Tanh(oa.FixedGain(oa.FramesSignal(f, 0.0), 1.0))))), opts)
    control.play(oa.Reinhard(oa.Tanh(oa.FramesSignal(f, 0.0))), opts)
    control.play(oa.FixedGain(oa.Tanh(oa.FramesSignal(f, 0.0)), -1.0), opts)
    control.play_buffered(oa.Reinhard(oa.Tanh(oa.Gain.new(oa.FramesSignal(f, 0.0))[1])), opts, 50.0, RATE, 0.1)
    out = scene.sample_n(INTERVAL, N)
    assert np.isfinite(out).all() and np.abs(out).max() > 0
    scene.close()


# General nests of the Seek wrappers (gain.rs:39-51, reinhard.rs:42-50, tanh.rs:36-44 are `impl<T: Seek> Seek`): FX_CHAIN sources
CHAINS_EXACT = (
    [("fixed", -2.0), ("fixed", 3.5)],                                  # FixedGain(FixedGain(x)): two roundings, not one product
    [("reinhard",), ("reinhard",)],
    [("fixed", 6.0), ("reinhard",), ("fixed", -1.5), ("reinhard",)],
    [("reinhard",), ("fixed", 2.0), ("fixed", 2.0)],
    [("fixed", 1.0), ("reinhard",)],                                    # (compact form, for company)
    [],
)
CHAINS_TANH = ([("tanh",), ("reinhard",)], [("fixed", 3.0), ("tanh",), ("tanh",)], [("reinhard",), ("tanh",), ("fixed", -2.0), ("tanh",)])


def _chain_scenes(seed, n_src, leaves, chains, mode, downmix_every=0):
    import oddio_amd as oa
    sc = synth.make_scene(seed, n_src, cube=12.0, vmax=22.0)
    control, scene = oa.SpatialScene(max_sources=n_src + 8, max_frames=N)
    scene.set_mode(mode)
    ref = oc.SpatialScene()
    for i in range(n_src):
        wrap = chains[i % len(chains)]
        vel = np.zeros(3, np.float32) if i % 6 == 5 else sc["velocity"][i]
        sigs = []
        for mod, clip_fn in ((oa, _clip_hip), (oc, _clip_oracle)):
            if downmix_every and i % downmix_every == 0:
                st = np.stack([synth.noise_clip(seed, 2 * i, 7000), synth.noise_clip(seed, 2 * i + 1, 7000)], axis=1)
                fr = mod.Frames.from_slice(44100 if i % 2 else 48000, st)
                sig = mod.Downmix(mod.FramesSignal(fr, 0.03))
                for w in wrap:
                    sig = mod.FixedGain(sig, w[1]) if w[0] == "fixed" else (mod.Reinhard(sig) if w[0] == "reinhard" else mod.Tanh(sig))
            else:
                sig = _build(mod, leaves[i % len(leaves)], i, seed, clip_fn, wrap)
            sigs.append(sig)
        control.play(sigs[0], oa.SpatialOptions(sc["position"][i], vel, 0.1))
        ref.play(sigs[1], oc.SpatialOptions(sc["position"][i], vel, 0.1))
    return control, scene, ref


@pytest.mark.parametrize("n_src", [7, 40, 150])
def test_general_seek_chains_ordered_bit_exact(n_src):
    """Reinhard(Reinhard(x)), FixedGain(FixedGain(x)), four-deep nests, Reinhard(.. Downmix(x)) over every leaf kind: ORDERED mode is
    the oracle bit for bit (single wave and contribution rows), through a listener rotation and a ragged callback."""
    import oddio_amd as oa
    control, scene, ref = _chain_scenes(900 + n_src, n_src, ("frames", "frames", "constant", "cycle", "frames"), CHAINS_EXACT, oa.MODE_ORDERED, downmix_every=4)
    for cb in range(5):
        if cb == 2:
            q = np.array([0.9800666, 0.0, 0.1986693, 0.0], np.float32)
            control.set_listener_rotation(q); ref.set_listener_rotation(q)
        n = N if cb != 3 else 600
        np.testing.assert_array_equal(scene.sample_n(INTERVAL, n), ref.sample_n(INTERVAL, n), err_msg=f"callback {cb}")
    scene.close()


@pytest.mark.parametrize("mode_name", ["FAST", "ORDERED", "TRACKED"])
def test_general_seek_chains_with_tanh_within_tolerance(mode_name):
    """Reinhard(Tanh(x)) and friends: the device's tanhf is a few ulp from glibc's -- the north_star's 1e-5; sine leaves too."""
    import oddio_amd as oa
    control, scene, ref = _chain_scenes(955, 90, ("frames", "sine", "frames", "constant", "cycle"), CHAINS_TANH + CHAINS_EXACT[:3] + (CHAINS_EXACT[4], [("reinhard",)]), getattr(oa, "MODE_" + mode_name),
                                        downmix_every=7)
    for cb in range(4):
        got = scene.sample_n(INTERVAL, N)
        want = ref.sample_n(INTERVAL, N)
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), cb
    scene.close()


def test_general_chain_ids_are_recycled_with_their_chains():
    """A chain lives in a table slot keyed by the handle id: a source that ends, is released and whose id is handed to a new source with
    ANOTHER chain must render the new chain (and the old one until it ends)."""
    import oddio_amd as oa
    control, scene = oa.SpatialScene(max_sources=4, max_frames=N)
    scene.set_mode(oa.MODE_ORDERED)
    ref = oc.SpatialScene()
    pos, vel = np.array([2.0, 0.5, -1.0], np.float32), np.zeros(3, np.float32)
    short = synth.noise_clip(3, 0, 1500)
    long_ = synth.noise_clip(3, 1, 9000)
    h = control.play(oa.FixedGain(oa.FixedGain(oa.FramesSignal(oa.Frames.from_slice(RATE, short), 0.0), 3.0), 2.0), oa.SpatialOptions(pos, vel, 0.1))
    ref.play(oc.FixedGain(oc.FixedGain(oc.FramesSignal(oc.Frames(RATE, short), 0.0), 3.0), 2.0), oc.SpatialOptions(pos, vel, 0.1))
    for cb in range(5):          # the short clip runs out and, a propagation delay later, the source is removed (spatial.rs:244-261)
        np.testing.assert_array_equal(scene.sample_n(INTERVAL, N), ref.sample_n(INTERVAL, N))
    assert len(scene) == len(ref) == 0 and h.is_finished()
    first_id = h.id
    h.release()
    h2 = control.play(oa.Reinhard(oa.Reinhard(oa.FramesSignal(oa.Frames.from_slice(RATE, long_), 0.0))), oa.SpatialOptions(pos, vel, 0.1))
    ref.play(oc.Reinhard(oc.Reinhard(oc.FramesSignal(oc.Frames(RATE, long_), 0.0))), oc.SpatialOptions(pos, vel, 0.1))
    assert h2.id == first_id
    for cb in range(2):
        np.testing.assert_array_equal(scene.sample_n(INTERVAL, N), ref.sample_n(INTERVAL, N))
    scene.close()
