"""Seek-set sources whose sample window is larger than spatial_mix's LDS stage (608 samples per 512-frame tile): clips of
88.2 / 96 / 176.4 / 192 kHz in a 48 kHz scene (resample ratios 1.8 - 4.0, src/frames.rs:176-201's slow branch), with and
without FixedGain, starting before the clip's first sample and running off its end (frames.rs:105-123).  Rendered in
sub-windows (kernels.h MULTI_STRIDE); ORDERED mode must stay bit-exact on every path: the one-wave walk (<= 1024 sources),
the contribution rows (> 1024), and next to ordinary sources in the same 16-source groups."""
import numpy as np
import pytest

import scenario  # noqa: F401
from oddio_amd import synth
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

INTERVAL = np.float32(1.0) / np.float32(48000)
RATES = (96000, 48000, 192000, 88200, 176400, 44100)


def build(n_src, seed, mode, max_frames=1024):
    import oddio_amd as oa
    control, scene = oa.SpatialScene(max_sources=max(64, n_src), max_frames=max_frames)
    scene.set_mode(mode)
    ref = oc.SpatialScene()
    sc = synth.make_scene(seed, n_src, cube=15.0, vmax=25.0)
    rng = np.random.default_rng(seed)
    for i in range(n_src):
        rate = RATES[i % len(RATES)]
        clip = synth.noise_clip(seed, i, int(rate * 0.09) + 17 * (i % 5))      # ~4 callbacks long: the later ones run off the end
        start = float(rng.uniform(-0.004, 0.01))                                 # some cursors start before the clip
        db = None if i % 3 else float(rng.uniform(-6.0, 3.0))
        sh = oa.FramesSignal(oa.Frames.from_slice(rate, clip), start)
        so = oc.FramesSignal(oc.Frames(rate, clip), start)
        if db is not None:
            sh, so = oa.FixedGain(sh, db), oc.FixedGain(so, db)
        control.play(sh, oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
        ref.play(so, oc.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
    return control, scene, ref


@pytest.mark.parametrize("n_src", [37, 700, 2600])
def test_large_windows_ordered_bit_exact(n_src):
    import oddio_amd as oa
    control, scene, ref = build(n_src, 300 + n_src, oa.MODE_ORDERED)
    for cb, n in enumerate((1024, 1024, 600, 1024, 1024)):
        a, b = ref.sample_n(INTERVAL, n), scene.sample_n(INTERVAL, n)
        np.testing.assert_array_equal(b, a, err_msg=f"{n_src} sources, callback {cb}")
        assert len(scene) == len(ref)
    scene.close()


def test_large_windows_fast_mode_tolerance():
    import oddio_amd as oa
    control, scene, ref = build(3000, 77, oa.MODE_FAST)
    for cb in range(3):
        a, b = ref.sample_n(INTERVAL, 1024), scene.sample_n(INTERVAL, 1024)
        assert np.abs(b - a).max() <= 1e-5 * np.abs(a).max(), f"callback {cb}"      # north_star tolerance
    scene.close()


def test_long_callback_with_large_windows():
    """2048-frame callbacks: the later tiles' records come from tile_records (kernels.h), sub-windows included."""
    import oddio_amd as oa
    control, scene, ref = build(90, 5, oa.MODE_ORDERED, max_frames=2048)
    for cb, n in enumerate((2048, 1500)):
        np.testing.assert_array_equal(scene.sample_n(INTERVAL, n), ref.sample_n(INTERVAL, n), err_msg=f"callback {cb}")
    scene.close()
