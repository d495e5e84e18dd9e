"""The two tree-sum modes.  With 17-32 sources a callback is rendered by eight waves -- a small scene's groups of 16 slots are
shared by four waves each (spatial_mix's `split`), every wave adding its four sources in descending slot order from zero --
whose sums are added in a fixed order: the two waves of a workgroup, then the workgroups in ascending order.  What
MODE_FAST_UNFUSED must produce can therefore be written down from single-source oracle renders -- every contribution with
the reference's roundings -- and compared bit for bit.  MODE_FAST fuses the lerp, the gain ramp and the accumulate in such multi-wave callbacks: it has to stay within
the north_star's 1e-5 of the reference (here: far inside it), not equal."""
import numpy as np
import pytest

import scenario  # noqa: F401  (path set-up shared with the other GPU tests)
from oddio_amd import synth
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

RATE = 48000
INTERVAL = np.float32(1.0) / np.float32(RATE)
N = 1024


def _render(mode, clips, sc, n_cb):
    import oddio_amd as oa
    control, scene = oa.SpatialScene(max_sources=64, max_frames=N)
    scene.set_mode(mode)
    for i, c in enumerate(clips):
        control.play(oa.FramesSignal(oa.Frames.from_slice(RATE, c), 0.02), oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
    outs = [scene.sample_n(INTERVAL, N).copy() for _ in range(n_cb)]
    scene.close()
    return outs


@pytest.mark.parametrize("n_src", [24, 32])
def test_unfused_tree_sum_is_the_sum_of_exact_contributions(n_src):
    import oddio_amd as oa
    n_cb = 3
    sc = synth.make_scene(91, n_src, cube=12.0, vmax=15.0)
    clips = [synth.noise_clip(91, i, 9000) for i in range(n_src)]
    # every source alone through the oracle: its contribution with the reference's own roundings (0 + s * gain)
    contrib = []
    for i in range(n_src):
        one = oc.SpatialScene()
        one.play(oc.FramesSignal(oc.Frames(RATE, clips[i]), 0.02), oc.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
        contrib.append([one.sample_n(INTERVAL, N).copy() for _ in range(n_cb)])
    ref = oc.SpatialScene()
    for i in range(n_src):
        ref.play(oc.FramesSignal(oc.Frames(RATE, clips[i]), 0.02), oc.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
    unfused = _render(oa.MODE_FAST_UNFUSED, clips, sc, n_cb)
    fused = _render(oa.MODE_FAST, clips, sc, n_cb)
    for cb in range(n_cb):
        waves = []
        for lo in range(0, 32, 4):                              # wave w: slots [4 w, 4 w + 4), descending slot order, from zero
            acc = np.zeros((N, 2), dtype=np.float32)
            for i in range(min(lo + 4, n_src) - 1, lo - 1, -1):
                acc = acc + contrib[i][cb]
            waves.append(acc)
        wgs = [waves[2 * k] + waves[2 * k + 1] for k in range(4)]   # the two waves of a workgroup, through LDS
        want = ((wgs[0] + wgs[1]) + wgs[2]) + wgs[3]              # reduce_partials: workgroups in ascending order
        np.testing.assert_array_equal(unfused[cb], want, err_msg=f"callback {cb}")
        reference = ref.sample_n(INTERVAL, N)
        scale = np.abs(reference).max()
        assert np.abs(unfused[cb] - reference).max() <= 1e-5 * scale
        assert np.abs(fused[cb] - reference).max() <= 1e-5 * scale       # the north_star's tolerance ...
        assert np.abs(fused[cb] - want).max() <= 4e-7 * scale            # ... and in fact a few ulp from the unfused tree sum
