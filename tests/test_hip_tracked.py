"""ODDIO_HIP_MODE_TRACKED on small scenes (ODDIO_HIP_PAIR_MIN_GROUPS=1 sends them through spatial_mix_pair, whose TRACK instantiation the
mode's second pass is): the mechanics -- partial sums, their prefixes in the reference's order (the buffered set's sum first), the
restarted running sums, the differences -- on scenes the oracle renders in seconds.  At this size every sum order is within 1e-5 of the
reference; what is checked is that TRACKED is much closer to the reference's sequential sum (src/spatial.rs:204,459-460) than the tree
sum is, with the source kinds, partial callbacks and set changes the pair kernel's own tests use.  The claim the mode exists for --
1e-6 where the tree sum is 2e-5 -- is tests/test_hip_large_scene.py's."""
import numpy as np
import pytest

import scenario  # noqa: F401
from oracle import oracle_c as oc
from test_hip_pair_kernel import INTERVAL, _events, _hip_signal, _oracle_signal, _sources

pytestmark = pytest.mark.gpu


def _render(monkeypatch, mode, sources, n_frames, n_cb, buffered=(), min_groups="1"):
    import oddio_amd as oa
    if min_groups is not None:
        monkeypatch.setenv("ODDIO_HIP_PAIR_MIN_GROUPS", min_groups)
    else:
        monkeypatch.delenv("ODDIO_HIP_PAIR_MIN_GROUPS", raising=False)
    control, scene = oa.SpatialScene(max_sources=1024, max_frames=1024)
    scene.set_mode(mode)
    handles = [control.play(_hip_signal(oa, s), oa.SpatialOptions(s["pos"], s["vel"], s["radius"])) for s in sources]
    for s in buffered:
        control.play_buffered(_hip_signal(oa, s), oa.SpatialOptions(s["pos"], s["vel"], s["radius"]), 40.0, 48000, 0.1)
    outs = []
    for cb in range(n_cb):
        _events(sources, cb, handles, control, 77)
        outs.append(scene.sample_n(INTERVAL, n_frames).copy())
    scene.close()
    return outs


def _render_oracle(sources, n_frames, n_cb, buffered=()):
    scene = oc.SpatialScene()
    handles = [scene.play(_oracle_signal(s), oc.SpatialOptions(s["pos"], s["vel"], s["radius"])) for s in sources]
    for s in buffered:
        scene.play_buffered(_oracle_signal(s), oc.SpatialOptions(s["pos"], s["vel"], s["radius"]), 40.0, 48000, 0.1)
    outs = []
    for cb in range(n_cb):
        _events(sources, cb, handles, scene, 77)
        outs.append(scene.sample_n(INTERVAL, n_frames).copy())
    return outs


@pytest.mark.parametrize("n_src,n_frames,with_buffered", [(900, 1024, False), (333, 700, False), (200, 1024, True), (500, 512, False), (260, 200, False)])
def test_tracked_follows_the_sequential_sum(monkeypatch, n_src, n_frames, with_buffered):
    import oddio_amd as oa
    n_cb = 4
    sources = _sources(900 + n_src, n_src, with_sine=False, short_clip=1800)
    buffered = _sources(17, 12, with_sine=False)[:4] if with_buffered else ()
    buffered = [b for b in buffered if b["kind"] == "frames" and not b.get("reinhard")]
    ref = _render_oracle(sources, n_frames, n_cb, buffered)
    tracked = _render(monkeypatch, oa.MODE_TRACKED, sources, n_frames, n_cb, buffered)
    fast = _render(monkeypatch, oa.MODE_FAST_UNFUSED, sources, n_frames, n_cb, buffered)
    e_t, e_f = [], []
    for cb in range(n_cb):
        scale = np.abs(ref[cb]).max()
        e_t.append(float(np.abs(tracked[cb] - ref[cb]).max() / scale))
        e_f.append(float(np.abs(fast[cb] - ref[cb]).max() / scale))
    print("tracked", e_t, "tree", e_f)
    assert max(e_t) <= 6e-7, (e_t, e_f)               # a few ulps of the peak
    assert np.mean(e_t) < 0.7 * np.mean(e_f), (e_t, e_f)


def test_tracked_falls_back_to_ordered_for_small_scenes(monkeypatch):
    """Scenes below the pair kernel's size (without the test override: 32 768 sources) are ORDERED ones: the reference's bits."""
    import oddio_amd as oa
    sources = _sources(31, 70, with_sine=False)
    ref = _render_oracle(sources, 480, 3)
    got = _render(monkeypatch, oa.MODE_TRACKED, sources, 480, 3, min_groups=None)
    for cb in range(3):
        np.testing.assert_array_equal(got[cb], ref[cb])
