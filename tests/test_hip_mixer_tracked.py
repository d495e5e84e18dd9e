"""ODDIO_HIP_MODE_TRACKED on a Mixer (mixer_kernels.h TRACK): the fast path's callbacks are mixed twice -- partial tiles, their prefixes
in the reverse walk's order (mixer.rs:100-117), a second pass whose running sums restart there -- which repeats the rounding errors of
the reference's sequential f32 sum.  Checked against ORDERED mode, which is that sum bit for bit
(tests/test_hip_mixer.py), next to the tree sum of FAST mode; through mixer_mix (mixed kinds and rates) and mixer_mix_unit (plain clips
at the output rate).  ODDIO_HIP_TRACK_MIN_SOURCES=1 lets mixers of a few thousand sources take the mode."""
import numpy as np
import pytest

import scenario  # noqa: F401
from oddio_amd import synth

pytestmark = pytest.mark.gpu
INTERVAL = np.float32(1.0) / np.float32(48000)


def _mixer(mode, n_src, unit, seed=33):
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=n_src + 8, max_frames=2048)
    mixer.set_mode(mode)
    rates = (48000,) if unit else (48000, 44100, 96000, 22050)
    clips = [synth.noise_clip(seed, k, 5000 + 997 * k) for k in range(12)]
    frames = {}
    hs = []
    for i in range(n_src):
        k, rate = i % 12, rates[i % len(rates)]
        if (k, rate) not in frames:
            frames[(k, rate)] = oa.Frames.from_slice(rate, clips[k])
        if not unit and i % 9 == 8:
            sig = oa.Constant(0.001 * (i % 50) - 0.02)
        else:
            sig = oa.FramesSignal(frames[(k, rate)], -0.002 * (i % 4) + 0.01 * (i % 3))
        if i % 5 == 0:
            sig = oa.FixedGain(sig, -3.0 - (i % 4))
        hs.append(control.play(oa.MonoToStereo(sig)))
    return mixer, hs


@pytest.mark.parametrize("n_src,n_frames,unit", [(6000, 1024, False), (3500, 700, False), (20000, 1024, True), (5000, 1536, True)])
def test_tracked_mixer_follows_the_sequential_sum(monkeypatch, n_src, n_frames, unit):
    import oddio_amd as oa
    monkeypatch.setenv("ODDIO_HIP_TRACK_MIN_SOURCES", "1")
    mixers = {name: _mixer(mode, n_src, unit) for name, mode in (("tracked", oa.MODE_TRACKED), ("fast", oa.MODE_FAST), ("ordered", oa.MODE_ORDERED))}
    e_t, e_f = [], []
    for cb in range(4):
        if cb == 2:
            for mixer, hs in mixers.values():
                for j in range(5, n_src, 97):
                    hs[j].stop()
        out = {name: mh[0].sample_n(INTERVAL, n_frames).copy() for name, mh in mixers.items()}
        scale = np.abs(out["ordered"]).max()
        e_t.append(float(np.abs(out["tracked"] - out["ordered"]).max() / scale))
        e_f.append(float(np.abs(out["fast"] - out["ordered"]).max() / scale))
        assert len(mixers["tracked"][0]) == len(mixers["ordered"][0])
    print("tracked", e_t, "tree", e_f)
    # (the sources share 12 clips and carry Constants: coherent sums, where the sequential sum's rounding errors -- and with them the tree
    # sum's distance from it -- are several times those of incoherent scenes: 2e-5 .. 9e-5 here, outside the north_star's 1e-5)
    assert max(e_t) <= 5e-6, (e_t, e_f)
    assert np.mean(e_t) < 0.2 * np.mean(e_f), (e_t, e_f)
    for mixer, _ in mixers.values():
        mixer.close()


def test_small_tracked_mixers_are_ordered_ones():
    """Below the mode's size (8 192 sources) a TRACKED mixer is an ORDERED one: the same bits."""
    import oddio_amd as oa
    a, _ = _mixer(oa.MODE_TRACKED, 700, False)
    b, _ = _mixer(oa.MODE_ORDERED, 700, False)
    for cb in range(3):
        np.testing.assert_array_equal(a.sample_n(INTERVAL, 1024), b.sample_n(INTERVAL, 1024))
    a.close(); b.close()
