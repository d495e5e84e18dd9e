"""Downmix<FramesSignal<[f32;2]>> (src/downmix.rs:18-47) played in a SpatialScene on the HIP path vs
the CPU oracle.  GPU only.  Bit-exact in ORDERED mode, including ragged callbacks, where the
reference's Downmix::sample advances the inner clip a whole 256-frame buffer per chunk."""
import numpy as np
import pytest

import scenario
from test_hip_parity import rel_err, run_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_frames", [1024, 300, 1, 700, 1300])
def test_downmix_alone_bit_exact(n_frames):
    spec = scenario.random_spec(500 + n_frames, 3, kinds=("downmix",), clip_len=20000, start=0.05, cube=8.0)
    ref, got, ob, hb = run_pair(spec, n_frames, 5, mode=1)
    assert np.abs(ref).max() > 0
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_downmix_mixed_removal_and_fixed_gain():
    spec = scenario.random_spec(51, 21, kinds=("frames", "downmix", "cycle"), gain_db=(None, -6.0, 2.0, None), clip_len=2500, start=0.0,
                                cube=5.0, cycle_len=200)
    ref, got, ob, hb = run_pair(spec, 1024, 7, mode=1)
    assert len(ob) == len(hb) == 7           # FramesSignal and Downmix sources ran out; the Cycles stay
    np.testing.assert_array_equal(got, ref)
    hb.close()


def test_downmix_fast_mode_and_resample_ratio():
    spec = scenario.random_spec(52, 90, kinds=("downmix", "frames"), clip_len=30000, rate=44100, start=0.3)
    ref, got, ob, hb = run_pair(spec, 1024, 3, mode=0, max_sources=128)
    assert rel_err(got, ref) <= 1e-5
    hb.close()


def test_downmix_rejects_mono_clip():
    import oddio_amd as oa
    clip = oa.Frames.from_slice(48000, np.zeros(16, np.float32))
    with pytest.raises(TypeError):
        oa.Downmix(oa.FramesSignal(clip, 0.0))
