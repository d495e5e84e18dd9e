"""oddio_hip_mixer_sample_device: Mixer::sample (src/mixer.rs:92-119) with the frames left in device memory and no wait.  The ids a
callback stopped (finished clips, Mixed::stop) reach the host in pinned snapshots that later calls settle: in ORDERED mode before every
call (the sum order is the set order: bit-exact against the oracle and against the host-output entry), in FAST mode when they have
arrived (stopped sources are skipped on the device at once: the mix is the reference's within tolerance, `len` lags)."""
import numpy as np
import pytest

import scenario  # noqa: F401
from oddio_amd import synth
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

RATE = 48000
INTERVAL = np.float32(1.0) / np.float32(RATE)
N = 1024


def _play_all(n_src, seed):
    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=n_src + 4, max_frames=N)
    ref = oc.Mixer(2)
    handles, rhandles = [], []
    for i in range(n_src):
        n = (1500, 2600, 9000, 9000, 4200)[i % 5] + 7 * i          # some clips end in the first callbacks
        clip = synth.noise_clip(seed, i, n)
        start = 0.001 * (i % 3)
        sig = oa.FramesSignal(oa.Frames.from_slice(RATE, clip), start)
        rsig = oc.FramesSignal(oc.Frames(RATE, clip), start)
        if i % 4 == 1:
            sig, rsig = oa.FixedGain(sig, -3.0), oc.FixedGain(rsig, -3.0)
        handles.append(control.play(oa.MonoToStereo(sig)))
        rhandles.append(ref.play(oc.MonoToStereo(rsig)))
    return control, mixer, ref, handles, rhandles


@pytest.mark.parametrize("n_src", [40, 700])
def test_device_output_ordered_is_bit_exact_with_removals(n_src):
    import torch

    import oddio_amd as oa
    control, mixer, ref, handles, rhandles = _play_all(n_src, 61)
    mixer.set_mode(oa.MODE_ORDERED)
    out = torch.zeros((N, 2), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
    for cb in range(7):
        if cb == 3:
            for j in (2, 3, n_src - 1):
                handles[j].stop(); rhandles[j].stop()
        mixer.sample_device(INTERVAL, out.data_ptr(), N)
        mixer.synchronize()
        want = ref.sample_n(INTERVAL, N)
        np.testing.assert_array_equal(out.cpu().numpy(), want, err_msg=f"callback {cb}")
    got = mixer.sample_n(INTERVAL, N)                      # a host-output call settles every snapshot
    np.testing.assert_array_equal(got, ref.sample_n(INTERVAL, N))
    assert len(mixer) == len(ref)
    assert [h.is_stopped() for h in handles] == [h.is_stopped() for h in rhandles]
    mixer.close()


def test_device_output_fast_enqueued_back_to_back():
    import torch

    import oddio_amd as oa
    n_src = 900
    control, mixer, ref, handles, rhandles = _play_all(n_src, 62)
    outs = [torch.zeros((N, 2), dtype=torch.float32, device="cuda:0") for _ in range(9)]
    for cb in range(9):                                     # nine callbacks in flight: more than the snapshot ring holds
        mixer.sample_device(INTERVAL, outs[cb].data_ptr(), N)
    mixer.synchronize()
    for cb in range(9):
        want = ref.sample_n(INTERVAL, N)
        scale = max(float(np.abs(want).max()), 1e-3)
        assert np.abs(outs[cb].cpu().numpy() - want).max() <= 1e-5 * scale, cb
    mixer.sample_n(INTERVAL, N); ref.sample_n(INTERVAL, N)
    assert len(mixer) == len(ref)
    mixer.close()


def test_mono_mixer_device_output():
    import torch

    import oddio_amd as oa
    control, mixer = oa.Mixer(max_sources=8, max_frames=N, channels=1)
    ref = oc.Mixer(1)
    for i in range(5):
        clip = synth.noise_clip(63, i, 5000)
        control.play(oa.FramesSignal(oa.Frames.from_slice(RATE, clip), 0.0))
        ref.play(oc.FramesSignal(oc.Frames(RATE, clip), 0.0))
    mixer.set_mode(oa.MODE_ORDERED)
    out = torch.zeros((N,), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
    for cb in range(3):
        mixer.sample_device(INTERVAL, out.data_ptr(), N)
        mixer.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), ref.sample_n(INTERVAL, N).reshape(-1))
    mixer.close()
