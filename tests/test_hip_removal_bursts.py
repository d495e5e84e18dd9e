"""set.remove() under load (src/spatial.rs:258-261, src/set.rs:183-188): thousands of sources stop inside ONE callback.
The device's compaction has three shapes -- a handful of removals, a sorted list of up to 4096, and a scan of every
slot when more stopped than the list holds -- and all of them must leave the set in the order Vec::swap_remove leaves
it, because ORDERED mode sums in slot order and is compared bit for bit with the CPU oracle.  GPU only."""
import numpy as np
import pytest

from oddio_amd import synth
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

RATE = 48000
INTERVAL = np.float32(1.0) / np.float32(RATE)
N = 1024


def _groups():
    """(count, clip_len, static position or None for scattered): groups A and B sit at one place each, so that each
    whole group crosses `finished_for > distance / 343` in the same callback."""
    return [(5000, 2500, (1.0, 0.0, -2.0)),      # > 4096 stop together: the flag-scan path
            (1800, 6000, (-3.0, 0.5, 1.0)),      # 1800 stop together: the sorted-list path
            (300, 30000, None)]                  # scattered, long-lived: the survivors that get swapped around


@pytest.mark.parametrize("device_mode", [False, True])
def test_thousands_of_sources_stop_in_one_callback(device_mode):
    import oddio_amd as oa
    groups = _groups()
    total = sum(g[0] for g in groups)
    control, scene = oa.SpatialScene(max_sources=total + 64, max_frames=N)
    scene.set_mode(oa.MODE_ORDERED)
    scene.set_exact_updates(True)      # the device-mode variant enqueues 12 callbacks without a wait
    ref = oc.SpatialScene()
    handles = []
    seed = 31
    for gi, (cnt, clip_len, where) in enumerate(groups):
        n_clips = 16
        clips = np.stack([synth.noise_clip(seed + gi, k, clip_len) for k in range(n_clips)])
        clip_of = np.arange(cnt) % n_clips
        if where is None:
            sc = synth.make_scene(seed, cnt, cube=6.0, vmax=3.0)
            pos, vel = sc["position"], sc["velocity"]
        else:
            pos = np.tile(np.asarray(where, dtype=np.float32), (cnt, 1))
            vel = np.zeros((cnt, 3), dtype=np.float32)
        rad = np.full(cnt, 0.1, dtype=np.float32)
        frames = [oa.Frames.from_slice(RATE, clips[k]) for k in range(n_clips)]
        handles += control.play_frames_batch([frames[k] for k in clip_of], np.zeros(cnt), pos, vel, rad)
        ref.play_frames_bulk(RATE, clips, 0.0, pos, vel, rad, clip_of=clip_of)
    if device_mode:
        import torch
        dev_out = torch.zeros((12, N, 2), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
    lens, wants = [], []
    late = None
    for cb in range(12):
        if cb == 5:      # a new source after the first mass removal: it must land at the END of the compacted set
            clip = synth.noise_clip(seed + 9, 0, 20000)
            late = control.play(oa.FramesSignal(oa.Frames.from_slice(RATE, clip), 0.0), oa.SpatialOptions([0.5, 0.2, -1.0], [0.0, 0.0, 0.0], 0.1))
            ref.play(oc.FramesSignal(oc.Frames(RATE, clip), 0.0), oc.SpatialOptions([0.5, 0.2, -1.0], [0.0, 0.0, 0.0], 0.1))
        want = ref.sample_n(INTERVAL, N)
        lens.append(len(ref))
        if device_mode:      # all 12 callbacks are enqueued without a host wait in between; compared afterwards
            scene.sample_device(INTERVAL, dev_out[cb].data_ptr(), N)
            wants.append(want)
        else:
            np.testing.assert_array_equal(scene.sample_n(INTERVAL, N), want, err_msg=f"callback {cb}")
            assert len(scene) == len(ref), f"callback {cb}"
    if device_mode:
        scene.synchronize()
        got = dev_out.cpu().numpy()
        for cb in range(12):
            np.testing.assert_array_equal(got[cb], wants[cb], err_msg=f"callback {cb}")
        assert len(scene) == len(ref)
    # the scenario did what it is for: one callback removed more than the list holds, another between 1000 and 4096
    assert lens[0] == total and lens[-1] <= 301 and any(a - b > 4096 for a, b in zip(lens, lens[1:])), lens
    assert any(1000 < a - b <= 4096 for a, b in zip(lens, lens[1:])), lens
    fin = [h.is_finished() for h in handles]
    assert all(fin[:6800]) and late is not None and not late.is_finished()
    scene.close()
