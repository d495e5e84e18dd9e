"""ORDERED mode above the two-kernel threshold, device-output callbacks enqueued back to back: the front of callback k + 1
(control updates, walk, render of the contribution rows, set compaction) runs on the scene's second stream while the
ordered sum of callback k runs on the first (scene_host.inc, `piped`).  Everything that can go wrong there is an
ordering bug -- a row buffer or the length snapshot overwritten early, a control update or a seek applied to the wrong
callback, a mode / profiling switch between pipelined and serial callbacks -- and shows up as a bit difference from the
oracle, which knows nothing of streams."""
import numpy as np
import pytest

import scenario  # noqa: F401  (path set-up shared with the other GPU tests)
from oddio_amd import synth
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

RATE = 48000
INTERVAL = np.float32(1.0) / np.float32(RATE)
N = 1024


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    torch.cuda.init()
    return torch


def _scenes(n_src, seed, lens):
    import oddio_amd as oa
    sc = synth.make_scene(seed, n_src, cube=15.0, vmax=8.0)
    clips = [synth.noise_clip(seed, i, int(lens[i])) for i in range(n_src)]
    control, scene = oa.SpatialScene(max_sources=n_src + 512, max_frames=N)
    scene.set_mode(oa.MODE_ORDERED)
    scene.set_exact_updates(True)
    ref = oc.SpatialScene()
    frames = [oa.Frames.from_slice(RATE, c) for c in clips]
    handles, rhandles = [], []
    for i in range(n_src):
        handles.append(control.play(oa.FramesSignal(frames[i], 0.01), oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1)))
        rhandles.append(ref.play(oc.FramesSignal(oc.Frames(RATE, clips[i]), 0.01), oc.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1)))
    return sc, clips, frames, control, scene, ref, handles, rhandles


def test_pipelined_callbacks_with_removals_plays_and_motion(torch_cuda):
    torch = torch_cuda
    import oddio_amd as oa
    n_src, n_cb = 3000, 12                                       # > 1024: the two-kernel path
    rng = np.random.default_rng(5)
    lens = np.where(np.arange(n_src) % 5 == 2, rng.integers(3 * N, 9 * N, n_src), 14 * N + 999)   # a fifth of the clips end on the way
    sc, clips, frames, control, scene, ref, handles, rhandles = _scenes(n_src, 31, lens)
    outs = torch.zeros((n_cb, N, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
    refs = []
    for cb in range(n_cb):
        if cb in (3, 7):                                         # new sources land behind whatever has been compacted by then
            for i in range(150):
                j = (cb * 150 + i) % n_src
                handles.append(control.play(oa.FramesSignal(frames[j], 0.0), oa.SpatialOptions(sc["position"][i], sc["velocity"][j], 0.1)))
                rhandles.append(ref.play(oc.FramesSignal(oc.Frames(RATE, clips[j]), 0.0), oc.SpatialOptions(sc["position"][i], sc["velocity"][j], 0.1)))
        if cb in (2, 5, 6):                                      # Motion updates between un-synchronised callbacks
            for i in range(0, n_src, 7):
                if not rhandles[i].is_finished():
                    p, v = sc["position"][(i + cb) % n_src], sc["velocity"][(i + 2 * cb) % n_src]
                    handles[i].set_motion(p, v, cb == 5)
                    rhandles[i].set_motion(p, v, cb == 5)
        if cb == 4:
            q = np.float32([0.9238795, 0.0, 0.3826834, 0.0])
            control.set_listener_rotation(q); ref.set_listener_rotation(q)
        scene.sample_device(INTERVAL, outs[cb].data_ptr(), N)    # enqueued, never waited for inside the loop
        want = np.zeros((N, 2), dtype=np.float32)
        oc.run(ref, RATE, want)
        refs.append(want)
    scene.synchronize()
    got = outs.cpu().numpy()
    assert len(ref) < n_src + 300 - 100, "sources are supposed to end on the way"
    for cb in range(n_cb):
        np.testing.assert_array_equal(got[cb], refs[cb], err_msg=f"callback {cb}")
    assert len(scene) == len(ref)
    scene.close()


def test_pipelined_and_serial_callbacks_interleaved(torch_cuda):
    """Pipelined callbacks next to ones that are not: a host-output call, a zero-frame call, profiling events, FAST mode
    for two callbacks, Seek::seek on every source (a kernel on the scene's own stream) -- each switch is a join point."""
    torch = torch_cuda
    import oddio_amd as oa
    n_src = 2048
    lens = np.full(n_src, 40 * N)
    sc, clips, frames, control, scene, ref, handles, rhandles = _scenes(n_src, 32, lens)
    out = torch.zeros((N, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
    plan = ["dev", "dev", "dev", "host", "dev", "dev", "zero", "dev", "prof", "prof", "dev", "dev", "fast", "fast", "dev", "dev", "seek", "dev", "dev", "dev"]
    pending = []                                                 # (device tensor, reference) of un-synchronised callbacks
    for k, what in enumerate(plan):
        if what == "seek":
            scene.seek_all(-0.05)                               # Seek::seek on every live source (frames.rs:211-213) ...
            for kid in ref._kids:                               # ... and on the oracle's sources (none has ended: the scene still owns them)
                kid.seek(-0.05)
            continue
        if what == "zero":
            scene.sample_device(INTERVAL, out.data_ptr(), 0)
            oc.run(ref, RATE, np.zeros((0, 2), dtype=np.float32))
            continue
        scene.set_profiling(1 if what == "prof" else 0)
        scene.set_mode(oa.MODE_FAST if what == "fast" else oa.MODE_ORDERED)
        want = np.zeros((N, 2), dtype=np.float32)
        oc.run(ref, RATE, want)
        if what == "host":
            np.testing.assert_array_equal(scene.sample_n(INTERVAL, N), want, err_msg=f"step {k} ({what})")
        else:
            buf = torch.zeros((N, 2), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
            scene.sample_device(INTERVAL, buf.data_ptr(), N)
            pending.append((k, what, buf, want))
    scene.synchronize()
    for k, what, buf, want in pending:
        got = buf.cpu().numpy()
        if what == "fast":
            assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), f"step {k} (fast)"
        else:
            np.testing.assert_array_equal(got, want, err_msg=f"step {k} ({what})")
    scene.close()
