"""Device-resident entry points of the C ABI (sample_device, borrowed clips, batch controls,
per-kernel timing).  GPU only; PyTorch is used purely as an HBM allocator here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INTERVAL = np.float32(1.0) / np.float32(48000)


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    torch.cuda.init()
    return torch


def build(oa, synth, n_src, clips_np, sc, device_ptrs=None, torch=None):
    control, scene = oa.SpatialScene(max_sources=n_src, max_frames=1024)
    if device_ptrs is None:
        frames = [oa.Frames.from_slice(48000, clips_np[i]) for i in range(n_src)]
    else:
        frames = [oa.Frames.from_device_ptr(48000, device_ptrs[i], clips_np.shape[1], copy=False) for i in range(n_src)]
    handles = control.play_frames_batch(frames, np.full(n_src, 0.3), sc["position"], sc["velocity"], sc["radius"])
    return control, scene, handles, frames


def test_sample_device_matches_host_path(torch_cuda):
    torch = torch_cuda
    import oddio_amd as oa
    from oddio_amd import synth
    n_src, L = 96, 20480
    sc = synth.make_scene(5, n_src)
    clips = np.stack([synth.noise_clip(5, i, L) for i in range(n_src)])
    dev_clips = torch.from_numpy(clips).cuda()
    ptrs = [dev_clips.data_ptr() + 4 * L * i for i in range(n_src)]
    ca, sa, ha, fa = build(oa, synth, n_src, clips, sc)
    cb, sb, hb, fb = build(oa, synth, n_src, clips, sc, device_ptrs=ptrs, torch=torch)
    sb.set_profiling(True)
    out = torch.zeros((1024, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
    for step in range(4):
        if step == 2:
            pos = sc["position"] + np.float32(0.25)
            for h, p, v in zip(ha, pos, sc["velocity"]):
                h.set_motion(p, v, False)
            cb.set_motion_batch(hb, pos, sc["velocity"], False)       # bulk form == per-handle form
        if step == 3:
            sa.seek_all(-0.01)
            sb.seek_all(-0.01)
        host = sa.sample_n(INTERVAL, 1024)
        sb.sample_device(INTERVAL, out.data_ptr(), 1024)
        sb.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), host)        # deterministic: same grid, same order
    hist = sb.kernel_ms_history(16)
    assert hist.shape == (4, 3) and (hist > 0).all() and (hist < 50).all()
    assert len(sb) == n_src
    sa.close()
    sb.close()


def test_postfx_device_and_stream(torch_cuda):
    torch = torch_cuda
    import ctypes as C
    import oddio_amd as oa
    from oddio_amd import _lib
    x = torch.linspace(-3, 3, 2048, device="cuda", dtype=torch.float32).reshape(1024, 2).contiguous()
    ref = (x / (1 + x.abs())).cpu().numpy()
    _lib.check(_lib.lib().oddio_hip_postfx_device(0, oa.POSTFX_REINHARD, C.c_void_p(x.data_ptr()), 1024, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(x.cpu().numpy(), ref)
    control, scene = oa.SpatialScene(max_sources=8, max_frames=256)
    scene.set_stream(torch.cuda.current_stream().cuda_stream)
    control.play(oa.Constant(1.0), oa.SpatialOptions(position=[0.0, 0.0, -1.0]))
    out = torch.zeros((256, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
    scene.sample_device(INTERVAL, out.data_ptr(), 256)
    torch.cuda.synchronize()        # the scene's work is on torch's stream now
    o = out.cpu().numpy()
    assert np.allclose(o[:, 0], o[:, 1]) and o[0, 0] > 0
    scene.close()


def test_sample_device_removal_and_insertion_order_is_the_references(torch_cuda):
    """Sources finish (propagation-delay rule, spatial.rs:243-261) and new ones are played while callbacks are
    enqueued back to back with sample_device -- no host synchronisation in between.  The set lives on the device
    (swap_remove and push in stream order), so slot order, hence the ORDERED-mode sum, is the reference's: bit
    exact against the oracle, which removes inside its walk (set.rs:183-188)."""
    torch = torch_cuda
    import oddio_amd as oa
    from oddio_amd import synth
    from oracle import oracle_c as oc
    rate, n_cb, n0 = 48000, 14, 40
    sc = synth.make_scene(77, 200, cube=6.0, vmax=3.0)
    # clips of very different lengths: sources run out at different callbacks (1024 frames each)
    lens = [2600 + 977 * (i % 9) for i in range(200)]
    clips = [synth.noise_clip(77, i, lens[i]) for i in range(200)]
    control, scene = oa.SpatialScene(max_sources=256, max_frames=1024)
    scene.set_mode(oa.MODE_ORDERED)
    scene.set_exact_updates(True)      # 14 callbacks with plays in between and no wait of the test's own: never let an update slip a callback
    oscene = oc.SpatialScene()
    handles = []
    nxt = [0]

    def play(k):
        for _ in range(k):
            i = nxt[0]
            nxt[0] += 1
            handles.append(control.play(oa.FramesSignal(oa.Frames.from_slice(rate, clips[i]), 0.02), oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1)))
            oscene.play(oc.FramesSignal(oc.Frames(rate, clips[i]), 0.02), oc.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
    play(n0)
    outs = torch.zeros((n_cb, 1024, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
    refs = []
    for cb in range(n_cb):
        if cb in (2, 3, 5, 6, 9):
            play(11)                                    # lands behind whatever the walk has compacted by then
        scene.sample_device(INTERVAL, outs[cb].data_ptr(), 1024)     # enqueued, never waited for inside the loop
        ref = np.zeros((1024, 2), dtype=np.float32)
        oc.run(oscene, rate, ref)
        refs.append(ref)
    scene.synchronize()
    got = outs.cpu().numpy()
    removed = n0 + 55 - len(oscene)
    assert removed >= 20, "the scenario is supposed to remove sources while others are inserted"
    for cb in range(n_cb):
        np.testing.assert_array_equal(got[cb], refs[cb], err_msg=f"callback {cb}")
    assert len(scene) == len(oscene)
    fin = [h.is_finished() for h in handles]
    assert sum(fin) == removed
    scene.close()
