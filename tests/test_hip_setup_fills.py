"""Set-up fills against a busy NULL stream (round 6).  hipMemset of device memory and a device-to-device hipMemcpy return before the
fill has run, and the fill runs on the NULL stream, which a scene's own (non-blocking) stream does not wait for: a scene, a buffered
set, a clip or a Mixer made while the NULL stream has work queued must still be whole when its first callback runs
(oddio_hip.hip memset_now / copy_now).  The NULL stream is kept busy with torch matmuls -- torch's default stream is the NULL stream.
Before the fix the first inserts could be wiped by late fills (a source missing from its first callbacks, once a memory fault); found
by the soak under two concurrent processes (profiles/r06_soak_last_tree.txt).  NOTE: this test is a guard, not the reproducer -- it
passes on the build before the fix too (work queued by this process does not delay the runtime's fill enough); what reproduces the
defect is a second process on the GPU: tools/dbg/contend.sh (6 of 40 runs failed before, 0 of 30 after)."""
import numpy as np
import pytest

from oddio_amd import synth
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

INTERVAL = np.float32(1.0) / np.float32(48000)


def _busy(torch, x, n=24):
    y = x
    for _ in range(n):          # ~0.1-0.3 s of work queued on the NULL stream; nothing waits for it here
        y = (x @ y) * 1e-4
    return y


def opts(mod, p, v, r=0.1):
    return mod.SpatialOptions(np.asarray(p, np.float32), np.asarray(v, np.float32), r)


def test_scene_buffered_set_and_clips_made_while_the_null_stream_is_busy():
    import torch
    import oddio_amd as oa
    x = torch.randn(8192, 8192, device="cuda")
    torch.cuda.synchronize()
    keep = _busy(torch, x)
    control, scene = oa.SpatialScene(max_sources=64, max_frames=1024)
    scene.set_mode(oa.MODE_ORDERED)
    ref = oc.SpatialScene()
    clip = synth.noise_clip(41, 0, 20000)
    stereo = np.stack([synth.noise_clip(41, 1, 9000), synth.noise_clip(42, 1, 9000)], axis=1)
    keep = _busy(torch, x)
    f_h = oa.Frames.from_slice(48000, clip)                    # its pad fill and upload queue behind the matmuls
    s_h = oa.Frames.from_slice(48000, stereo)
    control.play(oa.FramesSignal(f_h, 0.0), opts(oa, [3.0, 1.0, -2.0], [4.0, 0.0, 1.0]))
    ref.play(oc.FramesSignal(oc.Frames(48000, clip), 0.0), opts(oc, [3.0, 1.0, -2.0], [4.0, 0.0, 1.0]))
    control.play(oa.Downmix(oa.FramesSignal(s_h, 0.0)), opts(oa, [-2.0, 0.5, 3.0], [0.0, 2.0, 0.0]))
    ref.play(oc.Downmix(oc.FramesSignal(oc.Frames(48000, stereo), 0.0)), opts(oc, [-2.0, 0.5, 3.0], [0.0, 2.0, 0.0]))
    keep = _busy(torch, x)
    gc_h, g_h = oa.Gain.new(oa.FramesSignal(f_h, 0.0))         # the first play_buffered makes the buffered set's tables
    og = oc.Gain(oc.FramesSignal(oc.Frames(48000, clip), 0.0))
    control.play_buffered(g_h, opts(oa, [6.0, 0.0, -4.0], [-8.0, 0.0, 0.0]), 100.0, 48000, 0.1)
    ref.play_buffered(og, opts(oc, [6.0, 0.0, -4.0], [-8.0, 0.0, 0.0]), 100.0, 48000, 0.1)
    keep = _busy(torch, x)
    control.play(oa.Cycle(oa.Frames.from_slice(48000, clip[:700])), opts(oa, [1.0, 0.0, 2.0], [1.0, 1.0, 0.0]))   # the Cycle rows and lists
    ref.play(oc.Cycle(oc.Frames(48000, clip[:700])), opts(oc, [1.0, 0.0, 2.0], [1.0, 1.0, 0.0]))
    control.play(oa.Reinhard(oa.FixedGain(oa.Reinhard(oa.FramesSignal(f_h, 0.0)), 3.0)), opts(oa, [2.0, 0.0, 2.0], [0.0, 0.0, 1.0]))   # the chain table
    ref.play(oc.Reinhard(oc.FixedGain(oc.Reinhard(oc.FramesSignal(oc.Frames(48000, clip), 0.0)), 3.0)), opts(oc, [2.0, 0.0, 2.0], [0.0, 0.0, 1.0]))
    for cb in range(4):
        a, b = ref.sample_n(INTERVAL, 1024), scene.sample_n(INTERVAL, 1024)
        assert np.abs(a).max() > 0
        np.testing.assert_array_equal(b, a, err_msg=f"callback {cb}")
        assert (len(scene), scene.len_buffered()) == (len(ref), ref.len_buffered())
    del keep
    scene.close()


def test_mixer_made_while_the_null_stream_is_busy():
    import torch
    import oddio_amd as oa
    x = torch.randn(8192, 8192, device="cuda")
    torch.cuda.synchronize()
    keep = _busy(torch, x)
    control, mixer = oa.Mixer(max_sources=32, max_frames=1024)
    ref = oc.Mixer(2)
    clip = synth.noise_clip(43, 0, 12000)
    keep = _busy(torch, x)
    f_h = oa.Frames.from_slice(48000, clip)
    control.play(oa.MonoToStereo(oa.FramesSignal(f_h, 0.0)))
    ref.play(oc.MonoToStereo(oc.FramesSignal(oc.Frames(48000, clip), 0.0)))
    control.play(oa.MonoToStereo(oa.Sine(0.25, 330.0)))
    ref.play(oc.MonoToStereo(oc.Sine(0.25, 330.0)))
    for cb in range(3):
        b = oa.run(mixer, 48000, np.zeros((1024, 2), np.float32))
        a = oc.run(ref, 48000, np.zeros((1024, 2), np.float32))
        assert np.abs(b - a).max() <= 1e-5 * np.abs(a).max(), f"callback {cb}"
    del keep
    mixer.close()
