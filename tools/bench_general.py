#!/usr/bin/env python
"""Timing of the paths beside the headline one: buffered spatial sources, the Mixer's general path, Seek-set Cycle
sources (one wave per source since round 2; every shape since round 3).  Prints ms per 1024-frame callback through the host-output entry point
(stream sync + 8 KiB D2H included) for a few set sizes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ODDIO_HIP_MAX_CYCLE", "4096")      # (Seek-set Cycle sources per scene: 1024 by default)
import oddio_amd as oa  # noqa: E402
from oddio_amd import synth  # noqa: E402

INTERVAL = np.float32(1.0) / np.float32(48000)


def time_calls(sig, n=1024, reps=20):
    for _ in range(3):
        sig.sample_n(INTERVAL, n)
    t0 = time.perf_counter()
    for _ in range(reps):
        sig.sample_n(INTERVAL, n)
    return (time.perf_counter() - t0) / reps * 1e3


def scale_table(n_src):
    """Buffered spatial sources at scale, by leaf: Gain<FramesSignal> / Gain<Sine> / Gain<Constant> / Gain<Cycle> (device-output
    callbacks enqueued back to back; `slow` = sources the callback left to the one-wavefront-per-source general kernel)."""
    import torch
    sc = synth.make_scene(11, n_src)
    clip = oa.Frames.from_slice(48000, synth.noise_clip(1, 0, 480000))
    cyc = oa.Frames.from_slice(48000, synth.noise_clip(3, 0, 5000))
    out = torch.zeros((1024, 2), dtype=torch.float32, device="cuda:0")
    leaves = [("Gain<FramesSignal>", lambda i: oa.FramesSignal(clip, 0.0)), ("Gain<Sine>", lambda i: oa.Sine(0.1 * i, 110.0 + 0.01 * i)),
              ("Gain<Constant>", lambda i: oa.Constant(0.5)), ("Gain<Cycle>", lambda i: oa.Cycle(cyc))]
    base = None
    for name, mk in leaves:
        control, scene = oa.SpatialScene(max_sources=n_src, max_frames=1024)
        scene.reserve_buffered(n_src)
        for i in range(n_src):
            gc, g = oa.Gain.new(mk(i))
            control.play_buffered(g, oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1), 100.0, 48000, 0.1)
        for _ in range(6):
            scene.sample_device(INTERVAL, out.data_ptr(), 1024)
        scene.synchronize()
        t0 = time.perf_counter()
        reps = 12
        for _ in range(reps):
            scene.sample_device(INTERVAL, out.data_ptr(), 1024)
        scene.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        slow = scene.debug_buffered_slow() if hasattr(scene, "debug_buffered_slow") else -1
        base = ms if base is None else base
        print(f"buffered spatial sources at scale ({name}): {n_src:6d} -> {ms:8.3f} ms / 1024-frame callback ({ms / base:5.2f}x Gain<FramesSignal>; {slow} on the general kernel)", flush=True)
        scene.close()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--scale":
        scale_table(int(sys.argv[2]))
        return
    clip = oa.Frames.from_slice(48000, synth.noise_clip(1, 0, 480000))
    for n_src in (1, 64, 1024, 4096):
        control, scene = oa.SpatialScene(max_sources=8192, max_frames=1024)
        sc = synth.make_scene(5, n_src)
        scene.reserve_buffered(n_src)
        for i in range(n_src):
            gc, g = oa.Gain.new(oa.FramesSignal(clip, 0.0))
            control.play_buffered(g, oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1), 100.0, 48000, 0.1)
        print(f"buffered spatial sources (Gain<FramesSignal>): {n_src:5d} -> {time_calls(scene):8.3f} ms / 1024-frame callback")
        scene.close()
    for n_src in (1, 64, 1024):
        control, mixer = oa.Mixer(max_sources=1024, max_frames=1024)
        for i in range(n_src):
            gc, g = oa.Gain.new(oa.MonoToStereo(oa.FramesSignal(clip, 0.0)))
            control.play(g)
        print(f"mixer general path (Gain<MonoToStereo<FramesSignal>>): {n_src:5d} -> {time_calls(mixer):8.3f} ms / 1024-frame callback")
        mixer.close()
    # Seek-set sources beside the 48 kHz FramesSignal (round 4: the cliffs of the Seek set -- resample ratios above 1.11, Sine,
    # Downmix, Cycle -- against the staged-window path at the same source count)
    n_src = 4096
    sc = synth.make_scene(5, n_src)
    clips = {rate: oa.Frames.from_slice(rate, synth.noise_clip(9, rate // 1000, 40 * rate)) for rate in (48000, 96000, 192000)}
    st2 = oa.Frames.from_slice(48000, np.stack([synth.noise_clip(4, 0, 480000), synth.noise_clip(4, 1, 480000)], axis=1))
    cyc4 = oa.Frames.from_slice(48000, synth.noise_clip(2, 0, 5000))
    cyc48 = oa.Frames.from_slice(48000, synth.noise_clip(2, 1, 48000))
    makers = [
        ("FramesSignal, 48 kHz clip (the staged-window path)", lambda i: oa.FramesSignal(clips[48000], 1.0), n_src),
        ("FramesSignal, 96 kHz clip (resample ratio 2)", lambda i: oa.FramesSignal(clips[96000], 1.0), n_src),
        ("FramesSignal, 192 kHz clip (resample ratio 4)", lambda i: oa.FramesSignal(clips[192000], 1.0), n_src),
        ("Sine", lambda i: oa.Sine(0.1 * i, 110.0 + 0.25 * i), n_src),
        ("Downmix<FramesSignal<[f32;2]>>", lambda i: oa.Downmix(oa.FramesSignal(st2, 1.0)), n_src),
        ("Cycle, 5000-sample loop (a tenth of the tiles touches the loop's end: row path)", lambda i: oa.Cycle(cyc4), n_src),
        ("Cycle, 48000-sample loop", lambda i: oa.Cycle(cyc48), n_src),
    ]
    base_ms = None
    for name, mk, count in makers:
        control, scene = oa.SpatialScene(max_sources=8192, max_frames=1024)
        for i in range(count):
            control.play(mk(i), oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
        ms = time_calls(scene)
        if base_ms is None:
            base_ms = ms
        print(f"Seek-set {name}: {count:5d} -> {ms:8.3f} ms / 1024-frame callback ({ms / base_ms * (n_src / count):5.2f}x the 48 kHz FramesSignal time per source)")
        scene.close()
    for n_src in (64, 4096):
        control, scene = oa.SpatialScene(max_sources=8192, max_frames=1024)
        sc = synth.make_scene(5, n_src)
        cyc = oa.Frames.from_slice(48000, synth.noise_clip(2, 0, 5000))
        for i in range(min(n_src, 1024)):
            control.play(oa.Cycle(cyc), oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
        print(f"Seek-set Cycle sources: {min(n_src, 1024):5d} -> {time_calls(scene):8.3f} ms / 1024-frame callback")
        scene.close()

    for n_src in (64, 1024):
        control, scene = oa.SpatialScene(max_sources=8192, max_frames=1024)
        sc = synth.make_scene(6, n_src)
        scene.reserve_buffered(n_src)
        cyc = oa.Frames.from_slice(48000, synth.noise_clip(3, 0, 5000))
        for i in range(n_src):
            gc, g = oa.Gain.new(oa.Cycle(cyc))
            control.play_buffered(g, oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1), 100.0, 48000, 0.1)
        print(f"buffered spatial sources (Gain<Cycle>): {n_src:5d} -> {time_calls(scene):8.3f} ms / 1024-frame callback")
        scene.close()

    # the shapes that were rendered one thread per source until round 3: stereo clips, Sine / Constant / Stream leaves, Fader
    stereo = oa.Frames.from_slice(48000, np.stack([synth.noise_clip(4, 0, 480000), synth.noise_clip(4, 1, 480000)], axis=1))
    for n_src in (1, 64, 1024):
        control, mixer = oa.Mixer(max_sources=1024, max_frames=1024)
        for i in range(n_src):
            gc, g = oa.Gain.new(oa.FramesSignal(stereo, 0.0))
            control.play(g)
        print(f"mixer general path (Gain<FramesSignal<[f32;2]>>): {n_src:5d} -> {time_calls(mixer):8.3f} ms / 1024-frame callback")
        mixer.close()
    for n_src in (1, 64, 1024):
        control, mixer = oa.Mixer(max_sources=1024, max_frames=1024)
        for i in range(n_src):
            gc, g = oa.Gain.new(oa.MonoToStereo(oa.Sine(0.1 * i, 110.0 + i)))
            control.play(g)
        print(f"mixer general path (Gain<MonoToStereo<Sine>>): {n_src:5d} -> {time_calls(mixer):8.3f} ms / 1024-frame callback")
        mixer.close()
    for n_src in (1, 64, 1024):
        control, scene = oa.SpatialScene(max_sources=8192, max_frames=1024)
        sc = synth.make_scene(7, n_src)
        scene.reserve_buffered(n_src)
        for i in range(n_src):
            gc, g = oa.Gain.new(oa.Sine(0.1 * i, 110.0 + i))
            control.play_buffered(g, oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1), 100.0, 48000, 0.1)
        print(f"buffered spatial sources (Gain<Sine>): {n_src:5d} -> {time_calls(scene):8.3f} ms / 1024-frame callback")
        scene.close()
    for n_src in (1, 64):
        control, mixer = oa.Mixer(max_sources=1024, max_frames=1024)
        streams = []
        for i in range(n_src):
            sc_, st_ = oa.Stream.new(48000, 48000 * 2)
            sc_.write(synth.noise_clip(8, i, 48000 * 2 - 8))
            streams.append(sc_)
            control.play(oa.MonoToStereo(st_))
        print(f"mixer general path (MonoToStereo<Stream>, ring in pinned host memory): {n_src:5d} -> {time_calls(mixer, reps=10):8.3f} ms / 1024-frame callback")
        mixer.close()
    for n_src in (1, 64, 256):                                   # (a mixer holds at most 256 Faders)
        control, mixer = oa.Mixer(max_sources=1024, max_frames=1024)
        faders = []
        for i in range(n_src):
            fc, f = oa.Fader.new(oa.MonoToStereo(oa.FramesSignal(clip, 0.0)))
            control.play(f)
            faders.append(fc)
        idle = time_calls(mixer)
        for fc in faders:
            fc.fade_to(oa.MonoToStereo(oa.FramesSignal(clip, 1.0)), 30.0)     # a fade that outlasts the measurement
        print(f"mixer general path (Fader<MonoToStereo<FramesSignal>>): {n_src:5d} -> {idle:8.3f} ms idle, {time_calls(mixer):8.3f} ms while fading / 1024-frame callback")
        mixer.close()


if __name__ == "__main__":
    main()
