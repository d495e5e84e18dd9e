#!/usr/bin/env python
"""Seek-set source kinds beside the 48 kHz FramesSignal, at scale: ms per 1024-frame callback on the device
(oddio_hip_scene_sample_device enqueued back to back, one synchronisation at the end), FAST mode.

    python tools/bench_seek_kinds.py [--sources 65536] [--callbacks 24]

Every scene has `--sources` moving sources of ONE kind (positions / velocities of the bench generator):
  frames48 / frames96 / frames192   FramesSignal over clips of that rate (resample ratio 1 / 2 / 4 at 48 kHz output),
                                    4096 distinct clips, scattered
  reinhard / tanh                   Reinhard<FramesSignal> / Tanh<FramesSignal>: a per-source soft clip around a 48 kHz clip source
  sine                              Sine (closed form, sinf per sample)
  downmix                           Downmix<FramesSignal<[f32;2]>> over 48 kHz stereo clips (round 6: FAST mode renders the clips' mono sums,
                                    device_types.h; ODDIO_HIP_DOWNMIX_PRESUM=0: the interleaved stereo windows)
  cycle / cycle48k                  Cycle over 5000-sample loops (0.1 s: a tenth of all tiles touches the loop's end and takes the
                                    row path) / 48000-sample loops (ODDIO_HIP_MAX_CYCLE raised to the source count)
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

RATE = 48000
N = 1024


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sources", type=int, default=65536)
    ap.add_argument("--callbacks", type=int, default=24)
    ap.add_argument("--warm", type=int, default=64, help="untimed callbacks first (the chip leaves its idle clock state under load only)")
    ap.add_argument("--kinds", default="frames48,frames96,frames192,reinhard,tanh,sine,downmix,cycle,cycle48k")
    args = ap.parse_args()
    os.environ.setdefault("ODDIO_HIP_MAX_CYCLE", str(args.sources))
    import torch

    import oddio_amd as oa
    from oddio_amd import synth
    dev = torch.device("cuda", 0)
    S = args.sources
    sc = synth.make_scene(2024, S)
    interval = np.float32(1.0) / np.float32(RATE)
    out = torch.zeros((N, 2), dtype=torch.float32, device=dev)
    n_clips = min(4096, S)
    pick = ((np.arange(S, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(20)) % np.uint64(n_clips)
    base_ms = None
    for kind in args.kinds.split(","):
        control, scene = oa.SpatialScene(max_sources=S, max_frames=N)
        keep = []
        if kind.startswith("frames") or kind in ("downmix", "reinhard", "tanh"):
            rate = int(kind[6:]) * 1000 if kind.startswith("frames") else RATE
            ch = 2 if kind == "downmix" else 1
            length = (int(rate * (1.0 + (args.callbacks + args.warm + 8) * N / RATE * 1.15)) + 4096) & ~3
            clips = (torch.rand((n_clips, length * ch), device=dev, dtype=torch.float32) * 2.0 - 1.0).contiguous()
            keep.append(clips)
            if ch == 1 and kind in ("reinhard", "tanh"):
                # per-source soft clips (reinhard.rs:22-50, tanh.rs:16-44): Reinhard is rendered inline by the staged loops, Tanh by the exact per-lane path
                frames = [oa.Frames.from_device_ptr(rate, clips.data_ptr() + 4 * length * i, length, device=0, copy=False) for i in range(n_clips)]
                wrap = oa.Reinhard if kind == "reinhard" else oa.Tanh
                for i in range(S):
                    control.play(wrap(oa.FramesSignal(frames[int(pick[i])], 1.0)), oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
            elif ch == 1:
                frames = [oa.Frames.from_device_ptr(rate, clips.data_ptr() + 4 * length * i, length, device=0, copy=False) for i in range(n_clips)]
                control.play_frames_batch([frames[int(k)] for k in pick], np.full(S, 1.0), sc["position"], sc["velocity"], sc["radius"])
            else:
                host = clips[:64].cpu().numpy().reshape(64, length, 2)
                frames = [oa.Frames.from_slice(rate, host[i]) for i in range(64)]
                for i in range(S):
                    control.play(oa.Downmix(oa.FramesSignal(frames[int(pick[i]) % 64], 1.0)), oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
        elif kind == "sine":
            for i in range(S):
                control.play(oa.Sine(float(sc["phase"][i]), float(sc["freq_hz"][i])), oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
        elif kind in ("cycle", "cycle48k"):
            frames = [oa.Frames.from_slice(RATE, synth.noise_clip(2, i, 5000 if kind == "cycle" else 48000)) for i in range(64)]
            for i in range(S):
                control.play(oa.Cycle(frames[i % 64]), oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
        else:
            raise SystemExit(f"unknown kind {kind}")
        for _ in range(args.warm):
            scene.sample_device(interval, out.data_ptr(), N)
        scene.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.callbacks):
            scene.sample_device(interval, out.data_ptr(), N)
        scene.synchronize()
        ms = (time.perf_counter() - t0) / args.callbacks * 1e3
        assert len(scene) == S, (kind, len(scene))
        assert bool(torch.isfinite(out).all())
        if base_ms is None:
            base_ms = ms
        print(f"{kind:10s} {S:7d} sources: {ms:8.4f} ms / callback  ({ms / base_ms:5.2f}x frames48)", flush=True)
        scene.close()
        del keep


if __name__ == "__main__":
    main()
