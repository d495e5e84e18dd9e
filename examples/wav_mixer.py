#!/usr/bin/env python
"""The reference's examples/wav.rs, offline: a stereo WAV is decoded to float frames
(sample / (2^(bits-1) - 1), examples/wav.rs:30-42), played through a Mixer as
`FramesSignal::from(Frames<[f32;2]>)` and resampled to the output rate by `oddio::run`
(examples/wav.rs:59-68); instead of a cpal stream the blocks are written to a WAV file.

    python examples/wav_mixer.py IN.wav [--out out.wav] [--rate 48000] [--block 1024]

`render(mod, make_mixer, ..)` is backend-agnostic: tests/test_hip_examples.py also runs it on the CPU
oracle and compares.

Without IN.wav a two-tone stereo test clip at 8 kHz is synthesised (the reference embeds
examples/wav/stereo-test.wav, 16-bit stereo 8 kHz, which is not shipped here).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oddio_amd import wav  # noqa: E402


def test_clip(rate=8000, seconds=2.0):
    t = np.arange(int(rate * seconds), dtype=np.float32) / np.float32(rate)
    left = np.sin(t * np.float32(2 * np.pi * 330.0)) * np.float32(0.6)
    right = np.sin(t * np.float32(2 * np.pi * 495.0)) * np.float32(0.4) * np.linspace(0, 1, len(t), dtype=np.float32)
    return rate, wav.to_i16(np.stack([left, right], axis=1)).astype(np.float32) / np.float32(32767)


def render(mod, make_mixer, src_rate, frames, out_rate, block):
    control, mixer = make_mixer()
    clip = mod.Frames.from_slice(src_rate, frames)
    control.play(mod.FramesSignal(clip, 0.0))
    n_blocks = int(np.ceil(len(frames) / src_rate * out_rate / block)) + 1
    out = [mod.run(mixer, out_rate, np.zeros((block, 2), dtype=np.float32)).copy() for _ in range(n_blocks)]
    return np.concatenate(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("wav", nargs="?")
    ap.add_argument("--out", default="wav_mixer.wav")
    ap.add_argument("--rate", type=int, default=48000)
    ap.add_argument("--block", type=int, default=1024)
    args = ap.parse_args()
    src_rate, frames = wav.read_wav(args.wav) if args.wav else test_clip()
    if frames.ndim != 2 or frames.shape[1] != 2:
        raise SystemExit("this example assumes the sound has two channels (examples/wav.rs:27)")
    import oddio_amd as oa

    def hip_mixer():
        control, mixer = oa.Mixer(max_sources=4, max_frames=args.block)
        mixer.set_mode(oa.MODE_ORDERED)
        return control, mixer
    out = render(oa, hip_mixer, src_rate, frames, args.rate, args.block)
    wav.write_wav(args.out, args.rate, out)
    print(f"wrote {args.out}: {len(out)} frames at {args.rate} Hz from {len(frames)} frames at {src_rate} Hz, peak {np.abs(out).max():.4f}")


if __name__ == "__main__":
    main()
