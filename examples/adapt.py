#!/usr/bin/env python
"""The reference's examples/adapt.rs through the HIP path: a Mixer wrapped in
`Adapt::new(mixer, 1e-3/sqrt(2), {tau 0.1, max_gain 1e6, low 0.1/sqrt 2, high 0.5/sqrt 2})`; a very
quiet 500 Hz sine plays for 2 s, then a loud 400 Hz sine joins for 2 s and is stopped, then 2 s more.
The adaptive gain pulls all three stretches into the audible range.

The device Mixer is `Mixer<[f32;2]>`, so the mono sines go through `MonoToStereo` (the reference's
example is mono); Adapt sums the channels (adapt.rs:73), i.e. sees twice the mono level.

    python examples/adapt.py [--out adapt.wav]

`render(mod, make_mixer)` is backend-agnostic: tests/test_hip_examples.py also runs it on the CPU oracle
and compares.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oddio_amd import wav  # noqa: E402

DURATION_SECS, RATE, BLOCK_SIZE = 2, 44100, 512


def render(mod, make_mixer):
    control, mixer = make_mixer()
    r2 = np.sqrt(np.float32(2.0))
    signal = mod.Adapt(mixer, np.float32(1e-3) / r2, mod.AdaptOptions(tau=0.1, max_gain=1e6, low=np.float32(0.1) / r2, high=np.float32(0.5) / r2))
    blocks = []

    def drive():
        for _ in range(RATE * DURATION_SECS // BLOCK_SIZE):
            blocks.append(mod.run(signal, RATE, np.zeros((BLOCK_SIZE, 2), dtype=np.float32)).copy())
    control.play(mod.MonoToStereo(mod.FixedGain(mod.Sine(0.0, 5e2), -60.0)))
    drive()
    handle = control.play(mod.MonoToStereo(mod.FixedGain(mod.Sine(0.0, 4e2), -2.0)))
    drive()
    handle.stop()
    drive()
    return np.concatenate(blocks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="adapt.wav")
    args = ap.parse_args()
    import oddio_amd as oa

    def hip_mixer():
        control, mixer = oa.Mixer(max_sources=4, max_frames=BLOCK_SIZE)
        mixer.set_mode(oa.MODE_ORDERED)
        return control, mixer
    out = render(oa, hip_mixer)
    wav.write_wav(args.out, RATE, out)
    third = len(out) // 3
    rms = [float(np.sqrt(np.mean(out[k * third:(k + 1) * third, 0].astype(np.float64) ** 2))) for k in range(3)]
    print(f"wrote {args.out}: {len(out)} frames; per-stretch RMS {rms[0]:.3f} {rms[1]:.3f} {rms[2]:.3f}")


if __name__ == "__main__":
    main()
