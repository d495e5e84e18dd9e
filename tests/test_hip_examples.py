"""The ports of the reference's examples (examples/offline.rs, wav.rs, adapt.rs) run end to end on
the GPU with --check (each compares its render against the CPU oracle).  GPU only."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script", ["offline.py", "wav_mixer.py", "adapt.py"])
def test_example_runs_and_matches_oracle(script, tmp_path):
    out = tmp_path / "out.wav"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), "--out", str(out), "--check"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert out.stat().st_size > 44
    from oddio_amd import wav
    rate, frames = wav.read_wav(out)
    assert frames.ndim == 2 and frames.shape[1] == 2 and len(frames) > 1000
