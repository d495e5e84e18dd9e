"""BASELINE configs[4] at its REAL size: ONE SpatialScene of 2 097 152 moving sources in 8 contiguous index shards, one
PROCESS per shard, through the library's cross-rank reduce -- against the oracle, which renders the whole scene once on the
host (src/spatial.rs:204,456-463: one reverse walk, one sequential f32 sum).

No 8-GPU node has been available to any round, so the eight ranks share device 0 and the reduce is the deterministic
peer-to-peer one (`oddio_hip_scene_reduce_init_p2p`; ncclCommInitRank refuses ranks that share a device).  What is exercised
at size is everything but the xGMI hop: 262 144 live sources per shard, the shard arithmetic, TRACKED's exchange of the
first pass's totals between the passes (p2p_gather_base), the rank-ordered sum, the post filter after the reduce.

Sources draw on a shared bank of 16 384 white-noise clips of 8 192 samples (512 MiB per process on the device, once on
the host) so that clips and state fit in HBM and in host RAM; positions in a +-10 m cube.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import scenario  # noqa: F401
from oddio_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RATE, N, CLIP, START, SEED = 48000, 1024, 8192, 0.06, 2097
S_TOTAL, WORLD, N_BANK = 2097152, 8, 16384
INTERVAL = np.float32(1.0) / np.float32(RATE)
TRACKED_TOL = 3e-6      # |gpu - the reference's sequential f32 sum| / max|reference| (north_star: 1e-5)
FAST_TOL = 2e-6         # |gpu - f64-accumulated sum of the same contributions| / max|reference|


def clip_of(n_src, n_bank):
    """Source i plays clip clip_of[i] of the bank (the multiplicative hash bench.py's parity check uses)."""
    idx = ((np.arange(n_src, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(20)) % np.uint64(n_bank)
    return idx.astype(np.uint32)


_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
dist.init_process_group("gloo", rank=rank, world_size=world)
import oddio_amd as oa
from oddio_amd import sharding, synth
from test_hip_large_scene import gpu_noise_clips
from test_hip_config5_full_size import clip_of, RATE, N, CLIP, START, SEED, S_TOTAL, N_BANK
interval = np.float32(1.0) / np.float32(RATE)
bank = gpu_noise_clips(SEED, N_BANK, CLIP, torch.device("cuda", 0))
frames = [oa.Frames.from_device_ptr(RATE, bank.data_ptr() + 4 * CLIP * i, CLIP, device=0, copy=False) for i in range(N_BANK)]
sc = synth.make_scene(SEED, S_TOTAL, cube=10.0)
idx = clip_of(S_TOTAL, N_BANK)
outs = {{}}
for name, mode, postfx in (("tracked", oa.MODE_TRACKED, 0), ("fast_reinhard", oa.MODE_FAST, oa.POSTFX_REINHARD)):
    sh = sharding.ShardedSpatialScene(0, S_TOTAL, N, rank, world, None, postfx=postfx, reduce="p2p", dist=dist)
    sh.scene.set_mode(mode)
    lo, hi = sh.shard
    handles = sh.play_frames_batch([frames[int(k)] for k in idx[lo:hi]], np.full(hi - lo, START), sc["position"][lo:hi], sc["velocity"][lo:hi],
                                   sc["radius"][lo:hi])
    got = []
    for cb in range(2):
        if cb == 1:      # every 4th source of the WHOLE scene gets a new Motion (spatial.rs:217-224)
            moved = np.arange(lo + (-lo) % 4, hi, 4)
            ids = np.array([handles[i - lo].id for i in moved], dtype=np.uint32)
            sh.control.set_motion_batch(ids, sc["position"][moved] + np.float32(0.02) * sc["velocity"][moved], (-sc["velocity"][moved]).astype(np.float32), False)
        dist.barrier()       # the reduce's in-kernel waits are bounded (2 s): the ranks enter a callback together
        got.append(sh.sample(interval, np.zeros((N, 2), np.float32)).copy())
    assert len(sh.scene) == hi - lo
    outs[name] = np.stack(got)
    dist.barrier()
    sh.scene.close()
    del sh, handles
np.savez(os.path.join({tmp!r}, f"rank{{rank}}.npz"), **outs)
dist.barrier(); dist.destroy_process_group()
"""


def test_config5_2097152_sources_in_8_shards_against_the_oracle(tmp_path):
    from oracle import oracle_c as oc
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, tmp=str(tmp_path)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(WORLD), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
             for r in range(WORLD)]
    # the oracle renders the whole scene on the host while the ranks run
    bank = np.stack([synth.noise_clip(SEED, i, CLIP) for i in range(N_BANK)])
    sc = synth.make_scene(SEED, S_TOTAL, cube=10.0)
    idx = clip_of(S_TOTAL, N_BANK)
    moved = np.arange(0, S_TOTAL, 4)
    new_pos = (sc["position"][moved] + np.float32(0.02) * sc["velocity"][moved]).astype(np.float32)
    new_vel = (-sc["velocity"][moved]).astype(np.float32)
    ref32, ref64 = [], []
    for acc64 in (False, True):
        scene = oc.SpatialScene()
        scene.play_frames_bulk(RATE, bank, START, sc["position"], sc["velocity"], sc["radius"], clip_of=idx)
        for cb in range(2):
            if cb == 1:
                from oracle.oracle_c import _fp, _vec3, lib
                for k, i in enumerate(moved):
                    lib().oo_scene_set_motion(scene._h, int(i), _fp(_vec3(new_pos[k])), _fp(_vec3(new_vel[k])), 0)
            if acc64:
                ref64.append(scene.sample_f64acc(INTERVAL, N))
            else:
                out = np.zeros((N, 2), dtype=np.float32)
                oc.run(scene, RATE, out)
                ref32.append(out)
        assert len(scene) == S_TOTAL
        del scene
    assert [p.wait(timeout=1500) for p in procs] == [0] * WORLD
    got = [np.load(tmp_path / f"rank{r}.npz") for r in range(WORLD)]
    for r in range(1, WORLD):      # every rank holds the same bits
        np.testing.assert_array_equal(got[0]["tracked"], got[r]["tracked"])
        np.testing.assert_array_equal(got[0]["fast_reinhard"], got[r]["fast_reinhard"])
    report = []
    for cb in range(2):
        scale = float(np.abs(ref32[cb]).max())
        assert scale > 0
        d_tracked = float(np.abs(got[0]["tracked"][cb] - ref32[cb]).max()) / scale
        ref_err = float(np.abs(ref32[cb].astype(np.float64) - ref64[cb]).max()) / scale    # the reference's own distance from the exact sum
        # Reinhard after the reduce (src/reinhard.rs:32), which is 1-Lipschitz: the clipped mix against the clipped exact sum
        clipped = ref64[cb] / (1.0 + np.abs(ref64[cb]))
        d_fast = float(np.abs(got[0]["fast_reinhard"][cb].astype(np.float64) - clipped).max()) / scale
        report.append({"max_abs_reference": scale, "tracked_vs_reference": d_tracked, "fast_reinhard_vs_f64": d_fast, "reference_vs_f64": ref_err})
        assert np.abs(got[0]["fast_reinhard"][cb]).max() < 1.0
    print("configs[4], 2 097 152 sources in 8 shards (p2p reduce, one GPU):", report)
    for rep in report:
        assert rep["tracked_vs_reference"] <= TRACKED_TOL, report
        assert rep["fast_reinhard_vs_f64"] <= FAST_TOL, report
