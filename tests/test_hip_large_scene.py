"""Parity at the size the metric is quoted on (BASELINE configs[2]: 262 144 moving FramesSignal
sources in ONE SpatialScene) and for the sharded pattern of configs[3]/[4], all through the C ABI.

The oracle renders the same scene on the host (clips borrowed, set up from C): its sequential f32
sum in the reference's order (src/spatial.rs:204,456-463) and an f64-accumulated sum of the same
contributions.  Own white-noise clips of 8 192 samples per source (8 GiB) keep host RAM bounded;
positions in a +-10 m cube so that the propagation delay (<= 50 ms) stays inside the clip.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import scenario  # noqa: F401  (puts tests/ helpers on the path the same way the other GPU tests do)
from oddio_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RATE = 48000
INTERVAL = np.float32(1.0) / np.float32(RATE)
N = 1024
S_BIG = 262144
# Tolerances of the FAST (tree-sum) mode, stated against what they are measured against:
NORTH_STAR_TOL = 1e-5          # BASELINE.json north_star: relative to max|reference|
FAST_VS_EXACT_TOL = 2e-6       # |gpu - f64-accumulated sum| (measured 2.4e-7 at 262 144 sources)
# The reference's OWN sequential f32 sum is ~sqrt(S) * eps from the exact sum: measured 1.5e-5 / 1.8e-5 at
# 262 144 sources, i.e. already outside NORTH_STAR_TOL of the exact value.  No sum order but the reference's can then be
# within 1e-5 of the reference: FAST mode does NOT meet the north_star tolerance at this size (ORDERED mode, bit-exact,
# does).  What FAST is held to against the reference here is this documented bound, not 1e-5:
FAST_VS_REFERENCE_BOUND_262144 = 4e-5
FAST_VS_REFERENCE_BOUND_65536 = 2e-5
CLIP = 8192
START = 0.06
SEED = 4242


def gpu_noise_clips(seed, n, length, device):
    """synth.noise_clip for clips [0, n) at once, on the GPU (SplitMix64 in wrapping int64 arithmetic)."""
    import torch
    G, M1, M2 = -7046029254386353131, -4658895280553007687, -7723592293110705685   # the u64 constants as i64

    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)
    idx = torch.arange(n, dtype=torch.int64, device=device)
    state0 = torch.tensor((seed ^ 0x5EED), dtype=torch.int64, device=device) ^ (idx * G)
    out = torch.empty((n, length), dtype=torch.float32, device=device)
    steps = torch.arange(1, length + 1, dtype=torch.int64, device=device) * G
    rows = max(1, (1 << 26) // length)
    for r0 in range(0, n, rows):
        z = state0[r0:r0 + rows, None] + steps[None, :]
        z = (z ^ lsr(z, 30)) * M1
        z = (z ^ lsr(z, 27)) * M2
        z = z ^ lsr(z, 31)
        out[r0:r0 + rows] = lsr(z, 40).to(torch.float32) * (2.0 ** -24) * 2.0 - 1.0
    return out


@pytest.fixture(scope="module")
def big():
    """Clips on the GPU and on the host, scene parameters, and the oracle's outputs for 2 callbacks."""
    import torch

    from oracle import oracle_c as oc
    dev = torch.device("cuda", 0)
    clips = gpu_noise_clips(SEED, S_BIG, CLIP, dev)
    host = clips.cpu().numpy()
    for i in (0, 1, 77777, S_BIG - 1):   # the GPU generator is the numpy one
        np.testing.assert_array_equal(host[i], synth.noise_clip(SEED, i, CLIP))
    sc = synth.make_scene(SEED, S_BIG, cube=10.0)
    # every 4th source gets a new Motion before the second callback (exercises spatial.rs:217-224)
    moved = np.arange(0, S_BIG, 4)
    new_pos = sc["position"][moved] + np.float32(0.02) * sc["velocity"][moved]
    new_vel = (-sc["velocity"][moved]).astype(np.float32)
    ref32, ref64 = [], []
    for acc64 in (False, True):
        scene = oc.SpatialScene()
        scene.play_frames_bulk(RATE, host, START, sc["position"], sc["velocity"], sc["radius"])
        handles = None
        for cb in range(2):
            if cb == 1:
                from oracle.oracle_c import lib, _fp, _vec3
                for k, i in enumerate(moved):
                    lib().oo_scene_set_motion(scene._h, int(i), _fp(_vec3(new_pos[k])), _fp(_vec3(new_vel[k])), 0)
            if acc64:
                ref64.append(scene.sample_f64acc(INTERVAL, N))
            else:
                out = np.zeros((N, 2), dtype=np.float32)
                oc.run(scene, RATE, out)
                ref32.append(out)
        del scene, handles
    return {"clips": clips, "host": host, "sc": sc, "moved": moved, "new_pos": new_pos, "new_vel": new_vel,
            "ref32": ref32, "ref64": ref64}


def play_shard(big, lo, hi, mode=0, postfx=0):
    import oddio_amd as oa
    control, scene = oa.SpatialScene(device=0, max_sources=hi - lo, max_frames=N)
    scene.set_mode(mode)
    if postfx:
        scene.set_postfx(postfx)
    base = big["clips"].data_ptr()
    frames = [oa.Frames.from_device_ptr(RATE, base + 4 * CLIP * i, CLIP, device=0, copy=False) for i in range(lo, hi)]
    sc = big["sc"]
    handles = control.play_frames_batch(frames, np.full(hi - lo, START), sc["position"][lo:hi], sc["velocity"][lo:hi], sc["radius"][lo:hi])
    return control, scene, handles, frames


def apply_motion(big, control, handles, lo, hi):
    sel = (big["moved"] >= lo) & (big["moved"] < hi)
    ids = np.array([handles[i - lo].id for i in big["moved"][sel]], dtype=np.uint32)
    if len(ids):
        control.set_motion_batch(ids, big["new_pos"][sel], big["new_vel"][sel], False)


def test_config3_fast_mode_vs_sequential_and_f64_oracle(big):
    """262 144 sources, FAST mode (deterministic tree sum over waves / workgroups)."""
    control, scene, handles, frames = play_shard(big, 0, S_BIG)
    report = []
    for cb in range(2):
        if cb == 1:
            apply_motion(big, control, handles, 0, S_BIG)
        got = scene.sample_n(INTERVAL, N)
        ref, ref64 = big["ref32"][cb], big["ref64"][cb]
        scale = float(np.abs(ref).max())
        d_ref = float(np.abs(got - ref).max())
        err_gpu = float(np.abs(got.astype(np.float64) - ref64).max())
        err_ref = float(np.abs(ref.astype(np.float64) - ref64).max())
        report.append((scale, d_ref / scale, err_gpu / scale, err_ref / scale))
        assert scale > 0
        # (a) the tree sum against the exact (f64-accumulated) sum of the same contributions
        assert err_gpu <= FAST_VS_EXACT_TOL * scale, report
        # (b) ... and never further from it than 4x the reference's own rounding error
        assert err_gpu <= 4 * err_ref + 1e-7 * scale, report
        # (c) against the reference itself: NOT the north_star's 1e-5 at this size (see FAST_VS_REFERENCE_BOUND_262144);
        # the conforming statement is the ORDERED test below
        assert d_ref <= FAST_VS_REFERENCE_BOUND_262144 * scale, report
        fast_conforms = d_ref <= NORTH_STAR_TOL * scale
        report[-1] = report[-1] + (fast_conforms,)
    print("config3 FAST: (max|ref|, |gpu-ref|/s, |gpu-f64|/s, |ref-f64|/s, within 1e-5 of the reference) per callback:", report)
    assert len(scene) == S_BIG


def test_config3_ordered_mode_is_bit_exact(big):
    """The same scene summed in the reference's order by one wavefront: identical bits at full size."""
    control, scene, handles, frames = play_shard(big, 0, S_BIG, mode=1)
    for cb in range(2):
        if cb == 1:
            apply_motion(big, control, handles, 0, S_BIG)
        got = scene.sample_n(INTERVAL, N)
        np.testing.assert_array_equal(got, big["ref32"][cb])


TRACKED_VS_REFERENCE_TOL = 3e-6      # ODDIO_HIP_MODE_TRACKED against the reference's sequential sum (measured ~1e-6): inside the north_star's 1e-5


@pytest.mark.parametrize("n_src,n_frames", [(S_BIG, N), (65536, N), (65536, 512), (S_BIG, 384), (S_BIG, 768)])   # (768: spatial_mix_pair<.., LANE16>)
def test_tracked_mode_is_within_the_north_star_tolerance_of_the_reference(big, n_src, n_frames):
    """ODDIO_HIP_MODE_TRACKED (pair_kernels.h TRACK): two passes of the FAST-mode kernel whose second one restarts every workgroup's
    running sums at the prefix of the first one's partial sums -- the reference's sequential f32 sum, rounding errors included, to
    ~1e-6 of the peak, where the tree sum is 1-2e-5 from it.  Against the oracle's sequential sum at every size (65 536 sources and
    callbacks of up to 512 frames: the tile kernel's TRACK instantiations)."""
    import oddio_amd as oa
    control, scene, handles, frames = play_shard(big, 0, n_src, mode=oa.MODE_TRACKED)
    if n_src == S_BIG and n_frames == N:
        refs = big["ref32"]
    else:
        # the oracle itself (round 6; before: HIP ORDERED, which is the oracle's bits only transitively)
        from oracle import oracle_c as oc
        from oracle.oracle_c import _fp, _vec3, lib
        sc = big["sc"]
        o = oc.SpatialScene()
        o.play_frames_bulk(RATE, big["host"][:n_src], START, sc["position"][:n_src], sc["velocity"][:n_src], sc["radius"][:n_src])
        refs = []
        for cb in range(2):
            if cb == 1:
                for k, i in enumerate(big["moved"]):
                    if i < n_src:
                        lib().oo_scene_set_motion(o._h, int(i), _fp(_vec3(big["new_pos"][k])), _fp(_vec3(big["new_vel"][k])), 0)
            out = np.zeros((n_frames, 2), dtype=np.float32)
            o.sample(INTERVAL, out)
            refs.append(out)
        del o
    report = []
    for cb in range(2):
        if cb == 1:
            apply_motion(big, control, handles, 0, n_src)
        got = scene.sample_n(INTERVAL, n_frames)
        scale = float(np.abs(refs[cb]).max())
        d = float(np.abs(got - refs[cb]).max()) / scale
        report.append(d)
        assert d <= TRACKED_VS_REFERENCE_TOL, report
        assert d <= NORTH_STAR_TOL
    print(f"TRACKED, {n_src} sources, {n_frames} frames: |gpu - reference| / max|reference| per callback:", report)
    assert len(scene) == n_src
    scene.close()


def test_config4_per_gpu_scene_65536_vs_oracle(big):
    """BASELINE configs[3] runs one 65 536-source SpatialScene per GPU: that scene size (a different grid shape from
    configs[2]: fewer source groups per wavefront) against the oracle's sequential f32 and f64-accumulated sums."""
    from oracle import oracle_c as oc
    S = 65536
    sc = big["sc"]
    refs = {}
    for acc64 in (False, True):
        o = oc.SpatialScene()
        o.play_frames_bulk(RATE, big["host"][:S], START, sc["position"][:S], sc["velocity"][:S], sc["radius"][:S])
        outs = []
        for cb in range(2):
            if acc64:
                outs.append(o.sample_f64acc(INTERVAL, N))
            else:
                out = np.zeros((N, 2), dtype=np.float32)
                oc.run(o, RATE, out)
                outs.append(out)
        refs[acc64] = outs
        del o
    control, scene, handles, frames = play_shard(big, 0, S)
    for cb in range(2):
        got = scene.sample_n(INTERVAL, N)
        ref, ref64 = refs[False][cb], refs[True][cb]
        scale = float(np.abs(ref).max())
        err_gpu = float(np.abs(got.astype(np.float64) - ref64).max())
        err_ref = float(np.abs(ref.astype(np.float64) - ref64).max())
        d_ref = float(np.abs(got - ref).max())
        assert scale > 0
        assert err_gpu <= FAST_VS_EXACT_TOL * scale, (cb, err_gpu / scale)
        assert err_gpu <= 4 * err_ref + 1e-7 * scale, (cb, err_gpu / scale, err_ref / scale)
        assert d_ref <= FAST_VS_REFERENCE_BOUND_65536 * scale, (cb, d_ref / scale)      # documented bound, not the north_star's 1e-5 (top of file)
    assert len(scene) == S
    scene.close()
    control2, scene2, handles2, frames2 = play_shard(big, 0, S, mode=1)
    for cb in range(2):
        np.testing.assert_array_equal(scene2.sample_n(INTERVAL, N), refs[False][cb])   # ORDERED: the reference's bits
    scene2.close()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_partials_equal_unsharded_hip(big, world):
    """configs[3]/[4] arithmetic on the HIP path: contiguous index shards rendered as separate scenes (what
    each rank does), partial buffers summed in rank order, Reinhard after the sum -- against the unsharded HIP
    scene and the oracle.  (The transport, RCCL, is covered by the world-1 test below and the 2-GPU test.)"""
    from oddio_amd.sharding import shard_range
    S = 65536                       # 8 x 8192 / 2 x 32768: keeps the test short
    control, scene, handles, frames = play_shard(big, 0, S)
    whole = [scene.sample_n(INTERVAL, N).copy() for _ in range(2)]
    parts = []
    for r in range(world):
        lo, hi = shard_range(S, world, r)
        c, s, h, f = play_shard(big, lo, hi)
        parts.append([s.sample_n(INTERVAL, N).copy() for _ in range(2)])
    for cb in range(2):
        total = np.zeros((N, 2), dtype=np.float32)
        for r in range(world):
            total = total + parts[r][cb]
        scale = np.abs(whole[cb]).max()
        assert scale > 0
        assert np.abs(total - whole[cb]).max() <= 2e-6 * scale      # both are tree sums of the same contributions
        clipped = total / (np.float32(1.0) + np.abs(total))         # Reinhard after the reduce (src/reinhard.rs:32)
        ref = whole[cb] / (np.float32(1.0) + np.abs(whole[cb]))
        assert np.abs(clipped - ref).max() <= 2e-6 * np.abs(ref).max()


def test_rccl_reduce_world_1(big):
    """The C ABI's collective entry points on one GPU: unique id, communicator, an all-reduce per callback on
    the scene's stream, the post filter after it."""
    import oddio_amd as oa
    from oddio_amd import api
    S = 4096
    control, scene, handles, frames = play_shard(big, 0, S, postfx=oa.POSTFX_REINHARD)
    plain = [scene.sample_n(INTERVAL, N).copy() for _ in range(2)]
    control2, scene2, handles2, frames2 = play_shard(big, 0, S, postfx=oa.POSTFX_REINHARD)
    scene2.reduce_init(0, 1, api.reduce_unique_id())
    for cb in range(2):
        got = scene2.sample_n(INTERVAL, N)
        np.testing.assert_array_equal(got, plain[cb])     # a world of one: the sum of one partial, then Reinhard
    scene2.reduce_destroy()


def test_rccl_group_of_one_in_tracked_mode(big, monkeypatch):
    """ODDIO_HIP_MODE_TRACKED in an RCCL reduce group: the exchange between the two passes is an ncclAllGather of the ranks' totals +
    track_base.  With a world of one the base is zero and every block leaves end - start: the same bits as the mode without a group
    (the only execution of that branch this image allows; the p2p branch runs with 2 and 8 ranks below)."""
    import oddio_amd as oa
    from oddio_amd import api
    monkeypatch.setenv("ODDIO_HIP_PAIR_MIN_GROUPS", "1")
    S = 4096
    for n_frames in (N, 448):
        control, scene, handles, frames = play_shard(big, 0, S, mode=oa.MODE_TRACKED)
        alone = [scene.sample_n(INTERVAL, n_frames).copy() for _ in range(2)]
        scene.close()
        control2, scene2, handles2, frames2 = play_shard(big, 0, S, mode=oa.MODE_TRACKED)
        scene2.reduce_init(0, 1, api.reduce_unique_id())
        for cb in range(2):
            np.testing.assert_array_equal(scene2.sample_n(INTERVAL, n_frames), alone[cb])
        scene2.reduce_destroy()
        scene2.close()
        c3, s3, h3, f3 = play_shard(big, 0, S, mode=oa.MODE_ORDERED)
        for cb in range(2):
            ref = s3.sample_n(INTERVAL, n_frames)
            assert np.abs(alone[cb] - ref).max() <= 6e-7 * np.abs(ref).max()
        s3.close()


_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
device = rank if {own_gpu} else 0
torch.cuda.set_device(device)
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
dist.init_process_group("gloo", rank=rank, world_size=world)
import oddio_amd as oa
from oddio_amd import sharding, synth
S, CLIP, N, RATE, START, SEED = 8192, 8192, 1024, 48000, 0.06, 99
uid = sharding.exchange_unique_id(dist) if {reduce!r} == "rccl" else None
sh = sharding.ShardedSpatialScene(device, S, N, rank, world, uid, postfx={postfx}, reduce={reduce!r}, dist=dist)
if {mode}:
    sh.scene.set_mode({mode})
lo, hi = sh.shard
sc = synth.make_scene(SEED, S, cube=10.0)
frames = [oa.Frames.from_slice(RATE, synth.noise_clip(SEED, i, CLIP), device=device) for i in range(lo, hi)]
sh.play_frames_batch(frames, np.full(hi - lo, START), sc["position"][lo:hi], sc["velocity"][lo:hi], sc["radius"][lo:hi])
outs = [sh.sample(np.float32(1.0) / np.float32(RATE), np.zeros((N, 2), np.float32)).copy() for _ in range(2)]
np.save(os.path.join({tmp!r}, f"rank{{rank}}.npy"), np.stack(outs))
dist.barrier(); dist.destroy_process_group()
"""


def _run_sharded_workers(tmp_path, world, own_gpu, reduce, mode=0, postfx=1):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, tmp=str(tmp_path), own_gpu=own_gpu, reduce=reduce, mode=mode, postfx=postfx))
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
             for r in range(world)]
    assert [p.wait(timeout=600) for p in procs] == [0] * world
    return [np.load(tmp_path / f"rank{r}.npy") for r in range(world)]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_scene_p2p_reduce_ranks_share_one_gpu(tmp_path, world):
    """One seeded scene in `world` contiguous index shards, one PROCESS per rank, all on device 0, through the library's
    deterministic peer-to-peer reduce (rank-ordered sum on rank 0) -- against the same scene unsharded on the HIP path,
    and against the rank-ordered sum of separately rendered shards (bit for bit: the reduce is that sum)."""
    import oddio_amd as oa
    from oddio_amd.sharding import shard_range
    S, CLIP2, SEED2 = 8192, 8192, 99
    got = _run_sharded_workers(tmp_path, world, own_gpu=False, reduce="p2p")
    for r in range(1, world):
        np.testing.assert_array_equal(got[0], got[r])        # every rank holds the mix
    sc = synth.make_scene(SEED2, S, cube=10.0)

    def render(lo, hi, postfx):
        control, scene = oa.SpatialScene(device=0, max_sources=hi - lo, max_frames=N)
        if postfx:
            scene.set_postfx(postfx)
        frames = [oa.Frames.from_slice(RATE, synth.noise_clip(SEED2, i, CLIP2), device=0) for i in range(lo, hi)]
        control.play_frames_batch(frames, np.full(hi - lo, START), sc["position"][lo:hi], sc["velocity"][lo:hi], sc["radius"][lo:hi])
        return [scene.sample_n(INTERVAL, N).copy() for _ in range(2)]
    whole = render(0, S, oa.POSTFX_REINHARD)
    parts = [render(*shard_range(S, world, r), 0) for r in range(world)]
    for cb in range(2):
        assert np.abs(got[0][cb] - whole[cb]).max() <= 2e-6 * np.abs(whole[cb]).max()
        total = parts[0][cb].copy()
        for r in range(1, world):
            total = total + parts[r][cb]                      # ((p0 + p1) + p2): the reduce's order
        expect = (total / (np.float32(1.0) + np.abs(total))).astype(np.float32)   # Reinhard after the sum (reinhard.rs:32)
        np.testing.assert_array_equal(got[0][cb], expect)


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_scene_tracked_mode_follows_the_unsharded_reference_sum(tmp_path, monkeypatch, world):
    """ODDIO_HIP_MODE_TRACKED on a sharded scene: between its two passes every rank hands the ranks below it the total of its first
    pass (p2p_gather_base) and starts its running sums at the sum of the ranks above it -- where the reference's reverse walk of the
    WHOLE scene stands when it enters the shard.  Against the unsharded scene in ORDERED mode (the reference's sequential sum, bit
    for bit)."""
    import oddio_amd as oa
    S, CLIP2, SEED2 = 8192, 8192, 99
    monkeypatch.setenv("ODDIO_HIP_PAIR_MIN_GROUPS", "1")        # (shards of 1 024 .. 4 096 sources take the tracked path)
    got = _run_sharded_workers(tmp_path, world, own_gpu=False, reduce="p2p", mode=oa.MODE_TRACKED, postfx=0)
    plain = _run_sharded_workers(tmp_path, world, own_gpu=False, reduce="p2p", mode=oa.MODE_FAST_UNFUSED, postfx=0)     # (the tree sum, for scale)
    for r in range(1, world):
        np.testing.assert_array_equal(got[0], got[r])
    sc = synth.make_scene(SEED2, S, cube=10.0)
    control, scene = oa.SpatialScene(device=0, max_sources=S, max_frames=N)
    scene.set_mode(oa.MODE_ORDERED)
    frames = [oa.Frames.from_slice(RATE, synth.noise_clip(SEED2, i, CLIP2), device=0) for i in range(S)]
    control.play_frames_batch(frames, np.full(S, START), sc["position"], sc["velocity"], sc["radius"])
    errs, tree = [], []
    for cb in range(2):
        ref = scene.sample_n(INTERVAL, N)
        errs.append(float(np.abs(got[0][cb] - ref).max() / np.abs(ref).max()))
        tree.append(float(np.abs(plain[0][cb] - ref).max() / np.abs(ref).max()))
    print(f"sharded TRACKED, world {world}: |gpu - reference| / max|reference| per callback:", errs, "the tree sum:", tree)
    assert max(errs) <= 8e-7, (errs, tree)
    assert np.mean(errs) < 0.5 * np.mean(tree), (errs, tree)
    scene.close()


def test_bench_self_launch_two_ranks_sharded():
    """`python bench.py --gpus 2 --mode sharded --reduce p2p`: the sharded bench path end to end on ONE device."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-devices", "--mode", "sharded", "--reduce", "p2p",
                        "--sources", "4096", "--clip-len", "65536", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["value"] > 0
    assert "p2p" in d["config"]["parallelism"]
    chk = d["config"]["multi_gpu_selfcheck"]              # the sharded-vs-unsharded scene check that precedes the timed region
    assert chk["reduce"] == "p2p" and chk["max_rel_err"] <= 1e-5 and chk["sources"] == 4096
    assert chk["tracked_max_rel_err_vs_ordered"] <= 3e-6          # the sharded scene in TRACKED mode against the unsharded ORDERED one
    assert len(d["config"]["roofline_frac_by_rank"]) == 2 and min(d["config"]["roofline_frac_by_rank"]) > 0
    # the conforming figure of a multi-GPU line: TRACKED on every rank, whole-job aggregate
    assert d["value_conforming_mode"] == "TRACKED" and d["value_conforming"] > 0      # (at 4 096 sources per rank a TRACKED callback is an ORDERED one: no slower than FAST)
    assert d["tracked_mode_ms_per_step"] > 0 and len(d["config"]["tracked_ms_per_step_by_rank"]) == 2


def test_sharded_scene_two_gpus_p2p(tmp_path):
    """The peer-to-peer reduce ACROSS devices (hipIpcOpenMemHandle with peer access, stores over xGMI): two ranks, one GPU
    each, against the rank-ordered sum -- the first thing to look at on a multi-GPU box (conftest.py runs the two-GPU tests first)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs; the 1-GPU box runs the same reduce between processes that share the device")
    got = _run_sharded_workers(tmp_path, 2, own_gpu=True, reduce="p2p")
    np.testing.assert_array_equal(got[0], got[1])
    assert np.isfinite(got[0]).all() and np.abs(got[0]).max() > 0


def test_sharded_scene_two_gpus_rccl(tmp_path):
    """One seeded scene on 2 GPUs through the library's RCCL all-reduce vs the same scene on one GPU."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's multi-GPU box); the 1-GPU box covers the arithmetic and world=1")
    import oddio_amd as oa
    S, CLIP2, SEED2 = 8192, 8192, 99
    script = tmp_path / "worker.py"
    got = _run_sharded_workers(tmp_path, 2, own_gpu=True, reduce="rccl")
    np.testing.assert_array_equal(got[0], got[1])            # all-reduce: every rank holds the mix
    control, scene = oa.SpatialScene(device=0, max_sources=S, max_frames=N)
    scene.set_postfx(oa.POSTFX_REINHARD)
    sc = synth.make_scene(SEED2, S, cube=10.0)
    frames = [oa.Frames.from_slice(RATE, synth.noise_clip(SEED2, i, CLIP2), device=0) for i in range(S)]
    control.play_frames_batch(frames, np.full(S, START), sc["position"], sc["velocity"], sc["radius"])
    for cb in range(2):
        ref = scene.sample_n(INTERVAL, N)
        assert np.abs(got[0][cb] - ref).max() <= 2e-6 * np.abs(ref).max()


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` spawns its own ranks (here both on device 0) and prints one JSON line."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-devices", "--sources", "4096", "--clip-len", "65536",
                        "--steps", "4", "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["value"] > 0
    assert d["roofline"]["frac"] > 0 and d["roofline"]["frac_callback"] > 0
    # the default reduce is RCCL; two ranks on ONE device cannot form a communicator (ncclCommInitRank refuses duplicate devices):
    # the self-check falls back to the peer-to-peer reduce, says why, and still holds the sharded scene to the unsharded one
    chk = d["config"]["multi_gpu_selfcheck"]
    assert chk["reduce"] == "p2p" and chk["rccl_error"] and chk["max_rel_err"] <= 1e-5
    assert chk["tracked_max_rel_err_vs_ordered"] <= 3e-6
    assert len(d["config"]["roofline_frac_by_rank"]) == 2
    assert d["value_conforming_mode"] == "TRACKED" and d["value_conforming"] > 0 and len(d["config"]["tracked_ms_per_step_by_rank"]) == 2


@pytest.mark.parametrize("fault,mode,expect_line", [("hang", "scenes", True), ("crash", "scenes", True), ("hang", "sharded", False)])
def test_bench_survives_a_selfcheck_that_hangs_or_dies(fault, mode, expect_line):
    """RCCL with more than one rank has never run on the boxes this was developed on.  The multi-GPU self-check therefore runs in a
    child process per rank with a timeout: in --mode scenes (no data-path collective) a self-check that hangs or dies costs the run
    its self-check -- the line is printed, the failure is in it; in --mode sharded (the self-check IS the data path) the run ends
    with a message instead of a number."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-devices", "--mode", mode, "--reduce", "p2p", "--sources", "4096",
                        "--clip-len", "65536", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--precondition-ms", "0", "--selfcheck-timeout", "25"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, ODDIO_BENCH_SELFCHECK_FAULT=fault))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if expect_line:
        assert r.returncode == 0, r.stderr[-2000:]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["value"] > 0 and "error" in d["config"]["multi_gpu_selfcheck"]
        assert d["value_conforming"] > 0            # independent scenes track on their own: no collective, timed anyway
    else:
        assert r.returncode != 0 and not lines and "self-check failed" in r.stderr


def test_bench_under_torchrun_two_ranks():
    """The driver's launch line for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` (here two ranks on the one device)."""
    import json
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-devices", "--sources", "4096", "--clip-len", "65536", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    chk = d["config"]["multi_gpu_selfcheck"]
    assert "error" not in chk and chk["max_rel_err"] <= 1e-5 and chk["tracked_max_rel_err_vs_ordered"] <= 3e-6
    assert d["value_conforming"] > 0
