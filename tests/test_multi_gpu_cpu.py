"""The N>1 path on CPU: world_size-2 gloo process groups (no GPU needed).

Covers the sharding arithmetic and the stereo-buffer reduce of oddio_amd/sharding.py.  Each rank
renders its contiguous source shard with the CPU oracle (tests may use the oracle as a stand-in
renderer; on the GPU box the same code path renders shards with the HIP scene), the partial
buffers are summed with torch.distributed, and the result is checked against the single-process
reference within the large-scene tolerance (summation order differs)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



def apply_postfx_numpy(x, postfx):
    """Host-side Reinhard (exact IEEE ops, src/reinhard.rs:32) for the CPU-side check of a reduced buffer."""
    if postfx == 1:
        x = x.astype(np.float32)
        return (x / (np.float32(1.0) + np.abs(x))).astype(np.float32)
    assert postfx == 0
    return x


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, seed, n_src, n_frames, n_cb, postfx, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import scenario
    from oddio_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = scenario.random_spec(seed, n_src, clip_len=9000, cube=10.0, start=0.06)
    lo, hi = sharding.shard_range(n_src, world, rank)
    ob = scenario.play_all(scenario.OracleBackend(), {"sources": spec["sources"][lo:hi]})
    interval = np.float32(1.0) / np.float32(48000)
    outs = []
    for cb in range(n_cb):
        part = torch.from_numpy(ob.sample(interval, n_frames).copy())
        sharding.reduce_stereo(part, dist, dst=None)        # all_reduce: every rank gets the mix
        outs.append(apply_postfx_numpy(part.numpy(), postfx))
    q.put((rank, lo, hi, np.stack(outs)))
    dist.barrier()
    dist.destroy_process_group()


class _OracleScene:
    """Stand-in for the HIP scene behind ShardedSpatialScene on a box without a GPU: same control / sample surface,
    rendered by the oracle."""

    def __init__(self, ob):
        self.ob = ob

    def play_frames_batch(self, frames_list, start_seconds, positions, velocities, radii):
        for f, st, p, v, r in zip(frames_list, start_seconds, positions, velocities, radii):
            self.ob.play({"kind": "frames", "clip": f["clip"], "rate": f["rate"], "start": float(st), "pos": p, "vel": v, "radius": float(r), "gain_db": None})
        return list(range(len(frames_list)))

    def sample(self, interval, out):
        out[...] = self.ob.sample(interval, out.shape[0])
        return out


def _sharded_worker(rank, world, port, seed, n_src, n_frames, n_cb, postfx, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import scenario
    from oddio_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = scenario.random_spec(seed, n_src, clip_len=9000, cube=10.0, start=0.06, kinds=("frames",), gain_db=(None,))

    def factory(max_sources, max_frames):
        sc = _OracleScene(scenario.OracleBackend())
        return sc, sc
    sh = sharding.ShardedSpatialScene(0, n_src, n_frames, rank, world, None, postfx=postfx, reduce="dist", dist=dist, scene_factory=factory)
    lo, hi = sh.shard
    src = spec["sources"][lo:hi]
    sh.play_frames_batch([{"clip": s_["clip"], "rate": s_["rate"]} for s_ in src], [s_["start"] for s_ in src],
                         [s_["pos"] for s_ in src], [s_["vel"] for s_ in src], [s_["radius"] for s_ in src])
    interval = np.float32(1.0) / np.float32(48000)
    outs = [sh.sample(interval, np.zeros((n_frames, 2), np.float32)).copy() for _ in range(n_cb)]
    q.put((rank, lo, hi, np.stack(outs)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_src,postfx", [(2, 11, 1), (2, 1, 0)])
def test_sharded_spatial_scene_control_flow_gloo(world, n_src, postfx):
    """ShardedSpatialScene itself (shard bookkeeping, play_frames_batch of the rank's shard, sample + cross-rank sum +
    post filter AFTER the sum), world 2 over gloo; the renderer behind it is the oracle (no GPU here)."""
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenario
    seed, n_frames, n_cb = 78, 1024, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, seed, n_src, n_frames, n_cb, postfx, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    spec = scenario.random_spec(seed, n_src, clip_len=9000, cube=10.0, start=0.06, kinds=("frames",), gain_db=(None,))
    ob = scenario.play_all(scenario.OracleBackend(), spec)
    interval = np.float32(1.0) / np.float32(48000)
    ref = np.stack([apply_postfx_numpy(ob.sample(interval, n_frames).copy(), postfx) for _ in range(n_cb)])
    np.testing.assert_array_equal(results[0][3], results[1][3])
    assert np.abs(results[0][3] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30)


@pytest.mark.parametrize("world,n_src,postfx", [(2, 13, 0), (2, 8, 1)])
def test_sharded_scene_reduce_gloo(world, n_src, postfx):
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenario
    from oddio_amd import sharding

    seed, n_frames, n_cb = 77, 1024, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, n_src, n_frames, n_cb, postfx, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # shards tile [0, n_src) exactly once, contiguously
    assert results[0][1] == 0 and results[-1][2] == n_src
    assert all(results[i][2] == results[i + 1][1] for i in range(world - 1))
    # single-process reference
    spec = scenario.random_spec(seed, n_src, clip_len=9000, cube=10.0, start=0.06)
    ob = scenario.play_all(scenario.OracleBackend(), spec)
    interval = np.float32(1.0) / np.float32(48000)
    ref = np.stack([apply_postfx_numpy(ob.sample(interval, n_frames).copy(), postfx) for _ in range(n_cb)])
    for rank, lo, hi, got in results:
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    np.testing.assert_array_equal(results[0][3], results[1][3])   # all_reduce: identical on every rank


def test_shard_range_properties():
    from oddio_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 2097152, 262145):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)
