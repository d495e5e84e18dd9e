"""Multi-GPU for the hot path: one process per GPU.

The path shards naturally: sources are independent until the final
per-frame sum (src/spatial.rs:460), independent scenes share nothing.

  * scene-parallel (BASELINE configs[3]): every rank owns whole scenes -> no communication at all;
    nothing here is needed except `shard_range` to split a list of scenes.
  * one huge scene (BASELINE configs[4]): contiguous source-index shards (keeps the reference's
    reverse walk order inside a shard); every rank renders a partial 2*N-float stereo buffer; ONE
    all-reduce (sum) of that 8 KiB buffer per callback, issued by libodd_hip.so itself on the
    scene's stream (`oddio_hip_scene_reduce_init`, RCCL over xGMI; latency bound, ring vs tree is
    irrelevant at this size); the post-mix soft clip (Reinhard/Tanh wraps the *scene*,
    src/reinhard.rs:7-10) runs after the sum.  The summation order differs from the single
    sequential f32 sum of the reference, so the result is within the tolerance policy of
    SURVEY.md H2, not bit-identical.

`reduce_stereo` is the same reduction through torch.distributed (used by the gloo CPU tests, where
there is no RCCL).
"""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of `n_items` for `rank` (first n % world ranks get one more)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def reduce_stereo(partial, dist=None, group=None, dst: int | None = 0):
    """Sum the ranks' partial stereo buffers (torch tensor, [n_frames, 2] f32, on the backend's
    device).  dst=None -> all_reduce (every rank gets the mix); otherwise reduce to `dst`."""
    if dist is None:
        import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return partial
    if dst is None:
        dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(partial, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return partial


def exchange_unique_id(dist=None, src: int = 0) -> bytes:
    """Rank `src` makes the reduce group's id (ncclGetUniqueId through the C ABI) and hands it to the
    other ranks over the process group the host program already has (any backend)."""
    from . import api
    if dist is None:
        import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return api.reduce_unique_id()
    box = [api.reduce_unique_id() if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def exchange_p2p_handle(scene, rank: int, world: int, dist=None, src: int = 0) -> None:
    """Join the deterministic peer-to-peer reduce (`oddio_hip_scene_reduce_init_p2p`): rank `src` creates the slab and
    hands its handle to the other ranks over the host program's process group (any backend)."""
    if dist is None:
        import torch.distributed as dist
    if src != 0:
        raise ValueError("the slab of the peer-to-peer reduce lives on rank 0 (the rank-ordered sum runs there): src must be 0")
    if world > 1 and not (dist.is_initialized() and dist.get_world_size() == world):
        raise RuntimeError(f"exchange_p2p_handle: a process group of {world} ranks is needed to hand the slab's handle over "
                           f"(torch.distributed is {'not initialised' if not dist.is_initialized() else 'of another size'})")
    if rank == src:
        handle = scene.reduce_init_p2p(0, world)
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast_object_list([handle], src=src)
    else:
        box = [None]
        dist.broadcast_object_list(box, src=src)
        scene.reduce_init_p2p(rank, world, box[0])
    # every rank has the slab open before any of them renders a callback: the in-kernel waits of the reduce are bounded (2 s),
    # and a rank that started sampling that much ahead of the others would fail the group (oddio_hip_scene_reduce_init_p2p)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class ShardedSpatialScene:
    """One logical SpatialScene whose sources are split into contiguous index shards, one per rank.

    `play_frames_batch` is called with the rank's OWN shard (`self.shard` = [lo, hi) of the global
    source index); `sample_device` renders it on the HIP path, the library sums the partial stereo
    buffers on the scene's stream -- RCCL all-reduce (`reduce="rccl"`) or the rank-ordered peer-to-peer
    reduce (`reduce="p2p"`) -- and then applies the scene-level post filter, so every rank ends up with
    the scene's mix."""

    def __init__(self, device: int, n_sources_total: int, max_frames: int, rank: int, world: int, unique_id: bytes | None,
                 postfx: int = 0, reduce: str = "rccl", dist=None, scene_factory=None):
        """reduce: "rccl" (the library's all-reduce, needs `unique_id`), "p2p" (the library's rank-ordered reduce; the
        handle travels over `dist`), or "dist" (the partial buffer is reduced by the host program's own
        torch.distributed group -- any backend -- and the post filter applied afterwards with `postfx_host`; what the
        gloo tests use).  `scene_factory(max_sources, max_frames) -> (control, scene)` replaces the HIP scene (tests)."""
        self.rank, self.world, self.device = rank, world, device
        self.reduce, self.dist, self.postfx = reduce, dist, postfx
        self.shard = shard_range(n_sources_total, world, rank)
        lo, hi = self.shard
        if scene_factory is None:
            import oddio_amd as oa
            self.control, self.scene = oa.SpatialScene(device=device, max_sources=max(hi - lo, 1), max_frames=max_frames)
        else:
            self.control, self.scene = scene_factory(max(hi - lo, 1), max_frames)
        if reduce == "dist":
            return                        # the scene-level filter follows the cross-rank sum: applied in sample()
        if postfx:
            self.scene.set_postfx(postfx)
        if reduce == "p2p":
            # summed in rank order on rank 0 (deterministic; the ranks may share one GPU)
            exchange_p2p_handle(self.scene, rank, world, dist)
        elif unique_id is not None:
            self.scene.reduce_init(rank, world, unique_id)

    def play_frames_batch(self, frames_list, start_seconds, positions, velocities, radii):
        assert len(frames_list) == self.shard[1] - self.shard[0], "pass this rank's shard of the source list"
        if not frames_list:
            return []
        return self.control.play_frames_batch(frames_list, np.asarray(start_seconds), np.asarray(positions), np.asarray(velocities),
                                              np.asarray(radii))

    def sample_device(self, interval, dev_ptr: int, n_frames: int):
        assert self.reduce != "dist", "the host-side reduce works on host buffers: use sample()"
        self.scene.sample_device(interval, dev_ptr, n_frames)

    def sample(self, interval, out: np.ndarray):
        out = self.scene.sample(interval, out)
        if self.reduce == "dist":
            import torch
            part = torch.from_numpy(out)
            reduce_stereo(part, self.dist, dst=None)
            out[...] = postfx_host(part.numpy(), self.postfx)
        return out


def postfx_host(x: np.ndarray, postfx: int) -> np.ndarray:
    """The scene-level soft clip on a host buffer, in exact f32 operations: Reinhard `x / (1 + |x|)` (src/reinhard.rs:32),
    Tanh `tanhf(x)` (src/tanh.rs:26)."""
    x = np.asarray(x, dtype=np.float32)
    if postfx == 1:
        return (x / (np.float32(1.0) + np.abs(x))).astype(np.float32)
    if postfx == 2:
        return np.tanh(x).astype(np.float32)
    return x
