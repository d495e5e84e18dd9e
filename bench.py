#!/usr/bin/env python
"""bench.py -- mixed source-frames/sec of the SpatialScene hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one SpatialScene callback: every source of the scene resampled (Doppler + propagation
delay), gain-ramped, panned and summed into one 1024-frame 48 kHz stereo buffer that stays in HBM.
Workload at N=1 = BASELINE.json configs[2] (the configuration the metric's roofline target is
quoted on): one SpatialScene with 262 144 moving FramesSignal sources, each with its own
65 536-sample clip (64 GiB of clips, generated on the GPU).  At N>1 every rank renders its own
independent scene of the same size (BASELINE configs[3] pattern: scene-parallel, no data-path
collective) => weak scaling; `value` is the whole-job aggregate.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     achieved = algorithmic bytes per mix launch / average mix-kernel duration measured
               with hipEvents on the scene's stream over the timed region
  cpu_baseline the C restatement of the reference algorithm (oracle/, kind "port"), single thread
               (the reference's execution model), on a bounded slice of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RATE = 48000
N_FRAMES = 1024
HBM_PEAK_GBPS = 8000.0   # MI355X spec (MI355X_MICROARCH.md); ~6300 is what a float4 copy reaches
PARAM_BYTES = 128        # per-source parameter/state budget P of SURVEY.md section 8(d)


def algorithmic_bytes(n_sources: int, n_frames: int) -> float:
    """B = S * (4 * (N * r + 32) + P) + 8 * N with r = 1 (mean resample ratio)."""
    return n_sources * (4.0 * (n_frames + 32) + PARAM_BYTES) + 8.0 * n_frames


def build_gpu_scene(device: int, n_sources: int, clip_len: int, seed: int, start_seconds: float, n_clips: int = 0):
    """Clips are synthesised on the GPU (A*sin(2*pi*f*n/48000), f from the shared generator) and
    borrowed zero-copy as oddio Frames; scene parameters come from oddio_amd.synth."""
    import torch

    import oddio_amd as oa
    from oddio_amd import synth

    sc = synth.make_scene(seed, n_sources)
    dev = torch.device("cuda", device)
    n_clips = n_sources if n_clips <= 0 else min(n_clips, n_sources)   # < n_sources: clips are shared, scattered
    clips = torch.empty((n_clips, clip_len), dtype=torch.float32, device=dev)
    # Host-side set-up first (handles, the set insertion), the GPU-side clip synthesis last: the timed
    # callbacks then follow seconds of GPU load instead of seconds of idling (clock ramp).
    control, scene = oa.SpatialScene(device=device, max_sources=n_sources, max_frames=N_FRAMES)
    base = clips.data_ptr()
    frames = [oa.Frames.from_device_ptr(RATE, base + 4 * clip_len * i, clip_len, device=device, copy=False) for i in range(n_clips)]
    if n_clips < n_sources:
        # scattered assignment: sources that share a clip must not be neighbours in the set walk, or
        # their windows would hit in L2/MALL and flatter the HBM figure
        pick = ((np.arange(n_sources, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(20)) % np.uint64(n_clips)
    handles = control.play_frames_batch(frames if n_clips == n_sources else [frames[int(k)] for k in pick],
                                        np.full(n_sources, start_seconds), sc["position"], sc["velocity"], sc["radius"])
    ids = np.array([h.id for h in handles], dtype=np.uint32)
    # a zero-frame callback is `sample(interval, &mut [])`: set.update() inserts the sources, no time passes
    prime = torch.zeros((1, 2), dtype=torch.float32, device=dev)
    scene.sample_device(np.float32(1.0) / np.float32(RATE), prime.data_ptr(), 0)
    scene.synchronize()
    freq = torch.from_numpy(sc["freq_hz"][:n_clips]).to(dev).double()
    n = torch.arange(clip_len, device=dev, dtype=torch.float64)
    chunk = max(1, (1 << 28) // clip_len)
    for s0 in range(0, n_clips, chunk):
        s1 = min(n_clips, s0 + chunk)
        ph = (2.0 * np.pi / RATE) * freq[s0:s1, None] * n[None, :]
        clips[s0:s1] = torch.sin(ph).float()
        del ph
    torch.cuda.synchronize(dev)
    return {"control": control, "scene": scene, "clips": clips, "frames": frames, "handles": handles, "ids": ids, "spec": sc}


def cpu_baseline(seed: int, budget_s: float = 12.0) -> dict:
    """Single-thread C oracle on a 4096-source slice of the same workload (same generator)."""
    from oddio_amd import synth
    from oracle import oracle_c as oc

    n_src, clip_len, start = 4096, 40960, 0.6
    sc = synth.make_scene(seed, n_src)
    frames = [oc.Frames(RATE, synth.sine_clip(sc["freq_hz"][i], clip_len, RATE)) for i in range(n_src)]
    cb_per_round = (clip_len - int(start * RATE)) // N_FRAMES - 1
    out = np.zeros((N_FRAMES, 2), dtype=np.float32)
    total_cb, t_total = 0, 0.0
    while t_total < budget_s:
        scene = oc.SpatialScene()
        for i in range(n_src):
            scene.play(oc.FramesSignal(frames[i], start), oc.SpatialOptions(sc["position"][i], sc["velocity"][i], sc["radius"][i]))
        oc.run(scene, RATE, out)   # first callback also drains the insert queue: untimed
        t0 = time.perf_counter()
        for _ in range(cb_per_round):
            oc.run(scene, RATE, out)
        t_total += time.perf_counter() - t0
        total_cb += cb_per_round
        assert len(scene) == n_src
        del scene
    sfps = n_src * N_FRAMES * total_cb / t_total
    return {
        "value": sfps, "unit": "source-frames/s", "cores": 1, "kind": "port",
        "sample": f"{n_src}-source slice of the workload (same generator), {total_cb} callbacks of {N_FRAMES} frames, "
                  f"C restatement of the reference (oracle/oddio_oracle.c, -O2 -ffp-contract=off), single thread = the reference's one audio thread",
        "host_cores_available": os.cpu_count(),
        "max_realtime_sources_per_core": sfps / RATE,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: after the idle scene set-up the GPU needs ~100 callbacks (~30 ms) of load before the mix kernel
    # reaches its steady duration (clock ramp); warmup + steps stay below the
    # 320-callback motion reset, so the timed region is the hot path only
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--sources", type=int, default=262144, help="sources per GPU (config 3: 262144; config 2: 4096)")
    ap.add_argument("--clip-len", type=int, default=65536)
    ap.add_argument("--clips", type=int, default=0,
                    help="distinct clips (default: one per source, the BASELINE workload); fewer lets --sources exceed what 288 GB of "
                         "own clips allows, e.g. to run AT the reported max_realtime_sources")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--mode", choices=["scenes", "sharded"], default="scenes",
                    help="N>1: 'scenes' = one independent scene per GPU (configs[3] pattern, no collective); "
                         "'sharded' = ONE scene of N*sources split into contiguous index shards with an RCCL sum-reduce "
                         "of the 8 KiB stereo buffer per callback (configs[4] pattern)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    device = local_rank if world > 1 else 0
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        dist = dist_mod

    S, L = args.sources, args.clip_len
    # clips start 1.0 s in: the propagation delay (<= 0.25 s at set-up) may grow by the drift of the
    # constant-velocity sources (<= 34.6 m/s) for `reset_every` callbacks without reading before the clip
    start_seconds = 1.0
    g = build_gpu_scene(device, S, L, args.seed + rank, start_seconds, args.clips)
    scene, control = g["scene"], g["control"]
    out = torch.zeros((N_FRAMES, 2), dtype=torch.float32, device=torch.device("cuda", device))
    interval = np.float32(1.0) / np.float32(RATE)
    # callbacks a clip lasts before sources would run off its end; rewind before that
    span = max(1, (L - int(start_seconds * RATE)) // N_FRAMES - 8)
    rewind_seconds = -float(span * N_FRAMES) / RATE
    reset_every = 320   # callbacks; bounds the drift of the constant-velocity sources (a host-side batch set_motion)

    step_no = 0

    sharded = args.mode == "sharded" and dist is not None
    if sharded:
        # one logical scene: this rank's scene IS its contiguous shard (seed + rank streams); the
        # partial stereo buffers are summed over RCCL on the same stream as the kernels
        from oddio_amd import sharding
        scene.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def one_step():
        nonlocal step_no
        if step_no and step_no % span == 0:
            scene.seek_all(rewind_seconds)              # Seek::seek on every source (a tiny kernel, timed)
        if step_no and step_no % reset_every == 0:
            control.set_motion_batch(g["ids"], g["spec"]["position"], g["spec"]["velocity"], True)
        scene.sample_device(interval, out.data_ptr(), N_FRAMES)
        if sharded:
            sharding.reduce_stereo(out, dist, dst=0)    # 2048 floats, sum, to rank 0
        step_no += 1

    for _ in range(args.warmup):
        one_step()
    scene.set_profiling(True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    hist = scene.kernel_ms_history(min(args.steps, 512))
    assert len(scene) == S, "sources finished inside the timed region"
    assert bool(torch.isfinite(out).all()) and float(out.abs().max()) > 0.0

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cuda", device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_units = float(S) * N_FRAMES * args.steps * world
        value = total_units / elapsed
        mix_ms = float(hist[:, 1].mean())
        b_alg = algorithmic_bytes(S, N_FRAMES)
        achieved = b_alg / (mix_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("sources") == S and j.get("kernel", "").startswith("spatial_mix"):
                    traffic = j.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "mixed source-frames/sec (48 kHz stereo)",
            "value": value,
            "unit": "source-frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ((f"BASELINE configs[2]: SpatialScene, {S} moving FramesSignal sources (own {L}-sample clip each), " if args.clips <= 0 or args.clips >= S
                              else f"real-time confirmation run: SpatialScene, {S} moving FramesSignal sources sharing {args.clips} clips of {L} samples, ")
                             +
                             f"Doppler + propagation delay, 48 kHz stereo, {N_FRAMES}-frame callbacks"
                             + ("" if world == 1 else (f"; ONE scene of {world * S} sources in {world} index shards + RCCL reduce (configs[4] pattern)" if sharded
                                                      else f"; {world} independent scenes, one per GPU (configs[3] pattern)"))),
                "sources_per_gpu": S, "frames_per_callback": N_FRAMES, "sample_rate": RATE, "clip_len": L,
                "parallelism": ("single-gpu" if world == 1 else ("source-sharded scene + RCCL stereo-buffer reduce" if sharded else "scene-parallel")),
            },
            "max_realtime_sources": value / RATE,
            "realtime_factor_per_gpu": (N_FRAMES / RATE) / (elapsed / args.steps),
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "kernel": "spatial_mix", "avg_kernel_ms": mix_ms, "algorithmic_bytes_per_launch": b_alg,
                "prepass_ms": float(hist[:, 0].mean()), "reduce_ms": float(hist[:, 2].mean()),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.seed, args.cpu_budget)
            line["cpu_baseline"]["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
