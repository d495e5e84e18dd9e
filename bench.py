#!/usr/bin/env python
"""bench.py -- mixed source-frames/sec of the SpatialScene hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 spawns its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one SpatialScene callback: every source of the scene resampled (Doppler + propagation
delay), gain-ramped, panned and summed into one 1024-frame 48 kHz stereo buffer that stays in HBM.
Workload at N=1 = BASELINE.json configs[2] (the configuration the metric's roofline target is
quoted on): one SpatialScene with 262 144 moving FramesSignal sources, each with its own
65 536-sample clip (64 GiB of clips, generated on the GPU).
N>1, default (`--mode scenes`): every rank renders its own independent scene of the SAME size
(BASELINE configs[3] pattern: scene-parallel, no data-path collective; per-GPU work is fixed across
N, so the driver's scaling efficiency compares like with like; `--sources 65536` gives configs[3]'s
literal 8 x 65 536).  `--mode sharded`: ONE seeded scene of N * sources split into contiguous index
shards (BASELINE configs[4]); the partial stereo buffers are summed by the library's RCCL
all-reduce on the kernels' stream.  `value` is the whole-job aggregate in both modes.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     achieved = algorithmic bytes per mix launch / average mix-kernel duration measured
               with hipEvents on the scene's stream over the timed region; frac_callback = the same
               bytes / the whole callback's wall time (prepass, reduce and launch gaps included)
  cpu_baseline the C restatement of the reference algorithm (oracle/, kind "port") on this box's
               host cores: one thread (the reference's execution model) and all cores
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RATE = 48000
N_FRAMES = 1024
HBM_PEAK_GBPS = 8000.0   # MI355X spec (MI355X_MICROARCH.md); a float4 copy reaches ~6300-6650
PARAM_BYTES = 128        # per-source parameter/state budget P of SURVEY.md section 8(d)


def algorithmic_bytes(n_sources: int, n_frames: int) -> float:
    """B = S * (4 * (N * r + 32) + P) + 8 * N with r = 1 (mean resample ratio)."""
    return n_sources * (4.0 * (n_frames + 32) + PARAM_BYTES) + 8.0 * n_frames


def build_gpu_scene(device: int, n_sources: int, clip_len: int, seed: int, start_seconds: float, n_clips: int = 0,
                    first_index: int = 0, scene_factory=None):
    """Clips are synthesised on the GPU (A*sin(2*pi*f*n/48000), f from the shared generator) and
    borrowed zero-copy as oddio Frames; scene parameters come from oddio_amd.synth (`first_index`:
    this rank's offset into the one seeded source list of a sharded scene)."""
    import torch

    import oddio_amd as oa
    from oddio_amd import synth

    sc = synth.make_scene(seed, n_sources, first_index=first_index)
    dev = torch.device("cuda", device)
    n_clips = n_sources if n_clips <= 0 else min(n_clips, n_sources)   # < n_sources: clips are shared, scattered
    clips = torch.empty((n_clips, clip_len), dtype=torch.float32, device=dev)
    # Host-side set-up first (handles, the set insertion), the GPU-side clip synthesis last: the timed
    # callbacks then follow seconds of GPU load instead of seconds of idling.
    if scene_factory is None:
        control, scene = oa.SpatialScene(device=device, max_sources=n_sources, max_frames=N_FRAMES)
    else:
        control, scene = scene_factory()
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
    torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
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


# ---------------------------------------------------------------------------------------------------
# CPU baseline: the C restatement (oracle/oddio_oracle.c) on the host cores of this box
# ---------------------------------------------------------------------------------------------------
def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_scene(oc, clips, first, n_src, seed, start):
    """An oracle SpatialScene of sources [first, first + n_src) of the bench generator; source i plays
    clip i % len(clips) (borrowed)."""
    from oddio_amd import synth
    sc = synth.make_scene(seed, n_src, first_index=first)
    idx = (np.arange(first, first + n_src) % clips.shape[0]).astype(np.uint32)
    scene = oc.SpatialScene()
    scene.play_frames_bulk(RATE, clips, start, sc["position"], sc["velocity"], sc["radius"], clip_of=idx)
    return scene


def _time_scene(oc, make_scene, cb_per_round, budget_s):
    out = np.zeros((N_FRAMES, 2), dtype=np.float32)
    total_cb, t_total = 0, 0.0
    while t_total < budget_s:
        scene = make_scene()
        oc.run(scene, RATE, out)   # first callback also drains the insert queue: untimed
        t0 = time.perf_counter()
        for _ in range(cb_per_round):
            oc.run(scene, RATE, out)
        t_total += time.perf_counter() - t0
        total_cb += cb_per_round
        del scene
    return total_cb, t_total


def parity_check(oc, bank, seed: int, start: float, device: int, n_src: int) -> dict:
    """Part of the cpu_baseline leg (the oracle is the checker): ONE callback of `n_src` sources of the bench generator
    -- the headline source count, clips drawn from the baseline's bank -- rendered by the oracle (the reference's
    sequential f32 sum, and the same contributions accumulated in f64) and by the HIP scene in both of its modes."""
    import torch

    import oddio_amd as oa
    from oddio_amd import synth
    sc = synth.make_scene(seed, n_src)
    idx = ((np.arange(n_src, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(20)) % np.uint64(bank.shape[0])
    idx = idx.astype(np.uint32)
    interval = np.float32(1.0) / np.float32(RATE)
    ref = np.zeros((N_FRAMES, 2), dtype=np.float32)
    scene = oc.SpatialScene()
    scene.play_frames_bulk(RATE, bank, start, sc["position"], sc["velocity"], sc["radius"], clip_of=idx)
    t0 = time.perf_counter()
    oc.run(scene, RATE, ref)
    t_ref = time.perf_counter() - t0
    del scene
    scene = oc.SpatialScene()
    scene.play_frames_bulk(RATE, bank, start, sc["position"], sc["velocity"], sc["radius"], clip_of=idx)
    ref64 = scene.sample_f64acc(interval, N_FRAMES)
    del scene
    dev_bank = torch.from_numpy(bank).to(torch.device("cuda", device))
    clip_len = bank.shape[1]
    frames = [oa.Frames.from_device_ptr(RATE, dev_bank.data_ptr() + 4 * clip_len * i, clip_len, device=device, copy=False) for i in range(bank.shape[0])]
    got = {}
    for mode, name in ((oa.MODE_FAST, "fast"), (oa.MODE_ORDERED, "ordered"), (oa.MODE_TRACKED, "tracked")):
        control, hscene = oa.SpatialScene(device=device, max_sources=n_src, max_frames=N_FRAMES)
        hscene.set_mode(mode)
        control.play_frames_batch([frames[int(k)] for k in idx], np.full(n_src, start), sc["position"], sc["velocity"], sc["radius"])
        got[name] = hscene.sample_n(interval, N_FRAMES)
        del control, hscene
    scale = float(np.abs(ref).max())
    rel = lambda a, b: float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) / scale   # noqa: E731
    # TRACKED is a statistical statement, so it is also measured on the two scenes built against it
    # (synth.adversarial_scene: coherent sources cancelling in pairs; a running sum parked at a power of two), same source count
    adversarial = {}
    for kind in ("cancelling", "parked"):
        adv = synth.adversarial_scene(kind, n_src)
        o = oc.SpatialScene()
        a_start = 0.06                   # (inside the bank's 8 192-sample clips, behind the propagation delay)
        o.play_frames_bulk(RATE, adv["bank"], a_start, adv["position"], adv["velocity"], adv["radius"], clip_of=adv["clip_of"])
        aref = np.zeros((N_FRAMES, 2), dtype=np.float32)
        oc.run(o, RATE, aref)
        del o
        abank = torch.from_numpy(adv["bank"]).to(torch.device("cuda", device))
        al = adv["bank"].shape[1]
        aframes = [oa.Frames.from_device_ptr(RATE, abank.data_ptr() + 4 * al * i, al, device=device, copy=False) for i in range(adv["bank"].shape[0])]
        res = {"max_abs_reference": float(np.abs(aref).max())}
        for mode, name in ((oa.MODE_TRACKED, "tracked"), (oa.MODE_FAST, "fast")):
            control, hscene = oa.SpatialScene(device=device, max_sources=n_src, max_frames=N_FRAMES)
            hscene.set_mode(mode)
            control.play_frames_batch([aframes[int(k)] for k in adv["clip_of"]], np.full(n_src, a_start), adv["position"], adv["velocity"], adv["radius"])
            res[f"{name}_rel_err_vs_reference"] = float(np.abs(hscene.sample_n(interval, N_FRAMES) - aref).max()) / res["max_abs_reference"]
            del control, hscene
        adversarial[kind] = res
        del aframes, abank
    tracked_worst = max([rel(got["tracked"], ref)] + [a["tracked_rel_err_vs_reference"] for a in adversarial.values()])
    return {
        "sources": n_src, "callbacks": 1, "max_abs_reference": scale,
        "ordered_bit_exact": bool(np.array_equal(got["ordered"], ref)),
        "fast_rel_err_vs_reference": rel(got["fast"], ref),
        "tracked_rel_err_vs_reference": rel(got["tracked"], ref),
        "tracked_worst_case_rel_err_vs_reference": tracked_worst,      # over this scene and the two adversarial ones below
        "tracked_conforms": bool(tracked_worst <= 1e-5),
        "adversarial_scenes": adversarial,
        "fast_rel_err_vs_f64": rel(got["fast"], ref64),
        "reference_rel_err_vs_f64": rel(ref, ref64),
        "tolerance": 1e-5,
        "note": "relative to max|reference|; the reference is the oracle's sequential f32 sum in the reference's walk order "
                "(oracle/oddio_oracle.c), f64 = the same contributions accumulated in f64.  At this source count the reference's "
                "own f32 sum is further than 1e-5 from the exact sum, so no other sum order can be within 1e-5 of it: ORDERED is the same "
                "order (bit-exact); TRACKED restarts every workgroup's running sums at the prefix of a first pass's partial sums and so "
                "repeats the reference's rounding errors (~1e-6); FAST is the deterministic tree sum the throughput is quoted in.",
        "oracle_seconds_per_callback_1_thread": t_ref,
    }


def cpu_worker(spec: str) -> None:
    """One process of the all-cores leg: `per` sources of the bench generator starting at `first`, on a small clip bank
    of its own (64 clips; sources share them, which flatters the CPU's caches, never the GPU)."""
    from oddio_amd import synth
    from oracle import oracle_c as oc
    per, rounds, seed, first = (int(x) for x in spec.split(","))
    base_len, reps, start, n_bank = 40960, 4, 0.6, 64
    clip_len = base_len * reps
    cb_per_round = (clip_len - int(start * RATE)) // N_FRAMES - 1
    sc = synth.make_scene(seed, n_bank, first_index=first)
    one = np.sin((2.0 * np.pi / RATE) * sc["freq_hz"][:, None].astype(np.float64) * np.arange(base_len, dtype=np.float64)[None, :]).astype(np.float32)
    bank = np.tile(one, (1, reps))
    out = np.zeros((N_FRAMES, 2), dtype=np.float32)
    scenes = []
    for _ in range(rounds):
        scene = _oracle_scene(oc, bank, first, per, seed, start)
        oc.run(scene, RATE, out)                     # insert queue drained, untimed
        scenes.append(scene)
    print("ready", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    acc = 0.0
    for scene in scenes:
        for _ in range(cb_per_round - 1):
            oc.run(scene, RATE, out)
        acc += float(np.abs(out).sum())
    print(rounds * (cb_per_round - 1), time.perf_counter() - t0, acc, flush=True)


def cpu_baseline(seed: int, budget_s: float = 20.0, parity_device: int | None = None, parity_sources: int = 0) -> dict:
    """single thread (the reference's execution model: one audio thread) on configs 1, 2
    and a 16 384-source slice of config 3, then ALL host cores by scene sharding (T independent partial
    scenes, partial buffers summed at the end -- what a user of the reference would have to do)."""
    from oddio_amd import synth
    from oracle import oracle_c as oc

    base_len, reps, start, n_bank = 40960, 4, 0.6, 4096
    clip_len = base_len * reps                                  # 3.4 s of audio: rounds of 130 callbacks
    cb_per_round = (clip_len - int(start * RATE)) // N_FRAMES - 1
    sc = synth.make_scene(seed, n_bank)
    n = np.arange(base_len, dtype=np.float64)
    bank = np.empty((n_bank, clip_len), dtype=np.float32)
    for s0 in range(0, n_bank, 256):
        one = np.sin((2.0 * np.pi / RATE) * sc["freq_hz"][s0:s0 + 256, None].astype(np.float64) * n[None, :]).astype(np.float32)
        bank[s0:s0 + 256] = np.tile(one, (1, reps))             # the content does not matter for the timing, the length does
    legs = {}
    share = budget_s / 4.0

    # config 1: Mixer of 64 MonoToStereo<Sine> (examples/simple.rs style), plumbing
    def mixer64():
        m = oc.Mixer(channels=2)
        for k in range(64):
            m.play(oc.MonoToStereo(oc.Sine(float(sc["phase"][k]), 110.0 * 2.0 ** (k / 12.0))))
        return m
    out = np.zeros((N_FRAMES, 2), dtype=np.float32)
    m = mixer64()
    oc.run(m, RATE, out)
    t0, cbs = time.perf_counter(), 0
    while time.perf_counter() - t0 < min(share, 2.0):
        oc.run(m, RATE, out)
        cbs += 1
    legs["config1_mixer_64_sines_1_thread"] = 64 * N_FRAMES * cbs / (time.perf_counter() - t0)

    # config 2 shape, one thread (own clip per source)
    cb, t = _time_scene(oc, lambda: _oracle_scene(oc, bank, 0, 4096, seed, start), cb_per_round, share)
    single = 4096 * N_FRAMES * cb / t
    legs["config2_4096_sources_1_thread"] = single
    # 16 384-source slice of config 3, one thread (clips shared 4 ways: flatters the CPU's caches, never the GPU)
    cb, t = _time_scene(oc, lambda: _oracle_scene(oc, bank, 0, 16384, seed, start), cb_per_round, share)
    legs["config3_slice_16384_sources_1_thread"] = 16384 * N_FRAMES * cb / t

    parity = parity_check(oc, bank, seed, start, parity_device, parity_sources) if parity_device is not None and parity_sources > 0 else None

    # all cores: T PROCESSES, each with its own partial scene of 1024 sources (what a user of the reference would have
    # to do: one audio thread per partial scene, partial buffers summed).  Processes, not threads: in-process threads
    # around the C calls did not scale on either box (3.8 % parallel efficiency on the 256-thread GPU host), separate
    # processes do.  Every worker sets up, reports ready, and all are released together.
    def run_workers(T, per, rounds):
        """T worker processes, released together -> (wall seconds, slowest worker's seconds, total callbacks)."""
        workers = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", f"{per},{rounds},{seed},{t_ * per}"],
                                    stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for t_ in range(T)]
        for w in workers:
            assert w.stdout.readline().strip() == "ready", "cpu worker failed to start"
        t0 = time.perf_counter()
        for w in workers:
            w.stdin.write("go\n")
            w.stdin.flush()
        done = [w.stdout.readline().split() for w in workers]       # "<callbacks> <seconds> <checksum>"
        wall = time.perf_counter() - t0
        for w in workers:
            w.wait()
        assert all(np.isfinite(float(d[2])) for d in done)
        return wall, max(float(d[1]) for d in done), sum(int(d[0]) for d in done)

    # How many cores does this job really get?  os.cpu_count() and the affinity mask both said 256 on the GPU box while
    # 256 busy processes ran 30x slower than one (a container CPU quota neither of them shows): measured instead, with
    # short runs of the same worker -- the largest T (x2 steps) whose slowest worker is within 1.6x of a lone worker's time.
    n_max = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    _, t_one, _ = run_workers(1, 256, 1)
    T = 1
    while T * 2 <= n_max:
        _, t_slow, _ = run_workers(T * 2, 256, 1)
        if t_slow > 1.6 * t_one:
            _, t_slow, _ = run_workers(T * 2, 256, 1)           # (once more: a busy host makes a single short run noisy)
            if t_slow > 1.6 * t_one:
                break
        T *= 2
    per, rounds = 1024, 4
    t_all, slowest, n_cb_all = run_workers(T, per, rounds)
    all_cores = per * N_FRAMES * n_cb_all / t_all
    efficiency = all_cores / (T * single)
    legs[f"usable_cores_{T}_processes_x_{per}_sources"] = all_cores
    if efficiency < 0.5:
        print(f"bench.py: WARNING: the {T}-process CPU baseline reached only {efficiency:.2f} of {T} x the single-thread rate "
              f"(slowest worker {slowest:.2f} s of {t_all:.2f} s wall): not a usable baseline on this host", file=sys.stderr, flush=True)
    return {
        "value": single, "unit": "source-frames/s", "cores": 1, "kind": "port",
        "sample": f"4096-source slice of the workload (same generator), {cb_per_round}-callback rounds of {N_FRAMES} frames, "
                  f"C restatement of the reference (oracle/oddio_oracle.c, -O2 -ffp-contract=off, not rustc output), "
                  f"single thread = the reference's one audio thread",
        "all_cores": {"value": all_cores, "cores": T, "parallel_efficiency": efficiency, "valid": bool(efficiency >= 0.5),
                      "sample": f"{T} processes (the largest count, in x2 steps up to the {n_max} of the affinity mask, that still ran at a lone worker's speed) x "
                                f"{per}-source partial scenes, {n_cb_all // T} callbacks each in {rounds} rounds, released together; "
                                f"wall {t_all:.2f} s, slowest worker {slowest:.2f} s"},
        "legs": legs,
        "parity": parity,
        "cpu_model": _cpu_model(),
        "host_cores_available": os.cpu_count(),
        "cores_in_affinity_mask": n_max,
        "cores_measured_usable": T,
        "max_realtime_sources_per_core": single / RATE,
        "max_realtime_sources_all_cores": all_cores / RATE,
    }


# ---------------------------------------------------------------------------------------------------
# --workload buffered: the path Gain / Speed sources take into a scene (play_buffered, src/spatial.rs:314-340,395-433)
# ---------------------------------------------------------------------------------------------------
BUF_MAX_DISTANCE = 100.0     # metres: ring = ceil((100 / 343 + 0.1) * 48000) + 1 = 18 795 samples per source
BUF_DURATION = 0.1


def buffered_algorithmic_bytes(speeds: np.ndarray, n_frames: int) -> dict:
    """Per callback: every source's leaf samples (N * speed), its ring extended by N samples, the ring window both ears read
    (N + 32), its parameters, and the stereo buffer:  B = sum_s [4 N speed_s + 4 N + 4 (N + 32) + P] + 8 N."""
    s = float(len(speeds))
    leaf = 4.0 * n_frames * float(np.sum(speeds.astype(np.float64)))
    ring_w = 4.0 * n_frames * s
    ring_r = 4.0 * (n_frames + 32) * s
    par = PARAM_BYTES * s
    return {"leaf_read": leaf, "ring_write": ring_w, "ring_read": ring_r, "parameters": par, "output": 8.0 * n_frames,
            "write_kernel": leaf + ring_w + par / 2, "read_kernel": ring_r + par / 2 + 8.0 * n_frames,
            "total": leaf + ring_w + ring_r + par + 8.0 * n_frames}


def buffered_traffic(n_sources: int):
    """HBM bytes per callback of the buffered path's kernels from the last PMC passes (tools/make_pmc_json.py), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_buffered_latest.json")
    try:
        j = json.load(open(path))
        return j.get("hbm_bytes_per_callback") if j.get("sources") == n_sources else None
    except Exception:
        return None


def buffered_cpu_and_parity(device: int, seed: int, budget_s: float) -> tuple[dict, dict]:
    """The oracle (C restatement of the reference) on the same kind of scene: throughput of one thread, and parity of the HIP
    path against it -- 4 096 Gain<Speed<FramesSignal>> sources, 6 callbacks with a gain store to every source before the third."""
    import oddio_amd as oa
    from oddio_amd import synth
    from oracle import oracle_c as oc
    n_src, clip_len, n_cb = 4096, 16384, 6
    sc = synth.make_scene(seed, n_src)
    st = synth.SplitMixStreams(seed ^ 0xB0F, n_src)
    speeds = st.uniform(0.9, 1.1)
    gains = st.uniform(0.5, 1.0)
    bank = np.stack([synth.sine_clip(float(sc["freq_hz"][k]), clip_len) for k in range(256)])
    interval = np.float32(1.0) / np.float32(RATE)

    def oracle_scene():
        ref = oc.SpatialScene()
        ctl = []
        frames = [oc.Frames(RATE, bank[k]) for k in range(256)]
        for i in range(n_src):
            sp = oc.Speed(oc.FramesSignal(frames[i % 256], 0.02))
            sp.set_speed(speeds[i])
            g = oc.Gain(sp)
            ref.play_buffered(g, oc.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1), BUF_MAX_DISTANCE, RATE, BUF_DURATION)
            ctl.append(g)
        return ref, ctl

    ref, ctl = oracle_scene()
    outs_ref, t_cb = [], []
    for cb in range(n_cb):
        if cb == 2:
            for i, g in enumerate(ctl):
                g.set_amplitude_ratio(gains[i])
        t0 = time.perf_counter()
        outs_ref.append(ref.sample_n(interval, N_FRAMES))
        t_cb.append(time.perf_counter() - t0)
    # more callbacks for the timing if the budget allows (the first ones include page faults of the rings)
    t_extra, n_extra = 0.0, 0
    while t_extra + sum(t_cb) < budget_s and n_extra < 3:
        t0 = time.perf_counter()
        ref.sample_n(interval, N_FRAMES)
        t_extra += time.perf_counter() - t0
        n_extra += 1
    per_cb = (sum(t_cb[1:]) + t_extra) / (n_cb - 1 + n_extra)
    del ref, ctl
    res = {}
    for mode, name in ((oa.MODE_ORDERED, "ordered"), (oa.MODE_FAST, "fast")):
        control, scene = oa.SpatialScene(device=device, max_sources=n_src, max_frames=N_FRAMES)
        scene.reserve_buffered(n_src)
        scene.set_mode(mode)
        frames = [oa.Frames.from_slice(RATE, bank[k]) for k in range(256)]
        ids = control.play_buffered_frames_batch([frames[i % 256] for i in range(n_src)], np.full(n_src, 0.02), [oa.FILTER_SPEED, oa.FILTER_GAIN],
                                                 np.stack([speeds, np.ones(n_src, np.float32)], axis=1), sc["position"], sc["velocity"], sc["radius"],
                                                 BUF_MAX_DISTANCE, RATE, BUF_DURATION)
        outs = []
        for cb in range(n_cb):
            if cb == 2:
                control.set_control_batch(ids, 1, gains)
            outs.append(scene.sample_n(interval, N_FRAMES))
        res[name] = outs
        assert scene.debug_buffered_slow() == 0
        scene.close()
    scale = max(float(np.abs(o).max()) for o in outs_ref)
    rel = max(float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) for a, b in zip(res["fast"], outs_ref)) / scale
    parity = {"sources": n_src, "callbacks": n_cb, "shape": "Gain<Speed<FramesSignal>>, a gain store to every source before the third callback",
              "ordered_bit_exact": bool(all(np.array_equal(a, b) for a, b in zip(res["ordered"], outs_ref))),
              "fast_rel_err_vs_reference": rel, "tolerance": 1e-5, "max_abs_reference": scale}
    cpu = {"value": n_src * N_FRAMES / per_cb, "unit": "source-frames/s", "cores": 1, "kind": "port",
           "sample": f"{n_src} buffered Gain<Speed<FramesSignal>> sources (same generator, 256 shared clips), {n_cb - 1 + n_extra} callbacks of {N_FRAMES} frames, "
                     f"C restatement of the reference (oracle/oddio_oracle.c, -O2 -ffp-contract=off, not rustc output), one thread = the reference's one audio thread",
           "cpu_model": _cpu_model(), "seconds_per_callback": per_cb}
    return cpu, parity


def multi_gpu_selfcheck(device: int, rank: int, world: int, dist, reduce_pref: str) -> dict:
    """Before anything is timed on more than one rank: ONE seeded 4 096-source scene rendered (a) whole on this rank's GPU and (b) in
    `world` contiguous index shards, one per rank, summed by the library's own reduce -- the RCCL all-reduce, or, when its communicator
    cannot be created (ncclCommInitRank fails: ranks that share a device, no usable librccl), the rank-ordered peer-to-peer reduce --
    and the two compared on every rank.  A sharded scene that does not reproduce the unsharded one ends the run here, not in a number.
    -> {"reduce": the reduce that works on this box, "rccl_error": what RCCL said if it failed, "max_rel_err": worst rank's}"""
    import torch

    import oddio_amd as oa
    from oddio_amd import sharding, synth
    S, CL, SEED = 4096, 8192, 777
    dev = torch.device("cuda", device)
    gen = torch.Generator(device=dev).manual_seed(SEED)
    clips = (torch.rand((S, CL), device=dev, dtype=torch.float32, generator=gen) * 2.0 - 1.0).contiguous()   # (the same on every rank)
    torch.cuda.synchronize(dev)   # (torch's stream made the clips; the scenes' streams do not wait for it)
    frames = [oa.Frames.from_device_ptr(RATE, clips.data_ptr() + 4 * CL * i, CL, device=device, copy=False) for i in range(S)]
    sc = synth.make_scene(SEED, S, cube=10.0)
    interval = np.float32(1.0) / np.float32(RATE)

    def render(control, scene, lo, hi):
        control.play_frames_batch(frames[lo:hi], np.full(hi - lo, 0.06), sc["position"][lo:hi], sc["velocity"][lo:hi], sc["radius"][lo:hi])
        outs = []
        for _ in range(2):
            o = torch.zeros((N_FRAMES, 2), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
            scene.sample_device(interval, o.data_ptr(), N_FRAMES)
            scene.synchronize()
            outs.append(o.cpu().numpy())
        return outs
    control, scene = oa.SpatialScene(device=device, max_sources=S, max_frames=N_FRAMES)
    whole = render(control, scene, 0, S)
    scene.close()
    kind, rccl_error = reduce_pref, None
    sh = None
    if kind == "rccl":
        uid = sharding.exchange_unique_id(dist)
        try:
            sh = sharding.ShardedSpatialScene(device, S, N_FRAMES, rank, world, uid, reduce="rccl", dist=dist)
            ok = True
        except Exception as e:      # ncclCommInitRank refused (oddio_hip_scene_reduce_init returns its error)
            ok, rccl_error = False, str(e)[:300]
        flags = [None] * world
        dist.all_gather_object(flags, (ok, rccl_error))
        if not all(f[0] for f in flags):
            rccl_error = next(f[1] for f in flags if not f[0])
            if sh is not None:
                sh.scene.close()
            sh, kind = None, "p2p"
    if sh is None:
        sh = sharding.ShardedSpatialScene(device, S, N_FRAMES, rank, world, None, reduce="p2p", dist=dist)
    lo, hi = sh.shard
    got = render(sh.control, sh.scene, lo, hi)
    sh.scene.close()
    err = max(float(np.abs(g - w).max()) / max(float(np.abs(w).max()), 1e-30) for g, w in zip(got, whole))
    errs = [None] * world
    dist.all_gather_object(errs, err)
    if max(errs) > 1e-5:
        raise SystemExit(f"multi-GPU self-check failed: the {world}-shard scene (reduce: {kind}) differs from the unsharded one by {max(errs):.3g} "
                         f"of its peak on some rank (per rank: {errs})")
    # ... and the conforming mode: the same scene sharded in ODDIO_HIP_MODE_TRACKED (the ranks exchange the totals of their first
    # pass between the passes: ncclAllGather / the slab's second block of rows) against the unsharded scene in ORDERED mode -- the
    # reference's sequential sum bit for bit (tests/test_hip_large_scene.py).  ODDIO_HIP_PAIR_MIN_GROUPS=1 (read when a scene is
    # created): shards of a few hundred sources take the two-pass kernels too, as the shards of a full-size scene do.
    control, scene = oa.SpatialScene(device=device, max_sources=S, max_frames=N_FRAMES)
    scene.set_mode(oa.MODE_ORDERED)
    ordered = render(control, scene, 0, S)
    scene.close()
    fast_result = {"sources": S, "reduce": kind, "rccl_error": rccl_error, "max_rel_err": max(errs)}
    saved = os.environ.get("ODDIO_HIP_PAIR_MIN_GROUPS")
    os.environ["ODDIO_HIP_PAIR_MIN_GROUPS"] = "1"
    err_t, exc_t = None, None
    try:
        uid = sharding.exchange_unique_id(dist) if kind == "rccl" else None
        sh = sharding.ShardedSpatialScene(device, S, N_FRAMES, rank, world, uid, reduce=kind, dist=dist)
        sh.scene.set_mode(oa.MODE_TRACKED)
        got = render(sh.control, sh.scene, lo, hi)
        sh.scene.close()
        err_t = max(float(np.abs(g - w).max()) / max(float(np.abs(w).max()), 1e-30) for g, w in zip(got, ordered))
    except Exception as e:           # noqa: BLE001  (reported: the FAST check above stands by itself)
        exc_t = f"{type(e).__name__}: {e}"[:300]
    finally:
        if saved is None:
            del os.environ["ODDIO_HIP_PAIR_MIN_GROUPS"]
        else:
            os.environ["ODDIO_HIP_PAIR_MIN_GROUPS"] = saved
    errs_t = [None] * world
    dist.all_gather_object(errs_t, (err_t, exc_t))
    if any(e_[1] for e_ in errs_t):
        fast_result["tracked_error"] = next(e_[1] for e_ in errs_t if e_[1])
        fast_result["tracked_max_rel_err_vs_ordered"] = None
        return fast_result
    errs_t = [e_[0] for e_ in errs_t]
    if max(errs_t) > 3e-6:       # (the conforming figure is then not reported; the FAST figures stand)
        fast_result["tracked_error"] = (f"the {world}-shard scene in TRACKED mode (reduce: {kind}) is {max(errs_t):.3g} of the peak from the unsharded scene's "
                                        f"sequential sum on some rank (per rank: {errs_t}); the bound is 3e-6")
        fast_result["tracked_max_rel_err_vs_ordered"] = None
        return fast_result
    del frames, clips
    return {"sources": S, "reduce": kind, "rccl_error": rccl_error, "max_rel_err": max(errs), "tracked_max_rel_err_vs_ordered": max(errs_t)}


def selfcheck_worker(args) -> None:
    """`bench.py --selfcheck-worker` (spawned by every rank of a multi-GPU run, see run_selfcheck): the sharded-scene self-check in a
    process of its own -- RCCL with more than one rank has never run on the boxes this was developed on, and a collective that hangs
    or crashes there must cost the run its self-check, not its JSON line.  Prints one JSON object."""
    fault = os.environ.get("ODDIO_BENCH_SELFCHECK_FAULT")        # (tests: what the parent does with a worker that dies or never returns)
    if fault == "crash" and os.environ.get("RANK") == "1":
        os._exit(3)
    if fault == "hang":
        time.sleep(3600)
    import torch
    import torch.distributed as dist
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        res = multi_gpu_selfcheck(device, rank, world, dist, args.reduce)
    except SystemExit as e:          # a sharded scene that does not reproduce the unsharded one
        res = {"error": str(e)}
    except Exception as e:           # noqa: BLE001
        res = {"error": f"{type(e).__name__}: {e}"[:500]}
    print(json.dumps({"selfcheck": res}), flush=True)
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:                # noqa: BLE001
        pass


def run_selfcheck(args, rank: int, world: int, dist) -> dict:
    """Every rank runs the self-check in a child process (its own rendezvous port, a timeout) and the ranks compare notes.
    -> multi_gpu_selfcheck's dict, or {"error": ...} when any rank's child failed, mismatched or did not finish."""
    box = [_free_port() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    env = dict(os.environ, MASTER_PORT=str(box[0]), MASTER_ADDR="127.0.0.1")
    for k in [k for k in env if k.startswith("TORCHELASTIC_")]:      # (under torchrun: the children rendezvous on their own store, not the agent's)
        del env[k]
    cmd = [sys.executable, os.path.abspath(__file__), "--selfcheck-worker", "--gpus", str(world), "--reduce", args.reduce]
    if args.share_devices:
        cmd.append("--share-devices")
    res = None
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.selfcheck_timeout)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        res = json.loads(lines[-1])["selfcheck"] if lines else {"error": f"self-check worker exited with {r.returncode}: {r.stderr[-300:]}"}
    except subprocess.TimeoutExpired:
        res = {"error": f"self-check worker did not finish within {args.selfcheck_timeout:.0f} s (killed)"}
    except Exception as e:           # noqa: BLE001
        res = {"error": f"{type(e).__name__}: {e}"[:300]}
    allres = [None] * world
    dist.all_gather_object(allres, res)
    bad = [(r_, a["error"]) for r_, a in enumerate(allres) if "error" in a]
    if bad:
        return {"error": f"rank {bad[0][0]}: {bad[0][1]}", "ranks_failed": [b[0] for b in bad]}
    return allres[0]


def bench_mixer(device: int, frames_bank) -> dict:
    """The Mixer leg (mixer.rs is on the north_star's path; BASELINE configs[0] is a Mixer): host-output callbacks, a few of each, and the
    headline-size Mixer through the device-output entry with its roofline; reported only.  `frames_bank`: Frames of the Seek workload's clips."""
    import oddio_amd as oa
    interval = np.float32(1.0) / np.float32(RATE)

    def timed(sig, reps, warm):
        for _ in range(warm):
            sig.sample_n(interval, N_FRAMES)
        t0 = time.perf_counter()
        for _ in range(reps):
            sig.sample_n(interval, N_FRAMES)
        return (time.perf_counter() - t0) / reps * 1e3

    out = {"frames_per_callback": N_FRAMES, "boundary": "oddio_hip_mixer_sample (host slice: stream sync + D2H per callback)"}
    # BASELINE configs[0]: 64 static Sine sources into one Mixer (examples/offline.rs style)
    control, mixer = oa.Mixer(device=device, max_sources=64, max_frames=N_FRAMES)
    for k in range(64):
        control.play(oa.MonoToStereo(oa.Sine(0.1 * k, float(110.0 * 2.0 ** (k / 12.0)))))
    out["config0_64_sines_ms_per_callback"] = timed(mixer, 24, 8)
    mixer.close()
    # a large Mixer<[f32;2]> of MonoToStereo<FramesSignal> sources, FAST and ORDERED (the reference's reverse slot order: bit-exact,
    # tests/test_hip_mixer.py)
    S = min(65536, len(frames_bank) * 16)
    control, mixer = oa.Mixer(device=device, max_sources=S, max_frames=N_FRAMES)
    for i in range(S):
        control.play(oa.MonoToStereo(oa.FramesSignal(frames_bank[(i * 2654435761) % len(frames_bank)], 0.25)))
    fast = timed(mixer, 16, 8)
    mixer.set_mode(oa.MODE_TRACKED)                    # the reference's sequential sum to ~1e-6 (tests/test_hip_mixer_tracked.py)
    tracked = timed(mixer, 8, 3)
    mixer.set_mode(oa.MODE_ORDERED)
    ordered = timed(mixer, 8, 3)
    mixer.close()
    out.update({"sources": S, "fast_ms_per_callback": fast, "tracked_ms_per_callback": tracked, "ordered_ms_per_callback": ordered,
                "fast_source_frames_per_s": float(S) * N_FRAMES / (fast * 1e-3), "ordered_source_frames_per_s": float(S) * N_FRAMES / (ordered * 1e-3)})
    # what each mode promises about a mixer's output at this size: three mixers with the same sources, one callback, against ORDERED
    # (the reference's sequential f32 sum in its reverse slot order, bit-exact: tests/test_hip_mixer.py)
    Sp = min(32768, S)
    outs = {}
    for mode, name in ((oa.MODE_ORDERED, "ordered"), (oa.MODE_FAST, "fast"), (oa.MODE_TRACKED, "tracked")):
        control, mixer = oa.Mixer(device=device, max_sources=Sp, max_frames=N_FRAMES)
        mixer.set_mode(mode)
        for i in range(Sp):
            control.play(oa.MonoToStereo(oa.FramesSignal(frames_bank[(i * 2654435761) % len(frames_bank)], 0.25)))
        outs[name] = mixer.sample_n(interval, N_FRAMES).copy()
        mixer.close()
    scale = float(np.abs(outs["ordered"]).max())
    out["parity"] = {"sources": Sp, "reference": "ORDERED mode (the reference's sum order, bit-exact against the oracle in the tests)",
                     "fast_rel_err_vs_ordered": float(np.abs(outs["fast"] - outs["ordered"]).max()) / scale,
                     "tracked_rel_err_vs_ordered": float(np.abs(outs["tracked"] - outs["ordered"]).max()) / scale, "tolerance": 1e-5}
    # The Mixer at the headline size through its device-output entry (oddio_hip_mixer_sample_device: callbacks enqueued back to back,
    # one synchronisation), with a roofline of its own.  Algorithmic bytes per callback: S * (4 * N + P) + 8 * N -- every source reads
    # N mono f32 samples once (the resample ratio of a Mixer source is 1: no Doppler) plus P = 128 bytes of per-source records, and
    # the stereo mix is written once.
    import torch
    S2 = min(262144, len(frames_bank))
    control, mixer = oa.Mixer(device=device, max_sources=S2, max_frames=N_FRAMES)
    for i in range(S2):
        control.play(oa.MonoToStereo(oa.FramesSignal(frames_bank[i], 0.25)))
    dev_out = torch.zeros((N_FRAMES, 2), dtype=torch.float32, device=torch.device("cuda", device))
    torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
    for _ in range(6):
        mixer.sample_device(interval, dev_out.data_ptr(), N_FRAMES)
    mixer.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        mixer.sample_device(interval, dev_out.data_ptr(), N_FRAMES)
    mixer.synchronize()
    dev_ms = (time.perf_counter() - t0) / reps * 1e3
    assert bool(torch.isfinite(dev_out).all()) and float(dev_out.abs().max()) > 0.0
    host_ms = timed(mixer, 8, 2)                       # the same mixer through the reference's own boundary (host slice)
    mixer.set_mode(oa.MODE_TRACKED)
    for _ in range(3):
        mixer.sample_device(interval, dev_out.data_ptr(), N_FRAMES)
    mixer.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        mixer.sample_device(interval, dev_out.data_ptr(), N_FRAMES)
    mixer.synchronize()
    dev_tracked_ms = (time.perf_counter() - t0) / 8 * 1e3
    assert len(mixer) == S2, len(mixer)
    mixer.close()
    b_alg = float(S2) * (4.0 * N_FRAMES + 128.0) + 8.0 * N_FRAMES
    out["device_output"] = {
        "sources": S2, "boundary": "oddio_hip_mixer_sample_device (frames stay in HBM, callbacks enqueued back to back)",
        "ms_per_callback": dev_ms, "host_output_ms_per_callback": host_ms, "tracked_ms_per_callback": dev_tracked_ms,
        "source_frames_per_s": float(S2) * N_FRAMES / (dev_ms * 1e-3),
        "roofline": {"bound": "hbm", "kernel": "mixer_prepass + mixer_mix + mixer_reduce (the whole callback)", "algorithmic_bytes_per_callback": b_alg,
                     "achieved": b_alg / (dev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": b_alg / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "frac_host_output": b_alg / (host_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}}
    # the same with a GainControl on every source (Gain<MonoToStereo<FramesSignal>>: the mixer's chain path, gain.rs + smooth.rs)
    control, mixer = oa.Mixer(device=device, max_sources=S, max_frames=N_FRAMES)
    gains = []
    for i in range(S):
        gc, gsig = oa.Gain.new(oa.MonoToStereo(oa.FramesSignal(frames_bank[(i * 2654435761) % len(frames_bank)], 0.25)))
        control.play(gsig)
        if i % 64 == 0:
            gains.append(gc)
    for k, gc in enumerate(gains):          # (a ramp is running on a 64th of the sources while the callbacks are timed)
        gc.set_amplitude_ratio(0.5 + 0.001 * (k % 100))
    out["gain_sources_fast_ms_per_callback"] = timed(mixer, 16, 2)
    mixer.set_mode(oa.MODE_ORDERED)
    out["gain_sources_ordered_ms_per_callback"] = timed(mixer, 8, 3)
    mixer.close()
    return out


def bench_buffered(args, device: int, shared=None) -> dict:
    """One SpatialScene whose sources are all played with play_buffered as Gain<Speed<FramesSignal>>: own clip each, speeds in
    [0.9, 1.1], a new gain target for every source before every 4th callback (so that every Gain is always ramping).
    `shared`: the clips (and their Frames) of the Seek workload that ran before, used instead of synthesising 64 GiB again."""
    import torch

    import oddio_amd as oa
    from oddio_amd import synth
    S, L = args.sources, args.clip_len
    dev = torch.device("cuda", device)
    sc = synth.make_scene(args.seed, S)
    st = synth.SplitMixStreams(args.seed ^ 0xB0F, S)
    speeds = st.uniform(0.9, 1.1)
    n_clips = S if args.clips <= 0 else min(args.clips, S)
    control, scene = oa.SpatialScene(device=device, max_sources=S, max_frames=N_FRAMES)   # (max_sources also bounds the buffered set and sizes the control queue)
    scene.reserve_buffered(S)
    if shared is not None:
        clips, frames = shared["clips"], shared["frames"]
        assert len(frames) == n_clips
    else:
        clips = torch.empty((n_clips, L), dtype=torch.float32, device=dev)
        base = clips.data_ptr()
        frames = [oa.Frames.from_device_ptr(RATE, base + 4 * L * i, L, device=device, copy=False) for i in range(n_clips)]
    if n_clips < S:
        pick = ((np.arange(S, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(20)) % np.uint64(n_clips)
        flist = [frames[int(k)] for k in pick]
    else:
        flist = frames
    start_seconds = 0.5
    t_play = time.perf_counter()
    ids = control.play_buffered_frames_batch(flist, np.full(S, start_seconds), [oa.FILTER_SPEED, oa.FILTER_GAIN],
                                             np.stack([speeds, np.ones(S, np.float32)], axis=1), sc["position"], sc["velocity"], sc["radius"],
                                             BUF_MAX_DISTANCE, RATE, BUF_DURATION)
    t_play = time.perf_counter() - t_play
    if shared is None:
        freq = torch.from_numpy(sc["freq_hz"][:n_clips]).to(dev).double()
        n = torch.arange(L, device=dev, dtype=torch.float64)
        chunk = max(1, (1 << 28) // L)
        for s0 in range(0, n_clips, chunk):
            s1 = min(n_clips, s0 + chunk)
            clips[s0:s1] = torch.sin((2.0 * np.pi / RATE) * freq[s0:s1, None] * n[None, :]).float()
    d_ids = torch.from_numpy(ids.astype(np.int64)).to(dev).to(torch.int32)        # uint32 values < 2^31: same bits
    d_pos = torch.from_numpy(sc["position"]).to(dev).contiguous()
    d_vel = torch.from_numpy(sc["velocity"]).to(dev).contiguous()
    gain_sets = [(0.5 + 0.5 * torch.rand(S, device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(args.seed + k))).contiguous()
                 for k in range(4)]
    torch.cuda.synchronize(dev)
    out = torch.zeros((N_FRAMES, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
    interval = np.float32(1.0) / np.float32(RATE)
    # callbacks a clip lasts at the highest speed; every source's FramesSignal clock is put back before that
    span = max(4, int((L - int(start_seconds * RATE)) / (N_FRAMES * 1.1)) - 4)
    step_no = 0

    def before_step():
        nonlocal step_no
        if step_no and step_no % span == 0:
            scene.debug_reset_buffered_clock(start_seconds)
        if step_no and step_no % args.reset_every == 0:
            control.set_motion_device(S, d_ids.data_ptr(), d_pos.data_ptr(), d_vel.data_ptr(), True)
        if step_no % 4 == 0:      # GainControl::set_amplitude_ratio on every source: a 0.1 s ramp (4.7 callbacks) is always running
            control.set_control_device(S, d_ids.data_ptr(), 1, gain_sets[(step_no // 4) % 4].data_ptr())
        step_no += 1

    def one_step():
        before_step()
        scene.sample_device(interval, out.data_ptr(), N_FRAMES)

    def sync_all():
        scene.synchronize()
        torch.cuda.synchronize()

    if args.precondition_ms > 0:
        hold = args.precondition_hold
        for k in range(int(args.precondition_ms / 1.0)):
            one_step()
            if hold > 0 and k % hold == hold - 1:
                control.set_motion_device(S, d_ids.data_ptr(), d_pos.data_ptr(), d_vel.data_ptr(), True)
            if k % 32 == 31:
                scene.synchronize()
        control.set_motion_device(S, d_ids.data_ptr(), d_pos.data_ptr(), d_vel.data_ptr(), True)
        scene.debug_reset_buffered_clock(start_seconds)
        step_no = 0
    for _ in range(args.warmup):
        one_step()
    scene.set_profiling(1)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync_all()
    elapsed = time.perf_counter() - t0
    stages = scene.buffered_ms_history(min(args.steps, 512))
    assert len(stages) == min(args.steps, 512), (len(stages), args.steps)
    scene.set_profiling(False)
    n_slow = scene.debug_buffered_slow()
    assert scene.len_buffered() == S, "sources finished inside the timed region"
    assert bool(torch.isfinite(out).all()) and float(out.abs().max()) > 0.0
    # host-output calls (the reference's boundary hands a host slice), untimed by `value`
    host_buf = np.zeros((N_FRAMES, 2), dtype=np.float32)
    for _ in range(2):
        before_step(); scene.sample(interval, host_buf)
    th0 = time.perf_counter()
    for _ in range(6):
        before_step(); scene.sample(interval, host_buf)
    host_ms = (time.perf_counter() - th0) / 6 * 1e3
    # the general kernel on the same scene (what every buffered source took before round 4), a few callbacks
    scene.set_buffered_fast(False)
    for _ in range(2):
        one_step()
    scene.synchronize()
    tg0 = time.perf_counter()
    for _ in range(4):
        one_step()
    scene.synchronize()
    general_ms = (time.perf_counter() - tg0) / 4 * 1e3
    scene.set_buffered_fast(True)
    # ORDERED mode (the reference's sum order: contribution rows + ordered_sum), the figure that conforms to the 1e-5 tolerance at
    # this source count (tests/test_hip_buffered_fast.py: bit-exact at 65 536 buffered sources)
    scene.set_mode(oa.MODE_ORDERED)
    for _ in range(3):
        one_step()
    scene.synchronize()
    to0 = time.perf_counter()
    for _ in range(8):
        one_step()
    scene.synchronize()
    ordered_ms = (time.perf_counter() - to0) / 8 * 1e3
    # TRACKED: two passes of the ring reads whose second one restarts every wave's sums at the prefix of the first (~1e-6 of the
    # reference's sum: tests/test_hip_buffered_fast.py)
    scene.set_mode(oa.MODE_TRACKED)
    for _ in range(3):
        one_step()
    scene.synchronize()
    tt0 = time.perf_counter()
    for _ in range(8):
        one_step()
    scene.synchronize()
    tracked_ms = (time.perf_counter() - tt0) / 8 * 1e3
    scene.set_mode(oa.MODE_FAST)
    assert scene.len_buffered() == S, "sources finished inside the ORDERED leg"

    b = buffered_algorithmic_bytes(speeds, N_FRAMES)
    walk_ms, write_ms, read_ms = (float(stages[:, k].mean()) for k in range(3))
    path_ms = walk_ms + write_ms + read_ms
    ms_per_step = elapsed / args.steps * 1e3
    value = float(S) * N_FRAMES * args.steps / elapsed
    gbps = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9     # noqa: E731
    line = {
        "metric": "mixed source-frames/sec (48 kHz stereo)",
        "value": value, "unit": "source-frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"buffered path (north_star: speed.rs, gain.rs, smooth.rs, ring.rs): SpatialScene, {S} moving Gain<Speed<FramesSignal>> sources played "
                        f"with play_buffered (own {L}-sample clip each, speeds uniform in [0.9, 1.1], a GainControl store to every source before every 4th callback), "
                        f"rings of {int(np.ceil((BUF_MAX_DISTANCE / 343.0 + BUF_DURATION) * RATE)) + 1} samples, Doppler + propagation delay, 48 kHz stereo, {N_FRAMES}-frame callbacks",
            "sources_per_gpu": S, "frames_per_callback": N_FRAMES, "sample_rate": RATE, "clip_len": L,
            "sum_mode": "FAST: deterministic tree sum over waves and workgroups; ring contents are the reference's bits in every mode",
            "parallelism": "single-gpu", "play_seconds": t_play, "sources_on_general_kernel": int(n_slow),
        },
        "value_conforming": float(S) * N_FRAMES / (min(tracked_ms, ordered_ms) * 1e-3),      # the faster of TRACKED (~1e-6 of the reference's sum) and ORDERED (its bits)
        "value_conforming_mode": "TRACKED" if tracked_ms < ordered_ms else "ORDERED",
        "value_bit_exact": float(S) * N_FRAMES / (ordered_ms * 1e-3),
        "tracked_mode_ms_per_step": tracked_ms,
        "ordered_mode_ms_per_step": ordered_ms,
        "max_realtime_sources": value / RATE,
        "host_output_ms_per_step": host_ms,
        "general_kernel_ms_per_step": general_ms,
        "precondition_ms": args.precondition_ms,
        "roofline": {
            "bound": "hbm", "achieved": gbps(b["total"], path_ms), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps(b["total"], path_ms) / HBM_PEAK_GBPS,
            "traffic": buffered_traffic(S), "traffic_source": "profiles/pmc_buffered_latest.json (rocprofv3 --pmc passes of the three kernels, not measured in this run)",
            "kernel": "buffered_walk + buffered_write + spatial_mix<RING> (+ its reduce): the buffered set's whole path",
            "avg_kernel_ms": path_ms, "algorithmic_bytes_per_launch": b["total"], "bytes": b,
            "frac_callback": gbps(b["total"], ms_per_step) / HBM_PEAK_GBPS,
            "frac_host_output": gbps(b["total"], host_ms) / HBM_PEAK_GBPS,
            "walk_ms": walk_ms,
            "write_ms": write_ms, "write_frac": gbps(b["write_kernel"], write_ms) / HBM_PEAK_GBPS,
            "read_ms": read_ms, "read_frac": gbps(b["read_kernel"], read_ms) / HBM_PEAK_GBPS,
            "stage_timing": f"hipEvents around the three stages of every timed callback ({len(stages)} samples)",
        },
    }
    if not args.no_cpu_baseline:
        cpu, parity = buffered_cpu_and_parity(device, args.seed, min(args.cpu_budget, 10.0))
        cpu["gpu_over_cpu"] = value / cpu["value"]
        line["cpu_baseline"], line["parity"] = cpu, parity
    scene.close()
    return line


# ---------------------------------------------------------------------------------------------------
def _smi_clocks() -> str | None:
    """One `rocm-smi` reading (shader / memory clocks, package power, performance level); None when the tool is not there."""
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showperflevel"], capture_output=True, text=True, timeout=20)
        keep = [" ".join(ln.split()) for ln in r.stdout.splitlines() if any(k in ln for k in ("sclk", "mclk", "Power", "Performance Level"))]
        return " | ".join(keep) if keep else None
    except Exception:      # noqa: BLE001
        return None


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: spawn the N ranks ourselves (one per GPU)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: few enough callbacks that the constant-velocity sources are still the workload BASELINE defines
    # (positions in the +-50 m cube, radial velocities of both signs) -- after seconds of simulated time every
    # source is receding, windows shrink and the same kernel looks ~10 % faster
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--sources", type=int, default=262144, help="sources per GPU (config 3: 262144; config 2: 4096)")
    ap.add_argument("--clip-len", type=int, default=65536)
    ap.add_argument("--clips", type=int, default=0,
                    help="distinct clips (default: one per source, the BASELINE workload); fewer lets --sources exceed what 288 GB of "
                         "own clips allows, e.g. to run AT the reported max_realtime_sources")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--mode", choices=["scenes", "sharded"], default="scenes",
                    help="N>1: 'scenes' = one independent scene per GPU (configs[3] pattern, no collective); "
                         "'sharded' = ONE scene of N*sources split into contiguous index shards with an RCCL all-reduce "
                         "of the 8 KiB stereo buffer per callback (configs[4] pattern)")
    ap.add_argument("--reduce", choices=["rccl", "p2p"], default="rccl",
                    help="--mode sharded: the cross-rank sum of the stereo buffer -- RCCL all-reduce, or the library's deterministic "
                         "peer-to-peer reduce (rank-ordered sum on rank 0; works with several ranks on one GPU)")
    ap.add_argument("--selfcheck-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--selfcheck-timeout", type=float, default=300.0, help="seconds the multi-GPU self-check's child processes get before they are killed")
    ap.add_argument("--no-selfcheck", action="store_true", help="--gpus N > 1: skip the sharded-vs-unsharded scene check that precedes the timed run")
    ap.add_argument("--share-devices", action="store_true", help="smoke-testing on a box with fewer GPUs than ranks: rank r uses device r %% count")
    ap.add_argument("--reset-every", type=int, default=320, help="callbacks between host-side motion resets of all sources")
    ap.add_argument("--precondition-hold", type=int, default=32,
                    help="reset every source to its start every n pre-conditioning callbacks, so that they run BASELINE's workload too "
                         "(0: let the sources drift until the reset before the warm-up; measured 0-3 %% slower in the timed region, by box)")
    ap.add_argument("--event-stride", type=int, default=1,
                    help="bracket spatial_mix with hipEvents on every n-th timed callback (default: every one)")
    ap.add_argument("--precondition-ms", type=float, default=200.0,
                    help="GPU clock pre-conditioning before the warm-up steps: this many ms of untimed callbacks of the workload "
                         "itself, after which every source is put back to its starting state, so that a short warm-up starts from "
                         "loaded clocks instead of the idle state the host-side set-up leaves behind (DESIGN.md section 5); 0 disables it")
    ap.add_argument("--workload", choices=["seek", "buffered"], default="seek",
                    help="'seek' (default): BASELINE's FramesSignal sources played with play(); 'buffered': the same number of "
                         "Gain<Speed<FramesSignal>> sources played with play_buffered (single GPU)")
    ap.add_argument("--sustained", type=int, default=4096,
                    help="callbacks of the nested `sustained` leg (single GPU): the timed region's workload held for this many callbacks "
                         "(~0.25 ms each), with rocm-smi's clocks and power read while it runs; 0 skips it")
    ap.add_argument("--no-buffered", action="store_true", help="skip the nested buffered_path and mixer_path objects of the default (seek) workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    if args.cpu_worker:
        cpu_worker(args.cpu_worker)
        return
    if args.selfcheck_worker:
        selfcheck_worker(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    n_dev = torch.cuda.device_count()
    if world > n_dev and not args.share_devices:
        raise SystemExit(f"--gpus {world} but only {n_dev} device(s) visible (use --share-devices for a smoke run)")
    device = local_rank % n_dev
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        # the process group only carries the timing barrier / max and the reduce group's id: CPU tensors
        # over gloo.  The data path's one collective (sharded mode) is the library's own RCCL all-reduce.
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist_mod.init_process_group(backend="gloo", rank=rank, world_size=world)
        dist = dist_mod

    if args.workload == "buffered":
        if world != 1:
            raise SystemExit("--workload buffered is a single-GPU line")
        print(json.dumps(bench_buffered(args, device)), flush=True)
        return
    S, L = args.sources, args.clip_len
    selfcheck = None
    sharded = args.mode == "sharded" and world > 1
    if world > 1 and not args.no_selfcheck:
        # in a child process per rank, with a timeout: a collective that hangs or crashes costs the self-check, not the line
        selfcheck = run_selfcheck(args, rank, world, dist)
        if "error" in selfcheck:
            if rank == 0:
                print(f"[bench] multi-GPU self-check FAILED: {selfcheck['error']}", file=sys.stderr)
            if sharded:      # the sharded scene's own data path is what failed: no number can be trusted
                raise SystemExit(f"multi-GPU self-check failed: {selfcheck['error']}")
            # (--mode scenes has no data-path collective: the independent scenes are timed anyway, the failure is in the line)
        elif selfcheck["reduce"] != args.reduce:
            if rank == 0:
                print(f"[bench] RCCL reduce group unavailable ({selfcheck['rccl_error']}); the sharded scene uses the peer-to-peer reduce", file=sys.stderr)
            args.reduce = selfcheck["reduce"]
    # clips start 1.0 s in: the propagation delay (<= 0.25 s at set-up) may grow by the drift of the
    # constant-velocity sources (<= 34.6 m/s) for `reset_every` callbacks without reading before the clip
    start_seconds = 1.0
    if sharded:
        # ONE seeded scene of world * S sources; this rank owns the contiguous index shard [lo, hi)
        from oddio_amd import sharding
        uid = sharding.exchange_unique_id(dist) if args.reduce == "rccl" else None
        lo, hi = sharding.shard_range(world * S, world, rank)
        holder = {}

        def factory():
            holder["sh"] = sharding.ShardedSpatialScene(device, world * S, N_FRAMES, rank, world, uid, reduce=args.reduce, dist=dist)
            return holder["sh"].control, holder["sh"].scene
        g = build_gpu_scene(device, hi - lo, L, args.seed, start_seconds, args.clips, first_index=lo, scene_factory=factory)
    else:
        g = build_gpu_scene(device, S, L, args.seed + rank, start_seconds, args.clips)
    scene, control = g["scene"], g["control"]
    out = torch.zeros((N_FRAMES, 2), dtype=torch.float32, device=torch.device("cuda", device))
    torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
    interval = np.float32(1.0) / np.float32(RATE)
    # callbacks a clip lasts before sources would run off its end; rewind before that
    span = max(1, (L - int(start_seconds * RATE)) // N_FRAMES - 8)
    rewind_seconds = -float(span * N_FRAMES) / RATE
    reset_every = args.reset_every   # callbacks; bounds the drift of the constant-velocity sources (a host-side batch set_motion)

    step_no = 0

    def one_step():
        nonlocal step_no
        if step_no and step_no % span == 0:
            scene.seek_all(rewind_seconds)              # Seek::seek on every source (a tiny kernel, timed)
        if step_no and step_no % reset_every == 0:
            control.set_motion_batch(g["ids"], g["spec"]["position"], g["spec"]["velocity"], True)
        scene.sample_device(interval, out.data_ptr(), N_FRAMES)   # sharded: includes the RCCL all-reduce
        step_no += 1

    def sync_all():
        scene.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        scene.synchronize()
        torch.cuda.synchronize()

    if args.precondition_ms > 0:
        # DVFS: after seconds of host-side set-up the chip sits in a low clock state and needs ~30 ms of load to leave it;
        # the mix kernel is latency-bound enough to feel that (tools/ramp_probe.py).  Untimed callbacks of the workload
        # itself, then every source is put back where BASELINE's workload starts: clip cursors rewound, Motion reset.
        # (a fixed number of callbacks, ~0.3 ms each: the ranks of a sharded scene must all make the same calls)
        hold = args.precondition_hold
        for k in range(int(args.precondition_ms / 0.3)):
            one_step()
            if hold > 0 and k % hold == hold - 1:
                # keep the pre-conditioning callbacks on BASELINE's workload too (left alone the sources fly apart and the
                # windows shrink): every source back to its start, in stream order
                control.set_motion_batch(g["ids"], g["spec"]["position"], g["spec"]["velocity"], True)
                scene.seek_all(-float((step_no % span) * N_FRAMES) / RATE)
                scene.sample_device(interval, out.data_ptr(), 0)
                step_no = 0
            if k % 32 == 31:
                scene.synchronize()
        # the reset is host work (a batch set_motion of every source): ~25 ms of callbacks are queued first, so that the
        # GPU stays loaded while the host prepares it, and the reset itself is applied in stream order right behind them
        for _ in range(96):
            one_step()
        control.set_motion_batch(g["ids"], g["spec"]["position"], g["spec"]["velocity"], True)
        scene.seek_all(-float((step_no % span) * N_FRAMES) / RATE)
        scene.sample_device(interval, out.data_ptr(), 0)      # a zero-frame callback: the Motion updates are applied, no time passes
        step_no = 0
    for _ in range(args.warmup):
        one_step()
    # hipEvents around spatial_mix (the roofline kernel) inside the timed region, on every callback.  (A pair of events puts
    # ~12 us of gaps around the kernel it brackets -- kernel trace of round 3: 7.5 + 5.9 us against 0.9 us between
    # un-bracketed kernels -- but bracketing only every 4th callback did not pay: back to back the kernel itself runs
    # 1.5 % slower, step 0.2594-0.2614 vs 0.2613-0.2624 ms, mix 0.2339-0.2356 vs 0.2311-0.2319, same box; --event-stride 4.
    # Events attached to the kernel's own dispatch, hipExtLaunchKernelGGL, leave the same gaps: 7.1 + 4.7 us.)
    stride = max(1, args.event_stride)
    n_samples = (args.steps + stride - 1) // stride
    scene.set_profiling(1 + stride if stride >= 2 else 2)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync_all()
    elapsed = time.perf_counter() - t0
    hist = scene.kernel_ms_history(min(n_samples, 512))
    assert len(hist) == min(n_samples, 512), (len(hist), n_samples)
    # ---- `sustained`: the same workload for 4 096 callbacks (~1 s of GPU time: 200 x the timed region), clocks read while it runs ----
    sustained = None
    if world == 1 and args.sustained > 0:
        scene.set_profiling(False)
        dev_t = torch.device("cuda", device)
        d_ids = torch.from_numpy(np.ascontiguousarray(g["ids"], dtype=np.uint32).view(np.int32)).to(dev_t)
        d_pos = torch.from_numpy(np.ascontiguousarray(g["spec"]["position"], dtype=np.float32)).to(dev_t)
        d_vel = torch.from_numpy(np.ascontiguousarray(g["spec"]["velocity"], dtype=np.float32)).to(dev_t)
        hold = 32                      # every source back to BASELINE's start every 32 callbacks, by ONE message (device arrays): the host never falls behind
        smi_before = _smi_clocks()
        # one reading while the callbacks run: a thread that waits a third of the expected duration (the launch calls release the GIL)
        import threading
        during = {}
        expect_s = args.sustained * (elapsed / args.steps)

        def _read_during():
            time.sleep(0.3 * expect_s)
            during["t0"] = time.perf_counter()
            during["smi"] = _smi_clocks()
            during["t1"] = time.perf_counter()
        th = threading.Thread(target=_read_during)
        sync_all()
        ts0 = time.perf_counter()
        th.start()
        for k in range(args.sustained):
            if step_no and step_no % span == 0:
                scene.seek_all(rewind_seconds)              # (the clips last `span` callbacks: as in one_step)
            scene.sample_device(interval, out.data_ptr(), N_FRAMES)
            step_no += 1
            if k % hold == hold - 1:
                control.set_motion_device(len(g["ids"]), d_ids.data_ptr(), d_pos.data_ptr(), d_vel.data_ptr(), True)
        t_enq = time.perf_counter() - ts0
        sync_all()
        t_sus = time.perf_counter() - ts0
        th.join()
        # (kept only if the reading began and ended inside the run)
        smi_during = during.get("smi") if during.get("t1", 1e30) - ts0 <= t_sus else None
        sustained = {"callbacks": args.sustained, "ms_per_step": t_sus / args.sustained * 1e3, "value": float(S) * N_FRAMES * args.sustained / t_sus,
                     "frac_callback": algorithmic_bytes(len(g["ids"]), N_FRAMES) / (t_sus / args.sustained) / 1e9 / HBM_PEAK_GBPS,
                     "host_enqueue_s": t_enq, "gpu_s": t_sus, "rocm_smi_during_window_s": ([during["t0"] - ts0, during["t1"] - ts0] if "t1" in during else None), "rocm_smi_before": smi_before, "rocm_smi_during": smi_during,
                     "workload": f"the timed region's, every source's Motion put back to its start every {hold} callbacks (oddio_hip_scene_set_motion_device), clips rewound every {span}"}
        scene.set_profiling(2)
    # the other stages of a callback (walk, reduce): a few untimed callbacks with events around every stage
    # (four per callback cost 3-6 us of command-processor time, tools/event_overhead.py; not inside `value`)
    scene.set_profiling(1)
    for _ in range(8):
        one_step()
    stages = scene.kernel_ms_history(8)
    # the same kernel with the reference's arithmetic in every contribution (MODE_FAST_UNFUSED: no fused multiply-adds), untimed
    import oddio_amd as _oa
    # (on the workload of the timed region: every source put back where it started, the same warm-up, the same number of callbacks)
    scene.set_mode(_oa.MODE_FAST_UNFUSED)
    for _ in range(96):                    # (the reset is host work: keep the GPU loaded meanwhile, as before the timed region)
        one_step()
    control.set_motion_batch(g["ids"], g["spec"]["position"], g["spec"]["velocity"], True)
    scene.seek_all(-float((step_no % span) * N_FRAMES) / RATE)
    scene.sample_device(interval, out.data_ptr(), 0)
    step_no = 0
    for _ in range(args.warmup):
        one_step()
    scene.set_profiling(2)
    for _ in range(args.steps):
        one_step()
    unfused_ms = float(scene.kernel_ms_history(min(args.steps, 512))[:, 1].mean())
    scene.set_mode(_oa.MODE_FAST)
    scene.set_profiling(False)
    assert len(scene) == len(g["ids"]), "sources finished inside the timed region"
    assert bool(torch.isfinite(out).all()) and float(out.abs().max()) > 0.0

    # the reference's boundary hands a host slice (oddio::run): same callbacks through oddio_hip_scene_sample
    # (8 KiB D2H + a stream sync per callback), untimed by `value`, reported next to it
    host_buf = np.zeros((N_FRAMES, 2), dtype=np.float32)
    # (few callbacks: starting from an idle GPU every time, the same kernels run ~10 % slower than back to back, and
    # these launches are in the rocprofv3 statistics of this command next to the timed ones)
    n_host = 6
    for _ in range(2):                                   # untimed: first host-output calls touch the pinned buffer
        if step_no % span == 0:
            scene.seek_all(rewind_seconds)
        scene.sample(interval, host_buf)
        step_no += 1
    th0 = time.perf_counter()
    for _ in range(n_host):
        if step_no % span == 0:
            scene.seek_all(rewind_seconds)
        scene.sample(interval, host_buf)
        step_no += 1
    host_ms = (time.perf_counter() - th0) / n_host * 1e3

    # the bit-exact mode (reference's sum order, tests/test_hip_large_scene.py) on the same scene: a few callbacks, reported only
    ordered_ms = None
    ordered_latency_ms = None
    tracked_ms = None
    if world == 1:
        import oddio_amd as oa
        scene.set_mode(oa.MODE_ORDERED)
        n_warm, n_timed = 3, 32                # (the first launches after the mode switch run slower: the sum's first 1.06 ms, then 0.75;
                                               #  32 callbacks: the pipeline of front and sum fills once)
        for k in range(n_warm + n_timed):
            if step_no % span == 0:
                scene.seek_all(rewind_seconds)
            if k == n_warm:
                scene.synchronize()
                to0 = time.perf_counter()
            scene.sample_device(interval, out.data_ptr(), N_FRAMES)
            step_no += 1
        scene.synchronize()
        ordered_ms = (time.perf_counter() - to0) / n_timed * 1e3
        # one callback at a time (each waited for): what an interactive caller sees; enqueued back to back, above, the
        # front of callback k + 1 overlaps the sum of callback k
        tl0 = time.perf_counter()
        for _ in range(4):
            if step_no % span == 0:
                scene.seek_all(rewind_seconds)
            scene.sample_device(interval, out.data_ptr(), N_FRAMES)
            scene.synchronize()
            step_no += 1
        ordered_latency_ms = (time.perf_counter() - tl0) / 4 * 1e3
        # ODDIO_HIP_MODE_TRACKED: the reference's sequential sum to ~1e-6 (parity.tracked_rel_err_vs_reference,
        # tests/test_hip_large_scene.py) by two passes of the FAST-mode kernel
        scene.set_mode(oa.MODE_TRACKED)
        for k in range(3 + 16):
            if step_no % span == 0:
                scene.seek_all(rewind_seconds)
            if k == 3:
                scene.synchronize()
                tt0 = time.perf_counter()
            scene.sample_device(interval, out.data_ptr(), N_FRAMES)
            step_no += 1
        scene.synchronize()
        tracked_ms = (time.perf_counter() - tt0) / 16 * 1e3
        scene.set_mode(oa.MODE_FAST)

    tracked_by_rank = None
    if world > 1:
        # The conforming figure of a multi-GPU run: ODDIO_HIP_MODE_TRACKED on every rank (scenes: each scene tracks its own sum;
        # sharded: the ranks exchange their first pass's totals between the passes), the same protocol as the timed region --
        # barrier + synchronize on both sides, the slowest rank counts.  (ORDERED is not timed here: a sharded scene's rank-ordered
        # sum of per-shard ORDERED sums is not the reference's order across the shards.)
        import oddio_amd as oa
        # (a sharded scene's TRACKED callbacks issue a collective more -- ncclAllGather / the slab's second block: timed only when the
        # self-check has just run exactly that; independent scenes track on their own)
        tracked_verified = (not sharded) or args.no_selfcheck or bool(selfcheck and selfcheck.get("tracked_max_rel_err_vs_ordered") is not None)
        if tracked_verified:
            scene.set_mode(oa.MODE_TRACKED)
            n_warm, n_timed = 3, 16
            for _ in range(n_warm):
                one_step()
            sync_all()
            tt0 = time.perf_counter()
            for _ in range(n_timed):
                one_step()
            sync_all()
            t_tr = torch.tensor([time.perf_counter() - tt0], dtype=torch.float64)
            tracked_by_rank = [None] * world
            dist.all_gather_object(tracked_by_rank, float(t_tr.item()) / n_timed * 1e3)
            dist.all_reduce(t_tr, op=dist.ReduceOp.MAX)
            tracked_ms = float(t_tr.item()) / n_timed * 1e3
            scene.set_mode(oa.MODE_FAST)

    ranks_seen = 1
    per_rank_ms = [elapsed / args.steps * 1e3]
    per_rank_mix_ms = [float(hist[:, 1].mean())]
    reduce_info = scene.reduce_info() if hasattr(scene, "reduce_info") else {"kind": "none", "world": 1, "rccl_version": 0, "rccl_lib": None}
    if dist is not None:
        # every rank's own time (the line's ms_per_step is their maximum) and what every rank's library says about its reduce group
        gathered = [None] * world
        dist.all_gather_object(gathered, {"ms": elapsed / args.steps * 1e3, "reduce": reduce_info, "mix_ms": float(hist[:, 1].mean())})
        per_rank_ms = [g_["ms"] for g_ in gathered]
        per_rank_mix_ms = [g_["mix_ms"] for g_ in gathered]
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # ranks_seen: from the data path's own communicator when there is one (ncclCommCount / the P2P slab's rank count, on every
        # rank), else the number of ranks that reported a time
        worlds = {g_["reduce"]["world"] for g_ in gathered if g_["reduce"]["kind"] != "none"}
        ranks_seen = worlds.pop() if len(worlds) == 1 else len(gathered)

    if rank == 0:
        total_units = float(S) * N_FRAMES * args.steps * world
        value = total_units / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        mix_ms = float(hist[:, 1].mean())
        b_alg = algorithmic_bytes(len(g["ids"]), N_FRAMES)
        achieved = b_alg / (mix_ms * 1e-3) / 1e9
        traffic, traffic_source, ordered_traffic, traffic_stale = None, None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("sources") == S and j.get("kernel", "").startswith("spatial_mix"):
                    traffic = j.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/pmc_latest.json (rocprofv3 --pmc passes of this kernel, not measured in this run)"
                    ordered_traffic = j.get("ordered_hbm_bytes_per_callback")
                    # the counters were collected on the kernels whose source hash the file carries (tools/make_pmc_json.py)
                    from oddio_amd import _lib as _l
                    traffic_stale = j.get("mix_kernel_source_hash") != _l.mix_kernel_source_hash()
            except Exception:
                traffic = None
        if world == 1:
            shape = (f"BASELINE configs[2]: SpatialScene, {S} moving FramesSignal sources (own {L}-sample clip each), " if args.clips <= 0 or args.clips >= S
                     else f"real-time confirmation run: SpatialScene, {S} moving FramesSignal sources sharing {args.clips} clips of {L} samples, ")
        elif sharded:
            red = "RCCL all-reduce" if args.reduce == "rccl" else "rank-ordered peer-to-peer reduce"
            shape = f"BASELINE configs[4] pattern: ONE SpatialScene of {world * S} moving FramesSignal sources in {world} contiguous index shards + {red} of the stereo buffer, "
        else:
            shape = f"BASELINE configs[3] pattern: {world} independent SpatialScenes, one per GPU, {S} moving FramesSignal sources each (per-GPU work as at N=1), "
        line = {
            "metric": "mixed source-frames/sec (48 kHz stereo)",
            "value": value,
            "unit": "source-frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": shape + f"Doppler + propagation delay, 48 kHz stereo, {N_FRAMES}-frame callbacks",
                "sources_per_gpu": S, "frames_per_callback": N_FRAMES, "sample_rate": RATE, "clip_len": L,
                "sum_mode": "FAST: deterministic tree sum over waves and workgroups, fused multiply-adds in the lerp / gain ramp / accumulate "
                            "(within 1e-5 of the reference up to 4096 sources; `parity` has the figures at this size); TRACKED (~1e-6 of the reference) and the bit-exact ORDERED mode are "
                            "timed in tracked_mode_ms_per_step, ordered_mode_ms_per_step / ordered_mode_latency_ms",
                "parallelism": ("single-gpu" if world == 1 else ((f"source-sharded scene + stereo-buffer reduce ({args.reduce})") if sharded else "scene-parallel")),
                "ranks_seen": ranks_seen,
                "ms_per_step_by_rank": per_rank_ms,
                "tracked_ms_per_step_by_rank": tracked_by_rank,      # (N > 1) the conforming mode, every rank's own time
                # every rank's own roofline fraction (its mix kernel's average over the timed callbacks against the 8 TB/s peak)
                "roofline_frac_by_rank": [algorithmic_bytes(len(g["ids"]), N_FRAMES) / (m_ * 1e-3) / 1e9 / HBM_PEAK_GBPS for m_ in per_rank_mix_ms],
                "multi_gpu_selfcheck": selfcheck,        # (N > 1) the sharded-vs-unsharded scene check that ran before the timed region, and the reduce it found working
                "reduce_group": reduce_info,            # rank 0's: kind, world (ncclCommCount / slab), librccl path and version
            },
            # the figure that conforms to the north_star tolerance at this source count (`parity` below: FAST is outside 1e-5 of the
            # reference at this size because the reference's own sequential f32 sum is that far from the exact one): the faster of
            # TRACKED (the reference's sum with its rounding errors, ~1e-6: parity.tracked_rel_err_vs_reference) and ORDERED (the
            # reference's sum order, bit-exact: value_bit_exact), callbacks enqueued back to back
            # (N > 1: TRACKED on every rank, whole-job aggregate like `value`, the slowest rank's time)
            "value_conforming": ((float(S) * N_FRAMES / (min(tracked_ms or ordered_ms, ordered_ms) * 1e-3)) if ordered_ms else
                                 ((float(S) * N_FRAMES * world / (tracked_ms * 1e-3)) if tracked_ms else None)),
            "value_conforming_mode": (("TRACKED" if (tracked_ms and tracked_ms < ordered_ms) else "ORDERED") if ordered_ms else ("TRACKED" if tracked_ms else None)),
            "value_bit_exact": (float(S) * N_FRAMES / (ordered_ms * 1e-3)) if ordered_ms else None,    # ORDERED: the reference's bits
            "tracked_mode_ms_per_step": tracked_ms,             # the reference's sum to ~1e-6 (two passes of the FAST-mode kernel), callbacks back to back
            "max_realtime_sources": value / RATE,
            "realtime_factor_per_gpu": (N_FRAMES / RATE) / (elapsed / args.steps),
            "host_output_ms_per_step": host_ms,
            "ordered_mode_ms_per_step": ordered_ms,             # bit-exact (reference sum order) mode, same scene: callbacks enqueued back to back
            "ordered_mode_latency_ms": ordered_latency_ms,      # ... and one callback at a time
            "sustained": sustained,                             # the same workload held for thousands of callbacks (see --sustained)
            "precondition_ms": args.precondition_ms,
            "precondition_hold": args.precondition_hold,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_source": traffic_source,
                "traffic_stale": traffic_stale,      # true: the mix kernels' sources have changed since the PMC passes (re-run tools/profile_pmc.sh)
                # (callbacks of 513..1024 frames over >= 32 768 sources run spatial_mix_pair -- csrc/pair_kernels.h, ODDIO_HIP_PAIR=0 keeps the
                # 512-frame-tile kernel spatial_mix; smaller scenes always run spatial_mix)
                "kernel": ("spatial_mix_pair" if (os.environ.get("ODDIO_HIP_PAIR", "1") != "0" and len(g["ids"]) >= 32768) else "spatial_mix"),
                "avg_kernel_ms": mix_ms, "algorithmic_bytes_per_launch": b_alg,
                "frac_callback": b_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                # through the reference's own boundary -- a host slice per call (oddio::run, src/lib.rs:90-93): + an 8 KiB D2H copy and a
                # stream synchronisation per callback, the GPU idle while the host turns around
                "frac_host_output": b_alg / (host_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "prepass_ms": float(stages[:, 0].mean()),
                ("reduce_incl_collective_ms" if sharded else "reduce_ms"): float(stages[:, 2].mean()),
                "stage_timing": f"prepass/reduce: 8 untimed callbacks after the timed region; avg_kernel_ms: hipEvents around spatial_mix on every "
                                f"{stride}-th timed callback ({len(hist)} samples)",
                "kernel_samples": int(len(hist)), "event_stride": int(stride),
                "unfused_kernel_ms": unfused_ms,              # spatial_mix in MODE_FAST_UNFUSED, the timed region repeated after it (untimed)
                "unfused_frac": b_alg / (unfused_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                # ORDERED mode (bit-exact): the same algorithmic bytes over its whole callback (render of the contribution rows + the
                # sequential sum); ordered_traffic: what it really moves (every row crosses HBM twice), from the PMC passes
                "ordered_ms_per_step": ordered_ms,
                "ordered_frac": (b_alg / (ordered_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if ordered_ms else None,
                # TRACKED (the conforming mode at this size): the same algorithmic bytes over its two-pass callback; it moves them twice
                "tracked_ms_per_step": tracked_ms,
                "tracked_frac": (b_alg / (tracked_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if tracked_ms else None,
                "ordered_traffic": ordered_traffic,
            },
        }
        if world == 1 and not args.no_buffered:
            # the path Gain / Speed sources take into a scene (play_buffered), same source count, same clips: its own line, nested
            line["buffered_path"] = bench_buffered(args, device, shared={"clips": g["clips"], "frames": g["frames"]})
        if world == 1 and not args.no_buffered:
            line["mixer_path"] = bench_mixer(device, g["frames"])
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.seed, args.cpu_budget, parity_device=device, parity_sources=S)
            line["parity"] = line["cpu_baseline"].pop("parity")
            line["cpu_baseline"]["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
            if line["cpu_baseline"]["all_cores"]["valid"]:
                # (named after the number of cores the host really gave this job: see cpu_baseline.all_cores.cores / .sample)
                line["cpu_baseline"][f"gpu_over_cpu_usable_cores_{line['cpu_baseline']['all_cores']['cores']}"] = value / line["cpu_baseline"]["all_cores"]["value"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
