#!/usr/bin/env python
"""Generates the committed golden fixtures (inputs + expected outputs) for the hot path.

The reference is Rust and cannot run here, so the expected outputs come from the C restatement
(oracle/oddio_oracle.c) AFTER it has been checked, bit for bit, against the independent numpy
restatement (oracle/oracle_np.py) on the very same scenario -- the script refuses to write a
fixture on which the two disagree.  A fixture is data only: scene description arrays, the event
schedule and the rendered stereo buffers.

    python tests/golden/gen_golden.py [name ...]     # rewrites tests/golden/*.npz (or only the named fixtures)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import scenario  # noqa: E402
from oddio_amd import synth  # noqa: E402
from oracle import oracle_np as on  # noqa: E402

KIND_ID = {"frames": 0, "sine": 1, "constant": 2, "cycle": 3, "downmix": 4}     # (a downmix clip is stored interleaved: [l0, r0, l1, r1, ...])
ONLY = set(sys.argv[1:])


def pack(spec, events, n_frames, n_callbacks, interval, postfx, outputs):
    srcs = spec["sources"]
    n = len(srcs)
    clips = [np.asarray(s.get("clip", np.zeros(0, np.float32)), np.float32).reshape(-1) for s in srcs]
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(c) for c in clips])
    ev_motion = [(cb, e[1], *e[2], *e[3], int(e[4])) for cb, evs in events.items() for e in evs if e[0] == "motion"]
    ev_rot = [(cb, *e[1]) for cb, evs in events.items() for e in evs if e[0] == "rotation"]
    return dict(
        kind=np.array([KIND_ID[s["kind"]] for s in srcs], np.int32),
        pos=np.stack([s["pos"] for s in srcs]).astype(np.float32),
        vel=np.stack([s["vel"] for s in srcs]).astype(np.float32),
        radius=np.array([s["radius"] for s in srcs], np.float32),
        gain_db=np.array([np.nan if s.get("gain_db") is None else s["gain_db"] for s in srcs], np.float32),
        rate=np.array([s.get("rate", 0) for s in srcs], np.int64),
        start=np.array([s.get("start", 0.0) for s in srcs], np.float64),
        phase=np.array([s.get("phase", 0.0) for s in srcs], np.float32),
        hz=np.array([s.get("hz", 0.0) for s in srcs], np.float32),
        value=np.array([s.get("value", 0.0) for s in srcs], np.float32),
        clip_data=np.concatenate(clips).astype(np.float32) if offs[-1] else np.zeros(0, np.float32),
        clip_offsets=offs,
        ev_motion=np.array(ev_motion, np.float64).reshape(-1, 9),
        ev_rotation=np.array(ev_rot, np.float64).reshape(-1, 5),
        n_frames=np.int64(n_frames), n_callbacks=np.int64(n_callbacks), interval=np.float32(interval), postfx=np.int64(postfx),
        expected=outputs.astype(np.float32),
    )


def numpy_render(spec, events, n_frames, n_callbacks, interval, postfx):
    sc = on.Scene()
    handles = []
    for s in spec["sources"]:
        if s["kind"] == "frames":
            src = on.frames_source(s["rate"], s["clip"], s["start"], fixed_gain_db=s.get("gain_db"))
        elif s["kind"] == "cycle":
            src = on.cycle_source(s["rate"], s["clip"], fixed_gain_db=s.get("gain_db"))
        elif s["kind"] == "downmix":
            src = on.downmix_source(s["rate"], s["clip"], s["start"], fixed_gain_db=s.get("gain_db"))
        elif s["kind"] == "sine":
            src = on.sine_source(s["phase"], s["hz"], fixed_gain_db=s.get("gain_db"))
        else:
            src = on.constant_source(s["value"])
        handles.append(sc.play(src, s["pos"], s["vel"], s["radius"]))
    outs = []
    for cb in range(n_callbacks):
        for ev in events.get(cb, []):
            if ev[0] == "motion":
                sc.set_motion(handles[ev[1]], np.asarray(ev[2], np.float32), np.asarray(ev[3], np.float32), ev[4])
            else:
                sc.set_listener_rotation(np.asarray(ev[1], np.float32))
        o = sc.sample(interval, n_frames)
        if postfx == 1:
            o = on.reinhard(o)
        elif postfx == 2:
            o = on.tanh_clip(o)
        outs.append(o)
    return np.stack(outs)


def make(name, spec, events, n_frames, n_callbacks, postfx=0):
    if ONLY and name not in ONLY:
        return
    interval = np.float32(1.0) / np.float32(48000)
    ob = scenario.play_all(scenario.OracleBackend(), spec)
    if postfx:
        ob.set_postfx(postfx)
    ref = scenario.run_events([ob], spec, n_frames, n_callbacks, interval=interval, events=events)["oracle"]
    ref_np = numpy_render(spec, events, n_frames, n_callbacks, interval, postfx)
    if not np.array_equal(ref, ref_np):
        raise SystemExit(f"{name}: C oracle and numpy restatement disagree -- not writing a fixture")
    assert np.abs(ref).max() > 0
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **pack(spec, events, n_frames, n_callbacks, interval, postfx, ref))
    print(f"{name}: {len(spec['sources'])} sources, {n_callbacks}x{n_frames} frames, {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    rng = np.random.default_rng(7)
    # 1. FramesSignal sources only, full-tile callbacks, motion + rotation events
    spec = scenario.random_spec(1001, 5, clip_len=7200, cube=8.0, start=0.05)
    ev = {1: [("motion", 2, spec["sources"][2]["pos"] + np.float32(0.5), spec["sources"][2]["vel"], False)],
          2: [("rotation", [np.cos(0.2), 0.0, np.sin(0.2), 0.0])],
          3: [("motion", 4, spec["sources"][4]["pos"] - np.float32(1.5), spec["sources"][4]["vel"], True)]}
    make("frames_motion_rotation", spec, ev, 1024, 4)
    # 2. mixed kinds + FixedGain, ragged callback length (partial chunk)
    spec = scenario.random_spec(1002, 5, kinds=("frames", "sine", "constant", "frames"), gain_db=(None, -6.0, None, 3.0), clip_len=5200, cube=8.0, start=0.05)
    make("mixed_kinds_ragged", spec, {}, 700, 3)
    # 3. clip edges: negative start, run off the end, removal after propagation delay
    sc = synth.make_scene(1003, 4, cube=6.0)
    srcs = [{"kind": "frames", "clip": synth.noise_clip(1003, i, 600 + 500 * i), "rate": 48000, "start": -0.003 * i,
             "pos": sc["position"][i], "vel": sc["velocity"][i], "radius": 0.1, "gain_db": None} for i in range(4)]
    make("clip_edges_removal", {"sources": srcs}, {}, 1024, 5)
    # 4. resample ratios far from 1 + Reinhard post filter
    srcs = [{"kind": "frames", "clip": synth.noise_clip(1004, i, 4000), "rate": r, "start": 0.03,
             "pos": np.array([2.0 + i, 0.5, -1.0], np.float32), "vel": np.array([-25.0, 4.0, 9.0], np.float32), "radius": 0.1, "gain_db": None}
            for i, r in enumerate((44100, 22050, 96000))]
    make("resample_reinhard", {"sources": srcs}, {}, 512, 3, postfx=1)
    # 5. Cycle sources in the Seek set (short loops wrap many times per callback) between FramesSignals
    spec = scenario.random_spec(1005, 5, kinds=("cycle", "frames", "cycle"), gain_db=(None, None, -4.0), clip_len=5200, cube=8.0, start=0.05, cycle_len=150)
    spec["sources"][2]["clip"] = spec["sources"][2]["clip"][:7]
    ev = {1: [("motion", 0, spec["sources"][0]["pos"] + np.float32(0.7), spec["sources"][0]["vel"], False)]}
    make("cycle_in_scene", spec, ev, 700, 3)
    # 6. Cycle loops of several tiles (round 4: most of their tiles are rendered from the staged window like a clip's, the tiles that
    #    touch the loop's end by the row path), three clip rates, a jump that puts the ears of a source on different laps
    spec = scenario.random_spec(1006, 7, kinds=("cycle",), cube=9.0, cycle_len=8)
    for i, (n, rate) in enumerate(((2100, 48000), (9000, 96000), (3001, 44100), (1500, 48000), (20000, 192000), (5000, 48000), (700, 48000))):
        spec["sources"][i]["clip"] = synth.noise_clip(1006, i, n)
        spec["sources"][i]["rate"] = rate
        spec["sources"][i]["gain_db"] = (None, -3.5)[i % 2]
    ev = {2: [("motion", 1, spec["sources"][1]["pos"] + np.float32(40.0), spec["sources"][1]["vel"], True)],
          3: [("rotation", [np.cos(0.3), np.sin(0.3), 0.0, 0.0])]}
    make("cycle_long_loops", spec, ev, 1024, 5)
    # 7. Downmix<FramesSignal<[f32;2]>> (round 4: interleaved stereo windows): three clip rates, a source at rest (the constant-fract
    #    branch of frames.rs:180-187), a ragged callback length (downmix.rs:24-29 advances the inner clip a whole buffer per chunk)
    spec = scenario.random_spec(1007, 6, kinds=("downmix", "downmix", "frames"), gain_db=(None, 2.0, None), clip_len=6000, cube=7.0, start=-0.004)
    for i, rate in enumerate((48000, 96000, 48000, 44100, 48000, 22050)):
        spec["sources"][i]["rate"] = rate
    spec["sources"][4]["vel"] = np.zeros(3, np.float32)
    make("downmix_rates", spec, {}, 700, 5)


if __name__ == "__main__":
    main()
