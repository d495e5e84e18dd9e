"""Scenario specs playable on the CPU oracle and on the HIP path (identical inputs for both).

A spec is plain data: a list of source dicts plus timed events; `golden` fixtures are specs plus
the oracle's outputs.
"""
from __future__ import annotations

import numpy as np

from oddio_amd import synth


def random_spec(seed, n_src, kinds=("frames",), clip_len=24000, rate=48000, start=0.3, gain_db=(None,),
                cube=50.0, vmax=20.0, noise=True, cycle_len=1000):
    sc = synth.make_scene(seed, n_src, cube=cube, vmax=vmax)
    sources = []
    for i in range(n_src):
        kind = kinds[i % len(kinds)]
        src = {"kind": kind, "pos": sc["position"][i], "vel": sc["velocity"][i], "radius": float(sc["radius"][i]),
               "gain_db": gain_db[i % len(gain_db)]}
        if kind == "frames":
            src["clip"] = synth.noise_clip(seed, i, clip_len) if noise else synth.sine_clip(sc["freq_hz"][i], clip_len, rate)
            src["rate"] = rate
            src["start"] = start
        elif kind == "downmix":
            src["clip"] = np.stack([synth.noise_clip(seed, i, clip_len), synth.noise_clip(seed + 999, i, clip_len)], axis=1)
            src["rate"] = rate
            src["start"] = start
        elif kind == "cycle":
            src["clip"] = synth.noise_clip(seed, i, cycle_len) if noise else synth.sine_clip(sc["freq_hz"][i], cycle_len, rate)
            src["rate"] = rate
        elif kind == "sine":
            src["phase"] = float(sc["phase"][i])
            src["hz"] = float(sc["freq_hz"][i])
        else:
            src["value"] = 0.75
            src["gain_db"] = None
        sources.append(src)
    return {"sources": sources}


class OracleBackend:
    name = "oracle"

    def __init__(self):
        from oracle import oracle_c as oc
        self.oc = oc
        self.scene = oc.SpatialScene()
        self.handles = []
        self.top = self.scene

    def play(self, src):
        oc = self.oc
        if src["kind"] == "frames":
            sig = oc.FramesSignal(oc.Frames(src["rate"], src["clip"]), src["start"])
        elif src["kind"] == "downmix":
            sig = oc.Downmix(oc.FramesSignal(oc.Frames(src["rate"], src["clip"]), src["start"]))
        elif src["kind"] == "cycle":
            sig = oc.Cycle(oc.Frames(src["rate"], src["clip"]))
        elif src["kind"] == "sine":
            sig = oc.Sine(src["phase"], src["hz"])
        else:
            sig = oc.Constant(src["value"])
        if src.get("gain_db") is not None:
            sig = oc.FixedGain(sig, src["gain_db"])
        h = self.scene.play(sig, oc.SpatialOptions(src["pos"], src["vel"], src["radius"]))
        self.handles.append(h)
        return h

    def set_postfx(self, kind):
        self.top = {0: lambda s: s, 1: self.oc.Reinhard, 2: self.oc.Tanh}[kind](self.scene)

    def set_listener_rotation(self, q):
        self.scene.set_listener_rotation(q)

    def sample(self, interval, n):
        return self.top.sample_n(interval, n)

    def sample_f64(self, interval, n):
        return self.scene.sample_f64acc(interval, n)

    def __len__(self):
        return len(self.scene)


class HipBackend:
    name = "hip"

    def __init__(self, max_sources=4096, max_frames=4096, mode=0, device=0):
        import oddio_amd as oa
        self.oa = oa
        self.control, self.scene = oa.SpatialScene(device=device, max_sources=max_sources, max_frames=max_frames)
        self.scene.set_mode(mode)
        self.handles = []
        self.top = self.scene
        self._clips = {}

    def play(self, src):
        oa = self.oa
        if src["kind"] == "frames":
            key = id(src["clip"])
            if key not in self._clips:
                self._clips[key] = oa.Frames.from_slice(src["rate"], src["clip"])
            sig = oa.FramesSignal(self._clips[key], src["start"])
        elif src["kind"] == "downmix":
            key = id(src["clip"])
            if key not in self._clips:
                self._clips[key] = oa.Frames.from_slice(src["rate"], src["clip"])
            sig = oa.Downmix(oa.FramesSignal(self._clips[key], src["start"]))
        elif src["kind"] == "cycle":
            key = id(src["clip"])
            if key not in self._clips:
                self._clips[key] = oa.Frames.from_slice(src["rate"], src["clip"])
            sig = oa.Cycle(self._clips[key])
        elif src["kind"] == "sine":
            sig = oa.Sine(src["phase"], src["hz"])
        else:
            sig = oa.Constant(src["value"])
        if src.get("gain_db") is not None:
            sig = oa.FixedGain(sig, src["gain_db"])
        h = self.control.play(sig, oa.SpatialOptions(src["pos"], src["vel"], src["radius"]))
        self.handles.append(h)
        return h

    def set_postfx(self, kind):
        self.scene.set_postfx(kind)

    def set_listener_rotation(self, q):
        self.control.set_listener_rotation(q)

    def sample(self, interval, n):
        return self.scene.sample_n(interval, n)

    def __len__(self):
        return len(self.scene)

    def close(self):
        self.scene.close()


def play_all(backend, spec):
    for src in spec["sources"]:
        backend.play(src)
    return backend


def run_events(backends, spec, n_frames, n_callbacks, interval=None, events=None):
    """Drive every backend through the same callbacks; returns {name: [n_callbacks, n_frames, 2]}."""
    interval = np.float32(1.0) / np.float32(48000) if interval is None else np.float32(interval)
    events = events or {}
    outs = {b.name: [] for b in backends}
    for cb in range(n_callbacks):
        for ev in events.get(cb, []):
            for b in backends:
                if ev[0] == "motion":
                    _, j, p, v, disc = ev
                    b.handles[j].set_motion(np.asarray(p, np.float32), np.asarray(v, np.float32), disc)
                elif ev[0] == "rotation":
                    b.set_listener_rotation(np.asarray(ev[1], np.float32))
                elif ev[0] == "play":
                    b.play(ev[1])
        for b in backends:
            outs[b.name].append(b.sample(interval, n_frames).copy())
    return {k: np.stack(v) for k, v in outs.items()}
