"""Filter-chain scenarios (FixedGain / Gain / Speed around a FramesSignal; play_buffered or play in a SpatialScene, or a Mixer) as
plain data, playable on the C oracle, the numpy restatement and the HIP path -- the fixtures of tests/golden/chains_*.npz
(tests/golden/gen_golden_chains.py, tests/test_golden_chains.py)."""
from __future__ import annotations

import numpy as np

FIXED, GAIN, SPEED = 1, 2, 3          # chain entries (kind, param), innermost first; GAIN's param: initial amplitude ratio or NaN
REINHARD, TANH = 4, 5                 # per-source soft clips (reinhard.rs:22-50, tanh.rs:16-44), param unused


def pack(spec, expected):
    srcs = spec["sources"]
    n = len(srcs)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(s["clip"]) for s in srcs])
    ck = np.zeros((n, 4), np.int32)
    cp = np.full((n, 4), np.nan, np.float32)
    for i, s in enumerate(srcs):
        for w, (k, p) in enumerate(s["chain"]):
            ck[i, w], cp[i, w] = k, p
    return dict(
        mixer=np.int64(spec["mixer"]), clip_data=np.concatenate([s["clip"] for s in srcs]).astype(np.float32), clip_offsets=offs,
        rate=np.array([s["rate"] for s in srcs], np.int64), start=np.array([s["start"] for s in srcs], np.float64),
        pos=np.stack([s["pos"] for s in srcs]).astype(np.float32), vel=np.stack([s["vel"] for s in srcs]).astype(np.float32),
        radius=np.array([s["radius"] for s in srcs], np.float32), chain_kind=ck, chain_param=cp,
        buffered=np.array([s["buffered"] for s in srcs], np.int32), ring_rate=np.array([s["ring_rate"] for s in srcs], np.int64),
        max_distance=np.array([s["max_distance"] for s in srcs], np.float32), buffer_duration=np.array([s["buffer_duration"] for s in srcs], np.float32),
        ev_ctl=np.array([(cb, i, w, v) for cb, i, w, v in spec["ctl"]], np.float64).reshape(-1, 4),
        ev_motion=np.array([(cb, j, *p, *v, int(d)) for cb, j, p, v, d in spec["motion"]], np.float64).reshape(-1, 9),
        ev_rotation=np.array([(cb, *q) for cb, q in spec["rotation"]], np.float64).reshape(-1, 5),
        n_frames=np.int64(spec["n_frames"]), n_callbacks=np.int64(spec["n_callbacks"]), interval=np.float32(spec["interval"]),
        expected=np.asarray(expected, np.float32))


def load(path):
    z = np.load(path)
    n = len(z["rate"])
    srcs = []
    for i in range(n):
        chain = [(int(z["chain_kind"][i, w]), float(z["chain_param"][i, w])) for w in range(4) if z["chain_kind"][i, w]]
        srcs.append({"clip": z["clip_data"][z["clip_offsets"][i]:z["clip_offsets"][i + 1]].copy(), "rate": int(z["rate"][i]), "start": float(z["start"][i]),
                     "pos": z["pos"][i], "vel": z["vel"][i], "radius": float(z["radius"][i]), "chain": chain, "buffered": bool(z["buffered"][i]),
                     "ring_rate": int(z["ring_rate"][i]), "max_distance": float(z["max_distance"][i]), "buffer_duration": float(z["buffer_duration"][i])})
    spec = {"mixer": bool(z["mixer"]), "sources": srcs,
            "ctl": [(int(r[0]), int(r[1]), int(r[2]), float(r[3])) for r in z["ev_ctl"]],
            "motion": [(int(r[0]), int(r[1]), r[2:5].astype(np.float32), r[5:8].astype(np.float32), bool(r[8])) for r in z["ev_motion"]],
            "rotation": [(int(r[0]), r[1:5].astype(np.float32)) for r in z["ev_rotation"]],
            "n_frames": int(z["n_frames"]), "n_callbacks": int(z["n_callbacks"]), "interval": np.float32(z["interval"])}
    return spec, z["expected"]


class _Backend:
    """play every source of the spec; `ctl(i, w, v)`: the control of source i's filter w (GainControl / SpeedControl) stores v"""

    def run_events(self, spec, cb):
        for c, i, w, v in spec["ctl"]:
            if c == cb:
                self.ctl(i, w, v)
        for c, j, p, v, d in spec["motion"]:
            if c == cb:
                self.motion(j, p, v, d)
        for c, q in spec["rotation"]:
            if c == cb:
                self.rotation(q)


class CBackend(_Backend):
    def __init__(self, spec):
        from oracle import oracle_c as oc
        self.top = oc.Mixer(2) if spec["mixer"] else oc.SpatialScene()
        self.handles, self.controls = [], []
        for s in spec["sources"]:
            sig = oc.FramesSignal(oc.Frames(s["rate"], s["clip"]), s["start"])
            ctl = []
            for kind, p in s["chain"]:
                if kind == FIXED:
                    sig = oc.FixedGain(sig, p)
                    ctl.append(None)
                elif kind in (REINHARD, TANH):
                    sig = oc.Reinhard(sig) if kind == REINHARD else oc.Tanh(sig)
                    ctl.append(None)
                elif kind == GAIN:
                    sig = oc.Gain(sig)
                    if not np.isnan(p):
                        sig.init_amplitude_ratio(p)
                    ctl.append(sig.set_amplitude_ratio)
                else:
                    sig = oc.Speed(sig)
                    sig.set_speed(p)
                    ctl.append(sig.set_speed)
            self.controls.append(ctl)
            if spec["mixer"]:
                self.handles.append(self.top.play(oc.MonoToStereo(sig)))
            elif s["buffered"]:
                self.handles.append(self.top.play_buffered(sig, oc.SpatialOptions(s["pos"], s["vel"], s["radius"]), s["max_distance"], s["ring_rate"], s["buffer_duration"]))
            else:
                self.handles.append(self.top.play(sig, oc.SpatialOptions(s["pos"], s["vel"], s["radius"])))

    def ctl(self, i, w, v):
        self.controls[i][w](v)

    def motion(self, j, p, v, d):
        self.handles[j].set_motion(p, v, d)

    def rotation(self, q):
        self.top.set_listener_rotation(q)

    def sample(self, interval, n):
        return self.top.sample_n(interval, n)


class NumpyBackend(_Backend):
    def __init__(self, spec):
        from oracle import oracle_np as on
        self.on = on
        self.top = on.Mixer(2) if spec["mixer"] else on.Scene()
        self.handles, self.controls = [], []
        for s in spec["sources"]:
            src = on.frames_source(s["rate"], s["clip"], s["start"])
            ctl = []
            for kind, p in s["chain"]:
                if kind == FIXED:
                    src = on.fixed_gain_filter(src, p)
                    ctl.append(None)
                elif kind in (REINHARD, TANH):
                    src = on.reinhard_filter(src) if kind == REINHARD else on.tanh_filter(src)
                    ctl.append(None)
                elif kind == GAIN:
                    src = on.gain_filter(src, None if np.isnan(p) else p)
                    ctl.append((on.gain_control_set, src))
                else:
                    src = on.speed_filter(src, p)
                    ctl.append((on.speed_control_set, src))
            self.controls.append(ctl)
            if spec["mixer"]:
                self.handles.append(self.top.play(src))
            elif s["buffered"]:
                self.handles.append(self.top.play_buffered(src, s["pos"], s["vel"], s["radius"], s["max_distance"], s["ring_rate"], s["buffer_duration"]))
            else:
                self.handles.append(self.top.play(src, s["pos"], s["vel"], s["radius"]))

    def ctl(self, i, w, v):
        f, src = self.controls[i][w]
        f(src, v)

    def motion(self, j, p, v, d):
        self.top.set_motion(self.handles[j], p, v, d)

    def rotation(self, q):
        self.top.set_listener_rotation(q)

    def sample(self, interval, n):
        return self.top.sample(interval, n)


class HipBackend(_Backend):
    def __init__(self, spec, mode=1):
        import oddio_amd as oa
        n = len(spec["sources"])
        if spec["mixer"]:
            self.control, self.top = oa.Mixer(max_sources=n + 8, max_frames=max(spec["n_frames"], 1))
        else:
            self.control, self.top = oa.SpatialScene(max_sources=n + 8, max_frames=max(spec["n_frames"], 1))
        self.top.set_mode(mode)
        self.handles, self.controls = [], []
        for s in spec["sources"]:
            sig = oa.FramesSignal(oa.Frames.from_slice(s["rate"], s["clip"]), s["start"])
            ctl = []
            for kind, p in s["chain"]:
                if kind == FIXED:
                    sig = oa.FixedGain(sig, p)
                    ctl.append(None)
                elif kind in (REINHARD, TANH):
                    sig = oa.Reinhard(sig) if kind == REINHARD else oa.Tanh(sig)
                    ctl.append(None)
                elif kind == GAIN:
                    gc, sig = oa.Gain.new(sig)
                    if not np.isnan(p):
                        gc.set_amplitude_ratio(p)      # before play: the initial ratio (Gain::set_amplitude_ratio)
                    ctl.append(gc.set_amplitude_ratio)
                else:
                    sc_, sig = oa.Speed.new(sig)
                    sc_.set_speed(p)
                    ctl.append(sc_.set_speed)
            self.controls.append(ctl)
            if spec["mixer"]:
                self.handles.append(self.control.play(oa.MonoToStereo(sig)))
            elif s["buffered"]:
                self.handles.append(self.control.play_buffered(sig, oa.SpatialOptions(s["pos"], s["vel"], s["radius"]), s["max_distance"], s["ring_rate"], s["buffer_duration"]))
            else:
                self.handles.append(self.control.play(sig, oa.SpatialOptions(s["pos"], s["vel"], s["radius"])))

    def ctl(self, i, w, v):
        self.controls[i][w](v)

    def motion(self, j, p, v, d):
        self.handles[j].set_motion(np.asarray(p, np.float32), np.asarray(v, np.float32), d)

    def rotation(self, q):
        self.control.set_listener_rotation(np.asarray(q, np.float32))

    def sample(self, interval, n):
        return self.top.sample_n(interval, n)

    def close(self):
        self.top.close()


def run(backend, spec):
    outs = []
    for cb in range(spec["n_callbacks"]):
        backend.run_events(spec, cb)
        outs.append(np.array(backend.sample(spec["interval"], spec["n_frames"]), np.float32).copy())
    return np.stack(outs)
