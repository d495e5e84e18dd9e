"""Randomised control-plane soak: the same seeded stream of operations (play / play_buffered /
set_motion / set_listener_rotation / gain and speed controls / handle drops / ragged callback
sizes) drives the CPU oracle and the HIP scene in ORDERED mode.  Only exactly reproducible source
kinds are used (FramesSignal, Cycle, Constant, FixedGain / Gain / Speed chains), so every callback
must be bit-identical, and set sizes and is_finished flags must agree throughout.  GPU only.

This is the test of the host side: slot bookkeeping under swap_remove (src/set.rs:170-188) for both
sets, handle-id reuse, latest-value-wins motion (src/swap.rs), propagation-delay removal
(src/spatial.rs:243-261)."""
import os

import numpy as np
import pytest

from oddio_amd import synth
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

INTERVAL = np.float32(1.0) / np.float32(48000)
# soak options (tests/soak_fuzz.py): more live sources / operations per callback than the default 60 / 0-4
LIVE_MAX = int(os.environ.get("ODDIO_FUZZ_LIVE", "60"))
OPS_MAX = int(os.environ.get("ODDIO_FUZZ_OPS", "5"))
# soak options: the callback lengths drawn from (default: the list every recorded seed was run with); ODDIO_FUZZ_FRAMES=528,640,768,960,1008,1024 with
# ODDIO_HIP_PAIR_MIN_GROUPS=1 hunts spatial_mix_pair's lane-granular instantiations (LANE16), ODDIO_FUZZ_PLAIN=1 keeps Cycle / Downmix sources -- which
# take a scene off the pair kernel -- out of the draw
FRAME_CHOICES = [int(v) for v in os.environ["ODDIO_FUZZ_FRAMES"].split(",")] if os.environ.get("ODDIO_FUZZ_FRAMES") else [1024, 1024, 512, 256, 1, 300, 1300, 1536]
PLAIN_KINDS = os.environ.get("ODDIO_FUZZ_PLAIN", "") not in ("", "0")
ODDIO_FUZZ_CHAINS = os.environ.get("ODDIO_FUZZ_CHAINS", "1") != "0"    # random nests of FixedGain / Reinhard (/ Tanh) around played sources


def _vec(rng, scale):
    return (rng.uniform(-1, 1, 3) * scale).astype(np.float32)


# 5094: a listener rotation inside the callback gave one ear a near-unit resample ratio and the other a full-size
# window (padded re-layout overran its buffer); 5173: a control update sent for a finished source whose handle id was
# released and handed to a new source before the next callback.  Both found by tests/soak_fuzz.py.
@pytest.mark.parametrize("seed", list(range(10)) + [5094, 5173])
def test_random_operations_bit_exact(seed):
    import oddio_amd as oa
    rng = np.random.default_rng(9000 + seed)
    rng_chain = np.random.default_rng(424242 + seed)
    control, scene = oa.SpatialScene(max_sources=LIVE_MAX + 36, max_frames=1536)
    if LIVE_MAX > 200:
        scene.reserve_buffered(LIVE_MAX + 36)
    fuzz_mode = os.environ.get("ODDIO_FUZZ_MODE", "")      # soak options: the multi-wavefront tree sum (fast / unfused) or the tracked sum, compared with a tolerance
    fast = fuzz_mode in ("fast", "unfused", "tracked")
    scene.set_mode({"fast": oa.MODE_FAST, "unfused": oa.MODE_FAST_UNFUSED, "tracked": oa.MODE_TRACKED}.get(fuzz_mode, oa.MODE_ORDERED))
    scene.set_postfx((0, 1, 0)[seed % 3])
    ref_scene = oc.SpatialScene()
    ref = oc.Reinhard(ref_scene) if seed % 3 == 1 else ref_scene
    live = []          # [hip handle, oracle handle, hip controls, oracle controls]
    clip_no = 0
    peak_len, removed_seen = 0, False
    sig_scale = 0.0
    for cb in range(60):
        n_ops = int(rng.integers(0, OPS_MAX))
        for _ in range(n_ops):
            op = rng.choice(["play", "play", "buffered", "motion", "motion", "rotation", "control", "drop"])
            if op in ("play", "buffered") and len(live) < LIVE_MAX:
                clip_no += 1
                kind = rng.choice(["frames", "frames", "constant"] if PLAIN_KINDS else ["frames", "frames", "cycle", "constant"] + (["downmix"] if op == "play" else []))
                rate = int(rng.choice([48000, 44100, 22050]))
                pos, vel = _vec(rng, 12.0), _vec(rng, 25.0)
                radius = float(rng.choice([0.1, 0.5]))
                hc, rc = [], []
                if kind == "frames":
                    clip = synth.noise_clip(seed, clip_no, int(rng.integers(1, 9000)))
                    start = float(rng.uniform(-0.01, 0.02))
                    sh, so = oa.FramesSignal(oa.Frames.from_slice(rate, clip), start), oc.FramesSignal(oc.Frames(rate, clip), start)
                elif kind == "downmix":
                    n = int(rng.integers(1, 9000))
                    clip = np.stack([synth.noise_clip(seed, clip_no, n), synth.noise_clip(seed + 77, clip_no, n)], axis=1)
                    start = float(rng.uniform(-0.01, 0.02))
                    sh = oa.Downmix(oa.FramesSignal(oa.Frames.from_slice(rate, clip), start))
                    so = oc.Downmix(oc.FramesSignal(oc.Frames(rate, clip), start))
                elif kind == "cycle":
                    n = int(rng.integers(1, 700))
                    if clip_no % 3 == 0:
                        n = 1500 + 97 * n          # loops of several tiles: most of their tiles take the staged-window path (round 4), some wrap
                    clip = synth.noise_clip(seed, clip_no, n)
                    sh, so = oa.Cycle(oa.Frames.from_slice(rate, clip)), oc.Cycle(oc.Frames(rate, clip))
                else:
                    val = float(rng.uniform(-1, 1))
                    sh, so = oa.Constant(val), oc.Constant(val)
                if rng.random() < 0.4 and kind != "constant":
                    db = float(rng.uniform(-12, 6))
                    sh, so = oa.FixedGain(sh, db), oc.FixedGain(so, db)
                if op == "play" and ODDIO_FUZZ_CHAINS and rng_chain.random() < 0.3:
                    # round 6: nests of the Seek wrappers around a played source (FX_CHAIN, and the compact one-gain-one-clip form); drawn
                    # from a stream of their own so that the seeds pinned above keep their scenes.  Tanh only where the compare is a tolerance.
                    for _ in range(int(rng_chain.integers(1, 4))):
                        w = rng_chain.choice(["fixed", "reinhard", "tanh"] if fast else ["fixed", "reinhard"])
                        if w == "fixed":
                            db2 = float(rng_chain.uniform(-9, 9))
                            sh, so = oa.FixedGain(sh, db2), oc.FixedGain(so, db2)
                        elif w == "reinhard":
                            sh, so = oa.Reinhard(sh), oc.Reinhard(so)
                        else:
                            sh, so = oa.Tanh(sh), oc.Tanh(so)
                if op == "buffered":
                    for _ in range(int(rng.integers(0, 3))):
                        if rng.random() < 0.5:
                            c, sh = oa.Gain.new(sh)
                            so = oc.Gain(so)
                        else:
                            c, sh = oa.Speed.new(sh)
                            so = oc.Speed(so)
                        hc.append(c)
                        rc.append(so)
                    h = control.play_buffered(sh, oa.SpatialOptions(pos, vel, radius), 60.0, 48000, 0.05)
                    r = ref_scene.play_buffered(so, oc.SpatialOptions(pos, vel, radius), 60.0, 48000, 0.05)
                else:
                    h = control.play(sh, oa.SpatialOptions(pos, vel, radius))
                    r = ref_scene.play(so, oc.SpatialOptions(pos, vel, radius))
                live.append([h, r, hc, rc])
            elif op == "motion" and live:
                for _ in range(int(rng.integers(1, 4))):      # several per callback: the latest must win
                    k = int(rng.integers(0, len(live)))
                    pos, vel, disc = _vec(rng, 12.0), _vec(rng, 25.0), bool(rng.random() < 0.3)
                    live[k][0].set_motion(pos, vel, disc)
                    live[k][1].set_motion(pos, vel, disc)
            elif op == "rotation":
                q = rng.normal(size=4).astype(np.float32)
                q = (q / np.linalg.norm(q)).astype(np.float32)
                control.set_listener_rotation(q)
                ref_scene.set_listener_rotation(q)
            elif op == "control":
                cands = [e for e in live if e[2]]
                if cands:
                    e = cands[int(rng.integers(0, len(cands)))]
                    i = int(rng.integers(0, len(e[2])))
                    if isinstance(e[2][i], oa.GainControl):
                        v = float(rng.uniform(0.0, 2.0))
                        e[2][i].set_amplitude_ratio(v)
                        e[3][i].set_amplitude_ratio(v)
                    else:
                        v = float(rng.uniform(0.5, 1.6))
                        e[2][i].set_speed(v)
                        e[3][i].set_speed(v)
            elif op == "drop" and live:
                k = int(rng.integers(0, len(live)))
                if live[k][0].is_finished():                 # dropping the handle frees the id for reuse
                    live[k][0].release()
                    live.pop(k)
        n = int(rng.choice(FRAME_CHOICES))
        a = ref.sample_n(INTERVAL, n)
        b = scene.sample_n(INTERVAL, n)
        if fast:
            # 1e-5 of the signal's scale: the largest |reference| seen so far, not of this callback alone -- a one-frame callback whose
            # sources happen to cancel has a "peak" far below the running sums that set every mode's rounding (soak seed 43071, round 6:
            # TRACKED, n = 1, |out| = 0.0029, off by one ulp of its running sum: 3.4e-8)
            sig_scale = max(sig_scale, float(np.abs(a).max()))
            np.testing.assert_allclose(b, a, rtol=0, atol=1e-5 * max(sig_scale, 1e-3), err_msg=f"seed {seed} callback {cb} n {n}")
        else:
            np.testing.assert_array_equal(b, a, err_msg=f"seed {seed} callback {cb} n {n}")
        assert (len(scene), scene.len_buffered()) == (len(ref_scene), ref_scene.len_buffered())
        peak_len = max(peak_len, len(scene) + scene.len_buffered())
        removed_seen = removed_seen or any(e[0].is_finished() for e in live)
        assert [e[0].is_finished() for e in live] == [e[1].is_finished() for e in live]
    assert clip_no >= 15 and peak_len >= 5 and removed_seen, (clip_no, peak_len, removed_seen)   # the run exercised something
    scene.close()


@pytest.mark.parametrize("mode", ["fast", "unfused", "tracked"])
@pytest.mark.parametrize("seed", [3, 11])
def test_random_operations_in_the_tolerance_modes(monkeypatch, seed, mode):
    """The same operation stream in the modes whose contract is the 1e-5 tolerance (tests/soak_fuzz.py runs them over thousands of
    seeds): FAST, FAST_UNFUSED and TRACKED, with spatial_mix_pair and the TRACK instantiations taking these small scenes too
    (ODDIO_HIP_PAIR_MIN_GROUPS=1)."""
    monkeypatch.setenv("ODDIO_FUZZ_MODE", mode)
    monkeypatch.setenv("ODDIO_HIP_PAIR_MIN_GROUPS", "1")
    test_random_operations_bit_exact(seed)


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("seed", range(6))
def test_random_operations_unsynchronised(seed, exact):
    """The same operation stream with the callbacks ENQUEUED (oddio_hip_scene_sample_device): inserts, removals, handle-id
    reuse and control updates all take effect in stream order on the device while the host runs ahead (decisions that
    need `is_finished` use the oracle's answer).  The library stages a callback's updates in one of three pinned slots;
    by default a callback that finds all of them still unread by the device leaves its updates queued for the next
    one (it never waits), so `exact=False` lets the host run at most three callbacks ahead.  `exact=True`
    (oddio_hip_scene_set_exact_updates) enqueues all 60 callbacks without a single host wait of the test's own: the
    library then waits for a slot instead of letting updates slip.  Outputs compared afterwards."""
    import torch
    import oddio_amd as oa
    rng = np.random.default_rng(19000 + seed)
    rng_chain = np.random.default_rng(525252 + seed)
    control, scene = oa.SpatialScene(max_sources=LIVE_MAX + 36, max_frames=1536)
    if LIVE_MAX > 200:
        scene.reserve_buffered(LIVE_MAX + 36)
    fast = False
    scene.set_exact_updates(exact)
    sizes, wants = [], []
    dev_out = torch.zeros((60, 1536, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()   # (the fill runs on torch's stream, which the library's own streams do not wait for)
    scene.set_mode(oa.MODE_FAST if fast else oa.MODE_ORDERED)
    scene.set_postfx((0, 1, 0)[seed % 3])
    ref_scene = oc.SpatialScene()
    ref = oc.Reinhard(ref_scene) if seed % 3 == 1 else ref_scene
    live = []          # [hip handle, oracle handle, hip controls, oracle controls]
    clip_no = 0
    peak_len, removed_seen = 0, False
    for cb in range(60):
        n_ops = int(rng.integers(0, OPS_MAX))
        for _ in range(n_ops):
            op = rng.choice(["play", "play", "buffered", "motion", "motion", "rotation", "control", "drop"])
            if op in ("play", "buffered") and len(live) < LIVE_MAX:
                clip_no += 1
                kind = rng.choice(["frames", "frames", "constant"] if PLAIN_KINDS else ["frames", "frames", "cycle", "constant"] + (["downmix"] if op == "play" else []))
                rate = int(rng.choice([48000, 44100, 22050]))
                pos, vel = _vec(rng, 12.0), _vec(rng, 25.0)
                radius = float(rng.choice([0.1, 0.5]))
                hc, rc = [], []
                if kind == "frames":
                    clip = synth.noise_clip(seed, clip_no, int(rng.integers(1, 9000)))
                    start = float(rng.uniform(-0.01, 0.02))
                    sh, so = oa.FramesSignal(oa.Frames.from_slice(rate, clip), start), oc.FramesSignal(oc.Frames(rate, clip), start)
                elif kind == "downmix":
                    n = int(rng.integers(1, 9000))
                    clip = np.stack([synth.noise_clip(seed, clip_no, n), synth.noise_clip(seed + 77, clip_no, n)], axis=1)
                    start = float(rng.uniform(-0.01, 0.02))
                    sh = oa.Downmix(oa.FramesSignal(oa.Frames.from_slice(rate, clip), start))
                    so = oc.Downmix(oc.FramesSignal(oc.Frames(rate, clip), start))
                elif kind == "cycle":
                    n = int(rng.integers(1, 700))
                    if clip_no % 3 == 0:
                        n = 1500 + 97 * n          # loops of several tiles: most of their tiles take the staged-window path (round 4), some wrap
                    clip = synth.noise_clip(seed, clip_no, n)
                    sh, so = oa.Cycle(oa.Frames.from_slice(rate, clip)), oc.Cycle(oc.Frames(rate, clip))
                else:
                    val = float(rng.uniform(-1, 1))
                    sh, so = oa.Constant(val), oc.Constant(val)
                if rng.random() < 0.4 and kind != "constant":
                    db = float(rng.uniform(-12, 6))
                    sh, so = oa.FixedGain(sh, db), oc.FixedGain(so, db)
                if op == "play" and ODDIO_FUZZ_CHAINS and rng_chain.random() < 0.3:
                    # round 6: nests of the Seek wrappers around a played source (FX_CHAIN, and the compact one-gain-one-clip form); drawn
                    # from a stream of their own so that the seeds pinned above keep their scenes.  Tanh only where the compare is a tolerance.
                    for _ in range(int(rng_chain.integers(1, 4))):
                        w = rng_chain.choice(["fixed", "reinhard", "tanh"] if fast else ["fixed", "reinhard"])
                        if w == "fixed":
                            db2 = float(rng_chain.uniform(-9, 9))
                            sh, so = oa.FixedGain(sh, db2), oc.FixedGain(so, db2)
                        elif w == "reinhard":
                            sh, so = oa.Reinhard(sh), oc.Reinhard(so)
                        else:
                            sh, so = oa.Tanh(sh), oc.Tanh(so)
                if op == "buffered":
                    for _ in range(int(rng.integers(0, 3))):
                        if rng.random() < 0.5:
                            c, sh = oa.Gain.new(sh)
                            so = oc.Gain(so)
                        else:
                            c, sh = oa.Speed.new(sh)
                            so = oc.Speed(so)
                        hc.append(c)
                        rc.append(so)
                    h = control.play_buffered(sh, oa.SpatialOptions(pos, vel, radius), 60.0, 48000, 0.05)
                    r = ref_scene.play_buffered(so, oc.SpatialOptions(pos, vel, radius), 60.0, 48000, 0.05)
                else:
                    h = control.play(sh, oa.SpatialOptions(pos, vel, radius))
                    r = ref_scene.play(so, oc.SpatialOptions(pos, vel, radius))
                live.append([h, r, hc, rc])
            elif op == "motion" and live:
                for _ in range(int(rng.integers(1, 4))):      # several per callback: the latest must win
                    k = int(rng.integers(0, len(live)))
                    pos, vel, disc = _vec(rng, 12.0), _vec(rng, 25.0), bool(rng.random() < 0.3)
                    live[k][0].set_motion(pos, vel, disc)
                    live[k][1].set_motion(pos, vel, disc)
            elif op == "rotation":
                q = rng.normal(size=4).astype(np.float32)
                q = (q / np.linalg.norm(q)).astype(np.float32)
                control.set_listener_rotation(q)
                ref_scene.set_listener_rotation(q)
            elif op == "control":
                cands = [e for e in live if e[2]]
                if cands:
                    e = cands[int(rng.integers(0, len(cands)))]
                    i = int(rng.integers(0, len(e[2])))
                    if isinstance(e[2][i], oa.GainControl):
                        v = float(rng.uniform(0.0, 2.0))
                        e[2][i].set_amplitude_ratio(v)
                        e[3][i].set_amplitude_ratio(v)
                    else:
                        v = float(rng.uniform(0.5, 1.6))
                        e[2][i].set_speed(v)
                        e[3][i].set_speed(v)
            elif op == "drop" and live:
                k = int(rng.integers(0, len(live)))
                if live[k][1].is_finished():                 # (the oracle's answer: the device may be callbacks behind)
                    live[k][0].release()
                    live.pop(k)
        n = int(rng.choice(FRAME_CHOICES))
        wants.append(ref.sample_n(INTERVAL, n))
        sizes.append(n)
        scene.sample_device(INTERVAL, dev_out[cb].data_ptr(), n)
        if not exact and cb % 3 == 2:
            scene.synchronize()
        peak_len = max(peak_len, len(ref_scene) + ref_scene.len_buffered())
        removed_seen = removed_seen or any(e[1].is_finished() for e in live)
    scene.synchronize()
    got = dev_out.cpu().numpy()
    for cb in range(60):
        np.testing.assert_array_equal(got[cb, :sizes[cb]], wants[cb], err_msg=f"seed {seed} callback {cb} n {sizes[cb]}")
    assert (len(scene), scene.len_buffered()) == (len(ref_scene), ref_scene.len_buffered())
    assert [e[0].is_finished() for e in live] == [e[1].is_finished() for e in live]
    assert clip_no >= 15 and peak_len >= 5 and removed_seen, (clip_no, peak_len, removed_seen)
    scene.close()


@pytest.mark.parametrize("seed", range(8))
def test_mixer_random_operations_bit_exact(seed):
    # Mixer<[f32;2]> (src/mixer.rs:92-119): plays of mono / stereo clips, Cycle, Constant under
    # FixedGain / Gain / Speed chains, stops, control changes, ragged callbacks; the mixer starts on
    # its fast kernels and switches to the general path at the first source that needs it.
    import oddio_amd as oa
    rng = np.random.default_rng(7000 + seed)
    control, mixer = oa.Mixer(max_sources=64, max_frames=2048)
    mixer.set_mode(oa.MODE_ORDERED)
    mixer.set_postfx(seed % 2)
    cm = oc.Mixer(2)
    ref = oc.Reinhard(cm) if seed % 2 else cm
    live, clip_no, stops = [], 0, 0
    for cb in range(50):
        for _ in range(int(rng.integers(0, 4))):
            op = rng.choice(["play", "play", "stop", "control"])
            if op == "play" and len(live) < 48:
                clip_no += 1
                simple = cb < 6 and seed % 2 == 0        # keep the fast path alive for a while on even seeds
                kind = rng.choice(["mono", "mono", "constant"] if simple else ["mono", "stereo", "cycle", "constant"])
                rate = int(rng.choice([48000, 44100, 16000]))
                hc, rc = [], []
                if kind == "mono":
                    clip = synth.noise_clip(seed, clip_no, int(rng.integers(1, 12000)))
                    start = float(rng.uniform(-0.01, 0.01))
                    sh, so = oa.FramesSignal(oa.Frames.from_slice(rate, clip), start), oc.FramesSignal(oc.Frames(rate, clip), start)
                elif kind == "stereo":
                    n = int(rng.integers(1, 9000))
                    clip = np.stack([synth.noise_clip(seed, clip_no, n), synth.noise_clip(seed + 50, clip_no, n)], axis=1)
                    sh, so = oa.FramesSignal(oa.Frames.from_slice(rate, clip), 0.0), oc.FramesSignal(oc.Frames(rate, clip), 0.0)
                elif kind == "cycle":
                    clip = synth.noise_clip(seed, clip_no, int(rng.integers(1, 900)))
                    sh, so = oa.Cycle(oa.Frames.from_slice(rate, clip)), oc.Cycle(oc.Frames(rate, clip))
                else:
                    val = float(rng.uniform(-0.5, 0.5))
                    sh, so = oa.Constant(val), oc.Constant(val)
                if simple:
                    if kind == "mono" and rng.random() < 0.5:
                        db = float(rng.uniform(-10, 4))
                        sh, so = oa.FixedGain(sh, db), oc.FixedGain(so, db)
                    sh, so = oa.MonoToStereo(sh), oc.MonoToStereo(so)
                else:
                    n_filters = int(rng.integers(0, 4))
                    lift_at = -1 if kind == "stereo" else int(rng.integers(0, n_filters + 1))   # where MonoToStereo sits in the nest
                    for f in range(n_filters + 1):
                        if f == lift_at:
                            sh, so = oa.MonoToStereo(sh), oc.MonoToStereo(so)
                        if f == n_filters:
                            break
                        which = rng.choice(["fixed", "gain", "speed"])
                        if which == "fixed":
                            db = float(rng.uniform(-10, 4))
                            sh, so = oa.FixedGain(sh, db), oc.FixedGain(so, db)
                        elif which == "gain":
                            c, sh = oa.Gain.new(sh)
                            so = oc.Gain(so)
                            hc.append(c); rc.append(so)
                        else:
                            c, sh = oa.Speed.new(sh)
                            so = oc.Speed(so)
                            hc.append(c); rc.append(so)
                live.append([control.play(sh), cm.play(so), hc, rc])
            elif op == "stop" and live:
                k = int(rng.integers(0, len(live)))
                live[k][0].stop(); live[k][1].stop()
                stops += 1
            elif op == "control":
                cands = [e for e in live if e[2]]
                if cands:
                    e = cands[int(rng.integers(0, len(cands)))]
                    i = int(rng.integers(0, len(e[2])))
                    if isinstance(e[2][i], oa.GainControl):
                        v = float(rng.uniform(0.0, 2.0))
                        e[2][i].set_amplitude_ratio(v); e[3][i].set_amplitude_ratio(v)
                    else:
                        v = float(rng.uniform(0.5, 1.6))
                        e[2][i].set_speed(v); e[3][i].set_speed(v)
        n = int(rng.choice([1024, 1024, 2048, 512, 1, 700, 1500]))
        a = ref.sample_n(INTERVAL, n)
        b = mixer.sample_n(INTERVAL, n)
        np.testing.assert_array_equal(b, a, err_msg=f"seed {seed} callback {cb} n {n}")
        assert len(mixer) == len(cm)
        assert [e[0].is_stopped() for e in live] == [e[1].is_stopped() for e in live]
    assert clip_no >= 15 and stops >= 3
    mixer.close()


def test_control_thread_races_with_audio_thread():
    # SpatialSceneControl / Spatial handles are Send and used from another thread while the audio thread
    # renders (src/spatial.rs:267-350, src/set.rs:125-126): hammer the C ABI's control calls concurrently
    # with sample(); nothing may fail, output stays finite, and every source played is accounted for.
    import threading
    import oddio_amd as oa
    control, scene = oa.SpatialScene(max_sources=2048, max_frames=1024)
    clips = [oa.Frames.from_slice(48000, synth.noise_clip(5, i, 4000 + 500 * i)) for i in range(8)]
    stop = threading.Event()
    errors, played = [], []

    def controller(seed):
        rng = np.random.default_rng(seed)
        mine = []
        try:
            while not stop.is_set():
                op = rng.integers(0, 4)
                if op == 0 and len(mine) < 400:
                    h = control.play(oa.FramesSignal(clips[int(rng.integers(0, 8))], float(rng.uniform(0, 0.02))),
                                     oa.SpatialOptions(_vec(rng, 10.0), _vec(rng, 20.0), 0.1))
                    mine.append(h)
                elif op == 1 and mine:
                    mine[int(rng.integers(0, len(mine)))].set_motion(_vec(rng, 10.0), _vec(rng, 20.0), bool(rng.random() < 0.2))
                elif op == 2:
                    q = rng.normal(size=4).astype(np.float32)
                    control.set_listener_rotation((q / np.linalg.norm(q)).astype(np.float32))
                elif mine:
                    mine[int(rng.integers(0, len(mine)))].is_finished()
        except Exception as e:     # noqa: BLE001
            errors.append(e)
        played.append(mine)

    threads = [threading.Thread(target=controller, args=(s,)) for s in (1, 2, 3)]
    for t in threads:
        t.start()
    try:
        for cb in range(300):
            out = scene.sample_n(INTERVAL, int((1024, 256, 700)[cb % 3]))
            assert np.isfinite(out).all()
    finally:
        stop.set()
        for t in threads:
            t.join()
    assert not errors, errors
    handles = [h for mine in played for h in mine]
    assert len(handles) > 50
    for _ in range(40):                      # clips are <= 0.16 s: everything finishes and is removed
        scene.sample_n(INTERVAL, 1024)
    assert len(scene) == 0
    assert all(h.is_finished() for h in handles)
    scene.close()
