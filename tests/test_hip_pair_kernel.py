"""spatial_mix_pair (csrc/pair_kernels.h): the two-wavefront-per-source mix kernel that large FAST-mode scenes take for
callbacks of 513..1024 frames.  ODDIO_HIP_PAIR_MIN_GROUPS=1 sends small scenes through it, where what it must produce can
be written down exactly: workgroup w renders groups of 16 slots, each ear in one wavefront that adds its sources in
descending slot order from zero (spatial.rs:204,459-460); reduce_partials adds the workgroups in ascending order.  In
MODE_FAST_UNFUSED every contribution carries the reference's own roundings, so the result is bit for bit that sum of
single-source oracle renders; MODE_FAST (fused lerp / ramp / accumulate) stays within the north_star's 1e-5 of the
reference.  The scenes mix every variant the kernel has: plain staged windows, the padded near-unit layout and the
constant-fract branch (static sources, frames.rs:180-187), FixedGain, cursors that start before the clip, clips that end
inside the callback, windows larger than the stage (96 kHz clips: the exact per-lane path), Constant and Sine sources (out
of line on parked accumulators), motion updates and a listener rotation inside the run."""
import numpy as np
import pytest

import scenario  # noqa: F401  (path set-up shared with the other GPU tests)
from oddio_amd import synth
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

RATE = 48000
INTERVAL = np.float32(1.0) / np.float32(RATE)


def _sources(seed, n_src, with_sine, short_clip=5440):
    rng = np.random.default_rng(seed)
    sc = synth.make_scene(seed, n_src, cube=14.0, vmax=25.0)
    out = []
    for i in range(n_src):
        k = i % 12
        pos, vel = sc["position"][i].copy(), sc["velocity"][i].copy()
        src = {"pos": pos, "vel": vel, "radius": 0.1 if i % 3 else 0.5, "gain_db": None}
        if k == 3:
            src["kind"] = "constant"; src["value"] = float(rng.uniform(-1, 1))
        elif k == 7 and with_sine:
            src["kind"] = "sine"; src["phase"] = float(sc["phase"][i]); src["hz"] = float(sc["freq_hz"][i])
        else:
            src["kind"] = "frames"
            src["rate"] = RATE
            src["start"] = 0.06
            n = 9000
            if k == 1:
                vel[:] = 0.0                                    # ds == 1 exactly: padded layout, constant-fract branch
            elif k == 2:
                vel *= np.float32(0.02)                         # |ds - 1| < PAD_EPS: padded layout, running cursor
            elif k == 4:
                src["gain_db"] = -4.5                           # FixedGain
            elif k == 5:
                src["start"] = -0.004                           # the cursor starts before the clip
            elif k == 6:
                n = short_clip                                  # the clip ends inside the run (5440: in its third callback, so that the
                                                                # source is still in the set -- and the slots where they were -- when the run ends)
            elif k == 8:
                src["rate"] = 96000                             # window larger than the stage: exact per-lane path
            elif k == 9:
                src["rate"] = 44100
            elif k == 10:
                src["gain_db"] = 3.0; vel[:] = 0.0
            elif k == 11:
                src["reinhard"] = True                          # per-source Reinhard around FixedGain around the clip (reinhard.rs:22-50)
                src["gain_db"] = 6.0
            elif k == 0 and i % 24 == 12:
                src["reinhard"] = True; vel[:] = 0.0            # ... on the constant-fract branch
            elif k == 0 and with_sine:
                src["tanh"] = True                              # per-source Tanh (tanh.rs:22-29): tanh_fast in the fused kernel (FAST-mode test only:
                src["gain_db"] = 12.0 if i % 48 == 0 else -38.0  # the device's tanh is not libm's bit for bit); loud and very quiet
            src["clip"] = synth.noise_clip(seed, i, n)
        out.append(src)
    return out


def _oracle_signal(src):
    if src["kind"] == "frames":
        sig = oc.FramesSignal(oc.Frames(src["rate"], src["clip"]), src["start"])
    elif src["kind"] == "sine":
        sig = oc.Sine(src["phase"], src["hz"])
    else:
        sig = oc.Constant(src["value"])
    if src["gain_db"] is not None:
        sig = oc.FixedGain(sig, src["gain_db"])
    if src.get("reinhard"):
        sig = oc.Reinhard(sig)
    if src.get("tanh"):
        sig = oc.Tanh(sig)
    return sig


def _hip_signal(oa, src):
    if src["kind"] == "frames":
        sig = oa.FramesSignal(oa.Frames.from_slice(src["rate"], src["clip"]), src["start"])
    elif src["kind"] == "sine":
        sig = oa.Sine(src["phase"], src["hz"])
    else:
        sig = oa.Constant(src["value"])
    if src["gain_db"] is not None:
        sig = oa.FixedGain(sig, src["gain_db"])
    if src.get("reinhard"):
        sig = oa.Reinhard(sig)
    if src.get("tanh"):
        sig = oa.Tanh(sig)
    return sig


def _events(sources, cb, handles, scene_like, rng_seed):
    """The same control updates for every backend: a listener rotation before callback 1, new motions before callback 2."""
    if cb == 1:
        q = np.array([0.9950042, 0.0, 0.0998334, 0.0], np.float32)
        scene_like.set_listener_rotation(q)
    if cb == 2:
        rng = np.random.default_rng(rng_seed)
        for j in range(0, len(sources), 5):
            p = (sources[j]["pos"] + rng.uniform(-0.5, 0.5, 3)).astype(np.float32)
            v = rng.uniform(-20, 20, 3).astype(np.float32)
            disc = bool(j % 10 == 0)
            if handles[j] is not None:
                handles[j].set_motion(p, v, disc)


def _render_hip(monkeypatch, mode, sources, n_frames, n_cb, pair):
    import oddio_amd as oa
    monkeypatch.setenv("ODDIO_HIP_PAIR", "1" if pair else "0")
    monkeypatch.setenv("ODDIO_HIP_PAIR_MIN_GROUPS", "1")
    control, scene = oa.SpatialScene(max_sources=256, max_frames=1024)
    scene.set_mode(mode)
    handles = [control.play(_hip_signal(oa, s), oa.SpatialOptions(s["pos"], s["vel"], s["radius"])) for s in sources]
    outs = []
    for cb in range(n_cb):
        _events(sources, cb, handles, control, 77)
        outs.append(scene.sample_n(INTERVAL, n_frames).copy())
    n_live = len(scene)
    scene.close()
    return outs, n_live


def _render_oracle(sources, n_frames, n_cb, only=None):
    scene = oc.SpatialScene()
    handles = []
    for j, s in enumerate(sources):
        if only is None or j == only:
            handles.append(scene.play(_oracle_signal(s), oc.SpatialOptions(s["pos"], s["vel"], s["radius"])))
        else:
            handles.append(None)
    outs = []
    for cb in range(n_cb):
        _events(sources, cb, handles, scene, 77)
        outs.append(scene.sample_n(INTERVAL, n_frames).copy())
    return outs, len(scene)


@pytest.mark.parametrize("n_src,n_frames", [(100, 1024), (57, 700), (16, 1024), (9, 1024), (57, 768), (40, 960)])   # (768 / 960: the LANE16 instantiations)
def test_pair_kernel_unfused_is_the_sum_of_exact_contributions(monkeypatch, n_src, n_frames):
    import oddio_amd as oa
    n_cb = 4
    sources = _sources(300 + n_src, n_src, with_sine=False)
    contrib = [_render_oracle(sources, n_frames, n_cb, only=j)[0] for j in range(n_src)]
    got, n_live = _render_hip(monkeypatch, oa.MODE_FAST_UNFUSED, sources, n_frames, n_cb, pair=True)
    ref, ref_live = _render_oracle(sources, n_frames, n_cb)
    assert n_live == ref_live
    n_groups = (n_src + 15) // 16
    for cb in range(n_cb):
        want = None
        for w in range(n_groups):                                  # one workgroup per group of 16 slots (PAIR_GROUP)
            acc = np.zeros((n_frames, 2), dtype=np.float32)
            for i in range(min(16 * w + 16, n_src) - 1, 16 * w - 1, -1):
                acc = acc + contrib[i][cb]
            want = acc if want is None else want + acc             # reduce_partials: workgroups in ascending order
        np.testing.assert_array_equal(got[cb], want, err_msg=f"callback {cb}")
        scale = np.abs(ref[cb]).max()
        assert np.abs(got[cb] - ref[cb]).max() <= 1e-5 * scale


@pytest.mark.parametrize("n_src,n_frames", [(100, 1024), (41, 513)])
def test_pair_kernel_fast_mode_within_tolerance_and_close_to_spatial_mix(monkeypatch, n_src, n_frames):
    import oddio_amd as oa
    n_cb = 4
    sources = _sources(500 + n_src, n_src, with_sine=True, short_clip=1800)      # (finished from the start: removed during the run)
    ref, ref_live = _render_oracle(sources, n_frames, n_cb)
    pair, n_live = _render_hip(monkeypatch, oa.MODE_FAST, sources, n_frames, n_cb, pair=True)
    tile, n_live_t = _render_hip(monkeypatch, oa.MODE_FAST, sources, n_frames, n_cb, pair=False)
    assert n_live == ref_live == n_live_t
    for cb in range(n_cb):
        scale = np.abs(ref[cb]).max()
        assert np.abs(pair[cb] - ref[cb]).max() <= 1e-5 * scale, cb      # the north_star's tolerance
        assert np.abs(pair[cb] - tile[cb]).max() <= 2e-6 * scale, cb     # the two kernels differ by their sum trees only


@pytest.mark.parametrize("mode_name,n_src,n_frames", [("FAST", 100, 1024), ("FAST_UNFUSED", 57, 700), ("TRACKED", 100, 1024)])
def test_walk_inside_the_mix_kernel_leaves_the_same_bits(monkeypatch, mode_name, n_src, n_frames):
    """ODDIO_HIP_FUSED_WALK=1 (pair_kernels.h WALK): the set walk at the top of spatial_mix_pair instead of a spatial_prepass launch --
    the same records from the same code, so the same output bits, through a listener rotation, motion updates and sources that finish
    and are removed (the walk's other products: clocks, finished flags, the stopped list the reduce compacts from)."""
    import oddio_amd as oa
    n_cb = 5
    sources = _sources(900 + n_src, n_src, with_sine=mode_name == "FAST", short_clip=1800)
    monkeypatch.setenv("ODDIO_HIP_FUSED_WALK", "0")
    plain, live0 = _render_hip(monkeypatch, getattr(oa, "MODE_" + mode_name), sources, n_frames, n_cb, pair=True)
    monkeypatch.setenv("ODDIO_HIP_FUSED_WALK", "1")
    fused, live1 = _render_hip(monkeypatch, getattr(oa, "MODE_" + mode_name), sources, n_frames, n_cb, pair=True)
    assert live0 == live1 < n_src            # sources were removed on the way
    for cb in range(n_cb):
        np.testing.assert_array_equal(fused[cb], plain[cb], err_msg=f"callback {cb}")


@pytest.mark.parametrize("mode_name,n_src,n_frames", [("FAST", 100, 768), ("FAST", 41, 528), ("FAST_UNFUSED", 57, 960), ("TRACKED", 100, 640),
                                                      ("TRACKED", 70, 1008)])
def test_lane_granular_callbacks_leave_the_same_bits(monkeypatch, mode_name, n_src, n_frames):
    """Callbacks of 16 k < 1024 frames take spatial_mix_pair<.., LANE16>: the full-callback loop on the lanes whose 16 frames lie inside
    the callback, the other lanes sitting the sources out (pair_kernels.h).  The same operations on the same values as the ragged
    instantiations (ODDIO_HIP_LANE16=0), so the same bits -- through every source variant of the kernel, a listener rotation, motion
    updates and sources that finish inside the run; and, FAST mode, within the north_star's 1e-5 of the reference."""
    import oddio_amd as oa
    n_cb = 5
    sources = _sources(1300 + n_src, n_src, with_sine=mode_name == "FAST", short_clip=1800)
    monkeypatch.setenv("ODDIO_HIP_LANE16", "0")
    ragged, live0 = _render_hip(monkeypatch, getattr(oa, "MODE_" + mode_name), sources, n_frames, n_cb, pair=True)
    monkeypatch.setenv("ODDIO_HIP_LANE16", "1")
    lane16, live1 = _render_hip(monkeypatch, getattr(oa, "MODE_" + mode_name), sources, n_frames, n_cb, pair=True)
    ref, ref_live = _render_oracle(sources, n_frames, n_cb)
    assert live0 == live1 == ref_live < n_src
    for cb in range(n_cb):
        np.testing.assert_array_equal(lane16[cb], ragged[cb], err_msg=f"callback {cb}")
        assert np.abs(lane16[cb] - ref[cb]).max() <= 1e-5 * np.abs(ref[cb]).max(), cb
