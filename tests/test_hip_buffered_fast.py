"""The batched path of the buffered set (oddio_amd/csrc/buffered_fast.h: buffered_walk + buffered_write +
spatial_mix<.., RING>) against the CPU oracle, through the C ABI.  GPU only.

What it must reproduce: SpatialSceneControl::play_buffered (src/spatial.rs:314-340), the buffered half of
SpatialScene::sample (:395-433), Ring (src/ring.rs:4-80) incl. the write that wraps and the read cursor's rewrite at
the ring's end, Gain + Smoothed (src/gain.rs:58-127, src/smooth.rs), Speed (src/speed.rs:26-40), FixedGain
(src/gain.rs:9-51) over FramesSignal (src/frames.rs:176-201, both branches).  ORDERED mode: bit-exact.  FAST mode: the
north-star tolerance, 1e-5 of max|ref|, written below."""
import numpy as np
import pytest

from oddio_amd import synth
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

INTERVAL = np.float32(1.0) / np.float32(48000)
FAST_TOL = 1e-5      # relative to max|ref| (BASELINE.json north_star)


def opts(mod, p, v, r=0.1):
    return mod.SpatialOptions(np.asarray(p, np.float32), np.asarray(v, np.float32), r)


def init_gain(g, ratio):
    """Gain::set_amplitude_ratio before the signal is played (src/gain.rs:90-93): no ramp"""
    if hasattr(g, "init_amplitude_ratio"):
        g.init_amplitude_ratio(ratio)       # the oracle's Gain
    else:
        g.set_amplitude_ratio(ratio)        # oddio_amd: a control that is not bound yet sets the filter's initial value


def build_chain(mod, shape, clip, clip_rate, start, speed, gain, db):
    """-> (signal, gain control or None, speed control or None) for one of the chain shapes"""
    leaf = mod.FramesSignal(mod.Frames.from_slice(clip_rate, clip) if mod is not oc else mod.Frames(clip_rate, clip), start)
    gc = sc = None
    if shape == "plain":
        sig = leaf
    elif shape == "gain":
        if mod is oc:
            sig = oc.Gain(leaf); gc = sig
        else:
            gc, sig = mod.Gain.new(leaf)
        init_gain(gc, gain)
    elif shape == "gain_speed":          # Gain<Speed<FramesSignal>>: the bench's shape
        if mod is oc:
            sp = oc.Speed(leaf); sc = sp
            sig = oc.Gain(sp); gc = sig
        else:
            sc, sp = mod.Speed.new(leaf)
            gc, sig = mod.Gain.new(sp)
        sc.set_speed(speed); init_gain(gc, gain)
    elif shape == "speed_gain_fixed":    # Speed<Gain<FixedGain<FramesSignal>>>
        fg = mod.FixedGain(leaf, db)
        if mod is oc:
            g = oc.Gain(fg); gc = g
            sig = oc.Speed(g); sc = sig
        else:
            gc, g = mod.Gain.new(fg)
            sc, sig = mod.Speed.new(g)
        sc.set_speed(speed); init_gain(gc, gain)
    elif shape == "two_gains":           # Gain<Gain<FramesSignal>>: two ramps at once
        if mod is oc:
            g1 = oc.Gain(leaf)
            sig = oc.Gain(g1); gc = sig
            init_gain(g1, gain * 0.5)
        else:
            c1, g1 = mod.Gain.new(leaf)
            gc, sig = mod.Gain.new(g1)
            init_gain(c1, gain * 0.5)
        sc = None
    else:
        raise ValueError(shape)
    return sig, gc, sc


SHAPES = ("plain", "gain", "gain_speed", "speed_gain_fixed", "two_gains")


def populate(n_src, seed, *, max_distance, buffer_duration, cube, vmax, clip_len=40000, rates=(48000, 44100, 48000, 32000), shapes=SHAPES,
             mode=None, max_frames=1024, speed_span=0.1):
    import oddio_amd as oa
    control, scene = oa.SpatialScene(max_sources=max(64, n_src), max_frames=max_frames)
    scene.reserve_buffered(max(256, n_src))
    if mode is not None:
        scene.set_mode(mode)
    ref = oc.SpatialScene()
    sc = synth.make_scene(seed, n_src, cube=cube, vmax=vmax)
    rng = np.random.default_rng(seed)
    ctl = []
    for i in range(n_src):
        shape = shapes[i % len(shapes)]
        clip_rate = rates[(i // len(shapes)) % len(rates)]
        clip = synth.noise_clip(seed, i, clip_len)
        start = float(rng.uniform(-0.01, 0.05))
        speed = np.float32(1.0 + rng.uniform(-speed_span, speed_span))
        gain = np.float32(rng.uniform(0.3, 1.5))
        db = float(rng.uniform(-9.0, 3.0))
        sh, gh, sph = build_chain(oa, shape, clip, clip_rate, start, speed, gain, db)
        so, go, spo = build_chain(oc, shape, clip, clip_rate, start, speed, gain, db)
        control.play_buffered(sh, opts(oa, sc["position"][i], sc["velocity"][i]), max_distance, 48000, buffer_duration)
        ref.play_buffered(so, opts(oc, sc["position"][i], sc["velocity"][i]), max_distance, 48000, buffer_duration)
        ctl.append(((gh, go), (sph, spo)))
    return oa, control, scene, ref, ctl, rng


def test_every_fast_shape_bit_exact_over_ring_wraps():
    """Rings of ~3 900 samples: the write wraps every fourth callback, the read cursors pass the ring's end as often; gain
    and speed stores land mid-ramp; ORDERED mode (one wave walks the set): bit-exact."""
    import oddio_amd as oa
    oa_, control, scene, ref, ctl, rng = populate(60, 5, max_distance=20.0, buffer_duration=0.02, cube=12.0, vmax=15.0, mode=oa.MODE_ORDERED)
    for cb in range(26):
        if cb in (2, 3, 7, 12, 13, 20):
            for k, ((gh, go), (sph, spo)) in enumerate(ctl):
                if gh is not None and (k + cb) % 2 == 0:
                    g = np.float32(rng.uniform(0.0, 2.0))
                    gh.set_amplitude_ratio(g); go.set_amplitude_ratio(g)
                if sph is not None and (k + cb) % 3 == 0:
                    sp = np.float32(rng.uniform(0.85, 1.15))
                    sph.set_speed(sp); spo.set_speed(sp)
        a, b = ref.sample_n(INTERVAL, 1024), scene.sample_n(INTERVAL, 1024)
        assert scene.debug_buffered_slow() == 0, f"callback {cb}: every source should take the batched path"
        np.testing.assert_array_equal(b, a, err_msg=f"callback {cb}")
    assert np.abs(a).max() > 0
    scene.close()


def test_ragged_callbacks_and_unit_speed_fast_branch():
    """Callback sizes that are not 1024 (the ring write is then not 64 x 16 frames; chunks of fewer than 256 frames), and
    sources whose resample ratio is exactly 1 (frames.rs:180-187's constant-fract branch in the leaf, padded windows)."""
    import oddio_amd as oa
    oa_, control, scene, ref, ctl, rng = populate(40, 9, max_distance=40.0, buffer_duration=0.05, cube=15.0, vmax=1.0, rates=(48000,),
                                                  shapes=("plain", "gain", "gain_speed"), mode=oa.MODE_ORDERED, speed_span=0.0)
    for cb, n in enumerate((1024, 700, 256, 1, 1000, 513, 1024, 64, 1024)):
        if cb == 3:
            for (gh, go), _ in ctl:
                if gh is not None:
                    gh.set_amplitude_ratio(0.5); go.set_amplitude_ratio(0.5)
        a, b = ref.sample_n(INTERVAL, n), scene.sample_n(INTERVAL, n)
        assert scene.debug_buffered_slow() == 0
        np.testing.assert_array_equal(b, a, err_msg=f"callback {cb} ({n} frames)")
    scene.close()


def test_fast_equals_general_kernel_and_other_shapes_ride_along():
    """Cycle leaves over loops shorter than a callback's window (and Stream leaves, Faders) keep the general kernel -- their slab
    rows are added at their place in the walk -- while a Constant leaf rides the batched path since round 5; the same scene rendered
    with the batched path switched off gives the same bits."""
    import oddio_amd as oa
    outs = []
    for fast in (True, False):
        oa_, control, scene, ref, ctl, rng = populate(24, 13, max_distance=30.0, buffer_duration=0.03, cube=10.0, vmax=8.0, mode=oa.MODE_ORDERED)
        scene.set_buffered_fast(fast)
        extra = [
            (lambda m: (m.Gain(m.Constant(0.5)) if m is oc else m.Gain.new(m.Constant(0.5))[1]), [1.0, 2.0, 2.0]),
            (lambda m: (m.Speed(m.Cycle(m.Frames(32000, synth.noise_clip(31, 0, 300)))) if m is oc      # (a loop shorter than a callback's window)
                        else m.Speed.new(m.Cycle(m.Frames.from_slice(32000, synth.noise_clip(31, 0, 300))))[1]), [6.0, 1.0, 2.0]),
        ]
        for mk, p in extra:
            control.play_buffered(mk(oa), opts(oa, p, [0.5, 0.0, -1.0]), 30.0, 48000, 0.03)
            ref.play_buffered(mk(oc), opts(oc, p, [0.5, 0.0, -1.0]), 30.0, 48000, 0.03)
        res = []
        for cb in range(9):
            a, b = ref.sample_n(INTERVAL, 1024), scene.sample_n(INTERVAL, 1024)
            if fast:
                assert scene.debug_buffered_slow() == 1            # the Speed<Cycle>
            np.testing.assert_array_equal(b, a, err_msg=f"fast={fast} callback {cb}")
            res.append(b)
        outs.append(np.stack(res))
        scene.close()
    np.testing.assert_array_equal(outs[0], outs[1])


def test_removal_motion_and_rotation_with_a_seek_set_beside():
    import oddio_amd as oa
    control, scene = oa.SpatialScene(max_sources=128, max_frames=1024)
    scene.set_mode(oa.MODE_ORDERED)
    ref = oc.SpatialScene()
    hb, rb = [], []
    for i in range(20):   # buffered: short clips that finish and are removed after their delay
        clip = synth.noise_clip(11, i, 3000 + 2500 * i)
        p, v = [3.0 + 2 * (i % 7), -1.0, 2.0 + 0.3 * i], [1.0, 0.5 * (i % 5), -2.0]
        gh, g_h = oa.Gain.new(oa.FramesSignal(oa.Frames.from_slice(48000, clip), 0.0))
        g_o = oc.Gain(oc.FramesSignal(oc.Frames(48000, clip), 0.0))
        hb.append(control.play_buffered(g_h, opts(oa, p, v), 120.0, 48000, 0.1))
        rb.append(ref.play_buffered(g_o, opts(oc, p, v), 120.0, 48000, 0.1))
    hs, rs = [], []
    for i in range(9):
        clip = synth.noise_clip(12, i, 26000)
        p, v = [-4.0 - i, 2.0, 1.0 + i], [5.0, -3.0, 0.5 * i]
        hs.append(control.play(oa.FramesSignal(oa.Frames.from_slice(48000, clip), 0.05), opts(oa, p, v)))
        rs.append(ref.play(oc.FramesSignal(oc.Frames(48000, clip), 0.05), opts(oc, p, v)))
    for cb in range(60):
        if cb in (2, 9):
            for h, r in ((hb[1], rb[1]), (hb[7], rb[7]), (hs[3], rs[3])):
                h.set_motion(np.float32([2.0, 2.0, 2.0]), np.float32([0.0, 1.0, 0.0]), cb == 2)
                r.set_motion(np.float32([2.0, 2.0, 2.0]), np.float32([0.0, 1.0, 0.0]), cb == 2)
        if cb == 4:
            q = np.float32([np.cos(0.4), 0.0, np.sin(0.4), 0.0])
            control.set_listener_rotation(q)
            ref.set_listener_rotation(q)
        a, b = ref.sample_n(INTERVAL, 1024), scene.sample_n(INTERVAL, 1024)
        np.testing.assert_array_equal(b, a, err_msg=f"callback {cb}")
        assert (scene.len_buffered(), len(scene)) == (ref.len_buffered(), len(ref))
        assert [h.is_finished() for h in hb + hs] == [r.is_finished() for r in rb + rs]
        if scene.len_buffered() == 0 and cb > 30:
            break
    assert scene.len_buffered() == 0
    scene.close()


def test_fast_mode_tolerance_and_getters():
    import oddio_amd as oa
    oa_, control, scene, ref, ctl, rng = populate(400, 21, max_distance=60.0, buffer_duration=0.05, cube=25.0, vmax=20.0)
    (gh, go), (sph, spo) = ctl[2]          # a gain_speed source
    gh.set_gain(-6.0); go.set_gain(-6.0)
    assert abs(gh.gain() - (-6.0)) < 1e-4 and abs(gh.amplitude_ratio() - 10.0 ** (-6.0 / 20.0)) < 1e-6
    sph.set_speed(1.05); spo.set_speed(1.05)
    assert abs(sph.speed() - 1.05) < 1e-7
    for cb in range(6):
        a, b = ref.sample_n(INTERVAL, 1024), scene.sample_n(INTERVAL, 1024)
        assert scene.debug_buffered_slow() == 0
        assert np.abs(b - a).max() <= FAST_TOL * np.abs(a).max(), f"callback {cb}"
    scene.close()


def test_ordered_rows_path_above_the_serial_threshold():
    """More than 1024 buffered sources in ORDERED mode: spatial_mix<.., STORE, .., RING> writes contribution rows and
    ordered_sum adds them in the reference's reverse walk order; a Seek set beside it is seeded with the result."""
    import oddio_amd as oa
    n_buf, n_seek = 2300, 1300
    control, scene = oa.SpatialScene(max_sources=4096, max_frames=1024)
    scene.reserve_buffered(n_buf)
    scene.set_mode(oa.MODE_ORDERED)
    ref = oc.SpatialScene()
    sc = synth.make_scene(77, n_buf + n_seek, cube=8.0, vmax=6.0)
    frames_h, clips = [], []
    bank = [synth.noise_clip(78, k, 24000) for k in range(64)]
    bank_h = [oa.Frames.from_slice(48000, c) for c in bank]
    kinds = [oa.FILTER_SPEED, oa.FILTER_GAIN]
    rng = np.random.default_rng(3)
    params = np.stack([1.0 + rng.uniform(-0.1, 0.1, n_buf), rng.uniform(0.2, 1.2, n_buf)], axis=1).astype(np.float32)
    starts = rng.uniform(0.0, 0.05, n_buf)
    ids = control.play_buffered_frames_batch([bank_h[i % 64] for i in range(n_buf)], starts, kinds, params, sc["position"][:n_buf], sc["velocity"][:n_buf],
                                             sc["radius"][:n_buf], 25.0, 48000, 0.03)
    gains_o = []
    for i in range(n_buf):
        sp = oc.Speed(oc.FramesSignal(oc.Frames(48000, bank[i % 64]), float(starts[i])))
        sp.set_speed(params[i, 0])
        g = oc.Gain(sp)
        g.init_amplitude_ratio(params[i, 1])
        gains_o.append(g)
        ref.play_buffered(g, oc.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1), 25.0, 48000, 0.03)
    for i in range(n_buf, n_buf + n_seek):
        control.play(oa.FramesSignal(bank_h[i % 64], 0.04), oa.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
        ref.play(oc.FramesSignal(oc.Frames(48000, bank[i % 64]), 0.04), oc.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1))
    for cb, n in enumerate((1024, 1024, 600, 1024, 1024)):
        if cb == 2:
            new = rng.uniform(0.0, 1.5, n_buf).astype(np.float32)
            control.set_control_batch(ids[::3], 1, new[::3])
            for i in range(0, n_buf, 3):
                gains_o[i].set_amplitude_ratio(new[i])
        a, b = ref.sample_n(INTERVAL, n), scene.sample_n(INTERVAL, n)
        assert scene.debug_buffered_slow() == 0
        np.testing.assert_array_equal(b, a, err_msg=f"callback {cb}")
    scene.close()


def test_65536_buffered_sources_ordered_bit_exact_fast_vs_f64():
    """BASELINE configs[3]'s per-GPU size with every source played through play_buffered as Gain<Speed<FramesSignal>>.
    ORDERED (contribution rows + ordered_sum): bit-exact.  FAST (tree sum over the whole chip): at this source count the
    reference's own sequential f32 sum is further than 1e-5 from the exact sum, so FAST is held to the exact
    (f64-accumulated) sum of the reference's contributions, and to the reference within the two sums' own distances."""
    import oddio_amd as oa
    n_src, n_cb = 65536, 3
    fast_vs_exact_tol = 2e-6
    sc = synth.make_scene(91, n_src, cube=10.0, vmax=10.0)
    rng = np.random.default_rng(91)
    speeds = (1.0 + rng.uniform(-0.1, 0.1, n_src)).astype(np.float32)
    gains0 = rng.uniform(0.3, 1.0, n_src).astype(np.float32)
    gains1 = rng.uniform(0.0, 1.2, n_src).astype(np.float32)
    starts = rng.uniform(0.0, 0.05, n_src)
    bank = [synth.noise_clip(92, k, 12000) for k in range(256)]

    def oracle_run(acc64):
        ref = oc.SpatialScene()
        frames = [oc.Frames(48000, c) for c in bank]
        ctl = []
        for i in range(n_src):
            sp = oc.Speed(oc.FramesSignal(frames[i % 256], float(starts[i])))
            sp.set_speed(speeds[i])
            g = oc.Gain(sp)
            g.init_amplitude_ratio(gains0[i])
            ref.play_buffered(g, oc.SpatialOptions(sc["position"][i], sc["velocity"][i], 0.1), 30.0, 48000, 0.03)
            ctl.append(g)
        outs = []
        for cb in range(n_cb):
            if cb == 1:
                for i in range(0, n_src, 3):
                    ctl[i].set_amplitude_ratio(gains1[i])
            outs.append(ref.sample_f64acc(INTERVAL, 1024) if acc64 else ref.sample_n(INTERVAL, 1024))
        return outs

    ref32, ref64 = oracle_run(False), oracle_run(True)
    got = {}
    for mode in (oa.MODE_ORDERED, oa.MODE_FAST, oa.MODE_TRACKED):
        control, scene = oa.SpatialScene(max_sources=n_src, max_frames=1024)
        scene.reserve_buffered(n_src)
        scene.set_mode(mode)
        bank_h = [oa.Frames.from_slice(48000, c) for c in bank]
        ids = control.play_buffered_frames_batch([bank_h[i % 256] for i in range(n_src)], starts, [oa.FILTER_SPEED, oa.FILTER_GAIN],
                                                 np.stack([speeds, gains0], axis=1), sc["position"], sc["velocity"], sc["radius"], 30.0, 48000, 0.03)
        outs = []
        for cb in range(n_cb):
            if cb == 1:
                control.set_control_batch(ids[::3], 1, gains1[::3])
            outs.append(scene.sample_n(INTERVAL, 1024))
            assert scene.debug_buffered_slow() == 0
        got[mode] = outs
        scene.close()
    for cb in range(n_cb):
        np.testing.assert_array_equal(got[oa.MODE_ORDERED][cb], ref32[cb], err_msg=f"ORDERED callback {cb}")
        scale = float(np.abs(ref32[cb]).max())
        g64 = got[oa.MODE_FAST][cb].astype(np.float64)
        e_exact = float(np.abs(g64 - ref64[cb]).max())
        e_ref = float(np.abs(g64 - ref32[cb].astype(np.float64)).max())
        e_refexact = float(np.abs(ref32[cb].astype(np.float64) - ref64[cb]).max())
        assert scale > 0
        assert e_exact <= fast_vs_exact_tol * scale, (cb, e_exact / scale)
        assert e_ref <= max(FAST_TOL * scale, e_exact + e_refexact), (cb, e_ref / scale, e_refexact / scale)
        # TRACKED (round 5): the ring reads mixed twice, the second pass restarted at the prefix of the first one's partial sums -- the
        # reference's sequential sum with its rounding errors, where the tree sum above is as far from it as the exact sum is
        e_tracked = float(np.abs(got[oa.MODE_TRACKED][cb] - ref32[cb]).max())
        print(f"buffered, callback {cb}: |TRACKED - ref| / max|ref| = {e_tracked / scale:.2e}, |FAST - ref| = {e_ref / scale:.2e}")
        assert e_tracked <= 3e-6 * scale, (cb, e_tracked / scale, e_ref / scale)


def test_control_and_motion_updates_from_device_memory():
    """oddio_hip_scene_set_control_device / _set_motion_device: handle ids and values in device memory, one message, applied in
    message order between ordinary control calls -- against the same stores made through the oracle's controls, bit-exact.  Ids
    that are not the scene's (beyond the handle table) are ignored."""
    import torch

    import oddio_amd as oa
    dev = torch.device("cuda", 0)
    n_src = 48
    control, scene = oa.SpatialScene(max_sources=64, max_frames=1024)
    scene.reserve_buffered(256)
    scene.set_mode(oa.MODE_ORDERED)
    ref = oc.SpatialScene()
    sc = synth.make_scene(33, n_src, cube=10.0, vmax=8.0)
    sc2 = synth.make_scene(34, n_src, cube=9.0, vmax=5.0)
    rng = np.random.default_rng(33)
    hh, rh, gains = [], [], []
    for i in range(n_src):
        clip = synth.noise_clip(33, i, 40000)
        gc, g_h = oa.Gain.new(oa.FramesSignal(oa.Frames.from_slice(48000, clip), 0.01))
        g_o = oc.Gain(oc.FramesSignal(oc.Frames(48000, clip), 0.01))
        hh.append(control.play_buffered(g_h, opts(oa, sc["position"][i], sc["velocity"][i]), 30.0, 48000, 0.03))
        rh.append(ref.play_buffered(g_o, opts(oc, sc["position"][i], sc["velocity"][i]), 30.0, 48000, 0.03))
        gains.append((gc, g_o))
    ids = np.array([h.id for h in hh], dtype=np.uint32)
    keep = []
    for cb in range(9):
        if cb == 1:      # a gain store to every source from device memory (+ an id that is nobody's), then an ordinary store behind it
            vals = rng.uniform(0.1, 1.4, n_src).astype(np.float32)
            d_ids = torch.from_numpy(np.concatenate([ids, [4000000000]]).astype(np.int64)).to(dev).to(torch.int32)
            d_vals = torch.from_numpy(np.concatenate([vals, [9.0]]).astype(np.float32)).to(dev)
            keep += [d_ids, d_vals]
            torch.cuda.synchronize()   # (torch's stream converted the ids: the scene's stream does not wait for it)
            gains[5][0].set_amplitude_ratio(0.77); gains[5][1].set_amplitude_ratio(0.77)      # ahead of the batch: the batch wins
            control.set_control_device(n_src + 1, d_ids.data_ptr(), 0, d_vals.data_ptr())
            for i in range(n_src):
                gains[i][1].set_amplitude_ratio(vals[i])
            gains[0][0].set_amplitude_ratio(0.33); gains[0][1].set_amplitude_ratio(0.33)      # behind it: this one wins
        if cb == 4:      # Spatial::set_motion for every source from device memory
            d_ids = torch.from_numpy(ids.astype(np.int64)).to(dev).to(torch.int32)
            d_pos = torch.from_numpy(sc2["position"]).to(dev).contiguous()
            d_vel = torch.from_numpy(sc2["velocity"]).to(dev).contiguous()
            keep += [d_ids, d_pos, d_vel]
            torch.cuda.synchronize()
            control.set_motion_device(n_src, d_ids.data_ptr(), d_pos.data_ptr(), d_vel.data_ptr(), False)
            for i in range(n_src):
                rh[i].set_motion(sc2["position"][i], sc2["velocity"][i], False)
        a, b = ref.sample_n(INTERVAL, 1024), scene.sample_n(INTERVAL, 1024)
        np.testing.assert_array_equal(b, a, err_msg=f"callback {cb}")
    scene.close()


@pytest.mark.parametrize("with_sine", [False, True])
def test_synthesised_leaves_take_the_batched_path(with_sine):
    """Round 5: Constant, Sine and Cycle leaves under FixedGain / Gain / Speed chains are rendered by buffered_write like clip sources
    (a synthesised leaf is computed where a clip's window would be read; a Cycle's (base, offset) cursor is scanned with its rewrite
    at the clip's end) -- no source of a 1024-frame callback is left to the general kernel.  Constant and Cycle leaves are exact
    (ORDERED: bit for bit, rings wrapping inside the run included); Sine leaves go through the device's sine (sin_small, ~1e-7
    absolute): the north_star's 1e-5."""
    import oddio_amd as oa
    n_src = 150
    sc = synth.make_scene(77, n_src, cube=15.0, vmax=18.0)
    control, scene = oa.SpatialScene(max_sources=n_src + 8, max_frames=1024)
    scene.reserve_buffered(n_src)
    scene.set_mode(oa.MODE_ORDERED)
    ref = oc.SpatialScene()
    gains = []
    for i in range(n_src):
        kind = i % 3
        if kind == 0:
            clip = synth.noise_clip(77, i, 20000)
            leaf_h, leaf_o = oa.FramesSignal(oa.Frames.from_slice(48000, clip), 0.0), oc.FramesSignal(oc.Frames(48000, clip), 0.0)
        elif kind == 1 and i % 2 == 1:
            # Cycle (cycle.rs:26-53): loops of 1.3 k .. 6 k samples at 48 / 44.1 kHz -- most callbacks pass the loop's end, some twice
            # per Ring::write (the ring wraps too)
            loop = synth.noise_clip(78, i, 1300 + 37 * i)
            leaf_h, leaf_o = oa.Cycle(oa.Frames.from_slice((48000, 44100)[i % 4 == 1], loop)), oc.Cycle(oc.Frames((48000, 44100)[i % 4 == 1], loop))
        elif kind == 1 or not with_sine:
            leaf_h, leaf_o = oa.Constant(0.3 + 0.001 * i), oc.Constant(0.3 + 0.001 * i)
        else:
            leaf_h, leaf_o = oa.Sine(0.05 * i, 200.0 + 3.0 * i), oc.Sine(0.05 * i, 200.0 + 3.0 * i)
        if i % 4 == 1:
            leaf_h, leaf_o = oa.FixedGain(leaf_h, -2.0), oc.FixedGain(leaf_o, -2.0)
        gc, sig_h = oa.Gain.new(leaf_h)
        sig_o = oc.Gain(leaf_o)
        if i % 5 == 2:
            spc, sig_h = oa.Speed.new(sig_h)
            spc.set_speed(1.03)
            sig_o = oc.Speed(sig_o)
            sig_o.set_speed(1.03)
        gains.append((gc, sig_o if i % 5 != 2 else None))
        o = (sc["position"][i], sc["velocity"][i], 0.1)
        # short rings (max_distance 20 m, 0.05 s): they wrap every few callbacks -- Ring::write's two inner.sample calls
        control.play_buffered(sig_h, oa.SpatialOptions(*o), 20.0, 48000, 0.05)
        ref.play_buffered(sig_o, oc.SpatialOptions(*o), 20.0, 48000, 0.05)
    for cb in range(10):
        if cb in (2, 5):
            for k, (gc, og) in enumerate(gains):
                if og is not None and k % 3 != 2:
                    gc.set_amplitude_ratio(0.4 + 0.05 * ((k + cb) % 9)); og.set_amplitude_ratio(0.4 + 0.05 * ((k + cb) % 9))
        got, want = scene.sample_n(INTERVAL, 1024), ref.sample_n(INTERVAL, 1024)
        assert scene.debug_buffered_slow() == 0, cb
        if with_sine:
            assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), cb
        else:
            np.testing.assert_array_equal(got, want, err_msg=f"callback {cb}")
    scene.close()
