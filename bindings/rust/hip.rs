// bindings/rust/hip.rs -- the module a maintainer adds to the oddio crate as `src/hip.rs`
// (behind a cargo feature, e.g. `hip`) so that the SpatialScene / Mixer hot path runs on an MI355X
// through libodd_hip.so.  build.rs: println!("cargo:rustc-link-lib=dylib=odd_hip");
//
// NOT COMPILED in the build image (no cargo/rustc there).  tests/test_rust_binding.py checks every
// `extern "C"` declaration below against include/oddio_hip.h (name, arity, argument types), and
// everything these functions call is exercised through the same C ABI by the ctypes binding
// (oddio_amd/_lib.py) in `pytest -m gpu`.
//
// What this gives the crate: `HipSpatialScene: Signal<Frame = [Sample; 2]>` and
// `HipMixer: Signal<Frame = [Sample; 2]>`, so `oddio::run(&mut scene, rate, out)` (src/lib.rs:90-93),
// `Reinhard::new(scene)`, `Adapt::new(scene, ..)` and any parent Mixer work unchanged.
//
// What cannot be redirected: `SpatialSceneControl::play<S: Seek>` (src/spatial.rs:289-302) and
// `MixerControl::play<S: Signal>` (src/mixer.rs:18-26) are open-ended -- any user type implementing
// the traits can be played (examples/realtime.rs plays such closures-as-signals).  A device path
// can only take signals it has a kernel for, so the HIP controls expose one method per supported
// signal shape (`play_frames`, `play_sine`, `play_buffered(leaf, filters)`, ...).  A user-defined
// `Signal` has to stay on the CPU scene; mix the two scenes' outputs with a parent `Mixer`.
use crate::{Frames, Sample, Signal, SpatialOptions};
use std::{
    ffi::CStr,
    os::raw::{c_char, c_int, c_void},
    sync::Arc,
};

#[repr(C)]
pub struct RawFrames {
    _p: [u8; 0],
}
#[repr(C)]
pub struct RawScene {
    _p: [u8; 0],
}
#[repr(C)]
pub struct RawMixer {
    _p: [u8; 0],
}
/// `oddio_hip_filter`: one wrapper around a leaf, innermost first.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct RawFilter {
    pub kind: c_int, // 1 FixedGain(dB), 2 Gain(initial amplitude ratio), 3 Speed(initial factor), 4 Reinhard, 5 Tanh (per-source soft clips)
    pub param: f32,
}
pub const LEAF_FRAMES: c_int = 0;
pub const LEAF_SINE: c_int = 1;
pub const LEAF_CONSTANT: c_int = 2;
pub const LEAF_CYCLE: c_int = 3;
pub const LEAF_DOWNMIX: c_int = 4; // Downmix::new(FramesSignal::new(stereo frames, ..)): oddio_hip_scene_play_filtered only
pub const POSTFX_NONE: c_int = 0;
pub const POSTFX_REINHARD: c_int = 1;
pub const POSTFX_TANH: c_int = 2;

extern "C" {
    fn oddio_hip_last_error() -> *const c_char;
    fn oddio_hip_frames_from_slice(device: c_int, rate: u32, samples: *const f32, len: usize, out: *mut *mut RawFrames) -> c_int;
    fn oddio_hip_frames_from_slice_stereo(device: c_int, rate: u32, interleaved: *const f32, n_frames: usize, out: *mut *mut RawFrames) -> c_int;
    fn oddio_hip_frames_release(f: *mut RawFrames) -> c_int;
    fn oddio_hip_scene_create(device: c_int, max_sources: u32, max_frames: u32, out: *mut *mut RawScene) -> c_int;
    fn oddio_hip_scene_destroy(s: *mut RawScene) -> c_int;
    fn oddio_hip_scene_play_frames(s: *mut RawScene, f: *mut RawFrames, start_seconds: f64, fixed_gain_db: f32, position: *const f32, velocity: *const f32, radius: f32, id: *mut u32) -> c_int;
    fn oddio_hip_scene_play_sine(s: *mut RawScene, phase: f32, frequency_hz: f32, fixed_gain_db: f32, position: *const f32, velocity: *const f32, radius: f32, id: *mut u32) -> c_int;
    fn oddio_hip_scene_play_constant(s: *mut RawScene, value: f32, position: *const f32, velocity: *const f32, radius: f32, id: *mut u32) -> c_int;
    fn oddio_hip_scene_play_cycle(s: *mut RawScene, f: *mut RawFrames, fixed_gain_db: f32, position: *const f32, velocity: *const f32, radius: f32, id: *mut u32) -> c_int;
    fn oddio_hip_scene_play_buffered(s: *mut RawScene, leaf_kind: c_int, f: *mut RawFrames, start_seconds: f64, phase: f32, frequency_hz_or_value: f32, filters: *const RawFilter, n_filters: c_int, position: *const f32, velocity: *const f32, radius: f32, max_distance: f32, rate: u32, buffer_duration: f32, id: *mut u32) -> c_int;
    fn oddio_hip_scene_play_filtered(s: *mut RawScene, leaf_kind: c_int, f: *mut RawFrames, start_seconds: f64, phase: f32, frequency_hz_or_value: f32, filters: *const RawFilter, n_filters: c_int, position: *const f32, velocity: *const f32, radius: f32, id: *mut u32) -> c_int;
    fn oddio_hip_source_set_gain(s: *mut RawScene, id: u32, filter_index: c_int, amplitude_ratio: f32) -> c_int;
    fn oddio_hip_source_set_gain_db(s: *mut RawScene, id: u32, filter_index: c_int, db: f32) -> c_int;
    fn oddio_hip_source_set_speed(s: *mut RawScene, id: u32, filter_index: c_int, factor: f32) -> c_int;
    fn oddio_hip_source_get_amplitude_ratio(s: *mut RawScene, id: u32, filter_index: c_int, amplitude_ratio: *mut f32) -> c_int;
    fn oddio_hip_source_get_gain_db(s: *mut RawScene, id: u32, filter_index: c_int, db: *mut f32) -> c_int;
    fn oddio_hip_source_get_speed(s: *mut RawScene, id: u32, filter_index: c_int, factor: *mut f32) -> c_int;
    fn oddio_hip_scene_reserve_buffered(s: *mut RawScene, max_buffered: u32) -> c_int;
    fn oddio_hip_scene_play_buffered_batch(s: *mut RawScene, n: usize, frames: *const *mut RawFrames, start_seconds: *const f64, filter_kinds: *const c_int, n_filters: c_int, filter_params: *const f32, positions: *const f32, velocities: *const f32, radii: *const f32, max_distance: f32, rate: u32, buffer_duration: f32, ids: *mut u32) -> c_int;
    fn oddio_hip_scene_set_control_batch(s: *mut RawScene, n: usize, ids: *const u32, filter_index: c_int, values: *const f32) -> c_int;
    fn oddio_hip_source_set_motion(s: *mut RawScene, id: u32, position: *const f32, velocity: *const f32, discontinuity: c_int) -> c_int;
    fn oddio_hip_source_is_finished(s: *mut RawScene, id: u32, finished: *mut c_int) -> c_int;
    fn oddio_hip_source_release(s: *mut RawScene, id: u32) -> c_int;
    fn oddio_hip_scene_set_listener_rotation(s: *mut RawScene, rotation_sxyz: *const f32) -> c_int;
    fn oddio_hip_scene_set_postfx(s: *mut RawScene, postfx: c_int) -> c_int;
    fn oddio_hip_scene_set_mode(s: *mut RawScene, mode: c_int) -> c_int;
    fn oddio_hip_scene_sample(s: *mut RawScene, interval: f32, out: *mut f32, n_frames: usize) -> c_int;
    fn oddio_hip_scene_reduce_init(s: *mut RawScene, rank: c_int, world: c_int, unique_id: *const c_void, unique_id_bytes: usize) -> c_int;
    fn oddio_hip_scene_reduce_init_p2p(s: *mut RawScene, rank: c_int, world: c_int, handle: *mut c_void, handle_bytes: usize) -> c_int;
    fn oddio_hip_mixer_create(device: c_int, max_sources: u32, max_frames: u32, out: *mut *mut RawMixer) -> c_int;
    fn oddio_hip_mixer_create_mono(device: c_int, max_sources: u32, max_frames: u32, out: *mut *mut RawMixer) -> c_int;
    fn oddio_hip_mixer_destroy(m: *mut RawMixer) -> c_int;
    fn oddio_hip_mixer_play_sine(m: *mut RawMixer, phase: f32, frequency_hz: f32, fixed_gain_db: f32, id: *mut u32) -> c_int;
    fn oddio_hip_mixer_play_frames(m: *mut RawMixer, f: *mut RawFrames, start_seconds: f64, fixed_gain_db: f32, id: *mut u32) -> c_int;
    fn oddio_hip_mixer_play_chain(m: *mut RawMixer, leaf_kind: c_int, f: *mut RawFrames, start_seconds: f64, phase: f32, frequency_hz_or_value: f32, filters: *const RawFilter, n_filters: c_int, id: *mut u32) -> c_int;
    fn oddio_hip_mixer_set_gain(m: *mut RawMixer, id: u32, filter_index: c_int, amplitude_ratio: f32) -> c_int;
    fn oddio_hip_mixer_set_speed(m: *mut RawMixer, id: u32, filter_index: c_int, factor: f32) -> c_int;
    fn oddio_hip_mixer_get_amplitude_ratio(m: *mut RawMixer, id: u32, filter_index: c_int, amplitude_ratio: *mut f32) -> c_int;
    fn oddio_hip_mixer_get_speed(m: *mut RawMixer, id: u32, filter_index: c_int, factor: *mut f32) -> c_int;
    fn oddio_hip_mixer_stop(m: *mut RawMixer, id: u32) -> c_int;
    fn oddio_hip_mixer_source_release(m: *mut RawMixer, id: u32) -> c_int;
    fn oddio_hip_mixer_is_stopped(m: *mut RawMixer, id: u32, stopped: *mut c_int) -> c_int;
    fn oddio_hip_mixer_set_postfx(m: *mut RawMixer, postfx: c_int) -> c_int;
    fn oddio_hip_mixer_set_mode(m: *mut RawMixer, mode: c_int) -> c_int;
    fn oddio_hip_mixer_sample(m: *mut RawMixer, interval: f32, out: *mut f32, n_frames: usize) -> c_int;
    fn oddio_hip_mixer_sample_device(m: *mut RawMixer, interval: f32, dev_out: *mut f32, n_frames: usize) -> c_int;
    fn oddio_hip_mixer_synchronize(m: *mut RawMixer) -> c_int;
}

fn check(rc: c_int) {
    if rc != 0 {
        // the reference's path is infallible (no Result anywhere); a HIP failure is a bug or a lost device
        let msg = unsafe { CStr::from_ptr(oddio_hip_last_error()) }.to_string_lossy().into_owned();
        panic!("oddio_hip error {rc}: {msg}");
    }
}

fn pv(o: &SpatialOptions) -> ([f32; 3], [f32; 3]) {
    ([o.position.x, o.position.y, o.position.z], [o.velocity.x, o.velocity.y, o.velocity.z])
}

/// `Arc<Frames<f32>>` uploaded once to HBM (src/frames.rs:26-47).
pub struct HipFrames(*mut RawFrames);
unsafe impl Send for HipFrames {}
unsafe impl Sync for HipFrames {}
impl HipFrames {
    pub fn from_frames(device: i32, frames: &Arc<Frames<f32>>) -> Arc<Self> {
        let mut h = std::ptr::null_mut();
        check(unsafe { oddio_hip_frames_from_slice(device, frames.rate(), frames.as_ptr(), frames.len(), &mut h) });
        Arc::new(Self(h))
    }
    /// `Arc<Frames<[f32; 2]>>` (Mixer chains, `Downmix`)
    pub fn from_stereo_frames(device: i32, frames: &Arc<Frames<[f32; 2]>>) -> Arc<Self> {
        let mut h = std::ptr::null_mut();
        check(unsafe { oddio_hip_frames_from_slice_stereo(device, frames.rate(), frames.as_ptr() as *const f32, frames.len(), &mut h) });
        Arc::new(Self(h))
    }
}
impl Drop for HipFrames {
    fn drop(&mut self) {
        unsafe { oddio_hip_frames_release(self.0) };
    }
}

struct SceneHandle(*mut RawScene);
unsafe impl Send for SceneHandle {}
unsafe impl Sync for SceneHandle {} // control calls are queued lock-free inside the library (include/oddio_hip.h)
impl Drop for SceneHandle {
    fn drop(&mut self) {
        unsafe { oddio_hip_scene_destroy(self.0) };
    }
}

/// Drop-in for `SpatialScene` (src/spatial.rs:160-189): the audio thread owns it.
pub struct HipSpatialScene(Arc<SceneHandle>);
/// Drop-in for `SpatialSceneControl` (src/spatial.rs:267-350).
pub struct HipSpatialSceneControl(Arc<SceneHandle>);
/// Drop-in for `Spatial` (src/spatial.rs:119-157).
pub struct HipSpatial {
    scene: Arc<SceneHandle>,
    id: u32,
}

impl HipSpatialScene {
    pub fn new(device: i32, max_sources: u32, max_frames: u32) -> (HipSpatialSceneControl, Self) {
        let mut h = std::ptr::null_mut();
        check(unsafe { oddio_hip_scene_create(device, max_sources, max_frames, &mut h) });
        let h = Arc::new(SceneHandle(h));
        (HipSpatialSceneControl(h.clone()), Self(h))
    }
    /// `Reinhard::new(scene)` / `Tanh::new(scene)` fused on the device (src/reinhard.rs, src/tanh.rs)
    pub fn with_postfx(self, postfx: c_int) -> Self {
        check(unsafe { oddio_hip_scene_set_postfx((self.0).0, postfx) });
        self
    }
    /// The reference's sequential f32 sum, bit for bit (ODDIO_HIP_MODE_ORDERED): about 5.5x the cost of the default
    /// deterministic tree sum; allocates the per-source contribution rows on the calling thread.
    pub fn bit_exact(self) -> Self {
        check(unsafe { oddio_hip_scene_set_mode((self.0).0, 1) });
        self
    }
    /// The tree sum with the reference's roundings in every contribution (ODDIO_HIP_MODE_FAST_UNFUSED): the default
    /// mode fuses the lerp, the gain ramp and the accumulate of large scenes (within 1e-5 of the reference either way).
    pub fn unfused(self) -> Self {
        check(unsafe { oddio_hip_scene_set_mode((self.0).0, 2) });
        self
    }
    /// The reference's sequential sum to ~1e-6 of the peak at about twice the default mode's cost (ODDIO_HIP_MODE_TRACKED): large
    /// scenes, where the tree sum is farther than 1e-5 from the reference's f32 sum.  Allocates like `bit_exact`.
    pub fn tracked(self) -> Self {
        check(unsafe { oddio_hip_scene_set_mode((self.0).0, 3) });
        self
    }
    /// One logical scene split by source index over `world` GPUs (BASELINE configs[4]): every rank
    /// builds its shard's scene, then joins the stereo-buffer reduce with an `ncclUniqueId` made by
    /// rank 0 (`oddio_hip_reduce_unique_id`) and handed to the others by the host program.
    pub fn join_reduce(&mut self, rank: i32, world: i32, unique_id: &[u8]) {
        check(unsafe { oddio_hip_scene_reduce_init((self.0).0, rank, world, unique_id.as_ptr() as *const c_void, unique_id.len()) });
    }
    /// The same reduction without RCCL, summed in rank order on rank 0 (deterministic; ranks may share a GPU).
    /// Rank 0 calls it first and hands the filled `handle` (ODDIO_HIP_P2P_HANDLE_BYTES) to the other ranks.
    pub fn join_reduce_p2p(&mut self, rank: i32, world: i32, handle: &mut [u8]) {
        check(unsafe { oddio_hip_scene_reduce_init_p2p((self.0).0, rank, world, handle.as_mut_ptr() as *mut c_void, handle.len()) });
    }
}

impl Signal for HipSpatialScene {
    type Frame = [Sample; 2];
    fn sample(&mut self, interval: f32, out: &mut [[Sample; 2]]) {
        // src/spatial.rs:376
        check(unsafe { oddio_hip_scene_sample((self.0).0, interval, out.as_mut_ptr() as *mut f32, out.len()) });
    }
    fn is_finished(&self) -> bool {
        false // src/spatial.rs:473-476
    }
}
// `oddio::run(&mut scene, sample_rate, out)` (src/lib.rs:90-93) works unchanged: it only calls `sample`.

impl HipSpatialSceneControl {
    fn spatial(&self, id: u32) -> HipSpatial {
        HipSpatial { scene: self.0.clone(), id }
    }
    /// play(FramesSignal::new(frames, start_seconds), options)   (src/spatial.rs:289-302)
    pub fn play_frames(&mut self, frames: &Arc<HipFrames>, start_seconds: f64, options: SpatialOptions) -> HipSpatial {
        let (p, v) = pv(&options);
        let mut id = 0u32;
        check(unsafe { oddio_hip_scene_play_frames((self.0).0, frames.0, start_seconds, f32::NAN, p.as_ptr(), v.as_ptr(), options.radius, &mut id) });
        self.spatial(id)
    }
    /// play(FixedGain::new(FramesSignal::new(frames, start_seconds), db), options)   (src/gain.rs:18-23)
    pub fn play_frames_fixed_gain(&mut self, frames: &Arc<HipFrames>, start_seconds: f64, db: f32, options: SpatialOptions) -> HipSpatial {
        let (p, v) = pv(&options);
        let mut id = 0u32;
        check(unsafe { oddio_hip_scene_play_frames((self.0).0, frames.0, start_seconds, db, p.as_ptr(), v.as_ptr(), options.radius, &mut id) });
        self.spatial(id)
    }
    /// play(Sine::new(phase, frequency_hz), options)   (src/sine.rs:18-23)
    pub fn play_sine(&mut self, phase: f32, frequency_hz: f32, options: SpatialOptions) -> HipSpatial {
        let (p, v) = pv(&options);
        let mut id = 0u32;
        check(unsafe { oddio_hip_scene_play_sine((self.0).0, phase, frequency_hz, f32::NAN, p.as_ptr(), v.as_ptr(), options.radius, &mut id) });
        self.spatial(id)
    }
    /// play(Constant::new(value), options)   (src/constant.rs)
    pub fn play_constant(&mut self, value: f32, options: SpatialOptions) -> HipSpatial {
        let (p, v) = pv(&options);
        let mut id = 0u32;
        check(unsafe { oddio_hip_scene_play_constant((self.0).0, value, p.as_ptr(), v.as_ptr(), options.radius, &mut id) });
        self.spatial(id)
    }
    /// play(Cycle::new(frames), options)   (src/cycle.rs:17-23)
    pub fn play_cycle(&mut self, frames: &Arc<HipFrames>, options: SpatialOptions) -> HipSpatial {
        let (p, v) = pv(&options);
        let mut id = 0u32;
        check(unsafe { oddio_hip_scene_play_cycle((self.0).0, frames.0, f32::NAN, p.as_ptr(), v.as_ptr(), options.radius, &mut id) });
        self.spatial(id)
    }
    /// play(Reinhard::new(FramesSignal::new(frames, start_seconds)), options) and the other Seek chains around a clip: `filters`
    /// innermost first, up to four of FixedGain and the soft clips (`RawFilter { kind: 4 /* Reinhard */ | 5 /* Tanh */, .. }`) in any
    /// order and multiplicity -- `Reinhard::new(Tanh::new(x))`, `FixedGain::new(FixedGain::new(x, a), b)` -- (src/reinhard.rs:42-50,
    /// src/tanh.rs:36-44, src/gain.rs:39-51 are the `impl<T: Seek> Seek` that make such a signal playable).  One FixedGain and one
    /// clip at most: rendered inline by the staged kernels; longer nests: the exact per-lane path (include/oddio_hip.h).
    pub fn play_frames_filtered(&mut self, frames: &Arc<HipFrames>, start_seconds: f64, filters: &[RawFilter], options: SpatialOptions) -> HipSpatial {
        let (p, v) = pv(&options);
        let mut id = 0u32;
        check(unsafe {
            oddio_hip_scene_play_filtered((self.0).0, LEAF_FRAMES, frames.0, start_seconds, 0.0, 0.0, filters.as_ptr(), filters.len() as c_int, p.as_ptr(), v.as_ptr(), options.radius, &mut id)
        });
        self.spatial(id)
    }
    /// play(filters(Downmix::new(FramesSignal::new(stereo_frames, start_seconds))), options): the same wrapper nests around a Downmix
    /// of a stereo clip (src/downmix.rs:18-47 is Seek when its inner signal is), e.g. `Reinhard::new(Downmix::new(x))`.
    pub fn play_downmix_filtered(&mut self, stereo_frames: &Arc<HipFrames>, start_seconds: f64, filters: &[RawFilter], options: SpatialOptions) -> HipSpatial {
        let (p, v) = pv(&options);
        let mut id = 0u32;
        check(unsafe {
            oddio_hip_scene_play_filtered((self.0).0, LEAF_DOWNMIX, stereo_frames.0, start_seconds, 0.0, 0.0, filters.as_ptr(), filters.len() as c_int, p.as_ptr(), v.as_ptr(), options.radius, &mut id)
        });
        self.spatial(id)
    }
    /// play_buffered(filters(FramesSignal), options, max_distance, rate, buffer_duration)   (src/spatial.rs:314-340);
    /// `filters` innermost first; the returned handle's `set_gain` / `set_speed` are the GainControl / SpeedControl.
    pub fn play_buffered_frames(&mut self, frames: &Arc<HipFrames>, start_seconds: f64, filters: &[RawFilter], options: SpatialOptions, max_distance: f32, rate: u32, buffer_duration: f32) -> HipSpatial {
        let (p, v) = pv(&options);
        let mut id = 0u32;
        check(unsafe {
            oddio_hip_scene_play_buffered((self.0).0, LEAF_FRAMES, frames.0, start_seconds, 0.0, 0.0, filters.as_ptr(), filters.len() as c_int, p.as_ptr(), v.as_ptr(), options.radius, max_distance, rate, buffer_duration, &mut id)
        });
        self.spatial(id)
    }
    /// Capacity of the buffered set, before the first play_buffered (default 256)
    pub fn reserve_buffered(&mut self, max_buffered: u32) {
        check(unsafe { oddio_hip_scene_reserve_buffered((self.0).0, max_buffered) });
    }
    /// `n` x play_buffered(filters(FramesSignal::new(frames[i], start[i])), ..) under one lock, the rings from one stretch of
    /// device memory (scenes with 10^5..10^6 Gain / Speed sources); `filter_params` is [n][filter_kinds.len()].  Returns the handle
    /// ids; `set_control_batch(ids, filter_index, values)` stores to their GainControls / SpeedControls.
    pub fn play_buffered_frames_batch(&mut self, frames: &[Arc<HipFrames>], start_seconds: &[f64], filter_kinds: &[c_int], filter_params: &[f32], positions: &[[f32; 3]], velocities: &[[f32; 3]], radii: &[f32], max_distance: f32, rate: u32, buffer_duration: f32) -> Vec<u32> {
        let n = frames.len();
        assert!(start_seconds.len() == n && positions.len() == n && velocities.len() == n && radii.len() == n && filter_params.len() == n * filter_kinds.len());
        let raw: Vec<*mut RawFrames> = frames.iter().map(|f| f.0).collect();
        let mut ids = vec![0u32; n];
        check(unsafe {
            oddio_hip_scene_play_buffered_batch((self.0).0, n, raw.as_ptr(), start_seconds.as_ptr(), filter_kinds.as_ptr(), filter_kinds.len() as c_int, filter_params.as_ptr(), positions.as_ptr() as *const f32, velocities.as_ptr() as *const f32, radii.as_ptr(), max_distance, rate, buffer_duration, ids.as_mut_ptr())
        });
        ids
    }
    pub fn set_control_batch(&mut self, ids: &[u32], filter_index: i32, values: &[f32]) {
        assert_eq!(ids.len(), values.len());
        check(unsafe { oddio_hip_scene_set_control_batch((self.0).0, ids.len(), ids.as_ptr(), filter_index, values.as_ptr()) });
    }
    pub fn set_listener_rotation(&mut self, rotation: mint::Quaternion<f32>) {
        // src/spatial.rs:345-349 (the library stores the inverse, like the reference)
        let q = [rotation.s, rotation.v.x, rotation.v.y, rotation.v.z];
        check(unsafe { oddio_hip_scene_set_listener_rotation((self.0).0, q.as_ptr()) });
    }
}

impl HipSpatial {
    pub fn set_motion(&mut self, position: mint::Point3<f32>, velocity: mint::Vector3<f32>, discontinuity: bool) {
        // src/spatial.rs:137-149
        let (p, v) = ([position.x, position.y, position.z], [velocity.x, velocity.y, velocity.z]);
        check(unsafe { oddio_hip_source_set_motion(self.scene.0, self.id, p.as_ptr(), v.as_ptr(), discontinuity as c_int) });
    }
    pub fn is_finished(&self) -> bool {
        // src/spatial.rs:154-156
        let mut f: c_int = 0;
        check(unsafe { oddio_hip_source_is_finished(self.scene.0, self.id, &mut f) });
        f != 0
    }
    /// GainControl::set_amplitude_ratio (src/gain.rs:158-160) of filter `index` of a buffered source
    pub fn set_amplitude_ratio(&mut self, index: i32, factor: f32) {
        check(unsafe { oddio_hip_source_set_gain(self.scene.0, self.id, index, factor) });
    }
    /// GainControl::set_gain (src/gain.rs:141-143)
    pub fn set_gain(&mut self, index: i32, db: f32) {
        check(unsafe { oddio_hip_source_set_gain_db(self.scene.0, self.id, index, db) });
    }
    /// SpeedControl::set_speed (src/speed.rs:52-54)
    pub fn set_speed(&mut self, index: i32, factor: f32) {
        check(unsafe { oddio_hip_source_set_speed(self.scene.0, self.id, index, factor) });
    }
    /// GainControl::amplitude_ratio (src/gain.rs:147-150)
    pub fn amplitude_ratio(&self, index: i32) -> f32 {
        let mut v = 1.0f32;
        check(unsafe { oddio_hip_source_get_amplitude_ratio(self.scene.0, self.id, index, &mut v) });
        v
    }
    /// GainControl::gain (src/gain.rs:133-135), decibels
    pub fn gain(&self, index: i32) -> f32 {
        let mut v = 0.0f32;
        check(unsafe { oddio_hip_source_get_gain_db(self.scene.0, self.id, index, &mut v) });
        v
    }
    /// SpeedControl::speed (src/speed.rs:47-49)
    pub fn speed(&self, index: i32) -> f32 {
        let mut v = 1.0f32;
        check(unsafe { oddio_hip_source_get_speed(self.scene.0, self.id, index, &mut v) });
        v
    }
}
impl Drop for HipSpatial {
    fn drop(&mut self) {
        unsafe { oddio_hip_source_release(self.scene.0, self.id) };
    }
}

struct MixerHandle(*mut RawMixer);
unsafe impl Send for MixerHandle {}
unsafe impl Sync for MixerHandle {}
impl Drop for MixerHandle {
    fn drop(&mut self) {
        unsafe { oddio_hip_mixer_destroy(self.0) };
    }
}
/// Drop-in for `Mixer<[Sample; 2]>` (src/mixer.rs:46-120)
pub struct HipMixer(Arc<MixerHandle>);
/// Drop-in for `MixerControl` (src/mixer.rs:9-27)
pub struct HipMixerControl(Arc<MixerHandle>);
/// Drop-in for `Mixed` (src/mixer.rs:30-44)
pub struct HipMixed {
    mixer: Arc<MixerHandle>,
    id: u32,
}

impl HipMixer {
    pub fn new(device: i32, max_sources: u32, max_frames: u32) -> (HipMixerControl, Self) {
        let mut h = std::ptr::null_mut();
        check(unsafe { oddio_hip_mixer_create(device, max_sources, max_frames, &mut h) });
        let h = Arc::new(MixerHandle(h));
        (HipMixerControl(h.clone()), Self(h))
    }
    pub fn with_postfx(self, postfx: c_int) -> Self {
        check(unsafe { oddio_hip_mixer_set_postfx((self.0).0, postfx) });
        self
    }
    /// The reference's sequential f32 sum over the sources (src/mixer.rs:100-117), bit for bit (ODDIO_HIP_MODE_ORDERED).
    pub fn bit_exact(self) -> Self {
        check(unsafe { oddio_hip_mixer_set_mode((self.0).0, 1) });
        self
    }
    /// That sum to ~1e-6 of the peak at about twice the default mode's cost (ODDIO_HIP_MODE_TRACKED): mixers of thousands of sources,
    /// coherent mixes above all, where the default tree sum is farther than 1e-5 from the reference's f32 sum.
    pub fn tracked(self) -> Self {
        check(unsafe { oddio_hip_mixer_set_mode((self.0).0, 3) });
        self
    }
}
impl HipMixer {
    /// `Mixer::sample` with the frames left in device memory (`dev_out`: 2 * n_frames floats on the mixer's device) and no wait, for
    /// hosts that consume the mix on the GPU or enqueue several callbacks; `synchronize` waits for the mixer's stream.  Not part of
    /// the reference's interface (src/mixer.rs:92-119 fills a host slice): the counterpart of `HipSpatialScene::sample_device`.
    ///
    /// # Safety
    /// `dev_out` must be device memory of at least `2 * n_frames` floats that stays valid until the callback has executed.
    pub unsafe fn sample_device(&mut self, interval: f32, dev_out: *mut f32, n_frames: usize) {
        check(oddio_hip_mixer_sample_device((self.0).0, interval, dev_out, n_frames));
    }
    pub fn synchronize(&mut self) {
        check(unsafe { oddio_hip_mixer_synchronize((self.0).0) });
    }
}
impl Signal for HipMixer {
    type Frame = [Sample; 2];
    fn sample(&mut self, interval: f32, out: &mut [[Sample; 2]]) {
        // src/mixer.rs:92-119
        check(unsafe { oddio_hip_mixer_sample((self.0).0, interval, out.as_mut_ptr() as *mut f32, out.len()) });
    }
}

/// Drop-in for `Mixer<f32>` (src/mixer.rs:46-81 with `impl Frame for f32`, src/frame.rs:53-61): mono signals only; the
/// control half is the same `HipMixerControl` (its stereo-clip plays are refused by the library).
pub struct HipMonoMixer(Arc<MixerHandle>);
impl HipMonoMixer {
    pub fn new(device: i32, max_sources: u32, max_frames: u32) -> (HipMixerControl, Self) {
        let mut h = std::ptr::null_mut();
        check(unsafe { oddio_hip_mixer_create_mono(device, max_sources, max_frames, &mut h) });
        let h = Arc::new(MixerHandle(h));
        (HipMixerControl(h.clone()), Self(h))
    }
}
impl Signal for HipMonoMixer {
    type Frame = Sample;
    fn sample(&mut self, interval: f32, out: &mut [Sample]) {
        // src/mixer.rs:92-119 with T = f32
        check(unsafe { oddio_hip_mixer_sample((self.0).0, interval, out.as_mut_ptr(), out.len()) });
    }
}
impl HipMixerControl {
    /// play(MonoToStereo::new(Sine::new(phase, frequency_hz)))   (examples/simple.rs:42)
    pub fn play_sine(&mut self, phase: f32, frequency_hz: f32) -> HipMixed {
        let mut id = 0u32;
        check(unsafe { oddio_hip_mixer_play_sine((self.0).0, phase, frequency_hz, f32::NAN, &mut id) });
        HipMixed { mixer: self.0.clone(), id }
    }
    /// play(MonoToStereo::new(FramesSignal::new(frames, start_seconds)))
    pub fn play_frames(&mut self, frames: &Arc<HipFrames>, start_seconds: f64) -> HipMixed {
        let mut id = 0u32;
        check(unsafe { oddio_hip_mixer_play_frames((self.0).0, frames.0, start_seconds, f32::NAN, &mut id) });
        HipMixed { mixer: self.0.clone(), id }
    }
    /// play(filters(FramesSignal<[f32;2]> | MonoToStereo<FramesSignal<f32>>)): Gain / Speed / FixedGain chains
    pub fn play_chain_frames(&mut self, frames: &Arc<HipFrames>, start_seconds: f64, filters: &[RawFilter]) -> HipMixed {
        let mut id = 0u32;
        check(unsafe { oddio_hip_mixer_play_chain((self.0).0, LEAF_FRAMES, frames.0, start_seconds, 0.0, 0.0, filters.as_ptr(), filters.len() as c_int, &mut id) });
        HipMixed { mixer: self.0.clone(), id }
    }
}
impl Drop for HipMixed {
    fn drop(&mut self) {
        unsafe { oddio_hip_mixer_source_release(self.mixer.0, self.id) };
    }
}
impl HipMixed {
    pub fn stop(&mut self) {
        // src/mixer.rs:34-37
        check(unsafe { oddio_hip_mixer_stop(self.mixer.0, self.id) });
    }
    pub fn is_stopped(&self) -> bool {
        // src/mixer.rs:40-43
        let mut f: c_int = 0;
        check(unsafe { oddio_hip_mixer_is_stopped(self.mixer.0, self.id, &mut f) });
        f != 0
    }
    pub fn set_amplitude_ratio(&mut self, index: i32, factor: f32) {
        check(unsafe { oddio_hip_mixer_set_gain(self.mixer.0, self.id, index, factor) });
    }
    pub fn set_speed(&mut self, index: i32, factor: f32) {
        check(unsafe { oddio_hip_mixer_set_speed(self.mixer.0, self.id, index, factor) });
    }
    /// GainControl::amplitude_ratio (src/gain.rs:147-150)
    pub fn amplitude_ratio(&self, index: i32) -> f32 {
        let mut v = 1.0f32;
        check(unsafe { oddio_hip_mixer_get_amplitude_ratio(self.mixer.0, self.id, index, &mut v) });
        v
    }
    /// SpeedControl::speed (src/speed.rs:47-49)
    pub fn speed(&self, index: i32) -> f32 {
        let mut v = 1.0f32;
        check(unsafe { oddio_hip_mixer_get_speed(self.mixer.0, self.id, index, &mut v) });
        v
    }
}
