// oddio_hip.hip -- host side of libodd_hip.so: the scene/mixer objects and the C ABI of
// include/oddio_hip.h.  gfx950 only.  The host logic mirrors the reference's control plane at the
// boundary: `Set` insert/update/swap_remove (src/set.rs:55-66,141-188), the `swap` "latest value"
// hand-off for Motion / listener rotation (src/swap.rs:36-64), and SpatialScene::sample's
// prologue (src/spatial.rs:376-394).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: the functions are resolved with dlopen on first use

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/oddio_hip.h"
#include "kernels.h"
#include "mixer_kernels.h"
#include "buffered_kernels.h"

using namespace oddio_hip;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail((int)_e, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ---------------------------------------------------------------------------------------------
// Frames
// ---------------------------------------------------------------------------------------------
struct oddio_hip_frames {
    int device = 0;
    uint32_t rate = 0;
    size_t len = 0;          // frames
    uint32_t channels = 1;   // 1: Frames<f32>; 2: Frames<[f32;2]> (interleaved), Mixer general path only
    float* dev = nullptr;
    bool owned = true;
    void* pinned_block = nullptr;   // Stream rings: the hipHostMalloc'ed block `dev` points into
    std::atomic<int> refs{1};
};

static int frames_alloc(int device, uint32_t rate, size_t len, uint32_t channels, oddio_hip_frames** out) {
    if (!out) return fail(ODDIO_HIP_EINVAL, "out is NULL");
    if (len == 0) return fail(ODDIO_HIP_EINVAL, "empty clip (the reference panics in Frames::get_pair, frames.rs:111)");
    if (len > 0x7fffff00u) return fail(ODDIO_HIP_EINVAL, "clip too long (%zu samples)", len);
    if (rate == 0) return fail(ODDIO_HIP_EINVAL, "rate must be > 0");
    auto* f = new oddio_hip_frames();
    f->device = device; f->rate = rate; f->len = len; f->channels = channels;
    const size_t padded = (len * channels + 3) & ~size_t(3);
    DeviceGuard g(device);
    if (!g.ok) { delete f; return fail(ODDIO_HIP_ENODEV, "hipSetDevice(%d) failed", device); }
    hipError_t e = hipMalloc(&f->dev, padded * sizeof(float));
    if (e != hipSuccess) { delete f; return fail(ODDIO_HIP_ENOMEM, "hipMalloc(%zu floats): %s", padded, hipGetErrorString(e)); }
    e = hipMemset(f->dev + (padded - 4), 0, 4 * sizeof(float));   // zero tail pad: S(i) = 0 for i >= len
    if (e != hipSuccess) { (void)hipFree(f->dev); delete f; return fail((int)e, "hipMemset: %s", hipGetErrorString(e)); }
    *out = f;
    return 0;
}

extern "C" int oddio_hip_frames_from_slice(int device, uint32_t rate, const float* samples, size_t len, oddio_hip_frames** out) {
    if (!samples && len) return fail(ODDIO_HIP_EINVAL, "samples is NULL");
    oddio_hip_frames* f = nullptr;
    int rc = frames_alloc(device, rate, len, 1, &f);
    if (rc) return rc;
    DeviceGuard g(device);
    hipError_t e = hipMemcpy(f->dev, samples, len * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(f->dev); delete f; return fail((int)e, "hipMemcpy H2D: %s", hipGetErrorString(e)); }
    *out = f;
    return 0;
}

extern "C" int oddio_hip_frames_from_slice_stereo(int device, uint32_t rate, const float* interleaved, size_t n_frames, oddio_hip_frames** out) {
    if (!interleaved && n_frames) return fail(ODDIO_HIP_EINVAL, "samples is NULL");
    oddio_hip_frames* f = nullptr;
    int rc = frames_alloc(device, rate, n_frames, 2, &f);
    if (rc) return rc;
    DeviceGuard g(device);
    hipError_t e = hipMemcpy(f->dev, interleaved, 2 * n_frames * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(f->dev); delete f; return fail((int)e, "hipMemcpy H2D: %s", hipGetErrorString(e)); }
    *out = f;
    return 0;
}

extern "C" int oddio_hip_frames_from_device(int device, uint32_t rate, const float* dev_samples, size_t len, int copy, oddio_hip_frames** out) {
    if (!dev_samples || !out) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    if (copy) {
        oddio_hip_frames* f = nullptr;
        int rc = frames_alloc(device, rate, len, 1, &f);
        if (rc) return rc;
        DeviceGuard g(device);
        hipError_t e = hipMemcpy(f->dev, dev_samples, len * sizeof(float), hipMemcpyDeviceToDevice);
        if (e != hipSuccess) { (void)hipFree(f->dev); delete f; return fail((int)e, "hipMemcpy D2D: %s", hipGetErrorString(e)); }
        *out = f;
        return 0;
    }
    if (len == 0 || (len & 3) || ((uintptr_t)dev_samples & 15)) return fail(ODDIO_HIP_EINVAL, "borrowed clips need len %% 4 == 0 and 16-byte alignment");
    if (len > 0x7fffff00u || rate == 0) return fail(ODDIO_HIP_EINVAL, "bad len/rate");
    auto* f = new oddio_hip_frames();
    f->device = device; f->rate = rate; f->len = len; f->dev = const_cast<float*>(dev_samples); f->owned = false;
    *out = f;
    return 0;
}

extern "C" int oddio_hip_frames_retain(oddio_hip_frames* f) {
    if (!f) return fail(ODDIO_HIP_EINVAL, "NULL frames");
    f->refs.fetch_add(1, std::memory_order_relaxed);
    return 0;
}
extern "C" int oddio_hip_frames_release(oddio_hip_frames* f) {
    if (!f) return fail(ODDIO_HIP_EINVAL, "NULL frames");
    if (f->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        if (f->pinned_block) { DeviceGuard g(f->device); (void)hipHostFree(f->pinned_block); }
        else if (f->owned && f->dev) { DeviceGuard g(f->device); (void)hipFree(f->dev); }
        delete f;
    }
    return 0;
}
extern "C" int oddio_hip_frames_info(const oddio_hip_frames* f, uint32_t* rate, size_t* len) {
    if (!f) return fail(ODDIO_HIP_EINVAL, "NULL frames");
    if (rate) *rate = f->rate;
    if (len) *len = f->len;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Stream (stream.rs): StreamControl on the host, the SPSC ring (spsc.rs) in pinned GPU-visible memory
// ---------------------------------------------------------------------------------------------
struct oddio_hip_stream {
    oddio_hip_frames* ring = nullptr;   // refcounted owner of the pinned block; ->dev = device address of the samples
    StreamHeader* hdr = nullptr;        // host addresses
    float* data = nullptr;
    uint32_t size = 0;                  // capacity + 1 slots (spsc.rs:12)
    uint32_t channels = 1, rate = 0;
    bool played = false;
};

extern "C" int oddio_hip_stream_create(int device, uint32_t rate, size_t size_frames, uint32_t channels, oddio_hip_stream** out) {
    if (!out) return fail(ODDIO_HIP_EINVAL, "out is NULL");
    if (rate == 0 || (channels != 1 && channels != 2)) return fail(ODDIO_HIP_EINVAL, "rate must be > 0, channels 1 or 2");
    if (size_frames >= 0x7ffffff0u) return fail(ODDIO_HIP_EINVAL, "stream too large");
    DeviceGuard g(device);
    if (!g.ok) return fail(ODDIO_HIP_ENODEV, "hipSetDevice(%d) failed", device);
    const uint32_t size = (uint32_t)size_frames + 1u;
    void* block = nullptr;
    const size_t bytes = sizeof(StreamHeader) + (size_t)size * channels * sizeof(float);
    hipError_t e = hipHostMalloc(&block, bytes, hipHostMallocCoherent | hipHostMallocMapped);
    if (e != hipSuccess) return fail(ODDIO_HIP_ENOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    memset(block, 0, bytes);
    void* dblock = nullptr;
    e = hipHostGetDevicePointer(&dblock, block, 0);
    if (e != hipSuccess) { (void)hipHostFree(block); return fail((int)e, "hipHostGetDevicePointer: %s", hipGetErrorString(e)); }
    auto* f = new oddio_hip_frames();
    f->device = device; f->rate = rate; f->len = size; f->channels = channels; f->owned = false; f->pinned_block = block;
    f->dev = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(dblock) + sizeof(StreamHeader));
    auto* st = new oddio_hip_stream();
    st->ring = f;
    st->hdr = reinterpret_cast<StreamHeader*>(block);
    st->data = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(block) + sizeof(StreamHeader));
    st->size = size; st->channels = channels; st->rate = rate;
    *out = st;
    return 0;
}

// spsc::Sender::free (spsc.rs:75-86)
static uint32_t stream_free_slots(const oddio_hip_stream* st, uint32_t write, uint32_t read) {
    if (write < read) return read - write - 1u;
    if (read >= 1u) return st->size - write + (read - 1u);
    return st->size - write - 1u;
}

extern "C" int oddio_hip_stream_free(oddio_hip_stream* st, size_t* n_frames) {
    if (!st || !n_frames) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    const uint32_t write = st->hdr->write;
    const uint32_t read = __atomic_load_n(&st->hdr->read, __ATOMIC_ACQUIRE);
    *n_frames = stream_free_slots(st, write, read);
    return 0;
}

// spsc::Sender::send_from_slice (spsc.rs:27-67): append a prefix of `samples`, report how much fitted
extern "C" int oddio_hip_stream_write(oddio_hip_stream* st, const float* samples, size_t n_frames, size_t* consumed) {
    if (!st || (!samples && n_frames)) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    if (st->hdr->closed) return fail(ODDIO_HIP_ESTATE, "the stream control has been dropped");
    const uint32_t write = st->hdr->write;
    const uint32_t read = __atomic_load_n(&st->hdr->read, __ATOMIC_ACQUIRE);
    const uint32_t C = st->channels;
    size_t n1_cap, n2_cap;
    if (write < read) { n1_cap = read - write - 1u; n2_cap = 0; }
    else if (read >= 1u) { n1_cap = st->size - write; n2_cap = read - 1u; }
    else { n1_cap = st->size - write - 1u; n2_cap = 0; }
    const size_t n1 = std::min(n1_cap, n_frames);
    memcpy(st->data + (size_t)write * C, samples, n1 * C * sizeof(float));
    const size_t n2 = std::min(n2_cap, n_frames - n1);
    if (n2) memcpy(st->data, samples + n1 * C, n2 * C * sizeof(float));
    const size_t n = n1 + n2;
    __atomic_store_n(&st->hdr->write, (uint32_t)((write + n) % st->size), __ATOMIC_RELEASE);
    if (consumed) *consumed = n;
    return 0;
}

// drop(StreamControl): the receiver sees is_closed (spsc.rs:165-167) and the Stream finishes once it
// has been drained (stream.rs:71-73, :88-90).  The handle is invalid afterwards.
extern "C" int oddio_hip_stream_drop(oddio_hip_stream* st) {
    if (!st) return fail(ODDIO_HIP_EINVAL, "NULL stream");
    __atomic_store_n(&st->hdr->closed, 1u, __ATOMIC_RELEASE);
    oddio_hip_frames_release(st->ring);
    delete st;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// library
// ---------------------------------------------------------------------------------------------
extern "C" int oddio_hip_abi_version(void) { return ODDIO_HIP_ABI_VERSION; }
extern "C" const char* oddio_hip_last_error(void) { return g_last_error.c_str(); }
extern "C" int oddio_hip_device_count(int* count) {
    if (!count) return fail(ODDIO_HIP_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(ODDIO_HIP_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return 0;
}

// RCCL entry points of the sharded-scene reduce, resolved at first use (include/oddio_hip.h)
namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return GetUniqueId && CommInitRank && CommDestroy && AllReduce && GetErrorString; }
};
RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // a host process that already uses RCCL (e.g. through PyTorch) has librccl.so.1 mapped: share it
        api.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!api.handle) api.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!api.handle) return;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.handle, "ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
    });
    return &api;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// SpatialScene
// ---------------------------------------------------------------------------------------------
namespace {

struct HandleRec {
    uint32_t slot = 0xffffffffu;   // current slot while in the set
    bool in_set = false;           // inserted (after update()) and not yet removed
    bool queued = false;           // play() called, not yet update()d
    bool finished = false;         // Spatial::is_finished
    bool released = false;         // handle dropped by the user
    bool buffered = false;         // lives in the buffered set (play_buffered)
    float* ring = nullptr;         // device Ring of a buffered source
    uint64_t motion_epoch = 0;     // dedupe stamp for set_motion
    oddio_hip_frames* frames = nullptr;
    uint32_t fader = 0;                          // 1 + FaderRec index of a buffered Fader source
    std::vector<oddio_hip_frames*> fade_frames;  // clips of the signals handed to fade_to
};

struct PendingPlay { uint32_t id; SrcStatic st; SrcDyn dyn; };
struct PendingMotion { uint32_t id; float pos[3]; float vel[3]; uint32_t disc; };
struct PendingPlayB { uint32_t id; BufStatic st; BufDyn dyn; };
struct PendingControl { uint32_t id; uint32_t index; float value; };

constexpr uint32_t SCENE_FADER_CAP = 256;   // Fader sources per scene
constexpr uint32_t STOPPED_CAP = 4096;   // ids returned inline with each callback; more => second fetch
constexpr int RING = 2;

}  // namespace

// Adapt::new(signal, initial_rms, options) wrapped around a scene / mixer (adapt.rs:14-61)
struct AdaptHost {
    bool on = false;
    float tau = 0.1f, max_gain = INFINITY, low = 0.0f, high = 0.0f;
    float* d_state = nullptr;          // avg_squared
};

static int adapt_configure(AdaptHost& a, int device, hipStream_t stream, int enable, float initial_rms, float tau, float max_gain,
                           float low, float high) {
    if (!enable) { a.on = false; return 0; }
    if (!(tau > 0.0f)) return fail(ODDIO_HIP_EINVAL, "AdaptOptions::tau must be > 0");
    DeviceGuard g(device);
    if (!g.ok) return fail(ODDIO_HIP_ENODEV, "hipSetDevice(%d) failed", device);
    if (!a.d_state) HIP_TRY(hipMalloc(&a.d_state, sizeof(float)));
    const float avg_squared = initial_rms * initial_rms;                                        // adapt.rs:28
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipMemcpy(a.d_state, &avg_squared, sizeof(float), hipMemcpyHostToDevice));
    a.tau = tau; a.max_gain = max_gain; a.low = low; a.high = high;
    a.on = true;
    return 0;
}

// Launches the epilogue on `buf` (interleaved stereo, raw sum) and applies `postfx` after it.
static int adapt_launch(const AdaptHost& a, hipStream_t stream, float interval, float* buf, size_t n_frames, int postfx) {
    if (n_frames == 0) return 0;
    AdaptParams A;
    A.alpha = 1.0f - expf(-interval / a.tau);                                                   // adapt.rs:70
    A.one_minus_alpha = 1.0f - A.alpha;
    A.max_gain = a.max_gain; A.low = a.low; A.high = a.high;
    hipLaunchKernelGGL(adapt_kernel, dim3(1), dim3(256), 0, stream, buf, (uint32_t)n_frames, A, a.d_state, postfx);
    HIP_TRY(hipGetLastError());
    return 0;
}

struct oddio_hip_scene {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    uint32_t max_sources = 0, max_frames = 0, waves_cap = 0, tiles_max = 0;
    // device arrays
    SrcStatic* d_static = nullptr;
    SrcDyn* d_dyn = nullptr;
    SrcPending* d_pend = nullptr;
    EarParams* d_ear = nullptr;
    uint32_t* d_stopped[RING] = {nullptr, nullptr};
    float* d_partials = nullptr;
    float* d_out = nullptr;
    float* d_stage1 = nullptr;
    // buffered set (play_buffered): allocated on first use by the control thread
    uint32_t max_buffered = 0, len_b = 0;
    BufStatic* d_bstatic = nullptr;
    BufDyn* d_bdyn = nullptr;
    SrcPending* d_bpend = nullptr;
    float* d_contrib = nullptr;
    uint32_t* d_bskip = nullptr;
    float* d_outb = nullptr;
    BufMove* d_bmoves = nullptr;
    ControlUpdate* d_ctrl = nullptr;
    std::vector<uint32_t> id_of_slot_b;
    std::vector<PendingPlayB> pending_plays_b;
    std::vector<PendingControl> pending_controls;
    size_t live_count_b = 0;
    std::vector<float*> ring_garbage;      // rings of removed sources, freed off the audio thread
    // Seek-set Cycle sources: contribution rows, allocated by the control thread on first use
    float* d_cycle_rows = nullptr;
    uint32_t cycle_cap = 0, cycle_next = 0;
    std::vector<uint32_t> cycle_free;      // rows of removed sources
    uint32_t cycle_live = 0;               // Cycle sources in the Seek set (audio thread)
    AdaptHost adapt;
    // Fader sources in the buffered set (fader.rs): records + 1024-frame scratch, allocated by the control thread
    FaderRec* d_faders = nullptr;
    float* d_fader_scratch = nullptr;
    uint32_t fader_count = 0;
    std::vector<std::pair<uint32_t, FaderPending>> pending_fades;   // (1 + record index, command)
    MotionUpdate* d_motion = nullptr;
    MotionUpdate* d_bmotion = nullptr;     // staging for buffered sources' motion updates (max_buffered entries)
    MotionUpdate* h_motion = nullptr;      // pinned staging for set_motion batches (max_sources entries)
    hipEvent_t ev_motion = nullptr;        // its last H2D copy
    bool motion_copy_pending = false;
    std::vector<PendingMotion> motion_scratch;   // reused capacity: no per-callback heap churn for large batches
    SlotMove* d_moves = nullptr;
    // pinned staging
    uint32_t* h_stopped[RING] = {nullptr, nullptr};
    uint32_t* hd_stopped[RING] = {nullptr, nullptr};   // the same pinned buffers as the device addresses them
    float* h_out = nullptr;
    hipEvent_t ev_stopped[RING] = {nullptr, nullptr};
    bool ring_busy[RING] = {false, false};
    uint32_t ring_nsrc[RING] = {0, 0};
    uint64_t call_index = 0;
    // control plane (guarded by mu)
    std::mutex mu;
    std::vector<PendingPlay> pending_plays;
    std::vector<PendingMotion> pending_motion;
    bool rot_fresh = false;
    float rot_pending[4] = {1, 0, 0, 0};
    std::vector<HandleRec> handles;
    std::vector<uint32_t> free_ids;
    size_t live_count = 0;                 // queued + in_set
    uint64_t motion_epoch = 0;
    // audio-thread state
    float rot[4] = {1, 0, 0, 0};
    uint32_t len = 0;                      // live slots
    std::vector<uint32_t> id_of_slot;
    int postfx = 0, mode = 0;
    ncclComm_t comm = nullptr;             // sharded scene: RCCL communicator of the stereo-buffer reduce
    int comm_world = 1;
    bool profiling = false;
    static constexpr int PROF_RING = 512;
    std::vector<hipEvent_t> ev_prof;       // PROF_RING x 4 events, created on first use
    uint64_t prof_calls = 0;               // profiled calls so far
};

static int scene_free(oddio_hip_scene* s) {
    DeviceGuard g(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->comm) { (void)rccl_api()->CommDestroy(s->comm); s->comm = nullptr; }
    for (auto& h : s->handles) if (h.frames) { oddio_hip_frames_release(h.frames); h.frames = nullptr; }
    for (auto& p : s->pending_plays) (void)p;
    (void)hipFree(s->d_static); (void)hipFree(s->d_dyn); (void)hipFree(s->d_pend); (void)hipFree(s->d_ear);
    (void)hipFree(s->d_partials); (void)hipFree(s->d_out); (void)hipFree(s->d_stage1);
    (void)hipFree(s->d_bstatic); (void)hipFree(s->d_bdyn); (void)hipFree(s->d_bpend); (void)hipFree(s->d_contrib); (void)hipFree(s->d_bskip);
    (void)hipFree(s->d_outb); (void)hipFree(s->d_bmoves); (void)hipFree(s->d_bmotion); (void)hipFree(s->d_ctrl); (void)hipFree(s->d_cycle_rows); (void)hipFree(s->adapt.d_state); (void)hipFree(s->d_faders); (void)hipFree(s->d_fader_scratch);
    for (auto& h : s->handles) { for (auto* f : h.fade_frames) oddio_hip_frames_release(f); h.fade_frames.clear(); }
    for (auto& h : s->handles) if (h.ring) { (void)hipFree(h.ring); h.ring = nullptr; }
    for (float* r : s->ring_garbage) (void)hipFree(r); (void)hipFree(s->d_motion); (void)hipFree(s->d_moves);
    for (int r = 0; r < RING; ++r) {
        (void)hipFree(s->d_stopped[r]);
        if (s->h_stopped[r]) (void)hipHostFree(s->h_stopped[r]);
        if (s->ev_stopped[r]) (void)hipEventDestroy(s->ev_stopped[r]);
    }
    if (s->h_out) (void)hipHostFree(s->h_out);
    if (s->h_motion) (void)hipHostFree(s->h_motion);
    if (s->ev_motion) (void)hipEventDestroy(s->ev_motion);
    for (auto& e : s->ev_prof) if (e) (void)hipEventDestroy(e);
    if (s->stream && s->owns_stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return 0;
}

extern "C" int oddio_hip_scene_create(int device, uint32_t max_sources, uint32_t max_frames, oddio_hip_scene** out) {
    if (!out) return fail(ODDIO_HIP_EINVAL, "out is NULL");
    if (max_sources == 0 || max_frames == 0) return fail(ODDIO_HIP_EINVAL, "max_sources and max_frames must be > 0");
    if (max_frames > (1u << 20)) return fail(ODDIO_HIP_EINVAL, "max_frames too large");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return fail(ODDIO_HIP_ENODEV, "no HIP device %d (count %d)", device, ndev);
    DeviceGuard g(device);
    if (!g.ok) return fail(ODDIO_HIP_ENODEV, "hipSetDevice(%d) failed", device);
    auto* s = new oddio_hip_scene();
    s->device = device; s->max_sources = max_sources; s->max_frames = max_frames;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete s; return fail(ODDIO_HIP_ENODEV, "hipGetDeviceProperties failed"); }
    {   // resident mix-kernel waves per CU, summed over the tiles of a callback; tunable for experiments
        uint32_t per_cu = MIX_WAVES_PER_CU;
        if (const char* e = getenv("ODDIO_HIP_WAVES_PER_CU")) { int v = atoi(e); if (v > 0 && v <= 64) per_cu = (uint32_t)v; }
        s->waves_cap = (uint32_t)prop.multiProcessorCount * per_cu;
        if (s->waves_cap == 0) s->waves_cap = 2048;
    }
    s->tiles_max = (max_frames + TILE_FRAMES - 1) / TILE_FRAMES;
    const size_t cap = max_sources;
    const uint32_t groups = (max_sources + MIX_GROUP - 1) / MIX_GROUP;
    const uint32_t wgs_max = (std::min(groups, s->waves_cap) + MIX_WG_WAVES - 1) / MIX_WG_WAVES + 1;
#define SC_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { int rc = fail(_e == hipErrorOutOfMemory ? ODDIO_HIP_ENOMEM : (int)_e, "%s: %s", #expr, hipGetErrorString(_e)); scene_free(s); return rc; } } while (0)
    SC_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    SC_TRY(hipMalloc(&s->d_static, cap * sizeof(SrcStatic)));
    SC_TRY(hipMalloc(&s->d_dyn, cap * sizeof(SrcDyn)));
    SC_TRY(hipMalloc(&s->d_pend, cap * sizeof(SrcPending)));
    SC_TRY(hipMalloc(&s->d_ear, cap * 2 * sizeof(EarParams)));
    SC_TRY(hipMalloc(&s->d_partials, (size_t)s->tiles_max * wgs_max * 2 * TILE_FRAMES * sizeof(float)));
    SC_TRY(hipMalloc(&s->d_out, (size_t)s->tiles_max * 2 * TILE_FRAMES * sizeof(float)));
    SC_TRY(hipMalloc(&s->d_stage1, (size_t)RED_SPLIT * s->tiles_max * 2 * TILE_FRAMES * sizeof(float)));
    SC_TRY(hipMalloc(&s->d_motion, cap * sizeof(MotionUpdate)));
    SC_TRY(hipHostMalloc(&s->h_motion, cap * sizeof(MotionUpdate), hipHostMallocDefault));
    SC_TRY(hipEventCreateWithFlags(&s->ev_motion, hipEventDisableTiming));
    SC_TRY(hipMalloc(&s->d_moves, cap * sizeof(SlotMove)));
    SC_TRY(hipHostMalloc(&s->h_out, (size_t)2 * max_frames * sizeof(float), hipHostMallocDefault));
    for (int r = 0; r < RING; ++r) {
        SC_TRY(hipMalloc(&s->d_stopped[r], (1 + STOPPED_CAP) * sizeof(uint32_t)));
        SC_TRY(hipHostMalloc(&s->h_stopped[r], (1 + STOPPED_CAP) * sizeof(uint32_t), hipHostMallocMapped));
        SC_TRY(hipHostGetDevicePointer((void**)&s->hd_stopped[r], s->h_stopped[r], 0));
        SC_TRY(hipMemsetAsync(s->d_stopped[r], 0, sizeof(uint32_t), s->stream));
        s->h_stopped[r][0] = 0;
        SC_TRY(hipEventCreateWithFlags(&s->ev_stopped[r], hipEventDisableTiming));
    }
    SC_TRY(hipMemsetAsync(s->d_pend, 0, cap * sizeof(SrcPending), s->stream));
    SC_TRY(hipStreamSynchronize(s->stream));
#undef SC_TRY
    s->id_of_slot.resize(cap);
    *out = s;
    return 0;
}

extern "C" int oddio_hip_scene_destroy(oddio_hip_scene* s) {
    if (!s) return fail(ODDIO_HIP_EINVAL, "NULL scene");
    return scene_free(s);
}

// ---- control plane --------------------------------------------------------------------------
static uint32_t alloc_id_locked(oddio_hip_scene* s) {
    if (!s->free_ids.empty()) { uint32_t id = s->free_ids.back(); s->free_ids.pop_back(); s->handles[id] = HandleRec(); return id; }
    s->handles.emplace_back();
    return (uint32_t)s->handles.size() - 1;
}

static float db_to_gain(float db) { return std::isnan(db) ? 1.0f : powf(10.0f, db / 20.0f); }   // gain.rs:20

static int scene_play_common(oddio_hip_scene* s, SrcStatic st, double start_seconds, float phase, oddio_hip_frames* frames,
                             const float pos[3], const float vel[3], float radius, uint32_t* source_id) {
    if (!s || !pos || !vel) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    st.radius = radius;
    SrcDyn d = {};
    d.t = start_seconds;
    d.phase = phase;
    d.state_dt = 0.0f;                                   // State::new, spatial.rs:495-499
    for (int k = 0; k < 3; ++k) { d.tgt_pos[k] = pos[k]; d.tgt_vel[k] = vel[k]; d.prev_pos[k] = pos[k]; }
    d.finished_for = 0.0f;
    d.flags = 0;
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->live_count >= s->max_sources) return fail(ODDIO_HIP_ENOMEM, "scene is full (max_sources = %u)", s->max_sources);
    s->live_count++;
    const uint32_t id = alloc_id_locked(s);
    d.id = id;
    HandleRec& h = s->handles[id];
    h.queued = true;
    h.frames = frames;
    if (frames) oddio_hip_frames_retain(frames);
    s->pending_plays.push_back({id, st, d});
    if (source_id) *source_id = id;
    return 0;
}

extern "C" int oddio_hip_scene_play_frames(oddio_hip_scene* s, oddio_hip_frames* frames, double start_seconds, float fixed_gain_db,
                                           const float position[3], const float velocity[3], float radius, uint32_t* source_id) {
    if (!s || !frames) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    if (frames->device != s->device) return fail(ODDIO_HIP_EINVAL, "frames live on device %d, scene on %d", frames->device, s->device);
    if (frames->channels != 1) return fail(ODDIO_HIP_EINVAL, "spatial scenes take mono clips (Frame = Sample, spatial.rs:291)");
    SrcStatic st = {};
    st.clip = frames->dev; st.clip_len = (uint32_t)frames->len; st.clip_rate = frames->rate;
    st.fixed_gain = db_to_gain(fixed_gain_db); st.kind = KIND_FRAMES;
    return scene_play_common(s, st, start_seconds, 0.0f, frames, position, velocity, radius, source_id);
}

extern "C" int oddio_hip_scene_play_frames_downmix(oddio_hip_scene* s, oddio_hip_frames* frames, double start_seconds, float fixed_gain_db,
                                                   const float position[3], const float velocity[3], float radius, uint32_t* source_id) {
    if (!s || !frames) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    if (frames->device != s->device) return fail(ODDIO_HIP_EINVAL, "frames live on device %d, scene on %d", frames->device, s->device);
    if (frames->channels != 2) return fail(ODDIO_HIP_EINVAL, "Downmix takes a stereo clip (oddio_hip_frames_from_slice_stereo)");
    SrcStatic st = {};
    st.clip = frames->dev; st.clip_len = (uint32_t)frames->len; st.clip_rate = frames->rate;
    st.fixed_gain = db_to_gain(fixed_gain_db); st.kind = KIND_DOWNMIX;
    return scene_play_common(s, st, start_seconds, 0.0f, frames, position, velocity, radius, source_id);
}

extern "C" int oddio_hip_scene_play_frames_batch(oddio_hip_scene* s, size_t n, oddio_hip_frames* const* frames, const double* start_seconds,
                                                 const float* fixed_gain_db, const float* positions, const float* velocities,
                                                 const float* radii, uint32_t* ids) {
    if (!s || !frames || !start_seconds || !positions || !velocities || !radii) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (s->live_count + n > s->max_sources) return fail(ODDIO_HIP_ENOMEM, "scene would overflow (max_sources = %u)", s->max_sources);
        for (size_t i = 0; i < n; ++i)
            if (!frames[i] || frames[i]->device != s->device || frames[i]->channels != 1) return fail(ODDIO_HIP_EINVAL, "frames[%zu] invalid", i);
        s->live_count += n;
        s->pending_plays.reserve(s->pending_plays.size() + n);
        s->handles.reserve(s->handles.size() + n);
        for (size_t i = 0; i < n; ++i) {
            oddio_hip_frames* f = frames[i];
            SrcStatic st = {};
            st.clip = f->dev; st.clip_len = (uint32_t)f->len; st.clip_rate = f->rate;
            st.fixed_gain = fixed_gain_db ? db_to_gain(fixed_gain_db[i]) : 1.0f; st.kind = KIND_FRAMES;
            st.radius = radii[i];
            SrcDyn d = {};
            d.t = start_seconds[i];
            for (int k = 0; k < 3; ++k) { d.tgt_pos[k] = positions[3 * i + k]; d.tgt_vel[k] = velocities[3 * i + k]; d.prev_pos[k] = positions[3 * i + k]; }
            const uint32_t id = alloc_id_locked(s);
            d.id = id;
            HandleRec& h = s->handles[id];
            h.queued = true; h.frames = f;
            oddio_hip_frames_retain(f);
            s->pending_plays.push_back({id, st, d});
            if (ids) ids[i] = id;
        }
    }
    return 0;
}

extern "C" int oddio_hip_scene_play_sine(oddio_hip_scene* s, float phase, float frequency_hz, float fixed_gain_db,
                                         const float position[3], const float velocity[3], float radius, uint32_t* source_id) {
    SrcStatic st = {};
    st.freq_or_value = frequency_hz * ODDIO_TAU;   // sine.rs:21
    st.fixed_gain = db_to_gain(fixed_gain_db); st.kind = KIND_SINE;
    return scene_play_common(s, st, 0.0, phase, nullptr, position, velocity, radius, source_id);
}

extern "C" int oddio_hip_scene_play_constant(oddio_hip_scene* s, float value, const float position[3], const float velocity[3],
                                             float radius, uint32_t* source_id) {
    SrcStatic st = {};
    st.freq_or_value = value; st.fixed_gain = 1.0f; st.kind = KIND_CONSTANT;
    return scene_play_common(s, st, 0.0, 0.0f, nullptr, position, velocity, radius, source_id);
}

constexpr uint32_t CYCLE_ROWS_DEFAULT = 1024;   // Seek-set Cycle sources per scene (ODDIO_HIP_MAX_CYCLE overrides)

extern "C" int oddio_hip_scene_play_cycle(oddio_hip_scene* s, oddio_hip_frames* frames, float fixed_gain_db,
                                          const float position[3], const float velocity[3], float radius, uint32_t* source_id) {
    if (!s || !frames) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    if (frames->device != s->device) return fail(ODDIO_HIP_EINVAL, "frames live on device %d, scene on %d", frames->device, s->device);
    if (frames->channels != 1) return fail(ODDIO_HIP_EINVAL, "spatial scenes take mono clips (Frame = Sample, spatial.rs:291)");
    uint32_t row;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (!s->d_cycle_rows) {   // control thread, never inside sample()
            uint32_t cap = std::min(CYCLE_ROWS_DEFAULT, s->max_sources);
            if (const char* e = getenv("ODDIO_HIP_MAX_CYCLE")) { long v = atol(e); if (v > 0) cap = (uint32_t)std::min<long>(v, s->max_sources); }
            DeviceGuard g(s->device);
            if (!g.ok) return fail(ODDIO_HIP_ENODEV, "hipSetDevice(%d) failed", s->device);
            hipError_t e = hipMalloc(&s->d_cycle_rows, (size_t)cap * 2 * s->max_frames * sizeof(float));
            if (e != hipSuccess) { s->d_cycle_rows = nullptr; return fail(ODDIO_HIP_ENOMEM, "hipMalloc(cycle rows): %s", hipGetErrorString(e)); }
            s->cycle_cap = cap;
        }
        if (!s->cycle_free.empty()) { row = s->cycle_free.back(); s->cycle_free.pop_back(); }
        else if (s->cycle_next < s->cycle_cap) row = s->cycle_next++;
        else return fail(ODDIO_HIP_ENOMEM, "too many Cycle sources in the Seek set (%u); raise ODDIO_HIP_MAX_CYCLE or use play_buffered", s->cycle_cap);
    }
    SrcStatic st = {};
    st.clip = frames->dev; st.clip_len = (uint32_t)frames->len; st.clip_rate = frames->rate;
    st.fixed_gain = db_to_gain(fixed_gain_db); st.kind = KIND_CYCLE;
    memcpy(&st.freq_or_value, &row, sizeof(row));
    int rc = scene_play_common(s, st, 0.0, 0.0f, frames, position, velocity, radius, source_id);   // Cycle::new: cursor 0 (cycle.rs:17-23)
    if (rc) { std::lock_guard<std::mutex> lk(s->mu); s->cycle_free.push_back(row); }
    return rc;
}

extern "C" int oddio_hip_source_set_motion(oddio_hip_scene* s, uint32_t id, const float position[3], const float velocity[3], int discontinuity) {
    if (!s || !position || !velocity) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (id >= s->handles.size() || s->handles[id].released) return fail(ODDIO_HIP_ESTATE, "unknown source id %u", id);
    PendingMotion m;
    m.id = id;
    for (int k = 0; k < 3; ++k) { m.pos[k] = position[k]; m.vel[k] = velocity[k]; }
    m.disc = discontinuity ? 1u : 0u;
    s->pending_motion.push_back(m);
    return 0;
}

extern "C" int oddio_hip_scene_set_motion_batch(oddio_hip_scene* s, size_t n, const uint32_t* ids, const float* positions,
                                                const float* velocities, int discontinuity) {
    if (!s || !ids || !positions || !velocities) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    s->pending_motion.reserve(s->pending_motion.size() + n);
    for (size_t i = 0; i < n; ++i) {
        if (ids[i] >= s->handles.size() || s->handles[ids[i]].released) return fail(ODDIO_HIP_ESTATE, "unknown source id %u", ids[i]);
        PendingMotion m;
        m.id = ids[i];
        for (int k = 0; k < 3; ++k) { m.pos[k] = positions[3 * i + k]; m.vel[k] = velocities[3 * i + k]; }
        m.disc = discontinuity ? 1u : 0u;
        s->pending_motion.push_back(m);
    }
    return 0;
}

extern "C" int oddio_hip_source_is_finished(oddio_hip_scene* s, uint32_t id, int* finished) {
    if (!s || !finished) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (id >= s->handles.size() || s->handles[id].released) return fail(ODDIO_HIP_ESTATE, "unknown source id %u", id);
    *finished = s->handles[id].finished ? 1 : 0;
    return 0;
}

extern "C" int oddio_hip_source_release(oddio_hip_scene* s, uint32_t id) {
    if (!s) return fail(ODDIO_HIP_EINVAL, "NULL scene");
    std::lock_guard<std::mutex> lk(s->mu);
    if (id >= s->handles.size() || s->handles[id].released) return fail(ODDIO_HIP_ESTATE, "unknown source id %u", id);
    HandleRec& h = s->handles[id];
    h.released = true;
    if (!h.in_set && !h.queued) s->free_ids.push_back(id);   // already removed: id reusable now
    return 0;
}

extern "C" int oddio_hip_scene_set_listener_rotation(oddio_hip_scene* s, const float q[4]) {
    if (!s || !q) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    s->rot_pending[0] = q[0]; s->rot_pending[1] = -q[1]; s->rot_pending[2] = -q[2]; s->rot_pending[3] = -q[3];   // invert_quat, spatial.rs:346
    s->rot_fresh = true;
    return 0;
}

extern "C" int oddio_hip_scene_set_postfx(oddio_hip_scene* s, int postfx) {
    if (!s || postfx < 0 || postfx > 2) return fail(ODDIO_HIP_EINVAL, "bad postfx");
    s->postfx = postfx;
    return 0;
}
extern "C" int oddio_hip_scene_set_adapt(oddio_hip_scene* s, int enable, float initial_rms, float tau, float max_gain, float low, float high) {
    if (!s) return fail(ODDIO_HIP_EINVAL, "NULL scene");
    return adapt_configure(s->adapt, s->device, s->stream, enable, initial_rms, tau, max_gain, low, high);
}
extern "C" int oddio_hip_scene_set_mode(oddio_hip_scene* s, int mode) {
    if (!s || mode < 0 || mode > 1) return fail(ODDIO_HIP_EINVAL, "bad mode");
    s->mode = mode;
    return 0;
}
extern "C" int oddio_hip_scene_len(oddio_hip_scene* s, size_t* len) {
    if (!s || !len) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    *len = s->len;
    return 0;
}
extern "C" int oddio_hip_scene_len_buffered(oddio_hip_scene* s, size_t* len) {
    if (!s || !len) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    *len = s->len_b;
    return 0;
}
extern "C" int oddio_hip_scene_set_profiling(oddio_hip_scene* s, int enable) {
    if (!s) return fail(ODDIO_HIP_EINVAL, "NULL scene");
    DeviceGuard g(s->device);
    if (enable && s->ev_prof.empty()) {
        s->ev_prof.assign((size_t)oddio_hip_scene::PROF_RING * 4, nullptr);
        for (auto& e : s->ev_prof) HIP_TRY(hipEventCreate(&e));
    }
    s->profiling = enable != 0;
    s->prof_calls = 0;
    return 0;
}
extern "C" int oddio_hip_scene_kernel_ms_history(oddio_hip_scene* s, float* ms, size_t max_calls, size_t* n_calls) {
    if (!s || !ms || !n_calls) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    DeviceGuard g(s->device);
    const uint64_t have = std::min<uint64_t>(s->prof_calls, oddio_hip_scene::PROF_RING);
    const uint64_t n = std::min<uint64_t>(have, max_calls);
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t call = s->prof_calls - n + i;
        hipEvent_t* ev = &s->ev_prof[(call % oddio_hip_scene::PROF_RING) * 4];
        HIP_TRY(hipEventSynchronize(ev[3]));
        for (int k = 0; k < 3; ++k) HIP_TRY(hipEventElapsedTime(&ms[3 * i + k], ev[k], ev[k + 1]));
    }
    *n_calls = (size_t)n;
    return 0;
}
extern "C" int oddio_hip_scene_last_kernel_ms(oddio_hip_scene* s, float ms[3]) {
    size_t n = 0;
    int rc = oddio_hip_scene_kernel_ms_history(s, ms, 1, &n);
    if (rc) return rc;
    if (n == 0) return fail(ODDIO_HIP_ESTATE, "no profiled call yet");
    return 0;
}
extern "C" int oddio_hip_scene_synchronize(oddio_hip_scene* s) {
    if (!s) return fail(ODDIO_HIP_EINVAL, "NULL scene");
    DeviceGuard g(s->device);
    HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}
extern "C" int oddio_hip_scene_set_stream(oddio_hip_scene* s, void* stream) {
    if (!s) return fail(ODDIO_HIP_EINVAL, "NULL scene");
    DeviceGuard g(s->device);
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->owns_stream && s->stream) (void)hipStreamDestroy(s->stream);
    s->stream = (hipStream_t)stream;   // NULL selects the legacy default stream
    s->owns_stream = false;
    return 0;
}
extern "C" int oddio_hip_scene_stream(oddio_hip_scene* s, void** stream) {
    if (!s || !stream) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    *stream = (void*)s->stream;
    return 0;
}

// ---- sharded scene: stereo-buffer all-reduce over RCCL (include/oddio_hip.h) -------------------

extern "C" int oddio_hip_reduce_unique_id(void* unique_id, size_t unique_id_bytes) {
    if (!unique_id || unique_id_bytes < sizeof(ncclUniqueId)) return fail(ODDIO_HIP_EINVAL, "unique_id needs %zu bytes", sizeof(ncclUniqueId));
    static_assert(sizeof(ncclUniqueId) == ODDIO_HIP_UNIQUE_ID_BYTES, "ODDIO_HIP_UNIQUE_ID_BYTES == sizeof(ncclUniqueId)");
    RcclApi* R = rccl_api();
    if (!R->ok()) return fail(ODDIO_HIP_ENODEV, "librccl.so.1 could not be loaded: %s", dlerror());
    ncclUniqueId id;
    const ncclResult_t r = R->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(ODDIO_HIP_ENODEV, "ncclGetUniqueId: %s", R->GetErrorString(r));
    memcpy(unique_id, &id, sizeof(id));
    return 0;
}

extern "C" int oddio_hip_scene_reduce_init(oddio_hip_scene* s, int rank, int world, const void* unique_id, size_t unique_id_bytes) {
    if (!s || !unique_id || unique_id_bytes < sizeof(ncclUniqueId)) return fail(ODDIO_HIP_EINVAL, "bad argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(ODDIO_HIP_EINVAL, "rank %d of %d", rank, world);
    if (s->comm) return fail(ODDIO_HIP_ESTATE, "the scene already belongs to a reduce group");
    RcclApi* R = rccl_api();
    if (!R->ok()) return fail(ODDIO_HIP_ENODEV, "librccl.so.1 could not be loaded: %s", dlerror());
    DeviceGuard g(s->device);
    if (!g.ok) return fail(ODDIO_HIP_ENODEV, "hipSetDevice(%d) failed", s->device);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = R->CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) return fail(ODDIO_HIP_ENODEV, "ncclCommInitRank(rank %d of %d): %s", rank, world, R->GetErrorString(r));
    s->comm = comm;
    s->comm_world = world;
    return 0;
}

extern "C" int oddio_hip_scene_reduce_destroy(oddio_hip_scene* s) {
    if (!s) return fail(ODDIO_HIP_EINVAL, "NULL scene");
    if (!s->comm) return 0;
    DeviceGuard g(s->device);
    (void)hipStreamSynchronize(s->stream);
    (void)rccl_api()->CommDestroy(s->comm);
    s->comm = nullptr;
    s->comm_world = 1;
    return 0;
}

// ---- buffered sources (play_buffered, spatial.rs:314-340) -------------------------------------
static int ensure_buffered_locked(oddio_hip_scene* s, uint32_t want) {
    if (s->d_bstatic) return want <= s->max_buffered ? 0 : fail(ODDIO_HIP_ENOMEM, "buffered capacity is %u (call oddio_hip_scene_reserve_buffered first)", s->max_buffered);
    const uint32_t cap = std::min<uint32_t>(std::max<uint32_t>(want, 256u), s->max_sources);
    DeviceGuard g(s->device);
    const size_t n_out = (size_t)2 * s->max_frames;
#define BF_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(_e == hipErrorOutOfMemory ? ODDIO_HIP_ENOMEM : (int)_e, "%s: %s", #expr, hipGetErrorString(_e)); } while (0)
    BF_TRY(hipMalloc(&s->d_bstatic, cap * sizeof(BufStatic)));
    BF_TRY(hipMalloc(&s->d_bdyn, cap * sizeof(BufDyn)));
    BF_TRY(hipMalloc(&s->d_bpend, cap * sizeof(SrcPending)));
    BF_TRY(hipMalloc(&s->d_contrib, cap * n_out * sizeof(float)));
    BF_TRY(hipMalloc(&s->d_bskip, cap * sizeof(uint32_t)));
    BF_TRY(hipMalloc(&s->d_outb, n_out * sizeof(float)));
    BF_TRY(hipMalloc(&s->d_bmoves, cap * sizeof(BufMove)));
    BF_TRY(hipMalloc(&s->d_bmotion, cap * sizeof(MotionUpdate)));
    BF_TRY(hipMalloc(&s->d_ctrl, 4096 * sizeof(ControlUpdate)));
    BF_TRY(hipMemset(s->d_bpend, 0, cap * sizeof(SrcPending)));
#undef BF_TRY
    s->id_of_slot_b.resize(cap);
    s->max_buffered = cap;
    return 0;
}

extern "C" int oddio_hip_scene_reserve_buffered(oddio_hip_scene* s, uint32_t max_buffered) {
    if (!s || max_buffered == 0) return fail(ODDIO_HIP_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->d_bstatic) return max_buffered <= s->max_buffered ? 0 : fail(ODDIO_HIP_ESTATE, "buffered capacity already fixed at %u", s->max_buffered);
    return ensure_buffered_locked(s, max_buffered);
}

static int scene_play_buffered_impl(oddio_hip_scene* s, int leaf_kind, oddio_hip_frames* frames, double start_seconds, float phase,
                                    float freq_hz_or_value, const oddio_hip_filter* filters, int n_filters,
                                    const float position[3], const float velocity[3], float radius, float max_distance,
                                    uint32_t rate, float buffer_duration, uint32_t* source_id, bool fader = false);

extern "C" int oddio_hip_scene_play_buffered(oddio_hip_scene* s, int leaf_kind, oddio_hip_frames* frames, double start_seconds, float phase,
                                             float freq_hz_or_value, const oddio_hip_filter* filters, int n_filters,
                                             const float position[3], const float velocity[3], float radius, float max_distance,
                                             uint32_t rate, float buffer_duration, uint32_t* source_id) {
    if (leaf_kind == (int)KIND_STREAM) return fail(ODDIO_HIP_EINVAL, "streams are played with oddio_hip_scene_play_buffered_stream");
    return scene_play_buffered_impl(s, leaf_kind, frames, start_seconds, phase, freq_hz_or_value, filters, n_filters, position, velocity, radius,
                                    max_distance, rate, buffer_duration, source_id);
}

extern "C" int oddio_hip_scene_play_buffered_stream(oddio_hip_scene* s, oddio_hip_stream* stream, const oddio_hip_filter* filters, int n_filters,
                                                    const float position[3], const float velocity[3], float radius, float max_distance,
                                                    uint32_t rate, float buffer_duration, uint32_t* source_id) {
    if (!s || !stream) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    if (stream->played) return fail(ODDIO_HIP_ESTATE, "a Stream is moved into the scene once (stream.rs:24-34)");
    if (stream->channels != 1) return fail(ODDIO_HIP_EINVAL, "spatial scenes take mono streams (Frame = Sample, spatial.rs:291)");
    int rc = scene_play_buffered_impl(s, (int)KIND_STREAM, stream->ring, 0.0, 0.0f, 0.0f, filters, n_filters, position, velocity, radius,
                                      max_distance, rate, buffer_duration, source_id);
    if (!rc) stream->played = true;
    return rc;
}

// play_buffered(Fader::new(chain).1, ..) (fader.rs:16-28) and FaderControl::fade_to (:83-93) for a buffered source
extern "C" int oddio_hip_scene_play_buffered_fader(oddio_hip_scene* s, int leaf_kind, oddio_hip_frames* frames, double start_seconds, float phase,
                                                   float freq_hz_or_value, const oddio_hip_filter* filters, int n_filters,
                                                   const float position[3], const float velocity[3], float radius, float max_distance,
                                                   uint32_t rate, float buffer_duration, uint32_t* source_id) {
    if (leaf_kind == (int)KIND_STREAM) return fail(ODDIO_HIP_EINVAL, "Fader<..Stream..> is not implemented");
    return scene_play_buffered_impl(s, leaf_kind, frames, start_seconds, phase, freq_hz_or_value, filters, n_filters, position, velocity, radius,
                                    max_distance, rate, buffer_duration, source_id, true);
}

static int scene_build_signal(oddio_hip_scene* s, int leaf_kind, oddio_hip_frames*& frames, double start_seconds, float phase,
                              float freq_hz_or_value, const oddio_hip_filter* filters, int n_filters, BufStatic& st, BufDyn& d);

extern "C" int oddio_hip_source_fade_to(oddio_hip_scene* s, uint32_t id, int leaf_kind, oddio_hip_frames* frames, double start_seconds, float phase,
                                        float freq_hz_or_value, const oddio_hip_filter* filters, int n_filters, float duration) {
    if (!s) return fail(ODDIO_HIP_EINVAL, "NULL scene");
    if (leaf_kind == (int)KIND_STREAM) return fail(ODDIO_HIP_EINVAL, "fading to a Stream is not implemented");
    if (!(duration > 0.0f)) return fail(ODDIO_HIP_EINVAL, "fade_to: duration must be > 0");
    FaderPending cmd = {};
    int rc = scene_build_signal(s, leaf_kind, frames, start_seconds, phase, freq_hz_or_value, filters, n_filters, cmd.st, cmd.dyn);
    if (rc) return rc;
    cmd.duration = duration;
    cmd.fresh = 1u;
    std::lock_guard<std::mutex> lk(s->mu);
    if (id >= s->handles.size() || !s->handles[id].fader) return fail(ODDIO_HIP_ESTATE, "source %u is not a Fader", id);
    if (frames) { oddio_hip_frames_retain(frames); s->handles[id].fade_frames.push_back(frames); }
    s->pending_fades.emplace_back(s->handles[id].fader, cmd);
    return 0;
}

// leaf + filters (innermost first) of a mono signal for the thread-per-source paths of a scene
static int scene_build_signal(oddio_hip_scene* s, int leaf_kind, oddio_hip_frames*& frames, double start_seconds, float phase,
                              float freq_hz_or_value, const oddio_hip_filter* filters, int n_filters, BufStatic& st, BufDyn& d) {
    if (n_filters < 0 || n_filters > MAX_WRAP || (n_filters && !filters)) return fail(ODDIO_HIP_EINVAL, "0..%d filters", MAX_WRAP);
    st = BufStatic{};
    d = BufDyn{};
    if (leaf_kind == (int)KIND_FRAMES) {
        if (!frames) return fail(ODDIO_HIP_EINVAL, "frames is NULL");
        if (frames->device != s->device) return fail(ODDIO_HIP_EINVAL, "frames live on another device");
        if (frames->channels != 1) return fail(ODDIO_HIP_EINVAL, "spatial scenes take mono clips");
        st.clip = frames->dev; st.clip_len = (uint32_t)frames->len; st.clip_rate = frames->rate;
        d.common.t = start_seconds;
    } else if (leaf_kind == (int)KIND_CYCLE) {
        if (!frames || frames->device != s->device || frames->channels != 1) return fail(ODDIO_HIP_EINVAL, "Cycle needs a mono clip on the scene device");
        st.clip = frames->dev; st.clip_len = (uint32_t)frames->len; st.clip_rate = frames->rate;
        d.common.t = 0.0;   // Cycle::new (cycle.rs:17-23)
    } else if (leaf_kind == (int)KIND_STREAM) {
        if (!frames || frames->device != s->device) return fail(ODDIO_HIP_EINVAL, "the stream lives on another device");
        st.clip = frames->dev; st.clip_len = (uint32_t)frames->len; st.clip_rate = frames->rate;   // ring of capacity + 1 slots
        d.common.phase = 0.0f;   // Stream::t (stream.rs:30)
    } else if (leaf_kind == (int)KIND_SINE) {
        st.freq_or_value = freq_hz_or_value * ODDIO_TAU;   // sine.rs:21
        d.common.phase = phase;
        frames = nullptr;
    } else if (leaf_kind == (int)KIND_CONSTANT) {
        st.freq_or_value = freq_hz_or_value;
        frames = nullptr;
    } else {
        return fail(ODDIO_HIP_EINVAL, "unknown leaf kind %d", leaf_kind);
    }
    st.kind = (uint32_t)leaf_kind;
    st.channels = 1;
    st.n_wrap = (uint32_t)n_filters;
    for (int w = 0; w < n_filters; ++w) {
        st.wrap_kind[w] = (uint32_t)filters[w].kind;
        d.shared[w] = 1.0f; d.sm_prev[w] = 1.0f; d.sm_next[w] = 1.0f; d.sm_progress[w] = 1.0f;
        switch (filters[w].kind) {
        case ODDIO_HIP_FILTER_FIXED_GAIN: st.wrap_param[w] = powf(10.0f, filters[w].param / 20.0f); break;                 // gain.rs:20
        case ODDIO_HIP_FILTER_GAIN: d.shared[w] = filters[w].param; d.sm_prev[w] = d.sm_next[w] = filters[w].param; break;    // Gain::set_amplitude_ratio, gain.rs:90-93
        case ODDIO_HIP_FILTER_SPEED: d.shared[w] = filters[w].param; break;                                                    // speed.rs:18
        default: return fail(ODDIO_HIP_EINVAL, "unknown filter kind %d", filters[w].kind);
        }
    }
    return 0;
}

static int scene_play_buffered_impl(oddio_hip_scene* s, int leaf_kind, oddio_hip_frames* frames, double start_seconds, float phase,
                                    float freq_hz_or_value, const oddio_hip_filter* filters, int n_filters,
                                    const float position[3], const float velocity[3], float radius, float max_distance,
                                    uint32_t rate, float buffer_duration, uint32_t* source_id, bool fader) {
    if (!s || !position || !velocity) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    if (rate == 0) return fail(ODDIO_HIP_EINVAL, "rate must be > 0");
    BufStatic st;
    BufDyn d;
    int brc = scene_build_signal(s, leaf_kind, frames, start_seconds, phase, freq_hz_or_value, filters, n_filters, st, d);
    if (brc) return brc;
    // SpatialSignalBuffered::new (spatial.rs:31-56)
    const float max_delay = max_distance / ODDIO_SPEED_OF_SOUND + buffer_duration;
    const float cap_f = ceilf(max_delay * (float)rate);
    if (!(cap_f >= 0.0f) || cap_f > 2.0e8f) return fail(ODDIO_HIP_EINVAL, "ring of %g samples is out of range", (double)cap_f);
    const uint32_t ring_len = (uint32_t)cap_f + 1u;
    st.ring_len = ring_len; st.rate = rate; st.max_delay = max_delay; st.radius = radius;
    {   // queue.delay(rate, min(|position| / c, max_delay))  (ring.rs:45-47)
        float n2 = 0.0f;
        n2 = n2 + position[0] * position[0]; n2 = n2 + position[1] * position[1]; n2 = n2 + position[2] * position[2];
        const float dist = sqrtf(n2);
        const float dt = fminf(dist / ODDIO_SPEED_OF_SOUND, max_delay);
        d.ring_write = fmodf(0.0f + (float)rate * dt, (float)ring_len);
    }
    for (int k = 0; k < 3; ++k) { d.common.tgt_pos[k] = position[k]; d.common.tgt_vel[k] = velocity[k]; d.common.prev_pos[k] = position[k]; }
    std::lock_guard<std::mutex> lk(s->mu);
    int rc = ensure_buffered_locked(s, (uint32_t)s->live_count_b + 1);
    if (rc) return rc;
    if (s->live_count_b >= s->max_buffered) return fail(ODDIO_HIP_ENOMEM, "buffered set is full (%u)", s->max_buffered);
    {   // control thread: allocate the ring (zeroed, Ring::new ring.rs:10-15) and free rings of removed sources
        DeviceGuard g(s->device);
        for (float* r : s->ring_garbage) (void)hipFree(r);
        s->ring_garbage.clear();
        hipError_t e = hipMalloc(&st.ring, (size_t)ring_len * sizeof(float));
        if (e != hipSuccess) return fail(ODDIO_HIP_ENOMEM, "hipMalloc(ring of %u floats): %s", ring_len, hipGetErrorString(e));
        e = hipMemset(st.ring, 0, (size_t)ring_len * sizeof(float));
        if (e != hipSuccess) { (void)hipFree(st.ring); return fail((int)e, "hipMemset(ring): %s", hipGetErrorString(e)); }
    }
    uint32_t fader_tag = 0;
    if (fader) {   // Fader::new(signal): record with progress 1.0 (fader.rs:21); control thread, never inside sample()
        DeviceGuard g(s->device);
        if (!s->d_faders) {
            hipError_t e = hipMalloc(&s->d_faders, SCENE_FADER_CAP * sizeof(FaderRec));
            if (e == hipSuccess) e = hipMalloc(&s->d_fader_scratch, (size_t)SCENE_FADER_CAP * FADER_BUF * sizeof(float));
            if (e != hipSuccess) { (void)hipFree(s->d_faders); s->d_faders = nullptr; (void)hipFree(st.ring); return fail(ODDIO_HIP_ENOMEM, "hipMalloc(fader records): %s", hipGetErrorString(e)); }
        }
        if (s->fader_count >= SCENE_FADER_CAP) { (void)hipFree(st.ring); return fail(ODDIO_HIP_ENOMEM, "a scene holds at most %u Fader sources", SCENE_FADER_CAP); }
        FaderRec rec = {};
        rec.progress = 1.0f;
        hipError_t e = hipMemcpy(s->d_faders + s->fader_count, &rec, sizeof(rec), hipMemcpyHostToDevice);   // not referenced by any slot yet
        if (e != hipSuccess) { (void)hipFree(st.ring); return fail((int)e, "hipMemcpy(fader record): %s", hipGetErrorString(e)); }
        fader_tag = ++s->fader_count;
        st.fader = fader_tag;
    }
    s->live_count_b++;
    const uint32_t id = alloc_id_locked(s);
    d.common.id = id;
    HandleRec& h = s->handles[id];
    h.fader = fader_tag;
    h.queued = true; h.buffered = true; h.frames = frames; h.ring = st.ring;
    if (frames) oddio_hip_frames_retain(frames);
    s->pending_plays_b.push_back({id, st, d});
    if (source_id) *source_id = id;
    return 0;
}

static int push_control(oddio_hip_scene* s, uint32_t id, int filter_index, float value) {
    if (!s) return fail(ODDIO_HIP_EINVAL, "NULL scene");
    if (filter_index < 0 || filter_index >= MAX_WRAP) return fail(ODDIO_HIP_EINVAL, "bad filter index");
    std::lock_guard<std::mutex> lk(s->mu);
    if (id >= s->handles.size() || s->handles[id].released || !s->handles[id].buffered) return fail(ODDIO_HIP_ESTATE, "source %u is not a buffered source", id);
    s->pending_controls.push_back({id, (uint32_t)filter_index, value});
    return 0;
}
extern "C" int oddio_hip_source_set_gain(oddio_hip_scene* s, uint32_t id, int filter_index, float amplitude_ratio) {
    return push_control(s, id, filter_index, amplitude_ratio);                              // gain.rs:158-160
}
extern "C" int oddio_hip_source_set_gain_db(oddio_hip_scene* s, uint32_t id, int filter_index, float db) {
    return push_control(s, id, filter_index, powf(10.0f, db / 20.0f));                       // gain.rs:141-143
}
extern "C" int oddio_hip_source_set_speed(oddio_hip_scene* s, uint32_t id, int filter_index, float factor) {
    return push_control(s, id, filter_index, factor);                                       // speed.rs:52-54
}

// ---- the audio-thread side ------------------------------------------------------------------

// Applies one harvested list of stopped ids: Set::remove == Vec::swap_remove in the walk's
// descending slot order (spatial.rs:204,258-261; set.rs:183-188).
static int apply_stopped(oddio_hip_scene* s, std::vector<uint32_t>& ids) {
    if (ids.empty()) return 0;
    std::vector<uint32_t> slots_of[2];   // [0] seekable set, [1] buffered set
    {
        std::lock_guard<std::mutex> lk(s->mu);
        for (uint32_t id : ids) {
            if (id >= s->handles.size()) continue;
            HandleRec& h = s->handles[id];
            if (!h.in_set) continue;
            slots_of[h.buffered ? 1 : 0].push_back(h.slot);
        }
    }
    for (int set = 0; set < 2; ++set) {
        std::vector<uint32_t>& slots = slots_of[set];
        if (slots.empty()) continue;
        std::vector<uint32_t>& id_of_slot = set ? s->id_of_slot_b : s->id_of_slot;
        std::sort(slots.begin(), slots.end(), std::greater<uint32_t>());
        // simulate the swap_removes on the id map, tracking where each surviving element started
        std::unordered_map<uint32_t, uint32_t> origin;   // current slot -> original slot of its (moved) occupant
        uint32_t len = set ? s->len_b : s->len;
        std::vector<uint32_t> removed_ids;
        for (uint32_t slot : slots) {
            removed_ids.push_back(id_of_slot[slot]);
            const uint32_t last = len - 1;
            if (slot != last) {
                auto it = origin.find(last);
                origin[slot] = it == origin.end() ? last : it->second;
                id_of_slot[slot] = id_of_slot[last];
            }
            origin.erase(last);
            len--;
        }
        std::vector<SlotMove> moves;
        for (auto& kv : origin) if (kv.first < len) moves.push_back({kv.first, kv.second});
        {
            std::lock_guard<std::mutex> lk(s->mu);
            for (uint32_t id : removed_ids) {
                HandleRec& h = s->handles[id];
                h.in_set = false; h.finished = true; h.slot = 0xffffffffu;
                if (set) s->live_count_b--; else s->live_count--;
                if (h.frames) { oddio_hip_frames_release(h.frames); h.frames = nullptr; }
                if (h.ring) { s->ring_garbage.push_back(h.ring); h.ring = nullptr; }   // freed by the control thread / destroy
                if (h.released) s->free_ids.push_back(id);
            }
            for (const SlotMove& m : moves) s->handles[id_of_slot[m.dst]].slot = m.dst;
        }
        if (set) s->len_b = len; else s->len = len;
        if (!moves.empty()) {
            // sources move from original (surviving) slots into stopped slots: reads and writes are disjoint
            static_assert(sizeof(SlotMove) == sizeof(BufMove), "move records share a staging buffer layout");
            void* d_mv = set ? (void*)s->d_bmoves : (void*)s->d_moves;
            HIP_TRY(hipMemcpyAsync(d_mv, moves.data(), moves.size() * sizeof(SlotMove), hipMemcpyHostToDevice, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));   // `moves` is pageable host memory
            const uint32_t n = (uint32_t)moves.size();
            if (set)
                hipLaunchKernelGGL(apply_buf_moves, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->d_bmoves, n, s->d_bstatic, s->d_bdyn, s->d_bpend);
            else
                hipLaunchKernelGGL(apply_slot_moves, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->d_moves, n, s->d_static, s->d_dyn, s->d_pend);
            HIP_TRY(hipGetLastError());
        }
    }
    return 0;
}

static int harvest_ring(oddio_hip_scene* s, int r) {
    if (!s->ring_busy[r]) return 0;
    HIP_TRY(hipEventSynchronize(s->ev_stopped[r]));
    s->ring_busy[r] = false;
    uint32_t count = s->h_stopped[r][0];
    if (count == 0) return 0;
    std::vector<uint32_t> ids;
    if (count <= STOPPED_CAP) {
        ids.assign(s->h_stopped[r] + 1, s->h_stopped[r] + 1 + count);
    } else {
        // mass removal: more ids than the inline list holds -> read the flags of every slot
        HIP_TRY(hipStreamSynchronize(s->stream));
        std::vector<SrcDyn> dyn(s->ring_nsrc[r]);
        HIP_TRY(hipMemcpy(dyn.data(), s->d_dyn, dyn.size() * sizeof(SrcDyn), hipMemcpyDeviceToHost));
        for (const SrcDyn& d : dyn) if (d.flags & DYN_STOPPED) ids.push_back(d.id);
        if (s->len_b) {
            std::vector<BufDyn> bd(s->len_b);
            HIP_TRY(hipMemcpy(bd.data(), s->d_bdyn, bd.size() * sizeof(BufDyn), hipMemcpyDeviceToHost));
            for (const BufDyn& d : bd) if (d.common.flags & DYN_STOPPED) ids.push_back(d.common.id);
        }
    }
    return apply_stopped(s, ids);
}

static int scene_sample_impl(oddio_hip_scene* s, float interval, float* host_out, float* dev_out, size_t n_frames) {
    if (!s) return fail(ODDIO_HIP_EINVAL, "NULL scene");
    if (n_frames > s->max_frames) return fail(ODDIO_HIP_ENOMEM, "n_frames %zu > max_frames %u", n_frames, s->max_frames);
    if (n_frames && !host_out && !dev_out) return fail(ODDIO_HIP_EINVAL, "out is NULL");
    DeviceGuard g(s->device);
    if (!g.ok) return fail(ODDIO_HIP_ENODEV, "hipSetDevice(%d) failed", s->device);
    const int r = (int)(s->call_index & 1);
    s->call_index++;
    int rc = harvest_ring(s, r);   // the list this ring slot still holds (two calls ago in device mode)
    if (rc) return rc;

    // ---- set.update(): drain control messages (set.rs:141-168) ----
    bool motion_applied = false;   // a seekable source got a fresh Motion this callback
    std::vector<PendingPlay> plays;
    std::vector<PendingPlayB> plays_b;
    std::vector<PendingControl> controls;
    std::vector<PendingMotion>& motions = s->motion_scratch;
    motions.clear();
    bool rot_fresh;
    float rot_new[4];
    {
        std::lock_guard<std::mutex> lk(s->mu);
        plays.swap(s->pending_plays);
        plays_b.swap(s->pending_plays_b);
        controls.swap(s->pending_controls);
        motions.swap(s->pending_motion);
        rot_fresh = s->rot_fresh;
        s->rot_fresh = false;
        memcpy(rot_new, s->rot_pending, sizeof(rot_new));
    }
    if (!plays.empty()) {
        const uint32_t first = s->len;
        const size_t k = plays.size();
        if (first + k > s->max_sources) return fail(ODDIO_HIP_ENOMEM, "scene overflow");
        std::vector<SrcStatic> hs(k);
        std::vector<SrcDyn> hd(k);
        {
            std::lock_guard<std::mutex> lk(s->mu);
            for (size_t i = 0; i < k; ++i) {
                hs[i] = plays[i].st; hd[i] = plays[i].dyn;
                if (hs[i].kind == KIND_CYCLE) s->cycle_live++;
                const uint32_t slot = first + (uint32_t)i;
                s->id_of_slot[slot] = plays[i].id;
                HandleRec& h = s->handles[plays[i].id];
                h.slot = slot; h.in_set = true; h.queued = false;
            }
        }
        HIP_TRY(hipMemcpyAsync(s->d_static + first, hs.data(), k * sizeof(SrcStatic), hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(s->d_dyn + first, hd.data(), k * sizeof(SrcDyn), hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemsetAsync(s->d_pend + first, 0, k * sizeof(SrcPending), s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));   // hs/hd are pageable and go out of scope
        s->len = first + (uint32_t)k;
    }
    if (!plays_b.empty()) {   // buffered set: set.update() pushes in send order too
        const uint32_t first = s->len_b;
        const size_t k = plays_b.size();
        std::vector<BufStatic> hs(k);
        std::vector<BufDyn> hd(k);
        {
            std::lock_guard<std::mutex> lk(s->mu);
            for (size_t i = 0; i < k; ++i) {
                hs[i] = plays_b[i].st; hd[i] = plays_b[i].dyn;
                const uint32_t slot = first + (uint32_t)i;
                s->id_of_slot_b[slot] = plays_b[i].id;
                HandleRec& h = s->handles[plays_b[i].id];
                h.slot = slot; h.in_set = true; h.queued = false;
            }
        }
        HIP_TRY(hipMemcpyAsync(s->d_bstatic + first, hs.data(), k * sizeof(BufStatic), hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(s->d_bdyn + first, hd.data(), k * sizeof(BufDyn), hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemsetAsync(s->d_bpend + first, 0, k * sizeof(SrcPending), s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        s->len_b = first + (uint32_t)k;
    }
    if (!controls.empty()) {   // GainControl / SpeedControl: relaxed "latest value" stores (gain.rs:158-160, speed.rs:52-54)
        std::vector<ControlUpdate> ups;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            for (const PendingControl& c : controls) {
                const HandleRec& h = s->handles[c.id];
                if (!h.in_set || !h.buffered) continue;
                ups.push_back({h.slot, c.index, c.value, 0u});
            }
        }
        if (!ups.empty()) {
            HIP_TRY(hipMemcpyAsync(s->d_ctrl, ups.data(), ups.size() * sizeof(ControlUpdate), hipMemcpyHostToDevice, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
            const uint32_t n = (uint32_t)ups.size();
            // updates of one callback are applied in send order by a single thread block per 256; a later
            // value for the same (slot, index) must win -> serialise them
            hipLaunchKernelGGL(apply_control_updates_serial, dim3(1), dim3(1), 0, s->stream, s->d_ctrl, n, s->d_bdyn);
            HIP_TRY(hipGetLastError());
        }
    }
    {   // FaderControl::fade_to commands: swap.rs keeps only the latest flushed one per Fader
        std::vector<std::pair<uint32_t, FaderPending>> fades;
        { std::lock_guard<std::mutex> lk(s->mu); fades.swap(s->pending_fades); }
        if (!fades.empty()) {
            for (size_t i = 0; i < fades.size(); ++i) {
                bool superseded = false;
                for (size_t j = i + 1; j < fades.size(); ++j) superseded = superseded || fades[j].first == fades[i].first;
                if (superseded) continue;
                HIP_TRY(hipMemcpyAsync(&s->d_faders[fades[i].first - 1u].pend, &fades[i].second, sizeof(FaderPending), hipMemcpyHostToDevice, s->stream));
            }
            HIP_TRY(hipStreamSynchronize(s->stream));   // `fades` is pageable host memory
        }
    }
    if (!motions.empty()) {
        // swap.rs semantics: only the latest value per source survives until refresh().  Seekable
        // sources' updates are written straight into the pinned staging buffer (front), buffered
        // ones behind them (back), and go to the device with asynchronous copies.
        if (s->motion_copy_pending) { HIP_TRY(hipEventSynchronize(s->ev_motion)); s->motion_copy_pending = false; }
        size_t n_seek = 0;
        std::vector<MotionUpdate> ups_b;                     // buffered sources: few, keep the simple path
        {
            std::lock_guard<std::mutex> lk(s->mu);
            s->motion_epoch++;
            for (size_t i = motions.size(); i-- > 0;) {      // newest first: the latest value per source wins
                HandleRec& h = s->handles[motions[i].id];
                if (h.motion_epoch == s->motion_epoch) continue;
                h.motion_epoch = s->motion_epoch;
                if (!h.in_set) continue;
                MotionUpdate tmp;
                MotionUpdate& u = h.buffered ? tmp : s->h_motion[n_seek++];   // distinct seekable handles <= max_sources
                u.slot = h.slot;
                for (int c = 0; c < 3; ++c) { u.pos[c] = motions[i].pos[c]; u.vel[c] = motions[i].vel[c]; }
                u.discontinuity = motions[i].disc;
                if (h.buffered) ups_b.push_back(tmp);
            }
        }
        if (!ups_b.empty()) {
            const uint32_t n = (uint32_t)ups_b.size();
            HIP_TRY(hipMemcpyAsync(s->d_bmotion, ups_b.data(), ups_b.size() * sizeof(MotionUpdate), hipMemcpyHostToDevice, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));        // pageable source
            hipLaunchKernelGGL(apply_motion_updates, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->d_bmotion, n, s->d_bpend);
            HIP_TRY(hipGetLastError());
        }
        if (n_seek) {
            motion_applied = true;
            const uint32_t n = (uint32_t)n_seek;
            HIP_TRY(hipMemcpyAsync(s->d_motion, s->h_motion, n_seek * sizeof(MotionUpdate), hipMemcpyHostToDevice, s->stream));
            hipLaunchKernelGGL(apply_motion_updates, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->d_motion, n, s->d_pend);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(s->ev_motion, s->stream));
            s->motion_copy_pending = true;
        }
    }

    // ---- listener rotation (spatial.rs:382-386) ----
    SceneParams P;
    memcpy(P.prev_rot, s->rot, sizeof(P.prev_rot));
    if (rot_fresh) memcpy(s->rot, rot_new, sizeof(s->rot));
    memcpy(P.rot, s->rot, sizeof(P.rot));
    P.interval = interval;
    P.elapsed = interval * (float)n_frames;   // spatial.rs:394
    P.n_frames = (uint32_t)n_frames;
    P.n_sources = s->len;
    P.cycle_rows = s->d_cycle_rows;
    P.cycle_plane = s->max_frames;
    P.pad = 0;

    float* out_dev = dev_out ? dev_out : s->d_out;
    const bool prof = s->profiling && !s->ev_prof.empty();
    hipEvent_t* pev = prof ? &s->ev_prof[(s->prof_calls % oddio_hip_scene::PROF_RING) * 4] : nullptr;
    if (prof) HIP_TRY(hipEventRecord(pev[0], s->stream));
    if (s->len > 0) {
        hipLaunchKernelGGL(spatial_prepass, dim3((s->len + 255) / 256), dim3(256), 0, s->stream, P, s->d_static, s->d_dyn, s->d_pend,
                           s->d_ear, s->d_stopped[r], STOPPED_CAP, motion_applied ? 1 : 0);
        HIP_TRY(hipGetLastError());
        if (s->cycle_live > 0) {
            hipLaunchKernelGGL(cycle_sources, dim3((s->len + 63) / 64), dim3(64), 0, s->stream, P, s->d_static, s->d_dyn, s->d_ear);
            HIP_TRY(hipGetLastError());
        }
    }
    const float* init = nullptr;
    if (s->len_b > 0) {
        hipLaunchKernelGGL(buffered_sources, dim3((s->len_b + 63) / 64), dim3(64), 0, s->stream, P, s->len_b, s->d_bstatic, s->d_bdyn, s->d_bpend,
                           s->d_contrib, s->d_bskip, s->d_stopped[r], STOPPED_CAP, s->d_faders, s->d_fader_scratch);
        HIP_TRY(hipGetLastError());
        if (n_frames > 0) {
            const uint32_t n_out = 2u * (uint32_t)n_frames;
            hipLaunchKernelGGL(buffered_reduce, dim3((n_out + 255) / 256), dim3(256), 0, s->stream, s->d_contrib, s->d_bskip, s->len_b,
                               (uint32_t)n_frames, s->d_outb);
            HIP_TRY(hipGetLastError());
            init = s->d_outb;
        }
    }
    if (prof) HIP_TRY(hipEventRecord(pev[1], s->stream));
    uint32_t n_wgs = 0;
    const uint32_t n_tiles = ((uint32_t)n_frames + TILE_FRAMES - 1) / TILE_FRAMES;
    if (n_frames > 0 && s->len > 0) {
        const uint32_t n_groups = (s->len + MIX_GROUP - 1) / MIX_GROUP;
        // one round of workgroups: the waves of ALL tiles together fill the chip once (waves_cap resident
        // waves); more, shorter workgroups only add partial tiles and per-workgroup overhead
        const uint32_t cap_tile = std::max<uint32_t>(MIX_WG_WAVES, s->waves_cap / std::max(1u, n_tiles));
        uint32_t waves = s->mode == ODDIO_HIP_MODE_ORDERED ? 1u : std::min(n_groups, cap_tile);
        const uint32_t gpw = (n_groups + waves - 1) / waves;
        waves = (n_groups + gpw - 1) / gpw;
        // whole workgroups of MIX_WG_WAVES independent waves (trailing waves get an empty range)
        n_wgs = (waves + MIX_WG_WAVES - 1) / MIX_WG_WAVES;
        const bool full = (n_frames % TILE_FRAMES) == 0;
        if (full)
            hipLaunchKernelGGL(spatial_mix<true>, dim3(n_wgs, n_tiles), dim3(64 * MIX_WG_WAVES), 0, s->stream, P, s->d_static, s->d_ear, s->d_partials, init, gpw, n_groups);
        else
            hipLaunchKernelGGL(spatial_mix<false>, dim3(n_wgs, n_tiles), dim3(64 * MIX_WG_WAVES), 0, s->stream, P, s->d_static, s->d_ear, s->d_partials, init, gpw, n_groups);
        HIP_TRY(hipGetLastError());
    }
    if (prof) HIP_TRY(hipEventRecord(pev[2], s->stream));
    bool stopped_published = false;
    if (n_frames > 0) {
        const uint32_t n_out = 2u * (uint32_t)n_frames;
        // with Adapt the filter order is Reinhard(Adapt(scene)); in a sharded scene the filters follow the cross-GPU sum
        const int fused_postfx = (s->adapt.on || s->comm) ? 0 : s->postfx;
        if (n_wgs > 0) {
            hipLaunchKernelGGL(reduce_stage1, dim3(((uint32_t)n_frames + 31) / 32, RED_SPLIT), dim3(256), 0, s->stream, s->d_partials, s->d_stage1,
                               n_wgs, (uint32_t)n_frames);
            hipLaunchKernelGGL(reduce_stage2, dim3((n_out + 255) / 256), dim3(256), 0, s->stream, s->d_stage1, out_dev, n_wgs,
                               (uint32_t)n_frames, fused_postfx, s->d_stopped[r], s->hd_stopped[r], STOPPED_CAP);
            stopped_published = true;
        } else if (init) {
            hipLaunchKernelGGL(copy_postfx_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s->stream, init, out_dev, n_out, fused_postfx);
        } else {
            hipLaunchKernelGGL(zero_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s->stream, out_dev, n_out);   // spatial.rs:389-391
        }
        HIP_TRY(hipGetLastError());
        if (s->comm) {
            // the shard's partial buffer -> the scene's mix: one 8 KiB sum over xGMI, same stream, no host sync
            const ncclResult_t nr = rccl_api()->AllReduce(out_dev, out_dev, n_out, ncclFloat32, ncclSum, s->comm, s->stream);
            if (nr != ncclSuccess) return fail(ODDIO_HIP_ENODEV, "ncclAllReduce: %s", rccl_api()->GetErrorString(nr));
            if (!s->adapt.on && s->postfx) {
                hipLaunchKernelGGL(postfx_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s->stream, out_dev, n_out, s->postfx);
                HIP_TRY(hipGetLastError());
            }
        }
        if (s->adapt.on) { rc = adapt_launch(s->adapt, s->stream, interval, out_dev, n_frames, s->postfx); if (rc) return rc; }
    }
    if (prof) { HIP_TRY(hipEventRecord(pev[3], s->stream)); s->prof_calls++; }

    // ---- results back ----
    if (!stopped_published) {   // paths without a reduce (empty Seek set, zero frames)
        hipLaunchKernelGGL(publish_stopped, dim3(1), dim3(256), 0, s->stream, s->d_stopped[r], s->hd_stopped[r], STOPPED_CAP);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(s->ev_stopped[r], s->stream));
    s->ring_busy[r] = true;
    s->ring_nsrc[r] = s->len;
    if (host_out || !dev_out) {
        if (n_frames > 0 && host_out) {
            HIP_TRY(hipMemcpyAsync(s->h_out, out_dev, 2 * n_frames * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        }
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (n_frames > 0 && host_out) memcpy(host_out, s->h_out, 2 * n_frames * sizeof(float));
        // host mode: removals take effect within the call, like the reference's walk
        rc = harvest_ring(s, r ^ 1);
        if (rc) return rc;
        rc = harvest_ring(s, r);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int oddio_hip_scene_sample(oddio_hip_scene* s, float interval, float* out, size_t n_frames) {
    return scene_sample_impl(s, interval, out, nullptr, n_frames);
}
extern "C" int oddio_hip_scene_run(oddio_hip_scene* s, uint32_t sample_rate, float* out, size_t n_frames) {
    const float interval = 1.0f / (float)sample_rate;   // lib.rs:91
    return scene_sample_impl(s, interval, out, nullptr, n_frames);
}
extern "C" int oddio_hip_scene_sample_device(oddio_hip_scene* s, float interval, float* dev_out, size_t n_frames) {
    if (!dev_out && n_frames) return fail(ODDIO_HIP_EINVAL, "dev_out is NULL");
    return scene_sample_impl(s, interval, nullptr, dev_out, n_frames);
}

extern "C" int oddio_hip_postfx_device(int device, int postfx, float* dev_buf, size_t n_frames, void* stream) {
    if (!dev_buf || postfx < 0 || postfx > 2) return fail(ODDIO_HIP_EINVAL, "bad argument");
    if (postfx == 0 || n_frames == 0) return 0;
    DeviceGuard g(device);
    const uint32_t n = 2u * (uint32_t)n_frames;
    hipLaunchKernelGGL(postfx_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dev_buf, n, postfx);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int oddio_hip_scene_seek_all(oddio_hip_scene* s, float seconds) {
    if (!s) return fail(ODDIO_HIP_EINVAL, "NULL scene");
    if (s->len == 0) return 0;
    DeviceGuard g(s->device);
    hipLaunchKernelGGL(seek_all_kernel, dim3((s->len + 255) / 256), dim3(256), 0, s->stream, s->d_dyn, s->d_static, s->len, seconds);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int oddio_hip_source_playback_position(oddio_hip_scene* s, uint32_t id, double* seconds) {
    if (!s || !seconds) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    uint32_t slot;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (id >= s->handles.size() || !s->handles[id].in_set) return fail(ODDIO_HIP_ESTATE, "source %u is not in the set", id);
        slot = s->handles[id].slot;
    }
    DeviceGuard g(s->device);
    HIP_TRY(hipStreamSynchronize(s->stream));
    SrcDyn d;
    SrcStatic st;
    HIP_TRY(hipMemcpy(&d, s->d_dyn + slot, sizeof(d), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&st, s->d_static + slot, sizeof(st), hipMemcpyDeviceToHost));
    if (st.kind != KIND_FRAMES && st.kind != KIND_DOWNMIX) return fail(ODDIO_HIP_ESTATE, "source %u is not a FramesSignal", id);
    // frames.rs:199-200, :238-240: (t * rate) as isize, read back as isize as f64 / rate
    const double rate = (double)st.clip_rate;
    double sp = d.t * rate;
    sp = sp != sp ? 0.0 : std::trunc(sp);
    *seconds = sp / rate;
    return 0;
}

// Debug: what the runtime thinks about the mix kernel's residency (blocks of 64 threads per CU).
extern "C" int oddio_hip_debug_mix_occupancy(int device, int* blocks_per_cu, int* num_cus, int* vgprs, int* lds_bytes) {
    DeviceGuard g(device);
    if (!g.ok) return fail(ODDIO_HIP_ENODEV, "hipSetDevice(%d) failed", device);
    int nb = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spatial_mix<true>, 64 * MIX_WG_WAVES, 0));
    hipFuncAttributes fa;
    HIP_TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(spatial_mix<true>)));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (blocks_per_cu) *blocks_per_cu = nb;
    if (num_cus) *num_cus = prop.multiProcessorCount;
    if (vgprs) *vgprs = fa.numRegs;
    if (lds_bytes) *lds_bytes = (int)fa.sharedSizeBytes;
    return 0;
}

#include "mixer_host.inc"
