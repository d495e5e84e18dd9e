// oddio_hip.hip -- host side of libodd_hip.so: the scene/mixer objects and the C ABI of
// include/oddio_hip.h.  gfx950 only.  The host logic mirrors the reference's control plane at the
// boundary: `Set` insert/update/swap_remove (src/set.rs:55-66,141-188), the `swap` "latest value"
// hand-off for Motion / listener rotation (src/swap.rs:36-64), and SpatialScene::sample's
// prologue (src/spatial.rs:376-394).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>   // types only: the functions are resolved with dlopen on first use

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/oddio_hip.h"
#include "kernels.h"
#include "pair_kernels.h"
#include "mixer_kernels.h"
#include "buffered_kernels.h"
#include "buffered_fast.h"
#include "set_kernels.h"

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

// Set-up fills and device-to-device copies made by the control thread.  hipMemset of device memory (and a D2D hipMemcpy) returns before
// the fill has run, as their CUDA namesakes do, and the fill runs on the NULL stream -- which the scenes' and mixers' own streams
// (hipStreamNonBlocking) do not wait for.  Normally the fill is over long before the first kernel of a `sample` call gets to the memory;
// with another process keeping the GPU busy it is not: round 6's soak under two concurrent processes met sources that vanished for a
// callback (a late fill of `d_len` after the first insert), and once a memory fault (a late 0xff fill of the id -> slot table).
// Every such fill is waited for before the call that made it returns; these are control-thread paths, never inside `sample`.
static hipError_t memset_now(void* p, int value, size_t bytes) {
    const hipError_t e = hipMemset(p, value, bytes);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}
static hipError_t copy_now(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    const hipError_t e = hipMemcpy(dst, src, bytes, kind);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}

// The host-output calls wait for a ticket that the callback's last kernel stores in pinned memory (a D2H copy + hipStreamSynchronize
// costs 30-50 us of wake-up latency per callback, the ticket ~5).  The spin is bounded by what the caller's callbacks have been
// taking -- 8 x the longest recent wait, at least 2 ms, at most 200 ms -- after which the caller falls back to the stream
// synchronisation (a wedged device then surfaces as that call's error instead of 200 ms of a spinning core per callback).
// `ewma_us`: the caller's running estimate of its waits, updated here.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("isb" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}
static inline bool spin_for_ticket(const uint32_t* flag, uint32_t ticket, double* ewma_us) {
    const double budget_us = std::min(200000.0, std::max(2000.0, 8.0 * *ewma_us));
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    double us = 0.0;
    for (uint32_t it = 0;; ++it) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == ticket) { seen = true; break; }
        if ((it & 255u) == 255u) {
            us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us > budget_us) break;
        }
        cpu_relax();
    }
    if (seen) us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    *ewma_us = seen ? std::max(us, 0.9 * *ewma_us) : std::max(*ewma_us, budget_us);   // (decays slowly, jumps up at once)
    return seen;
}

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
    bool has_mono_sum = false;      // stereo clips: L + R sits behind the frames (device_types.h downmix_presum_offset)
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
    // a stereo clip: + the mono sum L + R behind the frames (device_types.h downmix_presum_offset), for Downmix sources in FAST mode
    const size_t mono_off = channels == 2 ? (size_t)oddio_hip::downmix_presum_offset((uint32_t)len) : 0;
    const size_t mono_padded = channels == 2 ? ((len + 3) & ~size_t(3)) + 4 : 0;
    const size_t total = channels == 2 ? mono_off + mono_padded : padded;
    DeviceGuard g(device);
    if (!g.ok) { delete f; return fail(ODDIO_HIP_ENODEV, "hipSetDevice(%d) failed", device); }
    hipError_t e = hipMalloc(&f->dev, total * sizeof(float));
    if (e != hipSuccess) { delete f; return fail(ODDIO_HIP_ENOMEM, "hipMalloc(%zu floats): %s", total, hipGetErrorString(e)); }
    e = memset_now(f->dev + (padded - 4), 0, (total - (padded - 4)) * sizeof(float));   // zero tail pad: S(i) = 0 for i >= len (and the mono sum's pads)
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
    hipError_t e = copy_now(f->dev, samples, len * sizeof(float), hipMemcpyHostToDevice);
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
    hipError_t e = copy_now(f->dev, interleaved, 2 * n_frames * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        // the mono sum (0.0 + L) + R, downmix.rs:27-29's channels().sum() on the frames themselves
        std::vector<float> mono(n_frames);
        for (size_t i = 0; i < n_frames; ++i) mono[i] = (0.0f + interleaved[2 * i]) + interleaved[2 * i + 1];
        e = copy_now(f->dev + oddio_hip::downmix_presum_offset((uint32_t)n_frames), mono.data(), n_frames * sizeof(float), hipMemcpyHostToDevice);
        f->has_mono_sum = e == hipSuccess;
    }
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
        hipError_t e = copy_now(f->dev, dev_samples, len * sizeof(float), hipMemcpyDeviceToDevice);
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
extern "C" int oddio_hip_frames_refcount(const oddio_hip_frames* f, int* count) {
    if (!f || !count) return fail(ODDIO_HIP_EINVAL, "NULL argument");
    *count = f->refs.load(std::memory_order_acquire);
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
extern "C" int oddio_hip_bounds_checked(void) {
#ifdef ODDIO_HIP_BOUNDS
    return 1;
#else
    return 0;
#endif
}
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
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;   // (optional: TRACKED mode of a sharded scene)
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;      // (optional: reporting only)
    ncclResult_t (*GetVersion)(int*) = nullptr;
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
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.handle, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.handle, "ncclCommCount"));
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(api.handle, "ncclGetVersion"));
    });
    return &api;
}
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
    HIP_TRY(copy_now(a.d_state, &avg_squared, sizeof(float), hipMemcpyHostToDevice));
    a.tau = tau; a.max_gain = max_gain; a.low = low; a.high = high;
    a.on = true;
    return 0;
}

// Launches the epilogue on `buf` (interleaved stereo, or mono frames with channels == 1; raw sum) and applies `postfx` after it.
static int adapt_launch(const AdaptHost& a, hipStream_t stream, float interval, float* buf, size_t n_frames, int postfx, int channels = 2) {
    if (n_frames == 0) return 0;
    AdaptParams A;
    A.alpha = 1.0f - expf(-interval / a.tau);                                                   // adapt.rs:70
    A.one_minus_alpha = 1.0f - A.alpha;
    A.max_gain = a.max_gain; A.low = a.low; A.high = a.high;
    if (channels == 1) hipLaunchKernelGGL(adapt_kernel<1>, dim3(1), dim3(256), 0, stream, buf, (uint32_t)n_frames, A, a.d_state, postfx);
    else hipLaunchKernelGGL(adapt_kernel<2>, dim3(1), dim3(256), 0, stream, buf, (uint32_t)n_frames, A, a.d_state, postfx);
    HIP_TRY(hipGetLastError());
    return 0;
}

#include "scene_host.inc"

extern "C" int oddio_hip_postfx_device(int device, int postfx, float* dev_buf, size_t n_frames, void* stream) {
    if (!dev_buf || postfx < 0 || postfx > 2) return fail(ODDIO_HIP_EINVAL, "bad argument");
    if (postfx == 0 || n_frames == 0) return 0;
    DeviceGuard g(device);
    const uint32_t n = 2u * (uint32_t)n_frames;
    hipLaunchKernelGGL(postfx_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dev_buf, n, postfx);
    HIP_TRY(hipGetLastError());
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
