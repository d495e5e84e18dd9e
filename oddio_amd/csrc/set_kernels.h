// set_kernels.h -- the reference's `Set` (src/set.rs) kept on the device.
//
// A scene's source tables are indexed by slot == position in the reference's `Set`.  All three
// mutations happen in stream order on the GPU, so that the audio thread never waits for the device
// and slot order is the reference's even when callbacks are enqueued back to back:
//   insert_sources   set.update(): `push` the sources played since the last callback, in send order,
//                    at the current end of the set (set.rs:141-168)
//   compact_set      set.remove(i) == Vec::swap_remove for every source the walk stopped, in the walk's
//                    descending slot order (spatial.rs:204,258-261; set.rs:183-188); publishes the
//                    removed handle ids to a ring in pinned host memory (the reference returns the boxes
//                    to the control thread through its `free` channel, set.rs:84-122) and the new length
//   *_by_id          control traffic addresses sources by handle id; the slot is looked up here
#pragma once
#include "buffered_kernels.h"

namespace oddio_hip {

constexpr uint32_t SLOT_INVALID = 0xffffffffu;
constexpr uint32_t SLOT_BUFFERED_BIT = 0x80000000u;   // slot_of_id: the source lives in the buffered set

__device__ __forceinline__ SrcDyn& dyn_common(SrcDyn& d) { return d; }
__device__ __forceinline__ SrcDyn& dyn_common(BufDyn& d) { return d.common; }

// What the device tells the host about a scene, in pinned host memory (written with system-scope stores).
struct SetPublish {
    volatile uint64_t len_and_inserted[2];   // per set (0 seek, 1 buffered): live length | cumulative inserts << 32
    volatile uint32_t removed_total[2];      // per set: entries ever written to its removed-id ring
    uint32_t reduce_timeouts;                // the P2P reduce gave up waiting for a rank (reported by the next sample call)
};

template <class S, class D>
__global__ void insert_sources(const S* __restrict__ src_st, const D* __restrict__ src_dyn, uint32_t k, S* __restrict__ st, D* __restrict__ dyn,
                               SrcPending* __restrict__ pend, const uint32_t* __restrict__ d_len, uint32_t* __restrict__ slot_of_id,
                               uint32_t set_bit) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const uint32_t slot = *d_len + i;
    const S s = src_st[i];
    D d = src_dyn[i];
    st[slot] = s;
    dyn[slot] = d;
    SrcPending p = {};
    pend[slot] = p;
    slot_of_id[dyn_common(d).id] = slot | set_bit;
}
// after insert_sources: d_len[0] += k, d_len[1] (cumulative inserts) += k
__global__ void bump_len(uint32_t* d_len, uint32_t k) { d_len[0] += k; d_len[1] += k; }

// Arguments of one set's compaction (by value into the kernels that run it).
template <class S, class D> struct CompactArgs {
    uint32_t* stopped_hdr;      // [0] = number of sources the walk stopped this callback, [1..] their handle ids (first `cap`;
    uint32_t cap;               //   when more stopped, every slot's flags are scanned instead)
    S* st; D* dyn; SrcPending* pend;
    uint32_t* d_len;            // [0] live length, [1] cumulative inserts, [2] cumulative removals, [3] inserts last published
    uint32_t* slot_of_id;
    uint32_t set_index;         // 0 seekable, 1 buffered
    uint32_t* removed_ring; uint32_t ring_mask;
    unsigned char* finished;
    SetPublish* pub;
};

// One thread block (any size).  Runs after every reader of this callback's slot layout.
template <class S, class D>
__device__ void compact_set_block(const CompactArgs<S, D>& A) {
    __shared__ uint32_t slots[4096];
    __shared__ uint32_t sorted[4096];
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t set_bit = A.set_index ? SLOT_BUFFERED_BIT : 0u;
    const uint32_t count = A.stopped_hdr[0];
    uint32_t len = A.d_len[0];
    uint32_t total = A.d_len[2];
    __syncthreads();
    // remove slot s (block-uniform): the last element moves into it (Vec::swap_remove)
    auto remove_slot = [&](uint32_t s) {
        const uint32_t last = len - 1u;
        const uint32_t id_removed = dyn_common(A.dyn[s]).id;
        __syncthreads();
        if (s != last) {
            const uint32_t* a; uint32_t* b;
            a = reinterpret_cast<const uint32_t*>(&A.st[last]); b = reinterpret_cast<uint32_t*>(&A.st[s]);
            for (uint32_t w = tid; w < sizeof(S) / 4; w += nt) b[w] = a[w];
            a = reinterpret_cast<const uint32_t*>(&A.dyn[last]); b = reinterpret_cast<uint32_t*>(&A.dyn[s]);
            for (uint32_t w = tid; w < sizeof(D) / 4; w += nt) b[w] = a[w];
            a = reinterpret_cast<const uint32_t*>(&A.pend[last]); b = reinterpret_cast<uint32_t*>(&A.pend[s]);
            for (uint32_t w = tid; w < sizeof(SrcPending) / 4; w += nt) b[w] = a[w];
            if (tid == 0) A.slot_of_id[dyn_common(A.dyn[last]).id] = s | set_bit;
        }
        if (tid == 0) {
            A.slot_of_id[id_removed] = SLOT_INVALID;
            A.removed_ring[total & A.ring_mask] = id_removed;
            A.finished[id_removed] = 1;
        }
        __threadfence_block();
        __syncthreads();   // the copies of this step before the next step's reads
        total++;
        len--;
    };
    if (count > 0 && count <= A.cap && count <= 4096u) {
        for (uint32_t i = tid; i < count; i += nt) slots[i] = A.slot_of_id[A.stopped_hdr[1 + i]] & ~SLOT_BUFFERED_BIT;
        __syncthreads();
        for (uint32_t i = tid; i < count; i += nt) {        // rank sort, descending (slots are distinct)
            const uint32_t v = slots[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < count; ++j) rank += slots[j] > v ? 1u : 0u;
            sorted[rank] = v;
        }
        __syncthreads();
        for (uint32_t i = 0; i < count; ++i) remove_slot(sorted[i]);
    } else if (count > 0) {
        // mass removal: the list overflowed, walk every slot in the walk's (descending) order
        for (uint32_t hi = len; hi > 0;) {
            const uint32_t base = hi >= nt ? hi - nt : 0u;
            const uint32_t s = base + tid;
            slots[tid] = (s < hi && (dyn_common(A.dyn[s]).flags & DYN_STOPPED)) ? 1u : 0u;
            __syncthreads();
            for (uint32_t t = hi - base; t-- > 0;)
                if (slots[t]) remove_slot(base + t);
            __syncthreads();
            hi = base;
        }
    }
    if (tid == 0) {
        const uint32_t inserted = A.d_len[1];
        if (count > 0 || inserted != A.d_len[3]) {           // nothing removed, nothing inserted: the host's view is current
            A.d_len[0] = len;
            A.d_len[2] = total;
            A.d_len[3] = inserted;
            A.stopped_hdr[0] = 0u;                           // re-armed for the next callback
            __threadfence_system();
            A.pub->removed_total[A.set_index] = total;
            A.pub->len_and_inserted[A.set_index] = (uint64_t)len | ((uint64_t)inserted << 32);
            __threadfence_system();
        }
    }
}

template <class S, class D>
__global__ __launch_bounds__(256) void compact_set(CompactArgs<S, D> A) { compact_set_block(A); }

// The callback's last kernel: the fixed-order sum of the workgroup partial tiles + Reinhard / Tanh (blocks
// [0, n_red)), and -- in two extra blocks that need nothing from the others -- set.remove() of what the walk
// stopped (spatial.rs:258-261), so that neither a second reduce stage nor compaction costs a launch of its own.
template <int SEGS = RED_SEGS>
__global__ __launch_bounds__(2 * RED_FRAMES * SEGS) void reduce_partials_compact(const float* __restrict__ partials, float* __restrict__ out, uint32_t n_wgs,
                                                                                uint32_t n_frames, int postfx, uint32_t n_red,
                                                                                CompactArgs<SrcStatic, SrcDyn> seek, CompactArgs<BufStatic, BufDyn> buf) {
    if (blockIdx.x == n_red) { compact_set_block(seek); return; }
    if (blockIdx.x == n_red + 1u) { compact_set_block(buf); return; }
    reduce_partials_body<SEGS>(partials, out, n_wgs, n_frames, postfx, blockIdx.x);
}

// Motion updates carry handle ids (spatial.rs:137-149: the handle, not the set position)
struct MotionById { uint32_t id; float pos[3]; float vel[3]; uint32_t discontinuity; };

__global__ void apply_motion_by_id(const MotionById* __restrict__ up, uint32_t n, const uint32_t* __restrict__ slot_of_id,
                                   SrcPending* __restrict__ pend, SrcPending* __restrict__ pend_b) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const MotionById u = up[i];
    if (u.id == 0xffffffffu) return;                        // sent for the id's previous owner (scene_host.inc)
    const uint32_t sl = slot_of_id[u.id];
    if (sl == SLOT_INVALID) return;                          // already removed
    SrcPending p;
    p.pos[0] = u.pos[0]; p.pos[1] = u.pos[1]; p.pos[2] = u.pos[2];
    p.vel[0] = u.vel[0]; p.vel[1] = u.vel[1]; p.vel[2] = u.vel[2];
    p.flags = PEND_FRESH | (u.discontinuity ? PEND_DISCONTINUITY : 0u);
    p.pad = 0;
    if (sl & SLOT_BUFFERED_BIT) pend_b[sl & ~SLOT_BUFFERED_BIT] = p; else pend[sl] = p;
}

// GainControl / SpeedControl values by handle id: one thread per update (the host has kept only the last value per
// (id, filter) of the callback, so they are independent)
struct ControlById { uint32_t id; uint32_t index; float value; uint32_t pad; };
__global__ void apply_control_by_id(const ControlById* __restrict__ up, uint32_t n, const uint32_t* __restrict__ slot_of_id,
                                    BufDyn* __restrict__ bdyn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ControlById u = up[i];
    if (u.id == 0xffffffffu) return;                        // sent for the id's previous owner
    const uint32_t sl = slot_of_id[u.id];
    if (sl == SLOT_INVALID || !(sl & SLOT_BUFFERED_BIT)) return;
    bdyn[sl & ~SLOT_BUFFERED_BIT].shared[u.index & (MAX_WRAP - 1)] = u.value;
}

// The same two updates with ids and values in DEVICE memory (oddio_hip_scene_set_control_device / _set_motion_device):
// an engine that computes gains or positions on the GPU hands them over without a trip through the host; one message
// per batch on the control queue, applied in message order.
__global__ void apply_control_dev(const uint32_t* __restrict__ ids, const float* __restrict__ values, uint32_t n, uint32_t index,
                                  const uint32_t* __restrict__ slot_of_id, uint32_t id_cap, BufDyn* __restrict__ bdyn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = ids[i];
    if (id >= id_cap) return;                                  // not a handle id of this scene: ignored (the arrays are the caller's)
    const uint32_t sl = slot_of_id[id];
    if (sl == SLOT_INVALID || !(sl & SLOT_BUFFERED_BIT)) return;
    bdyn[sl & ~SLOT_BUFFERED_BIT].shared[index] = values[i];
}
__global__ void apply_motion_dev(const uint32_t* __restrict__ ids, const float* __restrict__ pos, const float* __restrict__ vel, uint32_t n,
                                 uint32_t discontinuity, const uint32_t* __restrict__ slot_of_id, uint32_t id_cap, SrcPending* __restrict__ pend,
                                 SrcPending* __restrict__ pend_b) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = ids[i];
    if (id >= id_cap) return;
    const uint32_t sl = slot_of_id[id];
    if (sl == SLOT_INVALID) return;
    SrcPending p;
    p.pos[0] = pos[3 * i]; p.pos[1] = pos[3 * i + 1]; p.pos[2] = pos[3 * i + 2];
    p.vel[0] = vel[3 * i]; p.vel[1] = vel[3 * i + 1]; p.vel[2] = vel[3 * i + 2];
    p.flags = PEND_FRESH | (discontinuity ? PEND_DISCONTINUITY : 0u);
    p.pad = 0;
    if (sl & SLOT_BUFFERED_BIT) pend_b[sl & ~SLOT_BUFFERED_BIT] = p; else pend[sl] = p;
}

// ---------------------------------------------------------------------------------------------------------------
// Sharded scene, deterministic reduce (SURVEY.md 8e's alternative to the RCCL all-reduce): every peer writes its
// partial stereo buffer into a slab in rank 0's memory (opened through hipIpc: xGMI peer-to-peer writes, or plain
// device writes when the ranks share one GPU), rank 0 adds the partials in FIXED rank order
//     mix = ((p0 + p1) + p2) + ... + p(world-1)
// and publishes the result, which every peer copies back.  One 8 KiB write, one 8 KiB read per peer and callback.
// The slab is fine-grained memory; everything that crosses a process goes through system-scope atomics (the
// per-XCD L2s and a peer's caches are not coherent for plain accesses inside a kernel).
//   slab layout (floats after the header): [header 256 B: arrived[world] | done][result: stride][partial of rank 1..][world rows: TRACKED mode's totals]
// Waits are bounded (~2 s): a rank that never arrives must not hang the GPU; *err is set instead, and the host makes the
// failure sticky for the scene (every later sample call returns ODDIO_HIP_ESTATE until the group is destroyed and re-made:
// the ranks' source cursors are a callback apart by then, nothing inside the group can re-align them).  The ranks must
// have passed a barrier of the host program between reduce_init_p2p and the first sample call.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t P2P_HEADER_WORDS = 64;      // 256 bytes: arrived[r] at word r (r < 60), done at word 63
constexpr uint32_t P2P_MAX_WORLD = 60;
constexpr unsigned long long P2P_TIMEOUT_TICKS = 200000000ull;   // wall_clock64 ticks of 10 ns: 2 s

// waits until *flag has reached `epoch` (wrap-safe: a peer that is already a callback further on satisfies the wait, so one
// late rank costs one timeout, not one per callback)
__device__ __forceinline__ bool p2p_wait_equal(const uint32_t* flag, uint32_t epoch) {
    const unsigned long long t0 = wall_clock64();
    while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
        if (wall_clock64() - t0 > P2P_TIMEOUT_TICKS) return false;
        __builtin_amdgcn_s_sleep(16);
    }
    return true;
}

// peer (rank > 0): partial -> its slab row, then arrived[rank] = epoch.  One block.
__global__ __launch_bounds__(1024) void p2p_publish(const float* __restrict__ partial, uint32_t* slab, uint32_t stride, uint32_t rank, uint32_t n_out,
                                                    uint32_t epoch) {
    float* row = reinterpret_cast<float*>(slab + P2P_HEADER_WORDS) + (size_t)rank * stride;
    for (uint32_t i = threadIdx.x; i < n_out; i += blockDim.x) __hip_atomic_store(row + i, partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(slab + rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// rank 0: waits for every peer, adds in rank order (its own partial is `buf`), leaves the mix in `buf` and in the slab's
// result row, then done = epoch.  One block.
__global__ __launch_bounds__(1024) void p2p_sum(float* __restrict__ buf, uint32_t* slab, uint32_t stride, uint32_t world, uint32_t n_out, uint32_t epoch,
                                                uint32_t* __restrict__ err) {
    if (threadIdx.x == 0) {
        bool all = true;
        for (uint32_t r = 1; r < world; ++r) all = p2p_wait_equal(slab + r, epoch) && all;
        if (!all) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    float* rows = reinterpret_cast<float*>(slab + P2P_HEADER_WORDS);
    for (uint32_t i = threadIdx.x; i < n_out; i += blockDim.x) {
        float t = buf[i];
        for (uint32_t r = 1; r < world; ++r) t = t + __hip_atomic_load(rows + (size_t)r * stride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        buf[i] = t;
        __hip_atomic_store(rows + i, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);       // row 0 = the result
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(slab + (P2P_HEADER_WORDS - 1), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// peer: waits for done == epoch, copies the mix into its own buffer.  One block.
__global__ __launch_bounds__(1024) void p2p_fetch(float* __restrict__ buf, const uint32_t* slab, uint32_t n_out, uint32_t epoch, uint32_t* __restrict__ err) {
    if (threadIdx.x == 0 && !p2p_wait_equal(slab + (P2P_HEADER_WORDS - 1), epoch)) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    const float* res = reinterpret_cast<const float*>(slab + P2P_HEADER_WORDS);
    for (uint32_t i = threadIdx.x; i < n_out; i += blockDim.x) buf[i] = __hip_atomic_load(res + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// TRACKED mode in a p2p group: every rank > 0 leaves its first pass's total in its row of the slab's SECOND block of rows
// (rows world .. 2 world - 1: the first block is the final reduce's and is rewritten later in the same callback), then arrived[rank]
// = epoch; every rank waits for the ranks above it and adds their totals in descending rank order -- the value the reference's
// running sum has when its walk enters this rank's shard.  One block.  (The final reduce of the callback uses epoch + 1.)
__global__ __launch_bounds__(1024) void p2p_gather_base(const float* __restrict__ total, uint32_t* slab, uint32_t stride, uint32_t rank, uint32_t world,
                                                        uint32_t n_out, uint32_t epoch, float* __restrict__ base, uint32_t* __restrict__ err) {
    float* rows = reinterpret_cast<float*>(slab + P2P_HEADER_WORDS) + (size_t)world * stride;
    if (rank > 0) {
        for (uint32_t i = threadIdx.x; i < n_out; i += blockDim.x) __hip_atomic_store(rows + (size_t)rank * stride + i, total[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(slab + rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (base == nullptr) return;                       // (a rank whose callback is not a tracked one only contributes its total)
    if (threadIdx.x == 0) {
        bool all = true;
        for (uint32_t r = rank + 1u; r < world; ++r) all = p2p_wait_equal(slab + r, epoch) && all;
        if (!all) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_out; i += blockDim.x) {
        float b = 0.0f;
        bool first = true;
        for (uint32_t r = world; r-- > rank + 1u;) {
            const float v = __hip_atomic_load(rows + (size_t)r * stride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            b = first ? v : b + v; first = false;
        }
        base[i] = b;
    }
}

// Seek::seek on every live source (signal.rs:48-51)
__global__ void seek_all_live(SrcDyn* __restrict__ dyn, const SrcStatic* __restrict__ st, const uint32_t* __restrict__ d_len, float seconds) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d_len[0]) return;
    if (st[i].kind == KIND_FRAMES || st[i].kind == KIND_DOWNMIX) dyn[i].t = dyn[i].t + (double)seconds;   // frames.rs:211-213
    else if (st[i].kind == KIND_SINE) dyn[i].phase = fmodf(dyn[i].phase + seconds * st[i].freq_or_value, ODDIO_TAU);
    else if (st[i].kind == KIND_CYCLE) dyn[i].t = f64_rem_euclid(dyn[i].t + (double)seconds * (double)st[i].clip_rate, (double)st[i].clip_len);
}

// Host-output calls: the callback's frames into pinned host memory, then the call's ticket (system-scope release): the audio
// thread spins on the ticket instead of paying a D2H copy and a stream synchronisation.  One block.
__global__ __launch_bounds__(1024) void publish_out(const float* __restrict__ out_dev, float* host_out, uint32_t n, uint32_t* flag, uint32_t ticket) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) __hip_atomic_store(host_out + i, out_dev[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The Mixer's host-output call: the frames and the head of the stopped-id list ([0] = count, then ids) into pinned host memory, then
// the ticket.  One block.
__global__ __launch_bounds__(1024) void mixer_publish(const float* __restrict__ out_dev, float* host_out, uint32_t n, uint32_t* __restrict__ stopped,
                                                      uint32_t* host_stopped, uint32_t stopped_words, uint32_t* flag, uint32_t ticket) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) __hip_atomic_store(host_out + i, out_dev[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t cnt = stopped[0];
    const uint32_t words = 1u + (cnt < stopped_words - 1u ? cnt : stopped_words - 1u);
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) __hip_atomic_store(host_stopped + i, stopped[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        stopped[0] = 0u;          // the next callback's walk counts from zero (no memset on the stream)
        __hip_atomic_store(flag, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// (bench) FramesSignal::t = seconds for the FramesSignal leaves of the buffered set
__global__ void reset_buffered_clock(BufDyn* __restrict__ dyn, const BufStatic* __restrict__ st, const uint32_t* __restrict__ d_len, double seconds) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d_len[0]) return;
    if (st[i].kind == KIND_FRAMES) dyn[i].common.t = seconds;
}

}  // namespace oddio_hip
