// pair_kernels.h -- spatial_mix_pair: the Seek-set mix kernel of large FAST-mode scenes (gfx950, wave64).
//
// spatial_mix (kernels.h) renders a 512-frame tile per wavefront, both ears: every source's window crosses HBM -> LDS twice
// per 1024-frame callback (once per tile, 2.3 KB each, with the ~40 samples the two tiles' windows share read twice), and
// every wavefront issues the three DMA instructions of its own window.  Measured (profiles/r05_*): with all windows
// served from L2 the kernel takes 0.203 ms, from HBM 0.216-0.219; random 2.3 KB reads reach 5.4-5.6 TB/s on this part,
// 4.3 KB reads 5.8-5.9.
//
// Here a WORKGROUP of two wavefronts renders the whole callback (up to 1024 frames = four 256-frame chunks,
// spatial.rs:393,456) of its sources, wave 0 the left ear and wave 1 the right: the source's window -- every sample both
// ears touch in the callback, ~4.3 KB -- is staged ONCE, each wave issuing half of its 1 KiB DMA pieces, and an s_barrier
// per source hands the two halves over (and frees the other window buffer for the next source).  Lane l of a wave owns
// the 16 consecutive frames 16 l .. 16 l + 15 of its ear (chunk l >> 4, block l & 15): the same 16 register
// accumulators, checkpointed exact cursor scan (frames.rs:189-196) and per-sample loop (mix_source_lds) as spatial_mix.
// The two waves write disjoint halves of the workgroup's partial tiles; no cross-wave sum.
//
// Per source the walk leaves one 96-byte PairRec (make_pair_rec) instead of two 64-byte TileRecs.
// LDS per workgroup: two window buffers of PAIR_WIN_CAP samples + one 5 KB block of stream checkpoints per wave
// = 19 712 B -> 8 workgroups = 16 waves per CU, the occupancy of spatial_mix.
//
// Used for FAST / FAST_UNFUSED callbacks of 513..1024 frames over scenes large enough to give every workgroup of a full
// chip at least one group of 16 sources (scene_host.inc); everything else keeps spatial_mix.  Sources without a staged
// variant here (windows beyond the stage, Downmix, Sine, Constant, Cycle rows) run out of line on parked accumulators,
// exactly as in spatial_mix.
#pragma once
#include "kernels.h"

namespace oddio_hip {

constexpr int PAIR_GROUP = MIX_GROUP;                                     // sources per cursor-scan step (16 sources x 4 chunks = 64 lanes)
// PAIR_DEPTH windows in flight behind the one being mixed (round 6 experiment, default 1).  The question: is the kernel's fetch side
// bound by concurrency -- one window in flight per workgroup = 2 048 x 4.3 KB = 8.8 MB outstanding, and 8.8 MB per 1.5 us of loaded
// memory latency would be exactly the 6.0 TB/s the fetch-only diagnostic build reaches -- or by the rate HBM gives random 4.3-KB reads?
// Round 5's three-buffer variant could not tell: its third buffer cost two workgroups per CU.  ODDIO_PAIR_DEPTH=2 pays for the third
// buffer with the stream blocks instead: a cursor checkpoint per 32 frames (a lane whose 16 frames start mid-interval replays the 16
// steps before them with the scan's own adds), 12 words per stream instead of 20 -- 3 x 4 736 + 2 x 3 072 = 20 352 B per workgroup,
// still 8 workgroups per CU, 109 VGPRs, no scratch; the younger window's DMA stays in flight across the per-source hand-over
// (`s_waitcnt vmcnt(2)`, a barrier without hipcc's fences).  Parity-green (every pair-kernel, TRACKED and bounds-build test).
// Measured, same box, alternating runs (profiles/r06_ab_pair_depth2.txt): the kernel 0.2325 against 0.2312 ms, its fetch alone
// (no sample loop) 0.2098 against 0.2023 -- more windows in flight do NOT raise the fetch rate: it is the rate of this access
// pattern, not a latency x concurrency product.  Kept as a build option; the default stays the two-buffer layout.
#ifndef ODDIO_PAIR_DEPTH
#define ODDIO_PAIR_DEPTH 1
#endif
constexpr int PAIR_DEPTH = ODDIO_PAIR_DEPTH;
static_assert(PAIR_DEPTH == 1 || PAIR_DEPTH == 2, "one or two windows in flight");
constexpr int PAIR_NBUF = PAIR_DEPTH + 1;
constexpr int PAIR_CK_SHIFT = PAIR_DEPTH == 2 ? 1 : 0;                    // checkpoints every 16 << shift frames
constexpr int PAIR_CK = 16 >> PAIR_CK_SHIFT;                              // checkpoints per stream (a 256-frame chunk)
constexpr int PAIR_STREAM_WORDS = PAIR_CK + 4;                            // + {4 * wrel, g0, dg, ds}
constexpr int PAIR_WIN_BYTES = PAIR_WIN_CAP * 4;
constexpr int PAIR_LDS_WIN0 = 0;
constexpr int PAIR_LDS_STREAM = PAIR_NBUF * PAIR_WIN_BYTES;
constexpr int PAIR_STREAM_BYTES = 64 * PAIR_STREAM_WORDS * 4;             // one wave's 64 stream blocks
constexpr int PAIR_LDS_TOTAL = PAIR_LDS_STREAM + 2 * PAIR_STREAM_BYTES;
constexpr int PAIR_TAIL_LANES = (PAIR_WIN_BYTES - 4096) / 16;             // lanes of the fifth 1 KiB piece that stay inside the window buffer
static_assert(PAIR_WIN_BYTES % 16 == 0 && PAIR_LDS_TOTAL % 16 == 0 && (PAIR_STREAM_WORDS * 4) % 16 == 0, "16-byte aligned window buffers and stream blocks");
static_assert(PAIR_LDS_TOTAL * 8 <= 160 * 1024, "8 workgroups (16 waves) per CU");
static_assert(16 * PARK_STRIDE * 4 <= PAIR_WIN_BYTES, "a wave parks its accumulators in one window buffer");
static_assert(PAIR_WIN_BYTES > 4096 && PAIR_WIN_BYTES <= 5120 && PAIR_TAIL_LANES > 0 && PAIR_TAIL_LANES <= 64, "five 1 KiB pieces cover a window buffer");

// s_barrier of the two waves with the LDS hand-over made explicit for the compiler (the builtin alone is not a memory barrier)
__device__ __forceinline__ void pair_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The per-source hand-over with two windows in flight (PAIR_DEPTH == 2): the same barrier without the fences, which hipcc lowers to
// `s_waitcnt vmcnt(0)` on both sides -- that would drain the younger window's DMA at every source.  What crosses the barrier here is
// LDS only: each wave has waited for its own half of the window (vmcnt, by hand) and for its own LDS accesses (lgkmcnt) before it
// arrives; the "memory" clobber keeps the compiler from moving accesses across.
__device__ __forceinline__ void pair_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// HBM -> LDS, this wave's half of a source's window: the 1 KiB pieces first, first + 2 (and 4 when first == 0) through the
// bounds-checked descriptor of the PairRec (window_desc: lanes outside the clip get zeros, frames.rs:105-123).
// The instruction offset moves the global and the LDS address alike (see window_dma).
__device__ __forceinline__ void pair_window_dma(uint32_t lds_dst, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t info, int lane16, int first) {
    u32x4 rsrc;
    rsrc.x = d0; rsrc.y = d1; rsrc.z = d2;
    rsrc.w = 0x00020000u;
    const int nvec = (int)((info >> 8) & 511u);
    const int neg = -16 * (int)((info >> 17) & 511u);
#ifdef ODDIO_HIP_BOUNDS
    if (nvec * 16 > PAIR_WIN_BYTES || (int)d2 > nvec * 16 + neg + 16) asm volatile("s_trap 2");
#endif
    const int voff = neg + lane16 + 1024 * first;
    if (ODDIO_DIAG & 8) return;
    const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_dst + 1024u * (uint32_t)first));   // (wave-uniform; M0 wants an SGPR)
    uint32_t keep;
    // (pieces 0-3 from every lane: lanes past the window write zeros inside the buffer, no traffic)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
                 "buffer_load_dwordx4 %1, %2, 0 offen" ODDIO_WIN_POLICY " lds\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:2048" ODDIO_WIN_POLICY " lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(m0v) : "memory");
    if (first == 0 && nvec > 256) {
        if (lane16 < 16 * PAIR_TAIL_LANES)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen" ODDIO_WIN_POLICY " lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff + 4096), "s"(rsrc), "s"(m0v + 4096u) : "memory");
    }
}

// Near-unit sources: plain -> padded layout in place (window_repack_padded), the two waves of the workgroup together:
// wave w moves the 16-byte vectors of the 1 KiB pieces w, w + 2, w + 4.  Every read of either wave precedes every write.
__device__ __forceinline__ void pair_repack_padded(unsigned char* win_bytes, int nvec, int lane, int wv, uint32_t* err) {
    asm volatile("" : "+v"(lane));
    u32x4 v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int q = lane + 64 * (2 * k + wv);
        v[k] = (q < nvec) ? *reinterpret_cast<const u32x4*>(win_bytes + 16 * q) : u32x4{0u, 0u, 0u, 0u};
    }
    pair_barrier();
    unsigned int* win = reinterpret_cast<unsigned int*>(win_bytes);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int q = lane + 64 * (2 * k + wv);
        if (q < nvec) {
            const int li = 4 * q;
            const int pos = li + (li >> 4);
            if (!ODDIO_BOUNDS_CHECK(err, pos + 3 < PAIR_WIN_CAP, BOUNDS_REPACK, pos, nvec)) continue;
            win[pos + 0] = v[k].x; win[pos + 1] = v[k].y; win[pos + 2] = v[k].z; win[pos + 3] = v[k].w;
            if ((li & 15) == 0 && li > 0) win[pos - 1] = v[k].x;
        }
    }
    pair_barrier();
}

// grid = workgroups; block = 128 (wave 0: left ear, wave 1: right ear).  Workgroup w walks groups [g_lo, g_hi) of 16 slots
// in DESCENDING order (the reference's reverse set walk, spatial.rs:204) and leaves its partial sums in spatial_mix's
// layout (kernels.h PART_BLOCK) for reduce_partials.
// TRACK (ODDIO_HIP_MODE_TRACKED, second pass): `init` holds, in the partial tiles' layout, the value the reference's running sum has --
// to a few hundred ulps -- when its walk reaches this workgroup's sources (track_prefix over the first pass's partial sums).  The
// wave restarts its running sums there, adds its sources in the reference's order with the reference's roundings (FUSED = false),
// and leaves what they added: end - start.  A sequential f32 sum rounds every addend to the ulp of the running sum's binade -- the
// rounding error of a step does not depend on the sum's lower bits -- so the restarted sums make the reference's rounding errors, and
// the workgroups' differences add up to the reference's sequential sum to ~1e-6 of the peak (the plain tree sum: 1-2e-5 at 262 144
// sources, which is the reference's own distance from the exact sum).
// WALK (round 6): the set walk of the callback -- spatial_prepass's work for this workgroup's own sources (walk_set, both EarStates, the
// clock bookkeeping, the PairRecs: prepass_source + make_pair_rec, one source per lane, the two waves on alternate blocks of 64) -- runs
// at the top of the kernel instead of in a launch of its own: 19 us of launch + a 96-byte record round trip per source through HBM
// become ~8 us inside the kernel (the records are read back through L2 by the workgroup that wrote them).  `ear_in` / `recs_in` /
// `n_sources_ptr` are then unused: the walk's own tables (PairWalk) are read back.  For scenes without Cycle rows (their scan sits
// between the walk and the mix) and callbacks the pair kernel renders; the TRACK second pass reads what the first pass's walk left.
struct PairWalk {
    SrcDyn* dyn; SrcPending* pend; EarParams* ear; PairRec* recs; uint32_t* stopped_hdr; const uint32_t* d_len; uint32_t* len_snap;
    uint32_t stopped_cap; int check_pending;
};
constexpr int PAIR_WALK_STAGE = 64 * (sizeof(PairRec) / 4 + 1) * 4;      // bytes of a wave's transposition stage (wave_aos_*)
static_assert(2 * PAIR_WALK_STAGE <= PAIR_LDS_TOTAL, "the walk's two transposition stages fit the workgroup's LDS");

// LANE16 (with FULL): a callback of 16 k < 1024 frames -- every lane's 16 frames lie wholly inside or wholly outside `out`, so the lanes
// inside run the full-callback loop (no per-sample select, sums accumulated in place) and the others sit the sources out; what the
// reference's sizes of 10 / 20 ms (480 / 960 frames) and any other multiple of 16 take instead of the ragged instantiations.
template <bool FULL, bool FUSED, bool TRACK = false, bool WALK = false, bool LANE16 = false>
__global__ __launch_bounds__(128, MIX_WAVES_PER_SIMD) void spatial_mix_pair(SceneParams P, const SrcStatic* __restrict__ st,
                                                                            const EarParams* __restrict__ ear_in, const PairRec* __restrict__ recs_in,
                                                                            float* __restrict__ partials, const float* __restrict__ init,
                                                                            uint32_t groups_per_wg, uint32_t n_groups,
                                                                            const uint32_t* __restrict__ n_sources_ptr, PairWalk W) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[PAIR_LDS_TOTAL];
    static_assert(!WALK || !TRACK, "the second pass of a tracked callback reads the records of the first pass's walk");
    static_assert(!LANE16 || (FULL && !WALK), "lane-granular callbacks run the full-callback loop on the lanes inside the callback");
    const uint32_t n_sources = WALK ? W.d_len[0] : *n_sources_ptr;
    // (WALK: the tables this workgroup has just written are read through the pointers they were written through)
    const EarParams* ear = WALK ? W.ear : ear_in;
    const PairRec* recs = WALK ? W.recs : recs_in;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));       // this wave's ear
    const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)smem);
    const int lane = threadIdx.x & 63;
    const int lane16 = 16 * lane;
    const uint32_t n_frames = P.n_frames;
    // n_groups = gridDim.x * groups_per_wg + rem: the first `rem` workgroups walk one group more
    const uint32_t rem = n_groups - gridDim.x * groups_per_wg;
    const uint32_t g_lo = blockIdx.x * groups_per_wg + (blockIdx.x < rem ? blockIdx.x : rem);
    const uint32_t g_hi = g_lo + groups_per_wg + (blockIdx.x < rem ? 1u : 0u);

    if (WALK) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *W.len_snap = n_sources;      // (the reduce's set compaction and a TRACK second pass read it)
        uint32_t* stage = reinterpret_cast<uint32_t*>(smem + (wv ? PAIR_WALK_STAGE : 0));
        const uint32_t s_end = g_hi * MIX_GROUP < n_sources ? g_hi * MIX_GROUP : n_sources;
        int lw = threadIdx.x & 63;
        asm volatile("" : "+v"(lw));
#pragma unroll 1
        for (uint32_t first = g_lo * MIX_GROUP + 64u * (uint32_t)wv; first < s_end; first += 128u) {
            const uint32_t n_valid = (s_end - first) < 64u ? (s_end - first) : 64u;
            const uint32_t i = first + (uint32_t)lw;
            SrcDyn d = {};
            SrcStatic s = {};
            wave_aos_load2(d, W.dyn, s, st, first, n_valid, lw, stage);
            EarPair ep = {};
            ep.e[0].flags = EAR_SKIP; ep.e[1].flags = EAR_SKIP;
            if (i < s_end) prepass_source(P, i, d, s, W.pend, ep.e[0], ep.e[1], W.stopped_hdr, W.stopped_cap, W.check_pending);
            const PairRec r = make_pair_rec(P, s, ep.e[0], ep.e[1]);
            const uint32_t path = r.info & 7u;
            const bool needs_ear = path != PATH_LDS && path != PATH_SKIP;      // (the out-of-line paths read the EarParams)
            wave_aos_store(r, W.recs, first, n_valid, lw, stage);
            if (__any(needs_ear)) wave_aos_store(ep, reinterpret_cast<EarPair*>(W.ear), first, n_valid, lw, stage);
            wave_aos_store(d, W.dyn, first, n_valid, lw, stage);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's records have left ...
        pair_barrier();                                       // ... and the other wave's: both read all of them below
    }
    float acc[16], fi[16];
    // phase-B role: chunk c, block b -> the 16 consecutive frames 256 c + 16 b ..
    const int cB = lane >> 4, bB = lane & 15;
    const uint32_t frame0 = 16u * (uint32_t)lane;
    const bool lane_on = !LANE16 || frame0 < P.n_frames;     // (LANE16: n_frames % 16 == 0)
    const float fbase = (float)frame0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { acc[k] = 0.0f; fi[k] = fbase + (float)k; }   // `i as f32` (spatial.rs:459)
    static_assert(!TRACK || !FUSED, "the tracked sums carry the reference's roundings");
    if (TRACK) {
        if (frame0 < n_frames) {
            const float* src = init + ((size_t)(2 * lane) * gridDim.x + blockIdx.x) * PART_BLOCK + (size_t)wv * PART_FRAMES;
            const size_t step = (size_t)gridDim.x * PART_BLOCK;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 v = reinterpret_cast<const float4*>(src + (q4 >> 1) * step)[q4 & 1];
                acc[4 * q4] = v.x; acc[4 * q4 + 1] = v.y; acc[4 * q4 + 2] = v.z; acc[4 * q4 + 3] = v.w;
            }
        }
    } else if (init != nullptr && blockIdx.x == 0) {
        // the buffered set is walked before the seekable one (spatial.rs:395-438): its sum is what the first source is added to
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (frame0 + (uint32_t)k < n_frames) acc[k] = init[2 * (frame0 + (uint32_t)k) + (uint32_t)wv];
    }
    // (the loads above are awaited here, once: left pending, hipcc puts its `s_waitcnt vmcnt(0)` for them in front of every loop of
    // the walk that touches the accumulators)
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(acc[k]));
    unsigned char* const sbase = smem + PAIR_LDS_STREAM + wv * PAIR_STREAM_BYTES;   // this wave's stream blocks: block 4 j + c
    unsigned char* const blkB0 = sbase + cB * (PAIR_STREAM_WORDS * 4);
    constexpr int BLK_SRC = 4 * PAIR_STREAM_WORDS * 4;

    // The records of a group -- lanes 0-15: {descriptor words, info} of source `lane`; lane (j, c): {ds, g0, dg} of this
    // wave's ear, the chunk's frac0 and wrel -- are fetched one group ahead (see spatial_mix).
#define PAIR_LOAD_GROUP(GG, V, Q, F, W)                                                                                   \
    {                                                                                                                     \
        int la_ = lane;                                                                                                   \
        asm volatile("" : "+v"(la_));                                                                                     \
        const PairRec* __restrict__ grec_ = recs + (size_t)(GG) * MIX_GROUP;                                              \
        V = make_uint4(0u, 0u, 0u, 0u); Q = f4u{0.0f, 0.0f, 0.0f, 0.0f}; F = 0.0f; W = 0u;                                \
        if (la_ < MIX_GROUP && (GG) * MIX_GROUP + (uint32_t)la_ < n_sources) V = *reinterpret_cast<const uint4*>(grec_ + la_); \
        if ((GG) * MIX_GROUP + (uint32_t)(la_ >> 2) < n_sources) {                                                        \
            const PairEar* pe_ = &grec_[la_ >> 2].ear[wv];                                                                \
            Q = *reinterpret_cast<const f4u*>(pe_);                              /* {ds, g0, dg, .} */                    \
            F = pe_->frac0[la_ & 3];                                                                                      \
            W = (uint32_t)pe_->wrel[la_ & 3];                                                                             \
        }                                                                                                                 \
    }
    uint4 pv = make_uint4(0u, 0u, 0u, 0u);
    f4u pq = {0.0f, 0.0f, 0.0f, 0.0f};
    float pf = 0.0f;
    uint32_t pw = 0u;
    if (g_hi > g_lo) PAIR_LOAD_GROUP(g_hi - 1u, pv, pq, pf, pw)
    int buf = 0;                 // the window buffer of the staged source being mixed (`cur`); its successors' follow cyclically
    int ahead = 0;               // PAIR_DEPTH == 2: the window of the staged source after `cur` is already in flight
    bool pre_issued = false;     // the last source of the previous group already started this group's first window
#define PAIR_WIN_OFF(B) ((uint32_t)(B) * (uint32_t)PAIR_WIN_BYTES)
#define PAIR_NEXT_BUF(B, K) (((B) + (K)) % PAIR_NBUF)
    // this wave's half of window `cur` has landed; with a younger window's DMA instructions behind it (`ahead`: always at least two,
    // pair_window_dma) those may stay in flight -- vmcnt counts in issue order, and anything issued between the two windows is older
#define PAIR_WINDOW_WAIT()                                                                                                \
    {                                                                                                                     \
        if (PAIR_DEPTH == 2 && ahead && !(ODDIO_DIAG & 8)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");               \
        else window_wait();                                                                                               \
    }
    for (uint32_t g = g_hi; g-- > g_lo;) {
        // ------------------------------ phase A ------------------------------
        const uint4 vdesc = pv;
        const f4u q = pq;
        const float frac0 = pf;
        const uint32_t wr0 = pw;
        bool need_prefetch = g > g_lo;
        int laneA = lane;
        asm volatile("" : "+v"(laneA));
        const int pj = (int)(vdesc.w & 7u);
        const unsigned lds_mask = (unsigned)__ballot(laneA < MIX_GROUP && pj == PATH_LDS) & 0xffffu;
        const unsigned rare_mask = (unsigned)__ballot(laneA < MIX_GROUP && pj != PATH_LDS && pj != PATH_SKIP) & 0xffffu;
        int cur = lds_mask ? 31 - __builtin_clz(lds_mask) : -1;
        uint32_t cur_info = 0;
#define PAIR_ISSUE_WINDOW_OF(VD, JN, BUF)                                                                                 \
    pair_window_dma(lds_base + PAIR_WIN_OFF(BUF), (uint32_t)__builtin_amdgcn_readlane((int)(VD).x, (JN)),                \
                    (uint32_t)__builtin_amdgcn_readlane((int)(VD).y, (JN)), (uint32_t)__builtin_amdgcn_readlane((int)(VD).z, (JN)), \
                    (uint32_t)__builtin_amdgcn_readlane((int)(VD).w, (JN)), lane16, wv ^ ((JN) & 1));
#define PAIR_ISSUE_WINDOW(JN, BUF) PAIR_ISSUE_WINDOW_OF(vdesc, JN, BUF)
        if (cur >= 0) {   // the first window is on its way while the cursors are scanned
            cur_info = (uint32_t)__builtin_amdgcn_readlane((int)vdesc.w, cur);
            if (!pre_issued) PAIR_ISSUE_WINDOW(cur, buf)
        }
        pre_issued = false;
        ahead = 0;
        {
            // exact f32 cursor scan (frames.rs:189-196) of stream (source j = lane >> 2, chunk c = lane & 3) of this wave's ear
            float* blk = reinterpret_cast<float*>(sbase + laneA * (PAIR_STREAM_WORDS * 4));
            const float ds = q.x;
            float x = frac0;
#pragma unroll 1
            for (int b = 0; b < 15; ++b) {
                if ((b & ((1 << PAIR_CK_SHIFT) - 1)) == 0) blk[b >> PAIR_CK_SHIFT] = x;
                if (ODDIO_DIAG & 4) { x = __builtin_fmaf(16.0f, ds, x); continue; }
#pragma unroll
                for (int i = 0; i < 16; ++i) x = x + ds;
            }
            if (PAIR_CK_SHIFT == 0) blk[15] = x;
            *reinterpret_cast<float4*>(blk + PAIR_CK) = make_float4(__uint_as_float(4u * wr0), q.y, q.z, q.x);
        }
        wave_sync();

        // ------------------------------ phase B ------------------------------
        float cx0 = 0.0f;
        float4 ct = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        // lane data of staged source J: the cursor at this lane's first frame -- its checkpoint, or (32-frame checkpoints, odd block) the
        // checkpoint before it advanced by the scan's own 16 adds -- and {4 * wrel, g0, dg, ds}
#define PAIR_LANE_DATA(J, X0, T)                                                                                          \
    {                                                                                                                     \
        const unsigned char* blk_ = blkB0 + (J) * BLK_SRC;                                                                \
        X0 = reinterpret_cast<const float*>(blk_)[bB >> PAIR_CK_SHIFT];                                                   \
        T = *reinterpret_cast<const float4*>(blk_ + 4 * PAIR_CK);                                                         \
        if (PAIR_CK_SHIFT && (bB & 1) && !(ODDIO_DIAG & 4)) {                                                             \
            _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) X0 = X0 + T.w;          /* frames.rs:194 */                 \
        }                                                                                                                 \
    }
        if (cur >= 0) PAIR_LANE_DATA(cur, cx0, ct)
        // VAR: 0 the common source, 1 padded layout (resample ratio within PAD_EPS of 1), 2 FixedGain and/or a cursor that starts negative
#define PAIR_VARIANT(INFO) ((((INFO) >> 3) & SFLAG_PAD) ? 1 : ((((INFO) >> 3) & (SFLAG_FG | SFLAG_NEG)) ? 2 : 0))
#define PAIR_STAGED_SOURCE(VAR, PRE)                                                                                      \
    {                                                                                                                     \
        const int flags_j = (int)((cur_info >> 3) & 31u);                                                                 \
        unsigned char* win_bytes = smem + PAIR_WIN_OFF(buf);                                                              \
        const int nvec_j = (int)((cur_info >> 8) & 511u);                                                                 \
        PAIR_WINDOW_WAIT()                                /* this wave's half of the window has landed */                 \
        /* ... and the other wave's; both are done with the buffer of the source before */                               \
        if (PAIR_DEPTH == 2) pair_barrier_lds();                                                                          \
        else {                                                                                                            \
            pair_barrier();                                                                                               \
            asm volatile("" : "+v"(pv.x), "+v"(pv.y), "+v"(pv.z), "+v"(pv.w), "+v"(pq.x), "+v"(pq.y), "+v"(pq.z), "+v"(pq.w), "+v"(pf), "+v"(pw)); \
        }                                                                                                                 \
        const bool fetched_before = !need_prefetch;                                                                       \
        if (need_prefetch) { PAIR_LOAD_GROUP(g - 1u, pv, pq, pf, pw) need_prefetch = false; }                             \
        const unsigned below = lds_mask & ((1u << cur) - 1u);                                                             \
        const int nxt = below ? 31 - __builtin_clz(below) : -1;                                                           \
        uint32_t nxt_info = 0;                                                                                            \
        float nx0 = 0.0f;                                                                                                 \
        float4 nt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                                                                  \
        int ahead_next = 0;                                                                                               \
        if (nxt >= 0) {          /* the staged sources behind this one: their windows land while we compute */            \
            nxt_info = (uint32_t)__builtin_amdgcn_readlane((int)vdesc.w, nxt);                                            \
            if (PAIR_DEPTH == 1 || !ahead) PAIR_ISSUE_WINDOW(nxt, PAIR_NEXT_BUF(buf, 1))                                  \
            if (PAIR_DEPTH == 2) {                                                                                        \
                const unsigned below2 = below & ((1u << nxt) - 1u);                                                       \
                if (below2) {    /* (its buffer is the one the source before this one was mixed from) */                  \
                    PAIR_ISSUE_WINDOW(31 - __builtin_clz(below2), PAIR_NEXT_BUF(buf, 2))                                  \
                    ahead_next = 1;                                                                                       \
                }                                                                                                         \
            }                                                                                                             \
            PAIR_LANE_DATA(nxt, nx0, nt)                                                                                  \
        } else if ((PRE) && g > g_lo && fetched_before) {                                                                 \
            /* last staged source of the group: the next window buffer is free for the next group's first window */      \
            int lb_ = lane;                                                                                               \
            asm volatile("" : "+v"(lb_));                                                                                 \
            const unsigned nm_ = (unsigned)__ballot(lb_ < MIX_GROUP && (int)(pv.w & 7u) == PATH_LDS) & 0xffffu;           \
            if (nm_) { PAIR_ISSUE_WINDOW_OF(pv, 31 - __builtin_clz(nm_), PAIR_NEXT_BUF(buf, 1)) pre_issued = true; }      \
        }                                                                                                                 \
        const int wrel4 = __float_as_int(ct.x);                                                                           \
        if (ODDIO_DIAG & 2) { acc[0] += cx0 + ct.y + ct.z + ct.w + __int_as_float(wrel4); }                              \
        else if ((VAR) == 1) {                                                                                            \
            const float fg = (flags_j & SFLAG_FG) ? st[g * MIX_GROUP + (uint32_t)cur].fixed_gain : 1.0f;   /* v * 1.0 == v */ \
            const float frac0_ = reinterpret_cast<const float*>(blkB0 + cur * BLK_SRC)[0];   /* checkpoint 0 */           \
            const int fast_e = wv ? (flags_j & SFLAG_FAST_R) : (flags_j & SFLAG_FAST_L);                                  \
            pair_repack_padded(win_bytes, nvec_j, lane, wv, P.bounds_err);                                                \
            if (lane_on) mix_source_lds<FULL, true, false, true, FUSED, false, false, PAIR_WIN_CAP, true, FUSED>(win_bytes, wrel4, cx0, bB, fast_e, frac0_, acc, fi, frame0, n_frames, fg, ct.y, ct.z, ct.w, 4 * nvec_j, P.bounds_err, \
                                                                                             0, 0, (int)((cur_info >> 28) & 7u)); \
        } else if ((VAR) == 0) {                                                                                          \
            if (lane_on) mix_source_lds<FULL, false, true, false, FUSED, false, false, PAIR_WIN_CAP, false, FUSED>(win_bytes, wrel4, cx0, bB, 0, 0.0f, acc, fi, frame0, n_frames, 1.0f, ct.y, ct.z, ct.w, 4 * nvec_j, P.bounds_err); \
        } else {                                                                                                          \
            const float fg = (flags_j & SFLAG_FG) ? st[g * MIX_GROUP + (uint32_t)cur].fixed_gain : 1.0f;                  \
            if (lane_on) mix_source_lds<FULL, true, false, false, FUSED, false, false, PAIR_WIN_CAP, true, FUSED>(win_bytes, wrel4, cx0, bB, 0, 0.0f, acc, fi, frame0, n_frames, fg, ct.y, ct.z, ct.w, 4 * nvec_j, P.bounds_err, \
                                                                                              0, 0, (int)((cur_info >> 28) & 7u)); \
        }                                                                                                                 \
        buf = PAIR_NEXT_BUF(buf, 1);                                                                                      \
        ahead = ahead_next;                                                                                               \
        cur = nxt; cur_info = nxt_info; cx0 = nx0; ct = nt;                                                               \
    }
        // rare path: both waves park their accumulators (wave w over window buffer w: every window in flight is awaited first, and the
        // current one is fetched again afterwards; a second one in flight is simply issued again by the next staged source), run out of
        // line, fetch them back
#define PAIR_RARE_SOURCE(J)                                                                                               \
    {                                                                                                                     \
        const int path_j = __builtin_amdgcn_readlane((int)vdesc.w, (J)) & 7;                                              \
        float* park = reinterpret_cast<float*>(smem + wv * PAIR_WIN_BYTES);                                               \
        float ph_ = 0.0f;                                                                                                 \
        float4 t_ = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                                                                  \
        if (path_j == PATH_SINE_INLINE) { PAIR_LANE_DATA((J), ph_, t_) ph_ = reinterpret_cast<const float*>(blkB0 + (J) * BLK_SRC)[0]; } \
        window_wait();                                                                                                    \
        pair_barrier();                                   /* no window DMA of either wave is in flight any more */       \
        _Pragma("unroll") for (int k = 0; k < 16; ++k) park[k * PARK_STRIDE + lane] = acc[k];                             \
        wave_sync();                                                                                                      \
        if (path_j == PATH_SINE_INLINE)                                                                                   \
            mix_source_sine(park, lane, frame0, n_frames, ph_, t_.w, __int_as_float(__builtin_amdgcn_readlane((int)vdesc.x, (J))), \
                            __int_as_float(__builtin_amdgcn_readlane((int)vdesc.y, (J))), t_.y, t_.z);                    \
        else mix_source_rare_ear(park, lane, frame0, n_frames, (uint32_t)cB, path_j, st, ear, g * MIX_GROUP + (uint32_t)(J), P.cycle_rows, P.cycle_plane, wv); \
        wave_sync();                                                                                                      \
        _Pragma("unroll") for (int k = 0; k < 16; ++k) acc[k] = park[k * PARK_STRIDE + lane];                             \
        pair_barrier();                                   /* both park areas are free again */                            \
        ahead = 0;                                                                                                        \
        if (cur >= 0) PAIR_ISSUE_WINDOW(cur, buf)                                                                         \
    }
        {
            unsigned rm = rare_mask;
            while (cur >= 0 || rm) {
                const int rj = rm ? 31 - __builtin_clz(rm) : -1;
                if (rj > cur) {
                    rm &= ~(1u << rj);
                    PAIR_RARE_SOURCE(rj)
                    continue;
                }
#pragma unroll 1
                while (cur > rj && PAIR_VARIANT(cur_info) == 0) PAIR_STAGED_SOURCE(0, rm == 0)
#pragma unroll 1
                while (cur > rj && PAIR_VARIANT(cur_info) == 1) PAIR_STAGED_SOURCE(1, rm == 0)
#pragma unroll 1
                while (cur > rj && PAIR_VARIANT(cur_info) == 2) PAIR_STAGED_SOURCE(2, rm == 0)
            }
        }
#undef PAIR_RARE_SOURCE
#undef PAIR_STAGED_SOURCE
#undef PAIR_VARIANT
#undef PAIR_LANE_DATA
#undef PAIR_ISSUE_WINDOW
#undef PAIR_ISSUE_WINDOW_OF
        if (need_prefetch) PAIR_LOAD_GROUP(g - 1u, pv, pq, pf, pw)   // a group without a staged source
        wave_sync();   // before the next group's phase A overwrites the stream blocks
    }
#undef PAIR_WINDOW_WAIT
#undef PAIR_NEXT_BUF
#undef PAIR_WIN_OFF
#undef PAIR_LOAD_GROUP

    // ---- this wave's half of the workgroup's partial tiles: ear wv, frames 16 lane .. (tile = lane >> 5) ----
    int le = threadIdx.x & 63;
    asm volatile("" : "+v"(le));
    if (16u * (uint32_t)le < n_frames) {
        // the lane's 16 frames are two blocks of PART_FRAMES frames (kernels.h: the partial sums' layout)
        float* dst = partials + ((size_t)(2 * le) * gridDim.x + blockIdx.x) * PART_BLOCK + (size_t)wv * PART_FRAMES;
        const size_t dst_step = (size_t)gridDim.x * PART_BLOCK;
        if (TRACK && (P.track_all || blockIdx.x + 1u != gridDim.x)) {   // (the workgroup the reference's walk starts with keeps its start value: the buffered set's sum)
            const float* src = init + ((size_t)(2 * le) * gridDim.x + blockIdx.x) * PART_BLOCK + (size_t)wv * PART_FRAMES;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 v = reinterpret_cast<const float4*>(src + (q4 >> 1) * dst_step)[q4 & 1];
                acc[4 * q4] = acc[4 * q4] - v.x; acc[4 * q4 + 1] = acc[4 * q4 + 1] - v.y; acc[4 * q4 + 2] = acc[4 * q4 + 2] - v.z; acc[4 * q4 + 3] = acc[4 * q4 + 3] - v.w;
            }
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) reinterpret_cast<float4*>(dst + (q4 >> 1) * dst_step)[q4 & 1] = make_float4(acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]);
    }
}

// ODDIO_HIP_MODE_TRACKED, between the two passes: prefix[w] = the buffered set's sum (`init`, or 0) + the partial sums of the workgroups
// the reference's walk passes before workgroup w (w + 1 .. n_wgs - 1: it starts at the highest slot), in that order, for every output
// of the callback; partial-tile layout in and out.  grid = blocks of PART_FRAMES frames; block = 16 outputs x TRK_SEGS segments of
// the workgroup list: segment sums, their exclusive scan, then the running values.
constexpr int TRK_SEGS = 32;
constexpr int TRK_BATCH = 16;
__global__ __launch_bounds__(PART_BLOCK * TRK_SEGS) void track_prefix(const float* __restrict__ partials, const float* __restrict__ init, float* __restrict__ prefix,
                                                                       uint32_t n_wgs, uint32_t n_frames) {
    __shared__ float tot[TRK_SEGS][PART_BLOCK];
    const uint32_t ox = threadIdx.x & (PART_BLOCK - 1), seg = threadIdx.x / PART_BLOCK;
    const int per = (int)((n_wgs + TRK_SEGS - 1) / TRK_SEGS);
    const int hi = (int)n_wgs - 1 - (int)seg * per;                 // this segment: workgroups hi, hi - 1, .. lo
    const int lo = hi - per + 1 > 0 ? hi - per + 1 : 0;
    const float* p = partials + (size_t)blockIdx.x * n_wgs * PART_BLOCK + ox;
    float* q = prefix + (size_t)blockIdx.x * n_wgs * PART_BLOCK + ox;
    float sum = 0.0f;
    for (int w = hi; w >= lo; w -= TRK_BATCH) {
        float v[TRK_BATCH];
#pragma unroll
        for (int k = 0; k < TRK_BATCH; ++k) v[k] = (w - k >= lo) ? p[(size_t)(w - k) * PART_BLOCK] : 0.0f;
#pragma unroll
        for (int k = 0; k < TRK_BATCH; ++k) sum = sum + v[k];
    }
    tot[seg][ox] = sum;
    __syncthreads();
    const uint32_t f = blockIdx.x * PART_FRAMES + (ox % PART_FRAMES);
    float run = (init != nullptr && f < n_frames) ? init[2 * f + ox / PART_FRAMES] : 0.0f;
    for (uint32_t k = 0; k < seg; ++k) run = run + tot[k][ox];
    for (int w = hi; w >= lo; w -= TRK_BATCH) {
        float v[TRK_BATCH];
#pragma unroll
        for (int k = 0; k < TRK_BATCH; ++k) v[k] = (w - k >= lo) ? p[(size_t)(w - k) * PART_BLOCK] : 0.0f;
#pragma unroll
        for (int k = 0; k < TRK_BATCH; ++k)
            if (w - k >= lo) { q[(size_t)(w - k) * PART_BLOCK] = run; run = run + v[k]; }
    }
}

// Sharded scenes in TRACKED mode (scene_host.inc): the reference's walk passes the ranks in descending order (contiguous index shards:
// rank world - 1 holds the highest slots), so rank r's running sums start at base_r = total_{world-1} + ... + total_{r+1}, the
// totals being each rank's own sum of its first pass.  RCCL path: `gather` = ncclAllGather of the totals.
__global__ void track_base(const float* __restrict__ gather, uint32_t rank, uint32_t world, uint32_t n_out, float* __restrict__ base) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    float b = 0.0f;
    bool first = true;
    for (uint32_t r = world; r-- > rank + 1u;) { const float v = gather[(size_t)r * n_out + i]; b = first ? v : b + v; first = false; }
    base[i] = b;
}
__global__ void add_stereo(float* __restrict__ out, const float* __restrict__ add, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = out[i] + add[i];
}

}  // namespace oddio_hip
