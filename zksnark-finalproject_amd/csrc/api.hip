// libzkg16 C ABI (include/zkg16.h): context, proving-key / R1CS / witness residency, the prove pipeline and
// its O(1) host tail.  Mirrors ark-groth16 0.4 `create_proof_with_reduction_and_matrices` +
// `create_proof_with_assignment` (src/prover.rs) as reached from
// /root/reference/src/arkworks/backend/matrix_proof.rs:139-140.
//
// There is NO CPU fallback: without a HIP device zkg16_init fails with ZKG16_ERR_NO_DEVICE.
#include <chrono>
#include <functional>
#include <thread>

#include <exception>

#include "common.hpp"

using namespace zk;

namespace zk {
ScopedKernelTimer::ScopedKernelTimer(zkg16_ctx *c, const char *n, double u, hipStream_t st)
    : ctx(c), name(n), units(u), stream(st ? st : c->stream) {
    if (!ctx->kernel_timing) return;
    if (ctx->kernel_timing_accumulate_only && strncmp(n, "msm_accumulate", 14) != 0) return;
    ZK_HIP(hipEventCreate(&e0));
    ZK_HIP(hipEventCreate(&e1));
    ZK_HIP(hipEventRecord(e0, stream));
}
ScopedKernelTimer::~ScopedKernelTimer() {
    if (!e0) return;
    (void)hipEventRecord(e1, stream);
    ctx->pending_events.push_back(PendingEvent{name, units, e0, e1});
}
void kernel_timer_resolve(zkg16_ctx *ctx) {
    if (ctx->pending_events.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->wm_stream);
    for (auto &sl : ctx->slots)
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    for (auto &p : ctx->pending_events) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            auto &s = ctx->kstats[p.name];
            s.launches++;
            s.ms += ms;
            s.units += p.units;
        }
        (void)hipEventDestroy(p.e0);
        (void)hipEventDestroy(p.e1);
    }
    ctx->pending_events.clear();
}
}  // namespace zk

namespace {

const char *k_version = "zkg16 0.1 (gfx950; BLS12-381 Groth16 prove hot path)";

int fail(zkg16_ctx *ctx, const HipError &e) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", e.what, hipGetErrorString(e.err), e.file, e.line);
    if (ctx) ctx->last_error = buf;
    (void)hipGetLastError();
    if (e.err == hipErrorOutOfMemory) return ZKG16_ERR_OOM;
    if (e.err == hipErrorInvalidValue && strstr(e.what, "domain")) return ZKG16_ERR_DOMAIN_TOO_LARGE;
    return ZKG16_ERR_HIP;
}

#define ZK_API_BEGIN(ctx)                         \
    if (!(ctx)) return ZKG16_ERR_BAD_ARG;         \
    std::lock_guard<std::mutex> _lk((ctx)->mu);   \
    try {                                         \
        ZK_HIP(hipSetDevice((ctx)->device));
#define ZK_API_END(ctx)                           \
    }                                             \
    catch (const HipError &e) { return fail((ctx), e); } \
    catch (const std::bad_alloc &) { return ZKG16_ERR_OOM; } \
    return ZKG16_OK;

// ---- lanes (common.hpp: zkg16_ctx::lanes).  Proving entry points take a free lane for the duration of the call; everything
// else (key / matrix / assignment residency, setup, the stage entry points) runs on the root under its mutex, as before.
void copy_options(zkg16_ctx *dst, const zkg16_ctx *src) {
    dst->opt_window_bits = src->opt_window_bits; dst->opt_min_seg = src->opt_min_seg; dst->opt_ntt_mode = src->opt_ntt_mode;
    dst->opt_reduce_mode = src->opt_reduce_mode; dst->opt_b_filter = src->opt_b_filter; dst->opt_spmv_dict = src->opt_spmv_dict;
    dst->opt_wm_first = src->opt_wm_first; dst->opt_g1_waves = src->opt_g1_waves; dst->opt_fixup_aux = src->opt_fixup_aux;
    dst->opt_window_bits_h = src->opt_window_bits_h; dst->opt_reduce_chunk = src->opt_reduce_chunk; dst->opt_wm_concurrent = src->opt_wm_concurrent;
    dst->opt_ntt_radix = src->opt_ntt_radix; dst->opt_ntt_xcd = src->opt_ntt_xcd; dst->opt_acc_debug = src->opt_acc_debug;
    dst->opt_sort_mode = src->opt_sort_mode; dst->opt_acc_pipeline = src->opt_acc_pipeline; dst->opt_fuse_pointwise = src->opt_fuse_pointwise;
    dst->opt_matrix_parts = src->opt_matrix_parts; dst->opt_g2_lazy = src->opt_g2_lazy; dst->opt_fixed_base_bits = src->opt_fixed_base_bits;
    dst->opt_collect_threads = src->opt_collect_threads;
    dst->kernel_timing = src->kernel_timing; dst->kernel_timing_accumulate_only = src->kernel_timing_accumulate_only;
}
void create_streams(zkg16_ctx *ctx) {
    // Plain (equal-priority) streams.  Measured at n = 32: main low / witness-map high priority 16.6 ms per proof,
    // reversed 16.2 ms, no priorities 15.0 ms (profiles/kernel_timeline_r1_*.txt).
    ZK_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ZK_HIP(hipStreamCreateWithFlags(&ctx->wm_stream, hipStreamNonBlocking));
    // HIP hands streams to its 4 hardware queues round-robin in creation order, and which streams end up sharing a queue
    // moves a proof by ~4 % (13.9 vs 14.5 ms at n = 32).  All of a ctx's streams are therefore created here, in the order
    // the measured-best pairing needs (main | witness map | B2, L, A, B1, H reductions).  (A further ctx of the same
    // process starts three queues on; two such contexts together measured 76 proofs/s, two aligned ones 73.)
    for (int i : {0, 2, 3, 4, 1}) ZK_HIP(hipStreamCreateWithFlags(&ctx->slots[i].stream, hipStreamNonBlocking));
}
// everything a ctx (root or lane) owns on the device
void teardown(zkg16_ctx *ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->wm_stream);
    if (ctx->extra_host) (void)hipHostFree(ctx->extra_host);
    if (ctx->circuit_stage) (void)hipHostFree(ctx->circuit_stage);
    ctx->circuit_stage = nullptr;
    ctx->circuit_stage_bytes = 0;
    for (int i = 0; i < 2; i++) {
        if (ctx->stage_host[i]) (void)hipHostFree(ctx->stage_host[i]);
        if (ctx->stage_done[i]) (void)hipEventDestroy(ctx->stage_done[i]);
    }
    for (auto &sl : ctx->slots) {
        if (sl.wsums_host) (void)hipHostFree(sl.wsums_host);
        if (sl.acc_done) (void)hipEventDestroy(sl.acc_done);
        if (sl.red_done) (void)hipEventDestroy(sl.red_done);
        if (sl.acc_start) (void)hipEventDestroy(sl.acc_start);
        if (sl.red_start) (void)hipEventDestroy(sl.red_start);
        sl.buckets.release();
        sl.bucket_sum.release();
        sl.wsums_dev.release();
        sl.seg_head.release();
        sl.seg_tail.release();
        sl.seg_meta.release();
        sl.long_list.release();
        sl.long_sums.release();
        sl.red_a.release(); sl.red_b.release(); sl.red_c.release();
        if (sl.stream) { (void)hipStreamSynchronize(sl.stream); (void)hipStreamDestroy(sl.stream); }
    }
    if (ctx->wm_stream) (void)hipStreamDestroy(ctx->wm_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
}
// A free lane of `root` for one proof: the lowest free one (a single caller always gets lane 0 = the root itself, so nothing
// changes for it); callers beyond opt_lanes wait.  The lane's mutex is held for the lease.
struct LaneLease {
    zkg16_ctx *root, *lane = nullptr;
    int idx = -1;
    double t0 = 0;
    std::unique_lock<std::mutex> held;
    std::shared_lock<std::shared_mutex> keys;
    explicit LaneLease(zkg16_ctx *r) : root(r) {
        {
            std::unique_lock<std::mutex> lk(root->lane_mu);
            const int cap = root->opt_lanes < 1 ? 1 : root->opt_lanes > 8 ? 8 : root->opt_lanes;
            root->lane_cv.wait(lk, [&] {
                for (int i = 0; i < cap; i++)
                    if (!root->lane_busy[i]) { idx = i; return true; }
                return false;
            });
            if (idx > 0 && !root->lanes[idx - 1]) {
                ZK_HIP(hipSetDevice(root->device));
                auto l = std::make_unique<zkg16_ctx>();
                l->device = root->device;
                l->num_cus = root->num_cus;
                l->root = root;
                copy_options(l.get(), root);
                try {
                    create_streams(l.get());
                } catch (...) {
                    teardown(l.get());
                    throw;
                }
                root->lanes[idx - 1] = std::move(l);
            }
            root->lane_busy[idx] = true;
            lane = idx == 0 ? root : root->lanes[idx - 1].get();
        }
        held = std::unique_lock<std::mutex>(lane->mu);
        keys = std::shared_lock<std::shared_mutex>(root->key_rw);
        t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    ~LaneLease() {
        if (idx < 0) return;
        const double t1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
        if (keys.owns_lock()) keys.unlock();
        if (held.owns_lock()) held.unlock();
        {
            std::lock_guard<std::mutex> lk(root->lane_mu);
            root->lane_busy[idx] = false;
            root->last_lane = idx;
            if (root->lane_log.size() >= 256) root->lane_log.erase(root->lane_log.begin(), root->lane_log.begin() + 128);
            root->lane_log.push_back(zkg16_ctx::LaneLogEntry{idx, t0, t1});
        }
        root->lane_cv.notify_one();
    }
    LaneLease(const LaneLease &) = delete;
    LaneLease &operator=(const LaneLease &) = delete;
};
zkg16_ctx *lane_of(zkg16_ctx *root, int idx) { return idx <= 0 || idx > 7 || !root->lanes[idx - 1] ? root : root->lanes[idx - 1].get(); }
// the proving entry points: `ctx` is rebound to the leased lane for the body, `root` keeps the handle maps
#define ZK_LANE_BEGIN(ctx)                        \
    if (!(ctx)) return ZKG16_ERR_BAD_ARG;         \
    zkg16_ctx *const root = (ctx);                \
    try {                                         \
        LaneLease _lease(root);                   \
        (ctx) = _lease.lane;                      \
        try {                                     \
            ZK_HIP(hipSetDevice((ctx)->device));
#define ZK_LANE_END(ctx)                          \
        } catch (const HipError &e) {             \
            const int _rc = fail((ctx), e);       \
            if ((ctx) != root) { std::lock_guard<std::mutex> _l(root->lane_mu); root->last_error = (ctx)->last_error; } \
            return _rc;                           \
        }                                         \
    } catch (const HipError &e) { return fail(root, e); } \
    catch (const std::bad_alloc &) { return ZKG16_ERR_OOM; } \
    return ZKG16_OK;

// ---- host <-> ABI point conversions (u64 limbs and u32 limbs share the little-endian byte layout)
G1Affine g1_from_abi(const uint64_t *l, int inf) {
    G1Affine p;
    if (inf) return G1Affine::inf();
    memcpy(&p, l, sizeof p);
    return p;
}
G2Affine g2_from_abi(const uint64_t *l, int inf) {
    G2Affine p;
    if (inf) return G2Affine::inf();
    memcpy(&p, l, sizeof p);
    return p;
}
template <class A>
void point_to_abi(const A &p, uint64_t *out, uint8_t *inf) {
    if (p.is_inf()) {
        memset(out, 0, sizeof p);
        if (inf) *inf = 1;
    } else {
        memcpy(out, &p, sizeof p);
        if (inf) *inf = 0;
    }
}

// Upload a slice [lo, hi) of a saturated affine query vector and convert it into the unsaturated device form at
// dst[0 .. hi-lo); flagged-infinity points become (0,0).
template <class A> struct UOf;
template <> struct UOf<G1Affine> { using T = G1AffineU; };
template <> struct UOf<G2Affine> { using T = G2AffineU; };
inline void convert_bases(zkg16_ctx *ctx, const G1Affine *in, G1AffineU *out, size_t n) { convert_g1_bases(ctx, in, out, n); }
inline void convert_bases(zkg16_ctx *ctx, const G2Affine *in, G2AffineU *out, size_t n) { convert_g2_bases(ctx, in, out, n); }

template <class A>
void upload_points(zkg16_ctx *ctx, typename UOf<A>::T *dst, const uint64_t *src, const uint8_t *inf, size_t lo, size_t hi) {
    if (hi <= lo) return;
    const size_t n = hi - lo;
    const A *s = reinterpret_cast<const A *>(src) + lo;
    DevBuf stage(n * sizeof(A));
    if (!inf) {
        ZK_HIP(hipMemcpyAsync(stage.p, s, n * sizeof(A), hipMemcpyHostToDevice, ctx->stream));
    } else {
        std::vector<A> tmp(s, s + n);
        for (size_t i = 0; i < n; i++)
            if (inf[lo + i]) tmp[i] = A::inf();
        ZK_HIP(hipMemcpyAsync(stage.p, tmp.data(), n * sizeof(A), hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));      // tmp is freed at scope exit
    }
    convert_bases(ctx, stage.as<A>(), dst, n);
    ZK_HIP(hipStreamSynchronize(ctx->stream));
}

template <class A>
void upload_one(zkg16_ctx *ctx, typename UOf<A>::T *dst, const A &p) {
    const typename UOf<A>::T u{to_u(p.x), to_u(p.y)};     // host-side conversion (same templates)
    ZK_HIP(hipMemcpyAsync(dst, &u, sizeof(u), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
}

}  // namespace
namespace zk {
// ---- device allocation cache (see common.hpp)
namespace {
struct DevCache {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> free_blocks;     // by (device, size)
    size_t cached_bytes = 0;
    static constexpr size_t LIMIT = (size_t)96 << 30, MIN_CACHED = (size_t)1 << 20;
};
DevCache &dev_cache() {
    static DevCache c;
    return c;
}
}  // namespace
void *dev_acquire(size_t bytes, size_t *got) {
    DevCache &c = dev_cache();
    if (bytes >= DevCache::MIN_CACHED) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.free_blocks.lower_bound(std::make_pair(dev, bytes));
        if (it != c.free_blocks.end() && it->first.first == dev && it->first.second <= bytes + bytes / 4) {      // at most 25 % larger than asked
            void *p = it->second;
            *got = it->first.second;
            c.cached_bytes -= it->first.second;
            c.free_blocks.erase(it);
            return p;
        }
    }
    void *p = nullptr;
    const bool trace = getenv("ZKG16_TRACE_ALLOC") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, bytes);
    if (trace) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 1.0) fprintf(stderr, "dev_acquire: hipMalloc of %.1f MB took %.2f ms\n", bytes / 1048576.0, ms);
    }
    if (e != hipSuccess) {                      // out of memory with blocks parked in the cache: give them back and retry once
        (void)hipGetLastError();
        dev_cache_flush();
        ZK_HIP(hipMalloc(&p, bytes));
    }
    *got = bytes;
    return p;
}
void dev_release(void *p, size_t bytes) noexcept {
    DevCache &c = dev_cache();
    if (bytes >= DevCache::MIN_CACHED) {
        const bool trace = getenv("ZKG16_TRACE_ALLOC") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        (void)hipDeviceSynchronize();           // what hipFree would have done
        if (trace) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > 1.0) fprintf(stderr, "dev_release: sync before caching %.1f MB took %.2f ms\n", bytes / 1048576.0, ms);
        }
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(c.mu);
        if (c.cached_bytes + bytes <= DevCache::LIMIT) {
            c.free_blocks.emplace(std::make_pair(dev, bytes), p);
            c.cached_bytes += bytes;
            return;
        }
    }
    const bool trace = getenv("ZKG16_TRACE_ALLOC") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    (void)hipFree(p);
    if (trace) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 1.0) fprintf(stderr, "dev_release: hipFree of %.1f MB took %.2f ms\n", bytes / 1048576.0, ms);
    }
}
void dev_cache_flush() noexcept {
    DevCache &c = dev_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &kv : c.free_blocks) {
        (void)hipSetDevice(kv.first.first);
        (void)hipFree(kv.second);
    }
    (void)hipSetDevice(cur);
    c.free_blocks.clear();
    c.cached_bytes = 0;
}
}  // namespace zk
namespace {

struct Partials {
    G1XYZZ h, l, a, b1;
    G2XYZZ b2;
    // un-sharded proofs: s*(a + alpha) and r*(b1 + beta) are formed on the host as soon as A and B1 are collected, while the
    // device still works on the remaining MSMs (they are ~0.35 ms of the 0.4 ms host tail)
    bool have_early = false;
    G1XYZZ s_a, r_b1;
};

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// events of one proof, destroyed on every exit path
struct EventSet {
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    EventSet() { for (auto &e : ev) ZK_HIP(hipEventCreate(&e)); }
    ~EventSet() { for (auto &e : ev) if (e) (void)hipEventDestroy(e); }
    EventSet(const EventSet &) = delete;
    EventSet &operator=(const EventSet &) = delete;
};

// The device part of a proof: witness map + the five MSMs over this ctx's pk shard.  A shard is a pair of index ranges:
// [z_lo, z_hi) of the a / b_g1 / b_g2 / l queries and [h_lo, h_hi) of h_query.  A rank whose h range is empty skips the
// witness map and the H MSM altogether, one whose z range is empty (and which does not carry the r, s, -rs terms) skips the
// four z-side MSMs — this is what lets the ranks of a multi-GPU proof take different roles (zkg16_shard_plan).
// zp (zkg16_prove_matrix): the assignment is still being produced — part k of it becomes valid when zp->produce(k) has queued
// its kernels on the main stream.  The z-side MSMs then run in rounds, one per part, over the terms of that part only (digits
// of the other scalars count as zero), each round's buckets are summed into the MSM's bucket array and ONE reduction follows;
// the witness map waits for the last part.
void prove_device(zkg16_ctx *ctx, PkDev &pk, R1csDev &rc, WitnessDev &wit, const Fr &r, const Fr &s, Partials &out,
                  const std::function<void()> *before_witness_map = nullptr, const ZParts *zp = nullptr) {
    const size_t m_total = rc.num_variables;
    if (wit.n != m_total || pk.m_total != m_total) throw HipError{hipErrorInvalidValue, "prove: assignment / key length mismatch", __FILE__, __LINE__};
    const size_t N = (size_t)1 << rc.log_n;
    if (pk.n_h_total != N - 1) throw HipError{hipErrorInvalidValue, "prove: h_query length != N-1", __FILE__, __LINE__};
    EventSet evs;                         // 0-1: z-side sort (main stream), 2-4: witness map / h-side sort (aux stream)
    hipEvent_t *ev = evs.ev;
    // A throw between the first enqueue and the last collect (e.g. out of memory in a slot's bucket array) must not leave
    // kernels of this proof in flight: the next proof on the ctx rewrites extra_host and reuses the workspaces and slots.
    struct DrainOnError {
        zkg16_ctx *c;
        bool ok = false;
        ~DrainOnError() {
            if (ok) return;
            (void)hipStreamSynchronize(c->stream);
            (void)hipStreamSynchronize(c->wm_stream);
            for (auto &sl : c->slots) {
                if (sl.stream) (void)hipStreamSynchronize(sl.stream);
                sl.active = sl.pending_reduce = sl.fixups_pending = sl.last_of_proof = false;
            }
        }
    } drain{ctx};
    const double t0 = now_ms();

    // ---- main stream: the z-side scalar vector (z-slice || r, s, -rs), read in place by the digit kernel -> digits -> sort.
    // It does not depend on the witness map, so the G2 accumulation can start while h is still being computed on the aux stream.
    const size_t nz = pk.z_hi - pk.z_lo;
    const size_t nzs = nz + 3;                                 // + r, s, -rs slots
    const size_t nh = pk.h_hi - pk.h_lo;
    const bool z_side = nz > 0 || pk.blinding;
    MsmPlan plan_z, plan_h, plan_zb;
    const bool b_sparse = pk.b_skipped * 20 > nzs;             // > 5 % of the terms: worth a second (0.3 ms) sort
    out.h = out.l = out.a = out.b1 = G1XYZZ::inf();
    out.b2 = G2XYZZ::inf();
    ZK_HIP(hipEventRecord(ev[0], ctx->stream));
    ctx->ws_z.last_tb = ctx->ws_zb.last_tb = ctx->ws_h.last_tb = 0;
    const bool trace = getenv("ZKG16_TRACE_HOST") != nullptr;
    MsmWorkspace &wsb = b_sparse ? ctx->ws_zb : ctx->ws_z;
    const MsmPlan &planb = b_sparse ? plan_zb : plan_z;
    const int parts = zp ? zp->parts : 1;
    const bool rounds = parts > 1 && z_side && zp->part_of != nullptr;
    // ---- the four z-side accumulations go onto the main stream BEFORE the witness map's ~40 launches (G2 first: its long
    // reduction then hides behind the G1 accumulations); their reductions — whose first packet is a wait — only after those
    // launches (msm_enqueue_reduce).  Kernel trace at n = 32: queued after the witness map, the G2 accumulation started
    // 2.0 ms into the proof with its inputs ready at 0.6 ms.
    // (Tried and dropped, 128x128: sorting the B-side list first and the full list on another stream underneath the G2
    // accumulation, with the witness map held back until the first list exists — the accumulation then starts 9 instead of
    // 15 ms into the proof, and the proof takes the same 171-172 ms: kernels that share the device slow each other by about
    // what the overlap saves, the proof is the SUM of its kernels' work.  A high-priority witness-map stream: +1 ms.)
    auto enqueue_z_accs = [&](int round) {
        if (!z_side) return;
        msm_g2_enqueue_acc(ctx, wsb, planb, pk.b2.as<G2AffineU>(), ctx->slots[0], round);
        msm_g1_enqueue_acc(ctx, ctx->ws_z, plan_z, pk.l.as<G1AffineU>(), ctx->slots[2], round);
        msm_g1_enqueue_acc(ctx, ctx->ws_z, plan_z, pk.a.as<G1AffineU>(), ctx->slots[3], round);
        msm_g1_enqueue_acc(ctx, wsb, planb, pk.b1.as<G1AffineU>(), ctx->slots[4], round);
    };
    if (zp && !rounds)
        for (int k = 0; k < parts; k++) zp->produce(k);       // no rounds (one part, or no z side here): the whole assignment first
    if (z_side) {
        if (!ctx->extra_host) ZK_HIP(hipHostMalloc(&ctx->extra_host, 3 * sizeof(Fr), hipHostMallocDefault));
        Fr *extra = reinterpret_cast<Fr *>(ctx->extra_host);
        extra[0] = pk.blinding ? r : Fr::zero();              // the r/s/-rs terms are added by one shard only
        extra[1] = pk.blinding ? s : Fr::zero();
        extra[2] = pk.blinding ? fp_neg(fp_mul(r, s)) : Fr::zero();
        // the digit kernel reads the three extra scalars straight from this pinned (device-visible) host buffer: no host-to-device
        // copy is queued; the buffer is rewritten only by the next proof, which starts after this one has been collected
        for (int k = 0; k < (rounds ? parts : 1); k++) {
            if (rounds) zp->produce(k);
            ScalarSrc zsrc{wit.z.as<Fr>() + pk.z_lo, nz, extra, 3, true, nullptr};
            if (rounds) { zsrc.part = zp->part_of; zsrc.want_part = k; }
            const int tz = pk.tab_c_z;
            msm_plan_build(ctx, ctx->ws_z, zsrc, plan_z, tz, tz != 0);
            if (b_sparse) {      // B1 and B2 share a plan without the terms whose bases are infinity (see b_density_mask_kernel)
                // (measured: 128x128 with window tables 154.5 -> 153.65 ms, 32x32 11.97 -> 11.68; with a plain key 168.55 -> 169.2, so only with tables
                // unless option b_filter = 1 asks for it)
                if ((ctx->opt_b_filter == 1 || (ctx->opt_b_filter == 0 && tz != 0)) && ctx->opt_sort_mode == 0) {
                    msm_plan_filter(ctx, ctx->ws_z, plan_z, pk.b_mask.as<uint8_t>(), ctx->ws_zb, plan_zb);
                } else {
                    zsrc.mask = pk.b_mask.as<uint8_t>();
                    msm_plan_build(ctx, ctx->ws_zb, zsrc, plan_zb, tz, tz != 0);
                }
            }
            if (rounds) enqueue_z_accs(k);
        }
    }
    ZK_HIP(hipEventRecord(ev[1], ctx->stream));
    if (zp) ZK_HIP(hipEventRecord(ev[5], ctx->stream));       // z is complete once the main stream gets here
    // the z-side accumulations either start at once (their kernels and the witness map's then share the device) or wait for the
    // witness map, which then has the device to itself and lets the h-side sort run underneath the accumulations.  Measured: with
    // window tables 128x128 154.5 vs 153.65 ms (32x32 11.7 vs 12.2, 46x46 19.7 vs 20.3), with a plain key 128x128 168.6 vs 171.2 —
    // so by default only from 2^23 on and only for keys with tables; never when the matrices are still being uploaded on the
    // witness map's stream (the accumulations are what hides that).
    const int wm_first = (!nh || rounds) ? 0 : ctx->opt_wm_first >= 0 ? ctx->opt_wm_first : (rc.log_n >= 23 && pk.tab_c_h != 0 && !before_witness_map) ? 1 : 0;
    if (!wm_first && !rounds) enqueue_z_accs(-1);

    // ---- R1CS -> QAP witness map (a3-a5 of SURVEY.md 8a) and the h-side sort, on a third stream concurrently with the
    // z-side work (measured in one process, n = 32: 18.15 vs 18.58 ms in order; n = 12: 10.05 vs 11.16 ms)
    const bool wm_concurrent = ctx->opt_wm_concurrent != 0;
    if (wm_concurrent) std::swap(ctx->stream, ctx->wm_stream);      // every launch helper targets ctx->stream
    try {
        // zkg16_prove (host pointers): the matrices are uploaded here, on the witness map's stream, while the z-side
        // accumulations queued above already keep the device busy
        if (before_witness_map) (*before_witness_map)();
        if (zp) ZK_HIP(hipStreamWaitEvent(ctx->stream, ev[5], 0));
        ZK_HIP(hipEventRecord(ev[2], ctx->stream));
        Fr *h = nullptr;
        if (nh) witness_map_run(ctx, rc, wit.z.as<Fr>(), &h);
        ZK_HIP(hipEventRecord(ev[3], ctx->stream));
        if (nh) {
            const ScalarSrc hsrc{h + pk.h_lo, nh, nullptr, 0, true, nullptr};
            msm_plan_build(ctx, ctx->ws_h, hsrc, plan_h, pk.tab_c_h ? pk.tab_c_h : ctx->opt_window_bits_h, pk.tab_c_h != 0);
        }
        ZK_HIP(hipEventRecord(ev[4], ctx->stream));
    } catch (...) {
        if (wm_concurrent) std::swap(ctx->stream, ctx->wm_stream);
        throw;
    }
    if (wm_concurrent) std::swap(ctx->stream, ctx->wm_stream);
    if (wm_first) {
        ZK_HIP(hipStreamWaitEvent(ctx->stream, wm_first == 2 ? ev[4] : ev[3], 0));
        enqueue_z_accs(-1);
    }

    if (z_side) {
        msm_g2_enqueue_reduce(ctx, ctx->slots[0]);
        msm_g1_enqueue_reduce(ctx, ctx->slots[2]);
        msm_g1_enqueue_reduce(ctx, ctx->slots[3]);
        msm_g1_enqueue_reduce(ctx, ctx->slots[4]);
    }
    if (trace) fprintf(stderr, "host: z-side msms queued at %.3f ms\n", now_ms() - t0);

    // ---- H last: it is the only MSM that waits for the witness map; then collect — each MSM's host Horner overlaps the
    // device work still queued behind it.  (With the witness map first H could go second — its list sorted underneath the G2
    // accumulation — and L, with a quarter of H's buckets and nothing to do on the host afterwards, last: measured 154.7 against
    // 153.65 ms, the shorter tail does not pay for the earlier scatter.)
    ZK_HIP(hipStreamWaitEvent(ctx->stream, ev[4], 0));
    ctx->slots[1].last_of_proof = true;
    if (nh) msm_g1_enqueue(ctx, ctx->ws_h, plan_h, pk.h.as<G1AffineU>(), ctx->slots[1]);
    if (trace) fprintf(stderr, "host: h queued at %.3f ms\n", now_ms() - t0);
    double tprev = now_ms();
    auto lap = [&](int idx) { const double t = now_ms(); ctx->timings[idx] = (float)(t - tprev); tprev = t; };
    const bool early = pk.full;
    auto collect_b2 = [&] { out.b2 = msm_g2_collect(ctx, ctx->slots[0]); };
    auto collect_l = [&] { out.l = msm_g1_collect(ctx, ctx->slots[2]); };
    auto collect_a = [&] {
        out.a = msm_g1_collect(ctx, ctx->slots[3]);
        if (early) {
            G1XYZZ A = out.a;
            xyzz_madd(A, pk.alpha_g1, false);
            const Fr sc = fp_from_mont(s);
            out.s_a = xyzz_mul(A, sc.l);
        }
    };
    auto collect_b1 = [&] {
        out.b1 = msm_g1_collect(ctx, ctx->slots[4]);
        if (early) {
            G1XYZZ B1 = out.b1;
            xyzz_madd(B1, pk.beta_g1, false);
            const Fr rc_ = fp_from_mont(r);
            out.r_b1 = xyzz_mul(B1, rc_.l);
        }
    };
    // A plain key's MSMs come back as one sum per window (and per weight bit with the bit-sliced reduction): ~0.3 ms of host
    // additions per G1 MSM and ~1 ms for the G2 one.  One after the other they outlast the device on small and mid-size circuits
    // (8x8: 2.4 ms of host work in a 4.4 ms proof), so each collect gets its own thread: it waits for its MSM's event, then combines.
    const bool threaded = z_side && ctx->opt_collect_threads != 0 &&
                          (ctx->opt_collect_threads == 1 || ctx->slots[0].nwin > 1 || ctx->slots[2].nwin > 1);
    if (z_side && threaded) {
        std::exception_ptr err[4];
        {
            ThreadGroup tg;
            const int device = ctx->device;
            auto guarded = [&err, device](int i, auto &job) {
                return [&err, device, i, &job] {
                    try {
                        (void)hipSetDevice(device);
                        job();
                    } catch (...) {
                        err[i] = std::current_exception();
                    }
                };
            };
            tg.run(guarded(0, collect_b2));
            tg.run(guarded(1, collect_l));
            tg.run(guarded(2, collect_a));
            tg.run(guarded(3, collect_b1));
            try {
                if (nh) out.h = msm_g1_collect(ctx, ctx->slots[1]);
                else ZK_HIP(hipStreamSynchronize(ctx->stream));
            } catch (...) {
                tg.join();
                throw;
            }
            lap(3);
            tg.join();
        }
        for (auto &e : err)
            if (e) std::rethrow_exception(e);
        out.have_early = early;
        ctx->timings[4] = ctx->timings[5] = ctx->timings[6] = 0;
        lap(7);      // what the slowest z-side collect took beyond H's
    } else {
        if (z_side) {
            collect_b2(); lap(7);
            collect_l(); lap(4);
            collect_a(); lap(5);
            collect_b1(); lap(6);
            out.have_early = early;
        } else {
            ctx->timings[4] = ctx->timings[5] = ctx->timings[6] = ctx->timings[7] = 0;
        }
        if (nh) out.h = msm_g1_collect(ctx, ctx->slots[1]);
        else ZK_HIP(hipStreamSynchronize(ctx->stream));
        lap(3);
    }
    float ms;
    // [1] witness map, [2] digits+sort of both vectors (device time); [3..7] host-observed completion gaps of H, L, A, B1, B2
    // (collected in the order B2, L, A, B1, H — the first gap contains most of the device time: NOT a breakdown);
    // [10..14] device time of the bucket accumulation (+ fix-ups) of H, L, A, B1, B2, [15..19] of their bucket reductions, from
    // event pairs on the streams they ran on — the per-stage times upstream's spans ("Compute C" = H + L, "Compute A",
    // "Compute B in G1", "Compute B in G2": ark-groth16 prover.rs) correspond to.  Kernels of different MSMs overlap, so these
    // sum to more than the proof.
    {
        const int slot_of[5] = {1, 2, 3, 4, 0};     // H, L, A, B1, B2
        for (int k = 0; k < 5; k++) {
            MsmSlot &sl = ctx->slots[slot_of[k]];
            const bool ran = k == 0 ? nh > 0 : z_side;
            float acc_ms = 0, red_ms = 0;
            if (ran && sl.acc_start && hipEventElapsedTime(&acc_ms, sl.acc_start, sl.acc_done) != hipSuccess) { acc_ms = 0; (void)hipGetLastError(); }
            if (ran && sl.red_start && hipEventElapsedTime(&red_ms, sl.red_start, sl.red_done) != hipSuccess) { red_ms = 0; (void)hipGetLastError(); }
            ctx->timings[10 + k] = acc_ms;
            ctx->timings[15 + k] = red_ms;
        }
    }
    // [20] host Horner of H's window sums (after the last device event of the proof: exposed), [21] of the other four (overlap H's device work)
    ctx->timings[20] = nh ? ctx->slots[1].collect_host_ms : 0;
    ctx->timings[21] = z_side ? ctx->slots[0].collect_host_ms + ctx->slots[2].collect_host_ms + ctx->slots[3].collect_host_ms + ctx->slots[4].collect_host_ms : 0;
    ctx->timings[0] = 0;
    ZK_HIP(hipEventElapsedTime(&ms, ev[2], ev[3]));
    ctx->timings[1] = ms;
    ZK_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
    ctx->timings[2] = ms;
    ZK_HIP(hipEventElapsedTime(&ms, ev[3], ev[4]));
    ctx->timings[2] += ms;
    ctx->timings[9] = (float)(now_ms() - t0);
    drain.ok = true;
}

// Host tail (a9): A = alpha + MSM_a, B = beta + MSM_b, C = s*A + r*B1 + MSM_l + MSM_h.
// (r*delta, s*delta and -rs*delta already ride inside the MSMs as three extra (base, scalar) slots.)
void prove_tail_pts(const G1Affine &alpha_g1, const G1Affine &beta_g1, const G2Affine &beta_g2, const Fr &r, const Fr &s,
                    const Partials &p, uint64_t *proof_out, uint8_t *inf_out) {
    G1XYZZ A = p.a;
    xyzz_madd(A, alpha_g1, false);
    G1XYZZ B1 = p.b1;
    xyzz_madd(B1, beta_g1, false);
    G2XYZZ B2 = p.b2;
    xyzz_madd(B2, beta_g2, false);
    G1XYZZ C, rB;
    if (p.have_early) {
        C = p.s_a;
        rB = p.r_b1;
    } else {
        const Fr rc = fp_from_mont(r), sc = fp_from_mont(s);
        C = xyzz_mul(A, sc.l);
        rB = xyzz_mul(B1, rc.l);
    }
    xyzz_add(C, rB);
    xyzz_add(C, p.l);
    xyzz_add(C, p.h);
    point_to_abi(xyzz_to_affine(A), proof_out, inf_out);
    point_to_abi(xyzz_to_affine(B2), proof_out + 12, inf_out + 1);
    point_to_abi(xyzz_to_affine(C), proof_out + 36, inf_out + 2);
}
void prove_tail(PkDev &pk, const Fr &r, const Fr &s, const Partials &p, uint64_t *proof_out, uint8_t *inf_out) {
    prove_tail_pts(pk.alpha_g1, pk.beta_g1, pk.beta_g2, r, s, p, proof_out, inf_out);
}

void sum_partials(Partials &p, const uint64_t *partials, const uint8_t *partial_inf, int n_ranks);

Fr fr_from_abi(const uint64_t *l) {
    Fr v;
    memcpy(&v, l, sizeof v);
    return v;
}

// validate + allocate (nothing is copied yet)
int r1cs_create(const uint64_t *const rp[3], const uint32_t *const col[3], const uint64_t *const cf[3], size_t num_instance,
                size_t num_constraints, size_t num_variables, std::unique_ptr<R1csDev> &out) {
    if (num_instance == 0) return ZKG16_ERR_BAD_ARG;
    for (int i = 0; i < 3; i++)
        if (!rp[i] || (rp[i][num_constraints] && (!col[i] || !cf[i]))) return ZKG16_ERR_BAD_ARG;
    const size_t dom = num_constraints + num_instance;
    int log_n = 0;
    while (((size_t)1 << log_n) < dom) log_n++;
    if (log_n > 32) return ZKG16_ERR_DOMAIN_TOO_LARGE;      // ark: SynthesisError::PolynomialDegreeTooLarge
    if (log_n > 28) return ZKG16_ERR_DOMAIN_TOO_LARGE;      // build limit (three-pass NTT covers 2^31; 32-bit entry indices cap the MSMs)
    auto r = std::make_unique<R1csDev>();
    r->num_instance = num_instance;
    r->num_constraints = num_constraints;
    r->num_variables = num_variables;
    r->log_n = log_n;
    for (int i = 0; i < 3; i++) {
        const size_t nnz = rp[i][num_constraints];
        // row pointers: start at 0, never decrease, end at nnz — spmv_kernel walks [rp[row], rp[row+1]) unchecked on the device
        if (rp[i][0] != 0) return ZKG16_ERR_BAD_ARG;
        // both scans in slices on a few host threads (86.6 M column indices in the 128x128 circuit: ~0.1 s on one)
        const int T = (nnz + num_constraints) >= ((size_t)1 << 22) ? 8 : 1;
        std::vector<int> bad(T, 0);
        auto scan = [&](int t) {
            const size_t r0 = num_constraints * t / T, r1 = num_constraints * (t + 1) / T;
            for (size_t row = r0; row < r1; row++)
                if (rp[i][row] > rp[i][row + 1]) { bad[t] = 1; return; }
            const size_t k0 = nnz * t / T, k1 = nnz * (t + 1) / T;
            uint32_t top = 0;
            for (size_t k = k0; k < k1; k++) top = col[i][k] > top ? col[i][k] : top;
            if (k1 > k0 && top >= num_variables) bad[t] = 1;
        };
        {
            ThreadGroup tg;
            for (int t = 1; t < T; t++) tg.run([&scan, t]() { scan(t); });
            scan(0);
        }
        for (int t = 0; t < T; t++)
            if (bad[t]) return ZKG16_ERR_BAD_ARG;
        r->nnz[i] = nnz;
        r->rp[i].alloc((num_constraints + 1) * sizeof(uint64_t));
        r->col[i].alloc(nnz * sizeof(uint32_t));
        r->cf[i].alloc(nnz * sizeof(Fr));
    }
    out = std::move(r);
    return ZKG16_OK;
}
// Host -> device copy of a large pageable buffer, queued on ctx->stream.  hipMemcpyAsync from pageable memory goes through
// the runtime's own single-threaded staging (~9 GB/s measured: the 3.5 GB of a 128x128 R1CS took 0.38 s of the 0.55 s
// host-pointer proof); here four host threads fill one half of a pinned ring while the DMA engine drains the other.
static constexpr size_t STAGE_BYTES = (size_t)64 << 20;
void upload_h2d(zkg16_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes < ((size_t)8 << 20)) {
        ZK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return;
    }
    for (int i = 0; i < 2; i++)
        if (!ctx->stage_host[i]) {
            ZK_HIP(hipHostMalloc(&ctx->stage_host[i], STAGE_BYTES, hipHostMallocDefault));
            ZK_HIP(hipEventCreateWithFlags(&ctx->stage_done[i], hipEventDisableTiming));
        }
    const unsigned char *s = static_cast<const unsigned char *>(src);
    unsigned char *d = static_cast<unsigned char *>(dst);
    int slot = 0;
    for (size_t off = 0; off < bytes; off += STAGE_BYTES, slot ^= 1) {
        const size_t len = bytes - off < STAGE_BYTES ? bytes - off : STAGE_BYTES;
        ZK_HIP(hipEventSynchronize(ctx->stage_done[slot]));         // the copy that last used this half has left it (a fresh event is complete)
        unsigned char *stage = static_cast<unsigned char *>(ctx->stage_host[slot]);
        constexpr int T = 4;
        const size_t part = (len / T + 4095) & ~(size_t)4095;
        {
            ThreadGroup tg;
            for (int t = 1; t < T; t++) {
                const size_t lo = part * t < len ? part * t : len, hi = part * (t + 1) < len ? part * (t + 1) : len;
                tg.run([=]() { if (hi > lo) memcpy(stage + lo, s + off + lo, hi - lo); });
            }
            memcpy(stage, s + off, part < len ? part : len);
        }
        ZK_HIP(hipMemcpyAsync(d + off, stage, len, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipEventRecord(ctx->stage_done[slot], ctx->stream));
    }
}
// host -> device copies of the three matrices, queued on ctx->stream (the call returns when the last piece is staged)
void r1cs_copy(zkg16_ctx *ctx, R1csDev &r, const uint64_t *const rp[3], const uint32_t *const col[3], const uint64_t *const cf[3]) {
    for (int i = 0; i < 3; i++) {
        upload_h2d(ctx, r.rp[i].p, rp[i], (r.num_constraints + 1) * sizeof(uint64_t));
        if (r.nnz[i]) {
            upload_h2d(ctx, r.col[i].p, col[i], r.nnz[i] * sizeof(uint32_t));
            upload_h2d(ctx, r.cf[i].p, cf[i], r.nnz[i] * sizeof(Fr));
        }
    }
}
int load_r1cs(zkg16_ctx *ctx, const uint64_t *const rp[3], const uint32_t *const col[3], const uint64_t *const cf[3],
              size_t num_instance, size_t num_constraints, size_t num_variables, uint64_t *handle) {
    if (!handle) return ZKG16_ERR_BAD_ARG;
    std::unique_ptr<R1csDev> r;
    const int rc = r1cs_create(rp, col, cf, num_instance, num_constraints, num_variables, r);
    if (rc) return rc;
    r1cs_copy(ctx, *r, rp, col, cf);
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    *handle = ctx->next_handle++;
    ctx->r1cs.put(*handle, std::move(r));
    return ZKG16_OK;
}

void sum_partials(Partials &p, const uint64_t *partials, const uint8_t *partial_inf, int n_ranks) {
    p.h = p.l = p.a = p.b1 = G1XYZZ::inf();
    p.b2 = G2XYZZ::inf();
    for (int k = 0; k < n_ranks; k++) {
        const uint64_t *q = partials + 72 * (size_t)k;
        const uint8_t *f = partial_inf + 5 * (size_t)k;
        xyzz_madd(p.h, g1_from_abi(q, f[0]), false);
        xyzz_madd(p.l, g1_from_abi(q + 12, f[1]), false);
        xyzz_madd(p.a, g1_from_abi(q + 24, f[2]), false);
        xyzz_madd(p.b1, g1_from_abi(q + 36, f[3]), false);
        xyzz_madd(p.b2, g2_from_abi(q + 48, f[4]), false);
    }
}

}  // namespace

extern "C" {

const char *zkg16_version(void) { return k_version; }

const char *zkg16_strerror(int status) {
    switch (status) {
        case ZKG16_OK: return "ok";
        case ZKG16_ERR_BAD_ARG: return "bad argument";
        case ZKG16_ERR_DOMAIN_TOO_LARGE: return "evaluation domain too large (ark: PolynomialDegreeTooLarge)";
        case ZKG16_ERR_HIP: return "HIP runtime error";
        case ZKG16_ERR_OOM: return "out of memory";
        case ZKG16_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
        case ZKG16_ERR_BAD_HANDLE: return "unknown handle";
        case ZKG16_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown status";
    }
}

const char *zkg16_last_error(zkg16_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int zkg16_init(const int *device_ids, int n_devices, zkg16_ctx **out) {
    if (!out) return ZKG16_ERR_BAD_ARG;
    *out = nullptr;
    if (n_devices != 1 && !(n_devices == 0 && !device_ids)) return ZKG16_ERR_UNSUPPORTED;   // one GPU per ctx / process
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return ZKG16_ERR_NO_DEVICE;
    }
    const int dev = (n_devices == 1 && device_ids) ? device_ids[0] : 0;
    if (dev < 0 || dev >= count) return ZKG16_ERR_BAD_ARG;
    auto *ctx = new (std::nothrow) zkg16_ctx();
    if (!ctx) return ZKG16_ERR_OOM;
    ctx->device = dev;
    try {
        ZK_HIP(hipSetDevice(dev));
        hipDeviceProp_t prop;
        ZK_HIP(hipGetDeviceProperties(&prop, dev));
        ctx->num_cus = prop.multiProcessorCount;
        // the code object holds gfx950 kernels only (no fallback path, no other ISA): any other device is "no device"
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            ctx->last_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
            delete ctx;
            return ZKG16_ERR_NO_DEVICE;
        }
        create_streams(ctx);
    } catch (const HipError &e) {
        int rc = fail(ctx, e);
        delete ctx;
        return rc;
    }
    *out = ctx;
    return ZKG16_OK;
}

void zkg16_destroy(zkg16_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto &l : ctx->lanes)
        if (l) {
            {
                std::lock_guard<std::mutex> lk(l->mu);      // a proof still running on that lane finishes first
                teardown(l.get());
            }
            l.reset();
        }
    teardown(ctx);
    ctx->pks.clear();
    ctx->r1cs.clear();
    ctx->wits.clear();
    ctx->ntt_tables.clear();
    delete ctx;
    dev_cache_flush();      // a destroyed ctx really returns its memory (other live contexts simply allocate afresh)
}

}  // extern "C"
namespace {
int set_option_one(zkg16_ctx *ctx, const char *name, int64_t value) {
    if (!strcmp(name, "window_bits")) {
        if (value != 0 && (value < 2 || value > 20)) return ZKG16_ERR_BAD_ARG;
        ctx->opt_window_bits = (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "min_seg")) {
        if (value < 0 || value > 4096) return ZKG16_ERR_BAD_ARG;
        ctx->opt_min_seg = (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "ntt_mode")) {
        if (value != 0 && value != 1 && value != 3) return ZKG16_ERR_BAD_ARG;      // 3: unsaturated, but three passes above 2^22
        ctx->opt_ntt_mode = (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "reduce_mode")) {
        // 4 = classic everywhere (0 restores the default, 3); 5 = bit-sliced wherever it applies; 6 = the default without the bit-sliced form
        if (value < 0 || value > 6) return ZKG16_ERR_BAD_ARG;
        ctx->opt_reduce_mode = value == 0 ? 3 : (value == 4 ? 0 : (int)value);
        return ZKG16_OK;
    }
    if (!strcmp(name, "g1_waves")) {
        if (value < 0 || value > 4) return ZKG16_ERR_BAD_ARG;
        ctx->opt_g1_waves = (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "fixup_aux")) {
        ctx->opt_fixup_aux = value ? 1 : 0;
        return ZKG16_OK;
    }
    if (!strcmp(name, "window_bits_h")) {
        if (value != 0 && (value < 2 || value > 20)) return ZKG16_ERR_BAD_ARG;
        ctx->opt_window_bits_h = (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "wm_concurrent")) {      // -1 auto (default), 0 in-order, 1 third stream
        ctx->opt_wm_concurrent = (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "ntt_radix")) {          // 1 (default; also 0): the last seven stages by lane exchanges, the others one per LDS trip; 2: every stage through the LDS; 4: two per trip (radix 4)
        if (value != 0 && value != 1 && value != 2 && value != 3 && value != 4) return ZKG16_ERR_BAD_ARG;      // 3: the top seven stages by lane exchanges too
        ctx->opt_ntt_radix = value == 0 ? 1 : (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "ntt_xcd")) {            // 1 (default): XCD-aware tile order in the NTT passes; 2 = off (0 restores the default)
        if (value < 0 || value > 2) return ZKG16_ERR_BAD_ARG;
        ctx->opt_ntt_xcd = value == 2 ? 0 : 1;
        return ZKG16_OK;
    }
    if (!strcmp(name, "acc_debug")) {          // timing probes of the accumulation kernels; results are WRONG while set
        if (value < 0 || value > 15) return ZKG16_ERR_BAD_ARG;
        ctx->opt_acc_debug = (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "sort_mode")) {          // 0 (default): hand-written wave-ballot bucket scatter; 1: rocPRIM device radix sort
        if (value < 0 || value > 1) return ZKG16_ERR_BAD_ARG;
        ctx->opt_sort_mode = (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "acc_pipeline")) {       // bit 0: G1, bit 1: G2 software-pipelined gather; 0 (default) and 4: neither.  With four G1
        // waves per SIMD and window tables the plain form (gather right before its addition) measured 153.6 against 157.4 ms per
        // 128x128 proof, and the same at every other size and with plain keys (profiles/ab_options_r2_final.txt)
        if (value < 0 || value > 4) return ZKG16_ERR_BAD_ARG;
        ctx->opt_acc_pipeline = value == 4 ? 0 : (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "fuse_pointwise")) {     // 1 (default): (ab - c)/Z on the load of the seventh transform; 0: own pass
        ctx->opt_fuse_pointwise = value ? 1 : 0;
        return ZKG16_OK;
    }
    if (!strcmp(name, "wm_first")) {           // -1 (default): 1 from 2^23 on for keys with window tables; 0: z-side accumulations start at once; 1: after the witness map; 2: after the h-side sort too
        if (value < -1 || value > 2) return ZKG16_ERR_BAD_ARG;
        ctx->opt_wm_first = (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "spmv_dict")) {          // 0 / 1 (default): 16-bit coefficient dictionary in the SpMV; 2: 32-byte coefficients
        if (value < 0 || value > 2) return ZKG16_ERR_BAD_ARG;
        ctx->opt_spmv_dict = (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "b_filter")) {           // B-side term list = the sorted full list minus the masked terms: 0 (default) with window tables, 1 always; 2: second sort
        if (value < 0 || value > 2) return ZKG16_ERR_BAD_ARG;
        ctx->opt_b_filter = (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "collect_threads")) {    // host combination of the MSMs' window sums: 0 = by key kind (threads for plain keys), 1 = always threaded, 2 = never
        if (value < 0 || value > 2) return ZKG16_ERR_BAD_ARG;
        ctx->opt_collect_threads = value == 0 ? -1 : value == 2 ? 0 : 1;
        return ZKG16_OK;
    }
    if (!strcmp(name, "fixed_base_bits")) {    // setup's fixed-base windows: 0 = by batch size, else 4..14 (ladder-built table) or 16 / 18 / 20 (two-level)
        if (value != 0 && (value < 4 || value > 20 || (value > 14 && (value & 1)))) return ZKG16_ERR_BAD_ARG;
        ctx->opt_fixed_base_bits = (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "g2_lazy")) {            // G2 bucket accumulation: 0 / 1 (default) = Fq2 products as two fused two-product reductions (LDS-parked operands), 2 = Karatsuba with three
        if (value < 0 || value > 2) return ZKG16_ERR_BAD_ARG;
        ctx->opt_g2_lazy = value == 2 ? 0 : 1;
        return ZKG16_OK;
    }
    if (!strcmp(name, "matrix_parts")) {       // zkg16_prove_matrix: slices of the host sponges the proof is fed in (0 = five growing slices; k = k equal ones; 1 = no overlap: assignment first)
        if (value < 0 || value > 8) return ZKG16_ERR_BAD_ARG;
        ctx->opt_matrix_parts = (int)value;
        return ZKG16_OK;
    }
    if (!strcmp(name, "reduce_chunk")) {
        if (value != 0 && (value < 1 || value > 64 || (value & (value - 1)))) return ZKG16_ERR_BAD_ARG;
        ctx->opt_reduce_chunk = (int)value;
        return ZKG16_OK;
    }
    return ZKG16_ERR_UNSUPPORTED;
}
}  // namespace
extern "C" {

// Options apply to every lane of the ctx (a lane created later copies the root's).  "lanes": proofs this ctx runs at a time
// (1..8, default 2; callers beyond that wait).
int zkg16_set_option(zkg16_ctx *ctx, const char *name, int64_t value) {
    if (!ctx || !name) return ZKG16_ERR_BAD_ARG;
    if (!strcmp(name, "lanes")) {
        if (value < 1 || value > 8) return ZKG16_ERR_BAD_ARG;
        std::lock_guard<std::mutex> lk(ctx->lane_mu);
        ctx->opt_lanes = (int)value;
        return ZKG16_OK;
    }
    int rc;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        rc = set_option_one(ctx, name, value);
    }
    if (rc != ZKG16_OK) return rc;
    std::vector<zkg16_ctx *> ls;
    {
        std::lock_guard<std::mutex> lk(ctx->lane_mu);
        for (auto &l : ctx->lanes)
            if (l) ls.push_back(l.get());
    }
    for (zkg16_ctx *l : ls) {
        std::lock_guard<std::mutex> lk(l->mu);
        (void)set_option_one(l, name, value);
    }
    return ZKG16_OK;
}

int zkg16_pk_load_range(zkg16_ctx *ctx,
                        const uint64_t *a_query, const uint8_t *a_inf, size_t n_a,
                        const uint64_t *b_g1_query, const uint8_t *b_g1_inf, size_t n_b1,
                        const uint64_t *b_g2_query, const uint8_t *b_g2_inf, size_t n_b2,
                        const uint64_t *h_query, const uint8_t *h_inf, size_t n_h,
                        const uint64_t *l_query, const uint8_t *l_inf, size_t n_l,
                        const uint64_t alpha_g1[12], const uint64_t beta_g1[12], const uint64_t beta_g2[24],
                        const uint64_t delta_g1[12], const uint64_t delta_g2[24],
                        size_t num_instance, size_t z_lo, size_t z_hi, size_t h_lo, size_t h_hi, int blinding, uint64_t *pk_handle) {
    if (!pk_handle || !a_query || !b_g1_query || !b_g2_query || (!h_query && n_h) || (!l_query && n_l) || !alpha_g1 || !beta_g1 ||
        !beta_g2 || !delta_g1 || !delta_g2)
        return ZKG16_ERR_BAD_ARG;
    if (n_a == 0 || n_a != n_b1 || n_a != n_b2 || num_instance == 0 || num_instance > n_a || n_l != n_a - num_instance) return ZKG16_ERR_BAD_ARG;
    if (z_lo > z_hi || z_hi > n_a || h_lo > h_hi || h_hi > n_h) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    auto pk = std::make_unique<PkDev>();
    pk->num_instance = num_instance;
    pk->m_total = n_a;
    pk->n_h_total = n_h;
    pk->z_lo = z_lo; pk->z_hi = z_hi; pk->h_lo = h_lo; pk->h_hi = h_hi;
    pk->blinding = blinding != 0;
    pk->full = z_lo == 0 && z_hi == n_a && h_lo == 0 && h_hi == n_h && pk->blinding;
    const size_t nz = pk->z_hi - pk->z_lo, nh = pk->h_hi - pk->h_lo;
    pk->a.alloc((nz + 3) * sizeof(G1AffineU));
    pk->b1.alloc((nz + 3) * sizeof(G1AffineU));
    pk->l.alloc((nz + 3) * sizeof(G1AffineU));
    pk->b2.alloc((nz + 3) * sizeof(G2AffineU));
    pk->h.alloc((nh ? nh : 1) * sizeof(G1AffineU));
    ZK_HIP(hipMemsetAsync(pk->a.p, 0, pk->a.bytes, ctx->stream));      // (0,0) = infinity everywhere by default
    ZK_HIP(hipMemsetAsync(pk->b1.p, 0, pk->b1.bytes, ctx->stream));
    ZK_HIP(hipMemsetAsync(pk->l.p, 0, pk->l.bytes, ctx->stream));
    ZK_HIP(hipMemsetAsync(pk->b2.p, 0, pk->b2.bytes, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    upload_points<G1Affine>(ctx, pk->a.as<G1AffineU>(), a_query, a_inf, pk->z_lo, pk->z_hi);
    upload_points<G1Affine>(ctx, pk->b1.as<G1AffineU>(), b_g1_query, b_g1_inf, pk->z_lo, pk->z_hi);
    upload_points<G2Affine>(ctx, pk->b2.as<G2AffineU>(), b_g2_query, b_g2_inf, pk->z_lo, pk->z_hi);
    upload_points<G1Affine>(ctx, pk->h.as<G1AffineU>(), h_query, h_inf, pk->h_lo, pk->h_hi);
    // l_query[j] pairs with z[num_instance + j]: place it at the same index as its scalar in this shard's z slice
    {
        const size_t lo = pk->z_lo > num_instance ? pk->z_lo : num_instance, hi = pk->z_hi;
        if (hi > lo)
            upload_points<G1Affine>(ctx, pk->l.as<G1AffineU>() + (lo - pk->z_lo), l_query, l_inf, lo - num_instance, hi - num_instance);
    }
    pk->alpha_g1 = g1_from_abi(alpha_g1, 0);
    pk->beta_g1 = g1_from_abi(beta_g1, 0);
    pk->delta_g1 = g1_from_abi(delta_g1, 0);
    pk->beta_g2 = g2_from_abi(beta_g2, 0);
    pk->delta_g2 = g2_from_abi(delta_g2, 0);
    // extra slots (scalars r, s, -rs):  a += r*delta1 ; b1 += s*delta1 ; b2 += s*delta2 ; l += (-rs)*delta1
    upload_one<G1Affine>(ctx, pk->a.as<G1AffineU>() + nz + 0, pk->delta_g1);
    upload_one<G1Affine>(ctx, pk->b1.as<G1AffineU>() + nz + 1, pk->delta_g1);
    upload_one<G2Affine>(ctx, pk->b2.as<G2AffineU>() + nz + 1, pk->delta_g2);
    upload_one<G1Affine>(ctx, pk->l.as<G1AffineU>() + nz + 2, pk->delta_g1);
    pk->b_mask.alloc(nz + 3);
    pk->b_skipped = b_density_mask_run(ctx, pk->b1.as<G1AffineU>(), pk->b2.as<G2AffineU>(), nz + 3, pk->b_mask.as<uint8_t>());
    *pk_handle = ctx->next_handle++;
    ctx->pks.put(*pk_handle, std::move(pk));
    ZK_API_END(ctx)
}

int zkg16_pk_load(zkg16_ctx *ctx,
                  const uint64_t *a_query, const uint8_t *a_inf, size_t n_a,
                  const uint64_t *b_g1_query, const uint8_t *b_g1_inf, size_t n_b1,
                  const uint64_t *b_g2_query, const uint8_t *b_g2_inf, size_t n_b2,
                  const uint64_t *h_query, const uint8_t *h_inf, size_t n_h,
                  const uint64_t *l_query, const uint8_t *l_inf, size_t n_l,
                  const uint64_t alpha_g1[12], const uint64_t beta_g1[12], const uint64_t beta_g2[24],
                  const uint64_t delta_g1[12], const uint64_t delta_g2[24],
                  size_t num_instance, int shard_index, int shard_count, uint64_t *pk_handle) {
    if (shard_count < 1 || shard_index < 0 || shard_index >= shard_count) return ZKG16_ERR_BAD_ARG;
    return zkg16_pk_load_range(ctx, a_query, a_inf, n_a, b_g1_query, b_g1_inf, n_b1, b_g2_query, b_g2_inf, n_b2, h_query, h_inf, n_h, l_query,
                               l_inf, n_l, alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, num_instance,
                               n_a * (size_t)shard_index / shard_count, n_a * (size_t)(shard_index + 1) / shard_count,
                               n_h * (size_t)shard_index / shard_count, n_h * (size_t)(shard_index + 1) / shard_count, shard_index == 0,
                               pk_handle);
}

// A shard of a key that is already resident (zkg16_setup_resident / an un-sharded zkg16_pk_load): device-to-device copies of
// the index ranges, nothing crosses PCIe.
int zkg16_pk_slice(zkg16_ctx *ctx, uint64_t src_handle, size_t z_lo, size_t z_hi, size_t h_lo, size_t h_hi, int blinding,
                   uint64_t *pk_handle) {
    if (!pk_handle) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    auto src_ref = ctx->pks.get(src_handle); PkDev *src = src_ref.get();
    if (!src) return ZKG16_ERR_BAD_HANDLE;
    if (!src->full) return ZKG16_ERR_BAD_ARG;
    if (z_lo > z_hi || z_hi > src->m_total || h_lo > h_hi || h_hi > src->n_h_total) return ZKG16_ERR_BAD_ARG;
    auto pk = std::make_unique<PkDev>();
    pk->num_instance = src->num_instance;
    pk->m_total = src->m_total;
    pk->n_h_total = src->n_h_total;
    pk->z_lo = z_lo; pk->z_hi = z_hi; pk->h_lo = h_lo; pk->h_hi = h_hi;
    pk->blinding = blinding != 0;
    pk->full = z_lo == 0 && z_hi == src->m_total && h_lo == 0 && h_hi == src->n_h_total && pk->blinding;
    const size_t nz = z_hi - z_lo, nh = h_hi - h_lo, sm = src->m_total;
    pk->a.alloc((nz + 3) * sizeof(G1AffineU));
    pk->b1.alloc((nz + 3) * sizeof(G1AffineU));
    pk->l.alloc((nz + 3) * sizeof(G1AffineU));
    pk->b2.alloc((nz + 3) * sizeof(G2AffineU));
    pk->h.alloc((nh ? nh : 1) * sizeof(G1AffineU));
    auto cp = [&](DevBuf &dst, const DevBuf &from, size_t elem) {
        if (nz) ZK_HIP(hipMemcpyAsync(dst.p, static_cast<const unsigned char *>(from.p) + z_lo * elem, nz * elem, hipMemcpyDeviceToDevice, ctx->stream));
        ZK_HIP(hipMemcpyAsync(static_cast<unsigned char *>(dst.p) + nz * elem, static_cast<const unsigned char *>(from.p) + sm * elem, 3 * elem,
                              hipMemcpyDeviceToDevice, ctx->stream));      // the three trailing delta slots
    };
    cp(pk->a, src->a, sizeof(G1AffineU));
    cp(pk->b1, src->b1, sizeof(G1AffineU));
    cp(pk->l, src->l, sizeof(G1AffineU));
    cp(pk->b2, src->b2, sizeof(G2AffineU));
    if (nh) ZK_HIP(hipMemcpyAsync(pk->h.p, src->h.as<G1AffineU>() + h_lo, nh * sizeof(G1AffineU), hipMemcpyDeviceToDevice, ctx->stream));
    pk->alpha_g1 = src->alpha_g1; pk->beta_g1 = src->beta_g1; pk->delta_g1 = src->delta_g1;
    pk->beta_g2 = src->beta_g2; pk->delta_g2 = src->delta_g2;
    pk->b_mask.alloc(nz + 3);
    pk->b_skipped = b_density_mask_run(ctx, pk->b1.as<G1AffineU>(), pk->b2.as<G2AffineU>(), nz + 3, pk->b_mask.as<uint8_t>());
    *pk_handle = ctx->next_handle++;
    ctx->pks.put(*pk_handle, std::move(pk));
    ZK_API_END(ctx)
}

// Rank roles for one proof over n_ranks GPUs (host-only, no ctx).  Work is counted in G1 mixed additions: a z-side term
// costs W_z * (2 + density * (1 + kappa)) (L, A, and the B1 / B2 terms that are not infinity; kappa = G2 : G1 addition cost),
// an h term W_h, the witness map omega per domain element.  The first k ranks run the witness map and share h_query; every
// rank takes a share of the z ranges proportional to the time it has left, so that all finish together at
//   T(k) = max( (Z + H + k * WM) / n_ranks,  WM + H / k ),
// and k is the one that minimises T (k = n_ranks is the homogeneous split of round 1: every rank repeats the witness map).
static int default_window_bits(size_t n) {
    if (n >= ((size_t)1 << 23)) return 17;
    if (n >= ((size_t)1 << 20)) return 16;
    if (n >= ((size_t)1 << 17)) return 15;
    if (n >= ((size_t)1 << 14)) return 13;
    int lg = 0;
    while (((size_t)2 << lg) <= n) lg++;
    return lg - 3 < 4 ? 4 : lg - 3;
}
int zkg16_shard_plan(int n_ranks, size_t m_total, size_t n_h, double b_density, int h_ranks, const float *z_cost, uint64_t *ranges,
                     uint8_t *blinding, int *h_ranks_out) {
    return zkg16_shard_plan_tables(n_ranks, m_total, n_h, b_density, h_ranks, z_cost, 0, ranges, blinding, h_ranks_out);
}
int zkg16_shard_plan_tables(int n_ranks, size_t m_total, size_t n_h, double b_density, int h_ranks, const float *z_cost, int window_tables,
                            uint64_t *ranges, uint8_t *blinding, int *h_ranks_out) {
    if (n_ranks < 1 || m_total == 0 || !ranges || !blinding || h_ranks < 0 || h_ranks > n_ranks) return ZKG16_ERR_BAD_ARG;
    if (!(b_density > 0.0) || b_density > 1.0) b_density = 0.8;
    // calibrated on one MI355X playing every rank in turn (tools/shard_calibrate.py, profiles/shard_calibration_r2.txt, 128x128):
    // a z-only shard takes 3.6 ms + 108 ms x its fraction of the z cost (97 ms with window tables on the shard), an h-only shard
    // 23.8 ms (the witness map) + 1.2 ms + 46.0 ms x its fraction of h_query (41.2 ms with tables).  In additions at 6.2 G/s:
    // z side 1.33x (1.19x) its additions, h side 1.13x (1.015x), witness map 8.8 per domain element.
    constexpr double KAPPA = 2.8, OMEGA = 8.8;
    const double Z_OVERHEAD = window_tables ? 1.19 : 1.33, H_OVERHEAD = window_tables ? 0.965 : 1.13;
    const int G = n_ranks;
    const double Wz = 254 / default_window_bits(m_total + 3) + 1, Wh = n_h ? 254 / default_window_bits(n_h) + 1 : 0;
    // z-side work: uniform model, or the caller's per-index costs (in G1 mixed additions: entries of the scalar times the
    // queries in which its base is not the point at infinity, the G2 one counted KAPPA times) — the witness of a real circuit
    // is not uniform (runs of 0 / 1 values, variables absent from B), so equal index ranges are not equal work
    double Z = (double)m_total * Wz * (2.0 + b_density * (1.0 + KAPPA));
    if (z_cost) {
        Z = 0;
        for (size_t i = 0; i < m_total; i++) Z += z_cost[i] > 0 ? (double)z_cost[i] : 0.0;
        if (!(Z > 0)) Z = 1.0;
    }
    Z *= Z_OVERHEAD;
    const double H = H_OVERHEAD * (double)n_h * Wh, WM = n_h ? OMEGA * (double)(n_h + 1) : 0.0;
    // fixed cost of taking part at all (latency chains that do not shrink with the share: the scatter passes and the four / one
    // bucket reductions): ~1.9 ms for the z side, ~0.5 ms for the h side, in additions at 6.2 G/s.  With them a rank whose time
    // is used up by the witness map and its h share takes no z work at all, and small circuits use fewer witness-map ranks.
    // (round 3, profiles/shard_calibration_r3.txt: with the bit-sliced reductions a z-only shard takes 1.9 ms + 96.8 ms x its fraction,
    // an h-only one 24.26 ms (the witness map + 0.46 ms) + 39.1 ms x its fraction: round 2's 3.5 / 1.15 ms became 1.9 / 0.46, and the
    // h-side factor with tables 0.965 — with 1.5 ms / 1.015 the model preferred five witness-map ranks of eight, measured 36.0 against 34.5 ms)
    constexpr double F_Z = 11.8e6, F_H = 2.9e6;
    // time of the plan with k witness-map ranks: smallest T with  sum_i max(0, T - busy_i - F_Z) >= Z,  busy_i = WM + F_H + H/k (i < k)
    auto busy_of = [&](int k, int i) { return i < k ? WM + (n_h ? F_H : 0.0) + H / k : 0.0; };
    auto T_of = [&](int k) {
        double lo = busy_of(k, 0), hi = lo + F_Z + Z + 1.0;
        for (int it = 0; it < 80; it++) {
            const double T = 0.5 * (lo + hi);
            double c = 0;
            for (int i = 0; i < G; i++) { const double x = T - busy_of(k, i) - F_Z; if (x > 0) c += x; }
            if (c >= Z) hi = T; else lo = T;
        }
        return hi;
    };
    int k = h_ranks;
    if (k == 0) {
        k = 1;
        for (int c = 2; c <= G; c++)
            if (T_of(c) < T_of(k) * (1.0 - 1e-9)) k = c;
    }
    const double T = T_of(k);
    std::vector<double> cap(G);
    double cap_sum = 0;
    for (int i = 0; i < G; i++) {
        cap[i] = T - busy_of(k, i) - F_Z;
        if (cap[i] < 0) cap[i] = 0;
        cap_sum += cap[i];
    }
    if (!(cap_sum > 0)) { cap.assign(G, 1.0); cap_sum = G; }
    double acc = 0, run = 0;
    size_t prev = 0, pos = 0;
    bool blind_given = false;
    for (int i = 0; i < G; i++) {
        acc += cap[i];
        size_t hi;
        if (i == G - 1) {
            hi = m_total;
        } else if (!z_cost) {
            hi = (size_t)((double)m_total * (acc / cap_sum) + 0.5);
        } else {                                              // advance until this rank's share of the total cost is reached
            const double target = Z * (acc / cap_sum);
            while (pos < m_total && run < target) { run += Z_OVERHEAD * (z_cost[pos] > 0 ? (double)z_cost[pos] : 0.0); pos++; }
            hi = pos;
        }
        if (hi < prev) hi = prev;
        if (hi > m_total) hi = m_total;
        ranges[4 * i + 0] = prev;
        ranges[4 * i + 1] = hi;
        ranges[4 * i + 2] = i < k ? n_h * (size_t)i / k : 0;
        ranges[4 * i + 3] = i < k ? n_h * (size_t)(i + 1) / k : 0;
        blinding[i] = (!blind_given && hi > prev) ? 1 : 0;
        blind_given = blind_given || blinding[i];
        prev = hi;
    }
    if (h_ranks_out) *h_ranks_out = k;
    return ZKG16_OK;
}

// Window tables for a resident key or shard (msm.hip, "window tables").  window_bits_* = 0: chosen from the query length;
// < 0: leave that side as it is.  All four z-side queries share one width (A and L share a sorted term list, so do B1 and B2).
// Width chosen by a cost model in mixed additions: one per (scalar, window) term plus ~7 per bucket (its two additions of
// the reduction, the lost first slot of its run, its share of the fix-ups), over the widths that end on a window boundary
// (15, 14, 13, 12 windows).  Measured (ms per proof, plain key -> table): 32x32 13.4 -> 12.05 at 17 bits (13.15 at 19, 14.0 at
// 20); 46x46 22.6 -> 19.65 at 17 (21.7 at 19); 128x128 181 -> 171.0 at 20 / 22 for z / h (172.0 at 20 / 20, 172.4 at 22 / 22,
// 176.4 at 19 / 22).  Below 17 bits a bucket run spans more than the four lanes the short fix-up path handles (one resident round
// of accumulation waves is 2^17 lanes) and everything goes through the long path: 46x46 at 16 bits 27.8 ms, at 15 bits 40 ms.
// Every query gets a table by default: with the bit-sliced bucket reduction one bucket set of 2^15 / 2^16 buckets is reduced in ~20
// dependent additions, so even small keys gain (profiles/table_sweep_r3.txt: 4x4 3.5 -> 2.45 ms and 8x8 3.9 -> 3.1 ms at 16 bits, 16x16
// 5.6 -> 4.7 ms and the PrimeCircuit 5.9 -> 4.9 ms at 17; narrower tables lose: few buckets, each a long dependent chain).  Round 2
// left queries under 3 * 2^17 terms plain because the reduction of 2^16 buckets then cost a G2 MSM 6 ms.
static int default_table_bits(size_t n) {
    if (n < ((size_t)1 << 16)) return 16;
    int best = 17;
    double best_cost = 0;
    for (int c : {17, 19, 20, 22}) {
        const double cost = (double)n * (254 / c + 1) + 7.0 * (double)((size_t)1 << (c - 1));
        if (c == 17 || cost < best_cost) { best = c; best_cost = cost; }
    }
    return best;
}
int zkg16_pk_precompute(zkg16_ctx *ctx, uint64_t pk_handle, int window_bits_z, int window_bits_h, uint64_t *table_bytes) {
    if (window_bits_z > 24 || window_bits_h > 24 || (window_bits_z > 0 && window_bits_z < 4) || (window_bits_h > 0 && window_bits_h < 4))
        return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    std::unique_lock<std::shared_mutex> keys(ctx->key_rw);      // the key's buffers are replaced: no proof on any lane meanwhile
    auto pk_ref = ctx->pks.get(pk_handle); PkDev *pk = pk_ref.get();
    if (!pk) return ZKG16_ERR_BAD_HANDLE;
    if ((window_bits_z >= 0 && pk->tab_c_z) || (window_bits_h >= 0 && pk->tab_c_h)) return ZKG16_ERR_BAD_ARG;      // already built
    const size_t nz = pk->z_hi - pk->z_lo, nzs = nz + 3, nh = pk->h_hi - pk->h_lo;
    const int cz = window_bits_z < 0 || (nz == 0 && !pk->blinding) ? 0 : window_bits_z ? window_bits_z : default_table_bits(nzs);
    const int ch = window_bits_h < 0 || nh == 0 ? 0 : window_bits_h ? window_bits_h : default_table_bits(nh);
    if ((cz && nzs * (size_t)(254 / cz + 1) >= ((size_t)1 << 31)) || (ch && nh * (size_t)(254 / ch + 1) >= ((size_t)1 << 31))) return ZKG16_ERR_BAD_ARG;
    uint64_t added = 0;
    if (cz) {
        DevBuf a = msm_tables_build_g1(ctx, pk->a, nzs, cz);
        DevBuf l = msm_tables_build_g1(ctx, pk->l, nzs, cz);
        DevBuf b1 = msm_tables_build_g1(ctx, pk->b1, nzs, cz);
        DevBuf b2 = msm_tables_build_g2(ctx, pk->b2, nzs, cz);
        pk->a = std::move(a); pk->l = std::move(l); pk->b1 = std::move(b1); pk->b2 = std::move(b2);      // all four or none
        pk->tab_c_z = cz;
        added += (uint64_t)(254 / cz) * nzs * (3 * sizeof(G1AffineU) + sizeof(G2AffineU));
    }
    if (ch) {
        pk->h = msm_tables_build_g1(ctx, pk->h, nh, ch);
        pk->tab_c_h = ch;
        added += (uint64_t)(254 / ch) * nh * sizeof(G1AffineU);
    }
    if (table_bytes) *table_bytes = added;
    ZK_API_END(ctx)
}

int zkg16_pk_table_bits(zkg16_ctx *ctx, uint64_t pk_handle, int *window_bits_z, int *window_bits_h) {
    ZK_API_BEGIN(ctx)
    auto pk_ref = ctx->pks.get(pk_handle); PkDev *pk = pk_ref.get();
    if (!pk) return ZKG16_ERR_BAD_HANDLE;
    if (window_bits_z) *window_bits_z = pk->tab_c_z;
    if (window_bits_h) *window_bits_h = pk->tab_c_h;
    ZK_API_END(ctx)
}

void zkg16_pk_free(zkg16_ctx *ctx, uint64_t h) {
    if (!ctx) return;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    ctx->pks.erase(h);
}

int zkg16_r1cs_load(zkg16_ctx *ctx,
                    const uint64_t *a_row_ptr, const uint32_t *a_col, const uint64_t *a_coeff,
                    const uint64_t *b_row_ptr, const uint32_t *b_col, const uint64_t *b_coeff,
                    const uint64_t *c_row_ptr, const uint32_t *c_col, const uint64_t *c_coeff,
                    size_t num_instance, size_t num_constraints, size_t num_variables, uint64_t *r1cs_handle) {
    ZK_API_BEGIN(ctx)
    const uint64_t *rp[3] = {a_row_ptr, b_row_ptr, c_row_ptr};
    const uint32_t *col[3] = {a_col, b_col, c_col};
    const uint64_t *cf[3] = {a_coeff, b_coeff, c_coeff};
    int rc = load_r1cs(ctx, rp, col, cf, num_instance, num_constraints, num_variables, r1cs_handle);
    if (rc) return rc;
    ZK_API_END(ctx)
}

// A synthesized circuit (zkg16_circuit_*) loaded straight onto the device: the handles zkg16_r1cs_load / zkg16_witness_load would
// return for zkg16_circuit_export's arrays, without those arrays crossing the ABI.  The arrays are written into pinned staging memory
// the ctx keeps (grown on demand): a caller that exported into fresh buffers per request paid for 65 MB of allocation, page faults
// and unmapping around every PrimeCircuit request — and the unmapping slowed the NEXT synthesis from 18 to 45-60 ms on the GPU box.
int zkg16_circuit_load(zkg16_ctx *ctx, const zkg16_circuit *c, uint64_t *r1cs_handle, uint64_t *witness_handle) {
    if (!c || !r1cs_handle || !witness_handle) return ZKG16_ERR_BAD_ARG;
    size_t ni = 0, nw = 0, nc = 0, nnz[3] = {0, 0, 0};
    if (zkg16_circuit_dims(c, &ni, &nw, &nc, nnz) != ZKG16_OK) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    zkg16_ctx *root = ctx->root ? ctx->root : ctx;
    // layout of the staging block: 3 row-pointer arrays, 3 column arrays, 3 coefficient arrays, the assignment; 64-byte aligned
    size_t off[10], total = 0;
    auto place = [&](int i, size_t bytes) { off[i] = total; total += (bytes + 63) & ~(size_t)63; };
    for (int m = 0; m < 3; m++) place(m, (nc + 1) * sizeof(uint64_t));
    for (int m = 0; m < 3; m++) place(3 + m, (nnz[m] ? nnz[m] : 1) * sizeof(uint32_t));
    for (int m = 0; m < 3; m++) place(6 + m, (nnz[m] ? nnz[m] : 1) * sizeof(Fr));
    place(9, (ni + nw) * sizeof(Fr));
    if (root->circuit_stage_bytes < total) {
        if (root->circuit_stage) (void)hipHostFree(root->circuit_stage);
        root->circuit_stage = nullptr;
        root->circuit_stage_bytes = 0;
        ZK_HIP(hipHostMalloc(&root->circuit_stage, total + total / 8, hipHostMallocDefault));
        root->circuit_stage_bytes = total + total / 8;
    }
    uint8_t *base = static_cast<uint8_t *>(root->circuit_stage);
    uint64_t *rp[3], *cf[3], *z = reinterpret_cast<uint64_t *>(base + off[9]);
    uint32_t *col[3];
    for (int m = 0; m < 3; m++) {
        rp[m] = reinterpret_cast<uint64_t *>(base + off[m]);
        col[m] = reinterpret_cast<uint32_t *>(base + off[3 + m]);
        cf[m] = reinterpret_cast<uint64_t *>(base + off[6 + m]);
    }
    if (zkg16_circuit_export(c, rp, col, cf, z) != ZKG16_OK) return ZKG16_ERR_BAD_ARG;
    const uint64_t *crp[3] = {rp[0], rp[1], rp[2]}, *ccf[3] = {cf[0], cf[1], cf[2]};
    const uint32_t *ccol[3] = {col[0], col[1], col[2]};
    auto w = std::make_unique<WitnessDev>();
    w->n = ni + nw;
    w->z.alloc(w->n * sizeof(Fr));
    ZK_HIP(hipMemcpyAsync(w->z.p, z, w->n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    const int rc = load_r1cs(ctx, crp, ccol, ccf, ni, nc, ni + nw, r1cs_handle);      // synchronises the stream: the staging block is free again
    if (rc) return rc;
    *witness_handle = ctx->next_handle++;
    ctx->wits.put(*witness_handle, std::move(w));
    ZK_API_END(ctx)
}

// The MatrixCircuit's R1CS of size n written on the device (matrix_r1cs.hip): a handle as zkg16_r1cs_load would return for the
// arrays of zkg16_circuit_matrix + zkg16_circuit_export, without synthesising or uploading them.
int zkg16_r1cs_matrix(zkg16_ctx *ctx, size_t n, uint64_t *r1cs_handle) {
    if (!r1cs_handle || n < 2 || n > 1024) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    int st = ZKG16_OK;
    std::shared_ptr<R1csDev> r = matrix_r1cs_on_device(ctx, n, &st);
    if (!r) return st;
    *r1cs_handle = ctx->next_handle++;
    ctx->r1cs.put(*r1cs_handle, std::move(r));
    ZK_API_END(ctx)
}

// The arrays behind an r1cs handle, copied back (tests compare the device-written MatrixCircuit with the host synthesis).  Each
// pointer may be null; sizes as at load (num_constraints + 1 row pointers, nnz columns / coefficients per matrix).
int zkg16_r1cs_read(zkg16_ctx *ctx, uint64_t r1cs_handle, uint64_t *const row_ptr[3], uint32_t *const col[3], uint64_t *const coeff[3],
                    size_t *num_instance, size_t *num_constraints, size_t *num_variables, size_t nnz[3]) {
    ZK_API_BEGIN(ctx)
    auto rc_ref = ctx->r1cs.get(r1cs_handle); R1csDev *rc = rc_ref.get();
    if (!rc) return ZKG16_ERR_BAD_HANDLE;
    if (num_instance) *num_instance = rc->num_instance;
    if (num_constraints) *num_constraints = rc->num_constraints;
    if (num_variables) *num_variables = rc->num_variables;
    for (int m = 0; m < 3; m++) {
        if (nnz) nnz[m] = rc->nnz[m];
        if (row_ptr && row_ptr[m]) ZK_HIP(hipMemcpyAsync(row_ptr[m], rc->rp[m].p, (rc->num_constraints + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        if (col && col[m] && rc->nnz[m]) ZK_HIP(hipMemcpyAsync(col[m], rc->col[m].p, rc->nnz[m] * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        if (coeff && coeff[m] && rc->nnz[m]) ZK_HIP(hipMemcpyAsync(coeff[m], rc->cf[m].p, rc->nnz[m] * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    }
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_API_END(ctx)
}

void zkg16_r1cs_free(zkg16_ctx *ctx, uint64_t h) {
    if (!ctx) return;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    ctx->r1cs.erase(h);
}

int zkg16_witness_load(zkg16_ctx *ctx, const uint64_t *full_assignment, size_t n_assign, uint64_t *witness_handle) {
    if (!full_assignment || !witness_handle || n_assign == 0) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    auto w = std::make_unique<WitnessDev>();
    w->n = n_assign;
    w->z.alloc(n_assign * sizeof(Fr));
    ZK_HIP(hipMemcpyAsync(w->z.p, full_assignment, n_assign * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    *witness_handle = ctx->next_handle++;
    ctx->wits.put(*witness_handle, std::move(w));
    ZK_API_END(ctx)
}

// The assignment behind a witness handle, copied back to the host (tests compare the device-built MatrixCircuit assignment of
// zkg16_witness_matrix with the host builder's byte for byte).
int zkg16_witness_read(zkg16_ctx *ctx, uint64_t witness_handle, uint64_t *out, size_t n_assign) {
    if (!out) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    auto w_ref = ctx->wits.get(witness_handle); WitnessDev *w = w_ref.get();
    if (!w) return ZKG16_ERR_BAD_HANDLE;
    if (w->n != n_assign) return ZKG16_ERR_BAD_ARG;
    ZK_HIP(hipMemcpyAsync(out, w->z.p, n_assign * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_API_END(ctx)
}

void zkg16_witness_free(zkg16_ctx *ctx, uint64_t h) {
    if (!ctx) return;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    ctx->wits.erase(h);
}

int zkg16_prove_partial(zkg16_ctx *ctx, uint64_t pk_handle, uint64_t r1cs_handle, uint64_t witness_handle,
                        const uint64_t r[4], const uint64_t s[4], uint64_t partial_out[72], uint8_t partial_inf[5]) {
    if (!r || !s || !partial_out || !partial_inf) return ZKG16_ERR_BAD_ARG;
    ZK_LANE_BEGIN(ctx)
    auto pk_ref = root->pks.get(pk_handle); PkDev *pk = pk_ref.get();
    auto rc_ref = root->r1cs.get(r1cs_handle); R1csDev *rc = rc_ref.get();
    auto wit_ref = root->wits.get(witness_handle); WitnessDev *wit = wit_ref.get();
    if (!pk || !rc || !wit) return ZKG16_ERR_BAD_HANDLE;
    if (wit->n != rc->num_variables || pk->m_total != rc->num_variables || pk->num_instance != rc->num_instance ||
        pk->n_h_total != ((size_t)1 << rc->log_n) - 1)
        return ZKG16_ERR_BAD_ARG;
    Partials p;
    prove_device(ctx, *pk, *rc, *wit, fr_from_abi(r), fr_from_abi(s), p);
    point_to_abi(xyzz_to_affine(p.h), partial_out, partial_inf);
    point_to_abi(xyzz_to_affine(p.l), partial_out + 12, partial_inf + 1);
    point_to_abi(xyzz_to_affine(p.a), partial_out + 24, partial_inf + 2);
    point_to_abi(xyzz_to_affine(p.b1), partial_out + 36, partial_inf + 3);
    point_to_abi(xyzz_to_affine(p.b2), partial_out + 48, partial_inf + 4);
    ZK_LANE_END(ctx)
}

int zkg16_prove_finish(zkg16_ctx *ctx, uint64_t pk_handle, const uint64_t r[4], const uint64_t s[4],
                       const uint64_t *partials, const uint8_t *partial_inf, int n_ranks, uint64_t proof_out[48], uint8_t inf_out[3]) {
    if (!r || !s || !partials || !partial_inf || n_ranks < 1 || !proof_out || !inf_out) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    auto pk_ref = ctx->pks.get(pk_handle); PkDev *pk = pk_ref.get();
    if (!pk) return ZKG16_ERR_BAD_HANDLE;
    Partials p;
    sum_partials(p, partials, partial_inf, n_ranks);
    prove_tail(*pk, fr_from_abi(r), fr_from_abi(s), p, proof_out, inf_out);
    ZK_API_END(ctx)
}

int zkg16_combine_partials(const uint64_t alpha_g1[12], const uint64_t beta_g1[12], const uint64_t beta_g2[24],
                           const uint64_t r[4], const uint64_t s[4], const uint64_t *partials, const uint8_t *partial_inf,
                           int n_ranks, uint64_t proof_out[48], uint8_t inf_out[3]) {
    if (!alpha_g1 || !beta_g1 || !beta_g2 || !r || !s || !partials || !partial_inf || n_ranks < 1 || !proof_out || !inf_out)
        return ZKG16_ERR_BAD_ARG;
    Partials p;
    sum_partials(p, partials, partial_inf, n_ranks);
    prove_tail_pts(g1_from_abi(alpha_g1, 0), g1_from_abi(beta_g1, 0), g2_from_abi(beta_g2, 0), fr_from_abi(r), fr_from_abi(s), p,
                   proof_out, inf_out);
    return ZKG16_OK;
}

int zkg16_prove_resident(zkg16_ctx *ctx, uint64_t pk_handle, uint64_t r1cs_handle, uint64_t witness_handle,
                         const uint64_t r[4], const uint64_t s[4], uint64_t proof_out[48], uint8_t inf_out[3]) {
    if (!r || !s || !proof_out || !inf_out) return ZKG16_ERR_BAD_ARG;
    ZK_LANE_BEGIN(ctx)
    auto pk_ref = root->pks.get(pk_handle); PkDev *pk = pk_ref.get();
    auto rc_ref = root->r1cs.get(r1cs_handle); R1csDev *rc = rc_ref.get();
    auto wit_ref = root->wits.get(witness_handle); WitnessDev *wit = wit_ref.get();
    if (!pk || !rc || !wit) return ZKG16_ERR_BAD_HANDLE;
    if (!pk->full) return ZKG16_ERR_BAD_ARG;                            // sharded keys go through prove_partial/finish
    if (wit->n != rc->num_variables || pk->m_total != rc->num_variables || pk->num_instance != rc->num_instance ||
        pk->n_h_total != ((size_t)1 << rc->log_n) - 1)
        return ZKG16_ERR_BAD_ARG;
    Partials p;
    const Fr rr = fr_from_abi(r), ss = fr_from_abi(s);
    prove_device(ctx, *pk, *rc, *wit, rr, ss, p);
    const double t0 = now_ms();
    prove_tail(*pk, rr, ss, p, proof_out, inf_out);
    ctx->timings[8] = (float)(now_ms() - t0);
    ctx->timings[9] += ctx->timings[8];
    ZK_LANE_END(ctx)
}

// One MatrixCircuit request on matrices that are already resident: what the reference times as `proving_time`
// (matrix_proof.rs:138-145: Groth16::prove re-synthesises the circuit, then proves) with the per-request part of the synthesis —
// the assignment — produced WHILE the proof runs.  The three native sponges run on three host threads (sequential by
// construction); as soon as a quarter of their permutations is done the device expands those into their S-box values and the four
// z-side MSMs start on the terms that exist (prove_device's rounds); the witness map and the H MSM follow the last part.
int zkg16_prove_matrix(zkg16_ctx *ctx, uint64_t pk_handle, uint64_t r1cs_handle, size_t n, const uint64_t *a, const uint64_t *b,
                       const uint64_t r[4], const uint64_t s[4], uint64_t proof_out[48], uint8_t inf_out[3], uint64_t public_inputs[12],
                       float *timings_ms) {
    if (!r || !s || !proof_out || !inf_out || !a || !b || n < 2 || n > 1024) return ZKG16_ERR_BAD_ARG;
    if (!ctx) return ZKG16_ERR_BAD_ARG;
    const double t_call = now_ms();
    std::unique_ptr<MatrixWitnessStream, void (*)(MatrixWitnessStream *)> ms(nullptr, matrix_stream_free);
    try {
        // the chains start before the ctx is locked: they need neither it nor the device
        ms.reset(matrix_stream_start(n, a, b, ctx->opt_matrix_parts, ctx->opt_matrix_parts != 1));
    } catch (const std::bad_alloc &) {
        return ZKG16_ERR_OOM;
    }
    ZK_LANE_BEGIN(ctx)
    auto pk_ref = root->pks.get(pk_handle); PkDev *pk = pk_ref.get();
    auto rc_ref = root->r1cs.get(r1cs_handle); R1csDev *rc = rc_ref.get();
    if (!pk || !rc) return ZKG16_ERR_BAD_HANDLE;
    if (!pk->full) return ZKG16_ERR_BAD_ARG;
    const size_t total = matrix_stream_total(ms.get());
    if (total != rc->num_variables || pk->m_total != rc->num_variables || pk->num_instance != 4 || rc->num_instance != 4 ||
        pk->n_h_total != ((size_t)1 << rc->log_n) - 1)
        return ZKG16_ERR_BAD_ARG;                   // not the MatrixCircuit of this size
    WitnessDev wit;
    wit.n = total;
    wit.z.alloc(total * sizeof(Fr));
    // a throw below must not leave the stream object's copies and kernels in flight behind its destruction
    struct Drain { zkg16_ctx *c; ~Drain() { (void)hipStreamSynchronize(c->stream); } } drain{ctx};
    matrix_stream_attach(ms.get(), ctx, wit.z.as<Fr>(), 3);
    ZParts zp;
    zp.parts = matrix_stream_parts(ms.get());
    zp.part_of = matrix_stream_part_of(ms.get());
    MatrixWitnessStream *msp = ms.get();
    zp.produce = [ctx, msp](int k) { matrix_stream_produce(msp, ctx, k); };
    Partials p;
    const Fr rr = fr_from_abi(r), ss = fr_from_abi(s);
    prove_device(ctx, *pk, *rc, wit, rr, ss, p, nullptr, &zp);      // without part_of (matrix_parts = 1): the assignment first, then the proof
    const double t0 = now_ms();
    prove_tail(*pk, rr, ss, p, proof_out, inf_out);
    ctx->timings[8] = (float)(now_ms() - t0);
    ctx->timings[9] += ctx->timings[8];
    if (public_inputs) matrix_stream_hashes(ms.get(), public_inputs);
    if (timings_ms) {
        timings_ms[0] = (float)matrix_stream_chain_ms(ms.get());
        timings_ms[1] = (float)zp.parts;
        timings_ms[2] = (float)(now_ms() - t_call);
    }
    ZK_LANE_END(ctx)
}

int zkg16_prove(zkg16_ctx *ctx, uint64_t pk_handle, const uint64_t r[4], const uint64_t s[4],
                const uint64_t *a_row_ptr, const uint32_t *a_col, const uint64_t *a_coeff,
                const uint64_t *b_row_ptr, const uint32_t *b_col, const uint64_t *b_coeff,
                const uint64_t *c_row_ptr, const uint32_t *c_col, const uint64_t *c_coeff,
                size_t num_instance, size_t num_constraints, const uint64_t *full_assignment, size_t n_assign,
                uint64_t proof_out[48], uint8_t inf_out[3]) {
    if (!r || !s || !proof_out || !inf_out || !full_assignment || n_assign == 0) return ZKG16_ERR_BAD_ARG;
    ZK_LANE_BEGIN(ctx)
    auto pk_ref = root->pks.get(pk_handle); PkDev *pk = pk_ref.get();
    if (!pk) return ZKG16_ERR_BAD_HANDLE;
    if (!pk->full) return ZKG16_ERR_BAD_ARG;
    const uint64_t *rp[3] = {a_row_ptr, b_row_ptr, c_row_ptr};
    const uint32_t *col[3] = {a_col, b_col, c_col};
    const uint64_t *cf[3] = {a_coeff, b_coeff, c_coeff};
    std::unique_ptr<R1csDev> rc;
    const int st = r1cs_create(rp, col, cf, num_instance, num_constraints, n_assign, rc);
    if (st) return st;
    if (pk->m_total != n_assign || pk->num_instance != num_instance || pk->n_h_total != ((size_t)1 << rc->log_n) - 1) return ZKG16_ERR_BAD_ARG;
    // the assignment first (the z-side MSMs need only it); the matrices follow inside prove_device, behind the accumulations
    WitnessDev wit;
    wit.n = n_assign;
    wit.z.alloc(n_assign * sizeof(Fr));
    upload_h2d(ctx, wit.z.p, full_assignment, n_assign * sizeof(Fr));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    const std::function<void()> upload = [&]() { r1cs_copy(ctx, *rc, rp, col, cf); };
    Partials p;
    const Fr rr = fr_from_abi(r), ss = fr_from_abi(s);
    prove_device(ctx, *pk, *rc, wit, rr, ss, p, &upload);
    const double t0 = now_ms();
    prove_tail(*pk, rr, ss, p, proof_out, inf_out);
    ctx->timings[8] = (float)(now_ms() - t0);
    ctx->timings[9] += ctx->timings[8];
    ZK_LANE_END(ctx)
}

int zkg16_setup(zkg16_ctx *ctx, uint64_t r1cs_handle, const uint64_t trapdoor[20], const uint64_t g1_gen[12], const uint64_t g2_gen[24],
                uint64_t *a_query, uint8_t *a_inf, uint64_t *b_g1_query, uint8_t *b_g1_inf, uint64_t *b_g2_query, uint8_t *b_g2_inf,
                uint64_t *h_query, uint64_t *l_query, uint8_t *l_inf,
                uint64_t alpha_g1[12], uint64_t beta_g1[12], uint64_t beta_g2[24], uint64_t delta_g1[12], uint64_t delta_g2[24],
                uint64_t gamma_g2[24], uint64_t *gamma_abc_g1) {
    if (!trapdoor || !g1_gen || !g2_gen || !a_query || !b_g1_query || !b_g2_query || !h_query || !l_query || !alpha_g1 || !beta_g1 || !beta_g2 ||
        !delta_g1 || !delta_g2 || !gamma_g2 || !gamma_abc_g1)
        return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    auto rc_ref = ctx->r1cs.get(r1cs_handle); R1csDev *rc = rc_ref.get();
    if (!rc) return ZKG16_ERR_BAD_HANDLE;
    Fr trap[5];
    memcpy(trap, trapdoor, sizeof trap);
    for (int i = 0; i < 5; i++)
        if (trap[i].is_zero()) return ZKG16_ERR_BAD_ARG;
    SetupOut o{a_query, b_g1_query, b_g2_query, h_query, l_query, gamma_abc_g1, a_inf, b_g1_inf, b_g2_inf, l_inf,
               alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, gamma_g2};
    setup_run(ctx, *rc, trap, g1_from_abi(g1_gen, 0), g2_from_abi(g2_gen, 0), o);
    ZK_API_END(ctx)
}

int zkg16_setup_resident(zkg16_ctx *ctx, uint64_t r1cs_handle, const uint64_t trapdoor[20], const uint64_t g1_gen[12], const uint64_t g2_gen[24],
                         uint64_t *pk_handle, uint64_t alpha_g1[12], uint64_t beta_g2[24], uint64_t gamma_g2[24], uint64_t delta_g2[24],
                         uint64_t *gamma_abc_g1) {
    if (!trapdoor || !g1_gen || !g2_gen || !pk_handle || !alpha_g1 || !beta_g2 || !gamma_g2 || !delta_g2 || !gamma_abc_g1) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    auto rc_ref = ctx->r1cs.get(r1cs_handle); R1csDev *rc = rc_ref.get();
    if (!rc) return ZKG16_ERR_BAD_HANDLE;
    Fr trap[5];
    memcpy(trap, trapdoor, sizeof trap);
    for (int i = 0; i < 5; i++)
        if (trap[i].is_zero()) return ZKG16_ERR_BAD_ARG;
    auto pk = std::make_unique<PkDev>();
    uint64_t beta_g1[12], delta_g1[12];
    SetupOut o{nullptr, nullptr, nullptr, nullptr, nullptr, gamma_abc_g1, nullptr, nullptr, nullptr, nullptr,
               alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, gamma_g2};
    setup_run(ctx, *rc, trap, g1_from_abi(g1_gen, 0), g2_from_abi(g2_gen, 0), o, pk.get());
    *pk_handle = ctx->next_handle++;
    ctx->pks.put(*pk_handle, std::move(pk));
    ZK_API_END(ctx)
}

// ------------------------------------------------------------------------------------------------ stages
int zkg16_ntt(zkg16_ctx *ctx, uint64_t *data, size_t log_n, int inverse, int coset) {
    if (!data) return ZKG16_ERR_BAD_ARG;
    if (log_n > 32) return ZKG16_ERR_DOMAIN_TOO_LARGE;
    if (log_n > 28) return ZKG16_ERR_DOMAIN_TOO_LARGE;
    ZK_API_BEGIN(ctx)
    const size_t n = (size_t)1 << log_n;
    DevBuf d(n * sizeof(Fr)), t(n * sizeof(Fr));
    ZK_HIP(hipMemcpyAsync(d.p, data, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    const Fr *res = ntt_run(ctx, d.as<Fr>(), t.as<Fr>(), (int)log_n, inverse != 0, coset != 0);
    ZK_HIP(hipMemcpyAsync(data, res, n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_API_END(ctx)
}

int zkg16_bench_ntt(zkg16_ctx *ctx, size_t log_n, int inverse, int coset, int iters, float *ms_per_iter) {
    if (!ms_per_iter || iters < 1) return ZKG16_ERR_BAD_ARG;
    if (log_n > 28) return ZKG16_ERR_DOMAIN_TOO_LARGE;
    ZK_API_BEGIN(ctx)
    const size_t n = (size_t)1 << log_n;
    DevBuf d(n * sizeof(Fr)), t(n * sizeof(Fr));
    ZK_HIP(hipMemsetAsync(d.p, 0x5a, n * sizeof(Fr), ctx->stream));       // arbitrary (unreduced) limbs: timing only
    (void)ntt_get_tables(ctx, (int)log_n);
    ntt_run(ctx, d.as<Fr>(), t.as<Fr>(), (int)log_n, inverse != 0, coset != 0);
    hipEvent_t e0, e1;
    ZK_HIP(hipEventCreate(&e0));
    ZK_HIP(hipEventCreate(&e1));
    ZK_HIP(hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < iters; i++) {      // ping-pong, as the witness map does
        if (i & 1) ntt_run(ctx, t.as<Fr>(), d.as<Fr>(), (int)log_n, inverse != 0, coset != 0);
        else ntt_run(ctx, d.as<Fr>(), t.as<Fr>(), (int)log_n, inverse != 0, coset != 0);
    }
    ZK_HIP(hipEventRecord(e1, ctx->stream));
    ZK_HIP(hipEventSynchronize(e1));
    float ms = 0;
    ZK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_iter = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ZK_API_END(ctx)
}

// the whole R1CS -> QAP witness map (3 SpMV + 7 NTT + point-wise) alone on the device, repeated: stand-alone time per config
int zkg16_bench_witness_map(zkg16_ctx *ctx, uint64_t r1cs_handle, uint64_t witness_handle, int iters, float *ms_per_iter) {
    if (!ms_per_iter || iters < 1) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    auto rc_ref = ctx->r1cs.get(r1cs_handle); R1csDev *rc = rc_ref.get();
    auto wit_ref = ctx->wits.get(witness_handle); WitnessDev *wit = wit_ref.get();
    if (!rc || !wit) return ZKG16_ERR_BAD_HANDLE;
    if (wit->n != rc->num_variables) return ZKG16_ERR_BAD_ARG;
    Fr *h = nullptr;
    witness_map_run(ctx, *rc, wit->z.as<Fr>(), &h);
    EventSet evs;
    ZK_HIP(hipEventRecord(evs.ev[0], ctx->stream));
    for (int i = 0; i < iters; i++) witness_map_run(ctx, *rc, wit->z.as<Fr>(), &h);
    ZK_HIP(hipEventRecord(evs.ev[1], ctx->stream));
    ZK_HIP(hipEventSynchronize(evs.ev[1]));
    float ms = 0;
    ZK_HIP(hipEventElapsedTime(&ms, evs.ev[0], evs.ev[1]));
    *ms_per_iter = ms / iters;
    ZK_API_END(ctx)
}

}  // extern "C"

namespace {

template <class A, class X>
int msm_host_entry(zkg16_ctx *ctx, const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars, size_t n, int iters,
                   float *ms_per_iter, uint64_t *out_affine, uint8_t *out_inf, bool g2) {
    if ((!bases || !scalars) && n) return ZKG16_ERR_BAD_ARG;
    if (!out_affine) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    DevBuf d_bases((n ? n : 1) * sizeof(typename UOf<A>::T)), d_sc((n ? n : 1) * sizeof(Fr));
    if (n) {
        upload_points<A>(ctx, d_bases.as<typename UOf<A>::T>(), bases, inf, 0, n);
        ZK_HIP(hipMemcpyAsync(d_sc.p, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    }
    X total = X::inf();
    double ms_sum = 0;
    for (int it = 0; it < (iters < 1 ? 1 : iters); it++) {
        const double t0 = now_ms();
        MsmPlan plan;
        msm_plan_build(ctx, ctx->ws_h, d_sc.as<Fr>(), n, plan);
        if constexpr (sizeof(A) == sizeof(G1Affine)) total = msm_g1_exec(ctx, ctx->ws_h, plan, d_bases.as<G1AffineU>(), "msm");
        else total = msm_g2_exec(ctx, ctx->ws_h, plan, d_bases.as<G2AffineU>(), "msm");
        ms_sum += now_ms() - t0;
    }
    if (ms_per_iter) *ms_per_iter = (float)(ms_sum / (iters < 1 ? 1 : iters));
    point_to_abi(xyzz_to_affine(total), out_affine, out_inf);
    (void)g2;
    ZK_API_END(ctx)
}

}  // namespace

extern "C" {

int zkg16_msm_g1(zkg16_ctx *ctx, const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars_canonical, size_t n,
                 uint64_t out_affine[12], uint8_t *out_inf) {
    return msm_host_entry<G1Affine, G1XYZZ>(ctx, bases, inf, scalars_canonical, n, 1, nullptr, out_affine, out_inf, false);
}
int zkg16_msm_g2(zkg16_ctx *ctx, const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars_canonical, size_t n,
                 uint64_t out_affine[24], uint8_t *out_inf) {
    return msm_host_entry<G2Affine, G2XYZZ>(ctx, bases, inf, scalars_canonical, n, 1, nullptr, out_affine, out_inf, true);
}
int zkg16_bench_msm(zkg16_ctx *ctx, int group, const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars_canonical,
                    size_t n, int iters, float *ms_per_iter, uint64_t *out_affine, uint8_t *out_inf) {
    if (group == 1) return msm_host_entry<G1Affine, G1XYZZ>(ctx, bases, inf, scalars_canonical, n, iters, ms_per_iter, out_affine, out_inf, false);
    if (group == 2) return msm_host_entry<G2Affine, G2XYZZ>(ctx, bases, inf, scalars_canonical, n, iters, ms_per_iter, out_affine, out_inf, true);
    return ZKG16_ERR_BAD_ARG;
}

int zkg16_witness_map(zkg16_ctx *ctx, uint64_t r1cs_handle, uint64_t witness_handle, uint64_t *h_out, size_t *log_n_out) {
    if (!h_out) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    auto rc_ref = ctx->r1cs.get(r1cs_handle); R1csDev *rc = rc_ref.get();
    auto wit_ref = ctx->wits.get(witness_handle); WitnessDev *wit = wit_ref.get();
    if (!rc || !wit) return ZKG16_ERR_BAD_HANDLE;
    if (wit->n != rc->num_variables) return ZKG16_ERR_BAD_ARG;
    Fr *h = nullptr;
    witness_map_run(ctx, *rc, wit->z.as<Fr>(), &h);
    const size_t n = (size_t)1 << rc->log_n;
    ZK_HIP(hipMemcpyAsync(h_out, h, n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    if (log_n_out) *log_n_out = (size_t)rc->log_n;
    ZK_API_END(ctx)
}

int zkg16_fixed_base_g1(zkg16_ctx *ctx, const uint64_t base[12], const uint64_t *scalars_canonical, size_t n, uint64_t *out_affine,
                        uint8_t *out_inf) {
    if (!base || (!scalars_canonical && n) || (!out_affine && n)) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    if (n) {
        DevBuf d_sc(n * sizeof(Fr)), d_out(n * sizeof(G1Affine));
        ZK_HIP(hipMemcpyAsync(d_sc.p, scalars_canonical, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        fixed_base_g1_run(ctx, g1_from_abi(base, 0), d_sc.as<Fr>(), n, d_out.as<G1Affine>());
        ZK_HIP(hipMemcpyAsync(out_affine, d_out.p, n * sizeof(G1Affine), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
        if (out_inf) {
            const G1Affine *o = reinterpret_cast<const G1Affine *>(out_affine);
            for (size_t i = 0; i < n; i++) out_inf[i] = o[i].is_inf() ? 1 : 0;
        }
    }
    ZK_API_END(ctx)
}

int zkg16_fixed_base_g2(zkg16_ctx *ctx, const uint64_t base[24], const uint64_t *scalars_canonical, size_t n, uint64_t *out_affine,
                        uint8_t *out_inf) {
    if (!base || (!scalars_canonical && n) || (!out_affine && n)) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    if (n) {
        DevBuf d_sc(n * sizeof(Fr)), d_out(n * sizeof(G2Affine));
        ZK_HIP(hipMemcpyAsync(d_sc.p, scalars_canonical, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        fixed_base_g2_run(ctx, g2_from_abi(base, 0), d_sc.as<Fr>(), n, d_out.as<G2Affine>());
        ZK_HIP(hipMemcpyAsync(out_affine, d_out.p, n * sizeof(G2Affine), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
        if (out_inf) {
            const G2Affine *o = reinterpret_cast<const G2Affine *>(out_affine);
            for (size_t i = 0; i < n; i++) out_inf[i] = o[i].is_inf() ? 1 : 0;
        }
    }
    ZK_API_END(ctx)
}

// ------------------------------------------------------------------------------------------------ instrumentation
int zkg16_last_timings(zkg16_ctx *ctx, float *ms, int cap) {
    if (!ctx || !ms) return 0;
    int last;
    { std::lock_guard<std::mutex> lk(ctx->lane_mu); last = ctx->last_lane; }
    zkg16_ctx *l = lane_of(ctx, last);
    std::lock_guard<std::mutex> lk(l->mu);
    const int n = cap < 22 ? cap : 22;
    for (int i = 0; i < n; i++) ms[i] = l->timings[i];
    return n;
}

// Lengths of the sorted (scalar, window) term lists of the last proof on this ctx = mixed additions per MSM that uses the list:
// [0] the z list (A and L), [1] the B list (B1 and B2; 0 = they used the z list), [2] the h list.  Synchronises the ctx.
int zkg16_last_term_counts(zkg16_ctx *ctx, uint64_t counts[3]) {
    if (!counts || !ctx) return ZKG16_ERR_BAD_ARG;
    int last;
    { std::lock_guard<std::mutex> lk(ctx->lane_mu); last = ctx->last_lane; }
    zkg16_ctx *root = ctx;
    ctx = lane_of(root, last);
    ZK_API_BEGIN(ctx)
    ZK_HIP(hipDeviceSynchronize());
    MsmWorkspace *w[3] = {&ctx->ws_z, &ctx->ws_zb, &ctx->ws_h};
    for (int i = 0; i < 3; i++) {
        uint32_t v = 0;
        if (w[i]->last_tb && w[i]->offsets.p)
            ZK_HIP(hipMemcpy(&v, w[i]->offsets.as<uint32_t>() + w[i]->last_tb, sizeof v, hipMemcpyDeviceToHost));
        counts[i] = v;
    }
    ZK_API_END(ctx)
}

// G1 accumulation waves per SIMD of the last proof's three term lists (z, B, h; 0 = list not built): the occupancy the
// bucket accumulations actually ran at, which is the row of the bare-loop microbenchmark bench.py must compare them with.
int zkg16_last_acc_waves(zkg16_ctx *ctx, int waves[3]) {
    if (!ctx || !waves) return ZKG16_ERR_BAD_ARG;
    int last;
    { std::lock_guard<std::mutex> lk(ctx->lane_mu); last = ctx->last_lane; }
    zkg16_ctx *l = lane_of(ctx, last);
    std::lock_guard<std::mutex> lk(l->mu);
    MsmWorkspace *w[3] = {&l->ws_z, &l->ws_zb, &l->ws_h};
    for (int i = 0; i < 3; i++) waves[i] = w[i]->last_tb ? (int)(w[i]->last_lanes_g1 / ((uint32_t)l->num_cus * 4u * 64u)) : 0;
    return ZKG16_OK;
}

// waves of the accumulation kernels (as launched with the ctx's current options) that fit one SIMD at once: [0] G1, [1] G2
int zkg16_acc_resident_waves(zkg16_ctx *ctx, int waves[2]) {
    if (!waves) return ZKG16_ERR_BAD_ARG;
    ZK_API_BEGIN(ctx)
    waves[0] = msm_acc_resident_waves(ctx, false);
    waves[1] = msm_acc_resident_waves(ctx, true);
    ZK_API_END(ctx)
}

// the lanes (host intervals, steady-clock ms) of the most recent proofs on this ctx: rows of (lane, start, end); returns the
// number of rows written.  Two proofs whose intervals intersect on different lanes ran at the same time.
int zkg16_lane_log(zkg16_ctx *ctx, double *rows, int cap_rows) {
    if (!ctx || !rows || cap_rows < 0) return 0;
    std::lock_guard<std::mutex> lk(ctx->lane_mu);
    const int n = (int)ctx->lane_log.size() < cap_rows ? (int)ctx->lane_log.size() : cap_rows;
    for (int i = 0; i < n; i++) {
        const auto &e = ctx->lane_log[ctx->lane_log.size() - n + i];
        rows[3 * i] = e.lane; rows[3 * i + 1] = e.t0_ms; rows[3 * i + 2] = e.t1_ms;
    }
    return n;
}

}  // extern "C"
namespace {
std::vector<zkg16_ctx *> all_lanes(zkg16_ctx *root) {
    std::vector<zkg16_ctx *> v{root};
    std::lock_guard<std::mutex> lk(root->lane_mu);
    for (auto &l : root->lanes)
        if (l) v.push_back(l.get());
    return v;
}
}  // namespace
extern "C" {

int zkg16_kernel_timing(zkg16_ctx *ctx, int enable) {
    if (!ctx) return ZKG16_ERR_BAD_ARG;
    for (zkg16_ctx *l : all_lanes(ctx)) {
        std::lock_guard<std::mutex> lk(l->mu);
        l->kernel_timing = enable != 0;
        l->kernel_timing_accumulate_only = enable == 2;
    }
    return ZKG16_OK;
}

// summed over the lanes of the ctx
int zkg16_kernel_stats(zkg16_ctx *ctx, const char *kernel_name, uint64_t *launches, double *total_ms, double *units) {
    if (!ctx || !kernel_name) return ZKG16_ERR_BAD_ARG;
    uint64_t n = 0;
    double ms = 0, u = 0;
    for (zkg16_ctx *l : all_lanes(ctx)) {
        std::lock_guard<std::mutex> lk(l->mu);
        (void)hipSetDevice(l->device);
        kernel_timer_resolve(l);
        auto it = l->kstats.find(kernel_name);
        if (it == l->kstats.end()) continue;
        n += it->second.launches;
        ms += it->second.ms;
        u += it->second.units;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (units) *units = u;
    return ZKG16_OK;
}

void zkg16_kernel_stats_reset(zkg16_ctx *ctx) {
    if (!ctx) return;
    for (zkg16_ctx *l : all_lanes(ctx)) {
        std::lock_guard<std::mutex> lk(l->mu);
        (void)hipSetDevice(l->device);
        kernel_timer_resolve(l);
        l->kstats.clear();
    }
}

}  // extern "C"
