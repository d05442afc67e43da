// Shared host-side plumbing for libzkg16: context, device buffers, error handling, kernel timing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/zkg16.h"
#include "ec.cuh"
#include "ff.cuh"

namespace zk {

struct HipError {
    hipError_t err;
    const char *what;
    const char *file;
    int line;
};

#define ZK_HIP(expr)                                                 \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) throw zk::HipError{_e, #expr, __FILE__, __LINE__}; \
    } while (0)

// Device allocations go through a small process-wide cache: the reference's request flow builds and drops a proving key
// (GBs) per request, and hipMalloc / hipFree of buffers that size cost up to seconds on a busy allocator (n = 128: setup
// measured between 0.47 s and 8 s for identical requests before the cache).  dev_release() synchronises the device first,
// exactly as hipFree does, so a cached block is never handed out while a kernel may still touch it.
void *dev_acquire(size_t bytes, size_t *got);      // api.hip; throws HipError
void dev_release(void *p, size_t bytes) noexcept;
void dev_cache_flush() noexcept;

// Helper threads of one host-side job.  A std::thread that goes out of scope while joinable calls std::terminate, and the
// constructor itself can throw (EAGAIN): behind a C ABI neither may take the process down.  run() starts fn on a new thread,
// or — when none can be started — runs it right here; the destructor joins whatever was started, on every exit path.
class ThreadGroup {
    std::vector<std::thread> th_;
  public:
    ThreadGroup() = default;
    ThreadGroup(const ThreadGroup &) = delete;
    ThreadGroup &operator=(const ThreadGroup &) = delete;
    template <class Fn>
    void run(Fn fn) {
        try {
            th_.emplace_back(fn);
        } catch (const std::system_error &) {
            fn();
        }
    }
    void join() {
        for (auto &t : th_)
            if (t.joinable()) t.join();
        th_.clear();
    }
    ~ThreadGroup() { join(); }
};

// RAII device buffer.  Throws HipError on failure (mapped to a status at the ABI edge).
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    explicit DevBuf(size_t n) { alloc(n); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        p = dev_acquire(n, &bytes);
    }
    void ensure(size_t n) { if (n > bytes) alloc(n); }
    void release() { if (p) { dev_release(p, bytes); p = nullptr; bytes = 0; } }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// fixed-base scalar multiplication (setup): window table of the last generator used + scratch, kept across calls
struct FixedBaseCache {
    // window tables of the two most recently used generators (a request uses the standard generator to derive its own
    // random one, then that one for the whole key) + scratch shared by both
    struct Entry { std::vector<uint8_t> key; DevBuf table; uint64_t stamp = 0; int wbits = 0; } e[2];
    uint64_t clock = 0;
    DevBuf sums, pref, win_bases, table_xyzz;
};

// Per-kernel HIP-event timing on the ctx stream (bench.py's roofline leg reads these).
struct KernelStat { uint64_t launches = 0; double ms = 0, units = 0; };
struct PendingEvent;

struct NttTables {          // per domain size, built on device on first use
    int log_n = 0;
    std::mutex mu;          // the tables below that are built on first use (g_u / gi_u, wu): lanes of one ctx share this object
    DevBuf w;               // w[j]  = omega_N^j,            j < N
    DevBuf wu;              // the same in the unsaturated form (36 B per entry); only built for N > 2^22
    DevBuf g;               // g[i]  = 7^i                   (coset fft pre-multiply)
    DevBuf gi;              // gi[i] = 7^-i * N^-1           (coset ifft post-multiply)
    // the same two tables for the unsaturated kernels with the form conversion folded in (built on first use): a load there
    // skips the conversion product, which leaves a factor 2^-5 per pass behind; g_u = 2^10 g (its product is a full
    // conversion), gi_u = 2^(5 * passes) gi puts the missing factors back at the last store
    DevBuf g_u, gi_u;
    int u_passes = 0;
    Fr n_inv;               // N^-1 (plain ifft post-multiply)
    Fr zinv;                // (7^N - 1)^-1 : 1 / Z on the coset (witness map)
};

struct PkDev {              // proving key shard resident in HBM (affine AoS; (0,0) = infinity)
    size_t num_instance = 0, m_total = 0;            // m_total = num_instance + num_witness (full key)
    size_t z_lo = 0, z_hi = 0;                       // index range of the a/b1/b2 (and padded l) queries kept here
    size_t h_lo = 0, h_hi = 0, n_h_total = 0;        // index range of h_query kept here
    DevBuf a, b1, l;                                 // G1AffineU[z_hi - z_lo + 3]  (three trailing slots: r, s, -rs terms)
    DevBuf b2;                                       // G2AffineU[z_hi - z_lo + 3]
    DevBuf h;                                        // G1AffineU[h_hi - h_lo]
    DevBuf b_mask;                                   // u8[z_hi - z_lo + 3]: 1 where the term's base is infinity in both B queries
    size_t b_skipped = 0;                            // number of such terms (B-side plan is separate when this is worth a sort)
    G1Affine alpha_g1, beta_g1, delta_g1;            // host copies for the tail
    G2Affine beta_g2, delta_g2;
    bool blinding = true;                            // this shard's MSMs carry the r*delta, s*delta, -rs*delta terms
    bool full = true;                                // whole key (all ranges + blinding terms): zkg16_prove / _resident
    // window tables (zkg16_pk_precompute): a / b1 / l / b2 hold [254 / tab_c_z + 1][z_hi - z_lo + 3] points, h [254 / tab_c_h + 1][n_h];
    // level 0 is the query itself, so everything that reads the plain query keeps working.  0 = no table.
    int tab_c_z = 0, tab_c_h = 0;
};

struct R1csDev {
    size_t num_instance = 0, num_constraints = 0, num_variables = 0;
    int log_n = 0;
    DevBuf rp[3], col[3], cf[3];
    size_t nnz[3] = {0, 0, 0};
    // coefficient dictionary (poly.hip): R1CS coefficients are a few hundred distinct field elements (215 in the 128x128
    // MatrixCircuit's 86.6 M non-zeros), so the SpMV reads a 16-bit index per non-zero instead of 32 bytes.  Built on the
    // device the second time the witness map runs on this handle (a handle used once never pays for it); dict_state: 0 = not tried, 1 = in use, 2 = too many distinct values
    DevBuf ci[3], dict;
    uint32_t ndict = 0;
    int dict_state = 0;
    int spmv_uses = 0;              // witness maps run on this handle so far (the structures are built on the second)
    // rows ordered by length, longest first (poly.hip): lane t of the SpMV takes row perm[t], so the 64 rows of a wave have about
    // the same number of non-zeros.  Built with the dictionary; null = natural order
    DevBuf perm[3];
    bool perm_ok = false;
    std::mutex lazy_mu;             // the dictionary / row order are built once, by whichever lane gets there first
};

struct WitnessDev { size_t n = 0; DevBuf z; };

// handle -> resident object.  Objects are shared_ptr: a proof in flight on one lane keeps its key / matrices / assignment alive
// when another caller frees the handle meanwhile (the memory goes back when that proof ends).
template <class T>
class HandleMap {
    std::mutex mu_;
    std::map<uint64_t, std::shared_ptr<T>> m_;
  public:
    std::shared_ptr<T> get(uint64_t h) {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = m_.find(h);
        return it == m_.end() ? nullptr : it->second;
    }
    void put(uint64_t h, std::shared_ptr<T> v) {
        std::lock_guard<std::mutex> lk(mu_);
        m_[h] = std::move(v);
    }
    void erase(uint64_t h) {
        std::shared_ptr<T> dying;           // destroyed outside the lock (a DevBuf release synchronises the device)
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto it = m_.find(h);
            if (it == m_.end()) return;
            dying = std::move(it->second);
            m_.erase(it);
        }
    }
    void clear() {
        std::map<uint64_t, std::shared_ptr<T>> dying;
        {
            std::lock_guard<std::mutex> lk(mu_);
            dying.swap(m_);
        }
    }
};

struct MsmSlot {            // one in-flight MSM: written by the accumulate half (main stream), read by the reduce half (aux)
    DevBuf buckets, wsums_dev, seg_head, seg_tail, seg_meta, long_list, long_sums, red_a, red_b, red_c;
    DevBuf bucket_sum;              // an MSM fed in several rounds (streamed assignment): round 0 accumulates here, later rounds into `buckets` and are merged in
    hipStream_t stream = nullptr;   // this MSM's reduction runs here
    void *wsums_host = nullptr;   // pinned
    size_t host_bytes = 0;
    hipEvent_t acc_done = nullptr, red_done = nullptr;
    hipEvent_t acc_start = nullptr, red_start = nullptr;   // with acc_done / red_done: device time of this MSM's two halves (zkg16_last_timings)
    int nwin = 0, c = 0;
    bool active = false;
    bool pending_reduce = false;    // accumulation queued, reduction not yet (msm_*_enqueue_reduce)
    bool last_of_proof = false;     // this MSM's reduction is the tail of the proof (nothing left to overlap it)
    bool fixups_pending = false;    // fix-ups go with the reduction (aux stream) instead of the accumulation (main stream)
    alignas(16) unsigned char acc_args[128] = {0};
    unsigned acc_grid = 0;
    int two_level_k = 0;            // K of the work-efficient reduction when it was used for this MSM (0 = classic)
    float collect_host_ms = 0;      // host part of msm_collect (the Horner over windows / weight bits) of the last MSM on this slot
    int bit_sliced = 0;             // > 0: the bit-sliced reduction ran with this many weight bits (two_level_k = its chunk size)
    void *red_buckets = nullptr;
    size_t red_nb = 0;
};

struct MsmWorkspace {       // grown on demand, reused across proofs
    DevBuf keys, entries, offsets, seg_params, scalars, stage, sort_temp, codes;
    size_t last_tb = 0;         // offsets[last_tb] = length of the term list built last (0: none) — zkg16_last_term_counts
    uint32_t last_lanes_g1 = 0; // lanes of the G1 accumulation grid the last plan on this workspace asked for (zkg16_last_acc_waves)
};

}  // namespace zk

struct zkg16_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string last_error;
    std::map<int, std::unique_ptr<zk::NttTables>> ntt_tables;     // root only (guarded by ntt_mu): lanes share the tables
    std::mutex ntt_mu;
    zk::HandleMap<zk::PkDev> pks;                                 // root only
    zk::HandleMap<zk::R1csDev> r1cs;
    zk::HandleMap<zk::WitnessDev> wits;
    std::atomic<uint64_t> next_handle{1};
    // ---- lanes: a ctx proves up to `opt_lanes` proofs at a time (actix workers call prove concurrently: src/main.rs:37-43).  A
    // lane is a zkg16_ctx of its own — streams, MSM workspaces and slots, witness-map vectors, pinned buffers — whose `root` is
    // the ctx the caller holds; keys, matrices, assignments and NTT tables live in the root and are shared (HBM grows by a
    // lane's workspaces, not by a second copy of the key and its window tables).  Lane 0 is the root itself; further lanes are
    // created when a second caller arrives while the first is still proving.
    zkg16_ctx *root = nullptr;
    std::unique_ptr<zkg16_ctx> lanes[7];                          // root: lanes 1 .. 7, created on first use (fixed slots: readers never see a resize)
    std::mutex lane_mu;
    std::condition_variable lane_cv;
    bool lane_busy[8] = {false, false, false, false, false, false, false, false};
    int opt_lanes = 2;
    int last_lane = 0;                                            // lane of the proof that finished last (zkg16_last_timings)
    std::shared_mutex key_rw;                                     // proofs: shared; in-place changes of a resident key (zkg16_pk_precompute): exclusive
    struct LaneLogEntry { int lane; double t0_ms, t1_ms; };
    std::vector<LaneLogEntry> lane_log;                           // root (under lane_mu): the last proofs' lanes and host intervals
    zk::MsmWorkspace ws_z, ws_h, ws_zb;               // one workspace per scalar vector (z, h, and z masked by the B-query density)
    hipStream_t wm_stream = nullptr;                  // witness map + h-side sort of a proof, concurrent with the z-side MSMs
    zk::MsmSlot slots[5];                             // B2, H, L, A, B1 of one proof
    void *extra_host = nullptr;                       // pinned staging for the r, s, -rs scalars
    void *stage_host[2] = {nullptr, nullptr};         // pinned staging ring of upload_h2d (api.hip)
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    zk::DevBuf poly[4];                               // a, b, c, tmp vectors of the witness map
    float timings[24] = {0};
    bool kernel_timing = false;
    bool kernel_timing_accumulate_only = false;       // zkg16_kernel_timing(ctx, 2): only the bucket-accumulation launches
    std::map<std::string, zk::KernelStat> kstats;
    std::vector<zk::PendingEvent> pending_events;
    int opt_window_bits = 0;
    int opt_min_seg = 0;                              // shortest per-lane run of sorted entries in an accumulation (0 = default)
    int opt_ntt_mode = 1;                             // 1: unsaturated (29-bit limb) butterflies, 0: saturated
    int opt_reduce_mode = 3;                          // 0 classic (log-depth scan over all chunks), 1 work-efficient two-level, 2 = 1 except the proof's last MSM, 3 (default) = 2 from 16-bit windows on and bit-sliced for single bucket sets <= 2^19, 5 = bit-sliced wherever it applies
    int opt_b_filter = 0;                             // B-side term list filtered out of the full one: 0 = with window tables (default), 1 = always, 2 = never (second sort)
    int opt_spmv_dict = 0;                            // 0/1: coefficient dictionary in the SpMV (default); 2: plain kernel
    int opt_wm_first = -1;                            // see zkg16_set_option "wm_first"
    int opt_g1_waves = 0;                             // G1 accumulation waves per SIMD in the one resident round (0 = 2)
    int opt_fixup_aux = 0;                            // 1: fix-up kernels on the MSM's reduction stream
    int opt_window_bits_h = 0;                        // the H MSM's own plan (it is the last one: its reduction is not hidden)
    int opt_reduce_chunk = 0;
    int opt_wm_concurrent = -1;
    int opt_ntt_radix = 1;                            // 1 (default): the last seven butterfly stages by lane exchanges (ds_bpermute / DPP), 2: every stage through the LDS, 4: two stages per LDS trip
    int opt_ntt_xcd = 1;                              // NTT tiles in XCD-aware order (ntt.hip: xcd_tile)
    int opt_acc_debug = 0;                            // timing probes (wrong results): see AccArgs::debug
    int opt_sort_mode = 0;                            // 0: hand-written bucket scatter (bucket_sort.hip), 1: rocPRIM radix sort
    int opt_acc_pipeline = 0;                         // bit 0 / 1: G1 / G2 accumulation gathers the next base behind the last (inlined) product (default: neither)
    void *circuit_stage = nullptr;                    // pinned staging block of zkg16_circuit_load (root ctx; grown on demand)
    size_t circuit_stage_bytes = 0;
    int opt_collect_threads = -1;                     // -1: the z-side MSMs' window sums are combined on their own host threads when the key is plain; 1 always; 0 never
    int opt_fixed_base_bits = 0;                      // setup's fixed-base window width (0 = by batch size; even widths >= 16 are built in two levels)
    int opt_g2_lazy = 1;                              // G2 accumulation: Fq2 products with one reduction per component, operands parked in LDS (ffu.cuh: fq2u_mul_lazy)
    int opt_matrix_parts = 0;                         // zkg16_prove_matrix: gadget slices the assignment arrives in (0 = five growing slices, k = k equal ones, 1 = no overlap)
    int opt_fuse_pointwise = 1;                       // (ab - c)/Z on the load of the seventh transform (0: its own pass)
    int num_cus = 256;
    bool lds_attr_fixup[2] = {false, false}, lds_attr_ntt = false;      // hipFuncSetAttribute(max dynamic LDS) done on this device
    zk::FixedBaseCache fb_g1, fb_g2;
    zk::DevBuf poseidon_dev;                          // Poseidon MDS + round constants, Montgomery form (witness.hip), uploaded on first use
};

namespace zk {

// Brackets one kernel launch with a HIP-event pair on the ctx stream when ctx->kernel_timing is on.
// Recording is asynchronous (no host sync inside the timed region); pairs are resolved when stats are read.
struct PendingEvent { std::string name; double units; hipEvent_t e0, e1; };

struct ScopedKernelTimer {
    zkg16_ctx *ctx;
    const char *name;
    double units;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t stream;
    ScopedKernelTimer(zkg16_ctx *c, const char *n, double u, hipStream_t st = nullptr);
    ~ScopedKernelTimer();
};
void kernel_timer_resolve(zkg16_ctx *ctx);

// ---- entry points implemented per translation unit
// Out of place: returns `dst`, which holds the transform of `src`; `src` is scratch afterwards.  pw: optional fused
// point-wise stage on the input, src[i] <- (src[i] * b[i] - c[i]) * zinv (the witness map's (ab - c)/Z).
struct NttPointwise { const Fr *b, *c; Fr zinv; };
Fr *ntt_run(zkg16_ctx *ctx, Fr *src, Fr *dst, int log_n, bool inverse, bool coset, const NttPointwise *pw = nullptr);
NttTables *ntt_get_tables(zkg16_ctx *ctx, int log_n);

void spmv_run(zkg16_ctx *ctx, R1csDev &m, const Fr *z, Fr *a, Fr *b, Fr *c);
void pointwise_h_run(zkg16_ctx *ctx, Fr *ab_a, const Fr *b, const Fr *c, const Fr &zinv, size_t n);
void fr_from_mont_run(zkg16_ctx *ctx, const Fr *in, Fr *out, size_t n);
void witness_map_run(zkg16_ctx *ctx, R1csDev &m, const Fr *z, Fr **h_out);

// Scalar-vector side of Pippenger (shared by every MSM over the same scalars).
struct MsmPlan {
    size_t n = 0;            // scalars
    int c = 0, nwin = 0;     // window bits, window count (bucket sets: 1 with window tables)
    int nwin_digits = 0;     // digits per scalar = 254 / c + 1 (= nwin without tables)
    bool tabled = false;     // the bases are a window table [nwin_digits][n] (msm_tables_build): every digit lands in ONE bucket set
    size_t nb = 0;           // buckets per window = 2^(c-1)
    size_t total_entries = 0;            // upper bound n * windows (the exact count lives on the device)
    uint32_t lanes_g1 = 0, lanes_g2 = 0; // lanes of one resident round of accumulation waves (2 / 1 waves per SIMD)
};
void msm_plan_build(zkg16_ctx *ctx, MsmWorkspace &ws, const Fr *scalars_canonical, size_t n, MsmPlan &plan, int window_bits = 0);
// The scalar vector where it lives: n_main elements at `main` then n_extra at `extra`; `mont` = arkworks' Montgomery form
// (converted inside the digit kernel); mask[i] != 0 zeroes scalar i (B-query density filter).  All device pointers.
// part / want_part (streamed assignments, witness.hip): when part != nullptr only the scalars i with part[i] == want_part take
// part in this plan (the others count as zero: they may not even be computed yet).
struct ScalarSrc { const Fr *main; size_t n_main; const Fr *extra; size_t n_extra; bool mont; const uint8_t *mask; const uint8_t *part = nullptr; int want_part = 0; };
void msm_plan_build(zkg16_ctx *ctx, MsmWorkspace &ws, const ScalarSrc &src, MsmPlan &plan, int window_bits = 0, bool tabled = false);
// window tables of a resident key: bases[n] -> new buffer [254 / c + 1][n], level w = 2^(c w) * base (msm.hip)
DevBuf msm_tables_build_g1(zkg16_ctx *ctx, const DevBuf &bases, size_t n, int c);
DevBuf msm_tables_build_g2(zkg16_ctx *ctx, const DevBuf &bases, size_t n, int c);
// the term list of `plan_src` without the terms whose scalar index i has mask[i] != 0 (stable compaction; msm.hip)
void msm_plan_filter(zkg16_ctx *ctx, const MsmWorkspace &ws_src, const MsmPlan &plan_src, const uint8_t *mask, MsmWorkspace &ws_dst, MsmPlan &plan_dst);
void msm_sort_keys(zkg16_ctx *ctx, MsmWorkspace &ws, size_t count, unsigned key_bits);   // sort.hip (rocPRIM radix sort; option sort_mode 1)
// bucket_sort.hip: hand-written wave-ballot counting scatter (default); returns the per-window entry counts (device)
const uint32_t *msm_bucket_sort(zkg16_ctx *ctx, MsmWorkspace &ws, const uint32_t *codes, size_t n, int nwin, int c, uint2 *entries);
void radix_sort_hi32(zkg16_ctx *ctx, const uint64_t *in, uint64_t *out, size_t count, unsigned key_bits, DevBuf &temp, const char *timer_name);
// setup.hip: Groth16 key generation from a known trapdoor (discrete logs on device, then fixed-base batches)
struct SetupOut {
    uint64_t *a_query, *b_g1_query, *b_g2_query, *h_query, *l_query, *gamma_abc_g1;
    uint8_t *a_inf, *b_g1_inf, *b_g2_inf, *l_inf;
    uint64_t *alpha_g1, *beta_g1, *beta_g2, *delta_g1, *delta_g2, *gamma_g2;
};
void setup_run(zkg16_ctx *ctx, const R1csDev &m, const Fr trap[5], const G1Affine &g1, const G2Affine &g2, const SetupOut &out, PkDev *resident = nullptr);
void fr_powers_run(zkg16_ctx *ctx, Fr *out, const Fr &base, const Fr &scale, size_t n);
// Bases side: window sums -> host; returns the MSM value (XYZZ) after the host Horner.
// the two halves of an enqueue, for callers that interleave other launches between them (prove_device)
// round: -1 = the whole MSM in one accumulation (default); k >= 0 = round k of an MSM whose terms arrive in several rounds
// (same plan geometry every round): the rounds' bucket arrays are summed and ONE reduction follows (msm_*_enqueue_reduce)
int msm_acc_resident_waves(zkg16_ctx *ctx, bool g2);
void msm_g1_enqueue_acc(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const G1AffineU *bases, MsmSlot &slot, int round = -1);
void msm_g2_enqueue_acc(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const G2AffineU *bases, MsmSlot &slot, int round = -1);
void msm_g1_enqueue_reduce(zkg16_ctx *ctx, MsmSlot &slot);
void msm_g2_enqueue_reduce(zkg16_ctx *ctx, MsmSlot &slot);
void msm_g1_enqueue(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const G1AffineU *bases, MsmSlot &slot);
void msm_g2_enqueue(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const G2AffineU *bases, MsmSlot &slot);
G1XYZZ msm_g1_collect(zkg16_ctx *ctx, MsmSlot &slot);
G2XYZZ msm_g2_collect(zkg16_ctx *ctx, MsmSlot &slot);
G1XYZZ msm_g1_exec(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const G1AffineU *bases, const char *tag);
G2XYZZ msm_g2_exec(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const G2AffineU *bases, const char *tag);
// saturated (arkworks) affine points -> the unsaturated device form, on the ctx stream
void convert_g1_bases(zkg16_ctx *ctx, const G1Affine *in, G1AffineU *out, size_t n);
void convert_g2_bases(zkg16_ctx *ctx, const G2Affine *in, G2AffineU *out, size_t n);

// circuits.hip: the MatrixCircuit's R1CS of size n as a plan (matrix_plan.hpp), and its reference instantiation on the host
struct MatrixPlan;
bool matrix_plan_build(size_t n, MatrixPlan &plan);
void matrix_plan_instantiate_host(const MatrixPlan &plan, uint64_t *const rp[3], uint32_t *const col[3], Fr *const cf[3]);

// matrix_r1cs.hip: that R1CS written by kernels into a new R1csDev (status: a zkg16_status when the result is null)
std::shared_ptr<R1csDev> matrix_r1cs_on_device(zkg16_ctx *ctx, size_t n, int *status);

// witness.hip: the MatrixCircuit's assignment arriving on the device in parts (zkg16_witness_matrix: all at once;
// zkg16_prove_matrix: while the proof is already running).  slices_wanted gadget slices -> parts = slices + 1.
struct MatrixWitnessStream;
MatrixWitnessStream *matrix_stream_start(size_t n, const uint64_t *a, const uint64_t *b, int slices_wanted, bool overlap);   // host chains start here
int matrix_stream_parts(const MatrixWitnessStream *ms);
size_t matrix_stream_total(const MatrixWitnessStream *ms);
const uint8_t *matrix_stream_part_of(const MatrixWitnessStream *ms);       // device, total + n_extra bytes (null with one slice)
void matrix_stream_attach(MatrixWitnessStream *ms, zkg16_ctx *ctx, Fr *z, size_t n_extra);
void matrix_stream_produce(MatrixWitnessStream *ms, zkg16_ctx *ctx, int k);  // blocks for the chains, then queues part k on ctx->stream
double matrix_stream_chain_ms(const MatrixWitnessStream *ms);
void matrix_stream_hashes(const MatrixWitnessStream *ms, uint64_t out[12]);
void matrix_stream_free(MatrixWitnessStream *ms);

// An assignment that becomes valid in parts while its proof is already running (prove_device): produce(k) blocks until part k
// can be made, then queues on ctx->stream whatever writes it; part_of[i] = the part variable i (and the trailing r, s, -rs slots)
// belongs to.
struct ZParts {
    int parts = 1;
    const uint8_t *part_of = nullptr;
    std::function<void(int)> produce;
};

size_t b_density_mask_run(zkg16_ctx *ctx, const G1AffineU *b1, const G2AffineU *b2, size_t n, uint8_t *mask);
// device outputs; either may be null: saturated (arkworks layout, host-bound) and/or unsaturated (device-resident key)
void fixed_base_g1_run(zkg16_ctx *ctx, const G1Affine &base, const Fr *scalars_canonical, size_t n, G1Affine *out_sat, G1AffineU *out_u = nullptr);
void fixed_base_g2_run(zkg16_ctx *ctx, const G2Affine &base, const Fr *scalars_canonical, size_t n, G2Affine *out_sat, G2AffineU *out_u = nullptr);
// several scalar ranges in one pass (<= 8): range k = [start[k], start[k+1]) of the n concatenated scalars -> out_u[k] / out_sat[k]
// sync = false: everything is left queued on ctx->stream (the caller synchronises before touching the outputs)
void fixed_base_g1_multi(zkg16_ctx *ctx, const G1Affine &base, const Fr *scalars_canonical, size_t n, int nseg, const size_t *start,
                         G1AffineU *const *out_u, G1Affine *const *out_sat, bool sync = true);
void fixed_base_g2_multi(zkg16_ctx *ctx, const G2Affine &base, const Fr *scalars_canonical, size_t n, int nseg, const size_t *start,
                         G2AffineU *const *out_u, G2Affine *const *out_sat, bool sync = true);

}  // namespace zk
