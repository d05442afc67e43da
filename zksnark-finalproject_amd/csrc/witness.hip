// MatrixCircuit assignment generated ON THE DEVICE ("host-side circuit synthesis in C++ / on GPU").
//
// The reference's `proving_time` (/root/reference/src/arkworks/backend/matrix_proof.rs:138-145) covers
// `Groth16::prove(&pk, circuit, rng)`, which re-synthesises the circuit: the assignment z = instance || witness of
// MatrixCircuit (matrix_proof_of_work/constraints.rs:101-128) is recomputed on every request.  Its layout (same as
// circuits.hip's zkg16_circuit_matrix_witness, which is what the tests compare against byte for byte):
//
//   [1, hash_a, hash_b, hash_c] | a (n^2) | b (n^2) | sponge(a) gadget | sponge(b) gadget | n^2 zeros (constraints.rs:84)
//   | per (i, j): 0 (the sum's seed, :87) then the n products a_ik b_kj (:91) | sponge(c) gadget
//
// with each sponge gadget = per Poseidon permutation the five products x^2, x^4, x^8, x^16, x^17 of every S-box input that
// is not a constant (hashing_utils.rs:737-802: 8 full + 29 partial rounds, alpha = 17 -> 265 values; the first permutation
// of a hash has a constant capacity lane in round 0 -> 260).  75 % of z are those sponge values.
//
// Split of the work:
//   * A sponge is a sequential chain — permutation p + 1 needs the state permutation p leaves (hasher.rs:17-27) — so the
//     three chains run natively on three host threads over 64-bit limbs (hostff.hpp), recording only the state in front
//     of every permutation (96 B each).  c = a b comes first on the third thread, over the integers: every entry is
//     below n 2^128 < r, so no reduction is involved.
//   * Everything else is data-parallel and happens on the device, written in place into the buffer
//     zkg16_prove_resident reads: the u64 -> Montgomery conversions, the n^3 products, and for each of the 3 ceil(n^2/2)
//     permutations its 265 S-box values from the recorded entering state (one lane per permutation).
// Nothing of z crosses PCIe: the upload is a, b (16 n^2 B) + the entering states (288 B per two matrix entries) instead
// of 32 B per variable (278 MB at 128x128).
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "common.hpp"
#include "hostff.hpp"

using namespace zk;
using zk::h64::Fr64;

namespace {

#include "poseidon_params.inc"

constexpr int P_ROUNDS = POSEIDON_FULL + POSEIDON_PARTIAL, P_HALF = POSEIDON_FULL / 2;
constexpr size_t PERM_WITNESSES = 265, FIRST_PERM_SKIPPED = 5;

typedef unsigned __int128 u128;

#include "poseidon_h64.inc"

Fr64 fr64_from_u64(uint64_t v) {
    Fr64 c = Fr64::zero();
    c.l[0] = v;
    return h64::to_mont(c);
}

// c = a b over the integers (entries < n 2^128 < r), as Montgomery Fr.  bt = b transposed.
void matmul_u64(size_t n, const uint64_t *a, const uint64_t *b, Fr64 *c) {
    std::vector<uint64_t> bt(n * n);
    for (size_t k = 0; k < n; k++)
        for (size_t j = 0; j < n; j++) bt[j * n + k] = b[k * n + j];
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) {
            u128 lo = 0;
            uint64_t hi = 0;
            const uint64_t *ar = a + i * n, *br = bt.data() + j * n;
            for (size_t k = 0; k < n; k++) {
                const u128 p = (u128)ar[k] * br[k];
                lo += p;
                hi += lo < p ? 1 : 0;
            }
            Fr64 v = Fr64::zero();
            v.l[0] = (uint64_t)lo; v.l[1] = (uint64_t)(lo >> 64); v.l[2] = hi;
            c[i * n + j] = h64::to_mont(v);
        }
}

struct MatrixChains {
    size_t n = 0, nn = 0, perms = 0;
    std::vector<Fr64> elems[3];         // a, b, c as Montgomery Fr
    std::vector<Fr64> states[3];        // perms x 3 each
    Fr64 hash[3];
    std::atomic<size_t> done[3];        // permutations of chain h whose entering state has been written
    std::thread th[3];
    bool started[3] = {false, false, false};
    std::chrono::steady_clock::time_point t0;
    double chain_ms = 0;
    const uint64_t *a = nullptr, *b = nullptr;

    void chain(int h) {
        if (h == 2) matmul_u64(n, a, b, elems[2].data());
        else {
            const uint64_t *src = h == 0 ? a : b;
            for (size_t i = 0; i < nn; i++) elems[h][i] = fr64_from_u64(src[i]);
        }
        hash[h] = sponge_chain(elems[h].data(), nn, states[h].data(), &done[h]);
    }
    // the three chains, one host thread each (the third multiplies the matrices first).  inline_last: the third chain runs on the
    // calling thread (the one-shot entry point: nothing to overlap it with); otherwise all three run beside the caller, which
    // feeds the device from their progress (zkg16_prove_matrix).  a, b must stay valid until join().
    void start(size_t n_, const uint64_t *a_, const uint64_t *b_, bool inline_last) {
        t0 = std::chrono::steady_clock::now();
        n = n_; nn = n * n; a = a_; b = b_;
        perms = (nn + POSEIDON_RATE - 1) / POSEIDON_RATE;
        for (int h = 0; h < 3; h++) {
            elems[h].resize(nn);
            states[h].resize(3 * perms);
            done[h].store(0);
        }
        (void)pparams();
        // a thread that cannot be started (EAGAIN) is not fatal: its chain runs on the caller instead (in join at the latest)
        for (int h = 0; h < (inline_last ? 2 : 3); h++) {
            try {
                th[h] = std::thread([this, h]() { chain(h); });
                started[h] = true;
            } catch (const std::system_error &) {
            }
        }
        if (inline_last) { chain(2); ran_inline[2] = true; }
    }
    bool ran_inline[3] = {false, false, false};
    bool joined = false;
    void join() {
        if (joined) return;
        for (int h = 0; h < 3; h++) {
            if (started[h]) { th[h].join(); started[h] = false; ran_inline[h] = true; }
            else if (!ran_inline[h]) { chain(h); ran_inline[h] = true; }
        }
        chain_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        joined = true;
    }
    // blocks until every chain has recorded the entering states of its permutations [0, upto)
    void wait_for(size_t upto) {
        for (int h = 0; h < 3; h++) {
            if (!started[h] && !ran_inline[h]) { chain(h); ran_inline[h] = true; }      // its thread never started
            while (done[h].load(std::memory_order_acquire) < upto) std::this_thread::sleep_for(std::chrono::microseconds(30));
        }
    }
    ~MatrixChains() {
        for (int h = 0; h < 3; h++)
            if (started[h]) th[h].join();
    }
};

#ifndef ZKG16_HOST_ONLY        // (tests/test_host_sanitize.py compiles the host chains alone, without kernels, under ASan)
// ------------------------------------------------------------------------------------------------ device
struct PoseidonDev { Fr mds[3][3], ark[P_ROUNDS][3]; };

__device__ __forceinline__ void st32(Fr *p, const Fr &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
__device__ __forceinline__ Fr ld32(const Fr *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 lo = q[0], hi = q[1];
    Fr v;
    v.l[0] = lo.x; v.l[1] = lo.y; v.l[2] = lo.z; v.l[3] = lo.w; v.l[4] = hi.x; v.l[5] = hi.y; v.l[6] = hi.z; v.l[7] = hi.w;
    return v;
}

// z[0] = 1 | a, b as Montgomery Fr | n^2 zeros | per (i, j): 0, then a_ik b_kj for k < n.  One lane per element.
struct FillArgs {
    const uint64_t *a, *b;
    Fr *z;
    size_t n, nn, off_a, off_mc, off_mm, total;       // total = 2 nn (a, b) + nn (zeros) + nn (n + 1) (sums' seeds + products)
};
__global__ void __launch_bounds__(256) wit_matrix_fill_kernel(FillArgs g) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.total) return;
    if (t == 0) st32(g.z, Fr::one());
    Fr v = Fr::zero();
    Fr *dst;
    if (t < 2 * g.nn) {
        v.l[0] = (uint32_t)(t < g.nn ? g.a[t] : g.b[t - g.nn]);
        v.l[1] = (uint32_t)((t < g.nn ? g.a[t] : g.b[t - g.nn]) >> 32);
        v = fp_to_mont(v);
        dst = g.z + g.off_a + t;
    } else if (t < 3 * g.nn) {
        dst = g.z + g.off_mc + (t - 2 * g.nn);
    } else {
        const size_t e = t - 3 * g.nn, cell = e / (g.n + 1), k1 = e % (g.n + 1);
        dst = g.z + g.off_mm + e;
        if (k1) {
            const size_t i = cell / g.n, j = cell % g.n, k = k1 - 1;
            const uint64_t x = g.a[i * g.n + k], y = g.b[k * g.n + j];
            const uint64_t lo = x * y, hi = __umul64hi(x, y);
            v.l[0] = (uint32_t)lo; v.l[1] = (uint32_t)(lo >> 32); v.l[2] = (uint32_t)hi; v.l[3] = (uint32_t)(hi >> 32);
            v = fp_to_mont(v);       // the product of two F::from(u64) values, taken in Fr (constraints.rs:91): below 2^128 < r
        }
    }
    st32(dst, v);
}

// One lane per Poseidon permutation: from the state in front of it (host chain) the 265 values its S-boxes allocate, in the
// gadget's allocation order (round by round, lane 0..2, x^2, x^4, x^8, x^16, x^17), written where the sponge's segment of z
// puts them.  blockIdx.y = hash (a, b, c).
struct SpongeArgs {
    const Fr *states[3];            // perms x 3 each
    Fr *out[3];                     // first witness of each hash's gadget
    const PoseidonDev *params;
    uint32_t p_lo, p_hi;            // permutations [p_lo, p_hi) of every hash
};
__global__ void __launch_bounds__(64) wit_sponge_kernel(SpongeArgs g) {
    const uint32_t p = g.p_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.p_hi) return;
    const int h = blockIdx.y;
    const PoseidonDev *pp = g.params;
    Fr st[3];
    for (int i = 0; i < 3; i++) st[i] = ld32(g.states[h] + 3 * (size_t)p + i);
    Fr *out = g.out[h] + (p == 0 ? 0 : (size_t)p * PERM_WITNESSES - FIRST_PERM_SKIPPED);
    for (int r = 0; r < P_ROUNDS; r++) {
        const bool full = r < P_HALF || r >= P_HALF + POSEIDON_PARTIAL;
        for (int i = 0; i < 3; i++) st[i] = fp_add(st[i], pp->ark[r][i]);
        for (int i = 0; i < (full ? 3 : 1); i++) {
            const Fr x = st[i];
            const Fr x2 = fp_sqr(x), x4 = fp_sqr(x2), x8 = fp_sqr(x4), x16 = fp_sqr(x8), x17 = fp_mul(x16, x);
            if (!(p == 0 && r == 0 && i == 0)) {        // the capacity lane of a fresh sponge is a constant: no witnesses
                st32(out, x2); st32(out + 1, x4); st32(out + 2, x8); st32(out + 3, x16); st32(out + 4, x17);
                out += 5;
            }
            st[i] = x17;
        }
        Fr nst[3];
        for (int i = 0; i < 3; i++) {
            Fr acc = fp_mul(st[0], pp->mds[i][0]);
            acc = fp_add(acc, fp_mul(st[1], pp->mds[i][1]));
            nst[i] = fp_add(acc, fp_mul(st[2], pp->mds[i][2]));
        }
        for (int i = 0; i < 3; i++) st[i] = nst[i];
    }
}

// part_of[i] = the part of a streamed assignment in which variable i becomes valid: 0 = what needs only a and b (the constant,
// a, b, the zeros, the products; also the three trailing r / s / -rs slots of the z-side scalar vector), 1 + s = the S-box
// values of the permutations of slice s of every sponge, the last part also the three hashes.
struct PartArgs {
    uint8_t *part_of;
    size_t total, n_all, off_ha, off_hb, off_mc, off_hc, hw;
    uint32_t bounds[10];
    int slices;
};
__global__ void __launch_bounds__(256) wit_part_of_kernel(PartArgs g) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n_all) return;
    int part = 0;
    if (i >= 1 && i <= 3) part = g.slices;
    else if (i < g.total) {
        size_t o = g.total;
        if (i >= g.off_ha && i < g.off_ha + g.hw) o = i - g.off_ha;
        else if (i >= g.off_hb && i < g.off_hb + g.hw) o = i - g.off_hb;
        else if (i >= g.off_hc) o = i - g.off_hc;
        if (o != g.total) {
            const uint32_t p = o < PERM_WITNESSES - FIRST_PERM_SKIPPED ? 0u : (uint32_t)((o + FIRST_PERM_SKIPPED) / PERM_WITNESSES);
            int sl = 0;
            while (sl + 1 < g.slices && p >= g.bounds[sl + 1]) sl++;
            part = 1 + sl;
        }
    }
    g.part_of[i] = (uint8_t)part;
}

#endif  // ZKG16_HOST_ONLY
}  // namespace

#ifndef ZKG16_HOST_ONLY
namespace zk {

struct MatrixWitnessLayout {
    size_t n, nn, hw, ni, off_a, off_b, off_ha, off_hb, off_mc, off_mm, off_hc, total;
    explicit MatrixWitnessLayout(size_t n_) {
        n = n_; nn = n * n; ni = 4;
        hw = (nn + POSEIDON_RATE - 1) / POSEIDON_RATE * PERM_WITNESSES - FIRST_PERM_SKIPPED;
        off_a = ni; off_b = off_a + nn; off_ha = off_b + nn; off_hb = off_ha + hw; off_mc = off_hb + hw; off_mm = off_mc + nn;
        off_hc = off_mm + nn * (n + 1); total = off_hc + hw;
    }
};

// The assignment of one MatrixCircuit request arriving on the device in parts: part 0 needs only a and b, part 1 + s the
// entering states of slice s of the three host chains (common.hpp: MatrixWitnessStream is opaque to api.hip).
struct MatrixWitnessStream {
    MatrixWitnessLayout L;
    MatrixChains mc;
    int slices = 1;
    bool overlap = false;                   // the proof runs while the parts arrive: part_of is built
    std::vector<uint32_t> bounds;           // slices + 1 permutation indices
    DevBuf d_ab, d_states, part_of;
    Fr *z = nullptr;
    const uint64_t *a = nullptr, *b = nullptr;
    Fr inst[3];
    explicit MatrixWitnessStream(size_t n) : L(n) {}
};

MatrixWitnessStream *matrix_stream_start(size_t n, const uint64_t *a, const uint64_t *b, int slices_wanted, bool overlap) {
    auto *ms = new MatrixWitnessStream(n);
    ms->a = a; ms->b = b;
    const size_t perms = (ms->L.nn + POSEIDON_RATE - 1) / POSEIDON_RATE;
    // slices_wanted = 0: growing slices.  The device needs longer for a slice's terms than the host chain for its states
    // (128x128: 84 ms of z-side accumulation against 61 ms of chain), so once started it never waits again — what counts is
    // starting early: a first slice of 6 %, each later one about as long as the device is busy with the one before.
    static const double grow[5] = {0.06, 0.20, 0.45, 0.80, 1.0};
    int k = slices_wanted == 0 ? 5 : slices_wanted < 1 ? 1 : slices_wanted > 8 ? 8 : slices_wanted;
    const bool growing = slices_wanted == 0 && perms >= 4096;
    if (!growing && slices_wanted == 0) k = 4;
    while (k > 1 && !growing && perms / k < 512) k--;          // a slice shorter than ~4 ms of host chain is all fixed cost on the device side
    ms->slices = overlap ? k : 1;
    ms->overlap = overlap;
    ms->bounds.resize(ms->slices + 1);
    for (int i = 0; i <= ms->slices; i++)
        ms->bounds[i] = (growing && overlap) ? (i == 0 ? 0u : (uint32_t)((double)perms * grow[i - 1] + 0.5)) : (uint32_t)(perms * (size_t)i / ms->slices);
    ms->bounds[ms->slices] = (uint32_t)perms;
    try {
        ms->mc.start(n, a, b, !overlap);
    } catch (...) {
        delete ms;
        throw;
    }
    return ms;
}
int matrix_stream_parts(const MatrixWitnessStream *ms) { return ms->slices + 1; }
size_t matrix_stream_total(const MatrixWitnessStream *ms) { return ms->L.total; }
const uint8_t *matrix_stream_part_of(const MatrixWitnessStream *ms) { return ms->overlap ? ms->part_of.as<uint8_t>() : nullptr; }
double matrix_stream_chain_ms(const MatrixWitnessStream *ms) { return ms->mc.chain_ms; }
void matrix_stream_hashes(const MatrixWitnessStream *ms, uint64_t out[12]) {
    for (int h = 0; h < 3; h++) memcpy(out + 4 * h, ms->mc.hash[h].l, 32);
}
void matrix_stream_free(MatrixWitnessStream *ms) { delete ms; }

// device buffers of the request; z: total (+ whatever the caller appends) elements.  n_extra: trailing slots of the z-side scalar
// vector (r, s, -rs) that part_of must cover too.
void matrix_stream_attach(MatrixWitnessStream *ms, zkg16_ctx *ctx, Fr *z, size_t n_extra) {
    const MatrixWitnessLayout &L = ms->L;
    if (!ctx->poseidon_dev.p) {
        const PoseidonH &ph = pparams();
        PoseidonDev pd;
        static_assert(sizeof(PoseidonDev) == sizeof(Fr64) * (9 + 3 * P_ROUNDS), "layout");
        memcpy(pd.mds, ph.mds, sizeof pd.mds);
        memcpy(pd.ark, ph.ark, sizeof pd.ark);
        ctx->poseidon_dev.alloc(sizeof pd);
        ZK_HIP(hipMemcpy(ctx->poseidon_dev.p, &pd, sizeof pd, hipMemcpyHostToDevice));
    }
    ms->z = z;
    ms->d_ab.alloc(2 * L.nn * sizeof(uint64_t));
    ms->d_states.alloc(3 * ms->mc.perms * 3 * sizeof(Fr));
    if (ms->overlap) {
        const size_t n_all = L.total + n_extra;
        ms->part_of.alloc(n_all);
        PartArgs g;
        g.part_of = ms->part_of.as<uint8_t>();
        g.total = L.total; g.n_all = n_all; g.off_ha = L.off_ha; g.off_hb = L.off_hb; g.off_mc = L.off_mc; g.off_hc = L.off_hc; g.hw = L.hw;
        g.slices = ms->slices;
        for (int i = 0; i <= ms->slices; i++) g.bounds[i] = ms->bounds[i];
        hipLaunchKernelGGL(wit_part_of_kernel, dim3((unsigned)((n_all + 255) / 256)), dim3(256), 0, ctx->stream, g);
        ZK_HIP(hipGetLastError());
    }
}

// Queues on ctx->stream what makes part k of z valid; blocks on the host first until the chains have got that far.
void matrix_stream_produce(MatrixWitnessStream *ms, zkg16_ctx *ctx, int k) {
    const MatrixWitnessLayout &L = ms->L;
    Fr *z = ms->z;
    if (k == 0) {
        ZK_HIP(hipMemcpyAsync(ms->d_ab.p, ms->a, L.nn * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipMemcpyAsync(ms->d_ab.as<uint64_t>() + L.nn, ms->b, L.nn * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        FillArgs g;
        g.a = ms->d_ab.as<uint64_t>(); g.b = g.a + L.nn; g.z = z; g.n = L.n; g.nn = L.nn;
        g.off_a = L.off_a; g.off_mc = L.off_mc; g.off_mm = L.off_mm; g.total = 3 * L.nn + L.nn * (L.n + 1);
        ScopedKernelTimer kt(ctx, "wit_matrix_fill_kernel", (double)g.total);
        hipLaunchKernelGGL(wit_matrix_fill_kernel, dim3((unsigned)((g.total + 255) / 256)), dim3(256), 0, ctx->stream, g);
        ZK_HIP(hipGetLastError());
        return;
    }
    const uint32_t p_lo = ms->bounds[k - 1], p_hi = ms->bounds[k];
    const size_t perms = ms->mc.perms;
    if (k == ms->slices) ms->mc.join();          // the hashes exist once the chains have ended
    else ms->mc.wait_for(p_hi);
    for (int h = 0; h < 3; h++)
        ZK_HIP(hipMemcpyAsync(ms->d_states.as<Fr>() + (size_t)h * 3 * perms + 3 * (size_t)p_lo, ms->mc.states[h].data() + 3 * (size_t)p_lo,
                              3 * (size_t)(p_hi - p_lo) * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    if (k == ms->slices) {
        for (int h = 0; h < 3; h++) memcpy(ms->inst[h].l, ms->mc.hash[h].l, 32);
        ZK_HIP(hipMemcpyAsync(z + 1, ms->inst, sizeof ms->inst, hipMemcpyHostToDevice, ctx->stream));
    }
    SpongeArgs g;
    for (int h = 0; h < 3; h++) g.states[h] = ms->d_states.as<Fr>() + (size_t)h * 3 * perms;
    g.out[0] = z + L.off_ha; g.out[1] = z + L.off_hb; g.out[2] = z + L.off_hc;
    g.params = ctx->poseidon_dev.as<PoseidonDev>();
    g.p_lo = p_lo; g.p_hi = p_hi;
    ScopedKernelTimer kt(ctx, "wit_sponge_kernel", 3.0 * (double)(p_hi - p_lo));
    hipLaunchKernelGGL(wit_sponge_kernel, dim3((unsigned)((p_hi - p_lo + 63) / 64), 3), dim3(64), 0, ctx->stream, g);
    ZK_HIP(hipGetLastError());
}

}  // namespace zk
#endif  // ZKG16_HOST_ONLY

extern "C" {

// Host-only half (no ctx, no GPU): the three native sponges of the matrix handler (matrix_proof.rs:110-115) with the state in
// front of every permutation.  states (nullable): 3 hashes x ceil(n^2 / 2) permutations x 3 Fr; hashes: hash_a, hash_b, hash_c.
int zkg16_matrix_sponge_states(size_t n, const uint64_t *a, const uint64_t *b, uint64_t *states, uint64_t hashes[12]) {
    if (!a || !b || !hashes || n < 2 || n > 1024) return ZKG16_ERR_BAD_ARG;
    try {
        MatrixChains mc;
        mc.start(n, a, b, true);
        mc.join();
        for (int h = 0; h < 3; h++) {
            memcpy(hashes + 4 * h, mc.hash[h].l, 32);
            if (states) memcpy(states + (size_t)h * mc.perms * 12, mc.states[h].data(), mc.perms * 96);
        }
    } catch (const std::bad_alloc &) {
        return ZKG16_ERR_OOM;
    }
    return ZKG16_OK;
}

#ifndef ZKG16_HOST_ONLY
// The MatrixCircuit's full assignment for (a, b), built on the device: a witness handle as zkg16_witness_load would return for
// zkg16_circuit_matrix_witness's output.  public_inputs (nullable): hash_a, hash_b, hash_c (Montgomery), the handler's
// public inputs.  timings_ms (nullable, 3): host chains, upload + kernels (device time), whole call.
int zkg16_witness_matrix(zkg16_ctx *ctx, size_t n, const uint64_t *a, const uint64_t *b, uint64_t *witness_handle, uint64_t public_inputs[12],
                         float *timings_ms) {
    if (!a || !b || !witness_handle || n < 2 || n > 1024) return ZKG16_ERR_BAD_ARG;
    if (!ctx) return ZKG16_ERR_BAD_ARG;
    const auto t_call = std::chrono::steady_clock::now();
    if (MatrixWitnessLayout(n).total >= ((size_t)1 << 32)) return ZKG16_ERR_DOMAIN_TOO_LARGE;
    // the chains need neither the ctx nor the device: they run before the ctx is locked, so that other callers of this ctx are
    // not held up by the host arithmetic
    std::unique_ptr<MatrixWitnessStream, void (*)(MatrixWitnessStream *)> ms(nullptr, matrix_stream_free);
    try {
        ms.reset(matrix_stream_start(n, a, b, 1, false));
        ms->mc.join();
    } catch (const std::bad_alloc &) {
        return ZKG16_ERR_OOM;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    try {
        ZK_HIP(hipSetDevice(ctx->device));
        auto w = std::make_unique<WitnessDev>();
        w->n = ms->L.total;
        w->z.alloc(ms->L.total * sizeof(Fr));
        hipEvent_t e0, e1;
        ZK_HIP(hipEventCreate(&e0));
        ZK_HIP(hipEventCreate(&e1));
        struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{e0, e1};
        ZK_HIP(hipEventRecord(e0, ctx->stream));
        const bool trace = getenv("ZKG16_TRACE_HOST") != nullptr;
        auto lap = [&](const char *what) {
            if (trace) fprintf(stderr, "witness_matrix: %s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
        };
        lap("locked, z allocated");
        matrix_stream_attach(ms.get(), ctx, w->z.as<Fr>(), 0);
        lap("attached");
        matrix_stream_produce(ms.get(), ctx, 0);
        lap("part 0 queued");
        matrix_stream_produce(ms.get(), ctx, 1);
        lap("part 1 queued");
        ZK_HIP(hipEventRecord(e1, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));      // the copies above read the stream object's host buffers
        float dev_ms = 0;
        ZK_HIP(hipEventElapsedTime(&dev_ms, e0, e1));
        if (public_inputs) matrix_stream_hashes(ms.get(), public_inputs);
        *witness_handle = ctx->next_handle++;
        ctx->wits.put(*witness_handle, std::move(w));
        if (timings_ms) {
            timings_ms[0] = (float)ms->mc.chain_ms;
            timings_ms[1] = dev_ms;
            timings_ms[2] = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
        }
    } catch (const HipError &e) {
        (void)hipStreamSynchronize(ctx->stream);
        char buf[512];
        snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", e.what, hipGetErrorString(e.err), e.file, e.line);
        ctx->last_error = buf;
        (void)hipGetLastError();
        return e.err == hipErrorOutOfMemory ? ZKG16_ERR_OOM : ZKG16_ERR_HIP;
    } catch (const std::bad_alloc &) {
        (void)hipStreamSynchronize(ctx->stream);
        return ZKG16_ERR_OOM;
    }
    return ZKG16_OK;
}
#endif  // ZKG16_HOST_ONLY

}  // extern "C"
