// Pippenger multi-scalar multiplication on gfx950 for the five MSMs of a Groth16 proof
// (ark-ec 0.4.2 `VariableBaseMSM::msm_bigint` as called by ark-groth16 prover.rs for h_query, l_query,
// a_query, b_g1_query (G1) and b_g2_query (G2); reached from
// /root/reference/src/arkworks/backend/matrix_proof.rs:139-140).  The result is a group element, so only its
// value is pinned by the reference; the schedule below is MI355X-first:
//
//  scalar side (once per scalar vector, shared by every MSM over it — A, B1, B2 and L all use z):
//    msm_digits   signed radix-2^c digits (c up to 16) -> one 64-bit key (bucket id | index | sign) per (window, scalar)
//    radix sort   rocPRIM device radix sort on the bucket-id bits (sort.hip) — stable, so bucket order is reproducible
//    msm_offsets  bucket boundaries by binary search in the sorted keys
//  base side (per MSM):
//    msm_accumulate  one lane per fixed-length SEGMENT of the bucket-sorted entry list (not per bucket), so
//                    every lane of every wave performs exactly the same number of XYZZ mixed additions no matter
//                    how skewed the scalars are (the reference's witnesses are ~10% ones: SURVEY.md 8d);
//                    buckets fully inside a segment are written once, the <=2 straddling partials per lane
//                    go to a side list that msm_fixup folds in.
//    msm_reduce_level  weighted bucket sum  sum_b (b+1) B_b  as a log_K-depth tree: every level combines K
//                    adjacent blocks (S, T) -> (sum S, sum T + |block| * sum j*S_j) with running sums.
//    host: Horner over the W window sums (W*c doublings) — O(1), done in the proof tail.
//
// Arithmetic: 381-bit Montgomery on v_mad_u64_u32 — integer-ALU bound, not HBM bound: one mixed add is ~10 Fq
// products (~3k VALU ops) per 96-B base gathered.  No MFMA (no dense contraction); LDS is used by the scan.
#include <chrono>

#include "common.hpp"

namespace zk {

template <class F> struct FieldTraits;
template <> struct FieldTraits<FqU> { using Sat = Fq; static constexpr bool g2 = false; };
template <> struct FieldTraits<Fq2U> { using Sat = Fq2; static constexpr bool g2 = true; };
template <> struct FieldTraits<Fq> { using Sat = Fq; static constexpr bool g2 = false; };
template <> struct FieldTraits<Fq2> { using Sat = Fq2; static constexpr bool g2 = true; };

// ------------------------------------------------------------------------------------------------ vector ld/st
template <class T>
__device__ __forceinline__ T ldv(const T *p) {
    static_assert(sizeof(T) % 16 == 0, "16-byte multiple");
    T v;
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 *d = reinterpret_cast<uint4 *>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = q[i];
    return v;
}
template <class T>
__device__ __forceinline__ void stv(T *p, const T &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    const uint4 *s = reinterpret_cast<const uint4 *>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) q[i] = s[i];
}

// ------------------------------------------------------------------------------------------------ scalar side
// keys[w*n + i] = bucket id << 32 | i << 1 | neg, bucket id = w * 2^(c-1) + |d| - 1, or `invalid_bucket` (sorts last) for digit 0
// The scalar vector is read where it lives: n_main elements at `scalars` followed by n_extra at `extra` (the r, s, -rs
// terms of a proof), in arkworks' Montgomery form when `mont` (the conversion `into_bigint()` of prover.rs is done here, in
// registers — round 1 ran a separate fr_from_mont pass plus a staging copy per scalar vector), zeroed where mask[i] != 0
// (terms whose base is infinity in both B queries: b_density_mask_kernel).
struct DigitSrc {
    const uint32_t *scalars, *extra;
    size_t n_main, n;
    const uint8_t *mask;
    int mont;
    const uint8_t *part;       // streamed assignments: only scalars with part[i] == want take part (the others may not exist yet)
    int want;
};
__global__ void __launch_bounds__(256) msm_digits_kernel(DigitSrc src, int c, int nwin, size_t nb, uint64_t *keys, uint32_t *codes, uint32_t invalid_bucket) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = src.n;
    if (i >= n) return;
    uint32_t s[9];
    {
        const uint4 *q = reinterpret_cast<const uint4 *>(i < src.n_main ? src.scalars + 8 * i : src.extra + 8 * (i - src.n_main));
        uint4 lo = q[0], hi = q[1];
        if (src.mask && src.mask[i]) lo = hi = make_uint4(0, 0, 0, 0);
        if (src.part && (int)src.part[i] != src.want) lo = hi = make_uint4(0, 0, 0, 0);
        Fr v;
        v.l[0] = lo.x; v.l[1] = lo.y; v.l[2] = lo.z; v.l[3] = lo.w; v.l[4] = hi.x; v.l[5] = hi.y; v.l[6] = hi.z; v.l[7] = hi.w;
        if (src.mont) v = fp_from_mont(v);
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = v.l[k];
        s[8] = 0;
    }
    // scalars above (r-1)/2 are replaced by r - s with the sign of every digit flipped: the magnitude then fits 254
    // bits, so the top window never carries out and no carry-only window (one giant bucket) exists.
    constexpr uint32_t RH[8] = {0x80000000u, 0x7fffffffu, 0x7fff2dffu, 0xa9ded201u, 0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u};  // (r-1)/2
    bool flip = false;
#pragma unroll
    for (int k = 7; k >= 0; k--) {
        if (s[k] != RH[k]) { flip = s[k] > RH[k]; break; }
    }
    if (flip) {
        uint32_t borrow = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint64_t t = (uint64_t)FrP::mod(k) - s[k] - borrow;
            s[k] = (uint32_t)t;
            borrow = (uint32_t)(t >> 63);
        }
    }
    const uint32_t flipbit = flip ? 1u : 0u;
    uint32_t carry = 0;
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    for (int w = 0; w < nwin; w++) {
        const int bit = w * c;
        uint32_t v = 0;
        if (bit < 256) {
            const int limb = bit >> 5, off = bit & 31;
            uint64_t two = (uint64_t)s[limb] | ((uint64_t)s[limb + 1] << 32);
            v = (uint32_t)(two >> off) & mask;
        }
        v += carry;
        uint32_t key = 0;
        carry = 0;
        if (v > half) {                       // recenter: d = v - 2^c <= 0, carry into the next window
            const uint32_t mag = (1u << c) - v;
            carry = 1;
            if (mag) key = 1u + (((mag - 1u) << 1) | (1u ^ flipbit));
        } else if (v != 0) {
            key = 1u + (((v - 1u) << 1) | flipbit);
        }
        if (codes) {      // hand-written bucket scatter (bucket_sort.hip): (|d| - 1) << 1 | negate per (window, scalar), ~0 for digit 0
            codes[(size_t)w * n + i] = key - 1u;
        } else {          // rocPRIM path (sort.hip): 64-bit keys
            const uint32_t g = key ? (uint32_t)((size_t)w * nb + ((key - 1u) >> 1)) : invalid_bucket;
            keys[(size_t)w * n + i] = ((uint64_t)g << 32) | (uint64_t)(((uint32_t)i << 1) | ((key - 1u) & 1u));
        }
    }
}

// offsets[g] = first position of the sorted entry list whose bucket id is >= g  (g = 0 .. total_buckets; the last one is the
// number of valid entries: digit-0 keys carry bucket id = total_buckets and sort behind everything)
// count: length of the sorted list; with the hand-written scatter it is the sum of the per-window totals on the device
__global__ void __launch_bounds__(256) msm_offsets_kernel(const uint2 *sorted, size_t count, const uint32_t *win_total, int nwin, uint32_t *offsets,
                                                          size_t total_buckets) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > total_buckets) return;
    if (win_total) {
        count = 0;
        for (int w = 0; w < nwin; w++) count += win_total[w];
    }
    size_t lo = 0, hi = count;                    // answer in [lo, hi]
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if (sorted[mid].y < (uint32_t)g) lo = mid + 1; else hi = mid;
    }
    offsets[g] = (uint32_t)lo;
}

// Entries per lane so that ONE full round of resident waves covers the whole list with equal work per lane (no tail, no
// partially filled second round): seg_len = ceil(entries / lanes), one value per occupancy class
// (params[0]: G1 kernels, 2 waves/SIMD; params[1]: G2 kernels, 1 wave/SIMD).
// Small MSMs do not fill the round: a run is then kept at least about half an average bucket long (min_seg = 0), so that a
// bucket is split over at most ~3 lanes and its partial sums stay on msm_fixup's short path — with 4-entry runs a
// 6,476-constraint proof took 17 ms instead of 4.7 ms because every bucket went through the long fix-up.
__global__ void msm_seg_params_kernel(const uint32_t *total_ptr, uint32_t total_buckets, uint32_t lanes_g1, uint32_t lanes_g2, uint32_t min_seg,
                                      uint32_t *params) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const uint32_t total = *total_ptr;
        if (min_seg == 0) min_seg = total / total_buckets / 2 + 4;
        uint32_t s1 = (total + lanes_g1 - 1) / lanes_g1, s2 = (total + lanes_g2 - 1) / lanes_g2;
        params[0] = s1 < min_seg ? min_seg : s1;
        params[1] = s2 < min_seg ? min_seg : s2;
    }
}

// ------------------------------------------------------------------------------------------------ base side
template <class F>
struct AccArgs {
    const Affine<F> *bases;
    const uint2 *entries;        // .x = base index << 1 | negate, .y = global bucket id (window * 2^(c-1) + bucket)
    const uint32_t *offsets;
    XYZZ<F> *buckets, *seg_head, *seg_tail;
    int32_t *seg_meta;          // [2*t] = bucket of head partial or -1, [2*t+1] = bucket of tail partial or -1
    const uint32_t *total_ptr;  // number of sorted entries (= offsets[total_buckets]); read on the device, no host sync
    size_t total_buckets;
    const uint32_t *seg_len_ptr; // entries per lane, computed on the device from the exact entry count (msm_seg_params_kernel)
    uint32_t debug;              // timing probes only (option "acc_debug"; results are WRONG): bit 0 = no bucket stores, bit 1 = always gather base 0, bit 2 = gathers from the first 64 K bases only
};

// G1: 2 waves per SIMD (<= 256 registers) hide the base-gather latency; G2's live state needs the whole file.
// LAZY (G2 only): the mixed addition's Fq2 products with one reduction per component (ec.cuh: xyzz_madd_lazy)
template <class F, bool PIPE, bool LAZY = false>
__global__ void __launch_bounds__(64, FieldTraits<F>::g2 ? 1 : 2) msm_accumulate_kernel(AccArgs<F> a) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = *a.total_ptr;
    const uint32_t seg_len = *a.seg_len_ptr;
    const uint32_t start = t * seg_len;
    if (start >= total) return;
    const uint32_t end = start + seg_len < total ? start + seg_len : total;
    uint2 en = a.entries[start];
    const uint32_t g_first = en.y;
    const bool head_open = start > 0 && a.entries[start - 1].y == g_first;   // first bucket began in an earlier segment
    const uint32_t g_after = end < total ? a.entries[end].y : 0xffffffffu;   // bucket that continues into the next segment
    int32_t head_b = -1, tail_b = -1;
    XYZZ<F> acc = XYZZ<F>::inf();
    uint32_t cur = g_first;
    // software pipeline: entry p+1 and the base of entry p are in registers when iteration p starts.  The gather of base
    // p+1 (and the load of entry p+2) are issued between the two parts of the mixed addition: every field product of the
    // first part is a device-function call, and a call drains all outstanding loads, so a gather issued earlier would be
    // waited for at once; the second part is one inlined product (~600 instructions, no call), long enough to cover it.
    if constexpr (PIPE) {
        // G1: a finished bucket is not stored where it ends.  Its 14 stores would be in flight when the next field product is
        // called, and a call waits for ALL outstanding memory operations — measured 4.5 % of the kernel.  The sum is parked in
        // LDS (FLUSH_SLOTS lanes per iteration; further lanes of the same iteration store directly) and written out between
        // the two parts of the addition, next to the gather, where the inlined product covers it.
        constexpr bool DEFER = !FieldTraits<F>::g2;
        constexpr int FLUSH_SLOTS = 16, WORDS = (int)(sizeof(XYZZ<F>) / 4);
        __shared__ uint32_t flush_stage[DEFER ? FLUSH_SLOTS * WORDS : 1];
        XYZZ<F> *pend_dst = nullptr;
        uint32_t pend_slot = 0;
        uint2 en1 = start + 1 < end ? a.entries[start + 1] : en;
        Affine<F> bq = ldv(a.bases + (en.x >> 1));
        for (uint32_t p = start; p < end; p++) {
            const uint32_t e = en.x, gb = (a.debug & 8u) ? cur : en.y;      // bit 3: never leave the first bucket (no flush code runs)
            if (gb != cur) {
                // a bucket that began in an earlier segment is this segment's HEAD partial; one that continues into the
                // next is its TAIL partial (a bucket doing both is recorded as head only); everything else is complete.
                XYZZ<F> *dst = nullptr;
                if (cur == g_first && head_open) {
                    head_b = (int32_t)cur;
                    dst = a.seg_head + t;
                } else if (!acc.is_inf()) {
                    dst = a.buckets + cur;                                   // buckets[] is pre-zeroed = infinity (cur != g_after: it ended here)
                }
                if (a.debug & 1u) dst = nullptr;
                if constexpr (DEFER) {
                    const uint64_t want = __ballot(dst != nullptr);
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                    if (dst != nullptr && rank < (uint32_t)FLUSH_SLOTS) {
#pragma unroll
                        for (int k = 0; k < 14; k++) {
                            flush_stage[(k) * FLUSH_SLOTS + rank] = acc.x.l[k];
                            flush_stage[(14 + k) * FLUSH_SLOTS + rank] = acc.y.l[k];
                            flush_stage[(28 + k) * FLUSH_SLOTS + rank] = acc.zz.l[k];
                            flush_stage[(42 + k) * FLUSH_SLOTS + rank] = acc.zzz.l[k];
                        }
                        pend_dst = dst;
                        pend_slot = rank;
                    } else if (dst != nullptr) {
                        stv(dst, acc);
                    }
                } else if (dst != nullptr) {
                    stv(dst, acc);
                }
                acc = XYZZ<F>::inf();
                cur = gb;
            }
            MaddTail<F> tail;
            const bool normal = xyzz_madd_front(acc, bq, (e & 1u) != 0, tail);
            // unconditional (index clamped to the segment): a load inside a branch would be followed by the copies that merge its
            // result with the old value, and those wait for it at once.  The last iteration re-reads its own entry / base.
            en = en1;
            en1 = a.entries[p + 2 < end ? p + 2 : end - 1];
            bq = ldv(a.bases + ((a.debug & 2u) ? 0u : (a.debug & 4u) ? ((en.x >> 1) & 0xffffu) : (en.x >> 1)));      // bit 2: gathers confined to 64 K bases (cache-resident)
            if constexpr (DEFER) {
                if (pend_dst != nullptr) {
                    XYZZ<F> v;
#pragma unroll
                    for (int k = 0; k < 14; k++) {
                        v.x.l[k] = flush_stage[(k) * FLUSH_SLOTS + pend_slot];
                        v.y.l[k] = flush_stage[(14 + k) * FLUSH_SLOTS + pend_slot];
                        v.zz.l[k] = flush_stage[(28 + k) * FLUSH_SLOTS + pend_slot];
                        v.zzz.l[k] = flush_stage[(42 + k) * FLUSH_SLOTS + pend_slot];
                    }
                    stv(pend_dst, v);
                    pend_dst = nullptr;
                }
            }
            asm volatile("" ::: "memory");            // the loads above stay above the inlined product
            xyzz_madd_finish(acc, tail, normal);
        }
    } else {
        // round-1 form: the base is gathered right before its addition (kept selectable: option "acc_pipeline")
        uint32_t e = en.x, gb = en.y;
        for (uint32_t p = start; p < end; p++) {
            if (p + 1 < end) en = a.entries[p + 1];
            const uint32_t e_n = en.x, g_n = en.y;
            if (gb != cur) {
                if (cur == g_first && head_open) {
                    head_b = (int32_t)cur;
                    stv(a.seg_head + t, acc);
                } else if (!acc.is_inf()) {
                    stv(a.buckets + cur, acc);
                }
                acc = XYZZ<F>::inf();
                cur = gb;
            }
            const Affine<F> b = ldv(a.bases + (e >> 1));
            if constexpr (LAZY && FieldTraits<F>::g2) xyzz_madd_lazy(acc, b, (e & 1u) != 0);
            else xyzz_madd(acc, b, (e & 1u) != 0);
            e = e_n;
            gb = g_n;
        }
    }
    if (cur == g_first && head_open) {
        head_b = (int32_t)cur;
        stv(a.seg_head + t, acc);
    } else if (cur == g_after) {
        tail_b = (int32_t)cur;
        stv(a.seg_tail + t, acc);
    } else if (!acc.is_inf()) {
        stv(a.buckets + cur, acc);
    }
    a.seg_meta[2 * t] = head_b;
    a.seg_meta[2 * t + 1] = tail_b;
}

// One lane per segment whose last bucket spills over: add the head partials of the following segments.
// Chains longer than FIXUP_SHORT segments (a bucket holding thousands of terms: the value-1 scalars of the
// reference's witnesses; every bit-valued witness of the Fermat circuit) are cut into items of FIXUP_ITEM
// head partials and queued for msm_fixup_long (one workgroup per item); a chain of several items gets its item sums
// folded by msm_fixup_fold.  Without the cut a bucket spanning the whole grid (131,072 partials when every scalar is 1)
// kept one workgroup busy for 11 ms.
static constexpr int FIXUP_SHORT = 4;
static constexpr uint32_t FIXUP_ITEM = 1024;
// queue layout in MsmSlot::long_list: [0] item count, [1] chain count, then uint2 items[cap_items] = (segment, item index
// within its chain), then uint4 chains[] = (segment, first item, number of items, -)
struct FixQueue {
    uint32_t *counts;
    uint2 *items;
    uint4 *chains;
};

template <class F>
__global__ void __launch_bounds__(64) msm_fixup_kernel(AccArgs<F> a, FixQueue q) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t seg_len = *a.seg_len_ptr;
    if (t * seg_len >= (size_t)*a.total_ptr) return;
    const int32_t g = a.seg_meta[2 * t + 1];
    if (g < 0) return;
    const size_t last_seg = ((size_t)a.offsets[g + 1] - 1) / seg_len;   // segment holding the bucket's last term
    if (last_seg - t > FIXUP_SHORT) {
        const uint32_t n_items = (uint32_t)((last_seg - t + FIXUP_ITEM - 1) / FIXUP_ITEM);
        const uint32_t base = atomicAdd(q.counts, n_items);
        for (uint32_t j = 0; j < n_items; j++) q.items[base + j] = make_uint2((uint32_t)t, j);
        if (n_items > 1) q.chains[atomicAdd(q.counts + 1, 1u)] = make_uint4((uint32_t)t, base, n_items, 0u);
        return;
    }
    XYZZ<F> sum = ldv(a.seg_tail + t);
    for (size_t u = t + 1; u <= last_seg; u++) {
        const XYZZ<F> h = ldv(a.seg_head + u);
        xyzz_add(sum, h);
    }
    stv(a.buckets + g, sum);
}

// workgroup-wide sum of one XYZZ per lane (256 lanes) through LDS; result in sh[0]
template <class F>
__device__ __forceinline__ void block_fold(XYZZ<F> *sh, const XYZZ<F> &mine) {
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            XYZZ<F> x = sh[threadIdx.x];
            const XYZZ<F> y = sh[threadIdx.x + d];
            xyzz_add(x, y);
            sh[threadIdx.x] = x;
        }
        __syncthreads();
    }
}

// One 256-lane workgroup per item (grid-stride over the queue): lanes stride over the item's head partials (the first
// item of a chain also takes the tail partial), then a log-depth LDS tree folds the 256 lane sums.  A single-item chain
// goes straight to its bucket, the others to item_sums[].
template <class F>
__global__ void __launch_bounds__(256) msm_fixup_long_kernel(AccArgs<F> a, FixQueue q, XYZZ<F> *item_sums) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fix_smem[];
    XYZZ<F> *sh = reinterpret_cast<XYZZ<F> *>(fix_smem);
    const uint32_t n_items = q.counts[0];
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const size_t t = q.items[item].x, j = q.items[item].y;
        const int32_t g = a.seg_meta[2 * t + 1];
        const size_t last_seg = ((size_t)a.offsets[g + 1] - 1) / (size_t)*a.seg_len_ptr;
        const size_t lo = t + 1 + j * FIXUP_ITEM;
        const size_t hi = lo + FIXUP_ITEM - 1 < last_seg ? lo + FIXUP_ITEM - 1 : last_seg;
        XYZZ<F> sum = XYZZ<F>::inf();
        if (threadIdx.x == 0 && j == 0) sum = ldv(a.seg_tail + t);
        for (size_t u = lo + threadIdx.x; u <= hi; u += blockDim.x) {
            const XYZZ<F> h = ldv(a.seg_head + u);
            xyzz_add(sum, h);
        }
        block_fold(sh, sum);
        if (threadIdx.x == 0) {
            if (last_seg - t <= FIXUP_ITEM) stv(a.buckets + g, sh[0]);
            else stv(item_sums + item, sh[0]);
        }
        __syncthreads();
    }
}

// chains of several items: fold the item sums into the bucket
template <class F>
__global__ void __launch_bounds__(256) msm_fixup_fold_kernel(AccArgs<F> a, FixQueue q, const XYZZ<F> *item_sums) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fix_smem[];
    XYZZ<F> *sh = reinterpret_cast<XYZZ<F> *>(fix_smem);
    const uint32_t n_chains = q.counts[1];
    for (uint32_t c = blockIdx.x; c < n_chains; c += gridDim.x) {
        const uint4 ch = q.chains[c];
        XYZZ<F> sum = XYZZ<F>::inf();
        for (uint32_t i = threadIdx.x; i < ch.z; i += blockDim.x) {
            const XYZZ<F> h = ldv(item_sums + ch.y + i);
            xyzz_add(sum, h);
        }
        block_fold(sh, sum);
        if (threadIdx.x == 0) stv(a.buckets + a.seg_meta[2 * (size_t)ch.x + 1], sh[0]);
        __syncthreads();
    }
}

// A bucket that begins exactly at a segment start and spans it entirely is recorded as that segment's HEAD
// only if it began earlier; if it begins at the segment start it is a TAIL (gb_start == start, gb_end > end).
// Hence every spilled bucket has exactly one tail record followed by head records — what msm_fixup walks.

// ---- weighted bucket reduction:  W = sum_b (b+1) B_b = sum_b SS_b,  SS_b = sum_{b' >= b} B_b'  (sum of suffix sums).
// Depth matters more than work here (few thousand lanes, each a dependent chain of ~14-product additions), so the
// running sum is cut into chunks of K buckets and the cross-chunk carry comes from a log-depth suffix scan:
//   chunk_sums      S_c = sum of the chunk                                   (K-1 adds deep)
//   scan_step x log incl_c = sum_{c' >= c} S_c'  (Hillis-Steele, ping-pong)   (1 add deep each)
//   chunk_weighted  run = incl_{c+1}; for b in chunk, top down: run += B_b; acc += run     (2K adds deep)
//   sum_step x log  W = sum_c acc_c  (pairwise halving)                     (1 add deep each)
template <class F>
__global__ void __launch_bounds__(64, FieldTraits<F>::g2 ? 1 : 2) msm_chunk_sums_kernel(const XYZZ<F> *buckets, XYZZ<F> *sums, size_t nb, int k, int nwin) {
    const size_t nchunks = nb / k;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nchunks * (size_t)nwin) return;
    const size_t w = id / nchunks, c = id - w * nchunks;
    const XYZZ<F> *p = buckets + w * nb + c * k;
    XYZZ<F> acc = ldv(p);
    for (int j = 1; j < k; j++) {
        const XYZZ<F> q = ldv(p + j);
        xyzz_add(acc, q);
    }
    stv(sums + id, acc);
}

// out[c] = in[c] + in[c + d]  (within a window of m entries; beyond the end: unchanged)
template <class F>
__global__ void __launch_bounds__(64, FieldTraits<F>::g2 ? 1 : 2) msm_scan_step_kernel(const XYZZ<F> *in, XYZZ<F> *out, size_t m, size_t d, int nwin) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= m * (size_t)nwin) return;
    const size_t c = id % m;
    XYZZ<F> x = ldv(in + id);
    if (c + d < m) {
        const XYZZ<F> y = ldv(in + id + d);
        xyzz_add(x, y);
    }
    stv(out + id, x);
}

template <class F>
__global__ void __launch_bounds__(64, FieldTraits<F>::g2 ? 1 : 2) msm_chunk_weighted_kernel(const XYZZ<F> *buckets, const XYZZ<F> *incl, XYZZ<F> *out, size_t nb, int k, int nwin) {
    const size_t nchunks = nb / k;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nchunks * (size_t)nwin) return;
    const size_t w = id / nchunks, c = id - w * nchunks;
    const XYZZ<F> *p = buckets + w * nb + c * k;
    XYZZ<F> run = XYZZ<F>::inf();
    if (c + 1 < nchunks) run = ldv(incl + id + 1);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int j = k - 1; j >= 0; j--) {
        const XYZZ<F> q = ldv(p + j);
        xyzz_add(run, q);
        xyzz_add(acc, run);
    }
    stv(out + id, acc);
}

// buf[c] += buf[c + half] for c < half (within each window of `stride` entries)
template <class F>
__global__ void __launch_bounds__(64, FieldTraits<F>::g2 ? 1 : 2) msm_sum_step_kernel(XYZZ<F> *buf, size_t stride, size_t half, size_t live, int nwin) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= half * (size_t)nwin) return;
    const size_t w = id / half, c = id - w * half;
    if (c + half >= live) return;
    XYZZ<F> x = ldv(buf + w * stride + c);
    const XYZZ<F> y = ldv(buf + w * stride + c + half);
    xyzz_add(x, y);
    stv(buf + w * stride + c, x);
}

// ---- the work-efficient form (reduce_mode 1).  With b = cK + j:  (b+1) = (c+1)K - (K-1-j), so
//   W = K * sum_c (c+1) S_c  -  sum_c M_c,      S_c = sum_j B_{cK+j},   M_c = sum_j (K-1-j) B_{cK+j}
// chunk_local yields S_c and M_c in 2K-1 additions per chunk (one pass, running sum from the bottom); sum_c (c+1) S_c is
// the same weighted reduction on an array K times shorter (done with the kernels above); sum_c M_c is a pairwise tree.
// ~0.75 M additions per 17 x 2^14-bucket MSM instead of ~1.36 M, at the price of a longer dependent chain.
template <class F>
__global__ void __launch_bounds__(64, FieldTraits<F>::g2 ? 1 : 2) msm_chunk_local_kernel(const XYZZ<F> *buckets, XYZZ<F> *sums, XYZZ<F> *mom, size_t nb, int k, int nwin) {
    const size_t nchunks = nb / k;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nchunks * (size_t)nwin) return;
    const size_t w = id / nchunks, c = id - w * nchunks;
    const XYZZ<F> *p = buckets + w * nb + c * k;
    XYZZ<F> run = ldv(p);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int j = 1; j < k; j++) {
        xyzz_add(acc, run);
        const XYZZ<F> q = ldv(p + j);
        xyzz_add(run, q);
    }
    stv(sums + id, run);
    stv(mom + id, acc);
}

// ---- the bit-sliced form (reduce_mode 5; the default for one bucket set of up to 2^19 buckets, i.e. keys with window tables).
// The chains above are DEEP: ~70 dependent point additions for 2^16 buckets, and one G2 addition is ~70 us of one wave's
// instruction stream — a 32x32 proof spent 6.2 of its 11.5 ms there with 2 % of the lanes busy.  Here, after chunk_local
// (K = 4: 7 additions deep) the weights are cut by BIT:
//   sum_c (c+1) S_c = sum_t 2^t T_t + 2^bits S_top,   T_t = sum of the S_c whose c+1 has bit t set   (c+1 <= 2^bits, = only for the top)
// every T_t is a plain pairwise tree, all of them (and the tree of the M_c) side by side in ONE array of bits+1 pseudo-windows:
// msm_bit_pairs_kernel forms the first level, msm_sum_step_kernel halves it; the host does the 2 bits + 3 operations of the
// Horner over t (msm_collect).  Depth 7 + ~2 + log2(ns/2) additions instead of ~70; (bits+1) ns additions of work.
template <class F>
__global__ void __launch_bounds__(64, FieldTraits<F>::g2 ? 1 : 2) msm_bit_pairs_kernel(const XYZZ<F> *S, const XYZZ<F> *M, XYZZ<F> *out, size_t ns, int bits, int nwin) {
    const size_t half = ns / 2;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)(bits + 1) * nwin * half) return;
    const size_t t = id / ((size_t)nwin * half), rem = id - t * (size_t)nwin * half, w = rem / half, i = rem - w * half;
    const size_t c0 = 2 * i, c1 = c0 + 1;
    XYZZ<F> x;
    if ((int)t == bits) {                                    // the moments' tree
        x = ldv(M + w * ns + c0);
        const XYZZ<F> y = ldv(M + w * ns + c1);
        xyzz_add(x, y);
    } else {
        const bool take0 = ((c0 + 1) >> t) & 1, take1 = ((c1 + 1) >> t) & 1;
        x = XYZZ<F>::inf();
        if (take0) x = ldv(S + w * ns + c0);
        if (take1) {
            const XYZZ<F> y = ldv(S + w * ns + c1);
            if (take0) xyzz_add(x, y); else x = y;
        }
    }
    stv(out + id, x);
}

// ------------------------------------------------------------------------------------------------ representation changes
// proving-key bases: arkworks saturated Montgomery -> unsaturated 29-bit form (ffu.cuh), once at pk-load time
template <class FU>
__global__ void __launch_bounds__(256) convert_bases_kernel(const Affine<typename FieldTraits<FU>::Sat> *in, Affine<FU> *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Affine<typename FieldTraits<FU>::Sat> p = ldv(in + i);
    Affine<FU> o{to_u(p.x), to_u(p.y)};      // (0,0) stays (0,0)
    stv(out + i, o);
}
// window sums: entry w*stride of `in` -> saturated XYZZ for the host Horner
template <class FU>
__global__ void __launch_bounds__(64) convert_wsums_kernel(const XYZZ<FU> *in, size_t stride, XYZZ<typename FieldTraits<FU>::Sat> *out, int nwin) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwin) return;
    const XYZZ<FU> p = ldv(in + (size_t)w * stride);
    XYZZ<typename FieldTraits<FU>::Sat> o;
    if (p.is_inf()) o = XYZZ<typename FieldTraits<FU>::Sat>::inf();
    else o = XYZZ<typename FieldTraits<FU>::Sat>{to_sat(p.x), to_sat(p.y), to_sat(p.zz), to_sat(p.zzz)};
    stv(out + w, o);
}

// ------------------------------------------------------------------------------------------------ density of the B queries
// ark-groth16 keeps one b_query entry per variable even when the variable never occurs on the B side of the R1CS (its
// v_k(tau) is 0 and the entry is the point at infinity): ~19 % of the MatrixCircuit's variables.  A term whose base is
// infinity in BOTH b_g1_query and b_g2_query gets scalar 0 in the B-side plan, so it never becomes a bucket entry.
__global__ void __launch_bounds__(256) b_density_mask_kernel(const G1AffineU *b1, const G2AffineU *b2, size_t n, uint8_t *mask, uint32_t *count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1AffineU p = ldv(b1 + i);
    const G2AffineU q = ldv(b2 + i);
    const bool skip = p.is_inf() && q.is_inf();
    mask[i] = skip ? 1 : 0;
    if (skip) atomicAdd(count, 1u);
}
size_t b_density_mask_run(zkg16_ctx *ctx, const G1AffineU *b1, const G2AffineU *b2, size_t n, uint8_t *mask) {
    DevBuf cnt(sizeof(uint32_t));
    ZK_HIP(hipMemsetAsync(cnt.p, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(b_density_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, b1, b2, n, mask, cnt.as<uint32_t>());
    ZK_HIP(hipGetLastError());
    uint32_t h = 0;
    ZK_HIP(hipMemcpyAsync(&h, cnt.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return h;
}

// ------------------------------------------------------------------------------------------------ fixed base
// [s_i]G for many scalars (Groth16 setup, generator.rs's FixedBase::msm): 8-bit windows over a table of the 32 x 255 affine
// multiples d * 2^(8w) * G, all arithmetic in the unsaturated form.  Sums stay in XYZZ; the conversion to affine is batched
// (Montgomery's trick, one inversion per thread per K points) because an Fq inversion costs more than the 32 mixed adds.
template <class FU>
__global__ void __launch_bounds__(64, FieldTraits<FU>::g2 ? 1 : 2)
fixed_base_table_kernel(const Affine<typename FieldTraits<FU>::Sat> *win_bases /*nwin*/, XYZZ<FU> *table /*nwin * (2^wbits - 1)*/, int wbits,
                        int nwin) {
    const int per = (1 << wbits) - 1;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nwin * per) return;
    const int w = id / per, d = id % per + 1;
    const Affine<typename FieldTraits<FU>::Sat> bs = ldv(win_bases + w);
    const Affine<FU> b{to_u(bs.x), to_u(bs.y)};
    XYZZ<FU> acc = XYZZ<FU>::inf();
    for (int bit = wbits - 1; bit >= 0; bit--) {
        acc = xyzz_dbl(acc);
        if ((d >> bit) & 1) xyzz_madd(acc, b, false);
    }
    stv(table + id, acc);
}

// Wide windows in two levels: entry d = dh 2^k + dl of a 2k-bit window is half[2w + 1][dh] + half[2w][dl] — ONE addition on top of a
// table of k-bit windows (whose entries cost a k-step ladder each), so a 20-bit table (13.6 M entries) costs about what an 11-bit
// one did and the per-point cost of a large batch drops from 19 additions (14-bit windows) to 13.
template <class FU>
__global__ void __launch_bounds__(64, FieldTraits<FU>::g2 ? 1 : 2)
fixed_base_combine_kernel(const Affine<FU> *half /* 2 nwin x (2^k - 1) */, XYZZ<FU> *full /* nwin x (2^(2k) - 1) */, int k, int nwin) {
    const size_t per_full = ((size_t)1 << (2 * k)) - 1, per_half = ((size_t)1 << k) - 1;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)nwin * per_full) return;
    const size_t w = id / per_full, d = id % per_full + 1, dl = d & per_half, dh = d >> k;
    XYZZ<FU> acc = XYZZ<FU>::inf();
    if (dl) acc = XYZZ<FU>::from_affine(ldv(half + (2 * w) * per_half + (dl - 1)));
    if (dh) xyzz_madd(acc, ldv(half + (2 * w + 1) * per_half + (dh - 1)), false);
    stv(full + id, acc);
}

template <class FU>
__global__ void __launch_bounds__(64, FieldTraits<FU>::g2 ? 1 : 2)
fixed_base_kernel(const Affine<FU> *table, const uint32_t *scalars, size_t n, XYZZ<FU> *out, int wbits, int nwin) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int per = (1 << wbits) - 1;
    uint32_t k[9];
#pragma unroll
    for (int j = 0; j < 8; j++) k[j] = scalars[8 * i + j];
    k[8] = 0;
    XYZZ<FU> acc = XYZZ<FU>::inf();
    for (int w = 0; w < nwin; w++) {
        const int bit = w * wbits;
        const uint64_t two = (uint64_t)k[bit >> 5] | ((uint64_t)k[(bit >> 5) + 1] << 32);
        const uint32_t d = (uint32_t)(two >> (bit & 31)) & (uint32_t)per;
        if (d) {
            const Affine<FU> p = ldv(table + (size_t)w * (size_t)per + (d - 1));
            if constexpr (FieldTraits<FU>::g2) xyzz_madd_lazy(acc, p, false);      // one reduction per Fq2 component (one wave per block: ffu.cuh)
            else xyzz_madd(acc, p, false);
        }
    }
    stv(out + i, acc);
}

// outputs of one batched to-affine: consecutive index ranges [start[k], start[k+1]) go to their own arrays, each in the
// unsaturated (device-resident key) and / or arkworks' saturated form; several queries of a setup share one launch this way
template <class FU>
struct AffineSegs {
    int n;
    size_t start[8];
    Affine<FU> *out_u[8];
    Affine<typename FieldTraits<FU>::Sat> *out_sat[8];
};

// XYZZ -> affine for n points; thread t owns points t, t + T, t + 2T, ... (T = threads launched) so that a wave touches
// neighbouring points at every step.  `pref` is n field elements of scratch.  Either output may be null:
// out_u = unsaturated (device-resident key), out_sat = arkworks' saturated Montgomery form (host-bound).
template <class FU>
__global__ void __launch_bounds__(64, FieldTraits<FU>::g2 ? 1 : 2)
batch_affine_kernel(const XYZZ<FU> *pts, size_t n, FU *pref, AffineSegs<FU> segs) {
    using FS = typename FieldTraits<FU>::Sat;
    const size_t T = (size_t)gridDim.x * blockDim.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    FU acc = FU::one();
    size_t last = t;
    for (size_t i = t; i < n; i += T) {
        const XYZZ<FU> p = ldv(pts + i);
        pref[i] = acc;
        if (!p.is_inf()) acc = f_mul(acc, f_mul(p.zz, p.zzz));
        last = i;
    }
    FU inv = f_inv(acc);
    for (size_t i = last;; i -= T) {
        const XYZZ<FU> p = ldv(pts + i);
        Affine<FU> o = Affine<FU>::inf();
        if (!p.is_inf()) {
            const FU dinv = f_mul(inv, pref[i]);                  // (zz * zzz)^-1
            inv = f_mul(inv, f_mul(p.zz, p.zzz));
            o.x = f_tidy(f_mul(p.x, f_mul(dinv, p.zzz)));         // X / ZZ
            o.y = f_tidy(f_mul(p.y, f_mul(dinv, p.zz)));          // Y / ZZZ
        }
        int sg = 0;
        while (sg + 1 < segs.n && i >= segs.start[sg + 1]) sg++;
        const size_t j = i - segs.start[sg];
        if (segs.out_u[sg]) stv(segs.out_u[sg] + j, o);
        if (segs.out_sat[sg]) {
            Affine<FS> os = Affine<FS>::inf();
            if (!p.is_inf()) os = Affine<FS>{to_sat(o.x), to_sat(o.y)};
            stv(segs.out_sat[sg] + j, os);
        }
        if (i < T + t) break;       // i == t: first point of this thread
    }
}

// ------------------------------------------------------------------------------------------------ host drivers
static int pick_window_bits(zkg16_ctx *ctx, size_t n) {
    if (ctx->opt_window_bits >= 2 && ctx->opt_window_bits <= 20) return ctx->opt_window_bits;
    // 254-bit magnitudes: c = 16 and 15 leave a 14-bit top window, 13 a 7-bit one (c = 14 would leave 2 bits = 4 giant buckets)
    if (n >= ((size_t)1 << 23)) return 17;         // 128x128 circuit (8.7 M / 16.8 M terms): 15 windows instead of 16; measured 191.3 -> 185.7 ms
    if (n >= ((size_t)1 << 20)) return 16;
    if (n >= ((size_t)1 << 17)) return 15;
    if (n >= ((size_t)1 << 14)) return 13;
    int lg = 0;
    while (((size_t)2 << lg) <= n) lg++;          // floor(log2 n)
    int c = lg - 3;
    if (c < 4) c = 4;
    return c;
}

void msm_plan_build(zkg16_ctx *ctx, MsmWorkspace &ws, const Fr *scalars_canonical, size_t n, MsmPlan &plan, int window_bits) {
    const ScalarSrc src{scalars_canonical, n, nullptr, 0, false, nullptr};
    msm_plan_build(ctx, ws, src, plan, window_bits);
}
// tabled (window tables, msm_tables_build): the bases are [nwin_digits][n] with level w = 2^(c w) * base, so digit w of scalar i
// is an ordinary term (base w * n + i, digit) of ONE bucket set — the digit codes [w][i] already are that flat term list, and
// the scatter sees a single window of n * nwin_digits terms.  Fewer additions (larger c at the same bucket count) and one
// bucket reduction instead of one per window.
void msm_plan_build(zkg16_ctx *ctx, MsmWorkspace &ws, const ScalarSrc &src, MsmPlan &plan, int window_bits, bool tabled) {
    const size_t n = src.n_main + src.n_extra;
    plan.n = n;
    plan.tabled = tabled;
    plan.c = (window_bits >= 2 && window_bits <= (tabled ? 24 : 20)) ? window_bits : pick_window_bits(ctx, n);
    plan.nwin_digits = 254 / plan.c + 1;      // magnitudes are < 2^254 after the r - s fold (msm_digits_kernel)
    plan.nwin = tabled ? 1 : plan.nwin_digits;
    plan.nb = (size_t)1 << (plan.c - 1);
    plan.total_entries = 0;
    ws.last_tb = 0;
    if (n == 0) return;
    const size_t tb = plan.nb * plan.nwin;
    ws.last_tb = tb;
    const size_t tot = n * (size_t)plan.nwin_digits;
    if (tot >= ((size_t)1 << 31)) throw HipError{hipErrorInvalidValue, "msm: more than 2^31 (scalar, window) terms", __FILE__, __LINE__};
    ws.entries.ensure(tot * sizeof(uint64_t));
    ws.offsets.ensure((tb + 1) * sizeof(uint32_t));
    const bool own_sort = ctx->opt_sort_mode == 0 || tabled;
    const DigitSrc d{reinterpret_cast<const uint32_t *>(src.main), reinterpret_cast<const uint32_t *>(src.extra), src.n_main, n, src.mask,
                     src.mont ? 1 : 0, src.part, src.want_part};
    const uint32_t *win_total = nullptr;
    if (own_sort) {
        ws.codes.ensure(tot * sizeof(uint32_t));
        {
            ScopedKernelTimer kt(ctx, "msm_digits_kernel", (double)n, ctx->stream);
            hipLaunchKernelGGL(msm_digits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d, plan.c, plan.nwin_digits, plan.nb,
                               (uint64_t *)nullptr, ws.codes.as<uint32_t>(), (uint32_t)tb);
        }
        if (tabled) win_total = msm_bucket_sort(ctx, ws, ws.codes.as<uint32_t>(), tot, 1, plan.c, ws.entries.as<uint2>());
        else win_total = msm_bucket_sort(ctx, ws, ws.codes.as<uint32_t>(), n, plan.nwin, plan.c, ws.entries.as<uint2>());
    } else {
        ws.keys.ensure(tot * sizeof(uint64_t));
        {
            ScopedKernelTimer kt(ctx, "msm_digits_kernel", (double)n, ctx->stream);
            hipLaunchKernelGGL(msm_digits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d, plan.c, plan.nwin, plan.nb,
                               ws.keys.as<uint64_t>(), (uint32_t *)nullptr, (uint32_t)tb);
        }
        unsigned key_bits = 1;
        while (((size_t)1 << key_bits) <= tb) key_bits++;
        msm_sort_keys(ctx, ws, tot, key_bits);
    }
    {
        ScopedKernelTimer kt(ctx, "msm_offsets_kernel", (double)tb, ctx->stream);
        hipLaunchKernelGGL(msm_offsets_kernel, dim3((unsigned)((tb + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                           ws.entries.as<uint2>(), tot, win_total, plan.nwin, ws.offsets.as<uint32_t>(), tb);
    }
    ZK_HIP(hipGetLastError());
    // the exact entry count stays on the device (offsets[tb]); the accumulation grids are one resident round of waves and
    // the per-lane segment length is derived from the count on the device
    plan.total_entries = tot;
    // G1: two waves per SIMD, four from 2^25 terms on (124 registers: four fit; the extra waves hide what is left of the gather
    // latency — 128x128: 163.0 -> 160.8 ms; no change at 32x32, where a lane would get ~25 terms)
    plan.lanes_g1 = (uint32_t)ctx->num_cus * 4u * (uint32_t)(ctx->opt_g1_waves > 0 ? ctx->opt_g1_waves : tot >= ((size_t)1 << 25) ? 4 : 2) * 64u;
    plan.lanes_g2 = (uint32_t)ctx->num_cus * 4u * 1u * 64u;
    ws.last_lanes_g1 = plan.lanes_g1;
    ws.seg_params.ensure(2 * sizeof(uint32_t));
    hipLaunchKernelGGL(msm_seg_params_kernel, dim3(1), dim3(64), 0, ctx->stream, ws.offsets.as<uint32_t>() + tb, (uint32_t)tb, plan.lanes_g1,
                       plan.lanes_g2, (uint32_t)(ctx->opt_min_seg > 0 ? ctx->opt_min_seg : 0), ws.seg_params.as<uint32_t>());
    ZK_HIP(hipGetLastError());
}

// ---- the B-side term list as a FILTER of the full one.  B1 and B2 skip the terms whose base is the point at infinity in both B
// queries (mask[i] != 0).  Sorting the scalars a second time with those zeroed costs a whole digit + scatter run (3.5 ms at
// 128x128); the sorted full list minus the masked terms is the same list (the scatter is stable, and the masked terms are
// simply absent), so it is produced by one stable compaction: per-workgroup counts of kept terms, a scan, ballot ranks.
static constexpr int FILTER_CHUNK = 2048;        // terms per workgroup (256 threads x 8)
__device__ __forceinline__ bool filter_keep(uint2 e, uint32_t n, double inv_n, const uint8_t *mask) {
    const uint32_t idx = e.x >> 1;                               // window * n + i with window tables, i without
    uint32_t q = (uint32_t)((double)idx * inv_n);
    uint32_t i = idx - q * n;
    if ((int32_t)i < 0) i += n; else if (i >= n) i -= n;         // the estimate of q is off by at most one
    return mask[i] == 0;
}
__global__ void __launch_bounds__(256) filter_count_kernel(const uint2 *in, const uint32_t *total_ptr, uint32_t n, double inv_n, const uint8_t *mask,
                                                           uint32_t *block_counts) {
    __shared__ uint32_t wsum[4];
    const uint32_t total = *total_ptr;
    const size_t base = (size_t)blockIdx.x * FILTER_CHUNK;
    uint32_t mine = 0;
#pragma unroll
    for (int r = 0; r < FILTER_CHUNK / 256; r++) {
        const size_t k = base + (size_t)r * 256 + threadIdx.x;
        if (k < total && filter_keep(in[k], n, inv_n, mask)) mine++;
    }
    for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// one workgroup: exclusive scan of the per-workgroup counts (in place); counts[nblk] = total kept
__global__ void __launch_bounds__(1024) filter_scan_kernel(uint32_t *counts, uint32_t nblk) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t per = ((nblk + 15u) / 16u + 63u) & ~63u;
    const uint32_t lo = wv * per, hi = lo + per < nblk ? lo + per : nblk;
    uint32_t carry = 0;
    for (uint32_t b = lo; b < hi; b += 64) {
        const uint32_t i = b + lane;
        uint32_t v = i < hi ? counts[i] : 0u;
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        carry += v;
    }
    if (lane == 0) wave_tot[wv] = carry;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t x = 0; x < 16; x++) {
        const uint32_t t = wave_tot[x];
        if (x < wv) before += t;
        total += t;
    }
    carry = before;
    for (uint32_t b = lo; b < hi; b += 64) {
        const uint32_t i = b + lane;
        const uint32_t v = i < hi ? counts[i] : 0u;
        uint32_t inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += up;
        }
        if (i < hi) counts[i] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    __syncthreads();
    if (threadIdx.x == 0) counts[nblk] = total;
}
__global__ void __launch_bounds__(256) filter_write_kernel(const uint2 *in, const uint32_t *total_ptr, uint32_t n, double inv_n, const uint8_t *mask,
                                                           const uint32_t *block_base, uint2 *out) {
    __shared__ uint32_t wcount[FILTER_CHUNK / 256][4];
    const uint32_t total = *total_ptr;
    const size_t base = (size_t)blockIdx.x * FILTER_CHUNK;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint2 e[FILTER_CHUNK / 256];
    uint32_t below[FILTER_CHUNK / 256], keepmask = 0;
#pragma unroll
    for (int r = 0; r < FILTER_CHUNK / 256; r++) {
        const size_t k = base + (size_t)r * 256 + threadIdx.x;
        e[r] = make_uint2(0, 0);
        bool keep = false;
        if (k < total) {
            e[r] = in[k];
            keep = filter_keep(e[r], n, inv_n, mask);
        }
        const uint64_t peers = __ballot(keep);
        below[r] = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        if (lane == 0) wcount[r][wv] = (uint32_t)__popcll(peers);
        if (keep) keepmask |= 1u << r;
    }
    __syncthreads();
    uint32_t pos = block_base[blockIdx.x];
#pragma unroll
    for (int r = 0; r < FILTER_CHUNK / 256; r++) {       // element order = (round, wave, lane): the order of the input
        uint32_t before = 0, round_total = 0;
        for (uint32_t x = 0; x < 4; x++) {
            if (x < wv) before += wcount[r][x];
            round_total += wcount[r][x];
        }
        if (keepmask & (1u << r)) out[pos + before + below[r]] = e[r];
        pos += round_total;
    }
}
// plan_dst / ws_dst: the list of plan_src / ws_src without the terms whose scalar index is masked; same buckets, same widths
void msm_plan_filter(zkg16_ctx *ctx, const MsmWorkspace &ws_src, const MsmPlan &plan_src, const uint8_t *mask, MsmWorkspace &ws_dst, MsmPlan &plan_dst) {
    plan_dst = plan_src;
    ws_dst.last_tb = 0;
    if (plan_src.n == 0) return;
    const size_t tb = plan_src.nb * plan_src.nwin, tot = plan_src.total_entries;
    ws_dst.last_tb = tb;
    ws_dst.last_lanes_g1 = plan_dst.lanes_g1;
    ws_dst.entries.ensure(tot * sizeof(uint64_t));
    ws_dst.offsets.ensure((tb + 1) * sizeof(uint32_t));
    ws_dst.seg_params.ensure(2 * sizeof(uint32_t));
    const uint32_t nblk = (uint32_t)((tot + FILTER_CHUNK - 1) / FILTER_CHUNK);
    ws_dst.sort_temp.ensure(((size_t)nblk + 1) * sizeof(uint32_t));
    uint32_t *counts = ws_dst.sort_temp.as<uint32_t>();
    const uint32_t *total_src = ws_src.offsets.as<uint32_t>() + tb;
    const uint32_t n = (uint32_t)plan_src.n;
    const double inv_n = 1.0 / (double)n;
    {
        ScopedKernelTimer kt(ctx, "msm_plan_filter", (double)tot, ctx->stream);
        hipLaunchKernelGGL(filter_count_kernel, dim3(nblk), dim3(256), 0, ctx->stream, ws_src.entries.as<uint2>(), total_src, n, inv_n, mask, counts);
        hipLaunchKernelGGL(filter_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, counts, nblk);
        hipLaunchKernelGGL(filter_write_kernel, dim3(nblk), dim3(256), 0, ctx->stream, ws_src.entries.as<uint2>(), total_src, n, inv_n, mask, counts,
                           ws_dst.entries.as<uint2>());
        hipLaunchKernelGGL(msm_offsets_kernel, dim3((unsigned)((tb + 1 + 255) / 256)), dim3(256), 0, ctx->stream, ws_dst.entries.as<uint2>(), tot,
                           counts + nblk, 1, ws_dst.offsets.as<uint32_t>(), tb);
    }
    hipLaunchKernelGGL(msm_seg_params_kernel, dim3(1), dim3(64), 0, ctx->stream, ws_dst.offsets.as<uint32_t>() + tb, (uint32_t)tb, plan_dst.lanes_g1,
                       plan_dst.lanes_g2, (uint32_t)(ctx->opt_min_seg > 0 ? ctx->opt_min_seg : 0), ws_dst.seg_params.as<uint32_t>());
    ZK_HIP(hipGetLastError());
}

// An MSM runs in two halves on two HIP streams so that the latency-bound bucket reduction of one MSM overlaps the
// throughput-bound bucket accumulation of the next:
//   main stream: clear buckets -> accumulate -> fix-ups                      -> event acc_done
//   aux  stream: wait acc_done -> reduction -> convert window sums -> D2H   -> event red_done
//   host       : wait red_done -> Horner over the window sums (msm_collect)
// bucket clears go through an ordinary kernel on the MSM's own stream (not hipMemsetAsync)
__global__ void __launch_bounds__(256) zero_fill_kernel(uint4 *p, size_t n16, uint32_t *also) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0, 0, 0, 0);
    if (i == 0 && also) also[0] = also[1] = 0;
}
static void zero_fill(hipStream_t st, void *p, size_t bytes, uint32_t *also) {
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<uint4 *>(p), n16, also);
}

// head/tail partials of the lanes -> buckets (short chains per bucket), then the few buckets that span many lanes
template <class F>
static FixQueue fix_queue(MsmSlot &slot, size_t nseg) {
    const size_t cap_items = nseg + 16, cap_chains = nseg / FIXUP_ITEM + 16;
    slot.long_list.ensure(16 + cap_items * sizeof(uint2) + cap_chains * sizeof(uint4));
    slot.long_sums.ensure(cap_items * sizeof(XYZZ<F>));
    unsigned char *p = slot.long_list.as<unsigned char>();
    return FixQueue{reinterpret_cast<uint32_t *>(p), reinterpret_cast<uint2 *>(p + 16), reinterpret_cast<uint4 *>(p + 16 + cap_items * sizeof(uint2))};
}
// head/tail partials of the lanes -> buckets (short chains per bucket), then the few buckets that span many lanes
template <class F>
static void msm_launch_fixups(zkg16_ctx *ctx, MsmSlot &slot, hipStream_t fs) {
    AccArgs<F> a;
    memcpy(&a, slot.acc_args, sizeof a);
    const size_t psz = sizeof(XYZZ<F>);
    const FixQueue q = fix_queue<F>(slot, (size_t)slot.acc_grid * 64);
    XYZZ<F> *sums = slot.long_sums.as<XYZZ<F>>();
    {
        ScopedKernelTimer kt(ctx, FieldTraits<F>::g2 ? "msm_fixup_g2" : "msm_fixup_g1", (double)slot.acc_grid * 64, fs);
        hipLaunchKernelGGL(msm_fixup_kernel<F>, dim3(slot.acc_grid), dim3(64), 0, fs, a, q);
    }
    bool &lds_attr_set = ctx->lds_attr_fixup[FieldTraits<F>::g2 ? 1 : 0];     // per ctx (= per device): 256 * 448 B > the 64 KiB default for G2
    if (!lds_attr_set) {
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_fixup_long_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msm_fixup_fold_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_attr_set = true;
    }
    {
        ScopedKernelTimer kt(ctx, FieldTraits<F>::g2 ? "msm_fixup_long_g2" : "msm_fixup_long_g1", 0.0, fs);
        hipLaunchKernelGGL(msm_fixup_long_kernel<F>, dim3(512), dim3(256), 256 * psz, fs, a, q, sums);
        hipLaunchKernelGGL(msm_fixup_fold_kernel<F>, dim3(64), dim3(256), 256 * psz, fs, a, q, sums);
    }
}

// sum[i] += add[i] for the bucket arrays of two rounds of one MSM (a full XYZZ addition per bucket that got terms in the later
// round: <= 2^20 buckets per MSM, i.e. microseconds beside the accumulation it follows)
template <class F>
__global__ void __launch_bounds__(64, FieldTraits<F>::g2 ? 1 : 2) msm_bucket_merge_kernel(XYZZ<F> *sum, const XYZZ<F> *add, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const XYZZ<F> b = ldv(add + i);
    if (b.is_inf()) return;
    XYZZ<F> a = ldv(sum + i);
    xyzz_add(a, b);
    stv(sum + i, a);
}

template <class F>
static void msm_enqueue_acc(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const Affine<F> *bases, MsmSlot &slot, int round = -1) {
    if (round <= 0) {
        slot.active = false;
        slot.pending_reduce = false;
    }
    slot.nwin = plan.nwin;
    slot.c = plan.c;
    if (plan.n == 0) return;
    const size_t tb = plan.nb * plan.nwin;
    const size_t psz = sizeof(XYZZ<F>);
    if (!slot.acc_done) {
        ZK_HIP(hipEventCreate(&slot.acc_done));
        ZK_HIP(hipEventCreate(&slot.red_done));
        ZK_HIP(hipEventCreate(&slot.acc_start));
        ZK_HIP(hipEventCreate(&slot.red_start));
        if (!slot.stream) ZK_HIP(hipStreamCreateWithFlags(&slot.stream, hipStreamNonBlocking));
    }
    if (round <= 0) ZK_HIP(hipEventRecord(slot.acc_start, ctx->stream));
    XYZZ<F> *target;
    if (round == 0) {
        slot.bucket_sum.ensure(tb * psz);
        target = slot.bucket_sum.as<XYZZ<F>>();
    } else {
        slot.buckets.ensure(tb * psz);
        target = slot.buckets.as<XYZZ<F>>();
    }
    const size_t nseg = FieldTraits<F>::g2 ? plan.lanes_g2 : plan.lanes_g1;      // lanes of one resident round
    slot.seg_head.ensure(nseg * psz);
    slot.seg_tail.ensure(nseg * psz);
    slot.seg_meta.ensure(nseg * 2 * sizeof(int32_t));
    const FixQueue fq = fix_queue<F>(slot, (nseg + 63) / 64 * 64);
    zero_fill(ctx->stream, target, tb * psz, fq.counts);      // + the two fix-up queue counters
    AccArgs<F> a;
    a.bases = bases;
    a.entries = ws.entries.as<uint2>();
    a.offsets = ws.offsets.as<uint32_t>();
    a.buckets = target;
    a.seg_head = slot.seg_head.as<XYZZ<F>>();
    a.seg_tail = slot.seg_tail.as<XYZZ<F>>();
    a.seg_meta = slot.seg_meta.as<int32_t>();
    a.total_ptr = ws.offsets.as<uint32_t>() + tb;
    a.total_buckets = tb;
    a.seg_len_ptr = ws.seg_params.as<uint32_t>() + (FieldTraits<F>::g2 ? 1 : 0);
    a.debug = (uint32_t)ctx->opt_acc_debug;
    const unsigned grid = (unsigned)((nseg + 63) / 64);
    {
        ScopedKernelTimer kt(ctx, FieldTraits<F>::g2 ? "msm_accumulate_g2" : "msm_accumulate_g1", (double)plan.n, ctx->stream);
        const bool pipe = (ctx->opt_acc_pipeline >> (FieldTraits<F>::g2 ? 1 : 0)) & 1;
        if (pipe) hipLaunchKernelGGL((msm_accumulate_kernel<F, true>), dim3(grid), dim3(64), 0, ctx->stream, a);
        else if (FieldTraits<F>::g2 && ctx->opt_g2_lazy) hipLaunchKernelGGL((msm_accumulate_kernel<F, false, true>), dim3(grid), dim3(64), 0, ctx->stream, a);
        else hipLaunchKernelGGL((msm_accumulate_kernel<F, false>), dim3(grid), dim3(64), 0, ctx->stream, a);
    }
    static_assert(sizeof(AccArgs<F>) <= sizeof(slot.acc_args), "MsmSlot::acc_args too small");
    memcpy(slot.acc_args, &a, sizeof a);
    slot.acc_grid = grid;
    slot.fixups_pending = ctx->opt_fixup_aux != 0 && round < 0;      // rounds are merged right away: their fix-ups come first
    if (!slot.fixups_pending) msm_launch_fixups<F>(ctx, slot, ctx->stream);
    if (round > 0) {
        ScopedKernelTimer kt(ctx, FieldTraits<F>::g2 ? "msm_bucket_merge_g2" : "msm_bucket_merge_g1", (double)tb, ctx->stream);
        hipLaunchKernelGGL(msm_bucket_merge_kernel<F>, dim3((unsigned)((tb + 63) / 64)), dim3(64), 0, ctx->stream, slot.bucket_sum.as<XYZZ<F>>(), target, tb);
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipEventRecord(slot.acc_done, ctx->stream));
    slot.red_buckets = round >= 0 ? (void *)slot.bucket_sum.as<XYZZ<F>>() : (void *)a.buckets;
    slot.red_nb = plan.nb;
    slot.pending_reduce = true;
}

// ---- this slot's own stream: the weighted bucket reduction, while the main stream already runs the next MSM's
// accumulation; the five reductions of a proof are latency-bound chains on a few hundred waves each, so they also run
// concurrently with each other.  Queued separately from the accumulation: its first packet is a wait on acc_done, and
// HIP multiplexes streams onto a few hardware queues, so a wait queued early parks whatever is queued behind it.
template <class F>
static void msm_enqueue_reduce(zkg16_ctx *ctx, MsmSlot &slot) {
    using FS = typename FieldTraits<F>::Sat;
    if (!slot.pending_reduce) return;
    slot.pending_reduce = false;
    MsmPlan plan;
    plan.nb = slot.red_nb;
    plan.nwin = slot.nwin;
    const size_t tb = plan.nb * plan.nwin;
    const size_t psz = sizeof(XYZZ<F>);
    struct { XYZZ<F> *buckets; } a{reinterpret_cast<XYZZ<F> *>(slot.red_buckets)};
    hipStream_t aux = slot.stream;
    ZK_HIP(hipStreamWaitEvent(aux, slot.acc_done, 0));
    ZK_HIP(hipEventRecord(slot.red_start, aux));
    if (slot.fixups_pending) {      // the next accumulation on the main stream does not depend on them
        msm_launch_fixups<F>(ctx, slot, aux);
        slot.fixups_pending = false;
    }
    const char *rname = FieldTraits<F>::g2 ? "msm_reduce_g2" : "msm_reduce_g1";
    // buckets per lane: 8 when the reduction has enough lanes to matter as work; fewer when it is a short, purely
    // latency-bound chain (small circuits, 1/8 shards): measured 4.4 -> 3.7 ms at 6,476 constraints, 5.6 -> 5.1 ms at 60,684
    int kk = ctx->opt_reduce_chunk > 0 ? ctx->opt_reduce_chunk : (tb <= 16384 ? 2 : tb <= 131072 ? 4 : 8);
    while ((size_t)kk > plan.nb) kk >>= 1;
    const size_t nchunks = plan.nb / kk;
    const size_t tot = nchunks * plan.nwin;
    const int kk2 = 8;
    // work-efficient form: mode 1 everywhere; mode 2 everywhere except the MSM whose reduction is the exposed tail of the proof
    // (slot.last_of_proof, set by the caller): the others overlap the next accumulation, where work, not depth, is what costs
    // default (3): mode 2 from 16-bit windows on (measured 128x128: 184.1 -> 182.5 ms; 32x32: no change; a 6,476-constraint proof
    // 3.6 -> 4.8 ms with it, so small windows keep the short chain)
    const int rmode = ctx->opt_reduce_mode >= 5 ? 3 : ctx->opt_reduce_mode;          // 5 / 6: the default with / without the bit-sliced form
    const bool efficient = rmode == 1 || (rmode == 2 && !slot.last_of_proof) || (rmode == 3 && !slot.last_of_proof && slot.c >= 16);
    slot.two_level_k = (efficient && kk >= 2 && nchunks >= 2 * (size_t)kk2) ? kk : 0;
    // bit-sliced: bucket sets of a size where depth, not work, is what the reduction costs — one set of up to 2^19 buckets (window
    // tables), or the windows of a plain key from 13-bit windows on while they hold up to 2^19 buckets together (the host combines
    // (2 bits + 3) nwin sums: ~0.5 us per G1 and ~1.4 us per G2 operation, which smaller circuits cannot hide — 8x8: 4.4 -> 5.3 ms)
    slot.bit_sliced = 0;
    const bool pow2 = (plan.nb & (plan.nb - 1)) == 0;
    // (the proof's LAST reduction is an exposed tail whatever its size: bit-sliced up to 2^22 buckets — 128x128's H, 2^21: 3.0 -> 1.6 ms)
    const size_t one_set_max = slot.last_of_proof ? ((size_t)1 << 22) : ((size_t)1 << 19);
    const bool bs_auto = ctx->opt_reduce_mode == 3 && (plan.nwin == 1 ? plan.nb >= 256 && plan.nb <= one_set_max : plan.nb >= 4096 && tb <= ((size_t)1 << 19));
    if (pow2 && plan.nb >= 8 && (ctx->opt_reduce_mode == 5 || bs_auto)) {
        auto log2z = [](size_t v) { int l = 0; while (((size_t)1 << l) < v) l++; return l; };
        // chunk size by depth in dependent additions: 2K - 1 in chunk_local, the first tree level in rounds of the resident lanes (a G2
        // wave fills a SIMD, two G1 waves do; the following levels add about as much again), then one per halving
        const double resident = FieldTraits<F>::g2 ? 65536.0 : 131072.0;
        int kb = 2;
        double best = 1e30;
        for (int k = 2; k <= 16 && (size_t)k * 4 <= plan.nb; k <<= 1) {
            const size_t ns = plan.nb / k;
            const double rounds = (double)(log2z(ns) + 1) * plan.nwin * (ns / 2) / resident;
            const double depth = 2 * k - 1 + 2 * (rounds > 1 ? rounds : 1) + log2z(ns) - 1;
            if (depth < best) { best = depth; kb = k; }
        }
        if (ctx->opt_reduce_chunk > 1 && (ctx->opt_reduce_chunk & (ctx->opt_reduce_chunk - 1)) == 0) kb = ctx->opt_reduce_chunk;
        while (kb > 2 && (size_t)kb * 4 > plan.nb) kb >>= 1;
        slot.two_level_k = kb;
        slot.bit_sliced = log2z(plan.nb / kb);
    }
    const size_t nout = slot.bit_sliced ? (size_t)(slot.bit_sliced + 2) * plan.nwin : (slot.two_level_k ? 2 : 1) * (size_t)plan.nwin;
    const size_t tot_sm = slot.bit_sliced ? (plan.nb / slot.two_level_k) * plan.nwin : tot;      // chunk sums and moments
    slot.red_a.ensure(tot_sm * psz);
    slot.red_b.ensure(tot_sm * psz);
    slot.red_c.ensure((slot.bit_sliced ? (size_t)(slot.bit_sliced + 1) * plan.nwin * (plan.nb / slot.two_level_k / 2) : tot) * psz);
    // the window sums are written straight into pinned host memory by the conversion kernel (device-visible, coherent): no
    // device-to-host copy is queued, so no runtime blit kernel appears inside a proof
    if (slot.host_bytes < nout * sizeof(XYZZ<FS>)) {
        if (slot.wsums_host) (void)hipHostFree(slot.wsums_host);
        slot.host_bytes = 128 * sizeof(XYZZ<FS>) > nout * sizeof(XYZZ<FS>) ? 128 * sizeof(XYZZ<FS>) : nout * sizeof(XYZZ<FS>);
        ZK_HIP(hipHostMalloc(&slot.wsums_host, slot.host_bytes, hipHostMallocDefault));
    }
    XYZZ<FS> *wsums_out = reinterpret_cast<XYZZ<FS> *>(slot.wsums_host);
    XYZZ<F> *pa = slot.red_a.as<XYZZ<F>>(), *pb = slot.red_b.as<XYZZ<F>>(), *pc = slot.red_c.as<XYZZ<F>>();
    const unsigned rgrid = (unsigned)((tot + 63) / 64);
    // weighted reduction of `src_buckets` ([nwin][nb_] entries) with chunks of k_: result of window w at res[w * (nb_ / k_)]
    auto weighted = [&](const XYZZ<F> *src_buckets, size_t nb_, int k_, XYZZ<F> *p0, XYZZ<F> *p1, XYZZ<F> *res) {
        const size_t nch = nb_ / k_;
        const unsigned g = (unsigned)((nch * plan.nwin + 63) / 64);
        hipLaunchKernelGGL(msm_chunk_sums_kernel<F>, dim3(g), dim3(64), 0, aux, src_buckets, p0, nb_, k_, plan.nwin);
        XYZZ<F> *src = p0, *dst = p1;
        for (size_t d = 1; d < nch; d <<= 1) {
            hipLaunchKernelGGL(msm_scan_step_kernel<F>, dim3(g), dim3(64), 0, aux, src, dst, nch, d, plan.nwin);
            XYZZ<F> *t = src; src = dst; dst = t;
        }
        hipLaunchKernelGGL(msm_chunk_weighted_kernel<F>, dim3(g), dim3(64), 0, aux, src_buckets, src, res, nb_, k_, plan.nwin);
        size_t live = nch;
        while (live > 1) {
            const size_t half = (live + 1) / 2;
            hipLaunchKernelGGL(msm_sum_step_kernel<F>, dim3((unsigned)((half * plan.nwin + 63) / 64)), dim3(64), 0, aux, res, nch, half, live, plan.nwin);
            live = half;
        }
    };
    {
        ScopedKernelTimer kt(ctx, rname, (double)tb, aux);
        if (slot.bit_sliced) {
            const int kb = slot.two_level_k, bits = slot.bit_sliced;
            const size_t ns = plan.nb / kb, half = ns / 2;
            const int pw = (bits + 1) * plan.nwin;             // pseudo-windows: T_0 .. T_(bits-1), then the moments
            hipLaunchKernelGGL(msm_chunk_local_kernel<F>, dim3((unsigned)((ns * plan.nwin + 63) / 64)), dim3(64), 0, aux, a.buckets, pa, pb, plan.nb, kb, plan.nwin);
            hipLaunchKernelGGL(msm_bit_pairs_kernel<F>, dim3((unsigned)(((size_t)pw * half + 63) / 64)), dim3(64), 0, aux, pa, pb, pc, ns, bits, plan.nwin);
            size_t live = half;
            while (live > 1) {
                const size_t h2 = (live + 1) / 2;
                hipLaunchKernelGGL(msm_sum_step_kernel<F>, dim3((unsigned)((h2 * pw + 63) / 64)), dim3(64), 0, aux, pc, half, h2, live, pw);
                live = h2;
            }
            hipLaunchKernelGGL(convert_wsums_kernel<F>, dim3((pw + 63) / 64), dim3(64), 0, aux, pc, half, wsums_out, pw);
            hipLaunchKernelGGL(convert_wsums_kernel<F>, dim3((plan.nwin + 63) / 64), dim3(64), 0, aux, pa + (ns - 1), ns, wsums_out + pw, plan.nwin);
        } else if (!slot.two_level_k) {
            weighted(a.buckets, plan.nb, kk, pa, pb, pc);
            hipLaunchKernelGGL(convert_wsums_kernel<F>, dim3((plan.nwin + 63) / 64), dim3(64), 0, aux, pc, nchunks, wsums_out, plan.nwin);
        } else {
            // S -> pa, M -> pb; the second level works in pc, split in three
            hipLaunchKernelGGL(msm_chunk_local_kernel<F>, dim3(rgrid), dim3(64), 0, aux, a.buckets, pa, pb, plan.nb, kk, plan.nwin);
            const size_t n2 = (nchunks / kk2) * plan.nwin;
            weighted(pa, nchunks, kk2, pc, pc + n2, pc + 2 * n2);
            size_t live = nchunks;
            while (live > 1) {
                const size_t half = (live + 1) / 2;
                hipLaunchKernelGGL(msm_sum_step_kernel<F>, dim3((unsigned)((half * plan.nwin + 63) / 64)), dim3(64), 0, aux, pb, nchunks, half, live, plan.nwin);
                live = half;
            }
            hipLaunchKernelGGL(convert_wsums_kernel<F>, dim3((plan.nwin + 63) / 64), dim3(64), 0, aux, pc + 2 * n2, nchunks / kk2, wsums_out, plan.nwin);
            hipLaunchKernelGGL(convert_wsums_kernel<F>, dim3((plan.nwin + 63) / 64), dim3(64), 0, aux, pb, nchunks, wsums_out + plan.nwin, plan.nwin);
        }
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipEventRecord(slot.red_done, aux));
    slot.active = true;
}
template <class F>
static void msm_enqueue(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const Affine<F> *bases, MsmSlot &slot) {
    msm_enqueue_acc<F>(ctx, ws, plan, bases, slot);
    msm_enqueue_reduce<F>(ctx, slot);
}

template <class FS>
static XYZZ<FS> msm_collect(zkg16_ctx *ctx, MsmSlot &slot) {
    (void)ctx;
    if (!slot.active) return XYZZ<FS>::inf();
    ZK_HIP(hipEventSynchronize(slot.red_done));
    const auto host_t0 = std::chrono::steady_clock::now();
    struct Lap { const std::chrono::steady_clock::time_point t0; MsmSlot &s; ~Lap() { s.collect_host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } lap{host_t0, slot};
    const XYZZ<FS> *wsum = reinterpret_cast<const XYZZ<FS> *>(slot.wsums_host);
    auto window = [&](int w) {
        if (slot.bit_sliced) {          // K (2^bits S_top + sum_t 2^t T_t) - M   (msm_bit_pairs_kernel)
            const int bits = slot.bit_sliced;
            XYZZ<FS> v = wsum[(bits + 1) * slot.nwin + w];
            for (int t = bits - 1; t >= 0; t--) {
                v = xyzz_dbl(v);
                xyzz_add(v, wsum[t * slot.nwin + w]);
            }
            for (int k = slot.two_level_k; k > 1; k >>= 1) v = xyzz_dbl(v);
            xyzz_add(v, xyzz_neg(wsum[bits * slot.nwin + w]));
            return v;
        }
        if (!slot.two_level_k) return wsum[w];
        XYZZ<FS> v = wsum[w];                       // W_w = K * P_w - M_w  (msm_chunk_local_kernel)
        for (int k = slot.two_level_k; k > 1; k >>= 1) v = xyzz_dbl(v);
        xyzz_add(v, xyzz_neg(wsum[slot.nwin + w]));
        return v;
    };
    // host Horner over windows: sum_w 2^(c*w) W_w
    XYZZ<FS> total = window(slot.nwin - 1);
    for (int w = slot.nwin - 2; w >= 0; w--) {
        for (int q = 0; q < slot.c; q++) total = xyzz_dbl(total);
        xyzz_add(total, window(w));
    }
    slot.active = false;
    return total;
}

// Waves of the shipped accumulation kernels one SIMD can hold at once (registers: 255 for G1, 478 for G2 in this build): what the
// bare-loop comparison of bench.py has to be read against, beside the waves per SIMD the grid is sized for (zkg16_last_acc_waves).
int msm_acc_resident_waves(zkg16_ctx *ctx, bool g2) {
    int blocks = 0;
    const bool pipe = (ctx->opt_acc_pipeline >> (g2 ? 1 : 0)) & 1;
    hipError_t e;
    if (g2) {
        if (pipe) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, msm_accumulate_kernel<Fq2U, true>, 64, 0);
        else if (ctx->opt_g2_lazy) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, msm_accumulate_kernel<Fq2U, false, true>, 64, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, msm_accumulate_kernel<Fq2U, false>, 64, 0);
    } else {
        if (pipe) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, msm_accumulate_kernel<FqU, true>, 64, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, msm_accumulate_kernel<FqU, false>, 64, 0);
    }
    ZK_HIP(e);
    return blocks / 4;      // one-wave blocks per CU -> waves per SIMD
}

void msm_g1_enqueue_acc(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const G1AffineU *bases, MsmSlot &slot, int round) { msm_enqueue_acc<FqU>(ctx, ws, plan, bases, slot, round); }
void msm_g2_enqueue_acc(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const G2AffineU *bases, MsmSlot &slot, int round) { msm_enqueue_acc<Fq2U>(ctx, ws, plan, bases, slot, round); }
void msm_g1_enqueue_reduce(zkg16_ctx *ctx, MsmSlot &slot) { msm_enqueue_reduce<FqU>(ctx, slot); }
void msm_g2_enqueue_reduce(zkg16_ctx *ctx, MsmSlot &slot) { msm_enqueue_reduce<Fq2U>(ctx, slot); }
void msm_g1_enqueue(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const G1AffineU *bases, MsmSlot &slot) { msm_enqueue<FqU>(ctx, ws, plan, bases, slot); }
void msm_g2_enqueue(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const G2AffineU *bases, MsmSlot &slot) { msm_enqueue<Fq2U>(ctx, ws, plan, bases, slot); }
G1XYZZ msm_g1_collect(zkg16_ctx *ctx, MsmSlot &slot) { return msm_collect<Fq>(ctx, slot); }
G2XYZZ msm_g2_collect(zkg16_ctx *ctx, MsmSlot &slot) { return msm_collect<Fq2>(ctx, slot); }

G1XYZZ msm_g1_exec(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const G1AffineU *bases, const char *tag) {
    (void)tag;
    msm_enqueue<FqU>(ctx, ws, plan, bases, ctx->slots[0]);
    return msm_collect<Fq>(ctx, ctx->slots[0]);
}
G2XYZZ msm_g2_exec(zkg16_ctx *ctx, MsmWorkspace &ws, const MsmPlan &plan, const G2AffineU *bases, const char *tag) {
    (void)tag;
    msm_enqueue<Fq2U>(ctx, ws, plan, bases, ctx->slots[0]);
    return msm_collect<Fq2>(ctx, ctx->slots[0]);
}
void convert_g1_bases(zkg16_ctx *ctx, const G1Affine *in, G1AffineU *out, size_t n) {
    if (!n) return;
    hipLaunchKernelGGL(convert_bases_kernel<FqU>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, in, out, n);
    ZK_HIP(hipGetLastError());
}
void convert_g2_bases(zkg16_ctx *ctx, const G2Affine *in, G2AffineU *out, size_t n) {
    if (!n) return;
    hipLaunchKernelGGL(convert_bases_kernel<Fq2U>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, in, out, n);
    ZK_HIP(hipGetLastError());
}

template <class FU>
static void batch_affine_run(zkg16_ctx *ctx, const XYZZ<FU> *pts, size_t n, DevBuf &pref, const AffineSegs<FU> &segs) {
    // about one wave per SIMD: the serial inversion (~600 products) is amortised over n / threads points
    size_t threads = (size_t)ctx->num_cus * 4 * 64;
    if (threads * 4 > n) threads = (n + 3) / 4;
    const unsigned blocks = (unsigned)((threads + 63) / 64);
    pref.ensure(n * sizeof(FU));
    hipLaunchKernelGGL(batch_affine_kernel<FU>, dim3(blocks), dim3(64), 0, ctx->stream, pts, n, pref.as<FU>(), segs);
    ZK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ window tables
// A resident key may carry, next to every base P_i, its multiples 2^(c w) P_i for the windows w = 1 .. nwin-1 (affine, same
// unsaturated form): table[w][i].  A proof then needs no per-window bucket sets — digit w of scalar i is a term on base
// table[w][i] of ONE set of 2^(c-1) buckets — so c can grow (22 bits, 12 digits per scalar instead of 15 at 17 bits) at an
// unchanged bucket count, and the five reductions of a proof shrink to one window each.  Cost: nwin x the key in HBM
// (128x128 circuit: 69 GB of the 288 GB) and c * (nwin - 1) doublings per base once, at key-load time (zkg16_pk_precompute).
template <class FU>
__global__ void __launch_bounds__(64, FieldTraits<FU>::g2 ? 1 : 2) table_level_kernel(const Affine<FU> *prev, XYZZ<FU> *out, size_t n, int c) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Affine<FU> p = ldv(prev + i);
    XYZZ<FU> acc = XYZZ<FU>::inf();
    if (!p.is_inf()) {
        acc = xyzz_dbl_affine(p);
        for (int k = 1; k < c; k++) acc = xyzz_dbl(acc);
    }
    stv(out + i, acc);
}
template <class FU>
static DevBuf tables_build(zkg16_ctx *ctx, const DevBuf &bases, size_t n, int c) {
    const int nwin = 254 / c + 1;
    const size_t lvl = n * sizeof(Affine<FU>);
    DevBuf tab((size_t)nwin * lvl), xyzz(n * sizeof(XYZZ<FU>)), pref;
    ZK_HIP(hipMemcpyAsync(tab.p, bases.p, lvl, hipMemcpyDeviceToDevice, ctx->stream));
    Affine<FU> *t = tab.as<Affine<FU>>();
    for (int w = 1; w < nwin; w++) {
        hipLaunchKernelGGL(table_level_kernel<FU>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, t + (size_t)(w - 1) * n,
                           xyzz.as<XYZZ<FU>>(), n, c);
        ZK_HIP(hipGetLastError());
        AffineSegs<FU> seg{};
        seg.n = 1;
        seg.out_u[0] = t + (size_t)w * n;
        batch_affine_run<FU>(ctx, xyzz.as<XYZZ<FU>>(), n, pref, seg);
    }
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return tab;
}
DevBuf msm_tables_build_g1(zkg16_ctx *ctx, const DevBuf &bases, size_t n, int c) { return tables_build<FqU>(ctx, bases, n, c); }
DevBuf msm_tables_build_g2(zkg16_ctx *ctx, const DevBuf &bases, size_t n, int c) { return tables_build<Fq2U>(ctx, bases, n, c); }

// [s_i] base for the n concatenated scalars; results split over `segs`
template <class FU>
static void fixed_base_run(zkg16_ctx *ctx, FixedBaseCache &cache, const Affine<typename FieldTraits<FU>::Sat> &base, const Fr *scalars_canonical,
                           size_t n, const AffineSegs<FU> &segs, bool sync = true) {
    using FS = typename FieldTraits<FU>::Sat;
    if (n == 0) return;
    // 8-bit windows (32 additions per point, an 8,160-entry table) for small batches, 12- and 14-bit ones (22 / 19 additions) once
    // the batch is worth the larger table, and for the batches of a large key 16 / 18 / 20 bits (16 / 15 / 13 additions) built in
    // two levels (fixed_base_combine_kernel).  128x128 key: 51.5 M G1 points at 20 bits (1.5 GB table, 13 instead of 19 additions
    // each: 309 M additions saved for ~20 M spent on the table), 8.7 M G2 points at 18 bits.  Option "fixed_base_bits" forces a width.
    constexpr bool g2 = FieldTraits<FU>::g2;
    int wbits = n >= ((size_t)1 << 17) ? 12 : 8;
    if (!g2) { if (n >= ((size_t)1 << 25)) wbits = 20; else if (n >= ((size_t)1 << 24)) wbits = 18; else if (n >= ((size_t)1 << 21)) wbits = 16; }
    else { if (n >= ((size_t)1 << 23)) wbits = 18; else if (n >= ((size_t)1 << 20)) wbits = 16; }      // (64x64: setup 59.6 -> 54.3 ms with 16 / 16 instead of 18 / 12)
    if (ctx->opt_fixed_base_bits) wbits = ctx->opt_fixed_base_bits;
    const bool two_level = wbits >= 16;             // even widths only (checked by the option)
    const int nwin = (256 + wbits - 1) / wbits;
    const size_t tab_n = (size_t)nwin * (((size_t)1 << wbits) - 1);
    FixedBaseCache::Entry *ent = nullptr;
    for (auto &x : cache.e)
        if (x.wbits == wbits && x.key.size() == sizeof base && memcmp(x.key.data(), &base, sizeof base) == 0) ent = &x;
    const bool miss = ent == nullptr;
    if (miss) {
        ent = cache.e[0].stamp <= cache.e[1].stamp ? &cache.e[0] : &cache.e[1];      // replace the older one
        ent->key.clear();              // the victim matches nothing while its table is rebuilt: a failed build (OOM at large
        ent->wbits = 0;                // sizes) must not leave the old key attached to a half-built table
    }
    ent->stamp = ++cache.clock;
    if (miss) {
        // the ladder-built table: `lb`-bit windows, `lw` of them (the table itself, or the half-width level of a two-level one)
        const int lb = two_level ? wbits / 2 : wbits, lw = two_level ? 2 * nwin : nwin;
        const size_t ltab_n = (size_t)lw * (((size_t)1 << lb) - 1);
        // the window bases 2^(lb w) * base as affine points, one host inversion for all of them
        std::vector<XYZZ<FS>> wx(lw);
        XYZZ<FS> cur = XYZZ<FS>::from_affine(base);
        for (int w = 0; w < lw; w++) {
            wx[w] = cur;
            for (int q = 0; q < lb; q++) cur = xyzz_dbl(cur);
        }
        std::vector<Affine<FS>> wb(lw, Affine<FS>::inf());
        std::vector<FS> pre(lw);
        FS acc = FS::one();
        for (int w = 0; w < lw; w++) {
            pre[w] = acc;
            if (!wx[w].is_inf()) acc = f_mul(acc, f_mul(wx[w].zz, wx[w].zzz));
        }
        FS inv = f_inv(acc);
        for (int w = lw - 1; w >= 0; w--) {
            if (wx[w].is_inf()) continue;
            const FS dinv = f_mul(inv, pre[w]);
            inv = f_mul(inv, f_mul(wx[w].zz, wx[w].zzz));
            wb[w] = Affine<FS>{f_mul(wx[w].x, f_mul(dinv, wx[w].zzz)), f_mul(wx[w].y, f_mul(dinv, wx[w].zz))};
        }
        DevBuf &d_wb = cache.win_bases, &d_xyzz = cache.table_xyzz;      // kept in the cache: nothing here has to outlive a sync
        d_wb.ensure(lw * sizeof(Affine<FS>));
        d_xyzz.ensure((two_level ? tab_n : ltab_n) * sizeof(XYZZ<FU>));
        ent->table.ensure(tab_n * sizeof(Affine<FU>));
        DevBuf half_tab;
        if (two_level) half_tab.alloc(ltab_n * sizeof(Affine<FU>));
        ent->wbits = wbits;
        ZK_HIP(hipMemcpy(d_wb.p, wb.data(), lw * sizeof(Affine<FS>), hipMemcpyHostToDevice));       // 3-6 KB, from a host vector that goes away
        hipLaunchKernelGGL(fixed_base_table_kernel<FU>, dim3((unsigned)((ltab_n + 63) / 64)), dim3(64), 0, ctx->stream, d_wb.as<Affine<FS>>(),
                           d_xyzz.as<XYZZ<FU>>(), lb, lw);
        ZK_HIP(hipGetLastError());
        AffineSegs<FU> ts{};
        ts.n = 1;
        ts.out_u[0] = two_level ? half_tab.as<Affine<FU>>() : ent->table.as<Affine<FU>>();
        batch_affine_run<FU>(ctx, d_xyzz.as<XYZZ<FU>>(), ltab_n, cache.pref, ts);
        if (two_level) {
            hipLaunchKernelGGL(fixed_base_combine_kernel<FU>, dim3((unsigned)((tab_n + 63) / 64)), dim3(64), 0, ctx->stream, half_tab.as<Affine<FU>>(),
                               d_xyzz.as<XYZZ<FU>>(), lb, nwin);
            ZK_HIP(hipGetLastError());
            ts.out_u[0] = ent->table.as<Affine<FU>>();
            batch_affine_run<FU>(ctx, d_xyzz.as<XYZZ<FU>>(), tab_n, cache.pref, ts);
            ZK_HIP(hipStreamSynchronize(ctx->stream));          // half_tab goes out of scope
        }
        ent->key.assign(reinterpret_cast<const uint8_t *>(&base), reinterpret_cast<const uint8_t *>(&base) + sizeof base);
    }
    cache.sums.ensure(n * sizeof(XYZZ<FU>));
    hipLaunchKernelGGL(fixed_base_kernel<FU>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, ent->table.as<Affine<FU>>(),
                       reinterpret_cast<const uint32_t *>(scalars_canonical), n, cache.sums.as<XYZZ<FU>>(), wbits, nwin);
    ZK_HIP(hipGetLastError());
    batch_affine_run<FU>(ctx, cache.sums.as<XYZZ<FU>>(), n, cache.pref, segs);
    if (sync) ZK_HIP(hipStreamSynchronize(ctx->stream));
}
void fixed_base_g1_run(zkg16_ctx *ctx, const G1Affine &base, const Fr *sc, size_t n, G1Affine *out_sat, G1AffineU *out_u) {
    AffineSegs<FqU> s1{};
    s1.n = 1; s1.out_u[0] = out_u; s1.out_sat[0] = out_sat;
    fixed_base_run<FqU>(ctx, ctx->fb_g1, base, sc, n, s1);
}
void fixed_base_g2_run(zkg16_ctx *ctx, const G2Affine &base, const Fr *sc, size_t n, G2Affine *out_sat, G2AffineU *out_u) {
    AffineSegs<Fq2U> s1{};
    s1.n = 1; s1.out_u[0] = out_u; s1.out_sat[0] = out_sat;
    fixed_base_run<Fq2U>(ctx, ctx->fb_g2, base, sc, n, s1);
}
// several scalar ranges, one pass: scalars are the concatenation of the ranges; range k = [start[k], start[k+1]) -> out_*[k]
void fixed_base_g1_multi(zkg16_ctx *ctx, const G1Affine &base, const Fr *sc, size_t n, int nseg, const size_t *start, G1AffineU *const *out_u,
                         G1Affine *const *out_sat, bool sync) {
    AffineSegs<FqU> s{};
    s.n = nseg;
    for (int k = 0; k < nseg; k++) { s.start[k] = start[k]; s.out_u[k] = out_u[k]; s.out_sat[k] = out_sat[k]; }
    fixed_base_run<FqU>(ctx, ctx->fb_g1, base, sc, n, s, sync);
}
void fixed_base_g2_multi(zkg16_ctx *ctx, const G2Affine &base, const Fr *sc, size_t n, int nseg, const size_t *start, G2AffineU *const *out_u,
                         G2Affine *const *out_sat, bool sync) {
    AffineSegs<Fq2U> s{};
    s.n = nseg;
    for (int k = 0; k < nseg; k++) { s.start[k] = start[k]; s.out_u[k] = out_u[k]; s.out_sat[k] = out_sat[k]; }
    fixed_base_run<Fq2U>(ctx, ctx->fb_g2, base, sc, n, s, sync);
}

}  // namespace zk
