// Bucket scatter of an MSM's (scalar, window) digit entries — hand-written for gfx950 (north_star: "wavefront ballot /
// prefix-sum for bucket scatter").  Replaces the rocPRIM onesweep sort of round 1 on the prove path
// (kept behind option "sort_mode" = 1 for A/B; sort.hip).
//
// The digit kernel (msm.hip) leaves one 32-bit code per (window, scalar): (|d| - 1) << 1 | negate, or ~0 for digit 0.
// Within a window the entries must end up ordered by bucket = |d| - 1 (c - 1 bits); windows are independent, so the
// window index is never part of a key.  Stable least-significant-digit passes over the bucket bits: two of <= 10 bits up to
// 20 bucket bits, three of <= 8 bits above (the 21-bit buckets of a key with window tables, msm_tables_build):
//
//   count    one workgroup (4 waves) per 4096 consecutive elements of a window: per-WAVE digit histograms in LDS, no atomics —
//            the lanes holding the same digit find each other with one __ballot per digit bit (wave64 "match"), the lowest
//            of them adds their number to the wave's counter.                            -> counts[window][digit][block]
//   scan     one workgroup per (window, digit): exclusive prefix over the blocks of that digit (wave-wide shuffles)
//                                                                                        -> per-block start of every digit
//   scatter  the same sweep, keeping each element's rank among the equal digits of its wave (popcount of the lower lanes of
//            its ballot group + the wave's counter); the block then orders its chunk by digit in LDS (32 KiB) and writes
//            every digit's run with consecutive lanes on consecutive entries — whole cache lines instead of scattered 8-byte
//            stores (the first version scattered straight from the sweep: 192.8 ms per 128x128 proof against 187.7 with
//            rocPRIM; at 32x32 the two were equal).
//
// Everything is deterministic and stable (original index order inside a bucket): no atomics on positions, so every
// intermediate bucket sum of the accumulation is reproducible run to run.  Zero digits are dropped in the first pass, the
// last pass writes the final entry list compactly over all windows: entries[k] = (index << 1 | negate, global bucket id).
// Traffic per (scalar, window): 4 B digit code written once and read twice, 8 B entry written twice and read twice = 44 B
// against ~64 B for three onesweep passes over 64-bit keys; the arithmetic is ~1 VALU instruction per element per pass.
#include "common.hpp"

namespace zk {

static constexpr int SORT_WAVES = 4;             // waves per workgroup
static constexpr int SORT_ROUNDS = 16;           // 64-element rounds per wave
static constexpr int SORT_WCH = 64 * SORT_ROUNDS;            // elements per wave
static constexpr int SORT_BCH = SORT_WCH * SORT_WAVES;       // elements per workgroup (4096: 32 KiB of staged entries)
static constexpr int SORT_MAX_BITS = 10;         // digit bits per pass

struct SortPass {
    const uint32_t *codes;       // first pass input: digit codes [window][n]
    const uint2 *in;             // later passes' input: [window * n + k], k < in_count[window]
    uint2 *out;
    uint32_t *counts;            // [window][digit][block]  (after `scan`: exclusive prefix over the blocks)
    uint32_t *dig_total;         // [window][digit]
    uint32_t *win_total;         // [window]: elements of this pass's output per window (sort_window_total_kernel)
    const uint32_t *in_count;    // later passes: elements per window in `in` (= the previous pass's win_total); first pass: null (n each)
    size_t n;                    // scalars per window
    uint32_t nblk;               // workgroups per window = ceil(n / SORT_BCH)
    uint32_t nb;                 // buckets per window (the last pass writes global bucket ids)
    int nwin, shift, bits, pass;
};

// lanes of the wave holding the same `dg` as this lane (only among `valid` lanes): one ballot per digit bit
__device__ __forceinline__ uint64_t match_digit(uint32_t dg, bool valid, int bits) {
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < bits; b++) {
        const bool bit = (dg >> b) & 1u;
        const uint64_t m = __ballot(valid && bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {      // set bits of `mask` in lanes lower than this one
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ void wave_sync() {                           // lanes of one wave exchange data through LDS
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// per-wave digit histogram of the block's chunk in hist[wave][digit]; returns through el/dgv/rk the elements this lane
// holds, their digits and their rank among the equal digits of the same WAVE (original index order)
template <bool FIRST, bool KEEP>
__device__ __forceinline__ void sort_sweep(const SortPass &a, uint32_t w, size_t limit, uint32_t *hist, uint2 *el, uint32_t *dgv, uint32_t *rk,
                                            uint32_t &validmask) {
    const uint32_t lane = threadIdx.x & 63u, slot = threadIdx.x >> 6;
    const uint32_t ndig = 1u << a.bits;
    uint32_t *h = hist + slot * ndig;
    for (uint32_t d = lane; d < ndig; d += 64) h[d] = 0;
    wave_sync();
    const size_t k0 = (size_t)blockIdx.x * SORT_BCH + (size_t)slot * SORT_WCH;
    // all of the wave's elements are requested before the first one is ranked: the ranking rounds exchange data through LDS
    // behind compiler barriers, and a load issued inside a round would be waited for in that round (16 dependent round trips
    // to HBM per wave: the scatter of a 201 M-term list took 1.45 ms per pass that way, 2.2 TB/s)
    uint32_t dgs[SORT_ROUNDS], vmask = 0;
    uint2 es[SORT_ROUNDS];
    const uint32_t dmask = (1u << a.bits) - 1u;
    if (FIRST) {
        uint32_t code[SORT_ROUNDS];
#pragma unroll
        for (int r = 0; r < SORT_ROUNDS; r++) {
            const size_t k = k0 + (size_t)r * 64 + lane;
            code[r] = k < limit ? a.codes[(size_t)w * a.n + k] : 0xffffffffu;
        }
#pragma unroll
        for (int r = 0; r < SORT_ROUNDS; r++) {
            const size_t k = k0 + (size_t)r * 64 + lane;
            if (code[r] != 0xffffffffu) vmask |= 1u << r;                 // ~0 = digit 0: no entry
            es[r] = make_uint2(((uint32_t)k << 1) | (code[r] & 1u), code[r] >> 1);
            dgs[r] = (code[r] >> 1) & dmask;
        }
    } else {
#pragma unroll
        for (int r = 0; r < SORT_ROUNDS; r++) {
            const size_t k = k0 + (size_t)r * 64 + lane;
            es[r] = make_uint2(0, 0);
            if (k < limit) {
                es[r] = a.in[(size_t)w * a.n + k];
                vmask |= 1u << r;
            }
        }
#pragma unroll
        for (int r = 0; r < SORT_ROUNDS; r++) dgs[r] = (es[r].y >> a.shift) & dmask;
    }
    validmask = vmask;
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; r++) {
        const uint32_t dg = dgs[r];
        const bool valid = (vmask >> r) & 1u;
        const uint64_t peers = match_digit(dg, valid, a.bits);
        const uint32_t below = lanes_below(peers);
        uint32_t old = 0;
        if (valid) old = h[dg];                                  // every lane of a group reads the counter before its leader bumps it
        wave_sync();
        if (valid && below == 0) h[dg] = old + (uint32_t)__popcll(peers);     // one lane per distinct digit: no atomics
        wave_sync();
        if (KEEP) {
            el[r] = es[r];
            dgv[r] = dg;
            rk[r] = old + below;
        }
    }
}

template <bool FIRST>
__global__ void __launch_bounds__(64 * SORT_WAVES) sort_count_kernel(SortPass a) {
    extern __shared__ uint32_t sort_smem[];
    const uint32_t w = blockIdx.y, ndig = 1u << a.bits;
    const size_t limit = FIRST ? a.n : (size_t)a.in_count[w];
    uint32_t vm;
    sort_sweep<FIRST, false>(a, w, limit, sort_smem, nullptr, nullptr, nullptr, vm);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < ndig; d += blockDim.x) {
        uint32_t t = 0;
        for (int s = 0; s < SORT_WAVES; s++) t += sort_smem[s * ndig + d];
        a.counts[((size_t)w * ndig + d) * a.nblk + blockIdx.x] = t;
    }
}

// one workgroup per (window, digit): counts[w][d][0 .. nblk) -> exclusive prefix; total -> dig_total.  Each of the group's waves
// scans a contiguous share of the row with wave-wide shuffles and the shares are stitched through LDS: a single wave per row
// walked the 49 k blocks of a 201 M-term list in 768 dependent steps (0.45 ms per pass, a fifth of the whole scatter).
__global__ void __launch_bounds__(1024) sort_scan_kernel(SortPass a) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6, d = blockIdx.x, w = blockIdx.y;
    const uint32_t ndig = 1u << a.bits;
    uint32_t *row = a.counts + ((size_t)w * ndig + d) * a.nblk;
    const uint32_t per = ((a.nblk + nwv - 1) / nwv + 63u) & ~63u;          // blocks per wave, a multiple of 64
    const uint32_t lo = wv * per, hi = lo + per < a.nblk ? lo + per : a.nblk;
    uint32_t carry = 0;
    for (uint32_t base = lo; base < hi; base += 64) {                      // pass 1: the share's total
        const uint32_t i = base + lane;
        uint32_t v = i < hi ? row[i] : 0u;
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        carry += v;
    }
    if (lane == 0) wave_tot[wv] = carry;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t x = 0; x < nwv; x++) {
        const uint32_t t = wave_tot[x];
        if (x < wv) before += t;
        total += t;
    }
    carry = before;
    for (uint32_t base = lo; base < hi; base += 64) {                      // pass 2: exclusive prefix inside the share
        const uint32_t i = base + lane;
        const uint32_t v = i < hi ? row[i] : 0u;
        uint32_t inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += up;
        }
        if (i < hi) row[i] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (threadIdx.x == 0) a.dig_total[w * ndig + d] = total;
}

// one wave per window: win_total[w] = sum of the window's digit totals (what the passes use as window sizes / bases)
__global__ void __launch_bounds__(64) sort_window_total_kernel(SortPass a) {
    const uint32_t lane = threadIdx.x, w = blockIdx.x;
    const uint32_t ndig = 1u << a.bits;
    uint32_t s = 0;
    for (uint32_t d = lane; d < ndig; d += 64) s += a.dig_total[w * ndig + d];
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) a.win_total[w] = s;
}

// exclusive scan of one value per thread over the 256 threads of the block (scratch: SORT_WAVES words of LDS); returns the
// exclusive prefix, *total = block total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *scratch, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63u, slot = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o, 64);
        if ((int)lane >= o) inc += up;
    }
    __syncthreads();
    if (lane == 63) scratch[slot] = inc;
    __syncthreads();
    uint32_t before = 0, tot = 0;
    for (uint32_t s = 0; s < SORT_WAVES; s++) {
        const uint32_t x = scratch[s];
        if (s < slot) before += x;
        tot += x;
    }
    *total = tot;
    return before + inc - v;
}

// the block ranks its chunk (sweep), orders it by digit in LDS, and writes every digit's run with consecutive lanes on
// consecutive entries: runs average chunk / 2^bits entries (16 x 8 B at 8 bits), so the stores leave as whole cache lines
template <bool FIRST, bool LAST>
__global__ void __launch_bounds__(64 * SORT_WAVES) sort_scatter_kernel(SortPass a) {
    extern __shared__ uint32_t sort_smem[];
    const uint32_t w = blockIdx.y, ndig = 1u << a.bits, tid = threadIdx.x, slot = tid >> 6;
    uint32_t *hist = sort_smem;                                  // [SORT_WAVES][ndig] -> local start of (wave, digit)
    uint32_t *gdelta = hist + SORT_WAVES * ndig;                 // [ndig]: global position minus local position of a digit's run
    uint32_t *scratch = gdelta + ndig;                           // [2 * SORT_WAVES]
    uint2 *stage = reinterpret_cast<uint2 *>(scratch + 2 * SORT_WAVES);      // [SORT_BCH]
    const size_t limit = FIRST ? a.n : (size_t)a.in_count[w];
    if ((size_t)blockIdx.x * SORT_BCH >= limit) return;          // uniform over the block
    uint2 el[SORT_ROUNDS];
    uint32_t dgv[SORT_ROUNDS], rk[SORT_ROUNDS], vm;
    sort_sweep<FIRST, true>(a, w, limit, hist, el, dgv, rk, vm);
    __syncthreads();
    // window base: the last pass packs the windows (sum of the totals of the windows before), the others keep them apart (w * n)
    size_t win_base = LAST ? 0 : (size_t)w * a.n;
    if (LAST)
        for (uint32_t x = 0; x < w; x++) win_base += a.win_total[x];
    // digits are handled `per` at a time per thread, in digit order, so one block scan gives the local start of every digit
    // (from the block's own counts) and a second one the global start (from the totals of the whole window)
    const uint32_t per = (ndig + blockDim.x - 1) / blockDim.x;
    uint32_t loc = 0, glob = 0;
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t d = tid * per + j;
        if (d < ndig) {
            for (int s = 0; s < SORT_WAVES; s++) loc += hist[s * ndig + d];
            glob += a.dig_total[w * ndig + d];
        }
    }
    uint32_t block_count, dummy;
    uint32_t lrun = block_excl_scan(loc, scratch, &block_count);
    uint32_t grun = block_excl_scan(glob, scratch + SORT_WAVES, &dummy);
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t d = tid * per + j;
        if (d < ndig) {
            gdelta[d] = (uint32_t)win_base + grun + a.counts[((size_t)w * ndig + d) * a.nblk + blockIdx.x] - lrun;
            uint32_t p = lrun;
            for (int s = 0; s < SORT_WAVES; s++) {               // counts -> local start of (wave, digit)
                const uint32_t c = hist[s * ndig + d];
                hist[s * ndig + d] = p;
                p += c;
            }
            lrun = p;
            grun += a.dig_total[w * ndig + d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; r++)
        if (vm & (1u << r)) stage[hist[slot * ndig + dgv[r]] + rk[r]] = el[r];
    __syncthreads();
    const uint32_t mask = ndig - 1u, add = LAST ? w * a.nb : 0u;
    for (uint32_t i = tid; i < block_count; i += blockDim.x) {
        uint2 e = stage[i];
        const uint32_t d = (e.y >> a.shift) & mask;
        e.y += add;                                              // last pass: global bucket id
        a.out[gdelta[d] + i] = e;
    }
}

static size_t sort_lds_bytes(int bits, bool scatter) {
    const size_t ndig = (size_t)1 << bits;
    return scatter ? (SORT_WAVES * ndig + ndig + 2 * SORT_WAVES) * sizeof(uint32_t) + (size_t)SORT_BCH * sizeof(uint2) : SORT_WAVES * ndig * sizeof(uint32_t);
}

// codes: [window][n] digit codes -> entries sorted by (window, bucket), compact over the windows
// returns the device array of per-window entry counts (nwin values; their sum is the length of the list)
const uint32_t *msm_bucket_sort(zkg16_ctx *ctx, MsmWorkspace &ws, const uint32_t *codes, size_t n, int nwin, int c, uint2 *entries) {
    if (n == 0) return nullptr;
    const int bbits = c - 1;                                   // bucket bits per window
    const int npass = bbits <= 2 * SORT_MAX_BITS ? 2 : 3;      // 2 x <= 10 bits; above 20 bucket bits 3 x <= 8
    if (bbits > 24 || n >= ((size_t)1 << 31)) throw HipError{hipErrorInvalidValue, "bucket sort: window bits / length above build limit", __FILE__, __LINE__};
    int bits[3] = {0, 0, 0};
    for (int p = 0; p < npass; p++) bits[p] = bbits / npass + (p < bbits % npass ? 1 : 0);
    const uint32_t nblk = (uint32_t)((n + SORT_BCH - 1) / SORT_BCH);
    const size_t ndig_max = (size_t)1 << bits[0];
    const size_t counts_words = (size_t)nwin * ndig_max * nblk;
    const size_t small = (size_t)nwin * ndig_max + 3 * (size_t)nwin + 16;
    ws.sort_temp.ensure((counts_words + small) * sizeof(uint32_t));
    ws.keys.ensure((size_t)nwin * n * sizeof(uint2));           // intermediate entry list (window w at [w * n, ...))
    if (npass == 3) ws.stage.ensure((size_t)nwin * n * sizeof(uint2));      // second intermediate list
    uint32_t *counts = ws.sort_temp.as<uint32_t>();
    uint32_t *dig_total = counts + counts_words;
    uint32_t *win_tot = dig_total + (size_t)nwin * ndig_max;     // [pass][window]
    SortPass a{};
    a.codes = codes;
    a.counts = counts;
    a.dig_total = dig_total;
    a.n = n;
    a.nblk = nblk;
    a.nb = (uint32_t)1 << bbits;
    a.nwin = nwin;
    const dim3 grid(nblk, (unsigned)nwin), block(64 * SORT_WAVES);
    const unsigned scan_waves = nblk <= 256 ? 1u : nblk >= 4096 ? 16u : (nblk + 255u) / 256u;      // >= 4 steps of 64 blocks per wave
    ScopedKernelTimer kt(ctx, "msm_bucket_sort", (double)n * nwin, ctx->stream);
    int shift = 0;
    for (int p = 0; p < npass; p++) {
        const bool first = p == 0, last = p == npass - 1;
        a.pass = p; a.shift = shift; a.bits = bits[p];
        a.in = first ? nullptr : (p == 1 ? ws.keys.as<uint2>() : ws.stage.as<uint2>());
        a.in_count = first ? nullptr : win_tot + (size_t)(p - 1) * nwin;
        a.out = last ? entries : (p == 0 ? ws.keys.as<uint2>() : ws.stage.as<uint2>());
        a.win_total = win_tot + (size_t)p * nwin;
        const size_t lds_c = sort_lds_bytes(bits[p], false), lds_s = sort_lds_bytes(bits[p], true);
        if (first) hipLaunchKernelGGL(sort_count_kernel<true>, grid, block, lds_c, ctx->stream, a);
        else hipLaunchKernelGGL(sort_count_kernel<false>, grid, block, lds_c, ctx->stream, a);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(1u << bits[p], (unsigned)nwin), dim3(64 * scan_waves), 0, ctx->stream, a);
        hipLaunchKernelGGL(sort_window_total_kernel, dim3((unsigned)nwin), dim3(64), 0, ctx->stream, a);
        if (first) hipLaunchKernelGGL((sort_scatter_kernel<true, false>), grid, block, lds_s, ctx->stream, a);
        else if (last) hipLaunchKernelGGL((sort_scatter_kernel<false, true>), grid, block, lds_s, ctx->stream, a);
        else hipLaunchKernelGGL((sort_scatter_kernel<false, false>), grid, block, lds_s, ctx->stream, a);
        shift += bits[p];
    }
    ZK_HIP(hipGetLastError());
    return win_tot + (size_t)(npass - 1) * nwin;
}

}  // namespace zk
