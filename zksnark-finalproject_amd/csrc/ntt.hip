// BLS12-381 Fr radix-2 NTT for gfx950 — the seven transforms of the R1CS->QAP witness map
// (ark-poly 0.4.2 Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place as driven by
// ark-groth16 r1cs_to_qap.rs; reached from /root/reference/src/arkworks/backend/matrix_proof.rs:139-140).
//
// Shape: a "four-step" split N = N1 * N2 so that every butterfly stage runs out of LDS and HBM is touched
// exactly twice per element per transform (once for N <= 2048):
//   pass 1 (cols): for each group of C adjacent columns i2, load x[N2*i1 + i2] (C*32 B contiguous segments),
//                  N1-point DIF in LDS, multiply by the inter-pass twiddle w_N^(i2*k1), store in place.
//   pass 2 (rows): for each group of R adjacent rows k1, load N2 contiguous elements, N2-point DIF in LDS,
//                  write X[k1 + N1*k2] (R*32 B contiguous segments) to the output buffer.
// Natural order in, natural order out: the bit reversal of the DIF is absorbed by the LDS read-out.
// LDS tile = 2048 elements, limb-major (SoA) with an XOR bank swizzle so that both the stride-h butterfly
// accesses and the bit-reversed read-out are conflict-free; the sub-transform's twiddles (w_M^e, e < M/2) are
// staged once per workgroup in LDS.  No MFMA: there is no dense contraction in a modular butterfly.
// Coset shifts (g = 7) and the 1/N of the inverse ride on the load of pass 1 / the store of pass 2.
#include "common.hpp"
#include "fru.cuh"

namespace zk {

static constexpr int NTT_TILE_LOG = 11;
static constexpr int NTT_TILE = 1 << NTT_TILE_LOG;   // elements per workgroup tile (64 KiB of LDS)
static constexpr int NTT_THREADS = 256;
static constexpr int NTT_MAX_SUB_LOG = 11;           // a tile may hold one whole 2048-point sub-transform

__device__ __forceinline__ int swz(int e) { return e ^ ((e >> 5) & 31); }

__device__ __forceinline__ Fr lds_ld(const uint32_t *s, int stride, int e) {
    Fr v;
    const int p = swz(e);
#pragma unroll
    for (int k = 0; k < 8; k++) v.l[k] = s[k * stride + p];
    return v;
}
__device__ __forceinline__ void lds_st(uint32_t *s, int stride, int e, const Fr &v) {
    const int p = swz(e);
#pragma unroll
    for (int k = 0; k < 8; k++) s[k * stride + p] = v.l[k];
}

__device__ __forceinline__ Fr gld(const Fr *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fr v;
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
    v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
    return v;
}
__device__ __forceinline__ void gst(Fr *p, const Fr &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, each XCD has its own L2).  A column tile
// touches 32 * C contiguous bytes per row, a row tile writes 32 * R: with C (R) = 1 or 2 a 128-byte line is shared by 4 (2)
// neighbouring tiles.  Give those neighbours to consecutive blocks of ONE XCD, so the line is fetched / merged once in that
// XCD's L2 instead of once per XCD (measured at 2^24, C = 1: FETCH_SIZE of the column pass 2.79 GB against 0.54 GB of data).
__device__ __forceinline__ unsigned xcd_tile(unsigned b, unsigned nblocks, unsigned group) {
    if (group <= 1 || (nblocks % (8u * group)) != 0) return b;
    const unsigned xcd = b & 7u, s = b >> 3;
    return ((s / group) * 8u + xcd) * group + (s % group);
}

__device__ __forceinline__ int bitrev(int x, int bits) { return bits == 0 ? 0 : (int)(__brev((unsigned)x) >> (32 - bits)); }

struct NttPassArgs {
    const Fr *in;
    Fr *out;
    const Fr *w;       // w_N^j table
    const uint32_t *wu; // the same table in the unsaturated form (9 x u32 per entry), only for the global-twiddle kernels
    const Fr *pre;     // optional per-input-index multiplier (coset fft), else null
    // optional fused point-wise stage of the witness map on the FIRST pass's load (pre is null then):
    //   x[i] <- (in[i] * pw_b[i] - pw_c[i]) / Z   — ark-groth16 r1cs_to_qap.rs: ab = (a*b - c) * Z(g)^-1 before the coset ifft
    const Fr *pw_b, *pw_c;
    Fr pw_zinv;        // saturated kernels: 1/Z (Montgomery)
    FrU pw_zc;         // unsaturated kernels: 1/Z * 2^271 mod r, repacked (see load_u)
    const Fr *post;    // optional per-output-index multiplier (coset ifft), else null
    Fr post_const;     // used when post_const_on (plain ifft: N^-1)
    int post_const_on;
    int log_n, log_n1, log_n2;   // log_n = size of the w table (whole transform); this pass splits a 2^(log_n1+log_n2) sub-transform
    int inverse;
    // three-pass transforms (N > 2^22): passes 2 and 3 run batched over blockIdx.y = k0 on the rows of the outer split
    int tw_shift;                // inter-pass twiddle index is (i2*k1) << tw_shift   (w_M = w_N^(2^tw_shift))
    size_t batch_stride;         // elements between consecutive batches in `in` (and in `out` for in-place passes)
    int out_stride_log;          // rows pass: final index = batch + (k << out_stride_log)
    int xcd_order;               // 1: neighbouring tiles go to consecutive blocks of one XCD (xcd_tile); option "ntt_xcd"
};

// DIF over `ncols` independent sub-transforms of size 2^log_m laid out back to back in the LDS tile.
__device__ __forceinline__ void lds_dif(uint32_t *s_data, const uint32_t *s_tw, int log_m, int tw_stride) {
    const int tid = threadIdx.x;
    for (int s = log_m - 1; s >= 0; s--) {
        const int h = 1 << s;
        for (int u = tid; u < NTT_TILE / 2; u += NTT_THREADS) {
            const int c = u >> (log_m - 1);
            const int v = u & ((1 << (log_m - 1)) - 1);
            const int j = v & (h - 1);
            const int blk = v >> s;
            const int i0 = (c << log_m) + (blk << (s + 1)) + j;
            const int i1 = i0 + h;
            Fr a = lds_ld(s_data, NTT_TILE, i0);
            Fr b = lds_ld(s_data, NTT_TILE, i1);
            Fr sum = fp_add(a, b);
            Fr dif = fp_sub(a, b);
            if (s > 0) {   // the last stage's twiddle is w^0 = 1
                Fr tw = lds_ld(s_tw, tw_stride, j << (log_m - 1 - s));
                dif = fp_mul(dif, tw);
            }
            lds_st(s_data, NTT_TILE, i0, sum);
            lds_st(s_data, NTT_TILE, i1, dif);
        }
        __syncthreads();
    }
}

// twiddles of the 2^log_m sub-transform: w_M^e = w_N^(e * N/M), e < M/2 (inverse: w_N^-(...) = w[N - ...])
__device__ __forceinline__ void stage_twiddles(uint32_t *s_tw, int tw_stride, const Fr *w, int log_n, int log_m, int inverse) {
    const int half = 1 << (log_m > 0 ? log_m - 1 : 0);
    const unsigned nmask = (1u << log_n) - 1u;
    for (int e = threadIdx.x; e < half; e += NTT_THREADS) {
        unsigned idx = (unsigned)e << (log_n - log_m);
        if (inverse) idx = ((1u << log_n) - idx) & nmask;
        lds_st(s_tw, tw_stride, e, gld(w + idx));
    }
}

// pass 1: column groups.  grid = N2 / C, C = TILE >> log_n1.
__global__ void __launch_bounds__(NTT_THREADS) ntt_pass_cols(NttPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *s_data = smem;                         // [8][TILE]
    uint32_t *s_tw = smem + 8 * NTT_TILE;            // [8][tw_stride]
    const int tw_stride = 1 << (a.log_n1 - 1);
    const int n1 = 1 << a.log_n1;
    const int log_c = NTT_TILE_LOG - a.log_n1;
    const int C = 1 << log_c;
    const unsigned nmask = (1u << a.log_n) - 1u;
    const size_t n2 = (size_t)1 << a.log_n2;
    const size_t col0 = (size_t)blockIdx.x * C;
    const Fr *in = a.in + (size_t)blockIdx.y * a.batch_stride;
    Fr *out = a.out + (size_t)blockIdx.y * a.batch_stride;

    stage_twiddles(s_tw, tw_stride, a.w, a.log_n, a.log_n1, a.inverse);
    for (int t = threadIdx.x; t < NTT_TILE; t += NTT_THREADS) {
        const int c = t & (C - 1), i1 = t >> log_c;
        const size_t gi = (size_t)i1 * n2 + col0 + c;
        Fr v = gld(in + gi);
        if (a.pre) v = fp_mul(v, gld(a.pre + gi));
        else if (a.pw_b) v = fp_mul(fp_sub(fp_mul(v, gld(a.pw_b + gi)), gld(a.pw_c + gi)), a.pw_zinv);
        lds_st(s_data, NTT_TILE, (c << a.log_n1) + i1, v);
    }
    __syncthreads();
    lds_dif(s_data, s_tw, a.log_n1, tw_stride);
    for (int t = threadIdx.x; t < NTT_TILE; t += NTT_THREADS) {
        const int c = t & (C - 1), k1 = t >> log_c;
        Fr v = lds_ld(s_data, NTT_TILE, (c << a.log_n1) + bitrev(k1, a.log_n1));
        const size_t i2 = col0 + c;
        unsigned e = (unsigned)(((i2 * (size_t)k1) << a.tw_shift) & nmask);    // inter-pass twiddle w_M^(i2*k1), w_M = w_N^(2^tw_shift)
        if (a.inverse) e = ((1u << a.log_n) - e) & nmask;
        if (e) v = fp_mul(v, gld(a.w + e));
        gst(out + (size_t)k1 * n2 + i2, v);
    }
    (void)n1;
}

// pass 2: row groups.  grid = N1 / R, R = TILE >> log_n2.  Also the whole transform when log_n1 == 0.
__global__ void __launch_bounds__(NTT_THREADS) ntt_pass_rows(NttPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *s_data = smem;
    uint32_t *s_tw = smem + 8 * NTT_TILE;
    const int tw_stride = a.log_n2 > 0 ? 1 << (a.log_n2 - 1) : 1;
    const int log_r = NTT_TILE_LOG - a.log_n2;
    const int R = 1 << log_r;
    const size_t n1 = (size_t)1 << a.log_n1;
    const size_t n2 = (size_t)1 << a.log_n2;
    const size_t row0 = (size_t)blockIdx.x * R;
    const size_t n1_rows = n1;                        // rows that exist (R may exceed N1 for tiny transforms)
    const Fr *in = a.in + (size_t)blockIdx.y * a.batch_stride;

    stage_twiddles(s_tw, tw_stride, a.w, a.log_n, a.log_n2, a.inverse);
    for (int t = threadIdx.x; t < NTT_TILE; t += NTT_THREADS) {
        const int i2 = t & ((1 << a.log_n2) - 1), r = t >> a.log_n2;
        Fr v = Fr::zero();
        if (row0 + r < n1_rows) {
            const size_t gi = (row0 + r) * n2 + i2;
            v = gld(in + gi);
            if (a.pre) v = fp_mul(v, gld(a.pre + gi));
            else if (a.pw_b) v = fp_mul(fp_sub(fp_mul(v, gld(a.pw_b + gi)), gld(a.pw_c + gi)), a.pw_zinv);
        }
        lds_st(s_data, NTT_TILE, t, v);
    }
    __syncthreads();
    lds_dif(s_data, s_tw, a.log_n2, tw_stride);
    for (int t = threadIdx.x; t < NTT_TILE; t += NTT_THREADS) {
        const int r = t & (R - 1), k2 = t >> log_r;
        if (row0 + r >= n1_rows) continue;
        Fr v = lds_ld(s_data, NTT_TILE, (r << a.log_n2) + bitrev(k2, a.log_n2));
        const size_t k = (size_t)blockIdx.y + (((row0 + r) + n1 * (size_t)k2) << a.out_stride_log);
        if (a.post) v = fp_mul(v, gld(a.post + k));
        else if (a.post_const_on) v = fp_mul(v, a.post_const);
        gst(a.out + k, v);
    }
}

// ------------------------------------------------------------------------------------------------ unsaturated butterflies
// The same two passes with the arithmetic in FrU (fru.cuh): LDS holds 9 x 29-bit limbs per element; global memory keeps
// the saturated form, converted by the multiplication each load / store performs anyway (see fru.cuh).  Butterfly values
// stay below 2r: the sum gets a conditional subtraction of 2r, the difference (a - b + 2r < 4r) goes straight into the
// twiddle product.
static constexpr int NTT_THREADS_U = 1024;          // 4 waves per SIMD over one 2048-element tile: one butterfly per thread per stage
__device__ __forceinline__ FrU lds_ld_u(const uint32_t *s, int stride, int e) {
    FrU v;
    const int p = swz(e);
#pragma unroll
    for (int k = 0; k < 9; k++) v.l[k] = s[k * stride + p];
    return v;
}
__device__ __forceinline__ void lds_st_u(uint32_t *s, int stride, int e, const FrU &v) {
    const int p = swz(e);
#pragma unroll
    for (int k = 0; k < 9; k++) s[k * stride + p] = v.l[k];
}

// TL = log2 of the tile (11: twiddles of the sub-transform staged in LDS; 12: the tile fills the LDS, twiddles come from a
// U-form table in global memory — used above 2^22, where it makes the transform two passes instead of three)
template <int TL, bool GTW>
__device__ __forceinline__ FrU tw_u(const uint32_t *s_tw, int tw_stride, int e, int log_m, const NttPassArgs &a) {
    // twiddle w_M^e of the 2^log_m sub-transform, e < M/2: staged in LDS, or (GTW) read from the U-form table in global memory
    if (GTW) {
        const unsigned nmask = (1u << a.log_n) - 1u;
        unsigned idx = (unsigned)e << (a.log_n - log_m);
        if (a.inverse) idx = ((1u << a.log_n) - idx) & nmask;
        const uint32_t *t = a.wu + (size_t)idx * 9;
        FrU tw;
#pragma unroll
        for (int k = 0; k < 9; k++) tw.l[k] = t[k];
        return tw;
    }
    return lds_ld_u(s_tw, tw_stride, e);
}
// decimation-in-frequency butterfly on (x, y): x <- x + y (lazily reduced: < 2r + e, fru_add_lazy), y <- (x - y + 4r) * tw (< 2r);
// last stage (tw = 1): y <- x - y + 4r as it is (< 6r + e, not normalised) — the store multiplies it by the pass's factor anyway.
// (Until late in round 2: exact conditional subtraction of 2r after the sum and a normalised difference — 113 + 33 instructions
// against 45 + 9; 2^24: 3.06 -> 2.94 ms per transform.  Keeping the 256 twiddles of the last nine stages of the 4096-point tiles in
// the LDS left beside the tile instead of reading the global table: no change, 2.95 ms.)
template <bool LAST>
__device__ __forceinline__ void bfly_u(FrU &x, FrU &y, const FrU &tw) {
    const FrU sum = fru_add_lazy(x, y);
    const FrU dif = fru_sub_4r_raw(x, y);
    y = LAST ? dif : fru_mul(dif, tw);
    x = sum;
}

// TL = log2 of the tile (11: twiddles of the sub-transform staged in LDS; 12: the tile fills the LDS, twiddles come from a
// U-form table in global memory — used above 2^22, where it makes the transform two passes instead of three).
// R4 (option "ntt_radix" = 4): two butterfly stages per trip through the LDS — a thread holds the four elements i, i + h/2,
// i + h, i + 3h/2 of a stage pair in registers, so the tile is read and written log_m / 2 times and the barriers halve; the
// multiplications are the same four per group.  Measured: equal to one stage per trip up to 2^22, SLOWER at 2^24 (3.44 vs
// 3.07 ms per transform: the kernel is multiplier-bound, and the four live elements cost registers), so it is not the default.
// MODE 2 (option "ntt_radix" = 1): the last (up to) seven stages without the LDS — a thread keeps its pair in registers and, between
// two stages, swaps ONE element with the lane that holds the partner of the next stage (lane ^ 2^(s-1): ds_bpermute moves
// registers between lanes of a wave, no LDS memory, no barrier); the stages above stay as they are.  
// the value of lane ^ MASK: DPP quad permutes for 1 and 2 (a VALU move), ds_swizzle for 4 / 8 / 16 (no address register),
// ds_bpermute for 32
template <int MASK>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MASK == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
    else if constexpr (MASK == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
    else if constexpr (MASK == 4 || MASK == 8 || MASK == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (MASK << 10) | 0x1F);
    else return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ MASK) << 2, (int)v);
#else
    (void)lane;
    return v;
#endif
}
template <int MASK>
__device__ __forceinline__ FrU lane_xchg_u(const FrU &v, int lane) {
    FrU r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.l[k] = lane_xor_u32<MASK>(v.l[k], lane);
    return r;
}
// one in-register stage ST of the chain below: the butterfly, then (ST > 0) the exchange that forms the pairs of stage ST - 1
template <int TL, bool GTW, int ST>
__device__ __forceinline__ void chain_stage_u(FrU &x, FrU &y, int &ex, int lane, const uint32_t *s_tw, int tw_stride, int log_m, const NttPassArgs &a) {
    if constexpr (ST > 0) {
        const int j = ex & ((1 << ST) - 1);
        FrU tw;
        if constexpr (GTW) {           // the 64 twiddles the chain can ask for (w_M^(j' 2^(log_m - 7))) were staged in LDS: [9][64]
            const int jj = j << (6 - ST);
#pragma unroll
            for (int k = 0; k < 9; k++) tw.l[k] = s_tw[k * 64 + jj];
        } else {
            tw = tw_u<TL, GTW>(s_tw, tw_stride, j << (log_m - 1 - ST), log_m, a);
        }
        bfly_u<false>(x, y, tw);
        // next stage pairs elements that differ in bit ST - 1: the lane with that bit clear keeps x and takes the partner's x as
        // its y; the lane with it set keeps y and takes the partner's y as its x
        const bool hi = (lane >> (ST - 1)) & 1;
        FrU send;
#pragma unroll
        for (int k = 0; k < 9; k++) send.l[k] = hi ? x.l[k] : y.l[k];
        const FrU recv = lane_xchg_u<(1 << (ST - 1))>(send, lane);
#pragma unroll
        for (int k = 0; k < 9; k++) {
            if (hi) x.l[k] = recv.l[k]; else y.l[k] = recv.l[k];
        }
        if (hi) ex += 1 << (ST - 1);
    } else {
        bfly_u<true>(x, y, x);
    }
}

// stage Q (0 .. 6) of the UPPER chain (MODE 3): stage st = log_m - 1 - Q of the sub-transform; the lanes of a wave hold the pairs whose
// index bits log_m - 2 .. log_m - 7 spell the lane number, so the partner of the next stage is lane ^ (32 >> Q)
template <int TL, bool GTW, int Q>
__device__ __forceinline__ void chain_stage_hi_u(FrU &x, FrU &y, int &ex, int lane, const uint32_t *s_tw, int tw_stride, int log_m, const NttPassArgs &a) {
    const int st = log_m - 1 - Q;
    const int j = ex & ((1 << st) - 1);
    bfly_u<false>(x, y, tw_u<TL, GTW>(s_tw, tw_stride, j << Q, log_m, a));
    if constexpr (Q < 6) {
        const bool hi = (lane >> (5 - Q)) & 1;
        FrU send;
#pragma unroll
        for (int k = 0; k < 9; k++) send.l[k] = hi ? x.l[k] : y.l[k];
        const FrU recv = lane_xchg_u<(32 >> Q)>(send, lane);
#pragma unroll
        for (int k = 0; k < 9; k++) {
            if (hi) x.l[k] = recv.l[k]; else y.l[k] = recv.l[k];
        }
        if (hi) ex += 1 << (st - 1);
    }
}

template <int TL, bool GTW, int R4>
__device__ __forceinline__ void lds_dif_u(uint32_t *s_data, const uint32_t *s_tw, int log_m, int tw_stride, const NttPassArgs &a) {
    constexpr int TILE = 1 << TL;
    const int tid = threadIdx.x;
    int s = log_m - 1;
    if (R4 == 2 || R4 == 3) {
        if (log_m < 1) return;                                   // a one-point sub-transform: nothing to do
        int s_first = log_m - 1 < 6 ? log_m - 1 : 6;            // stages s_first .. 0 in registers
        if (R4 == 3 && log_m >= 8 && log_m <= 14) {
            // MODE 3: the TOP seven stages in registers as well (pairs dealt to the lanes by their upper index bits), ONE trip through
            // the LDS, then the remaining log_m - 7 stages by the chain below
            if (GTW && tid < 64) {
                uint32_t *tws = const_cast<uint32_t *>(s_tw);
                const unsigned nmask = (1u << a.log_n) - 1u;
                unsigned idx = ((unsigned)tid << (log_m - 7)) << (a.log_n - log_m);
                if (a.inverse) idx = ((1u << a.log_n) - idx) & nmask;
                const uint32_t *t = a.wu + (size_t)idx * 9;
#pragma unroll
                for (int k = 0; k < 9; k++) tws[k * 64 + tid] = t[k];
            }
            const int lane_a = tid & 63, lo_bits = log_m - 7;
            for (int u = tid; u < TILE / 2; u += NTT_THREADS_U) {
                const int rest = u >> 6;
                const int c = rest >> lo_bits, p_lo = rest & ((1 << lo_bits) - 1);
                int ex = (c << log_m) + (lane_a << lo_bits) + p_lo;
                FrU x = lds_ld_u(s_data, TILE, ex), y = lds_ld_u(s_data, TILE, ex + (1 << (log_m - 1)));
                chain_stage_hi_u<TL, GTW, 0>(x, y, ex, lane_a, s_tw, tw_stride, log_m, a);
                chain_stage_hi_u<TL, GTW, 1>(x, y, ex, lane_a, s_tw, tw_stride, log_m, a);
                chain_stage_hi_u<TL, GTW, 2>(x, y, ex, lane_a, s_tw, tw_stride, log_m, a);
                chain_stage_hi_u<TL, GTW, 3>(x, y, ex, lane_a, s_tw, tw_stride, log_m, a);
                chain_stage_hi_u<TL, GTW, 4>(x, y, ex, lane_a, s_tw, tw_stride, log_m, a);
                chain_stage_hi_u<TL, GTW, 5>(x, y, ex, lane_a, s_tw, tw_stride, log_m, a);
                chain_stage_hi_u<TL, GTW, 6>(x, y, ex, lane_a, s_tw, tw_stride, log_m, a);
                lds_st_u(s_data, TILE, ex, x);
                lds_st_u(s_data, TILE, ex + (1 << (log_m - 7)), y);
            }
            __syncthreads();
            s = log_m - 8;
            s_first = s;                                         // (<= 6) the rest is the lower chain, no LDS stage in between
        }
        if (GTW && tid < 64) {          // (log_m >= 11 on this path) the chain's twiddles, read once per block instead of once per butterfly
            uint32_t *tws = const_cast<uint32_t *>(s_tw);
            const unsigned nmask = (1u << a.log_n) - 1u;
            unsigned idx = ((unsigned)tid << (log_m - 7)) << (a.log_n - log_m);
            if (a.inverse) idx = ((1u << a.log_n) - idx) & nmask;
            const uint32_t *t = a.wu + (size_t)idx * 9;
#pragma unroll
            for (int k = 0; k < 9; k++) tws[k * 64 + tid] = t[k];
        }
        for (; s > s_first; s--) {                               // the stages above: through the LDS, as MODE 0
            const int h = 1 << s;
            for (int u = tid; u < TILE / 2; u += NTT_THREADS_U) {
                const int c = u >> (log_m - 1);
                const int v = u & ((1 << (log_m - 1)) - 1);
                const int j = v & (h - 1);
                const int blk = v >> s;
                const int i0 = (c << log_m) + (blk << (s + 1)) + j;
                FrU x = lds_ld_u(s_data, TILE, i0), y = lds_ld_u(s_data, TILE, i0 + h);
                bfly_u<false>(x, y, tw_u<TL, GTW>(s_tw, tw_stride, j << (log_m - 1 - s), log_m, a));
                lds_st_u(s_data, TILE, i0, x);
                lds_st_u(s_data, TILE, i0 + h, y);
            }
            __syncthreads();
        }
        const int lane = tid & 63;
        for (int u = tid; u < TILE / 2; u += NTT_THREADS_U) {
            const int c = u >> (log_m - 1);
            const int v = u & ((1 << (log_m - 1)) - 1);
            // the pair of stage s_first: bit s_first of the element index clear / set, the other bits from v
            int ex = (c << log_m) + ((v >> s_first) << (s_first + 1)) + (v & ((1 << s_first) - 1));
            FrU x = lds_ld_u(s_data, TILE, ex), y = lds_ld_u(s_data, TILE, ex + (1 << s_first));
            if (s_first >= 6) chain_stage_u<TL, GTW, 6>(x, y, ex, lane, s_tw, tw_stride, log_m, a);
            if (s_first >= 5) chain_stage_u<TL, GTW, 5>(x, y, ex, lane, s_tw, tw_stride, log_m, a);
            if (s_first >= 4) chain_stage_u<TL, GTW, 4>(x, y, ex, lane, s_tw, tw_stride, log_m, a);
            if (s_first >= 3) chain_stage_u<TL, GTW, 3>(x, y, ex, lane, s_tw, tw_stride, log_m, a);
            if (s_first >= 2) chain_stage_u<TL, GTW, 2>(x, y, ex, lane, s_tw, tw_stride, log_m, a);
            if (s_first >= 1) chain_stage_u<TL, GTW, 1>(x, y, ex, lane, s_tw, tw_stride, log_m, a);
            chain_stage_u<TL, GTW, 0>(x, y, ex, lane, s_tw, tw_stride, log_m, a);
            lds_st_u(s_data, TILE, ex, x);
            lds_st_u(s_data, TILE, ex + 1, y);
        }
        __syncthreads();
        return;
    }
    if (!R4) {                                  // one stage per trip (default)
        for (; s >= 0; s--) {
            const int h = 1 << s;
            for (int u = tid; u < TILE / 2; u += NTT_THREADS_U) {
                const int c = u >> (log_m - 1);
                const int v = u & ((1 << (log_m - 1)) - 1);
                const int j = v & (h - 1);
                const int blk = v >> s;
                const int i0 = (c << log_m) + (blk << (s + 1)) + j;
                FrU x = lds_ld_u(s_data, TILE, i0), y = lds_ld_u(s_data, TILE, i0 + h);
                if (s > 0) bfly_u<false>(x, y, tw_u<TL, GTW>(s_tw, tw_stride, j << (log_m - 1 - s), log_m, a));
                else bfly_u<true>(x, y, x);
                lds_st_u(s_data, TILE, i0, x);
                lds_st_u(s_data, TILE, i0 + h, y);
            }
            __syncthreads();
        }
        return;
    }
    for (; s >= 1; s -= 2) {
        const int h = 1 << s, q = h >> 1;
        for (int u = tid; u < TILE / 4; u += NTT_THREADS_U) {
            const int c = u >> (log_m - 2);
            const int v = u & ((1 << (log_m - 2)) - 1);
            const int j = v & (q - 1);
            const int blk = v >> (s - 1);
            const int i0 = (c << log_m) + (blk << (s + 1)) + j;
            FrU x0 = lds_ld_u(s_data, TILE, i0), x1 = lds_ld_u(s_data, TILE, i0 + q);
            FrU x2 = lds_ld_u(s_data, TILE, i0 + h), x3 = lds_ld_u(s_data, TILE, i0 + h + q);
            // stage s (half h): (x0, x2) with w^j, (x1, x3) with w^(j + q)
            bfly_u<false>(x0, x2, tw_u<TL, GTW>(s_tw, tw_stride, j << (log_m - 1 - s), log_m, a));
            bfly_u<false>(x1, x3, tw_u<TL, GTW>(s_tw, tw_stride, (j + q) << (log_m - 1 - s), log_m, a));
            // stage s - 1 (half q): (x0, x1) and (x2, x3), both with the same twiddle w'^j
            if (s > 1) {
                const FrU t1 = tw_u<TL, GTW>(s_tw, tw_stride, j << (log_m - s), log_m, a);
                bfly_u<false>(x0, x1, t1);
                bfly_u<false>(x2, x3, t1);
            } else {
                bfly_u<true>(x0, x1, x0);
                bfly_u<true>(x2, x3, x0);
            }
            lds_st_u(s_data, TILE, i0, x0);
            lds_st_u(s_data, TILE, i0 + q, x1);
            lds_st_u(s_data, TILE, i0 + h, x2);
            lds_st_u(s_data, TILE, i0 + h + q, x3);
        }
        __syncthreads();
    }
    if (s == 0) {                               // odd number of stages: the last one alone (twiddle 1)
        for (int u = tid; u < TILE / 2; u += NTT_THREADS_U) {
            const int i0 = u << 1;
            FrU x = lds_ld_u(s_data, TILE, i0), y = lds_ld_u(s_data, TILE, i0 + 1);
            bfly_u<true>(x, y, x);
            lds_st_u(s_data, TILE, i0, x);
            lds_st_u(s_data, TILE, i0 + 1, y);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void stage_twiddles_u(uint32_t *s_tw, int tw_stride, const Fr *w, int log_n, int log_m, int inverse) {
    const int half = 1 << (log_m > 0 ? log_m - 1 : 0);
    const unsigned nmask = (1u << log_n) - 1u;
    for (int e = threadIdx.x; e < half; e += NTT_THREADS_U) {
        unsigned idx = (unsigned)e << (log_n - log_m);
        if (inverse) idx = ((1u << log_n) - idx) & nmask;
        lds_st_u(s_tw, tw_stride, e, fru_from_sat(gld(w + idx)));
    }
}

// element load.  The stored (saturated Montgomery) limbs x 2^256 are used AS the U-form of x 2^-5: no conversion product.
// The transform is linear, so every pass that loads this way leaves a factor 2^-5 behind, and the multiplier of the LAST
// store carries 2^(5 * passes) (ntt_run).  The coset pre-multiply uses a table scaled by 2^10, which makes its one product a
// full conversion (X 2^256 * g 2^10 2^256 * 2^-261 = x g 2^261).  Fused point-wise stage (x * b - c) / Z: X (*) B = xb 2^251,
// C (*) 2^256 = c 2^251, their difference (+2r) times zc = 2^266 / Z is (xb - c)/Z * 2^256: like a plain load   ((*) = fru_mul).
__device__ __forceinline__ FrU load_u(const NttPassArgs &a, const Fr *in, size_t gi) {
    const FrU x = fru_repack(gld(in + gi));
    if (a.pre) return fru_mul(x, fru_repack(gld(a.pre + gi)));
    if (a.pw_b) {
        const FrU xb = fru_mul(x, fru_repack(gld(a.pw_b + gi)));
        const FrU c = fru_mul(fru_repack(gld(a.pw_c + gi)), fru_one_sat());
        return fru_mul(fru_sub_2r(xb, c), a.pw_zc);
    }
    return x;                                 // canonical (< r), limbs < 2^29: a valid butterfly operand
}

template <int TL, bool GTW, int R4>
__global__ void __launch_bounds__(NTT_THREADS_U) ntt_pass_cols_u(NttPassArgs a) {
    constexpr int TILE = 1 << TL;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *s_data = smem;                         // [9][TILE]
    uint32_t *s_tw = smem + 9 * TILE;                // [9][tw_stride] (not with global twiddles)
    const int tw_stride = 1 << (a.log_n1 - 1);
    const int log_c = TL - a.log_n1;
    const int C = 1 << log_c;
    const unsigned nmask = (1u << a.log_n) - 1u;
    const size_t n2 = (size_t)1 << a.log_n2;
    const size_t col0 = (size_t)xcd_tile(blockIdx.x, gridDim.x, (C >= 4 || !a.xcd_order) ? 1u : 4u / (unsigned)C) * C;
    const Fr *in = a.in + (size_t)blockIdx.y * a.batch_stride;
    Fr *out = a.out + (size_t)blockIdx.y * a.batch_stride;

    if (!GTW) stage_twiddles_u(s_tw, tw_stride, a.w, a.log_n, a.log_n1, a.inverse);
    for (int t = threadIdx.x; t < TILE; t += NTT_THREADS_U) {
        const int c = t & (C - 1), i1 = t >> log_c;
        const size_t gi = (size_t)i1 * n2 + col0 + c;
        lds_st_u(s_data, TILE, (c << a.log_n1) + i1, load_u(a, in, gi));
    }
    __syncthreads();
    lds_dif_u<TL, GTW, R4>(s_data, s_tw, a.log_n1, tw_stride, a);
    for (int t = threadIdx.x; t < TILE; t += NTT_THREADS_U) {
        const int c = t & (C - 1), k1 = t >> log_c;
        const FrU v = lds_ld_u(s_data, TILE, (c << a.log_n1) + bitrev(k1, a.log_n1));
        const size_t i2 = col0 + c;
        unsigned e = (unsigned)(((i2 * (size_t)k1) << a.tw_shift) & nmask);    // inter-pass twiddle w_M^(i2*k1), w_M = w_N^(2^tw_shift)
        if (a.inverse) e = ((1u << a.log_n) - e) & nmask;
        gst(out + (size_t)k1 * n2 + i2, fru_mul_to_sat(v, fru_repack(gld(a.w + e))));      // w[0] is the saturated one
    }
}

template <int TL, bool GTW, int R4>
__global__ void __launch_bounds__(NTT_THREADS_U) ntt_pass_rows_u(NttPassArgs a) {
    constexpr int TILE = 1 << TL;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *s_data = smem;
    uint32_t *s_tw = smem + 9 * TILE;
    const int tw_stride = a.log_n2 > 0 ? 1 << (a.log_n2 - 1) : 1;
    const int log_r = TL - a.log_n2;
    const int R = 1 << log_r;
    const size_t n1 = (size_t)1 << a.log_n1;
    const size_t n2 = (size_t)1 << a.log_n2;
    const size_t row0 = (size_t)xcd_tile(blockIdx.x, gridDim.x, (R >= 4 || !a.xcd_order) ? 1u : 4u / (unsigned)R) * R;
    const Fr *in = a.in + (size_t)blockIdx.y * a.batch_stride;

    if (!GTW) stage_twiddles_u(s_tw, tw_stride, a.w, a.log_n, a.log_n2, a.inverse);
    for (int t = threadIdx.x; t < TILE; t += NTT_THREADS_U) {
        const int i2 = t & ((1 << a.log_n2) - 1), r = t >> a.log_n2;
        FrU v;
#pragma unroll
        for (int k = 0; k < 9; k++) v.l[k] = 0;
        if (row0 + r < n1) v = load_u(a, in, (row0 + r) * n2 + i2);
        lds_st_u(s_data, TILE, t, v);
    }
    __syncthreads();
    lds_dif_u<TL, GTW, R4>(s_data, s_tw, a.log_n2, tw_stride, a);
    const FrU post_c = a.post_const_on ? fru_repack(a.post_const) : fru_one_sat();
    for (int t = threadIdx.x; t < TILE; t += NTT_THREADS_U) {
        const int r = t & (R - 1), k2 = t >> log_r;
        if (row0 + r >= n1) continue;
        const FrU v = lds_ld_u(s_data, TILE, (r << a.log_n2) + bitrev(k2, a.log_n2));
        const size_t k = (size_t)blockIdx.y + (((row0 + r) + n1 * (size_t)k2) << a.out_stride_log);
        gst(a.out + k, fru_mul_to_sat(v, a.post ? fru_repack(gld(a.post + k)) : post_c));
    }
}

// U-form copy of the w table (9 x u32 per entry) for the global-twiddle kernels
__global__ void __launch_bounds__(256) ntt_wu_kernel(const Fr *w, uint32_t *wu, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FrU u = fru_from_sat(gld(w + i));
#pragma unroll
    for (int k = 0; k < 9; k++) wu[i * 9 + k] = u.l[k];
}

// out[i] = scale * base^i: a thread raises base to its first index (square-and-multiply) and walks FR_POWERS_RUN
// consecutive powers from there (one product each) instead of exponentiating for every element
static constexpr int FR_POWERS_RUN = 8;
__global__ void fr_powers_kernel(Fr *out, Fr base, Fr scale, size_t n) {
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * FR_POWERS_RUN;
    if (i0 >= n) return;
    Fr cur = fp_mul(fp_pow_u64(base, (uint64_t)i0), scale);
    for (int k = 0; k < FR_POWERS_RUN && i0 + k < n; k++) {
        gst(out + i0 + k, cur);
        cur = fp_mul(cur, base);
    }
}
static unsigned fr_powers_grid(size_t n) { return (unsigned)(((n + FR_POWERS_RUN - 1) / FR_POWERS_RUN + 255) / 256); }

void fr_powers_run(zkg16_ctx *ctx, Fr *out, const Fr &base, const Fr &scale, size_t n) {
    if (!n) return;
    hipLaunchKernelGGL(fr_powers_kernel, dim3(fr_powers_grid(n)), dim3(256), 0, ctx->stream, out, base, scale, n);
    ZK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ host
static Fr fr_from_u64_host(uint64_t v) {
    Fr c = Fr::zero();
    c.l[0] = (uint32_t)v;
    c.l[1] = (uint32_t)(v >> 32);
    return fp_to_mont(c);
}

NttTables *ntt_get_tables(zkg16_ctx *ctx, int log_n) {
    // the tables belong to the root ctx and are shared by its lanes; the build below ends with a stream synchronisation, so a
    // table that is in the map is complete for every stream of every lane
    zkg16_ctx *owner = ctx->root ? ctx->root : ctx;
    std::lock_guard<std::mutex> lk(owner->ntt_mu);
    auto it = owner->ntt_tables.find(log_n);
    if (it != owner->ntt_tables.end()) return it->second.get();
    auto t = std::make_unique<NttTables>();
    t->log_n = log_n;
    const size_t n = (size_t)1 << log_n;
    // 2^32-th root of unity 7^((r-1)/2^32) (Montgomery), squared down to order N
    Fr root;
    {
        const uint32_t R32[8] = {0x5f0e466au, 0xb9b58d8cu, 0x1819d7ecu, 0x5b1b4c80u, 0x52a31e64u, 0x0af53ae3u, 0x19e9b27bu, 0x5bf3addau};
        for (int i = 0; i < 8; i++) root.l[i] = R32[i];
        for (int i = log_n; i < 32; i++) root = fp_sqr(root);
    }
    const Fr g = fr_from_u64_host(7);
    const Fr g_inv = fp_inv(g);
    t->n_inv = fp_inv(fr_from_u64_host((uint64_t)n));
    Fr gn = g;
    for (int i = 0; i < log_n; i++) gn = fp_sqr(gn);
    t->zinv = fp_inv(fp_sub(gn, Fr::one()));
    t->w.alloc(n * sizeof(Fr));
    t->g.alloc(n * sizeof(Fr));
    t->gi.alloc(n * sizeof(Fr));
    const int bs = 256;
    const unsigned grid = fr_powers_grid(n);
    hipLaunchKernelGGL(fr_powers_kernel, dim3(grid), dim3(bs), 0, ctx->stream, t->w.as<Fr>(), root, Fr::one(), n);
    hipLaunchKernelGGL(fr_powers_kernel, dim3(grid), dim3(bs), 0, ctx->stream, t->g.as<Fr>(), g, Fr::one(), n);
    hipLaunchKernelGGL(fr_powers_kernel, dim3(grid), dim3(bs), 0, ctx->stream, t->gi.as<Fr>(), g_inv, t->n_inv, n);
    ZK_HIP(hipGetLastError());
    // built once per domain size on whichever stream is current; the ctx's other streams (witness map, setup's G2 pass) read
    // the tables later without an event between them, so the build is completed here — and only cached once it has succeeded
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    NttTables *raw = t.get();
    owner->ntt_tables[log_n] = std::move(t);
    return raw;
}

// Out of place: the transform of `src` ends in `dst` (N elements each); `src` is scratch afterwards (the column passes
// run in place on it, the last pass writes `dst`) — callers ping-pong two buffers instead of copying a result back
// (round 1 ended every transform with a device-to-device copy: 7 x 64 N bytes per proof).
// pw (optional): the witness map's point-wise stage fused into the first pass's load: src[i] <- (src[i]*b[i] - c[i]) / Z.
Fr *ntt_run(zkg16_ctx *ctx, Fr *data, Fr *tmp, int log_n, bool inverse, bool coset, const NttPointwise *pw) {
    bool &lds_attr_set = ctx->lds_attr_ntt;          // per ctx (= per device)
    if (!lds_attr_set) {   // 64-72 KiB tile + up to 36 KiB of twiddles, or a 144 KiB tile: above the 64 KiB default dynamic-LDS cap
        for (const void *f : {reinterpret_cast<const void *>(ntt_pass_cols), reinterpret_cast<const void *>(ntt_pass_rows),
                              reinterpret_cast<const void *>(ntt_pass_cols_u<11, false, true>), reinterpret_cast<const void *>(ntt_pass_rows_u<11, false, true>),
                              reinterpret_cast<const void *>(ntt_pass_cols_u<12, true, true>), reinterpret_cast<const void *>(ntt_pass_rows_u<12, true, true>),
                              reinterpret_cast<const void *>(ntt_pass_cols_u<11, false, false>), reinterpret_cast<const void *>(ntt_pass_rows_u<11, false, false>),
                              reinterpret_cast<const void *>(ntt_pass_cols_u<12, true, false>), reinterpret_cast<const void *>(ntt_pass_rows_u<12, true, false>),
                              reinterpret_cast<const void *>(ntt_pass_cols_u<12, true, 2>), reinterpret_cast<const void *>(ntt_pass_rows_u<12, true, 2>),
                              reinterpret_cast<const void *>(ntt_pass_cols_u<11, false, 2>), reinterpret_cast<const void *>(ntt_pass_rows_u<11, false, 2>),
                              reinterpret_cast<const void *>(ntt_pass_cols_u<12, true, 3>), reinterpret_cast<const void *>(ntt_pass_rows_u<12, true, 3>),
                              reinterpret_cast<const void *>(ntt_pass_cols_u<11, false, 3>), reinterpret_cast<const void *>(ntt_pass_rows_u<11, false, 3>)})
            ZK_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_attr_set = true;
    }
    const bool uform = ctx->opt_ntt_mode != 0;          // 1 (default): unsaturated butterflies; 0: saturated (the first version)
    const bool r4 = ctx->opt_ntt_radix == 4;
    const bool xch = ctx->opt_ntt_radix == 1;       // lane exchanges for the last stages
    const bool xch2 = ctx->opt_ntt_radix == 3;      // ... and for the top seven stages
    auto *k_cols = uform ? (xch2 ? ntt_pass_cols_u<11, false, 3> : xch ? ntt_pass_cols_u<11, false, 2> : r4 ? ntt_pass_cols_u<11, false, true> : ntt_pass_cols_u<11, false, false>) : ntt_pass_cols;
    auto *k_rows = uform ? (xch2 ? ntt_pass_rows_u<11, false, 3> : xch ? ntt_pass_rows_u<11, false, 2> : r4 ? ntt_pass_rows_u<11, false, true> : ntt_pass_rows_u<11, false, false>) : ntt_pass_rows;
    const unsigned nthreads = uform ? NTT_THREADS_U : NTT_THREADS;
    if (log_n > 3 * NTT_MAX_SUB_LOG - 2) throw HipError{hipErrorInvalidValue, "ntt: domain above build limit 2^31", __FILE__, __LINE__};
    NttTables *t = ntt_get_tables(ctx, log_n);
    const size_t n = (size_t)1 << log_n;
    NttPassArgs a;
    memset(&a, 0, sizeof a);
    a.w = t->w.as<Fr>();
    a.log_n = log_n;
    a.inverse = inverse ? 1 : 0;
    a.xcd_order = ctx->opt_ntt_xcd;
    const Fr *pre = (!inverse && coset) ? t->g.as<Fr>() : nullptr;
    const Fr *post = (inverse && coset) ? t->gi.as<Fr>() : nullptr;
    int post_const_on = (inverse && !coset) ? 1 : 0;
    a.post_const = t->n_inv;
    if (uform) {
        // passes that load without a conversion product (load_u): all of them, except the first when it carries the coset
        // pre-multiply (a full conversion) or the fused point-wise stage (likewise)
        const bool two_big = ctx->opt_ntt_mode != 3 && log_n > 2 * NTT_MAX_SUB_LOG && log_n <= 24;
        const int passes = log_n <= NTT_MAX_SUB_LOG ? 1 : (two_big || log_n <= 2 * NTT_MAX_SUB_LOG) ? 2 : 3;
        std::unique_lock<std::mutex> lazy(t->mu);           // lanes share the table object: one of them builds, the others wait
        if (t->u_passes != passes || !t->g_u.p) {
            DevBuf gu(n * sizeof(Fr)), giu(n * sizeof(Fr));
            const Fr g = fr_from_u64_host(7), g_inv = fp_inv(g);
            Fr back = fr_from_u64_host(1);
            for (int i = 0; i < 5 * passes; i++) back = fp_add(back, back);
            const unsigned grid = fr_powers_grid(n);
            hipLaunchKernelGGL(fr_powers_kernel, dim3(grid), dim3(256), 0, ctx->stream, gu.as<Fr>(), g, fr_from_u64_host(1024), n);
            hipLaunchKernelGGL(fr_powers_kernel, dim3(grid), dim3(256), 0, ctx->stream, giu.as<Fr>(), g_inv, fp_mul(t->n_inv, back), n);
            ZK_HIP(hipGetLastError());
            ZK_HIP(hipStreamSynchronize(ctx->stream));      // first use only; cached once complete
            t->g_u = std::move(gu);
            t->gi_u = std::move(giu);
            t->u_passes = passes;
        }
        lazy.unlock();
        // (the fused point-wise load is scaled to leave the same 2^-5 as a plain load, so one gi_u table serves both)
        const int deficit = passes - (pre ? 1 : 0);
        Fr back = fr_from_u64_host(1);
        for (int i = 0; i < 5 * deficit; i++) back = fp_add(back, back);
        if (pre) pre = t->g_u.as<Fr>();
        if (post) post = t->gi_u.as<Fr>();
        a.post_const = post_const_on ? fp_mul(t->n_inv, back) : back;
        post_const_on = 1;                                  // the last store always multiplies (by 2^(5 deficit) at least)
    }
    if (pw) {
        if (pre) throw HipError{hipErrorInvalidValue, "ntt: point-wise fusion needs a transform without a coset pre-multiply", __FILE__, __LINE__};
        a.pw_zinv = pw->zinv;
        // zinv * 2^10 * 2^256 = zinv 2^266 as an integer: (xb - c) 2^251 (*) zinv 2^266 = (xb - c)/Z * 2^256, i.e. the value with
        // the same 2^-5 a plain load leaves (see load_u)
        a.pw_zc = fru_repack(fp_mul(pw->zinv, fr_from_u64_host((uint64_t)1 << 10)));
    }
    // the pass that touches the input first carries the coset pre-multiply / the fused point-wise stage
    auto first = [&](NttPassArgs &p) {
        p.pre = pre;
        if (pw) { p.pw_b = pw->b; p.pw_c = pw->c; }
    };
    auto lds_bytes = [uform](int log_m) { return (size_t)(uform ? 9 : 8) * 4 * (NTT_TILE + (log_m > 0 ? (1 << (log_m - 1)) : 1)); };

    if (uform && ctx->opt_ntt_mode != 3 && log_n > 2 * NTT_MAX_SUB_LOG && log_n <= 24) {
        // 2^23, 2^24: two passes over 4096-point tiles (N1 = 2^(log_n - 12) columns-first, N2 = 2^12) with the sub-transform
        // twiddles read from a U-form table in global memory, instead of three passes over 2048-point tiles
        std::unique_lock<std::mutex> lazy(t->mu);
        if (!t->wu.p) {
            DevBuf wu(n * 9 * sizeof(uint32_t));
            hipLaunchKernelGGL(ntt_wu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, t->w.as<Fr>(), wu.as<uint32_t>(), n);
            ZK_HIP(hipGetLastError());
            ZK_HIP(hipStreamSynchronize(ctx->stream));      // first use only: other streams of this ctx may read the table next
            t->wu = std::move(wu);                          // cached only once it is complete
        }
        lazy.unlock();
        a.wu = t->wu.as<uint32_t>();
        a.log_n2 = 12;
        a.log_n1 = log_n - 12;
        const size_t big_lds = (size_t)9 * 4 * 4096 + ((xch || xch2) ? 9 * 4 * 64 : 0);       // + the lane-exchange chain's 64 twiddles
        {
            NttPassArgs p1 = a;
            p1.in = data; p1.out = data;
            first(p1);
            p1.batch_stride = 0;
            const unsigned grid = (unsigned)(((size_t)1 << a.log_n2) >> (12 - a.log_n1));
            ScopedKernelTimer kt(ctx, "ntt_pass_cols", (double)n);
            if (xch2) hipLaunchKernelGGL((ntt_pass_cols_u<12, true, 3>), dim3(grid), dim3(NTT_THREADS_U), big_lds, ctx->stream, p1);
            else if (xch) hipLaunchKernelGGL((ntt_pass_cols_u<12, true, 2>), dim3(grid), dim3(NTT_THREADS_U), big_lds, ctx->stream, p1);
            else if (r4) hipLaunchKernelGGL((ntt_pass_cols_u<12, true, true>), dim3(grid), dim3(NTT_THREADS_U), big_lds, ctx->stream, p1);
            else hipLaunchKernelGGL((ntt_pass_cols_u<12, true, false>), dim3(grid), dim3(NTT_THREADS_U), big_lds, ctx->stream, p1);
        }
        {
            NttPassArgs p2 = a;
            p2.in = data; p2.out = tmp;
            p2.post = post;
            p2.post_const_on = post_const_on;
            const unsigned grid = (unsigned)((size_t)1 << a.log_n1);
            ScopedKernelTimer kt(ctx, "ntt_pass_rows", (double)n);
            if (xch2) hipLaunchKernelGGL((ntt_pass_rows_u<12, true, 3>), dim3(grid), dim3(NTT_THREADS_U), big_lds, ctx->stream, p2);
            else if (xch) hipLaunchKernelGGL((ntt_pass_rows_u<12, true, 2>), dim3(grid), dim3(NTT_THREADS_U), big_lds, ctx->stream, p2);
            else if (r4) hipLaunchKernelGGL((ntt_pass_rows_u<12, true, true>), dim3(grid), dim3(NTT_THREADS_U), big_lds, ctx->stream, p2);
            else hipLaunchKernelGGL((ntt_pass_rows_u<12, true, false>), dim3(grid), dim3(NTT_THREADS_U), big_lds, ctx->stream, p2);
        }
    } else if (log_n <= NTT_MAX_SUB_LOG) {
        a.log_n1 = 0;
        a.log_n2 = log_n;
        a.in = data;
        a.out = tmp;
        first(a);
        a.post = post;
        a.post_const_on = post_const_on;
        ScopedKernelTimer kt(ctx, "ntt_pass_rows", (double)n);
        hipLaunchKernelGGL(k_rows, dim3(1), dim3(nthreads), lds_bytes(log_n), ctx->stream, a);
    } else {
        // N = N0 * M (N0 = 1 for N <= 2^22): [outer column pass over N0] then the two-pass transform of size M, batched over k0 < N0
        const int log_m = log_n <= 2 * NTT_MAX_SUB_LOG ? log_n : 2 * NTT_MAX_SUB_LOG;
        const int log_n0 = log_n - log_m;
        if (log_n0 > 0) {
            NttPassArgs p0 = a;
            p0.in = data; p0.out = data;
            first(p0);
            p0.log_n1 = log_n0; p0.log_n2 = log_m;
            const unsigned grid = (unsigned)(((size_t)1 << log_m) >> (NTT_TILE_LOG - log_n0));
            ScopedKernelTimer kt(ctx, "ntt_pass_cols", (double)n);
            hipLaunchKernelGGL(k_cols, dim3(grid), dim3(nthreads), lds_bytes(log_n0), ctx->stream, p0);
        }
        a.log_n2 = log_m / 2;
        a.log_n1 = log_m - a.log_n2;
        const unsigned batches = 1u << log_n0;
        {
            NttPassArgs p1 = a;
            p1.in = data; p1.out = data;
            if (log_n0 == 0) first(p1);
            p1.tw_shift = log_n0;
            p1.batch_stride = (size_t)1 << log_m;
            const unsigned grid = (unsigned)(((size_t)1 << a.log_n2) >> (NTT_TILE_LOG - a.log_n1));
            ScopedKernelTimer kt(ctx, "ntt_pass_cols", (double)n);
            hipLaunchKernelGGL(k_cols, dim3(grid, batches), dim3(nthreads), lds_bytes(a.log_n1), ctx->stream, p1);
        }
        {
            NttPassArgs p2 = a;
            p2.in = data; p2.out = tmp;
            p2.post = post;
            p2.post_const_on = post_const_on;
            p2.batch_stride = (size_t)1 << log_m;
            p2.out_stride_log = log_n0;
            const unsigned grid = (unsigned)(((size_t)1 << a.log_n1) >> (NTT_TILE_LOG - a.log_n2));
            ScopedKernelTimer kt(ctx, "ntt_pass_rows", (double)n);
            hipLaunchKernelGGL(k_rows, dim3(grid, batches), dim3(nthreads), lds_bytes(a.log_n2), ctx->stream, p2);
        }
    }
    ZK_HIP(hipGetLastError());
    return tmp;
}

}  // namespace zk
