// BLS12-381 prime fields for gfx950 (and for the host-side proof tail): Fr (8 x u32) and Fq (12 x u32),
// Montgomery form with R = 2^256 / 2^384 — bit-identical in memory to arkworks'
// `Fp<MontBackend<_, 4|6>>` (4|6 little-endian u64 limbs), which is what the reference's
// ProvingKey / witness vectors hold (imports at /root/reference/src/arkworks/backend/matrix_proof.rs:13-20).
//
// Why u32 limbs: CDNA4's integer multiplier is 32x32 (v_mad_u64_u32 / v_mul_hi_u32); a u64 limb
// would be split by the compiler anyway.  All loops are fully unrolled so limbs live in VGPRs and the
// modulus limbs fold into instruction literals (no constant-memory traffic).
//
// Everything here is __host__ __device__: the same code runs in the kernels and in the O(1) host tail
// of the prover (window Horner, r/s scalar multiplications, affine normalisation).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD __host__ __device__ inline   // host pass: let the compiler decide (keeps build times sane)
#endif

namespace zk {

// ------------------------------------------------------------------------------------------------
// Field parameter packs.  mod(i)/r(i)/r2(i) are constexpr switch tables so that, after unrolling, every
// use is an immediate.
struct FrP {
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xffffffffu;  // -r^-1 mod 2^32
    static constexpr uint64_t INV64 = 0xfffffffeffffffffull;
    ZK_HD static constexpr uint32_t mod(int i) {
        constexpr uint32_t M[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                   0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return M[i];
    }
    ZK_HD static constexpr uint32_t one(int i) {  // R mod r
        constexpr uint32_t M[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                   0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return M[i];
    }
    ZK_HD static constexpr uint32_t r2(int i) {  // R^2 mod r
        constexpr uint32_t M[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                   0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return M[i];
    }
};

struct FqP {
    static constexpr int N = 12;
    static constexpr uint32_t INV = 0xfffcfffdu;  // -q^-1 mod 2^32
    static constexpr uint64_t INV64 = 0x89f3fffcfffcfffdull;
    ZK_HD static constexpr uint32_t mod(int i) {
        constexpr uint32_t M[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                    0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return M[i];
    }
    ZK_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t M[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                    0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return M[i];
    }
    ZK_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t M[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                    0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return M[i];
    }
};

// ------------------------------------------------------------------------------------------------
template <class P>
struct Fp {
    static constexpr int N = P::N;
    uint32_t l[N];

    ZK_HD static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    ZK_HD static Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::one(i);
        return r;
    }
    ZK_HD static Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::r2(i);
        return r;
    }
    ZK_HD bool is_zero() const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc |= l[i];
        return acc == 0;
    }
    ZK_HD bool operator==(const Fp &o) const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc |= l[i] ^ o.l[i];
        return acc == 0;
    }
    ZK_HD bool operator!=(const Fp &o) const { return !(*this == o); }
};

// r = a - p if a >= p (a < 2p assumed); branch-free: compute the difference, keep it if no borrow.
template <class P>
ZK_HD void fp_reduce_once(Fp<P> &a) {
    constexpr int N = P::N;
    uint32_t d[N];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)a.l[i] - P::mod(i) - borrow;
        d[i] = (uint32_t)t;
        borrow = (uint32_t)(t >> 63);
    }
#pragma unroll
    for (int i = 0; i < N; i++) a.l[i] = borrow ? a.l[i] : d[i];
}

template <class P>
ZK_HD Fp<P> fp_add(const Fp<P> &a, const Fp<P> &b) {
    constexpr int N = P::N;
    Fp<P> r;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)a.l[i] + b.l[i] + carry;
        r.l[i] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
    // 2p < 2^(32N) for both fields: no carry out of the top limb
    fp_reduce_once(r);
    return r;
}

template <class P>
ZK_HD Fp<P> fp_sub(const Fp<P> &a, const Fp<P> &b) {
    constexpr int N = P::N;
    Fp<P> r;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint32_t)t;
        borrow = (uint32_t)(t >> 63);
    }
    uint32_t mask = 0u - borrow;  // add p back iff we borrowed
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)r.l[i] + (P::mod(i) & mask) + carry;
        r.l[i] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
    return r;
}

template <class P>
ZK_HD Fp<P> fp_neg(const Fp<P> &a) {
    constexpr int N = P::N;
    Fp<P> r;
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < N; i++) nz |= a.l[i];
    uint32_t mask = nz ? 0xffffffffu : 0u;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)(P::mod(i) & mask) - a.l[i] - borrow;
        r.l[i] = (uint32_t)t;
        borrow = (uint32_t)(t >> 63);
    }
    return r;
}

template <class P>
ZK_HD Fp<P> fp_dbl(const Fp<P> &a) {
    return fp_add(a, a);
}

// Montgomery product a*b*R^-1 mod p (CIOS, 32-bit limbs).  Each inner step is one
// v_mad_u64_u32 (32x32+64) plus a 64-bit add; since the top bit of both moduli is clear the running
// value stays below 2p and fits N limbs + one carry word.
template <class P>
ZK_HD Fp<P> fp_mul_inline(const Fp<P> &a, const Fp<P> &b) {
    constexpr int N = P::N;
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
        const uint32_t bi = b.l[i];
#pragma unroll
        for (int j = 0; j < N; j++) {
            uint64_t x = (uint64_t)a.l[j] * bi + t[j] + c;
            t[j] = (uint32_t)x;
            c = x >> 32;
        }
        uint64_t top = (uint64_t)t[N] + c;  // < 2^33
        const uint32_t m = t[0] * P::INV;
        uint64_t x = (uint64_t)m * P::mod(0) + t[0];
        c = x >> 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            x = (uint64_t)m * P::mod(j) + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = x >> 32;
        }
        top += c;
        t[N - 1] = (uint32_t)top;
        t[N] = (uint32_t)(top >> 32);
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    // t[N] is always 0 here (value < 2p < 2^(32N))
    fp_reduce_once(r);
    return r;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host tail (window Horner, r/s scalar multiplications, normalisation): same values, 64-bit limbs + __int128
// (the byte layout of 2k u32 limbs and k u64 limbs is identical on little-endian hosts).
template <class P>
inline Fp<P> fp_mul_host64(const Fp<P> &a, const Fp<P> &b) {
    constexpr int M = P::N / 2;
    typedef unsigned __int128 u128;
    uint64_t A[M], B[M], MOD[M], t[M + 2];
    for (int i = 0; i < M; i++) {
        A[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
        B[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
        MOD[i] = (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32);
    }
    for (int i = 0; i < M + 2; i++) t[i] = 0;
    for (int i = 0; i < M; i++) {
        u128 c = 0;
        for (int j = 0; j < M; j++) {
            u128 x = (u128)A[j] * B[i] + t[j] + c;
            t[j] = (uint64_t)x;
            c = x >> 64;
        }
        u128 x = (u128)t[M] + c;
        t[M] = (uint64_t)x;
        t[M + 1] = (uint64_t)(x >> 64);
        const uint64_t m = t[0] * P::INV64;
        c = ((u128)m * MOD[0] + t[0]) >> 64;
        for (int j = 1; j < M; j++) {
            x = (u128)m * MOD[j] + t[j] + c;
            t[j - 1] = (uint64_t)x;
            c = x >> 64;
        }
        x = (u128)t[M] + c;
        t[M - 1] = (uint64_t)x;
        t[M] = t[M + 1] + (uint64_t)(x >> 64);
    }
    Fp<P> r;
    for (int i = 0; i < M; i++) {
        r.l[2 * i] = (uint32_t)t[i];
        r.l[2 * i + 1] = (uint32_t)(t[i] >> 32);
    }
    fp_reduce_once(r);
    return r;
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// Device: the 12-limb product is a real function call (one copy of the ~1.4k-instruction body per code
// object instead of one per use).  A point addition is 10-14 of these back to back; inlined, a single
// bucket-accumulation loop body would be >60 KB of code (larger than the instruction cache shared by a
// CU pair) and hipcc needs tens of minutes to schedule it.  Operands travel in 24 VGPRs: native vector
// types are passed directly by the AMDGPU calling convention (aggregates are capped at 16 registers).
typedef uint32_t zk_v4u __attribute__((ext_vector_type(4)));
struct FqRet { uint32_t l[12]; };
__device__ __noinline__ FqRet fq_mul_call(zk_v4u a0, zk_v4u a1, zk_v4u a2, zk_v4u b0, zk_v4u b1, zk_v4u b2);
#endif

template <class P>
ZK_HD Fp<P> fp_mul(const Fp<P> &a, const Fp<P> &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (P::N == 12) {
        zk_v4u a0 = {a.l[0], a.l[1], a.l[2], a.l[3]}, a1 = {a.l[4], a.l[5], a.l[6], a.l[7]}, a2 = {a.l[8], a.l[9], a.l[10], a.l[11]};
        zk_v4u b0 = {b.l[0], b.l[1], b.l[2], b.l[3]}, b1 = {b.l[4], b.l[5], b.l[6], b.l[7]}, b2 = {b.l[8], b.l[9], b.l[10], b.l[11]};
        FqRet r = fq_mul_call(a0, a1, a2, b0, b1, b2);
        Fp<P> o;
#pragma unroll
        for (int i = 0; i < 12; i++) o.l[i] = r.l[i];
        return o;
    } else {
        return fp_mul_inline(a, b);
    }
#else
    return fp_mul_host64(a, b);
#endif
}

template <class P>
ZK_HD Fp<P> fp_sqr(const Fp<P> &a) {
    return fp_mul(a, a);
}

// canonical <-> Montgomery
template <class P>
ZK_HD Fp<P> fp_to_mont(const Fp<P> &canon) {
    return fp_mul(canon, Fp<P>::r2());
}
template <class P>
ZK_HD Fp<P> fp_from_mont(const Fp<P> &a) {
    Fp<P> one = Fp<P>::zero();
    one.l[0] = 1;
    return fp_mul(a, one);
}

// a^e for a small run-time exponent (used for twiddle / coset power tables)
template <class P>
ZK_HD Fp<P> fp_pow_u64(const Fp<P> &a, uint64_t e) {
    Fp<P> acc = Fp<P>::one(), base = a;
    while (e) {
        if (e & 1) acc = fp_mul(acc, base);
        base = fp_sqr(base);
        e >>= 1;
    }
    return acc;
}

// a^(p-2).  Host tail + setup helpers only (never in a hot loop).
template <class P>
ZK_HD Fp<P> fp_inv(const Fp<P> &a) {
    constexpr int N = P::N;
    uint32_t e[N];  // p - 2
    uint32_t borrow = 2;
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)P::mod(i) - borrow;
        e[i] = (uint32_t)t;
        borrow = (uint32_t)(t >> 63);
    }
    Fp<P> acc = Fp<P>::one();
    bool started = false;
    for (int i = N * 32 - 1; i >= 0; i--) {
        if (started) acc = fp_sqr(acc);
        if ((e[i / 32] >> (i % 32)) & 1) {
            acc = started ? fp_mul(acc, a) : a;
            started = true;
        }
    }
    return acc;
}

using Fr = Fp<FrP>;
using Fq = Fp<FqP>;

#if defined(__HIP_DEVICE_COMPILE__)
// Every translation unit that multiplies in Fq on the device gets its own (internal-linkage-free, weak)
// copy of the callee: HIP code objects are linked per TU (no -fgpu-rdc).
__device__ __noinline__ __attribute__((weak)) FqRet fq_mul_call(zk_v4u a0, zk_v4u a1, zk_v4u a2, zk_v4u b0, zk_v4u b1, zk_v4u b2) {
    Fq a, b;
    a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w; a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w;
    a.l[8] = a2.x; a.l[9] = a2.y; a.l[10] = a2.z; a.l[11] = a2.w;
    b.l[0] = b0.x; b.l[1] = b0.y; b.l[2] = b0.z; b.l[3] = b0.w; b.l[4] = b1.x; b.l[5] = b1.y; b.l[6] = b1.z; b.l[7] = b1.w;
    b.l[8] = b2.x; b.l[9] = b2.y; b.l[10] = b2.z; b.l[11] = b2.w;
    Fq r = fp_mul_inline(a, b);
    FqRet o;
#pragma unroll
    for (int i = 0; i < 12; i++) o.l[i] = r.l[i];
    return o;
}
#endif

// ------------------------------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2+1)
struct Fq2 {
    Fq c0, c1;
    ZK_HD static Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
    ZK_HD static Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
    ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    ZK_HD bool operator==(const Fq2 &o) const { return c0 == o.c0 && c1 == o.c1; }
    ZK_HD bool operator!=(const Fq2 &o) const { return !(*this == o); }
};

// Uniform free-function interface over Fq / Fq2 so the curve code is written once.
ZK_HD Fq f_add(const Fq &a, const Fq &b) { return fp_add(a, b); }
ZK_HD Fq f_sub(const Fq &a, const Fq &b) { return fp_sub(a, b); }
ZK_HD Fq f_neg(const Fq &a) { return fp_neg(a); }
ZK_HD Fq f_dbl(const Fq &a) { return fp_dbl(a); }
ZK_HD Fq f_mul(const Fq &a, const Fq &b) { return fp_mul(a, b); }
ZK_HD Fq f_sqr(const Fq &a) { return fp_sqr(a); }
ZK_HD Fq f_inv(const Fq &a) { return fp_inv(a); }

ZK_HD Fr f_add(const Fr &a, const Fr &b) { return fp_add(a, b); }
ZK_HD Fr f_sub(const Fr &a, const Fr &b) { return fp_sub(a, b); }
ZK_HD Fr f_neg(const Fr &a) { return fp_neg(a); }
ZK_HD Fr f_mul(const Fr &a, const Fr &b) { return fp_mul(a, b); }
ZK_HD Fr f_sqr(const Fr &a) { return fp_sqr(a); }
ZK_HD Fr f_inv(const Fr &a) { return fp_inv(a); }

ZK_HD Fq2 f_add(const Fq2 &a, const Fq2 &b) { return Fq2{fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }
ZK_HD Fq2 f_sub(const Fq2 &a, const Fq2 &b) { return Fq2{fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }
ZK_HD Fq2 f_neg(const Fq2 &a) { return Fq2{fp_neg(a.c0), fp_neg(a.c1)}; }
ZK_HD Fq2 f_dbl(const Fq2 &a) { return Fq2{fp_dbl(a.c0), fp_dbl(a.c1)}; }
ZK_HD Fq2 f_mul(const Fq2 &a, const Fq2 &b) {
    // Karatsuba: 3 base-field products
    Fq v0 = fp_mul(a.c0, b.c0);
    Fq v1 = fp_mul(a.c1, b.c1);
    Fq s = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
    return Fq2{fp_sub(v0, v1), fp_sub(fp_sub(s, v0), v1)};
}
ZK_HD Fq2 f_sqr(const Fq2 &a) {
    // (c0 + c1 u)^2 = (c0+c1)(c0-c1) + 2 c0 c1 u : 2 base-field products
    Fq p = fp_mul(a.c0, a.c1);
    Fq r0 = fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1));
    return Fq2{r0, fp_dbl(p)};
}
ZK_HD Fq2 f_inv(const Fq2 &a) {
    Fq n = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
    return Fq2{fp_mul(a.c0, n), fp_neg(fp_mul(a.c1, n))};
}

}  // namespace zk
