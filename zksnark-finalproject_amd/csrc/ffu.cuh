// BLS12-381 Fq in an UNSATURATED radix-2^29 representation for the MSM kernels (14 x 29-bit limbs in u32).
//
// Why: on gfx950 v_mad_u64_u32 (32x32 + 64-bit addend) issues at the same rate as a 64-bit add or an add-with-carry
// (measured: ~4 cycles per wave64 instruction, profiles/microbench_r1.txt).  With saturated 32-bit limbs every partial
// product drags a carry instruction (and, because 64-bit operands must sit in even-aligned VGPR pairs, register
// moves) behind it: 288 multiply-adds became ~1,350 VALU instructions.  With 29-bit limbs a whole column of the
// product — up to 28 partial products of < 2^58 — accumulates in ONE 64-bit register pair by back-to-back
// v_mad_u64_u32 with no carry handling at all; carries are resolved once per column (a shift and a mask).
// 2 * 14^2 = 392 multiply-adds and ~100 other instructions per Montgomery product instead of ~1,350.
//
// Representation ("U-form"): x is held as x * 2^406 mod q (Montgomery radix R' = 2^(29*14)), as an integer in
// [0, 2^12 q) written in base 2^29 with every limb < 2^29 ("normalised").  Values are NOT kept below q:
//   * fqu_mul / fqu_sqr accept any two inputs < 2^12 q and return a value < 2q   (R' > 2^24 q makes the
//     final conditional subtraction of Montgomery's algorithm unnecessary);
//   * fqu_add is a limb-wise add + carry propagation (no modular reduction);
//   * fqu_sub<L>(a, b) = a + (M_L q - b) with M_L in {8, 32, 64}: the caller picks the level so that b <= (M_L - 1) q
//     (bounds are worked out per formula in DESIGN.md "value bounds"); the constant M_L q is stored with every
//     limb pre-borrowed to >= 2^29 so the limb-wise subtraction never underflows.
// Only exact zero limbs encode the additive identity used as a flag (point at infinity); "is this value 0 mod q"
// (the P == Q / P == -Q tests of the addition formulas) is fqu_is_zero_mod: value in {0, q, 2q, ...}.
//
// The arkworks-compatible saturated Montgomery form (x * 2^384 mod q, 12 x u32 — what the proving key and the proof
// use, include/zkg16.h) is converted at the boundaries: fqu_from_sat (one product by 2^(22+406) mod q) when the
// proving key is loaded, fqu_to_sat (one product by 2^384 mod q, canonical reduction, repack) on the five
// window-sum vectors a proof returns to the host.
#pragma once
#include "ff.cuh"

namespace zk {

struct FqU {
    static constexpr int N = 14;
    static constexpr int W = 29;
    static constexpr uint32_t MASK = (1u << 29) - 1u;
    uint32_t l[14];

    ZK_HD static FqU zero() {
        FqU r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.l[i] = 0;
        return r;
    }
    ZK_HD bool is_zero() const {   // exact zero limbs (flag encoding), NOT "0 mod q"
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) acc |= l[i];
        return acc == 0;
    }
    ZK_HD static FqU one();
};

struct FqUP {
    static constexpr uint32_t INV = 0x1ffcfffdu;   // -q^-1 mod 2^29
    ZK_HD static constexpr uint32_t mod(int i) {
        constexpr uint32_t M[14] = {0x1fffaaabu, 0x0ff7ffffu, 0x14ffffeeu, 0x17fffd62u, 0x0f6241eau, 0x09507b58u, 0x0afd9cc3u,
                                    0x109e70a2u, 0x1764774bu, 0x121a5d66u, 0x12c6e9edu, 0x12ffcd34u, 0x00111ea3u, 0x0000000du};
        return M[i];
    }
    ZK_HD static constexpr uint32_t one(int i) {      // 2^406 mod q
        constexpr uint32_t M[14] = {0x03a9fb84u, 0x0ba00690u, 0x071288f1u, 0x0f59bcc5u, 0x126cb614u, 0x0585bf36u, 0x1b85ac3du,
                                    0x1cf856fau, 0x1891ecbdu, 0x1a7eec05u, 0x155a88f0u, 0x0741ac6du, 0x1317c30fu, 0x00000009u};
        return M[i];
    }
    ZK_HD static constexpr uint32_t c_in(int i) {     // 2^(22+406) mod q : saturated-Montgomery -> U-form
        constexpr uint32_t M[14] = {0x1fddebbdu, 0x1a4f5474u, 0x0291f399u, 0x14d03b3cu, 0x0f6cad2cu, 0x1b4cabcau, 0x1592827cu,
                                    0x021c6ac7u, 0x1ec52a84u, 0x16fd5ec4u, 0x0c960da6u, 0x0fd2af6bu, 0x13263591u, 0x0000000bu};
        return M[i];
    }
    ZK_HD static constexpr uint32_t d_out(int i) {    // 2^384 mod q : U-form -> saturated-Montgomery
        constexpr uint32_t M[14] = {0x0002fffdu, 0x10480000u, 0x0300009du, 0x08001788u, 0x158baebfu, 0x0c2ba9e3u, 0x1d157d22u,
                                    0x0a6e0a4au, 0x0d77ce58u, 0x1d12b763u, 0x1701c6a5u, 0x1501c926u, 0x1f65ec3fu, 0x0000000au};
        return M[i];
    }
    // M*q with limbs 0..12 pre-borrowed to [2^29 - 1, 2^30): limb-wise "M*q - b" never underflows for normalised b <= (M-1) q
    ZK_HD static constexpr uint32_t m8(int i) {
        constexpr uint32_t M[14] = {0x3ffd5558u, 0x3fbffffeu, 0x27ffff72u, 0x3fffeb14u, 0x3b120f54u, 0x2a83dac2u, 0x37ece619u,
                                    0x24f38511u, 0x3b23ba5bu, 0x30d2eb34u, 0x36374f6bu, 0x37fe69a3u, 0x2088f51bu, 0x00000067u};
        return M[i];
    }
    ZK_HD static constexpr uint32_t m32(int i) {
        constexpr uint32_t M[14] = {0x3ff55560u, 0x3efffffeu, 0x3ffffdceu, 0x3fffac53u, 0x2c483d56u, 0x2a0f6b0eu, 0x3fb39868u,
                                    0x33ce1449u, 0x2c8ee96fu, 0x234bacd6u, 0x38dd3db1u, 0x3ff9a691u, 0x2223d471u, 0x0000019fu};
        return M[i];
    }
    ZK_HD static constexpr uint32_t m64(int i) {
        constexpr uint32_t M[14] = {0x3feaaac0u, 0x3dfffffeu, 0x3ffffb9eu, 0x3fff58a8u, 0x38907aaeu, 0x341ed61du, 0x3f6730d1u,
                                    0x279c2894u, 0x391dd2e0u, 0x269759adu, 0x31ba7b63u, 0x3ff34d24u, 0x2447a8e4u, 0x0000033fu};
        return M[i];
    }
    ZK_HD static constexpr uint32_t m128(int i) {
        constexpr uint32_t M[14] = {0x3fd55580u, 0x3bfffffeu, 0x3ffff73eu, 0x3ffeb152u, 0x3120f55eu, 0x283dac3cu, 0x3ece61a4u, 0x2f38512au, 0x323ba5c1u, 0x2d2eb35cu, 0x2374f6c7u, 0x3fe69a4au, 0x288f51cau, 0x0000067fu};
        return M[i];
    }
};

ZK_HD FqU FqU::one() {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = FqUP::one(i);
    return r;
}

// carry propagation: limbs back below 2^29 (the top limb keeps whatever is left; values stay < 2^406)
ZK_HD void fqu_normalise(FqU &a) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const uint32_t v = a.l[i] + c;
        a.l[i] = v & FqU::MASK;
        c = v >> 29;
    }
    a.l[13] += c;
}

ZK_HD FqU fqu_add(const FqU &a, const FqU &b) {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + b.l[i];
    fqu_normalise(r);
    return r;
}
ZK_HD FqU fqu_dbl(const FqU &a) { return fqu_add(a, a); }

// a - b + M*q ; L = 8 / 32 / 64 / 128 ; requires b normalised and b <= (L-1) q
template <int L>
ZK_HD FqU fqu_sub(const FqU &a, const FqU &b) {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint32_t m = L == 8 ? FqUP::m8(i) : (L == 32 ? FqUP::m32(i) : (L == 64 ? FqUP::m64(i) : FqUP::m128(i)));
        r.l[i] = a.l[i] + (m - b.l[i]);
    }
    fqu_normalise(r);
    return r;
}
ZK_HD FqU fqu_neg(const FqU &a) {   // 8q - a  (a <= 7q); keeps exact zero as exact zero so flags survive
    if (a.is_zero()) return a;
    return fqu_sub<8>(FqU::zero(), a);
}

// Montgomery product a*b*2^-406 mod q, product-scanning: one 64-bit accumulator per column, carries once per column.
// Inputs: normalised, values < 2^12 q.  Output: normalised, value < 2q.
template <bool SQR>
ZK_HD FqU fqu_mul_impl(const FqU &a, const FqU &b) {
    constexpr int N = 14;
    uint32_t m[N];
    uint32_t a2[N];
    if (SQR) {
#pragma unroll
        for (int i = 0; i < N; i++) a2[i] = a.l[i] << 1;
    }
    FqU r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; k++) {
        const int lo = k < N ? 0 : k - N + 1;
        const int hi = k < N ? k : N - 1;
        if (SQR) {
#pragma unroll
            for (int i = lo; i <= hi; i++) {
                const int j = k - i;
                if (i < j) acc += (uint64_t)a2[i] * a.l[j];
                else if (i == j) acc += (uint64_t)a.l[i] * a.l[i];
            }
        } else {
#pragma unroll
            for (int i = lo; i <= hi; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
        }
        if (k < N) {
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * FqUP::mod(k - i);
            m[k] = ((uint32_t)acc * FqUP::INV) & FqU::MASK;
            acc += (uint64_t)m[k] * FqUP::mod(0);
            acc >>= 29;
        } else {
#pragma unroll
            for (int i = lo; i <= hi; i++) acc += (uint64_t)m[i] * FqUP::mod(k - i);
            r.l[k - N] = (uint32_t)acc & FqU::MASK;
            acc >>= 29;
        }
    }
    r.l[N - 1] = (uint32_t)acc;
    return r;
}

// a*b + c*d with ONE Montgomery reduction: (ab + cd) 2^-406 mod q.  A column holds <= 28 partial products and <= 14 reduction
// products, each < 2^58: < 2^63.4, still one 64-bit accumulator.  Inputs normalised, < 2^12 q; output normalised, < 2q
// ((2 * 2^24 q^2 + 2^406 q) / 2^406 < 2q).  Always inlined (four operands = 56 registers do not fit the 32 argument VGPRs of
// a device-function call): used once per mixed addition, for Y3 = R (Q - X3) + (-Y1) PPP.
ZK_HD FqU fqu_mul2(const FqU &a, const FqU &b, const FqU &c, const FqU &d) {
    constexpr int N = 14;
    uint32_t m[N];
    FqU r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; k++) {
        const int lo = k < N ? 0 : k - N + 1;
        const int hi = k < N ? k : N - 1;
#pragma unroll
        for (int i = lo; i <= hi; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = lo; i <= hi; i++) acc += (uint64_t)c.l[i] * d.l[k - i];
        if (k < N) {
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * FqUP::mod(k - i);
            m[k] = ((uint32_t)acc * FqUP::INV) & FqU::MASK;
            acc += (uint64_t)m[k] * FqUP::mod(0);
            acc >>= 29;
        } else {
#pragma unroll
            for (int i = lo; i <= hi; i++) acc += (uint64_t)m[i] * FqUP::mod(k - i);
            r.l[k - N] = (uint32_t)acc & FqU::MASK;
            acc >>= 29;
        }
    }
    r.l[N - 1] = (uint32_t)acc;
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// one copy of each body per code object (see the note on fq_mul_call in ff.cuh); 28 VGPR arguments
typedef uint32_t zk_v2u __attribute__((ext_vector_type(2)));
struct FqURet { uint32_t l[14]; };
__device__ __noinline__ __attribute__((weak)) FqURet fqu_mul_call(zk_v4u a0, zk_v4u a1, zk_v4u a2, zk_v2u a3, zk_v4u b0, zk_v4u b1, zk_v4u b2, zk_v2u b3) {
    FqU a, b;
    a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w; a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w;
    a.l[8] = a2.x; a.l[9] = a2.y; a.l[10] = a2.z; a.l[11] = a2.w; a.l[12] = a3.x; a.l[13] = a3.y;
    b.l[0] = b0.x; b.l[1] = b0.y; b.l[2] = b0.z; b.l[3] = b0.w; b.l[4] = b1.x; b.l[5] = b1.y; b.l[6] = b1.z; b.l[7] = b1.w;
    b.l[8] = b2.x; b.l[9] = b2.y; b.l[10] = b2.z; b.l[11] = b2.w; b.l[12] = b3.x; b.l[13] = b3.y;
    const FqU r = fqu_mul_impl<false>(a, b);
    FqURet o;
#pragma unroll
    for (int i = 0; i < 14; i++) o.l[i] = r.l[i];
    return o;
}
__device__ __noinline__ __attribute__((weak)) FqURet fqu_sqr_call(zk_v4u a0, zk_v4u a1, zk_v4u a2, zk_v2u a3) {
    FqU a;
    a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w; a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w;
    a.l[8] = a2.x; a.l[9] = a2.y; a.l[10] = a2.z; a.l[11] = a2.w; a.l[12] = a3.x; a.l[13] = a3.y;
    const FqU r = fqu_mul_impl<true>(a, a);
    FqURet o;
#pragma unroll
    for (int i = 0; i < 14; i++) o.l[i] = r.l[i];
    return o;
}
#endif

ZK_HD FqU fqu_mul(const FqU &a, const FqU &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    zk_v4u a0 = {a.l[0], a.l[1], a.l[2], a.l[3]}, a1 = {a.l[4], a.l[5], a.l[6], a.l[7]}, a2 = {a.l[8], a.l[9], a.l[10], a.l[11]};
    zk_v2u a3 = {a.l[12], a.l[13]};
    zk_v4u b0 = {b.l[0], b.l[1], b.l[2], b.l[3]}, b1 = {b.l[4], b.l[5], b.l[6], b.l[7]}, b2 = {b.l[8], b.l[9], b.l[10], b.l[11]};
    zk_v2u b3 = {b.l[12], b.l[13]};
    const FqURet r = fqu_mul_call(a0, a1, a2, a3, b0, b1, b2, b3);
    FqU o;
#pragma unroll
    for (int i = 0; i < 14; i++) o.l[i] = r.l[i];
    return o;
#else
    return fqu_mul_impl<false>(a, b);
#endif
}
ZK_HD FqU fqu_sqr(const FqU &a) {
#if defined(__HIP_DEVICE_COMPILE__)
    zk_v4u a0 = {a.l[0], a.l[1], a.l[2], a.l[3]}, a1 = {a.l[4], a.l[5], a.l[6], a.l[7]}, a2 = {a.l[8], a.l[9], a.l[10], a.l[11]};
    zk_v2u a3 = {a.l[12], a.l[13]};
    const FqURet r = fqu_sqr_call(a0, a1, a2, a3);
    FqU o;
#pragma unroll
    for (int i = 0; i < 14; i++) o.l[i] = r.l[i];
    return o;
#else
    return fqu_mul_impl<true>(a, a);
#endif
}

// value == 0 mod q, for normalised a < 2^12 q.  The base-2^29 digits are unique, so a == 0 mod q iff a == k q as
// integers for the k suggested by the top limb (q / 2^377 = 13.002...); a one-limb filter rejects almost always.
ZK_HD bool fqu_is_zero_mod(const FqU &a) {
    const uint32_t kest = (a.l[13] * 20162u) >> 18;         // ~ a13 / 13.002 ; exact k is within +-1
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const uint32_t k = kest + (uint32_t)t - 1u;
        if (k > 4096u) continue;                              // also skips kest - 1 when kest == 0
        if (((k * FqUP::mod(0)) & FqU::MASK) != a.l[0]) continue;
        int64_t carry = 0;
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const int64_t v = (int64_t)a.l[i] - (int64_t)((uint64_t)k * FqUP::mod(i)) + carry;
            nz |= (uint32_t)(v & FqU::MASK);
            carry = v >> 29;                                  // arithmetic shift = floor
        }
        if (nz == 0 && carry == 0) return true;
    }
    return false;
}

// saturated Montgomery (x * 2^384 mod q, 12 x u32, canonical < q)  ->  U-form
ZK_HD FqU fqu_from_sat(const Fq &s) {
    FqU u;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const int bit = 29 * i;
        const int w = bit >> 5, off = bit & 31;
        uint64_t two = s.l[w];
        if (w + 1 < 12) two |= (uint64_t)s.l[w + 1] << 32;
        u.l[i] = (uint32_t)(two >> off) & FqU::MASK;
    }
    FqU c;
#pragma unroll
    for (int i = 0; i < 14; i++) c.l[i] = FqUP::c_in(i);
    if (u.is_zero()) return u;           // keep exact zeros exact (infinity flags)
    return fqu_mul(u, c);
}

// U-form (any value < 2^12 q)  ->  saturated Montgomery, canonical
ZK_HD Fq fqu_to_sat(const FqU &u) {
    FqU d;
#pragma unroll
    for (int i = 0; i < 14; i++) d.l[i] = FqUP::d_out(i);
    FqU v = fqu_mul(u, d);               // x * 2^384 mod q, in [0, 2q)
    // repack 14 x 29 -> 12 x 32 (value < 2q < 2^382)
    Fq s;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 32 * w;
        const int i = bit / 29, off = bit % 29;
        uint64_t acc = (uint64_t)v.l[i] >> off;
        int have = 29 - off;
        int j = i + 1;
        while (have < 32 && j < 14) {
            acc |= (uint64_t)v.l[j] << have;
            have += 29;
            j++;
        }
        s.l[w] = (uint32_t)acc;
    }
    fp_reduce_once(s);
    return s;
}

// ---- uniform free-function interface (see ec.cuh): Fq-like
ZK_HD FqU f_add(const FqU &a, const FqU &b) { return fqu_add(a, b); }
ZK_HD FqU f_sub(const FqU &a, const FqU &b) { return fqu_sub<32>(a, b); }     // b <= 31 q
ZK_HD FqU f_sub2(const FqU &a, const FqU &b) { return fqu_sub<64>(a, b); }    // b <= 63 q (stored coordinates)
ZK_HD FqU f_neg(const FqU &a) { return fqu_neg(a); }
ZK_HD FqU f_dbl(const FqU &a) { return fqu_dbl(a); }
ZK_HD FqU f_mul(const FqU &a, const FqU &b) { return fqu_mul(a, b); }
ZK_HD FqU f_sqr(const FqU &a) { return fqu_sqr(a); }
ZK_HD bool f_is_zero_mod(const FqU &a) { return fqu_is_zero_mod(a); }
// a*b - c*d for a stored coordinate c (<= 63 q): one reduction instead of two (fqu_mul2); result < 2q
ZK_HD FqU f_mul_sub(const FqU &a, const FqU &b, const FqU &c, const FqU &d) { return fqu_mul2(a, b, fqu_sub<64>(FqU::zero(), c), d); }

// ------------------------------------------------------------------------------------------------ Fq2 over U-form
struct Fq2U {
    FqU c0, c1;
    ZK_HD static Fq2U zero() { return Fq2U{FqU::zero(), FqU::zero()}; }
    ZK_HD static Fq2U one() { return Fq2U{FqU::one(), FqU::zero()}; }
    ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
};
ZK_HD Fq2U f_add(const Fq2U &a, const Fq2U &b) { return Fq2U{fqu_add(a.c0, b.c0), fqu_add(a.c1, b.c1)}; }
ZK_HD Fq2U f_sub(const Fq2U &a, const Fq2U &b) { return Fq2U{fqu_sub<32>(a.c0, b.c0), fqu_sub<32>(a.c1, b.c1)}; }
ZK_HD Fq2U f_sub2(const Fq2U &a, const Fq2U &b) { return Fq2U{fqu_sub<64>(a.c0, b.c0), fqu_sub<64>(a.c1, b.c1)}; }
ZK_HD Fq2U f_neg(const Fq2U &a) { return Fq2U{fqu_neg(a.c0), fqu_neg(a.c1)}; }
ZK_HD Fq2U f_dbl(const Fq2U &a) { return Fq2U{fqu_dbl(a.c0), fqu_dbl(a.c1)}; }
ZK_HD Fq2U f_mul(const Fq2U &a, const Fq2U &b) {
    // Karatsuba; component outputs < 2q + 8q
    const FqU v0 = fqu_mul(a.c0, b.c0);
    const FqU v1 = fqu_mul(a.c1, b.c1);
    const FqU s = fqu_mul(fqu_add(a.c0, a.c1), fqu_add(b.c0, b.c1));
    return Fq2U{fqu_sub<8>(v0, v1), fqu_sub<8>(s, fqu_add(v0, v1))};
}
// the same product with the three Fq products inlined (no call boundary): the last product of a G2 mixed addition, behind
// which the accumulation kernel hides the gather of the next base (ec.cuh: xyzz_madd_finish)
ZK_HD Fq2U fq2u_mul_inline(const Fq2U &a, const Fq2U &b) {
    const FqU v0 = fqu_mul_impl<false>(a.c0, b.c0);
    const FqU v1 = fqu_mul_impl<false>(a.c1, b.c1);
    const FqU s = fqu_mul_impl<false>(fqu_add(a.c0, a.c1), fqu_add(b.c0, b.c1));
    return Fq2U{fqu_sub<8>(v0, v1), fqu_sub<8>(s, fqu_add(v0, v1))};
}
#if defined(__HIP_DEVICE_COMPILE__)
// ---- Fq2 product with ONE reduction per component (bucket accumulation in G2, msm.hip):
//   c0 = a0 b0 + (-a1) b1,   c1 = a0 b1 + a1 b0   — two fqu_mul2 (four half-products, two reductions: the same 1,176
// multiply-adds as Karatsuba's three products, but two carry / quotient-digit passes instead of three, no sums of operands in
// front and no subtractions behind: ~1,390 instead of ~1,610 instructions, components < 2q instead of < 10q).
// fqu_mul2 takes four operands = 56 registers, the calling convention passes 32: two travel in registers, the other two are
// parked in LDS by the caller ([operand][quad][lane] x 16 bytes: one ds_write_b128 / ds_read_b128 per four limbs, conflict-free)
// and read back inside the callee while its first columns are already being multiplied.  Only kernels with ONE WAVE PER BLOCK may
// call this (the slot is indexed by the lane); 16 KiB of static LDS per block.
__shared__ uint4 zk_g2_opnd[4 * 4 * 64];
__device__ __forceinline__ void g2_park(int op, const FqU &v) {
    const unsigned lane = threadIdx.x & 63u;
    zk_g2_opnd[(op * 4 + 0) * 64 + lane] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    zk_g2_opnd[(op * 4 + 1) * 64 + lane] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    zk_g2_opnd[(op * 4 + 2) * 64 + lane] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
    zk_g2_opnd[(op * 4 + 3) * 64 + lane] = make_uint4(v.l[12], v.l[13], 0u, 0u);
}
__device__ __forceinline__ FqU g2_unpark(int op) {
    const unsigned lane = threadIdx.x & 63u;
    const uint4 q0 = zk_g2_opnd[(op * 4 + 0) * 64 + lane], q1 = zk_g2_opnd[(op * 4 + 1) * 64 + lane];
    const uint4 q2 = zk_g2_opnd[(op * 4 + 2) * 64 + lane], q3 = zk_g2_opnd[(op * 4 + 3) * 64 + lane];
    FqU v;
    v.l[0] = q0.x; v.l[1] = q0.y; v.l[2] = q0.z; v.l[3] = q0.w; v.l[4] = q1.x; v.l[5] = q1.y; v.l[6] = q1.z; v.l[7] = q1.w;
    v.l[8] = q2.x; v.l[9] = q2.y; v.l[10] = q2.z; v.l[11] = q2.w; v.l[12] = q3.x; v.l[13] = q3.y;
    return v;
}
// a b + c d with c, d = parked operands pair (0: slots 0, 1; 1: slots 2, 3)
__device__ __noinline__ __attribute__((weak)) FqURet fqu_mul2_lds_call(zk_v4u a0, zk_v4u a1, zk_v4u a2, zk_v2u a3, zk_v4u b0, zk_v4u b1, zk_v4u b2, zk_v2u b3, int pair) {
    FqU a, b;
    a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w; a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w;
    a.l[8] = a2.x; a.l[9] = a2.y; a.l[10] = a2.z; a.l[11] = a2.w; a.l[12] = a3.x; a.l[13] = a3.y;
    b.l[0] = b0.x; b.l[1] = b0.y; b.l[2] = b0.z; b.l[3] = b0.w; b.l[4] = b1.x; b.l[5] = b1.y; b.l[6] = b1.z; b.l[7] = b1.w;
    b.l[8] = b2.x; b.l[9] = b2.y; b.l[10] = b2.z; b.l[11] = b2.w; b.l[12] = b3.x; b.l[13] = b3.y;
    const FqU c = g2_unpark(2 * pair), d = g2_unpark(2 * pair + 1);
    const FqU r = fqu_mul2(a, b, c, d);
    FqURet o;
#pragma unroll
    for (int i = 0; i < 14; i++) o.l[i] = r.l[i];
    return o;
}
__device__ __forceinline__ FqU fqu_mul2_lds(const FqU &a, const FqU &b, int pair) {
    zk_v4u a0 = {a.l[0], a.l[1], a.l[2], a.l[3]}, a1 = {a.l[4], a.l[5], a.l[6], a.l[7]}, a2 = {a.l[8], a.l[9], a.l[10], a.l[11]};
    zk_v2u a3 = {a.l[12], a.l[13]};
    zk_v4u b0 = {b.l[0], b.l[1], b.l[2], b.l[3]}, b1 = {b.l[4], b.l[5], b.l[6], b.l[7]}, b2 = {b.l[8], b.l[9], b.l[10], b.l[11]};
    zk_v2u b3 = {b.l[12], b.l[13]};
    const FqURet r = fqu_mul2_lds_call(a0, a1, a2, a3, b0, b1, b2, b3, pair);
    FqU o;
#pragma unroll
    for (int i = 0; i < 14; i++) o.l[i] = r.l[i];
    return o;
}
// a, b: components normalised, <= 127 q (everything a mixed addition multiplies: coordinates, level-64 differences, products)
__device__ __forceinline__ Fq2U fq2u_mul_lazy(const Fq2U &a, const Fq2U &b) {
    g2_park(0, fqu_sub<128>(FqU::zero(), a.c1));
    g2_park(1, b.c1);
    g2_park(2, a.c1);
    g2_park(3, b.c0);
    Fq2U r;
    r.c0 = fqu_mul2_lds(a.c0, b.c0, 0);          // a0 b0 + (128 q - a1) b1
    r.c1 = fqu_mul2_lds(a.c0, b.c1, 1);          // a0 b1 + a1 b0
    return r;
}
#endif

ZK_HD Fq2U f_sqr(const Fq2U &a) {
    // (c0+c1)(c0-c1) + 2 c0 c1 u ; squared values are differences of stored coordinates (components < 74q, or 84q for
    // 2*Y in a doubling) -> level-128 subtraction
    const FqU p = fqu_mul(a.c0, a.c1);
    const FqU r0 = fqu_mul(fqu_add(a.c0, a.c1), fqu_sub<128>(a.c0, a.c1));
    return Fq2U{r0, fqu_dbl(p)};
}
ZK_HD bool f_is_zero_mod(const Fq2U &a) { return fqu_is_zero_mod(a.c0) && fqu_is_zero_mod(a.c1); }
ZK_HD Fq2U f_mul_sub(const Fq2U &a, const Fq2U &b, const Fq2U &c, const Fq2U &d) { return f_sub(f_mul(a, b), f_mul(c, d)); }

ZK_HD Fq2U fq2u_from_sat(const Fq2 &s) { return Fq2U{fqu_from_sat(s.c0), fqu_from_sat(s.c1)}; }
ZK_HD Fq2 fq2u_to_sat(const Fq2U &u) { return Fq2{fqu_to_sat(u.c0), fqu_to_sat(u.c1)}; }

// conversions by overload so templates can be written once
ZK_HD FqU to_u(const Fq &s) { return fqu_from_sat(s); }
ZK_HD Fq2U to_u(const Fq2 &s) { return fq2u_from_sat(s); }
ZK_HD Fq to_sat(const FqU &u) { return fqu_to_sat(u); }
ZK_HD Fq2 to_sat(const Fq2U &u) { return fq2u_to_sat(u); }

// ---- inversion (setup's batched to-affine only: one per thread per batch, never in the MSM inner loops)
ZK_HD FqU fqu_inv(const FqU &a) {          // a^(q-2); a < 2^12 q, result < 2q
    uint32_t e[12];
    uint32_t borrow = 2;
    for (int i = 0; i < 12; i++) {
        uint64_t t = (uint64_t)FqP::mod(i) - borrow;
        e[i] = (uint32_t)t;
        borrow = (uint32_t)(t >> 63);
    }
    FqU acc = a;                             // top bit of q - 2 (bit 380) is set
    for (int i = 379; i >= 0; i--) {
        acc = fqu_sqr(acc);
        if ((e[i / 32] >> (i % 32)) & 1) acc = fqu_mul(acc, a);
    }
    return acc;
}
ZK_HD FqU f_inv(const FqU &a) { return fqu_inv(a); }
ZK_HD Fq2U f_inv(const Fq2U &a) {           // conj(a) / (c0^2 + c1^2); components of a <= 31q
    const FqU n = fqu_inv(fqu_add(fqu_sqr(a.c0), fqu_sqr(a.c1)));
    return Fq2U{fqu_mul(a.c0, n), fqu_mul(fqu_sub<32>(FqU::zero(), a.c1), n)};
}
// bring every component back below 2q (a product by one): what stored proving-key coordinates must satisfy
ZK_HD FqU f_tidy(const FqU &a) { return a; }                 // callers pass products, already < 2q
ZK_HD Fq2U f_tidy(const Fq2U &a) { return Fq2U{fqu_mul(a.c0, FqU::one()), fqu_mul(a.c1, FqU::one())}; }

// saturated types: second-level subtraction and the mod test are the plain ones
ZK_HD Fq f_sub2(const Fq &a, const Fq &b) { return fp_sub(a, b); }
ZK_HD Fq2 f_sub2(const Fq2 &a, const Fq2 &b) { return f_sub(a, b); }
ZK_HD Fq f_mul_sub(const Fq &a, const Fq &b, const Fq &c, const Fq &d) { return f_sub(f_mul(a, b), f_mul(c, d)); }
ZK_HD Fq2 f_mul_sub(const Fq2 &a, const Fq2 &b, const Fq2 &c, const Fq2 &d) { return f_sub(f_mul(a, b), f_mul(c, d)); }
ZK_HD bool f_is_zero_mod(const Fq &a) { return a.is_zero(); }
ZK_HD bool f_is_zero_mod(const Fq2 &a) { return a.is_zero(); }

}  // namespace zk
