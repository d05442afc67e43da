// BLS12-381 Fr in an UNSATURATED radix-2^29 representation for the NTT butterflies (9 x 29-bit limbs in u32) — the
// same idea as ffu.cuh's FqU: on gfx950 a saturated 8-limb Montgomery product compiles to 128 v_mad_u64_u32 plus ~160
// 64-bit adds and ~360 register moves (carry chains need even-aligned VGPR pairs); with 29-bit limbs a column of
// partial products accumulates in one 64-bit register by back-to-back v_mad_u64_u32 and is carried once:
// 2 * 81 multiply-adds and ~60 other instructions.  r = 1 mod 2^29, so the Montgomery quotient digit is just the
// negated low limb (no multiplication).
//
// "U-form": x held as x * 2^261 mod r, any representative in [0, 2r) unless stated otherwise, limbs < 2^29.
//   fru_mul(a, b): a * b * 2^-261 mod r for a * b < 70 r^2 (2^261 / r = 70.66), result < 2r, no final subtraction.
// Memory keeps arkworks' saturated Montgomery form (x * 2^256 mod r, canonical, 8 x u32: what the ABI, the SpMV, the
// point-wise kernels and the MSM digit extraction use); conversions ride on multiplications the transform performs anyway:
//   load : repack(x 2^256) (*) C266            = x 2^261            (C266 = 2^266 mod r)
//   store: v 2^261 (*) repack(t 2^256)         = v t 2^256          (t = the inter-pass twiddle / coset or 1/N factor / 1)
// where (*) is fru_mul and repack() only moves bits (8 x 32 -> 9 x 29).
#pragma once
#include "ff.cuh"

namespace zk {

struct FrU {
    static constexpr uint32_t MASK = (1u << 29) - 1u;
    uint32_t l[9];
};

struct FrUP {
    ZK_HD static constexpr uint32_t mod(int i) {
        constexpr uint32_t M[9] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
        return M[i];
    }
    ZK_HD static constexpr uint32_t two_r(int i) {
        constexpr uint32_t M[9] = {0x00000002u, 0x1ffffff0u, 0x1f2dff7fu, 0x16900bffu, 0x1b00aa77u, 0x180809a1u, 0x0a4199ceu, 0x14ca675fu, 0x00e7db4eu};
        return M[i];
    }
    // 2r with limbs 0..7 pre-borrowed to [2^29 - 1, 2^30): limb-wise "2r - b" never underflows for normalised b < 2r
    ZK_HD static constexpr uint32_t m2(int i) {
        constexpr uint32_t M[9] = {0x20000002u, 0x3fffffefu, 0x3f2dff7eu, 0x36900bfeu, 0x3b00aa76u, 0x380809a0u, 0x2a4199cdu, 0x34ca675eu, 0x00e7db4du};
        return M[i];
    }
    // 4r with limbs 0..7 pre-borrowed: limb-wise "4r - b" never underflows for normalised b < 4r (butterflies with lazy sums)
    ZK_HD static constexpr uint32_t m4(int i) {
        constexpr uint32_t M[9] = {0x20000004u, 0x3fffffdfu, 0x3e5bfefeu, 0x2d2017feu, 0x360154eeu, 0x30101342u, 0x3483339cu, 0x2994cebdu, 0x01cfb69cu};
        return M[i];
    }
    ZK_HD static constexpr uint32_t c266(int i) {      // 2^266 mod r
        constexpr uint32_t M[9] = {0x1ffff72bu, 0x000046a7u, 0x1f5f3540u, 0x0ce3021cu, 0x118f3661u, 0x008176cbu, 0x054e487cu, 0x102e8190u, 0x001e092eu};
        return M[i];
    }
    ZK_HD static constexpr uint32_t c271(int i) {      // 2^271 mod r
        constexpr uint32_t M[9] = {0x1ffee558u, 0x0008d53fu, 0x0f2eaa00u, 0x0220139fu, 0x05e4224eu, 0x100eb2eau, 0x00c2a845u, 0x12a69488u, 0x0021b895u};
        return M[i];
    }
    ZK_HD static constexpr uint32_t one_sat(int i) {   // 2^256 mod r (the saturated form's one), repacked
        constexpr uint32_t M[9] = {0x1ffffffeu, 0x0000000fu, 0x00d20080u, 0x096ff400u, 0x04ff5588u, 0x07f7f65eu, 0x15be6631u, 0x0b3598a0u, 0x001824b1u};
        return M[i];
    }
};

ZK_HD FrU fru_c266() { FrU c; for (int i = 0; i < 9; i++) c.l[i] = FrUP::c266(i); return c; }
ZK_HD FrU fru_c271() { FrU c; for (int i = 0; i < 9; i++) c.l[i] = FrUP::c271(i); return c; }
ZK_HD FrU fru_one_sat() { FrU c; for (int i = 0; i < 9; i++) c.l[i] = FrUP::one_sat(i); return c; }

ZK_HD void fru_normalise(FrU &a) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t v = a.l[i] + c;
        a.l[i] = v & FrU::MASK;
        c = v >> 29;
    }
    a.l[8] += c;
}

// a * b * 2^-261 mod r; limbs of a, b < 2^30 (need not be normalised; or a < 3 * 2^29 with b normalised), a * b < 70 r^2; result
// normalised, < 2r
ZK_HD FrU fru_mul(const FrU &a, const FrU &b) {
    constexpr int N = 9;
    uint32_t m[N];
    FrU r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; k++) {
        const int lo = k < N ? 0 : k - N + 1;
        const int hi = k < N ? k : N - 1;
#pragma unroll
        for (int i = lo; i <= hi; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
        if (k < N) {
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * FrUP::mod(k - i);
            m[k] = (0u - (uint32_t)acc) & FrU::MASK;          // -r^-1 = -1 mod 2^29
            acc += m[k];                                      // m[k] * mod(0), mod(0) = 1
            acc >>= 29;
        } else {
#pragma unroll
            for (int i = lo; i <= hi; i++) acc += (uint64_t)m[i] * FrUP::mod(k - i);
            r.l[k - N] = (uint32_t)acc & FrU::MASK;
            acc >>= 29;
        }
    }
    r.l[N - 1] = (uint32_t)acc;
    return r;
}

// a + b, limbs normalised
ZK_HD FrU fru_add(const FrU &a, const FrU &b) {
    FrU r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    fru_normalise(r);
    return r;
}
// a - b + 2r for normalised b < 2r; result normalised, in (0, 2r + a]
ZK_HD FrU fru_sub_2r(const FrU &a, const FrU &b) {
    FrU r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (FrUP::m2(i) - b.l[i]);
    fru_normalise(r);
    return r;
}
// ---- the butterfly's own two operations (ntt.hip).  Values on the butterfly network are kept below 2r + e, e < 2^249:
// a + b, minus 2r when the two TOP LIMBS already say the sum has reached 2r (no borrow chain, no second compare): one signed
// carry pass does the addition, the subtraction and the normalisation.  With T = top limb of 2r: not subtracted -> the sum is
// below (T + 3) 2^232 < 2r + 2^234; subtracted -> it was at least (T + 2) 2^232 > 2r, so the result is positive and below the
// sum of the inputs' excesses + 2r — the excess e at most doubles per stage (2^234 -> 2^246 over the 12 stages of a pass) and
// every pass ends in a product, which brings the value back below 2r.
ZK_HD FrU fru_add_lazy(const FrU &a, const FrU &b) {
    const uint32_t top = a.l[8] + b.l[8];
    const bool sub = top > FrUP::two_r(8) + 1u;
    FrU r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int32_t v = (int32_t)(a.l[i] + b.l[i]) - (int32_t)(sub ? FrUP::two_r(i) : 0u) + c;
        r.l[i] = (uint32_t)v & FrU::MASK;
        c = v >> 29;                       // arithmetic shift: floor
    }
    r.l[8] = (uint32_t)((int32_t)top - (int32_t)(sub ? FrUP::two_r(8) : 0u) + c);
    return r;
}
// a - b + 4r for normalised b < 4r, NOT normalised (limbs < 3 * 2^29): goes straight into fru_mul, whose columns hold
// 9 x (3 * 2^29 x 2^29) + 9 x 2^58 < 2^63 with a normalised second operand
ZK_HD FrU fru_sub_4r_raw(const FrU &a, const FrU &b) {
    FrU r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (FrUP::m4(i) - b.l[i]);
    return r;
}

// x >= c ? x - c : x   for normalised x, c = 2r (TWO = true) or r
template <bool TWO>
ZK_HD FrU fru_cond_sub(const FrU &x) {
    FrU t;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int32_t d = (int32_t)x.l[i] - (int32_t)(TWO ? FrUP::two_r(i) : FrUP::mod(i)) + borrow;
        t.l[i] = (uint32_t)d & FrU::MASK;
        borrow = d >> 29;                  // arithmetic shift: 0 or -1
    }
    const int32_t top = (int32_t)x.l[8] - (int32_t)(TWO ? FrUP::two_r(8) : FrUP::mod(8)) + borrow;      // not a 29-bit digit
    t.l[8] = (uint32_t)top;
    FrU r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = top < 0 ? x.l[i] : t.l[i];
    return r;
}

// 8 x 32 -> 9 x 29: the same integer, only moved
ZK_HD FrU fru_repack(const Fr &s) {
    FrU u;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i;
        const int w = bit >> 5, off = bit & 31;
        uint64_t two = s.l[w];
        if (w + 1 < 8) two |= (uint64_t)s.l[w + 1] << 32;
        u.l[i] = (uint32_t)(two >> off) & FrU::MASK;
    }
    return u;
}
// 9 x 29 (normalised, value < 2^256) -> 8 x 32
ZK_HD Fr fru_unpack(const FrU &v) {
    Fr s;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int bit = 32 * w;
        const int i = bit / 29, off = bit % 29;
        uint64_t acc = (uint64_t)v.l[i] >> off;
        int have = 29 - off;
        int j = i + 1;
        while (have < 32 && j < 9) {
            acc |= (uint64_t)v.l[j] << have;
            have += 29;
            j++;
        }
        s.l[w] = (uint32_t)acc;
    }
    return s;
}
// saturated Montgomery (canonical) -> U-form
ZK_HD FrU fru_from_sat(const Fr &s) { return fru_mul(fru_repack(s), fru_c266()); }
// U-form value v (< 2r) times a saturated-form factor t -> saturated Montgomery form of the product, canonical
ZK_HD Fr fru_mul_to_sat(const FrU &v, const FrU &t_sat_repacked) {
    return fru_unpack(fru_cond_sub<false>(fru_mul(v, t_sat_repacked)));
}

}  // namespace zk
