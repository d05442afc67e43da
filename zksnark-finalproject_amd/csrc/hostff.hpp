// Host-only BLS12-381 field arithmetic on 64-bit limbs (Fr: 4, Fq: 6), Montgomery form — the same bytes as ff.cuh's
// Fp<P> (2k little-endian u32 limbs = k little-endian u64 limbs) and as arkworks' `Fp<MontBackend<_, 4|6>>`, so values
// move between the two representations by memcpy.
//
// Why a second implementation: ff.cuh is written for the GPU's 32x32 multiplier; compiled for the host its additions
// walk 32-bit limbs and its product repacks operands on every call (measured here: 46 ns per dependent Fr product +
// addition, 83 ns in Fq).  The two host-side consumers that are latency chains of field products —
//   * the native Poseidon sponges in front of the device witness generator (witness.hip; the reference's
//     hasher.rs:17-27 / hashing_utils.rs:737-877 run sequentially by construction: permutation p+1 needs p's output),
//   * the pairing verifier (verify.hip; matrix_proof.rs:200-205)
// — run on this header instead: fully unrolled CIOS with the "no-carry" shortcut both moduli allow (top bit clear), the
// same algorithm ark-ff 0.4's `MontBackend::mul_assign` uses for these fields.
#pragma once
#include <stdint.h>
#include <string.h>

#include "ff.cuh"

namespace zk {
namespace h64 {

typedef unsigned __int128 u128;

template <class P>
struct F {
    static constexpr int M = P::N / 2;
    uint64_t l[M];

    static constexpr uint64_t mod(int i) { return (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32); }
    static F zero() {
        F r;
        for (int i = 0; i < M; i++) r.l[i] = 0;
        return r;
    }
    static F one() {
        F r;
        for (int i = 0; i < M; i++) r.l[i] = (uint64_t)P::one(2 * i) | ((uint64_t)P::one(2 * i + 1) << 32);
        return r;
    }
    static F r2() {
        F r;
        for (int i = 0; i < M; i++) r.l[i] = (uint64_t)P::r2(2 * i) | ((uint64_t)P::r2(2 * i + 1) << 32);
        return r;
    }
    static F from(const Fp<P> &a) {
        F r;
        memcpy(r.l, a.l, sizeof r.l);
        return r;
    }
    Fp<P> to() const {
        Fp<P> r;
        memcpy(r.l, l, sizeof l);
        return r;
    }
    bool is_zero() const {
        uint64_t acc = 0;
        for (int i = 0; i < M; i++) acc |= l[i];
        return acc == 0;
    }
    bool operator==(const F &o) const {
        uint64_t acc = 0;
        for (int i = 0; i < M; i++) acc |= l[i] ^ o.l[i];
        return acc == 0;
    }
    bool operator!=(const F &o) const { return !(*this == o); }
};

// a - p if a >= p (a < 2p): the difference is kept when it does not borrow
template <class P>
inline void reduce_once(F<P> &a) {
    constexpr int M = F<P>::M;
    uint64_t d[M];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < M; i++) {
        const u128 t = (u128)a.l[i] - F<P>::mod(i) - borrow;
        d[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 127);
    }
#pragma unroll
    for (int i = 0; i < M; i++) a.l[i] = borrow ? a.l[i] : d[i];
}

template <class P>
inline F<P> add(const F<P> &a, const F<P> &b) {
    constexpr int M = F<P>::M;
    F<P> r;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < M; i++) {
        const u128 t = (u128)a.l[i] + b.l[i] + carry;
        r.l[i] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
    }
    reduce_once(r);      // 2p < 2^(64M): no carry out of the top limb
    return r;
}

template <class P>
inline F<P> sub(const F<P> &a, const F<P> &b) {
    constexpr int M = F<P>::M;
    F<P> r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < M; i++) {
        const u128 t = (u128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 127);
    }
    const uint64_t mask = 0 - borrow;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < M; i++) {
        const u128 t = (u128)r.l[i] + (F<P>::mod(i) & mask) + carry;
        r.l[i] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
    }
    return r;
}

template <class P>
inline F<P> neg(const F<P> &a) {
    return a.is_zero() ? a : sub(F<P>::zero(), a);
}
template <class P>
inline F<P> dbl(const F<P> &a) {
    return add(a, a);
}

// Montgomery product a b R^-1 mod p.  CIOS; because the modulus' top bit is clear the two carry chains of a row never
// overflow a limb together ("no-carry" form), so a row is 2M multiply-accumulates and nothing else.
template <class P>
inline F<P> mul(const F<P> &a, const F<P> &b) {
    constexpr int M = F<P>::M;
    uint64_t t[M];
#pragma unroll
    for (int i = 0; i < M; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < M; i++) {
        const uint64_t bi = b.l[i];
        u128 x = (u128)a.l[0] * bi + t[0];
        uint64_t c1 = (uint64_t)(x >> 64);
        const uint64_t k = (uint64_t)x * P::INV64;
        u128 y = (u128)k * F<P>::mod(0) + (uint64_t)x;
        uint64_t c2 = (uint64_t)(y >> 64);
#pragma unroll
        for (int j = 1; j < M; j++) {
            x = (u128)a.l[j] * bi + t[j] + c1;
            c1 = (uint64_t)(x >> 64);
            y = (u128)k * F<P>::mod(j) + (uint64_t)x + c2;
            c2 = (uint64_t)(y >> 64);
            t[j - 1] = (uint64_t)y;
        }
        t[M - 1] = c1 + c2;
    }
    F<P> r;
#pragma unroll
    for (int i = 0; i < M; i++) r.l[i] = t[i];
    reduce_once(r);
    return r;
}
template <class P>
inline F<P> sqr(const F<P> &a) {
    return mul(a, a);
}

template <class P>
inline F<P> to_mont(const F<P> &canon) {
    return mul(canon, F<P>::r2());
}
template <class P>
inline F<P> from_mont(const F<P> &a) {
    F<P> one = F<P>::zero();
    one.l[0] = 1;
    return mul(a, one);
}

// a^(p-2)
template <class P>
inline F<P> inv(const F<P> &a) {
    constexpr int M = F<P>::M;
    uint64_t e[M];
    uint64_t borrow = 2;
    for (int i = 0; i < M; i++) {
        const u128 t = (u128)F<P>::mod(i) - borrow;
        e[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 127);
    }
    F<P> acc = F<P>::one();
    bool started = false;
    for (int i = M * 64 - 1; i >= 0; i--) {
        if (started) acc = sqr(acc);
        if ((e[i / 64] >> (i % 64)) & 1) {
            acc = started ? mul(acc, a) : a;
            started = true;
        }
    }
    return acc;
}

using Fr64 = F<FrP>;
using Fq64 = F<FqP>;

}  // namespace h64
}  // namespace zk
