// BLS12-381 G1 (over Fq) and G2 (over Fq2) group arithmetic, y^2 = x^3 + b, a = 0.
//
// Device representation for bucket sums is extended Jacobian "XYZZ" (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// the mixed addition bucket += affine costs 8M + 2S (EFD madd-2008-s) versus 7M + 4S for Jacobian, and
// needs no inversion.  Infinity <=> ZZ == 0.  Affine points coming from the proving key use
// (x, y) = (0, 0) for the point at infinity (not on either curve), set at pk-load time from the
// caller's flag bytes (arkworks `Affine { x, y, infinity }`, see include/zkg16.h).
//
// The group element is what must match the reference (Proof{a, b, c} are affine, canonical); the
// projective representation is free.
#pragma once
#include "ff.cuh"
#include "ffu.cuh"

namespace zk {

template <class F>
struct Affine {
    F x, y;
    ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    ZK_HD static Affine inf() { return Affine{F::zero(), F::zero()}; }
};

template <class F>
struct XYZZ {
    F x, y, zz, zzz;
    ZK_HD bool is_inf() const { return zz.is_zero(); }
    ZK_HD static XYZZ inf() { return XYZZ{F::zero(), F::zero(), F::zero(), F::zero()}; }
    ZK_HD static XYZZ from_affine(const Affine<F> &p) {
        if (p.is_inf()) return inf();
        return XYZZ{p.x, p.y, F::one(), F::one()};
    }
};

// 2 * (affine p), p != inf  (EFD mdbl-2008-s-1)
template <class F>
ZK_HD XYZZ<F> xyzz_dbl_affine(const Affine<F> &p) {
    F U = f_dbl(p.y);
    F V = f_sqr(U);
    F W = f_mul(U, V);
    F S = f_mul(p.x, V);
    F X2 = f_sqr(p.x);
    F M = f_add(f_dbl(X2), X2);
    XYZZ<F> r;
    r.x = f_sub(f_sqr(M), f_dbl(S));
    r.y = f_sub(f_mul(M, f_sub2(S, r.x)), f_mul(W, p.y));
    r.zz = V;
    r.zzz = W;
    return r;
}

// 2 * p (EFD dbl-2008-s-1)
template <class F>
ZK_HD XYZZ<F> xyzz_dbl(const XYZZ<F> &p) {
    if (p.is_inf()) return p;
    F U = f_dbl(p.y);
    F V = f_sqr(U);
    F W = f_mul(U, V);
    F S = f_mul(p.x, V);
    F X2 = f_sqr(p.x);
    F M = f_add(f_dbl(X2), X2);
    XYZZ<F> r;
    r.x = f_sub(f_sqr(M), f_dbl(S));
    r.y = f_sub(f_mul(M, f_sub2(S, r.x)), f_mul(W, p.y));
    r.zz = f_mul(V, p.zz);
    r.zzz = f_mul(W, p.zzz);
    return r;
}

// acc += (neg ? -q : q), q affine (EFD madd-2008-s) with all exceptional cases.
template <class F>
ZK_HD void xyzz_madd(XYZZ<F> &acc, const Affine<F> &q_in, bool neg) {
    if (q_in.is_inf()) return;
    Affine<F> q = q_in;
    if (neg) q.y = f_neg(q.y);
    if (acc.is_inf()) {
        acc = XYZZ<F>{q.x, q.y, F::one(), F::one()};
        return;
    }
    F U2 = f_mul(q.x, acc.zz);
    F S2 = f_mul(q.y, acc.zzz);
    F Pp = f_sub2(U2, acc.x);
    F R = f_sub2(S2, acc.y);
    if (f_is_zero_mod(Pp)) {
        if (f_is_zero_mod(R)) acc = xyzz_dbl_affine(q);
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP = f_sqr(Pp);
    F PPP = f_mul(Pp, PP);
    F Q = f_mul(acc.x, PP);
    F X3 = f_sub(f_sqr(R), f_add(PPP, f_dbl(Q)));
    acc.y = f_mul_sub(R, f_sub2(Q, X3), acc.y, PPP);      // R (Q - X3) - Y1 PPP: one reduction in the unsaturated G1 form
    acc.x = X3;
    acc.zz = f_mul(acc.zz, PP);
    acc.zzz = f_mul(acc.zzz, PPP);
}

#if defined(__HIP_DEVICE_COMPILE__)
// The same mixed addition for G2 with the eight Fq2 products in the one-reduction-per-component form (ffu.cuh: fq2u_mul_lazy).
// Only for kernels with one wave per block (the bucket accumulation).
__device__ __forceinline__ void xyzz_madd_lazy(XYZZ<Fq2U> &acc, const Affine<Fq2U> &q_in, bool neg) {
    if (q_in.is_inf()) return;
    Affine<Fq2U> q = q_in;
    if (neg) q.y = f_neg(q.y);
    if (acc.is_inf()) {
        acc = XYZZ<Fq2U>{q.x, q.y, Fq2U::one(), Fq2U::one()};
        return;
    }
    Fq2U U2 = fq2u_mul_lazy(q.x, acc.zz);
    Fq2U S2 = fq2u_mul_lazy(q.y, acc.zzz);
    Fq2U Pp = f_sub2(U2, acc.x);
    Fq2U R = f_sub2(S2, acc.y);
    if (f_is_zero_mod(Pp)) {
        if (f_is_zero_mod(R)) acc = xyzz_dbl_affine(q);
        else acc = XYZZ<Fq2U>::inf();
        return;
    }
    Fq2U PP = f_sqr(Pp);
    Fq2U PPP = fq2u_mul_lazy(Pp, PP);
    Fq2U Q = fq2u_mul_lazy(acc.x, PP);
    Fq2U X3 = f_sub(f_sqr(R), f_add(PPP, f_dbl(Q)));
    acc.y = f_sub(fq2u_mul_lazy(R, f_sub2(Q, X3)), fq2u_mul_lazy(acc.y, PPP));
    acc.x = X3;
    acc.zz = fq2u_mul_lazy(acc.zz, PP);
    acc.zzz = fq2u_mul_lazy(acc.zzz, PPP);
}
#else
__host__ __device__ inline void xyzz_madd_lazy(XYZZ<Fq2U> &acc, const Affine<Fq2U> &q, bool neg) { xyzz_madd(acc, q, neg); }      // host pass of the kernel template only
#endif

// acc += q (EFD add-2008-s) with all exceptional cases.
template <class F>
ZK_HD void xyzz_add(XYZZ<F> &acc, const XYZZ<F> &q) {
    if (q.is_inf()) return;
    if (acc.is_inf()) {
        acc = q;
        return;
    }
    F U1 = f_mul(acc.x, q.zz);
    F U2 = f_mul(q.x, acc.zz);
    F S1 = f_mul(acc.y, q.zzz);
    F S2 = f_mul(q.y, acc.zzz);
    F Pp = f_sub(U2, U1);
    F R = f_sub(S2, S1);
    if (f_is_zero_mod(Pp)) {
        if (f_is_zero_mod(R)) acc = xyzz_dbl(acc);
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP = f_sqr(Pp);
    F PPP = f_mul(Pp, PP);
    F Q = f_mul(U1, PP);
    F X3 = f_sub(f_sqr(R), f_add(PPP, f_dbl(Q)));
    acc.y = f_sub(f_mul(R, f_sub2(Q, X3)), f_mul(S1, PPP));
    acc.x = X3;
    acc.zz = f_mul(f_mul(acc.zz, q.zz), PP);
    acc.zzz = f_mul(f_mul(acc.zzz, q.zzz), PPP);
}

// ---- the mixed addition in two parts for the bucket-accumulation kernels (msm.hip).  Every field product is a device-function
// call except ONE, which is inlined and done last (`finish`): a call boundary drains all outstanding memory operations
// (s_waitcnt vmcnt(0) at every function entry), so the gather of the NEXT base can only stay in flight across call-free code.
// The kernels issue that gather between `front` and `finish`; the ~600 inlined instructions of the last product hide it.
//   G1 (FqU):  finish = Y3 = R (Q - X3) + (-Y1) PPP   (the fused two-product reduction)
//   G2 (Fq2U): finish = ZZZ3 = ZZZ1 * PPP            (one Fq2 product, three inlined Fq products)
// front() does everything else, including all exceptional cases (then it returns false and finish() changes nothing).
template <class F> struct MaddTail { F a, b, c, d; };

// ZK_G1_INLINE_FRONT: the nine products of the G1 front part inlined too (no call at all on the main path; ~45 KB of loop body)
#ifndef ZK_G1_INLINE_FRONT
#define ZK_G1_INLINE_FRONT 0
#endif
#if ZK_G1_INLINE_FRONT
#define ZK_G1M(a, b) fqu_mul_impl<false>(a, b)
#define ZK_G1S(a) fqu_mul_impl<true>(a, a)
#else
#define ZK_G1M(a, b) f_mul(a, b)
#define ZK_G1S(a) f_sqr(a)
#endif
ZK_HD bool xyzz_madd_front(XYZZ<FqU> &acc, const Affine<FqU> &q_in, bool neg, MaddTail<FqU> &t) {
    if (q_in.is_inf()) { t.a = t.b = t.c = t.d = FqU::zero(); return false; }
    Affine<FqU> q = q_in;
    if (neg) q.y = f_neg(q.y);
    if (acc.is_inf()) {
        acc = XYZZ<FqU>{q.x, q.y, FqU::one(), FqU::one()};
        t.a = t.b = t.c = t.d = FqU::zero();
        return false;
    }
    FqU U2 = ZK_G1M(q.x, acc.zz);
    FqU S2 = ZK_G1M(q.y, acc.zzz);
    FqU Pp = f_sub2(U2, acc.x);
    FqU R = f_sub2(S2, acc.y);
    if (f_is_zero_mod(Pp)) {
        if (f_is_zero_mod(R)) acc = xyzz_dbl_affine(q);
        else acc = XYZZ<FqU>::inf();
        t.a = t.b = t.c = t.d = FqU::zero();
        return false;
    }
    // order chosen for register pressure: at most six field elements are live across any call (values that survive a call
    // must sit in the callee-saved half of the register file)
    FqU PP = ZK_G1S(Pp);
    FqU PPP = ZK_G1M(Pp, PP);
    acc.zz = ZK_G1M(acc.zz, PP);
    acc.zzz = ZK_G1M(acc.zzz, PPP);
    FqU Q = ZK_G1M(acc.x, PP);
    FqU X3 = f_sub(ZK_G1S(R), f_add(PPP, f_dbl(Q)));
    t.a = R;
    t.b = f_sub2(Q, X3);
    t.c = fqu_sub<64>(FqU::zero(), acc.y);
    t.d = PPP;
    acc.x = X3;
    return true;
}
ZK_HD void xyzz_madd_finish(XYZZ<FqU> &acc, const MaddTail<FqU> &t, bool normal) {
    const FqU y = fqu_mul2(t.a, t.b, t.c, t.d);
    if (normal) acc.y = y;
}

ZK_HD bool xyzz_madd_front(XYZZ<Fq2U> &acc, const Affine<Fq2U> &q_in, bool neg, MaddTail<Fq2U> &t) {
    if (q_in.is_inf()) { t.a = t.b = Fq2U::zero(); return false; }
    Affine<Fq2U> q = q_in;
    if (neg) q.y = f_neg(q.y);
    if (acc.is_inf()) {
        acc = XYZZ<Fq2U>{q.x, q.y, Fq2U::one(), Fq2U::one()};
        t.a = t.b = Fq2U::zero();
        return false;
    }
    // the first products inlined as well (ZK_G2_INLINE_FRONT of them): at one wave per SIMD a call's argument moves and waits are
    // not hidden by a partner wave
#ifndef ZK_G2_INLINE_FRONT
#define ZK_G2_INLINE_FRONT 0      // measured at 128x128: 46.5 ms (0), 46.9 (2), 48.0 (5): the tail product alone is best
#endif
    Fq2U U2 = ZK_G2_INLINE_FRONT >= 1 ? fq2u_mul_inline(q.x, acc.zz) : f_mul(q.x, acc.zz);
    Fq2U S2 = ZK_G2_INLINE_FRONT >= 2 ? fq2u_mul_inline(q.y, acc.zzz) : f_mul(q.y, acc.zzz);
    Fq2U Pp = f_sub2(U2, acc.x);
    Fq2U R = f_sub2(S2, acc.y);
    if (f_is_zero_mod(Pp)) {
        if (f_is_zero_mod(R)) acc = xyzz_dbl_affine(q);
        else acc = XYZZ<Fq2U>::inf();
        t.a = t.b = Fq2U::zero();
        return false;
    }
    Fq2U PP = f_sqr(Pp);
    Fq2U PPP = ZK_G2_INLINE_FRONT >= 3 ? fq2u_mul_inline(Pp, PP) : f_mul(Pp, PP);
    acc.zz = ZK_G2_INLINE_FRONT >= 4 ? fq2u_mul_inline(acc.zz, PP) : f_mul(acc.zz, PP);
    Fq2U Q = ZK_G2_INLINE_FRONT >= 5 ? fq2u_mul_inline(acc.x, PP) : f_mul(acc.x, PP);
    Fq2U X3 = f_sub(f_sqr(R), f_add(PPP, f_dbl(Q)));
    acc.y = f_sub(f_mul(R, f_sub2(Q, X3)), f_mul(acc.y, PPP));
    acc.x = X3;
    t.a = acc.zzz;
    t.b = PPP;
    return true;
}
ZK_HD void xyzz_madd_finish(XYZZ<Fq2U> &acc, const MaddTail<Fq2U> &t, bool normal) {
    const Fq2U z = fq2u_mul_inline(t.a, t.b);
    if (normal) acc.zzz = z;
}

template <class F>
ZK_HD XYZZ<F> xyzz_neg(const XYZZ<F> &p) {
    return XYZZ<F>{p.x, f_neg(p.y), p.zz, p.zzz};
}

// x = X/ZZ, y = Y/ZZZ.  One inversion: (ZZ*ZZZ)^-1.  Host tail only.
template <class F>
ZK_HD Affine<F> xyzz_to_affine(const XYZZ<F> &p) {
    if (p.is_inf()) return Affine<F>::inf();
    F inv = f_inv(f_mul(p.zz, p.zzz));
    F zz_inv = f_mul(inv, p.zzz);
    F zzz_inv = f_mul(inv, p.zz);
    return Affine<F>{f_mul(p.x, zz_inv), f_mul(p.y, zzz_inv)};
}

// [k]p for a 256-bit little-endian scalar (8 x u32, canonical).  Host tail + tests.
template <class F>
ZK_HD XYZZ<F> xyzz_mul(const XYZZ<F> &p, const uint32_t k[8]) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int i = 255; i >= 0; i--) {
        acc = xyzz_dbl(acc);
        if ((k[i / 32] >> (i % 32)) & 1) xyzz_add(acc, p);
    }
    return acc;
}

// value-bound bookkeeping for the unsaturated types (ffu.cuh): products are < 2q (Fq) / < 10q per component (Fq2);
// stored X, Y are one level-32 subtraction away from a product (< 42q); f_sub2 (level 64) is used wherever the
// subtrahend is a stored coordinate.  For the saturated types f_sub2 == f_sub.
using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1XYZZ = XYZZ<Fq>;
using G2XYZZ = XYZZ<Fq2>;
using G1AffineU = Affine<FqU>;   // device-resident proving-key bases / bucket sums (unsaturated form)
using G2AffineU = Affine<Fq2U>;

}  // namespace zk
