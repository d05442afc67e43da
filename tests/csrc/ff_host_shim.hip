// TEST-ONLY: exposes the __host__ side of csrc/ff.cuh + csrc/ec.cuh (the exact templates the kernels
// instantiate) so the limb arithmetic and curve formulas can be checked against the golden vectors on
// a machine without a GPU.  Built by tests/test_ff_host.py with `hipcc --offload-host-only`.
#include "ff.cuh"
#include "ec.cuh"
#include "fru.cuh"
#include <string.h>
using namespace zk;

template <class F> static F ld(const uint32_t *p) { F r; memcpy(&r, p, sizeof(F)); return r; }
template <class F> static void st(uint32_t *p, const F &v) { memcpy(p, &v, sizeof(F)); }

template <class F> static void field_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *o) {
    F x = ld<F>(a), y = ld<F>(b), r;
    switch (op) {
        case 0: r = f_add(x, y); break;
        case 1: r = f_sub(x, y); break;
        case 2: r = f_mul(x, y); break;
        case 3: r = f_sqr(x); break;
        case 4: r = f_inv(x); break;
        case 5: r = f_neg(x); break;
        default: r = x;
    }
    st(o, r);
}
template <class F> static void point_op(int op, const uint32_t *acc_in, const uint32_t *q, const uint32_t *k, int neg, uint32_t *out_xyzz, uint32_t *out_aff) {
    XYZZ<F> acc = ld<XYZZ<F>>(acc_in);
    switch (op) {
        case 0: xyzz_madd(acc, ld<Affine<F>>(q), neg != 0); break;
        case 1: xyzz_add(acc, ld<XYZZ<F>>(q)); break;
        case 2: acc = xyzz_dbl(acc); break;
        case 3: acc = xyzz_mul(acc, k); break;
        case 4: acc = XYZZ<F>::from_affine(ld<Affine<F>>(q)); break;
    }
    st(out_xyzz, acc);
    st(out_aff, xyzz_to_affine(acc));
}
extern "C" {
void ht_fr_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *o) { field_op<Fr>(op, a, b, o); }
void ht_fq_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *o) { field_op<Fq>(op, a, b, o); }
void ht_fq2_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *o) { field_op<Fq2>(op, a, b, o); }
void ht_fr_to_mont(const uint32_t *a, uint32_t *o) { st(o, fp_to_mont(ld<Fr>(a))); }
void ht_fr_from_mont(const uint32_t *a, uint32_t *o) { st(o, fp_from_mont(ld<Fr>(a))); }
void ht_g1_op(int op, const uint32_t *acc, const uint32_t *q, const uint32_t *k, int neg, uint32_t *ox, uint32_t *oa) { point_op<Fq>(op, acc, q, k, neg, ox, oa); }
void ht_g2_op(int op, const uint32_t *acc, const uint32_t *q, const uint32_t *k, int neg, uint32_t *ox, uint32_t *oa) { point_op<Fq2>(op, acc, q, k, neg, ox, oa); }
}

// ---------------------------------------------------------------------------------------------- unsaturated (U-form) types
template <class FS, class FU> static void fieldu_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *o) {
    FU x = to_u(ld<FS>(a)), y = to_u(ld<FS>(b)), r;
    switch (op) {
        case 0: r = f_add(x, y); break;
        case 1: r = f_sub(x, y); break;
        case 2: r = f_mul(x, y); break;
        case 3: r = f_sqr(x); break;
        case 4: r = f_inv(x); break;
        case 5: r = f_neg(x); break;
        case 6: r = f_sub2(x, y); break;
        case 8: r = f_tidy(f_mul(f_inv(f_mul(x, y)), y)); break;      // inverse of an un-reduced product, tidied: == 1/x
        case 9: {   // the fused Y3 of the mixed addition: a*b - c*d with c a stored coordinate at its documented bound (< 42q)
            FU cbig = f_sub(f_mul(x, x), f_add(f_mul(x, y), f_dbl(f_mul(y, y))));    // == x^2 - xy - 2y^2, value < 2q + 32q + ...
            r = f_mul_sub(f_sub2(x, y), f_add(x, y), cbig, f_sub2(y, x));
            break;
        }
        case 7: {   // bound stress: a long un-reduced expression inside the documented limits
            FU t = f_sub2(f_mul(x, y), f_sub(f_add(x, y), f_dbl(y)));      // < 2q + 64q
            FU u = f_sub(f_sqr(t), f_add(f_mul(t, x), f_dbl(f_mul(t, y)))); // < 42q
            r = f_sub(f_mul(t, f_sub2(x, u)), f_mul(u, y));
            break;
        }
        default: r = x;
    }
    st(o, to_sat(r));
}
template <class FS, class FU> static XYZZ<FU> xyzz_to_u(const XYZZ<FS> &p) { return XYZZ<FU>{to_u(p.x), to_u(p.y), to_u(p.zz), to_u(p.zzz)}; }
template <class FS, class FU> static XYZZ<FS> xyzz_to_sat(const XYZZ<FU> &p) { return XYZZ<FS>{to_sat(p.x), to_sat(p.y), to_sat(p.zz), to_sat(p.zzz)}; }
template <class FS, class FU> static void pointu_op(int op, const uint32_t *acc_in, const uint32_t *q, int neg, int reps, uint32_t *out_aff) {
    XYZZ<FU> acc = xyzz_to_u<FS, FU>(ld<XYZZ<FS>>(acc_in));
    for (int i = 0; i < reps; i++) {
        switch (op) {
            case 0: { Affine<FS> s = ld<Affine<FS>>(q); Affine<FU> u{to_u(s.x), to_u(s.y)}; xyzz_madd(acc, u, neg != 0); break; }
            case 1: { XYZZ<FU> u = xyzz_to_u<FS, FU>(ld<XYZZ<FS>>(q)); xyzz_add(acc, u); break; }
            case 2: acc = xyzz_dbl(acc); break;
        }
    }
    st(out_aff, xyzz_to_affine(xyzz_to_sat<FS, FU>(acc)));
}
extern "C" {
void ht_fqu_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *o) { fieldu_op<Fq, FqU>(op, a, b, o); }
void ht_fq2u_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *o) { fieldu_op<Fq2, Fq2U>(op, a, b, o); }
int ht_fqu_is_zero_mod_k(int k, int delta) {   // is_zero_mod(k*q + delta) for the test
    FqU x = FqU::zero();
    unsigned long long carry = 0;
    for (int i = 0; i < 14; i++) { unsigned long long v = (unsigned long long)k * FqUP::mod(i) + carry + (i == 0 ? (unsigned)delta : 0u); x.l[i] = (uint32_t)(v & FqU::MASK); carry = v >> 29; }
    return fqu_is_zero_mod(x) ? 1 : 0;
}
void ht_g1u_op(int op, const uint32_t *acc, const uint32_t *q, int neg, int reps, uint32_t *oa) { pointu_op<Fq, FqU>(op, acc, q, neg, reps, oa); }
void ht_g2u_op(int op, const uint32_t *acc, const uint32_t *q, int neg, int reps, uint32_t *oa) { pointu_op<Fq2, Fq2U>(op, acc, q, neg, reps, oa); }
}

// ---------------------------------------------------------------------------------------------- unsaturated Fr (csrc/fru.cuh)
extern "C" void ht_fru_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *o) {
    const Fr sa = ld<Fr>(a), sb = ld<Fr>(b);
    const FrU x = fru_from_sat(sa), y = fru_from_sat(sb);
    FrU r;
    switch (op) {
        case 0: r = fru_cond_sub<true>(fru_add(x, y)); break;
        case 1: r = fru_cond_sub<true>(fru_sub_2r(x, y)); break;
        case 2: r = fru_mul(x, y); break;
        case 3: r = fru_mul(fru_sub_2r(x, y), y); break;                    // (a - b) b with the un-reduced difference (< 4r)
        case 4: st(o, fru_mul_to_sat(x, fru_repack(sb))); return;           // store path: U-form times a saturated-form factor
        case 5: {   // coset load path: x times a table entry scaled by 2^10 (one product = full conversion)
            const Fr two10 = fp_to_mont([] { Fr c = Fr::zero(); c.l[0] = 1u << 10; return c; }());
            r = fru_mul(fru_repack(sa), fru_repack(fp_mul(sb, two10)));
            break;
        }
        case 7: r = fru_repack(sa); break;       // plain load: the stored limbs taken as the unsaturated form of x 2^-5
        case 6: {   // fused point-wise load path of the witness map's seventh transform: (a*b - b) * zc with zc = a * 2^266 (zinv := a):
                    // the result is the value times 2^-5 in the unsaturated form, exactly what a plain (conversion-free) load leaves
            const Fr two10 = fp_to_mont([] { Fr c = Fr::zero(); c.l[0] = 1u << 10; return c; }());
            const FrU zc = fru_repack(fp_mul(sa, two10));
            const FrU xb = fru_mul(fru_repack(sa), fru_repack(sb));
            const FrU c = fru_mul(fru_repack(sb), fru_one_sat());
            r = fru_mul(fru_sub_2r(xb, c), zc);
            break;
        }
        case 8: {   // the butterflies' lazily reduced sum, twelve stages deep on the path that is never multiplied: x + 12 y
            r = x;
            for (int i = 0; i < 12; i++) r = fru_add_lazy(r, y);
            break;
        }
        case 9: r = fru_mul(fru_sub_4r_raw(x, fru_add_lazy(fru_add_lazy(x, y), y)), y); break;      // (x - (x + 2y)) y: raw difference into the product
        case 10: {  // last stage: the raw (un-normalised, < 6r) difference goes straight into the store's product
            FrU s2 = x;
            for (int i = 0; i < 11; i++) s2 = fru_add_lazy(s2, x);          // 12 x, as large as a lazily reduced value gets
            st(o, fru_mul_to_sat(fru_sub_4r_raw(s2, y), fru_one_sat()));
            return;
        }
        default: r = x;
    }
    st(o, fru_mul_to_sat(r, fru_one_sat()));
}

// ---------------------------------------------------------------------------------------------- host-only 64-bit-limb fields (hostff.hpp)
// and the verifier's tower arithmetic (pairing_fast.inc), as verify.hip includes it
#include "hostff.hpp"
#include <vector>
namespace {
#include "final_exp.inc"
const uint64_t Z_ABS = 0xd201000000010000ULL;
struct Fq12 { Fq2 c[6]; };
#include "pairing_fast.inc"
template <class P> void h64_op(int op, const uint64_t *a, const uint64_t *b, uint64_t *o) {
    using F = zk::h64::F<P>;
    F x, y, r;
    memcpy(x.l, a, sizeof x.l);
    memcpy(y.l, b, sizeof y.l);
    switch (op) {
        case 0: r = zk::h64::add(x, y); break;
        case 1: r = zk::h64::sub(x, y); break;
        case 2: r = zk::h64::mul(x, y); break;
        case 3: r = zk::h64::sqr(x); break;
        case 4: r = zk::h64::inv(x); break;
        case 5: r = zk::h64::neg(x); break;
        case 6: r = zk::h64::to_mont(x); break;
        case 7: r = zk::h64::from_mont(x); break;
        default: r = x;
    }
    memcpy(o, r.l, sizeof r.l);
}
// ABI order of an Fq12 (ark's tower order) = the field order of pf::F12
pf::F12 f12_ld(const uint64_t *p) { pf::F12 a; memcpy(&a, p, sizeof a); return a; }
void f12_st(uint64_t *p, const pf::F12 &a) { memcpy(p, &a, sizeof a); }
}  // namespace
extern "C" {
void ht_h64_fr_op(int op, const uint64_t *a, const uint64_t *b, uint64_t *o) { h64_op<FrP>(op, a, b, o); }
void ht_h64_fq_op(int op, const uint64_t *a, const uint64_t *b, uint64_t *o) { h64_op<FqP>(op, a, b, o); }
// op 0 mul, 1 sqr, 2 inv, 3 mul_by_014 (b: l0 = b[0..12), l1 = b[12..24), l4 = b[24..36)), 4 frobenius (times = b[0]), 5 conj
void ht_f12_op(int op, const uint64_t *a, const uint64_t *b, uint64_t *o) {
    static_assert(sizeof(pf::F12) == 72 * 8, "layout");
    const pf::F12 x = f12_ld(a);
    pf::F12 r;
    switch (op) {
        case 0: r = pf::mul(x, f12_ld(b)); break;
        case 1: r = pf::sqr(x); break;
        case 2: r = pf::inv(x); break;
        case 3: { pf::F2 l[3]; memcpy(l, b, sizeof l); r = pf::mul_by_014(x, l[0], l[1], l[2]); break; }
        case 4: r = pf::frob(x, (int)b[0]); break;
        default: r = pf::conj(x);
    }
    f12_st(o, r);
}
// the shortcuts that only hold in the cyclotomic subgroup, against the generic arithmetic, on g = f^((q^6 - 1)(q^2 + 1)):
// bit 0: cyclotomic_sqr(g) == sqr(g), bit 1: conj(g) == inv(g), bit 2: pow_z(g) == generic g^|z| conjugated, bit 3: final_exp(f) is in
// the subgroup of order r (its r-th power is one is not checked here: its cube-free equality with the plain exponent is, in
// tests/test_verify_pairing.py); bit 4: the endomorphism constants were calibrated (fast subgroup tests in use)
int ht_f12_cyclotomic_checks(const uint64_t *a) {
    const pf::F12 f = f12_ld(a);
    const pf::F12 f1 = pf::mul(pf::conj(f), pf::inv(f));
    const pf::F12 g = pf::mul(pf::frob(f1, 2), f1);
    int ok = 0;
    if (pf::eq(pf::cyclotomic_sqr(g), pf::sqr(g))) ok |= 1;
    if (pf::eq(pf::mul(pf::conj(g), g), pf::f12_one())) ok |= 2;
    pf::F12 acc = g;
    for (int i = 62; i >= 0; i--) {
        acc = pf::sqr(acc);
        if ((Z_ABS >> i) & 1) acc = pf::mul(acc, g);
    }
    if (pf::eq(pf::pow_z(g), pf::conj(acc))) ok |= 4;
    const pf::F12 e = pf::final_exp(f);
    if (pf::eq(pf::mul(pf::conj(e), e), pf::f12_one())) ok |= 8;
    const pf::Endo &en = pf::endo();
    if (en.fast_g1 && en.fast_g2) ok |= 16;
    return ok;
}
}
