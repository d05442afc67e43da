// Groth16 verification on the host (scope row f-3; the other half of the reference's handlers:
// `Groth16::<Bls12_381>::verify_with_processed_vk` at /root/reference/src/arkworks/backend/matrix_proof.rs:200-205,
// fibbonaci_handler.rs:129-140, prime_snark.rs:191-200; upstream ark-groth16 verifier.rs + ark-ec bls12 pairing).
// Pure host code — verification is ~2 ms in the reference and not on the hot path; this version favours being obviously
// right over speed: optimal-ate Miller loop with G2 kept on the M-type twist (affine, slopes in Fq2), line values as
// sparse elements of Fq12 = Fq2[w]/(w^6 - (1+u)), and the final exponentiation as a plain exponentiation by
// (q^12 - 1)/r.  The loop count |z| is used without the sign correction (e(P,Q)^-1 consistently), which leaves
// pairing-product equations unchanged.
#include <string.h>

#include <vector>

#include "../../include/zkg16.h"
#include "ec.cuh"

using namespace zk;

namespace {

#include "final_exp.inc"

struct Fq12 { Fq2 c[6]; };

Fq12 fq12_one() {
    Fq12 r;
    for (auto &x : r.c) x = Fq2::zero();
    r.c[0] = Fq2::one();
    return r;
}
Fq2 mul_xi(const Fq2 &a) {      // a * (1 + u)
    return Fq2{fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1)};
}
Fq12 fq12_mul(const Fq12 &a, const Fq12 &b) {
    Fq2 t[11];
    for (auto &x : t) x = Fq2::zero();
    for (int i = 0; i < 6; i++) {
        if (a.c[i].is_zero()) continue;
        for (int j = 0; j < 6; j++) {
            if (b.c[j].is_zero()) continue;
            t[i + j] = f_add(t[i + j], f_mul(a.c[i], b.c[j]));
        }
    }
    Fq12 r;
    for (int k = 10; k >= 6; k--) t[k - 6] = f_add(t[k - 6], mul_xi(t[k]));
    for (int k = 0; k < 6; k++) r.c[k] = t[k];
    return r;
}
bool fq12_is_one(const Fq12 &a) {
    if (!(a.c[0] == Fq2::one())) return false;
    for (int k = 1; k < 6; k++)
        if (!a.c[k].is_zero()) return false;
    return true;
}
Fq12 fq12_pow(const Fq12 &a, const uint64_t *e, int nl) {
    Fq12 acc = fq12_one();
    bool started = false;
    for (int i = nl * 64 - 1; i >= 0; i--) {
        if (started) acc = fq12_mul(acc, acc);
        if ((e[i / 64] >> (i % 64)) & 1) {
            acc = started ? fq12_mul(acc, a) : a;
            started = true;
        }
    }
    return acc;
}

// line through twist points T, R (tangent when equal) evaluated at P in G1, scaled by w^3:
//   y_P w^3 - lambda x_P w^2 + (lambda x_T - y_T);   T <- T + R
Fq12 line_and_add(G2Affine &T, const G2Affine &R, const G1Affine &P) {
    Fq2 lam;
    if (T.x == R.x && T.y == R.y) {
        const Fq2 x2 = f_sqr(T.x);
        lam = f_mul(f_add(f_dbl(x2), x2), f_inv(f_dbl(T.y)));
    } else {
        lam = f_mul(f_sub(R.y, T.y), f_inv(f_sub(R.x, T.x)));
    }
    const Fq2 x3 = f_sub(f_sub(f_sqr(lam), T.x), R.x);
    const Fq2 y3 = f_sub(f_mul(lam, f_sub(T.x, x3)), T.y);
    Fq12 l;
    for (auto &x : l.c) x = Fq2::zero();
    l.c[0] = f_sub(f_mul(lam, T.x), T.y);
    l.c[2] = f_neg(Fq2{fp_mul(lam.c0, P.x), fp_mul(lam.c1, P.x)});
    l.c[3] = Fq2{P.y, Fq::zero()};
    T = G2Affine{x3, y3};
    return l;
}

Fq12 miller_loop(const G1Affine &P, const G2Affine &Q) {
    if (P.is_inf() || Q.is_inf()) return fq12_one();
    const uint64_t z = 0xd201000000010000ULL;
    Fq12 f = fq12_one();
    G2Affine T = Q;
    for (int i = 62; i >= 0; i--) {          // bit 63 is the leading one
        const Fq12 l = line_and_add(T, G2Affine(T), P);
        f = fq12_mul(fq12_mul(f, f), l);
        if ((z >> i) & 1) {
            const Fq12 l2 = line_and_add(T, Q, P);
            f = fq12_mul(f, l2);
        }
    }
    return f;
}

G1Affine g1_neg(const G1Affine &p) { return p.is_inf() ? p : G1Affine{p.x, fp_neg(p.y)}; }

template <class A>
A load_pt(const uint64_t *l, int inf) {
    if (inf) return A::inf();
    A p;
    memcpy(&p, l, sizeof p);
    return p;
}

}  // namespace

extern "C" {

// e(A, B) == e(alpha, beta) * e(sum_i z_i gamma_abc_i, gamma) * e(C, delta) ?
// gamma_abc_g1: num_instance points (the first pairs with the constant 1); public_inputs: (num_instance - 1) Montgomery Fr.
int zkg16_verify(const uint64_t alpha_g1[12], const uint64_t beta_g2[24], const uint64_t gamma_g2[24], const uint64_t delta_g2[24],
                 const uint64_t *gamma_abc_g1, size_t num_instance, const uint64_t *public_inputs,
                 const uint64_t proof[48], const uint8_t inf[3], int *ok) {
    if (!alpha_g1 || !beta_g2 || !gamma_g2 || !delta_g2 || !gamma_abc_g1 || num_instance == 0 || (!public_inputs && num_instance > 1) || !proof ||
        !inf || !ok)
        return ZKG16_ERR_BAD_ARG;
    G1XYZZ acc = G1XYZZ::from_affine(load_pt<G1Affine>(gamma_abc_g1, 0));
    for (size_t i = 1; i < num_instance; i++) {
        Fr z;
        memcpy(&z, public_inputs + 4 * (i - 1), sizeof z);
        const Fr zc = fp_from_mont(z);
        G1XYZZ t = xyzz_mul(G1XYZZ::from_affine(load_pt<G1Affine>(gamma_abc_g1 + 12 * i, 0)), zc.l);
        xyzz_add(acc, t);
    }
    const G1Affine X = xyzz_to_affine(acc);
    const G1Affine A = load_pt<G1Affine>(proof, inf[0]), C = load_pt<G1Affine>(proof + 36, inf[2]);
    const G2Affine B = load_pt<G2Affine>(proof + 12, inf[1]);
    Fq12 f = miller_loop(A, B);
    f = fq12_mul(f, miller_loop(g1_neg(load_pt<G1Affine>(alpha_g1, 0)), load_pt<G2Affine>(beta_g2, 0)));
    f = fq12_mul(f, miller_loop(g1_neg(X), load_pt<G2Affine>(gamma_g2, 0)));
    f = fq12_mul(f, miller_loop(g1_neg(C), load_pt<G2Affine>(delta_g2, 0)));
    *ok = fq12_is_one(fq12_pow(f, FINAL_EXP, FINAL_EXP_LIMBS)) ? 1 : 0;
    return ZKG16_OK;
}

}  // extern "C"
