// Groth16 verification on the host (scope row f-3; the other half of the reference's handlers:
// `Groth16::<Bls12_381>::verify_with_processed_vk` at /root/reference/src/arkworks/backend/matrix_proof.rs:200-205,
// fibbonaci_handler.rs:129-140, prime_snark.rs:191-200; upstream ark-groth16 verifier.rs + ark-ec bls12 pairing).
// Pure host code (no HIP call): verification is a few milliseconds in the reference and not on the hot path.
//
//   * Fq12 = Fq2[w]/(w^6 - xi), xi = 1 + u, kept flat as six Fq2 coefficients (the even ones form Fq6 = Fq2[v]/(v^3 - xi)
//     with v = w^2, which is what the inversion uses).
//   * One Miller loop for all pairs (the squaring of f is shared): optimal ate over |z| = 0xd201000000010000, G2 on the
//     M-type twist in homogeneous projective coordinates (no inversions), line values as sparse elements
//     c0 + c2 w^2 + c3 w^3 scaled by Fq2 factors the final exponentiation removes.
//   * Final exponentiation: easy part f^((q^6-1)(q^2+1)) with one inversion and a Frobenius; hard part via
//     3 (q^4 - q^2 + 1)/r = (z-1)^2 (z+q) (z^2+q^2-1) + 3 (Hayashida-Hayasaka-Teruya), i.e. five exponentiations by |z|.
//     The result is the cube of the reduced pairing; gcd(3, r) = 1, so "== 1" is unchanged.  The plain exponentiation by
//     (q^12-1)/r is kept as a cross-check (ZKG16_PAIRING_PLAIN_FINAL_EXP; tests compare the two verdicts).
//   * The loop count is used without the sign correction for z < 0 (every pairing comes out inverted, consistently),
//     which leaves pairing-product equations unchanged.
#include <string.h>

#include <algorithm>
#include <atomic>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/zkg16.h"
#include "ec.cuh"
#include "hostff.hpp"

using namespace zk;

namespace {

#include "final_exp.inc"

const uint64_t Z_ABS = 0xd201000000010000ULL;

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
Fq2 fq2_conj(const Fq2 &a) { return Fq2{a.c0, fp_neg(a.c1)}; }
Fq2 fq2_scale(const Fq2 &a, const Fq &k) { return Fq2{fp_mul(a.c0, k), fp_mul(a.c1, k)}; }
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
// x -> x^(q^6): w -> -w, Fq2 fixed.  On the cyclotomic subgroup (after the easy part) this is the inverse.
Fq12 fq12_conj(const Fq12 &a) {
    Fq12 r = a;
    for (int k = 1; k < 6; k += 2) r.c[k] = f_neg(a.c[k]);
    return r;
}

// ---- Frobenius: (sum c_i w^i)^q = sum conj(c_i) gamma_i w^i, gamma_i = xi^(i (q-1)/6)
struct FrobCoeffs {
    Fq2 g[6];
    FrobCoeffs() {
        const Fq2 xi{Fq::one(), Fq::one()};
        Fq2 base = Fq2::one();            // xi^((q-1)/6)
        bool started = false;
        for (int i = 6 * 64 - 1; i >= 0; i--) {
            if (started) base = f_sqr(base);
            if ((FROB_EXP[i / 64] >> (i % 64)) & 1) {
                base = started ? f_mul(base, xi) : xi;
                started = true;
            }
        }
        g[0] = Fq2::one();
        for (int i = 1; i < 6; i++) g[i] = f_mul(g[i - 1], base);
    }
};
const FrobCoeffs &frob_coeffs() {
    static FrobCoeffs f;
    return f;
}
Fq12 fq12_frob(const Fq12 &a, int times) {
    const FrobCoeffs &f = frob_coeffs();
    Fq12 r = a;
    for (int t = 0; t < times; t++)
        for (int i = 0; i < 6; i++) r.c[i] = f_mul(fq2_conj(r.c[i]), f.g[i]);
    return r;
}

// ---- inversion through Fq6 = Fq2[v]/(v^3 - xi): a = g + h w, a^-1 = (g - h w) / (g^2 - v h^2)
struct Fq6 { Fq2 a0, a1, a2; };
Fq6 fq6_mul(const Fq6 &x, const Fq6 &y) {
    const Fq2 t0 = f_mul(x.a0, y.a0), t1 = f_mul(x.a1, y.a1), t2 = f_mul(x.a2, y.a2);
    Fq6 r;
    r.a0 = f_add(t0, mul_xi(f_add(f_mul(x.a1, y.a2), f_mul(x.a2, y.a1))));
    r.a1 = f_add(f_add(f_mul(x.a0, y.a1), f_mul(x.a1, y.a0)), mul_xi(t2));
    r.a2 = f_add(f_add(f_mul(x.a0, y.a2), f_mul(x.a2, y.a0)), t1);
    return r;
}
Fq6 fq6_sub(const Fq6 &x, const Fq6 &y) { return Fq6{f_sub(x.a0, y.a0), f_sub(x.a1, y.a1), f_sub(x.a2, y.a2)}; }
Fq6 fq6_mul_v(const Fq6 &x) { return Fq6{mul_xi(x.a2), x.a0, x.a1}; }
Fq6 fq6_inv(const Fq6 &x) {
    const Fq2 t0 = f_sub(f_sqr(x.a0), mul_xi(f_mul(x.a1, x.a2)));
    const Fq2 t1 = f_sub(mul_xi(f_sqr(x.a2)), f_mul(x.a0, x.a1));
    const Fq2 t2 = f_sub(f_sqr(x.a1), f_mul(x.a0, x.a2));
    const Fq2 d = f_add(f_mul(x.a0, t0), mul_xi(f_add(f_mul(x.a2, t1), f_mul(x.a1, t2))));
    const Fq2 di = f_inv(d);
    return Fq6{f_mul(t0, di), f_mul(t1, di), f_mul(t2, di)};
}
Fq12 fq12_inv(const Fq12 &a) {
    const Fq6 g{a.c[0], a.c[2], a.c[4]}, h{a.c[1], a.c[3], a.c[5]};
    const Fq6 d = fq6_inv(fq6_sub(fq6_mul(g, g), fq6_mul_v(fq6_mul(h, h))));
    const Fq6 rg = fq6_mul(g, d), rh = fq6_mul(h, d);
    Fq12 r;
    r.c[0] = rg.a0; r.c[2] = rg.a1; r.c[4] = rg.a2;
    r.c[1] = f_neg(rh.a0); r.c[3] = f_neg(rh.a1); r.c[5] = f_neg(rh.a2);
    return r;
}

// a^z for a in the cyclotomic subgroup, z = -|z|
Fq12 pow_z(const Fq12 &a) {
    Fq12 acc = a;
    for (int i = 62; i >= 0; i--) {
        acc = fq12_mul(acc, acc);
        if ((Z_ABS >> i) & 1) acc = fq12_mul(acc, a);
    }
    return fq12_conj(acc);
}
// f^(3 (q^12 - 1)/r)
Fq12 final_exp_fast(const Fq12 &f) {
    const Fq12 f1 = fq12_mul(fq12_conj(f), fq12_inv(f));       // f^(q^6 - 1)
    const Fq12 g = fq12_mul(fq12_frob(f1, 2), f1);              // ^(q^2 + 1): now unitary, inverse = conjugate
    const Fq12 t0 = fq12_mul(pow_z(g), fq12_conj(g));           // g^(z - 1)
    const Fq12 t1 = fq12_mul(pow_z(t0), fq12_conj(t0));         // g^((z - 1)^2)
    const Fq12 t2 = fq12_mul(pow_z(t1), fq12_frob(t1, 1));      // ^(z + q)
    const Fq12 t3 = fq12_mul(fq12_mul(pow_z(pow_z(t2)), fq12_frob(t2, 2)), fq12_conj(t2));      // ^(z^2 + q^2 - 1)
    return fq12_mul(t3, fq12_mul(fq12_mul(g, g), g));           // * g^3
}

// ---- Miller loop, all pairs together
struct G2Proj { Fq2 x, y, z; };
struct MillerPair {
    G1Affine p;
    G2Affine q;
    G2Proj t;
};
const Fq2 &twist_b3() {      // 3 b' = 3 * 4 (1 + u)
    static const Fq2 v = [] {
        Fq c = Fq::zero();
        c.l[0] = 12;
        const Fq m = fp_to_mont(c);
        return Fq2{m, m};
    }();
    return v;
}
Fq12 line_value(const Fq2 &c0, const Fq2 &c2, const Fq2 &c3) {
    Fq12 l;
    for (auto &x : l.c) x = Fq2::zero();
    l.c[0] = c0; l.c[2] = c2; l.c[3] = c3;
    return l;
}
// tangent at T, evaluated at P, scaled by 2 Y Z^2 / Z: (Y^2 - 3b'Z^2) - 3X^2 x_P w^2 + 2YZ y_P w^3;  T <- 2T
Fq12 dbl_step(MillerPair &m) {
    const Fq2 &X = m.t.x, &Y = m.t.y, &Z = m.t.z;
    const Fq2 B = f_sqr(Y), C = f_sqr(Z), J = f_sqr(X);
    const Fq2 E = f_mul(twist_b3(), C);
    const Fq2 F = f_add(f_dbl(E), E);
    const Fq2 H = f_dbl(f_mul(Y, Z));
    const Fq2 XY = f_mul(X, Y);
    const Fq12 l = line_value(f_sub(B, E), f_neg(fq2_scale(f_add(f_dbl(J), J), m.p.x)), fq2_scale(H, m.p.y));
    // (X3, Y3, Z3) = (XY/2 (B - F), ((B + F)/2)^2 - 3E^2, B H); scaled by 4 to stay clear of halving:
    //   X3' = 2 XY (B - F), Y3' = (B + F)^2 - 12 E^2, Z3' = 4 B H
    const Fq2 E2 = f_sqr(E);
    G2Proj r;
    r.x = f_mul(f_dbl(XY), f_sub(B, F));
    Fq2 e12 = f_add(f_dbl(E2), E2);      // 3 E^2
    e12 = f_dbl(f_dbl(e12));             // 12 E^2
    r.y = f_sub(f_sqr(f_add(B, F)), e12);
    r.z = f_dbl(f_dbl(f_mul(B, H)));
    m.t = r;
    return l;
}
// chord through T and Q (affine), scaled by lambda: (theta x_Q - lambda y_Q) - theta x_P w^2 + lambda y_P w^3;  T <- T + Q
Fq12 add_step(MillerPair &m) {
    const Fq2 &X = m.t.x, &Y = m.t.y, &Z = m.t.z;
    const Fq2 theta = f_sub(Y, f_mul(m.q.y, Z));
    const Fq2 lambda = f_sub(X, f_mul(m.q.x, Z));
    const Fq2 C = f_sqr(theta), D = f_sqr(lambda);
    const Fq2 E = f_mul(lambda, D), F = f_mul(Z, C), G = f_mul(X, D);
    const Fq2 H = f_sub(f_add(E, F), f_dbl(G));
    const Fq12 l = line_value(f_sub(f_mul(theta, m.q.x), f_mul(lambda, m.q.y)), f_neg(fq2_scale(theta, m.p.x)), fq2_scale(lambda, m.p.y));
    G2Proj r;
    r.x = f_mul(lambda, H);
    r.y = f_sub(f_mul(theta, f_sub(G, H)), f_mul(E, Y));
    r.z = f_mul(Z, E);
    m.t = r;
    return l;
}
Fq12 multi_miller_loop(std::vector<MillerPair> &pairs) {
    Fq12 f = fq12_one();
    for (auto &m : pairs) m.t = G2Proj{m.q.x, m.q.y, Fq2::one()};
    for (int i = 62; i >= 0; i--) {          // bit 63 is the leading one
        f = fq12_mul(f, f);
        for (auto &m : pairs) f = fq12_mul(f, dbl_step(m));
        if ((Z_ABS >> i) & 1)
            for (auto &m : pairs) f = fq12_mul(f, add_step(m));
    }
    return f;
}
bool pairing_product_is_one(std::vector<MillerPair> &pairs, bool plain_final_exp) {
    std::vector<MillerPair> live;
    for (const auto &m : pairs)
        if (!m.p.is_inf() && !m.q.is_inf()) live.push_back(m);      // e(O, Q) = e(P, O) = 1
    const Fq12 f = multi_miller_loop(live);
    return fq12_is_one(plain_final_exp ? fq12_pow(f, FINAL_EXP, FINAL_EXP_LIMBS) : final_exp_fast(f));
}

// ------------------------------------------------------------------------------------------------ prepared keys
// ark-ec 0.4.2 models/bls12/g2.rs `G2Prepared::from(G2Affine)` and `Bls12::multi_miller_loop` / `ell`, restated: the
// reference ships `PreparedVerifyingKey` (io.rs:62-77: encode_pvk; matrix_proof.rs:134-136), whose serialization holds
// e(alpha, beta) in ark's Fq12 tower and, for -gamma and -delta, the 68 line-coefficient triples of the Miller loop over
// X = 0xd201000000010000 (63 doublings + 5 additions) in exactly ark's scaling (homogeneous projective, halvings by 2^-1).
struct EllCoeff { Fq2 c0, c1, c2; };
struct G2Prepared {
    std::vector<EllCoeff> ell;
    bool infinity = true;
};
const Fq &two_inv() {
    static const Fq v = [] {
        Fq c = Fq::zero();
        c.l[0] = 2;
        return fp_inv(fp_to_mont(c));
    }();
    return v;
}
const Fq2 &twist_b() {       // G2 COEFF_B = 4 (1 + u)
    static const Fq2 v = [] {
        Fq c = Fq::zero();
        c.l[0] = 4;
        const Fq m = fp_to_mont(c);
        return Fq2{m, m};
    }();
    return v;
}
// G2HomProjective::double_in_place (M twist): returns (i, 3j, -h)
EllCoeff ark_double(G2Proj &r) {
    const Fq2 a = fq2_scale(f_mul(r.x, r.y), two_inv());
    const Fq2 b = f_sqr(r.y);
    const Fq2 c = f_sqr(r.z);
    const Fq2 e = f_mul(twist_b(), f_add(f_dbl(c), c));
    const Fq2 f = f_add(f_dbl(e), e);
    const Fq2 g = fq2_scale(f_add(b, f), two_inv());
    const Fq2 h = f_sub(f_sqr(f_add(r.y, r.z)), f_add(b, c));
    const Fq2 i = f_sub(e, b);
    const Fq2 j = f_sqr(r.x);
    const Fq2 e2 = f_sqr(e);
    r.x = f_mul(a, f_sub(b, f));
    r.y = f_sub(f_sqr(g), f_add(f_dbl(e2), e2));
    r.z = f_mul(b, h);
    return EllCoeff{i, f_add(f_dbl(j), j), f_neg(h)};
}
// G2HomProjective::add_in_place (M twist): returns (j, -theta, lambda)
EllCoeff ark_add(G2Proj &r, const G2Affine &q) {
    const Fq2 theta = f_sub(r.y, f_mul(q.y, r.z));
    const Fq2 lambda = f_sub(r.x, f_mul(q.x, r.z));
    const Fq2 c = f_sqr(theta), d = f_sqr(lambda);
    const Fq2 e = f_mul(lambda, d), f = f_mul(r.z, c), g = f_mul(r.x, d);
    const Fq2 h = f_sub(f_add(e, f), f_dbl(g));
    const Fq2 ry = r.y;
    r.x = f_mul(lambda, h);
    r.y = f_sub(f_mul(theta, f_sub(g, h)), f_mul(e, ry));
    r.z = f_mul(r.z, e);
    const Fq2 j = f_sub(f_mul(theta, q.x), f_mul(lambda, q.y));
    return EllCoeff{j, f_neg(theta), lambda};
}
G2Prepared g2_prepare(const G2Affine &q) {
    G2Prepared p;
    if (q.is_inf()) return p;
    p.infinity = false;
    G2Proj r{q.x, q.y, Fq2::one()};
    for (int i = 62; i >= 0; i--) {          // BitIteratorBE::new(X).skip(1)
        p.ell.push_back(ark_double(r));
        if ((Z_ABS >> i) & 1) p.ell.push_back(ark_add(r, q));
    }
    return p;
}
// Bls12::ell (M twist): c2 *= p.y, c1 *= p.x, f.mul_by_014(c0, c1, c2) — in the flat basis: c0 + c1 w^2 + c2 w^3
void ark_ell(Fq12 &f, const EllCoeff &c, const G1Affine &p) { f = fq12_mul(f, line_value(c.c0, fq2_scale(c.c1, p.x), fq2_scale(c.c2, p.y))); }
// Bls12::multi_miller_loop over (G1 point, prepared G2) pairs, with the conjugation for the negative loop count
Fq12 ark_multi_miller_loop(const std::vector<std::pair<G1Affine, const G2Prepared *>> &all) {
    std::vector<std::pair<G1Affine, const G2Prepared *>> pairs;
    for (const auto &pr : all)
        if (!pr.first.is_inf() && !pr.second->infinity) pairs.push_back(pr);
    Fq12 f = fq12_one();
    size_t k = 0;
    for (int i = 62; i >= 0; i--) {
        f = fq12_mul(f, f);
        for (const auto &pr : pairs) ark_ell(f, pr.second->ell[k], pr.first);
        k++;
        if ((Z_ABS >> i) & 1) {
            for (const auto &pr : pairs) ark_ell(f, pr.second->ell[k], pr.first);
            k++;
        }
    }
    return fq12_conj(f);                     // X_IS_NEGATIVE: cyclotomic_inverse_in_place
}
bool fq12_eq(const Fq12 &a, const Fq12 &b) {
    for (int k = 0; k < 6; k++)
        if (!(a.c[k] == b.c[k])) return false;
    return true;
}
// ark's tower order: Fq12 = c0 + c1 w over Fq6 = Fq2[v]/(v^3 - xi) with v = w^2:  (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2)
// = flat coefficients of (1, w^2, w^4, w, w^3, w^5)
const int TOWER_ORDER[6] = {0, 2, 4, 1, 3, 5};
void fq12_to_abi(const Fq12 &a, uint64_t out[72]) {
    for (int k = 0; k < 6; k++) memcpy(out + 12 * k, &a.c[TOWER_ORDER[k]], sizeof(Fq2));
}
Fq12 fq12_from_abi(const uint64_t in[72]) {
    Fq12 a;
    for (int k = 0; k < 6; k++) memcpy(&a.c[TOWER_ORDER[k]], in + 12 * k, sizeof(Fq2));
    return a;
}
const size_t ELL_COUNT = 68;
void prepared_to_abi(const G2Prepared &p, uint64_t *out /* 68 x 36 */) {
    for (size_t i = 0; i < p.ell.size(); i++) memcpy(out + 36 * i, &p.ell[i], sizeof(EllCoeff));
}
G2Prepared prepared_from_abi(const uint64_t *in, size_t n) {
    G2Prepared p;
    p.infinity = n == 0;
    p.ell.resize(n);
    for (size_t i = 0; i < n; i++) memcpy(&p.ell[i], in + 36 * i, sizeof(EllCoeff));
    return p;
}
G2Affine g2_neg(const G2Affine &p) { return p.is_inf() ? p : G2Affine{p.x, f_neg(p.y)}; }

#include "pairing_fast.inc"

// ---- membership: on the curve and in the prime-order subgroup — what ark's deserialize_compressed validates.  The subgroup part
// is the endomorphism test of pairing_fast.inc (two / one scalar multiplications by the 64-bit |z| instead of one by the 255-bit
// r); the plain [r] P = O test remains behind it should the start-up calibration of the endomorphism constants ever fail.
const uint32_t *r_limbs() {
    static uint32_t l[8];
    static bool init = [] { for (int i = 0; i < 8; i++) l[i] = FrP::mod(i); return true; }();
    (void)init;
    return l;
}
bool g1_valid(const G1Affine &p) {
    if (p.is_inf()) return true;
    Fq c = Fq::zero();
    c.l[0] = 4;
    if (!(fp_sqr(p.y) == fp_add(fp_mul(fp_sqr(p.x), p.x), fp_to_mont(c)))) return false;
    const pf::Endo &en = pf::endo();
    if (en.fast_g1) return pf::g1_endo_test(pf::g1_pt(p), en.beta);
    return xyzz_mul(G1XYZZ::from_affine(p), r_limbs()).is_inf();
}
bool g2_valid(const G2Affine &p) {
    if (p.is_inf()) return true;
    if (!(f_sqr(p.y) == f_add(f_mul(f_sqr(p.x), p.x), twist_b()))) return false;
    const pf::Endo &en = pf::endo();
    if (en.fast_g2) return pf::g2_endo_test(pf::g2_pt(p), en.cx, en.cy);
    return xyzz_mul(G2XYZZ::from_affine(p), r_limbs()).is_inf();
}

G1Affine g1_neg(const G1Affine &p) { return p.is_inf() ? p : G1Affine{p.x, fp_neg(p.y)}; }

template <class A>
A load_pt(const uint64_t *l, int inf) {
    if (inf) return A::inf();
    A p;
    memcpy(&p, l, sizeof p);
    return p;
}

// ---- the fast path (pairing_fast.inc) behind the entry points
// gamma_abc[0] + sum_i z_i gamma_abc[i] -> affine (x, y) over the 64-bit-limb field; false = the point at infinity
bool prepared_inputs(const uint64_t *gamma_abc_g1, size_t num_instance, const uint64_t *public_inputs, pf::Fq64 &x, pf::Fq64 &y) {
    using pf::Fq64;
    auto pt = [&](size_t i) {
        const G1Affine p = load_pt<G1Affine>(gamma_abc_g1 + 12 * i, 0);
        return p.is_inf() ? pf::pt_inf<Fq64>() : pf::g1_pt(p);
    };
    pf::Pt<Fq64> acc = pt(0);
    for (size_t i = 1; i < num_instance; i++) {
        Fr z;
        memcpy(&z, public_inputs + 4 * (i - 1), sizeof z);
        const Fr zc = fp_from_mont(z);
        uint64_t k[4];
        memcpy(k, zc.l, sizeof k);
        if (!(k[0] | k[1] | k[2] | k[3])) continue;
        acc = pf::pt_add(acc, pf::pt_mul(pt(i), k, 4));
    }
    if (acc.inf) return false;
    const Fq64 inv = zk::h64::inv(zk::h64::mul(acc.zz, acc.zzz));
    x = zk::h64::mul(acc.x, zk::h64::mul(inv, acc.zzz));
    y = zk::h64::mul(acc.y, zk::h64::mul(inv, acc.zz));
    return true;
}
// prod e(P_i, Q_i) == 1 with the tower arithmetic: Q_i prepared on the fly
bool pairing_product_is_one_fast(const std::vector<MillerPair> &pairs) {
    std::vector<pf::Prepared> prep;
    prep.reserve(pairs.size());
    std::vector<pf::PairIn> in;
    for (const auto &m : pairs)
        if (!m.p.is_inf() && !m.q.is_inf()) prep.push_back(pf::prepare(m.q));
    size_t k = 0;
    for (const auto &m : pairs)
        if (!m.p.is_inf() && !m.q.is_inf()) in.push_back(pf::PairIn{pf::Fq64::from(m.p.x), pf::Fq64::from(m.p.y), &prep[k++]});
    return pf::eq(pf::final_exp(pf::miller_loop(in)), pf::f12_one());
}
pf::Prepared prepared_from_abi_fast(const uint64_t *in, size_t n) {
    pf::Prepared p;
    p.infinity = n == 0;
    p.ell.resize(n);
    for (size_t i = 0; i < n; i++) {
        EllCoeff e;
        memcpy(&e, in + 36 * i, sizeof e);
        p.ell[i] = pf::Ell{pf::from_sat(e.c0), pf::from_sat(e.c1), pf::from_sat(e.c2)};
    }
    return p;
}
void prepared_to_abi_fast(const pf::Prepared &p, uint64_t *out /* 68 x 36 */) {
    for (size_t i = 0; i < p.ell.size(); i++) {
        const EllCoeff e{pf::to_sat(p.ell[i].c0), pf::to_sat(p.ell[i].c1), pf::to_sat(p.ell[i].c2)};
        memcpy(out + 36 * i, &e, sizeof e);
    }
}

}  // namespace

extern "C" {

// prod_i e(P_i, Q_i) == 1 ?   g1: n x 12 limbs, g2: n x 24 limbs (arkworks layout), flag bytes nullable (= none at infinity)
int zkg16_pairing_check(const uint64_t *g1, const uint8_t *g1_inf, const uint64_t *g2, const uint8_t *g2_inf, size_t n, int flags, int *ok) {
    if ((n && (!g1 || !g2)) || !ok) return ZKG16_ERR_BAD_ARG;
    try {
        std::vector<MillerPair> pairs(n);
        for (size_t i = 0; i < n; i++) {
            pairs[i].p = load_pt<G1Affine>(g1 + 12 * i, g1_inf ? g1_inf[i] : 0);
            pairs[i].q = load_pt<G2Affine>(g2 + 24 * i, g2_inf ? g2_inf[i] : 0);
        }
        // default: tower arithmetic on 64-bit limbs (pairing_fast.inc); with the flag: the flat implementation and the plain
        // exponentiation by (q^12 - 1)/r — an independent second computation of the same verdict
        *ok = ((flags & ZKG16_PAIRING_PLAIN_FINAL_EXP) ? pairing_product_is_one(pairs, true) : pairing_product_is_one_fast(pairs)) ? 1 : 0;
    } catch (const std::bad_alloc &) {
        return ZKG16_ERR_OOM;
    }
    return ZKG16_OK;
}

int zkg16_scalar_mul_g1(const uint64_t base[12], const uint64_t k_canonical[4], uint64_t out_affine[12], uint8_t *out_inf) {
    if (!base || !k_canonical || !out_affine) return ZKG16_ERR_BAD_ARG;
    Fr k;
    memcpy(&k, k_canonical, sizeof k);
    const G1Affine r = xyzz_to_affine(xyzz_mul(G1XYZZ::from_affine(load_pt<G1Affine>(base, 0)), k.l));
    memcpy(out_affine, &r, sizeof r);
    if (out_inf) *out_inf = r.is_inf() ? 1 : 0;
    return ZKG16_OK;
}
int zkg16_scalar_mul_g2(const uint64_t base[24], const uint64_t k_canonical[4], uint64_t out_affine[24], uint8_t *out_inf) {
    if (!base || !k_canonical || !out_affine) return ZKG16_ERR_BAD_ARG;
    Fr k;
    memcpy(&k, k_canonical, sizeof k);
    const G2Affine r = xyzz_to_affine(xyzz_mul(G2XYZZ::from_affine(load_pt<G2Affine>(base, 0)), k.l));
    memcpy(out_affine, &r, sizeof r);
    if (out_inf) *out_inf = r.is_inf() ? 1 : 0;
    return ZKG16_OK;
}

// e(A, B) == e(alpha, beta) * e(sum_i z_i gamma_abc_i, gamma) * e(C, delta) ?
// gamma_abc_g1: num_instance points (the first pairs with the constant 1); public_inputs: (num_instance - 1) Montgomery Fr.
int zkg16_verify(const uint64_t alpha_g1[12], const uint64_t beta_g2[24], const uint64_t gamma_g2[24], const uint64_t delta_g2[24],
                 const uint64_t *gamma_abc_g1, size_t num_instance, const uint64_t *public_inputs,
                 const uint64_t proof[48], const uint8_t inf[3], int *ok) {
    if (!alpha_g1 || !beta_g2 || !gamma_g2 || !delta_g2 || !gamma_abc_g1 || num_instance == 0 || (!public_inputs && num_instance > 1) || !proof ||
        !inf || !ok)
        return ZKG16_ERR_BAD_ARG;
    const G1Affine A = load_pt<G1Affine>(proof, inf[0]), C = load_pt<G1Affine>(proof + 36, inf[2]);
    const G2Affine B = load_pt<G2Affine>(proof + 12, inf[1]);
    // ark's Proof::deserialize_compressed validates curve and subgroup membership before the handler ever verifies
    // (io.rs:53-60); a pairing on arbitrary coordinates is not a verification
    if (!g1_valid(A) || !g2_valid(B) || !g1_valid(C)) { *ok = 0; return ZKG16_OK; }
    try {
        const G1Affine alpha = load_pt<G1Affine>(alpha_g1, 0);
        const G2Affine beta = load_pt<G2Affine>(beta_g2, 0), gamma = load_pt<G2Affine>(gamma_g2, 0), delta = load_pt<G2Affine>(delta_g2, 0);
        // the key's own points: a caller that bypasses wire.py's decoder gets the guarantees of ark's deserialiser here too
        if (alpha.is_inf() || beta.is_inf() || gamma.is_inf() || delta.is_inf() || !g1_valid(alpha) || !g2_valid(beta) || !g2_valid(gamma) || !g2_valid(delta))
            return ZKG16_ERR_BAD_ARG;
        pf::Fq64 xx, xy;
        const bool have_x = prepared_inputs(gamma_abc_g1, num_instance, public_inputs, xx, xy);
        std::vector<MillerPair> pairs(4);
        pairs[0].p = A; pairs[0].q = B;
        pairs[1].p = g1_neg(alpha); pairs[1].q = beta;
        pairs[2].p = have_x ? g1_neg(G1Affine{xx.to(), xy.to()}) : G1Affine::inf(); pairs[2].q = gamma;
        pairs[3].p = g1_neg(C); pairs[3].q = delta;
        *ok = pairing_product_is_one_fast(pairs) ? 1 : 0;
    } catch (const std::bad_alloc &) {
        return ZKG16_ERR_OOM;
    }
    return ZKG16_OK;
}

// prepare_verifying_key (ark-groth16 verifier.rs): e(alpha, beta) and the prepared -gamma, -delta.  alpha_beta: 72 u64 = the six
// Fq2 of ark's Fq12 tower order (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2), Montgomery limbs; *_coeffs: 68 x 36 u64 (three Fq2
// per Miller-loop line, in loop order).  The bytes the reference puts on the wire are wire.encode_pvk's.
int zkg16_pvk_prepare(const uint64_t alpha_g1[12], const uint64_t beta_g2[24], const uint64_t gamma_g2[24], const uint64_t delta_g2[24],
                      uint64_t alpha_beta[72], uint64_t *gamma_neg_coeffs, uint64_t *delta_neg_coeffs, size_t *n_coeffs) {
    if (!alpha_g1 || !beta_g2 || !gamma_g2 || !delta_g2 || !alpha_beta || !gamma_neg_coeffs || !delta_neg_coeffs || !n_coeffs) return ZKG16_ERR_BAD_ARG;
    try {
        const G1Affine a = load_pt<G1Affine>(alpha_g1, 0);
        const G2Affine beta = load_pt<G2Affine>(beta_g2, 0), gamma = load_pt<G2Affine>(gamma_g2, 0), delta = load_pt<G2Affine>(delta_g2, 0);
        // not a key: points at infinity, off the curve or outside the prime-order subgroup (ark's deserialiser would have refused them)
        if (a.is_inf() || beta.is_inf() || gamma.is_inf() || delta.is_inf() || !g1_valid(a) || !g2_valid(beta) || !g2_valid(gamma) || !g2_valid(delta))
            return ZKG16_ERR_BAD_ARG;
        const pf::Prepared b = pf::prepare(beta), g = pf::prepare(g2_neg(gamma)), d = pf::prepare(g2_neg(delta));
        if (g.ell.size() != ELL_COUNT || d.ell.size() != ELL_COUNT) return ZKG16_ERR_BAD_ARG;
        const pf::F12 ab = pf::final_exp(pf::miller_loop({pf::PairIn{pf::Fq64::from(a.x), pf::Fq64::from(a.y), &b}}));
        fq12_to_abi(pf::to_flat(ab), alpha_beta);
        prepared_to_abi_fast(g, gamma_neg_coeffs);
        prepared_to_abi_fast(d, delta_neg_coeffs);
        *n_coeffs = ELL_COUNT;
    } catch (const std::bad_alloc &) {
        return ZKG16_ERR_OOM;
    }
    return ZKG16_OK;
}

// Groth16::verify_with_processed_vk (verifier.rs), restated: prepared_inputs = gamma_abc[0] + sum z_i gamma_abc[i];
// final_exponentiation(multi_miller_loop([(A, B), (prepared_inputs, -gamma prepared), (C, -delta prepared)])) == e(alpha, beta).
// Proof points are checked for curve and subgroup membership first (ark's deserialization does).
int zkg16_verify_prepared(const uint64_t *gamma_abc_g1, size_t num_instance, const uint64_t *public_inputs, const uint64_t alpha_beta[72],
                          const uint64_t *gamma_neg_coeffs, const uint64_t *delta_neg_coeffs, size_t n_coeffs,
                          const uint64_t proof[48], const uint8_t inf[3], int *ok) {
    if (!gamma_abc_g1 || num_instance == 0 || (!public_inputs && num_instance > 1) || !alpha_beta || !gamma_neg_coeffs || !delta_neg_coeffs ||
        n_coeffs != ELL_COUNT || !proof || !inf || !ok)
        return ZKG16_ERR_BAD_ARG;
    try {
        const G1Affine A = load_pt<G1Affine>(proof, inf[0]), C = load_pt<G1Affine>(proof + 36, inf[2]);
        const G2Affine B = load_pt<G2Affine>(proof + 12, inf[1]);
        if (!g1_valid(A) || !g2_valid(B) || !g1_valid(C)) { *ok = 0; return ZKG16_OK; }
        pf::Fq64 xx, xy;
        const bool have_x = prepared_inputs(gamma_abc_g1, num_instance, public_inputs, xx, xy);
        const pf::Prepared bp = pf::prepare(B), gp = prepared_from_abi_fast(gamma_neg_coeffs, n_coeffs), dp = prepared_from_abi_fast(delta_neg_coeffs, n_coeffs);
        std::vector<pf::PairIn> in;
        if (!A.is_inf() && !bp.infinity) in.push_back(pf::PairIn{pf::Fq64::from(A.x), pf::Fq64::from(A.y), &bp});
        if (have_x) in.push_back(pf::PairIn{xx, xy, &gp});
        if (!C.is_inf()) in.push_back(pf::PairIn{pf::Fq64::from(C.x), pf::Fq64::from(C.y), &dp});
        const pf::F12 test = pf::final_exp(pf::miller_loop(in));
        *ok = pf::eq(test, pf::from_flat(fq12_from_abi(alpha_beta))) ? 1 : 0;
    } catch (const std::bad_alloc &) {
        return ZKG16_ERR_OOM;
    }
    return ZKG16_OK;
}

// ---- the wire encodings of keys and proofs, natively (wire.py's pure-Python statement of the same rules is the reference the tests
// compare these with; a prepared verifying key holds 816 Fq values, and a key of the prime handler 258 G1 points: 55 + 18 ms of
// Python big-integer work per verify request, 2.5 ms of every prove request).
namespace {
using zk::h64::Fq64;
namespace hq = zk::h64;
struct FqConsts {
    uint64_t q[6], e[6];       // q, (q + 1) / 4
    Fq64 four;
    FqConsts() {
        for (int i = 0; i < 6; i++) q[i] = Fq64::mod(i);
        uint64_t carry = 1;
        for (int i = 0; i < 6; i++) { const uint64_t v = q[i] + carry; carry = v < carry ? 1 : 0; e[i] = v; }
        for (int i = 0; i < 6; i++) e[i] = (e[i] >> 2) | (i < 5 ? e[i + 1] << 62 : 0);
        four = Fq64::zero();
        four.l[0] = 4;
        four = hq::to_mont(four);
    }
};
const FqConsts &fq_consts() {
    static const FqConsts c;
    return c;
}
bool limbs_less(const uint64_t a[6], const uint64_t b[6]) {
    for (int i = 5; i >= 0; i--)
        if (a[i] != b[i]) return a[i] < b[i];
    return false;
}
// 48 big-endian bytes (the top three bits of byte 0 masked off when `flags`) -> canonical limbs; false if >= q
bool fq_from_be(const uint8_t *b, bool flags, Fq64 &canon) {
    for (int i = 0; i < 6; i++) {
        uint64_t v = 0;
        for (int t = 0; t < 8; t++) {
            uint8_t byte = b[47 - (8 * i + t)];
            if (flags && 8 * i + t == 47) byte &= 0x1F;
            v |= (uint64_t)byte << (8 * t);
        }
        canon.l[i] = v;
    }
    return limbs_less(canon.l, fq_consts().q);
}
void fq_to_be(const Fq64 &mont, uint8_t *b) {
    const Fq64 c = hq::from_mont(mont);
    for (int i = 0; i < 6; i++)
        for (int t = 0; t < 8; t++) b[47 - (8 * i + t)] = (uint8_t)(c.l[i] >> (8 * t));
}
// a^((q+1)/4) and whether it is a square root (q = 3 mod 4)
bool fq_sqrt(const Fq64 &a, Fq64 &r) {
    const FqConsts &k = fq_consts();
    Fq64 y = Fq64::one();
    bool started = false;
    for (int i = 383; i >= 0; i--) {
        if (started) y = hq::sqr(y);
        if ((k.e[i / 64] >> (i % 64)) & 1) { y = started ? hq::mul(y, a) : a; started = true; }
    }
    r = y;
    return hq::sqr(y) == a;
}
// y is the lexicographically larger of (y, -y): canonical y > q - y
bool fq_is_largest(const Fq64 &y) {
    const Fq64 yc = hq::from_mont(y), nyc = hq::from_mont(hq::neg(y));
    return limbs_less(nyc.l, yc.l);
}
// square root in Fq[u] / (u^2 + 1), the complex method (wire.py _sqrt_fq2)
bool fq2_sqrt(const Fq64 &a0, const Fq64 &a1, Fq64 &c0, Fq64 &c1) {
    if (a1.is_zero()) {
        Fq64 r;
        if (fq_sqrt(a0, r)) { c0 = r; c1 = Fq64::zero(); return true; }
        if (fq_sqrt(hq::neg(a0), r)) { c0 = Fq64::zero(); c1 = r; return true; }
        return false;
    }
    Fq64 n;
    if (!fq_sqrt(hq::add(hq::sqr(a0), hq::sqr(a1)), n)) return false;
    Fq64 two = Fq64::zero();
    two.l[0] = 2;
    const Fq64 inv2 = hq::inv(hq::to_mont(two));
    const Fq64 cand[2] = {hq::mul(hq::add(a0, n), inv2), hq::mul(hq::sub(a0, n), inv2)};
    for (int t = 0; t < 2; t++) {
        Fq64 r;
        if (!fq_sqrt(cand[t], r) || r.is_zero()) continue;
        const Fq64 i1 = hq::mul(a1, hq::inv(hq::dbl(r)));
        if (hq::sub(hq::sqr(r), hq::sqr(i1)) == a0) { c0 = r; c1 = i1; return true; }
    }
    return false;
}
}  // namespace

// n compressed G1 points (ark-serialize / zcash BLS12-381 encoding: 48 bytes, big-endian x, top bits = compressed, infinity, y is
// the lexicographically larger root) -> affine Montgomery limbs, strict as `G1Affine::deserialize_compressed`.  status[i]: 0 ok,
// 1 not a compressed encoding, 2 non-canonical infinity, 3 x not reduced, 4 x not on the curve, 5 not in the prime-order subgroup
// (only with validate).  Returns ZKG16_OK when every point decoded.
static int g1_decompress_range(const uint8_t *bytes, size_t lo, size_t hi, uint64_t *out, uint8_t *inf, int validate, int *status);

int zkg16_g1_decompress(const uint8_t *bytes, size_t n, uint64_t *out, uint8_t *inf, int validate, int *status) {
    if ((!bytes || !out || !inf) && n) return ZKG16_ERR_BAD_ARG;
    (void)fq_consts();
    (void)pf::endo();                // the function-local statics exist before any helper thread asks for them
    // a square root (~20 us) and a subgroup test (~50 us) per point: the 258 points of a prime-handler key on up to eight threads
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nthreads = n >= 64 && hw > 1 ? std::min<size_t>(std::min<size_t>(8, hw), n / 16) : 1;
    if (nthreads <= 1) return g1_decompress_range(bytes, 0, n, out, inf, validate, status) ? ZKG16_ERR_BAD_ARG : ZKG16_OK;
    std::atomic<int> bad{0};
    {
        struct Joiner {          // helper threads joined on every exit path; one that cannot be started runs inline
            std::vector<std::thread> th;
            ~Joiner() { for (auto &t : th) if (t.joinable()) t.join(); }
        } tg;
        const size_t per = (n + nthreads - 1) / nthreads;
        for (size_t t = 0; t < nthreads; t++) {
            const size_t lo = t * per, hi = std::min(n, lo + per);
            if (lo >= hi) break;
            auto job = [=, &bad] { bad += g1_decompress_range(bytes, lo, hi, out, inf, validate, status); };
            try {
                tg.th.emplace_back(job);
            } catch (const std::system_error &) {
                job();
            }
        }
    }
    return bad.load() ? ZKG16_ERR_BAD_ARG : ZKG16_OK;
}

static int g1_decompress_range(const uint8_t *bytes, size_t lo_k, size_t hi_k, uint64_t *out, uint8_t *inf, int validate, int *status) {
    const FqConsts &kc = fq_consts();
    int bad = 0;
    for (size_t k = lo_k; k < hi_k; k++) {
        const uint8_t *b = bytes + 48 * k;
        uint64_t *o = out + 12 * k;
        int st = 0;
        inf[k] = 0;
        memset(o, 0, 12 * sizeof(uint64_t));
        if (!(b[0] & 0x80)) st = 1;
        else if (b[0] & 0x40) {
            bool clean = b[0] == 0xC0;
            for (int i = 1; i < 48; i++) clean = clean && b[i] == 0;
            if (clean) inf[k] = 1; else st = 2;
        } else {
            Fq64 xc;
            if (!fq_from_be(b, true, xc)) st = 3;
            else {
                const Fq64 x = hq::to_mont(xc);
                Fq64 y;
                if (!fq_sqrt(hq::add(hq::mul(hq::sqr(x), x), kc.four), y)) st = 4;
                else {
                    if (fq_is_largest(y) != ((b[0] & 0x20) != 0)) y = hq::neg(y);
                    memcpy(o, x.l, 48);
                    memcpy(o + 6, y.l, 48);
                    if (validate && !g1_valid(load_pt<G1Affine>(o, 0))) st = 5;
                }
            }
        }
        if (status) status[k] = st;
        if (st) bad++;
    }
    return bad;
}

// the same for G2: 96 bytes = x.c1 || x.c0 big-endian, y compared as (c1, c0); out = n x 24 limbs (x.c0, x.c1, y.c0, y.c1)
int zkg16_g2_decompress(const uint8_t *bytes, size_t n, uint64_t *out, uint8_t *inf, int validate, int *status) {
    if ((!bytes || !out || !inf) && n) return ZKG16_ERR_BAD_ARG;
    const FqConsts &kc = fq_consts();
    int bad = 0;
    for (size_t k = 0; k < n; k++) {
        const uint8_t *b = bytes + 96 * k;
        uint64_t *o = out + 24 * k;
        int st = 0;
        inf[k] = 0;
        memset(o, 0, 24 * sizeof(uint64_t));
        if (!(b[0] & 0x80)) st = 1;
        else if (b[0] & 0x40) {
            bool clean = b[0] == 0xC0;
            for (int i = 1; i < 96; i++) clean = clean && b[i] == 0;
            if (clean) inf[k] = 1; else st = 2;
        } else {
            Fq64 x1c, x0c;
            if (!fq_from_be(b, true, x1c) || !fq_from_be(b + 48, false, x0c)) st = 3;
            else {
                const Fq64 x0 = hq::to_mont(x0c), x1 = hq::to_mont(x1c);
                // x^3 + 4 (1 + u)
                const Fq64 s0 = hq::sub(hq::sqr(x0), hq::sqr(x1)), s1 = hq::dbl(hq::mul(x0, x1));
                const Fq64 c0 = hq::add(hq::sub(hq::mul(s0, x0), hq::mul(s1, x1)), kc.four);
                const Fq64 c1 = hq::add(hq::add(hq::mul(s0, x1), hq::mul(s1, x0)), kc.four);
                Fq64 y0, y1;
                if (!fq2_sqrt(c0, c1, y0, y1)) st = 4;
                else {
                    // (y1, y0) > (-y1, -y0) lexicographically on canonical values
                    const Fq64 ny0 = hq::neg(y0), ny1 = hq::neg(y1);
                    const Fq64 y1c = hq::from_mont(y1), ny1c = hq::from_mont(ny1), y0c = hq::from_mont(y0), ny0c = hq::from_mont(ny0);
                    bool largest;
                    if (limbs_less(ny1c.l, y1c.l)) largest = true;
                    else if (limbs_less(y1c.l, ny1c.l)) largest = false;
                    else largest = limbs_less(ny0c.l, y0c.l);
                    if (largest != ((b[0] & 0x20) != 0)) { y0 = ny0; y1 = ny1; }
                    memcpy(o, x0.l, 48); memcpy(o + 6, x1.l, 48); memcpy(o + 12, y0.l, 48); memcpy(o + 18, y1.l, 48);
                    if (validate && !g2_valid(load_pt<G2Affine>(o, 0))) st = 5;
                }
            }
        }
        if (status) status[k] = st;
        if (st) bad++;
    }
    return bad ? ZKG16_ERR_BAD_ARG : ZKG16_OK;
}

// n affine points (Montgomery limbs; inf[i] != 0 = the point at infinity) -> compressed bytes (48 / 96 per point)
int zkg16_points_compress(int group, const uint64_t *points, const uint8_t *inf, size_t n, uint8_t *out) {
    if ((group != 1 && group != 2) || ((!points || !out) && n)) return ZKG16_ERR_BAD_ARG;
    const size_t w = group == 1 ? 12 : 24, nb = group == 1 ? 48 : 96;
    for (size_t k = 0; k < n; k++) {
        uint8_t *b = out + nb * k;
        if (inf && inf[k]) {
            memset(b, 0, nb);
            b[0] = 0xC0;
            continue;
        }
        const uint64_t *p = points + w * k;
        if (group == 1) {
            Fq64 x, y;
            memcpy(x.l, p, 48); memcpy(y.l, p + 6, 48);
            fq_to_be(x, b);
            b[0] |= 0x80 | (fq_is_largest(y) ? 0x20 : 0);
        } else {
            Fq64 x0, x1, y0, y1;
            memcpy(x0.l, p, 48); memcpy(x1.l, p + 6, 48); memcpy(y0.l, p + 12, 48); memcpy(y1.l, p + 18, 48);
            fq_to_be(x1, b);
            fq_to_be(x0, b + 48);
            const Fq64 y1c = hq::from_mont(y1), ny1c = hq::from_mont(hq::neg(y1));
            bool largest;
            if (limbs_less(ny1c.l, y1c.l)) largest = true;
            else if (limbs_less(y1c.l, ny1c.l)) largest = false;
            else largest = fq_is_largest(y0);
            b[0] |= 0x80 | (largest ? 0x20 : 0);
        }
    }
    return ZKG16_OK;
}

// n Fq values, Montgomery limbs <-> 48 little-endian canonical bytes each (ark's field serialization: the Fq12 and the line
// coefficients of a prepared verifying key); from-bytes refuses values >= q (status ZKG16_ERR_BAD_ARG)
int zkg16_fq_to_le_bytes(const uint64_t *limbs, size_t n, uint8_t *out) {
    if ((!limbs || !out) && n) return ZKG16_ERR_BAD_ARG;
    for (size_t k = 0; k < n; k++) {
        Fq64 v;
        memcpy(v.l, limbs + 6 * k, 48);
        const Fq64 c = hq::from_mont(v);
        memcpy(out + 48 * k, c.l, 48);          // little-endian host: canonical limbs are the bytes
    }
    return ZKG16_OK;
}
int zkg16_fq_from_le_bytes(const uint8_t *bytes, size_t n, uint64_t *out) {
    if ((!bytes || !out) && n) return ZKG16_ERR_BAD_ARG;
    for (size_t k = 0; k < n; k++) {
        Fq64 c;
        memcpy(c.l, bytes + 48 * k, 48);
        if (!limbs_less(c.l, fq_consts().q)) return ZKG16_ERR_BAD_ARG;
        const Fq64 m = hq::to_mont(c);
        memcpy(out + 6 * k, m.l, 48);
    }
    return ZKG16_OK;
}

// curve + prime-order-subgroup membership of one affine point (group: 1 = G1, 2 = G2); *ok = 1 iff both hold
int zkg16_point_check(int group, const uint64_t *point, int *ok) {
    if (!point || !ok || (group != 1 && group != 2)) return ZKG16_ERR_BAD_ARG;
    *ok = (group == 1 ? g1_valid(load_pt<G1Affine>(point, 0)) : g2_valid(load_pt<G2Affine>(point, 0))) ? 1 : 0;
    return ZKG16_OK;
}

}  // extern "C"
