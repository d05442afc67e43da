"""Pure-Python BLS12-381 pairing + Groth16 verifier (test infrastructure only).

The reference's own tests pin its prover through exactly one property: `Groth16::verify(...) == true`
(src/arkworks/matrix_proof_of_work/constraints.rs:231-272, constraints/fibbonaci.rs:192-232,
prime_snark/prime_circut.rs:273-357).  This module restates that check from the textbook definitions so the same
test strategy can run here: optimal-ate Miller loop over the M-type sextic twist (G2 kept on the twist in Fq2, line
values mapped into Fq12 = Fq2[w]/(w^6 - (1+u))), final exponentiation by plain exponentiation with (q^12 - 1)/r.
It uses the loop count |z| without the sign correction, i.e. it computes e(P, Q)^-1 consistently — pairing *equations*
are unaffected.  Deliberately simple and slow (~1 s per product of pairings)."""
import pyref as P

Q = P.Q_MOD
R = P.R_MOD
Z_ABS = 0xd201000000010000
Fq2 = P.Fq2
XI = Fq2(1, 1)


class Fq12:
    """polynomials of degree < 6 in w over Fq2, w^6 = xi = 1 + u"""
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = c

    @staticmethod
    def one():
        return Fq12([Fq2(1, 0)] + [Fq2(0, 0)] * 5)

    def __mul__(self, o):
        t = [Fq2(0, 0)] * 11
        for i, a in enumerate(self.c):
            if a.is_zero():
                continue
            for j, b in enumerate(o.c):
                if b.is_zero():
                    continue
                t[i + j] = t[i + j] + a * b
        for k in range(10, 5, -1):
            t[k - 6] = t[k - 6] + t[k] * XI
        return Fq12(t[:6])

    def __eq__(self, o):
        return all(a == b for a, b in zip(self.c, o.c))

    def pow(self, e):
        acc, base = Fq12.one(), self
        while e:
            if e & 1:
                acc = acc * base
            base = base * base
            e >>= 1
        return acc


def _line(T, Rp, Pp):
    """line through twist points T, Rp (T == Rp: tangent) evaluated at Pp in G1, scaled by w^3:
       y_P w^3 - lambda' x_P w^2 + (lambda' x_T - y_T);  returns (Fq12 value, T + Rp)"""
    xt, yt = T
    xr, yr = Rp
    if xt == xr and yt == yr:
        lam = (xt * xt * 3) * (yt * 2).inv()
    else:
        lam = (yr - yt) * (xr - xt).inv()
    x3 = lam * lam - xt - xr
    y3 = lam * (xt - x3) - yt
    xp, yp = Pp[0].v, Pp[1].v
    c = [Fq2(0, 0)] * 6
    c[0] = lam * xt - yt
    c[2] = -(lam * xp)
    c[3] = Fq2(yp, 0)
    return Fq12(c), (x3, y3)


def miller_loop(Pp, Qp):
    """Pp: G1 affine (pyref tuple of Fq1) or None; Qp: G2 affine on the twist (tuple of Fq2) or None"""
    if Pp is None or Qp is None:
        return Fq12.one()
    f = Fq12.one()
    T = Qp
    for bit in bin(Z_ABS)[3:]:
        l, T = _line(T, T, Pp)
        f = f * f * l
        if bit == "1":
            l, T = _line(T, Qp, Pp)
            f = f * l
    return f


FINAL_EXP = (Q ** 12 - 1) // R


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 ?"""
    f = Fq12.one()
    for Pp, Qp in pairs:
        f = f * miller_loop(Pp, Qp)
    return f.pow(FINAL_EXP) == Fq12.one()


def groth16_verify(vk, public_inputs, proof):
    """vk: dict alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1 (list, first entry pairs with the constant 1);
       public_inputs: list of ints (without the leading 1); proof: (A, B, C) pyref affine points.
       e(A, B) = e(alpha, beta) e(sum z_i gamma_abc_i, gamma) e(C, delta)   (ark-groth16 verifier.rs)"""
    A, B, C = proof
    acc = vk["gamma_abc_g1"][0]
    for z, g in zip(public_inputs, vk["gamma_abc_g1"][1:]):
        acc = P.ec_add(acc, P.ec_mul(g, z))
    return pairing_product_is_one([(A, B), (P.ec_neg(vk["alpha_g1"]), vk["beta_g2"]),
                                   (P.ec_neg(acc), vk["gamma_g2"]), (P.ec_neg(C), vk["delta_g2"])])
