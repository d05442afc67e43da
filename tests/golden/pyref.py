"""Pure-Python big-integer BLS12-381 / Groth16 arithmetic used ONLY to generate and
check golden vectors (tests/golden/*.json).  Test infrastructure: nothing in the product
path imports this.

It is deliberately written from the textbook definitions (Python ``int``/``pow``) so that
it shares no code and no limb tricks with either the C oracle (oracle/) or the HIP path:
 * fields are plain residues, no Montgomery form except in the explicit converters;
 * curve points are affine with the chord/tangent formulas;
 * the DFT is the O(N^2) definition;
 * the MSM is the naive sum;
 * Groth16 setup/prove follow SURVEY.md Appendix A.4-A.7 (ark-groth16 0.4 semantics).

Constants: SURVEY.md Appendix A.1.
"""
import random

# ---------------------------------------------------------------- fields
R_MOD = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001   # Fr modulus r
Q_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
FR_GEN = 7
FR_TWO_ADICITY = 32
FR_ROOT_2_32 = pow(FR_GEN, (R_MOD - 1) >> 32, R_MOD)
FR_MONT_R = (1 << 256) % R_MOD
FQ_MONT_R = (1 << 384) % Q_MOD


def fr_to_mont(x):
    return (x * FR_MONT_R) % R_MOD


def fr_from_mont(x):
    return (x * pow(FR_MONT_R, -1, R_MOD)) % R_MOD


def fq_to_mont(x):
    return (x * FQ_MONT_R) % Q_MOD


def fq_from_mont(x):
    return (x * pow(FQ_MONT_R, -1, Q_MOD)) % Q_MOD


def limbs64(x, n):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs64(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def root_of_unity(log_n):
    assert 0 <= log_n <= 32
    return pow(FR_ROOT_2_32, 1 << (32 - log_n), R_MOD)


# ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2+1)
class Fq2:
    __slots__ = ("c0", "c1")

    def __init__(self, c0, c1=0):
        self.c0 = c0 % Q_MOD
        self.c1 = c1 % Q_MOD

    def __add__(self, o):
        return Fq2(self.c0 + o.c0, self.c1 + o.c1)

    def __sub__(self, o):
        return Fq2(self.c0 - o.c0, self.c1 - o.c1)

    def __neg__(self):
        return Fq2(-self.c0, -self.c1)

    def __mul__(self, o):
        if isinstance(o, int):
            return Fq2(self.c0 * o, self.c1 * o)
        return Fq2(self.c0 * o.c0 - self.c1 * o.c1, self.c0 * o.c1 + self.c1 * o.c0)

    def __eq__(self, o):
        return self.c0 == o.c0 and self.c1 == o.c1

    def inv(self):
        n = pow(self.c0 * self.c0 + self.c1 * self.c1, -1, Q_MOD)
        return Fq2(self.c0 * n, -self.c1 * n)

    def is_zero(self):
        return self.c0 == 0 and self.c1 == 0

    def __repr__(self):
        return "Fq2(%x,%x)" % (self.c0, self.c1)


class Fq1:
    """Fq wrapped so the curve code below is generic over Fq / Fq2."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v % Q_MOD

    def __add__(self, o):
        return Fq1(self.v + o.v)

    def __sub__(self, o):
        return Fq1(self.v - o.v)

    def __neg__(self):
        return Fq1(-self.v)

    def __mul__(self, o):
        if isinstance(o, int):
            return Fq1(self.v * o)
        return Fq1(self.v * o.v)

    def __eq__(self, o):
        return self.v == o.v

    def inv(self):
        return Fq1(pow(self.v, -1, Q_MOD))

    def is_zero(self):
        return self.v == 0

    def __repr__(self):
        return "Fq(%x)" % self.v


# ---------------------------------------------------------------- curves (affine; None = infinity)
G1_B = Fq1(4)
G2_B = Fq2(4, 4)
G1_GEN = (
    Fq1(0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb),
    Fq1(0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
)
G2_GEN = (
    Fq2(0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
        0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
    Fq2(0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
        0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be),
)


def on_curve(P, b):
    if P is None:
        return True
    x, y = P
    return y * y == x * x * x + b


def ec_neg(P):
    if P is None:
        return None
    return (P[0], -P[1])


def ec_add(P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if y1 == y2 and not y1.is_zero():
            lam = (x1 * x1 * 3) * (y1 * 2).inv()
        else:
            return None
    else:
        lam = (y2 - y1) * (x2 - x1).inv()
    x3 = lam * lam - x1 - x2
    y3 = lam * (x1 - x3) - y1
    return (x3, y3)


def ec_mul(P, k):
    k %= R_MOD
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, P)
        P = ec_add(P, P)
        k >>= 1
    return acc


def g1_mul(k):
    return ec_mul(G1_GEN, k)


def g2_mul(k):
    return ec_mul(G2_GEN, k)


def msm_naive(points, scalars):
    acc = None
    for P, s in zip(points, scalars):
        acc = ec_add(acc, ec_mul(P, s))
    return acc


# -------- limb encodings identical to the C ABI (Montgomery, LE u64 limbs; see include/zkg16.h)
def g1_to_limbs(P):
    """-> (12 u64 limbs x||y Montgomery, infinity flag)"""
    if P is None:
        return [0] * 12, 1
    return limbs64(fq_to_mont(P[0].v), 6) + limbs64(fq_to_mont(P[1].v), 6), 0


def g2_to_limbs(P):
    """-> (24 u64 limbs x.c0||x.c1||y.c0||y.c1 Montgomery, infinity flag)"""
    if P is None:
        return [0] * 24, 1
    x, y = P
    out = []
    for v in (x.c0, x.c1, y.c0, y.c1):
        out += limbs64(fq_to_mont(v), 6)
    return out, 0


def g1_from_limbs(l, inf=0):
    if inf:
        return None
    return (Fq1(fq_from_mont(from_limbs64(l[0:6]))), Fq1(fq_from_mont(from_limbs64(l[6:12]))))


def g2_from_limbs(l, inf=0):
    if inf:
        return None
    v = [fq_from_mont(from_limbs64(l[6 * i:6 * i + 6])) for i in range(4)]
    return (Fq2(v[0], v[1]), Fq2(v[2], v[3]))


# ---------------------------------------------------------------- NTT by definition
def dft_naive(a, inverse=False, coset=False):
    """ark-poly Radix2EvaluationDomain semantics:
       fft: out[k] = sum_i a[i] w^(ik);  coset fft: a[i] *= g^i first;
       ifft: out[i] = N^-1 sum_k a[k] w^(-ik);  coset ifft: then out[i] *= g^-i."""
    n = len(a)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    w = root_of_unity(log_n)
    if not inverse:
        if coset:
            a = [(x * pow(FR_GEN, i, R_MOD)) % R_MOD for i, x in enumerate(a)]
        pw = [pow(w, i, R_MOD) for i in range(n)]
        return [sum(a[i] * pw[(i * k) % n] for i in range(n)) % R_MOD for k in range(n)]
    winv = pow(w, -1, R_MOD)
    pw = [pow(winv, i, R_MOD) for i in range(n)]
    ninv = pow(n, -1, R_MOD)
    out = [(sum(a[k] * pw[(i * k) % n] for k in range(n)) * ninv) % R_MOD for i in range(n)]
    if coset:
        ginv = pow(FR_GEN, -1, R_MOD)
        out = [(x * pow(ginv, i, R_MOD)) % R_MOD for i, x in enumerate(out)]
    return out


def poly_eval(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R_MOD
    return acc


# ---------------------------------------------------------------- R1CS / QAP / Groth16 (known trapdoor)
def next_pow2(n):
    p = 1
    while p < n:
        p <<= 1
    return p


def r1cs_eval(rows, z):
    return [sum(c * z[j] for c, j in row) % R_MOD for row in rows]


def lagrange_at(tau, N):
    """L_i(tau) for the radix-2 domain of size N (tau not in the domain)."""
    log_n = N.bit_length() - 1
    w = root_of_unity(log_n)
    zt = (pow(tau, N, R_MOD) - 1) % R_MOD
    ninv = pow(N, -1, R_MOD)
    out = []
    wi = 1
    for _ in range(N):
        out.append(zt * ninv % R_MOD * wi % R_MOD * pow((tau - wi) % R_MOD, -1, R_MOD) % R_MOD)
        wi = wi * w % R_MOD
    return out


def qap_at_tau(A, B, C, num_inputs, num_vars, tau):
    """u_k(tau), v_k(tau), w_k(tau) per SURVEY.md A.7 (LibsnarkReduction incl. the input rows)."""
    nc = len(A)
    N = next_pow2(nc + num_inputs)
    L = lagrange_at(tau, N)
    u = [0] * num_vars
    v = [0] * num_vars
    w = [0] * num_vars
    for i in range(nc):
        for c, j in A[i]:
            u[j] = (u[j] + c * L[i]) % R_MOD
        for c, j in B[i]:
            v[j] = (v[j] + c * L[i]) % R_MOD
        for c, j in C[i]:
            w[j] = (w[j] + c * L[i]) % R_MOD
    for k in range(num_inputs):
        u[k] = (u[k] + L[nc + k]) % R_MOD
    return N, u, v, w


def witness_map_h(A, B, C, num_inputs, z):
    """h coefficients by exact polynomial division, independent of any FFT:
       h(X) = (a(X) b(X) - c(X)) / (X^N - 1) with a,b,c the interpolants over the domain."""
    nc = len(A)
    N = next_pow2(nc + num_inputs)
    a = r1cs_eval(A, z) + [0] * (N - nc)
    b = r1cs_eval(B, z) + [0] * (N - nc)
    c = r1cs_eval(C, z) + [0] * (N - nc)
    for i in range(num_inputs):
        a[nc + i] = z[i]
    ac = dft_naive(a, inverse=True)
    bc = dft_naive(b, inverse=True)
    cc = dft_naive(c, inverse=True)
    prod = [0] * (2 * N)
    for i, x in enumerate(ac):
        if x:
            for j, y in enumerate(bc):
                prod[i + j] = (prod[i + j] + x * y) % R_MOD
    for i, x in enumerate(cc):
        prod[i] = (prod[i] - x) % R_MOD
    # divide by X^N - 1: prod = h*(X^N - 1)  =>  h[i] = prod[i+N] + h[i+N] (h[j]=0 for j>=N)
    h = [0] * N
    for i in range(N - 1, -1, -1):
        hi = prod[i + N] + (h[i + N] if i + N < N else 0)
        h[i] = hi % R_MOD
    # remainder check: prod[i] + h[i] == 0
    for i in range(N):
        assert (prod[i] + h[i]) % R_MOD == 0, "R1CS not satisfied"
    return N, h


def groth16_setup_logs(A, B, C, num_inputs, num_vars, trap):
    """Discrete logs (w.r.t. the fixed generators g1, g2 = [g1s]G1, [g2s]G2) of every pk element."""
    tau, alpha, beta, gamma, delta = trap["tau"], trap["alpha"], trap["beta"], trap["gamma"], trap["delta"]
    N, u, v, w = qap_at_tau(A, B, C, num_inputs, num_vars, tau)
    dinv = pow(delta, -1, R_MOD)
    ginv = pow(gamma, -1, R_MOD)
    zt = (pow(tau, N, R_MOD) - 1) % R_MOD
    logs = {
        "N": N,
        "a_query": u,
        "b_query": v,
        "h_query": [pow(tau, i, R_MOD) * zt % R_MOD * dinv % R_MOD for i in range(N - 1)],
        "l_query": [(beta * u[k] + alpha * v[k] + w[k]) % R_MOD * dinv % R_MOD for k in range(num_inputs, num_vars)],
        "gamma_abc": [(beta * u[k] + alpha * v[k] + w[k]) % R_MOD * ginv % R_MOD for k in range(num_inputs)],
        "alpha": alpha, "beta": beta, "gamma": gamma, "delta": delta,
    }
    return logs


def groth16_prove_logs(logs, h, z, num_inputs, r, s):
    """Logs (a, b, c) of the proof elements, and the in-the-exponent check (A.7)."""
    a = (logs["alpha"] + sum(zk * uk for zk, uk in zip(z, logs["a_query"])) + r * logs["delta"]) % R_MOD
    b = (logs["beta"] + sum(zk * vk for zk, vk in zip(z, logs["b_query"])) + s * logs["delta"]) % R_MOD
    c = (s * a + r * b - r * s % R_MOD * logs["delta"]
         + sum(zk * lk for zk, lk in zip(z[num_inputs:], logs["l_query"]))
         + sum(hi * qi for hi, qi in zip(h, logs["h_query"]))) % R_MOD
    pub = sum(zk * gk for zk, gk in zip(z[:num_inputs], logs["gamma_abc"])) % R_MOD
    ok = (a * b - logs["alpha"] * logs["beta"] - pub * logs["gamma"] - c * logs["delta"]) % R_MOD == 0
    return a, b, c, ok


def rand_fr(rng):
    return rng.randrange(R_MOD)
