"""Workload generators mirroring the reference's bench drivers (bench/matrix.py:10-24: all-ones n x n matrices,
sizes 2^k) for the in-process harness: shapes follow SURVEY.md Appendix B.  Pure Python/numpy — no oracle, no
reference code.  The C++ circuit builder (real MatrixCircuit + Poseidon constraints) replaces the synthetic
rows when it lands; the shapes (nc, instance, witness, domain) and the witness value mix are already exact."""
import numpy as np

R_MOD = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
MASK64 = (1 << 64) - 1

# standard BLS12-381 generators, Montgomery limbs in the C-ABI layout
_G1_GEN = [0x5cb38790fd530c16, 0x7817fc679976fff5, 0x154f95c7143ba1c1, 0xf0ae6acdf3d0e747, 0xedce6ecc21dbf440, 0x120177419e0bfb75,
           0xbaac93d50ce72271, 0x8c22631a7918fd8e, 0xdd595f13570725ce, 0x51ac582950405194, 0x0e1c8c3fad0059c0, 0x0bbc3efc5008a26a]
_G2_GEN = [0xf5f28fa202940a10, 0xb3f5fb2687b4961a, 0xa1a893b53e2ae580, 0x9894999d1a3caee9, 0x6f67b7631863366b, 0x058191924350bcd7,
           0xa5a9c0759e23f606, 0xaaa0c59dbccd60c3, 0x3bb17e18e2867806, 0x1b1ab6cc8541b367, 0xc2b6ed0ef2158547, 0x11922a097360edf3,
           0x4c730af860494c4a, 0x597cfa1f5e369c5a, 0xe7e6856caa0a635a, 0xbbefb5e96e0d495f, 0x07d3a975f0ef25a2, 0x0083fd8e7e80dae5,
           0xadc0fc92df64b05d, 0x18aa270a2b1461dc, 0x86adac6a3be4eba0, 0x79495c4ec93da33a, 0xe7175850a43ccaed, 0x0b2bc2a163de1bf2]


def g1_generator():
    return np.array(_G1_GEN, dtype=np.uint64)


def g2_generator():
    return np.array(_G2_GEN, dtype=np.uint64)


# ------------------------------------------------------------------------------------------------
# Matmul-shaped synthetic workload: exact (nc, num_instance, num_witness) of the
# reference's MatrixCircuit for an n x n product with all-ones inputs, row shapes of its two row families
# (matmul rows: 1 term per side; Poseidon rows: short linear combinations), and its witness value mix
# (ones / zeros / uniform).  Used until the C++ circuit builder exists; satisfiable by construction.
def matmul_shape(n):
    half = (n * n + 1) // 2
    pose = 3 * (half * 265 - 5)
    nc = 2 * n ** 3 + pose + 3
    w = n ** 3 + 4 * n * n + pose
    return dict(n=n, nc=nc, num_instance=4, num_witness=w, num_vars=4 + w, domain=1 << (nc + 4 - 1).bit_length())


def matmul_like_r1cs(n, seed=0x5EED0001):
    """Returns (r1cs arrays, z Montgomery (num_vars,4) u64, shape dict).
    Row families (Appendix B): n^3 product rows a*b = d (each defines one product witness, value 1 for the
    all-ones inputs), n^3 duplicate rows (the reference's second `mul_equals` constraint on the same triple),
    3(ceil(n^2/2)*265-5) Poseidon-like rows (1-3 term LCs with random coefficients, each defining one
    ~uniform witness), and 3 `enforce_equal`-like rows v*1 = v.  Free variables: instance (4) + the 4n^2
    matrix entries / pre-allocated zeros.  Satisfiable by construction (triangular)."""
    shp = matmul_shape(n)
    nc, nv = shp["nc"], shp["num_vars"]
    rng = np.random.default_rng(seed)
    n3 = n ** 3
    pose = nc - 2 * n3 - 3
    n_free = 4 + 4 * n * n
    assert nv == n_free + n3 + pose
    z_int = [0] * nv
    z_int[0] = 1
    for k in (1, 2, 3):
        z_int[k] = int(rng.integers(1, 1 << 62))      # hash-like public inputs
    n_ones = 2 * n * n
    for k in range(4, 4 + n_ones):
        z_int[k] = 1                                   # all-ones matrices (bench/matrix.py:11); the other 2n^2 stay 0
    a_cols, a_rp, a_cf = [], [0], []
    b_cols, b_rp, b_cf = [], [0], []
    c_cols = []
    small = [1, 2, 3, 5, 7]
    mm = []
    for i in range(n3):
        mm.append((4 + int(rng.integers(0, n_ones)), 4 + int(rng.integers(0, n_ones))))
    for i in range(nc):
        if i < 2 * n3:
            ja, jb = mm[i % n3]
            ta, tb = [(1, ja)], [(1, jb)]
            out = n_free + (i % n3)
        elif i < 2 * n3 + pose:
            out = n_free + n3 + (i - 2 * n3)
            ka, kb = int(rng.integers(1, 4)), int(rng.integers(1, 3))
            lo = max(0, out - 4096)
            ta = [(int(rng.integers(1, 1 << 62)) if rng.random() < .5 else small[int(rng.integers(0, 5))], int(rng.integers(lo, out))) for _ in range(ka)]
            tb = [(int(rng.integers(1, 1 << 62)) if rng.random() < .3 else 1, int(rng.integers(lo, out))) for _ in range(kb)]
        else:
            out = int(rng.integers(n_free, nv))
            ta, tb = [(1, out)], [(1, 0)]
        if i < n3 or 2 * n3 <= i < 2 * n3 + pose:
            av = sum(c * z_int[j] for c, j in ta) % R_MOD
            bv = sum(c * z_int[j] for c, j in tb) % R_MOD
            z_int[out] = av * bv % R_MOD
        for c, j in ta:
            a_cols.append(j); a_cf.append(c)
        for c, j in tb:
            b_cols.append(j); b_cf.append(c)
        a_rp.append(len(a_cols)); b_rp.append(len(b_cols))
        c_cols.append(out)

    def to_limbs(vals):
        out = np.zeros((len(vals), 4), dtype=np.uint64)
        for i, v in enumerate(vals):
            out[i, 0] = v & MASK64; out[i, 1] = (v >> 64) & MASK64; out[i, 2] = (v >> 128) & MASK64; out[i, 3] = v >> 192
        return out

    def from_canon(limbs):      # canonical -> Montgomery (x * 2^256 mod r), python ints
        vals = [(int(a) | int(b) << 64 | int(c) << 128 | int(d) << 192) for a, b, c, d in limbs]
        return to_limbs([(v << 256) % R_MOD for v in vals])

    r1cs = dict(
        a=(np.array(a_rp, dtype=np.uint64), np.array(a_cols, dtype=np.uint32), from_canon(to_limbs(a_cf))),
        b=(np.array(b_rp, dtype=np.uint64), np.array(b_cols, dtype=np.uint32), from_canon(to_limbs(b_cf))),
        c=(np.arange(nc + 1, dtype=np.uint64), np.array(c_cols, dtype=np.uint32), from_canon(to_limbs([1] * nc))),
        num_inputs=4, num_constraints=nc)
    return r1cs, from_canon(to_limbs(z_int)), shp


def prime_like_r1cs(nc=230000, seed=0x5EED0004):
    """Shape of the reference's Fermat-prime circuit (BASELINE configs[4]; SURVEY.md 8d row 4: ~1.3-2.6e5 constraints, domain
    2^18, 7 SHA-256 compressions + comparisons => nearly every witness is a bit): a satisfiable boolean circuit of AND / XOR /
    booleanity rows over bit-valued witnesses plus a sprinkling of 20-bit recomposition rows with field-sized quotients.
    This is a workload generator for the hot path (pathological bucket skew: every z-side scalar is 0 or 1), not a mirror of
    prime_snark/prime_circut.rs — its SHA-256 gadget constraint layout cannot be checked without ark-crypto-primitives.
    Returns (r1cs arrays, z Montgomery, shape dict)."""
    rng = np.random.default_rng(seed)
    ni = 1 + 1 + 256                                    # one, x, and the 256 digest bits as instance (prime_circut.rs:98-105)
    z = [1, int(rng.integers(1, 1 << 62))] + [int(b) for b in rng.integers(0, 2, size=256)]
    rows_a, rows_b, rows_c = [], [], []

    def new_var(v):
        z.append(v % R_MOD)
        return len(z) - 1

    bits = list(range(2, 258))                          # pool of bit-valued variables
    R1 = R_MOD - 1
    while len(rows_a) < nc:
        kind = len(rows_a) % 16
        a, b = bits[int(rng.integers(len(bits)))], bits[int(rng.integers(len(bits)))]
        if kind < 7:                                    # c = a AND b            : a * b = c
            c = new_var(z[a] * z[b])
            rows_a.append([(1, a)]); rows_b.append([(1, b)]); rows_c.append([(1, c)])
            bits.append(c)
        elif kind < 13:                                 # c = a XOR b            : (2a) * b = a + b - c
            c = new_var(z[a] ^ z[b])
            rows_a.append([(2, a)]); rows_b.append([(1, b)]); rows_c.append([(1, a), (1, b), (R1, c)])
            bits.append(c)
        elif kind < 15:                                 # booleanity of a fresh bit: b * (1 - b) = 0
            c = new_var(int(rng.integers(0, 2)))
            rows_a.append([(1, c)]); rows_b.append([(1, 0), (R1, c)]); rows_c.append([])
            bits.append(c)
        else:                                           # 20-bit recomposition times a random field element = witnessed product
            sel = [bits[int(rng.integers(len(bits)))] for _ in range(20)]
            val = sum(z[v] << k for k, v in enumerate(sel))
            q = int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) % R_MOD
            qv = new_var(q)
            pv = new_var(val * q)
            rows_a.append([(1 << k, v) for k, v in enumerate(sel)]); rows_b.append([(1, qv)]); rows_c.append([(1, pv)])
        if len(bits) > 4096:
            bits = bits[-4096:]

    def to_limbs(vals):
        out = np.zeros((len(vals), 4), dtype=np.uint64)
        for i, v in enumerate(vals):
            out[i, 0] = v & MASK64; out[i, 1] = (v >> 64) & MASK64; out[i, 2] = (v >> 128) & MASK64; out[i, 3] = v >> 192
        return out

    def mont(vals):
        return to_limbs([(v << 256) % R_MOD for v in vals])

    def csr(rows):
        rp, col, cf = [0], [], []
        for row in rows:
            for c, j in row:
                col.append(j); cf.append(c % R_MOD)
            rp.append(len(col))
        return np.array(rp, dtype=np.uint64), np.array(col, dtype=np.uint32), mont(cf) if cf else np.zeros((0, 4), dtype=np.uint64)

    nv = len(z)
    r1cs = dict(a=csr(rows_a), b=csr(rows_b), c=csr(rows_c), num_inputs=ni, num_constraints=nc)
    shp = dict(nc=nc, num_instance=ni, num_witness=nv - ni, num_vars=nv, domain=1 << (nc + ni - 1).bit_length())
    return r1cs, mont(z), shp
