"""Reader (and, for the self-test, writer) of the off-box arkworks fixture format produced by the `dump_fixture` example in
INTEGRATION.md section 6 — the way to pin this repo's parity against the real ark-groth16 prover on a machine that has cargo.

Layout (little-endian): magic b"ZKG16FX1"; u64 num_instance, num_witness, num_constraints, nnz_a, nnz_b, nnz_c, n_h;
then, all as raw in-memory Montgomery limbs (u64) exactly as arkworks stores them:
  r[4] s[4]
  per matrix m in A, B, C: row_ptr[(nc+1)] (u64), col[nnz] (u64), coeff[nnz][4]
  z[(ni+nw)][4]
  pk: alpha_g1[12] beta_g1[12] beta_g2[24] delta_g1[12] delta_g2[24];
      a_query[(ni+nw)][12] + inf bytes padded to 8; b_g1_query likewise; b_g2_query[(ni+nw)][24] + inf; h_query[n_h][12] + inf;
      l_query[nw][12] + inf
  proof: a[12] b[24] c[12] + 3 inf bytes padded to 8."""
import numpy as np

MAGIC = b"ZKG16FX1"


def _pad8(n):
    return (n + 7) // 8 * 8


def load(path):
    raw = open(path, "rb").read()
    assert raw[:8] == MAGIC, "not an arkworks fixture"
    pos = [8]

    def u64(n):
        a = np.frombuffer(raw, dtype="<u8", count=n, offset=pos[0]).copy()
        pos[0] += 8 * n
        return a

    def u8(n):
        a = np.frombuffer(raw, dtype=np.uint8, count=n, offset=pos[0]).copy()
        pos[0] += _pad8(n)
        return a

    ni, nw, nc, nnz_a, nnz_b, nnz_c, n_h = (int(x) for x in u64(7))
    r, s = u64(4), u64(4)
    r1cs = {}
    for name, nnz in (("a", nnz_a), ("b", nnz_b), ("c", nnz_c)):
        r1cs[name] = (u64(nc + 1), u64(nnz).astype(np.uint32), u64(4 * nnz).reshape(nnz, 4))
    r1cs["num_inputs"], r1cs["num_constraints"] = ni, nc
    z = u64(4 * (ni + nw)).reshape(-1, 4)
    pk = dict(alpha_g1=u64(12), beta_g1=u64(12), beta_g2=u64(24), delta_g1=u64(12), delta_g2=u64(24))
    for key, inf_key, n, w in (("a_query", "a_inf", ni + nw, 12), ("b_g1_query", "b_g1_inf", ni + nw, 12), ("b_g2_query", "b_g2_inf", ni + nw, 24),
                               ("h_query", "h_inf", n_h, 12), ("l_query", "l_inf", nw, 12)):
        pk[key] = u64(w * n).reshape(n, w)
        pk[inf_key] = u8(n)
    proof = u64(48)
    inf = u8(3)
    assert pos[0] == len(raw), "trailing bytes"
    return dict(num_instance=ni, num_witness=nw, r=r, s=s, r1cs=r1cs, z=z, pk=pk, proof=proof, inf=inf)


def dump(path, fx):
    """The same format from Python (used by the self-test, which writes a fixture from this repo's own oracle)."""
    out = [MAGIC]
    r1cs, pk = fx["r1cs"], fx["pk"]
    ni, nc = r1cs["num_inputs"], r1cs["num_constraints"]
    nw = fx["z"].shape[0] - ni
    hdr = [ni, nw, nc] + [len(r1cs[m][1]) for m in ("a", "b", "c")] + [pk["h_query"].shape[0]]
    out.append(np.array(hdr, dtype="<u8").tobytes())
    out += [np.asarray(fx["r"], dtype="<u8").tobytes(), np.asarray(fx["s"], dtype="<u8").tobytes()]
    for m in ("a", "b", "c"):
        rp, col, cf = r1cs[m]
        out += [np.asarray(rp, dtype="<u8").tobytes(), np.asarray(col, dtype="<u8").tobytes(), np.asarray(cf, dtype="<u8").tobytes()]
    out.append(np.asarray(fx["z"], dtype="<u8").tobytes())
    for k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2"):
        out.append(np.asarray(pk[k], dtype="<u8").tobytes())
    for key, inf_key in (("a_query", "a_inf"), ("b_g1_query", "b_g1_inf"), ("b_g2_query", "b_g2_inf"), ("h_query", "h_inf"), ("l_query", "l_inf")):
        q = np.asarray(pk[key], dtype="<u8")
        out.append(q.tobytes())
        fl = np.asarray(pk.get(inf_key, np.zeros(q.shape[0])), dtype=np.uint8).tobytes()
        out.append(fl + b"\0" * (_pad8(len(fl)) - len(fl)))
    out.append(np.asarray(fx["proof"], dtype="<u8").tobytes())
    fl = np.asarray(fx["inf"], dtype=np.uint8).tobytes()
    out.append(fl + b"\0" * (8 - len(fl)))
    open(path, "wb").write(b"".join(out))
