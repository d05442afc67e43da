"""Wire formats either side of the hot path (scope row f-2), as the reference ships them over its JSON API
(src/arkworks/matrix_proof_of_work/io.rs:45-88, backend/matrix_proof.rs:116-120,196-198):
  * Proof  -> `serialize_compressed` (ark-bls12-381 0.4: zcash-style BLS12-381 encoding, A 48 B | B 96 B | C 48 B) -> base64 STANDARD
  * Fr hash -> 32-byte little-endian canonical -> base64; decoded with from_le_bytes_mod_order
Point encoding: big-endian x; top three bits of the first byte = compressed (1), infinity, y-is-lexicographically-largest;
G2 is x.c1 || x.c0 and compares y by (c1, c0).  Pure Python integers (six field elements per proof: not a hot path)."""
import base64

import numpy as np

Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
_RQ_INV = pow(1 << 384, -1, Q)
_RR_INV = pow(1 << 256, -1, R)


def _int(limbs):
    return sum(int(v) << (64 * i) for i, v in enumerate(limbs))


def _limbs(x, n):
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def _fq(limbs):            # Montgomery limbs -> canonical int
    return _int(limbs) * _RQ_INV % Q


def _fq_mont(x):
    return _limbs((x << 384) % Q, 6)


def g1_compress(p12, inf):
    if inf:
        return bytes([0xC0]) + bytes(47)
    x, y = _fq(p12[:6]), _fq(p12[6:])
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if y > (Q - y) % Q else 0)
    return bytes(b)


def g2_compress(p24, inf):
    if inf:
        return bytes([0xC0]) + bytes(95)
    x0, x1, y0, y1 = (_fq(p24[6 * i:6 * i + 6]) for i in range(4))
    ny0, ny1 = (-y0) % Q, (-y1) % Q
    largest = (y1, y0) > (ny1, ny0)
    b = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if largest else 0)
    return bytes(b)


def _sqrt_fq(a):
    r = pow(a, (Q + 1) // 4, Q)            # q = 3 mod 4
    return r if r * r % Q == a % Q else None


def _sqrt_fq2(a0, a1):
    """sqrt in Fq[u]/(u^2+1) (complex method); returns (c0, c1) or None."""
    if a1 == 0:
        r = _sqrt_fq(a0)
        if r is not None:
            return r, 0
        r = _sqrt_fq((-a0) % Q)
        return (0, r) if r is not None else None
    n = _sqrt_fq((a0 * a0 + a1 * a1) % Q)
    if n is None:
        return None
    inv2 = pow(2, -1, Q)
    for cand in ((a0 + n) * inv2 % Q, (a0 - n) * inv2 % Q):
        c0 = _sqrt_fq(cand)
        if c0:
            c1 = a1 * pow(2 * c0, -1, Q) % Q
            if (c0 * c0 - c1 * c1) % Q == a0 % Q:
                return c0, c1
    return None


def _subgroup_check(group, limbs):
    """ark's deserialize_compressed validates: the point must lie in the prime-order subgroup (host-side zkg16_point_check)."""
    from .device import point_check
    if not point_check(group, limbs):
        raise ValueError("%s point is not in the prime-order subgroup" % group)


def g1_decompress(b, validate=True):
    """Strict, as `G1Affine::deserialize_compressed`: canonical x (< q), a clean infinity encoding, on the curve, in the subgroup."""
    if len(b) != 48 or not b[0] & 0x80:
        raise ValueError("not a compressed G1 point")
    if b[0] & 0x40:
        if b[0] != 0xC0 or any(b[1:]):
            raise ValueError("non-canonical encoding of the point at infinity")
        return np.zeros(12, dtype=np.uint64), 1
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big")
    if x >= Q:
        raise ValueError("x coordinate is not reduced")
    y = _sqrt_fq((x * x * x + 4) % Q)
    if y is None:
        raise ValueError("x not on curve")
    if (y > (Q - y) % Q) != bool(b[0] & 0x20):
        y = (Q - y) % Q
    out = np.concatenate([_fq_mont(x), _fq_mont(y)])
    if validate:
        _subgroup_check("g1", out)
    return out, 0


_G1_STATUS = {1: "not a compressed G1 point", 2: "non-canonical encoding of the point at infinity", 3: "x coordinate is not reduced",
              4: "x not on curve", 5: "g1 point is not in the prime-order subgroup"}


def g1_decompress_many(b, n, validate=True):
    """n compressed G1 points back to back -> (n x 12 Montgomery limbs, n infinity flags), the rules of g1_decompress applied by the
    library's host code (zkg16_g1_decompress): a verifying key with hundreds of gamma_abc_g1 points is decoded in a few ms instead
    of ~0.2 ms of Python big-integer square root per point."""
    return _decompress_many("g1", b, n, validate)


def _decompress_many(group, b, n, validate=True):
    import ctypes as C
    from . import _lib
    size, width = (48, 12) if group == "g1" else (96, 24)
    if len(b) != size * n:
        raise ValueError("not %d compressed %s points" % (n, group.upper()))
    raw = np.frombuffer(bytes(b), dtype=np.uint8)
    out = np.zeros((n, width), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    status = (C.c_int * max(n, 1))()
    fn = _lib.load().zkg16_g1_decompress if group == "g1" else _lib.load().zkg16_g2_decompress
    rc = fn(raw.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p), 1 if validate else 0, status)
    if rc:
        for i in range(n):
            if status[i]:
                raise ValueError("%s point %d: %s" % (group.upper(), i, _G1_STATUS.get(status[i], "invalid").replace("G1", group.upper()).replace("g1", group)))
        raise ValueError("%s decompression failed" % group.upper())
    return out, inf


def g2_decompress_many(b, n, validate=True):
    """n compressed G2 points back to back -> (n x 24 limbs, infinity flags) by the library's host code (zkg16_g2_decompress)."""
    return _decompress_many("g2", b, n, validate)


def points_compress(group, points, inf=None):
    """(n x 12 | n x 24 Montgomery limbs, optional infinity flags) -> the compressed bytes back to back (zkg16_points_compress)."""
    import ctypes as C
    from . import _lib
    width, size = (12, 48) if group == "g1" else (24, 96)
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, width)
    n = pts.shape[0]
    flags = np.zeros(n, dtype=np.uint8) if inf is None else np.ascontiguousarray(inf, dtype=np.uint8).reshape(n)
    out = np.zeros(size * n, dtype=np.uint8)
    rc = _lib.load().zkg16_points_compress(1 if group == "g1" else 2, pts.ctypes.data_as(C.c_void_p), flags.ctypes.data_as(C.c_void_p), n,
                                          out.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError("point compression failed")
    return out.tobytes()


def fq_to_le_bytes(limbs):
    """n x 6 Montgomery limbs -> 48 n bytes, little-endian canonical (zkg16_fq_to_le_bytes)."""
    import ctypes as C
    from . import _lib
    v = np.ascontiguousarray(limbs, dtype=np.uint64).reshape(-1, 6)
    out = np.zeros(48 * v.shape[0], dtype=np.uint8)
    if _lib.load().zkg16_fq_to_le_bytes(v.ctypes.data_as(C.c_void_p), v.shape[0], out.ctypes.data_as(C.c_void_p)):
        raise ValueError("field serialization failed")
    return out.tobytes()


def fq_from_le_bytes(b, n):
    """48 n little-endian canonical bytes -> n x 6 Montgomery limbs; refuses a value >= q (zkg16_fq_from_le_bytes)."""
    import ctypes as C
    from . import _lib
    if len(b) != 48 * n:
        raise ValueError("not %d field elements" % n)
    raw = np.frombuffer(bytes(b), dtype=np.uint8)
    out = np.zeros((n, 6), dtype=np.uint64)
    if _lib.load().zkg16_fq_from_le_bytes(raw.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p)):
        raise ValueError("field element is not reduced")
    return out


def g2_decompress(b, validate=True):
    if len(b) != 96 or not b[0] & 0x80:
        raise ValueError("not a compressed G2 point")
    if b[0] & 0x40:
        if b[0] != 0xC0 or any(b[1:]):
            raise ValueError("non-canonical encoding of the point at infinity")
        return np.zeros(24, dtype=np.uint64), 1
    x1 = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], "big")
    x0 = int.from_bytes(b[48:], "big")
    if x0 >= Q or x1 >= Q:
        raise ValueError("x coordinate is not reduced")
    # x^3 + 4(1 + u)
    a0 = (x0 * x0 - x1 * x1) % Q
    a1 = 2 * x0 * x1 % Q
    c0 = (a0 * x0 - a1 * x1 + 4) % Q
    c1 = (a0 * x1 + a1 * x0 + 4) % Q
    s = _sqrt_fq2(c0, c1)
    if s is None:
        raise ValueError("x not on curve")
    y0, y1 = s
    if ((y1, y0) > ((-y1) % Q, (-y0) % Q)) != bool(b[0] & 0x20):
        y0, y1 = (-y0) % Q, (-y1) % Q
    out = np.concatenate([_fq_mont(x0), _fq_mont(x1), _fq_mont(y0), _fq_mont(y1)])
    if validate:
        _subgroup_check("g2", out)
    return out, 0


def proof_serialize_compressed(proof48, inf3):
    """(A | B | C affine Montgomery limbs, infinity flags) -> 192 bytes (io.rs:45-51 before base64)."""
    p = np.asarray(proof48, dtype=np.uint64)
    return (points_compress("g1", p[:12], [inf3[0]]) + points_compress("g2", p[12:36], [inf3[1]]) + points_compress("g1", p[36:], [inf3[2]]))


def proof_deserialize_compressed(b):
    if len(b) != 192:
        raise ValueError("a compressed proof is 192 bytes")
    ac, iac = g1_decompress_many(bytes(b[:48]) + bytes(b[144:]), 2)
    bb, ib = g2_decompress_many(b[48:144], 1)
    return np.concatenate([ac[0], bb[0], ac[1]]), np.array([iac[0], ib[0], iac[1]], dtype=np.uint8)


def encode_proof(proof48, inf3):
    return base64.standard_b64encode(proof_serialize_compressed(proof48, inf3)).decode()


def decode_proof(s):
    return proof_deserialize_compressed(base64.standard_b64decode(s))


def encode_hash(fr_mont_limbs):
    """Fr (Montgomery limbs) -> into_bigint().to_bytes_le() -> base64 (matrix_proof.rs:116-120)."""
    v = _int(fr_mont_limbs) * _RR_INV % R
    return base64.standard_b64encode(v.to_bytes(32, "little")).decode()


def decode_hash(s):
    """base64 -> Fr::from_le_bytes_mod_order (matrix_proof.rs:196-198) -> Montgomery limbs"""
    v = int.from_bytes(base64.standard_b64decode(s), "little") % R
    return _limbs((v << 256) % R, 4)


def vk_serialize_compressed(vk):
    """VerifyingKey -> ark `CanonicalSerialize` compressed bytes: alpha_g1 (48) | beta_g2 (96) | gamma_g2 (96) | delta_g2 (96) |
    len(gamma_abc_g1) as u64 LE | gamma_abc_g1 (48 each)   (ark-groth16 data_structures.rs field order; Vec = u64 length
    prefix).  The reference ships the *prepared* key: pvk_serialize_compressed below."""
    gabc = np.asarray(vk["gamma_abc_g1"], dtype=np.uint64).reshape(-1, 12)
    g2s = np.stack([np.asarray(vk[k], dtype=np.uint64).reshape(24) for k in ("beta_g2", "gamma_g2", "delta_g2")])
    return (points_compress("g1", vk["alpha_g1"]) + points_compress("g2", g2s) + len(gabc).to_bytes(8, "little") + points_compress("g1", gabc))


# Untrusted bytes: every length prefix is checked against the bytes that are actually there BEFORE anything is allocated
# or indexed (a truncated or patched key must come back as ValueError, which the handlers answer with valid = False).
_MAX_INSTANCE = 1 << 24


def vk_deserialize_compressed(b):
    if len(b) < 344:
        raise ValueError("verifying key is truncated")
    n = int.from_bytes(b[336:344], "little")
    if n < 1 or n > _MAX_INSTANCE or len(b) != 344 + 48 * n:
        raise ValueError("verifying key length does not match its gamma_abc_g1 count")
    g1s, _ = g1_decompress_many(bytes(b[:48]) + bytes(b[344:344 + 48 * n]), n + 1)          # (infinity decodes to zero limbs)
    g2s, _ = g2_decompress_many(b[48:336], 3)
    return dict(alpha_g1=g1s[0], beta_g2=g2s[0], gamma_g2=g2s[1], delta_g2=g2s[2], gamma_abc_g1=g1s[1:])


def encode_vk(vk):
    return base64.standard_b64encode(vk_serialize_compressed(vk)).decode()


def decode_vk(s):
    return vk_deserialize_compressed(base64.standard_b64decode(s))


# ---- PreparedVerifyingKey (what the reference's handlers return: encode_pvk, io.rs:62-77; matrix_proof.rs:134-136)
# ark-groth16 0.4 data_structures.rs: PreparedVerifyingKey { vk, alpha_g1_beta_g2: Fq12, gamma_g2_neg_pc: G2Prepared, delta_g2_neg_pc: G2Prepared },
# derived CanonicalSerialize = the fields in order.  Fq12 = 12 base-field elements, each 48 bytes little-endian canonical, in
# tower order; G2Prepared { ell_coeffs: Vec<(Fq2, Fq2, Fq2)>, infinity: bool } = u64 LE length, 6 x 48 bytes per triple, 1 byte.
# The coefficient VALUES follow ark-ec's line-function formulas as restated in csrc/verify.hip (un-vendored crate): the layout is
# the published one, byte parity with a real arkworks pvk is unpinned (no fixture in the reference, no Rust toolchain here).
def _fq_le(limbs6):
    return _fq(limbs6).to_bytes(48, "little")


def _fq_from_le(b):
    v = int.from_bytes(b, "little")
    if v >= Q:
        raise ValueError("field element is not reduced")
    return _fq_mont(v)


def _prepared_bytes(coeffs):
    c = np.asarray(coeffs, dtype=np.uint64).reshape(-1, 36)
    return len(c).to_bytes(8, "little") + fq_to_le_bytes(c.reshape(-1, 6)) + b"\x00"                     # infinity = false


def pvk_serialize_compressed(pvk):
    """pvk: device.pvk_prepare(vk) -> bytes."""
    ab = np.asarray(pvk["alpha_beta"], dtype=np.uint64)
    return (vk_serialize_compressed(pvk) + fq_to_le_bytes(ab.reshape(12, 6)) + _prepared_bytes(pvk["gamma_neg_pc"]) + _prepared_bytes(pvk["delta_neg_pc"]))


PVK_COEFFS = 68          # line-coefficient triples of a BLS12-381 G2Prepared (63 doublings + 5 additions of |z|)


def pvk_deserialize_compressed(b):
    if len(b) < 344:
        raise ValueError("prepared verifying key is truncated")
    n = int.from_bytes(b[336:344], "little")
    if n < 1 or n > _MAX_INSTANCE:
        raise ValueError("prepared verifying key: implausible gamma_abc_g1 count")
    off = 344 + 48 * n
    if len(b) != off + 576 + 2 * (8 + 288 * PVK_COEFFS + 1):
        raise ValueError("prepared verifying key length does not match its counts")
    pvk = vk_deserialize_compressed(b[:off])
    pvk["alpha_beta"] = fq_from_le_bytes(b[off:off + 576], 12).reshape(72)
    off += 576
    for name in ("gamma_neg_pc", "delta_neg_pc"):
        m = int.from_bytes(b[off:off + 8], "little")
        off += 8
        if m != PVK_COEFFS or off + 288 * m + 1 > len(b):
            raise ValueError("G2Prepared coefficient count is not %d" % PVK_COEFFS)
        rows = fq_from_le_bytes(b[off:off + 288 * m], 6 * m).reshape(m, 36)
        off += 288 * m
        if b[off] not in (0, 1) or (b[off] == 1) != (m == 0):
            raise ValueError("inconsistent G2Prepared infinity flag")
        off += 1
        pvk[name] = rows
    if off != len(b):
        raise ValueError("trailing bytes after the prepared verifying key")
    return pvk


def encode_pvk(vk):
    """VerifyingKey dict -> prepare_verifying_key -> serialize_compressed -> base64 (io.rs:62-68)."""
    from .device import pvk_prepare
    return base64.standard_b64encode(pvk_serialize_compressed(pvk_prepare(vk))).decode()


def decode_pvk(s):
    return pvk_deserialize_compressed(base64.standard_b64decode(s))
