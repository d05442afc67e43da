// G2 arithmetic with every Fq2 value SPLIT OVER A PAIR OF LANES: lane 2j holds c0, lane 2j + 1 holds c1 (h = lane & 1).
//
// Why: one lane holding whole Fq2 coordinates needs 478 registers for the bucket accumulation (msm_accumulate_kernel<Fq2U>: 112 for the
// running sum, 56 for the base, the temporaries of the formula and four 14-limb operands per fused product), so a SIMD holds ONE such
// wave and nothing covers its waits (80 % issuing, profiles/rocprofv3_pmc_r3_sq.txt), and no G1 wave (255 registers) fits beside it.
// With the halves on two lanes every lane carries half the state (under 256 registers: two waves per SIMD, or a G1 wave beside it) and
// does exactly half the multiplications:
//   product   c0 = a0 b0 - a1 b1 (lane 0),  c1 = a0 b1 + a1 b0 (lane 1): each ONE fqu_mul2 (two half-products, one reduction) of the
//             lane's own halves and its partner's, fetched with DPP quad_perm [1, 0, 3, 2] (a VALU move, no LDS);
//   square    c0 = (a0 + a1)(a0 - a1),  c1 = 2 a0 a1: ONE fqu_mul per lane;
//   add / sub / double / negate: limb-wise on the lane's own half.
// Value bounds are those of the one-lane lazy form (ffu.cuh fq2u_mul_lazy, ec.cuh xyzz_madd_lazy): products < 2q per component,
// squares < 2q / < 4q, everything multiplied <= 127 q.  Control flow must be uniform per pair (both lanes of a pair take every branch
// together): DPP reads the partner's registers.
#pragma once
#include "../../zksnark-finalproject_amd/csrc/ec.cuh"

namespace zk {

// (the host pass of a kernel that uses these only needs them to parse: the device-only pieces are stubbed there)
__device__ __forceinline__ uint32_t h2_swap_u32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
#else
    return v;
#endif
}
__device__ __forceinline__ FqU h2_swap(const FqU &a) {      // the partner lane's value
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = h2_swap_u32(a.l[i]);
    return r;
}
__device__ __forceinline__ FqU h2_sel(bool c, const FqU &x, const FqU &y) {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = c ? x.l[i] : y.l[i];
    return r;
}
__device__ __forceinline__ bool h2_both(bool mine) { return mine && h2_swap_u32(mine ? 1u : 0u) != 0u; }

#if defined(__HIP_DEVICE_COMPILE__)
// own component of a * b; 28 argument registers + the parity
__device__ __noinline__ __attribute__((weak)) FqURet h2_mul_call(zk_v4u a0, zk_v4u a1, zk_v4u a2, zk_v2u a3, zk_v4u b0, zk_v4u b1, zk_v4u b2, zk_v2u b3, int h) {
    FqU a, b;
    a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w; a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w;
    a.l[8] = a2.x; a.l[9] = a2.y; a.l[10] = a2.z; a.l[11] = a2.w; a.l[12] = a3.x; a.l[13] = a3.y;
    b.l[0] = b0.x; b.l[1] = b0.y; b.l[2] = b0.z; b.l[3] = b0.w; b.l[4] = b1.x; b.l[5] = b1.y; b.l[6] = b1.z; b.l[7] = b1.w;
    b.l[8] = b2.x; b.l[9] = b2.y; b.l[10] = b2.z; b.l[11] = b2.w; b.l[12] = b3.x; b.l[13] = b3.y;
    const bool odd = h != 0;
    // lane 1 sends -b1 (lane 0 needs a1 * (-b1)), lane 0 sends b0
    const FqU send = h2_sel(odd, fqu_sub<128>(FqU::zero(), b), b);
    const FqU pb = h2_swap(send), pa = h2_swap(a);
    // lane 0: a0 b0 + a1 (-b1);  lane 1: a0 b1 + a1 b0
    const FqU r = fqu_mul2(h2_sel(odd, pa, a), b, h2_sel(odd, a, pa), pb);
    FqURet o;
#pragma unroll
    for (int i = 0; i < 14; i++) o.l[i] = r.l[i];
    return o;
}
#endif
__device__ __forceinline__ FqU h2_mul(const FqU &a, const FqU &b, int h) {
#if defined(__HIP_DEVICE_COMPILE__)
    zk_v4u a0 = {a.l[0], a.l[1], a.l[2], a.l[3]}, a1 = {a.l[4], a.l[5], a.l[6], a.l[7]}, a2 = {a.l[8], a.l[9], a.l[10], a.l[11]};
    zk_v2u a3 = {a.l[12], a.l[13]};
    zk_v4u b0 = {b.l[0], b.l[1], b.l[2], b.l[3]}, b1 = {b.l[4], b.l[5], b.l[6], b.l[7]}, b2 = {b.l[8], b.l[9], b.l[10], b.l[11]};
    zk_v2u b3 = {b.l[12], b.l[13]};
    const FqURet r = h2_mul_call(a0, a1, a2, a3, b0, b1, b2, b3, h);
    FqU o;
#pragma unroll
    for (int i = 0; i < 14; i++) o.l[i] = r.l[i];
    return o;
#else
    (void)b; (void)h;
    return a;
#endif
}
// own component of a^2 (components <= 84 q as in f_sqr(Fq2U)): lane 0 (a0 + a1)(a0 - a1) < 2q, lane 1 2 a0 a1 < 4q
__device__ __forceinline__ FqU h2_sqr(const FqU &a, int h) {
    const bool odd = h != 0;
    const FqU pa = h2_swap(a);
    const FqU x = h2_sel(odd, pa, fqu_add(a, pa));
    const FqU y = h2_sel(odd, a, fqu_sub<128>(a, pa));
    const FqU r = fqu_mul(x, y);
    FqU t;
#pragma unroll
    for (int i = 0; i < 14; i++) t.l[i] = odd ? r.l[i] : 0u;
    return fqu_add(r, t);
}
__device__ __forceinline__ bool h2_is_zero(const FqU &a) { return h2_both(a.is_zero()); }                 // exact zero (flags)
__device__ __forceinline__ bool h2_is_zero_mod(const FqU &a) { return h2_both(fqu_is_zero_mod(a)); }      // 0 mod q in both components

// One half of an XYZZ / affine G2 point.
struct H2Affine { FqU x, y; };
struct H2XYZZ { FqU x, y, zz, zzz; };

// memory layout = Affine<Fq2U> / XYZZ<Fq2U> (x.c0 x.c1 y.c0 y.c1 ...; 56 bytes per component): lane h reads / writes component h
__device__ __forceinline__ FqU h2_ld(const FqU *p) {
    const uint2 *q = reinterpret_cast<const uint2 *>(p);
    FqU v;
#pragma unroll
    for (int i = 0; i < 7; i++) { const uint2 w = q[i]; v.l[2 * i] = w.x; v.l[2 * i + 1] = w.y; }
    return v;
}
__device__ __forceinline__ void h2_st(FqU *p, const FqU &v) {
    uint2 *q = reinterpret_cast<uint2 *>(p);
#pragma unroll
    for (int i = 0; i < 7; i++) q[i] = make_uint2(v.l[2 * i], v.l[2 * i + 1]);
}
__device__ __forceinline__ H2Affine h2_ld_affine(const Affine<Fq2U> *p, int h) {
    const FqU *f = reinterpret_cast<const FqU *>(p);
    return H2Affine{h2_ld(f + h), h2_ld(f + 2 + h)};
}
__device__ __forceinline__ H2XYZZ h2_ld_xyzz(const XYZZ<Fq2U> *p, int h) {
    const FqU *f = reinterpret_cast<const FqU *>(p);
    return H2XYZZ{h2_ld(f + h), h2_ld(f + 2 + h), h2_ld(f + 4 + h), h2_ld(f + 6 + h)};
}
__device__ __forceinline__ void h2_st_xyzz(XYZZ<Fq2U> *p, const H2XYZZ &v, int h) {
    FqU *f = reinterpret_cast<FqU *>(p);
    h2_st(f + h, v.x); h2_st(f + 2 + h, v.y); h2_st(f + 4 + h, v.zz); h2_st(f + 6 + h, v.zzz);
}
__device__ __forceinline__ H2XYZZ h2_inf() { return H2XYZZ{FqU::zero(), FqU::zero(), FqU::zero(), FqU::zero()}; }
__device__ __forceinline__ FqU h2_one(int h) {      // Fq2 one = (1, 0)
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = h ? 0u : FqUP::one(i);
    return r;
}
__device__ __forceinline__ bool h2_is_inf(const H2XYZZ &p) { return h2_is_zero(p.zz); }
__device__ __forceinline__ bool h2_is_inf(const H2Affine &p) { return h2_both(p.x.is_zero() && p.y.is_zero()); }

// 2 * (affine q), q != inf  (ec.cuh xyzz_dbl_affine)
__device__ __forceinline__ H2XYZZ h2_dbl_affine(const H2Affine &q, int h) {
    const FqU U = fqu_dbl(q.y);
    const FqU V = h2_sqr(U, h);
    const FqU W = h2_mul(U, V, h);
    const FqU S = h2_mul(q.x, V, h);
    const FqU X2 = h2_sqr(q.x, h);
    const FqU M = fqu_add(fqu_dbl(X2), X2);
    H2XYZZ r;
    r.x = fqu_sub<32>(h2_sqr(M, h), fqu_dbl(S));
    r.y = fqu_sub<32>(h2_mul(M, fqu_sub<64>(S, r.x), h), h2_mul(W, q.y, h));
    r.zz = V;
    r.zzz = W;
    return r;
}

// acc += (neg ? -q : q)  (ec.cuh xyzz_madd_lazy, same formula, same bounds, same exceptional cases)
__device__ __forceinline__ void h2_madd(H2XYZZ &acc, const H2Affine &q_in, bool neg, int h) {
    if (h2_is_inf(q_in)) return;
    H2Affine q = q_in;
    if (neg) q.y = fqu_neg(q.y);
    if (h2_is_inf(acc)) {
        acc = H2XYZZ{q.x, q.y, h2_one(h), h2_one(h)};
        return;
    }
    const FqU U2 = h2_mul(q.x, acc.zz, h);
    const FqU S2 = h2_mul(q.y, acc.zzz, h);
    const FqU Pp = fqu_sub<64>(U2, acc.x);
    const FqU R = fqu_sub<64>(S2, acc.y);
    if (h2_is_zero_mod(Pp)) {
        if (h2_is_zero_mod(R)) acc = h2_dbl_affine(q, h);
        else acc = h2_inf();
        return;
    }
    const FqU PP = h2_sqr(Pp, h);
    const FqU PPP = h2_mul(Pp, PP, h);
    const FqU Q = h2_mul(acc.x, PP, h);
    const FqU X3 = fqu_sub<32>(h2_sqr(R, h), fqu_add(PPP, fqu_dbl(Q)));
    acc.y = fqu_sub<32>(h2_mul(R, fqu_sub<64>(Q, X3), h), h2_mul(acc.y, PPP, h));
    acc.x = X3;
    acc.zz = h2_mul(acc.zz, PP, h);
    acc.zzz = h2_mul(acc.zzz, PPP, h);
}

}  // namespace zk
