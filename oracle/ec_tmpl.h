/* TEST INFRASTRUCTURE (oracle) — short-Weierstrass (a = 0) curve template, included for G1 (over Fq)
 * and G2 (over Fq2).
 *
 * CPU restatement of ark-ec 0.4.2 `short_weierstrass::{Affine, Projective}` semantics
 * (ark-ec `src/models/short_weierstrass/{affine,group}.rs`; crate not vendored — SURVEY.md §8c):
 * Jacobian coordinates (X/Z^2, Y/Z^3), infinity <=> Z == 0, affine carries an explicit infinity flag.
 * Formulas: dbl-2009-l, madd-2007-bl, add-2007-bl (EFD), with the exceptional cases handled.
 *
 * Parameters: EC (prefix g1/g2), EF (field prefix fq/fq2).
 */
#define ECCAT_(a, b) a##_##b
#define ECCAT(a, b) ECCAT_(a, b)
#define EFT ECCAT(EF, t)
#define EFF(name) ECCAT(EF, name)
#define ECF(name) ECCAT(EC, name)
#define ECAFF ECCAT(EC, aff_t)
#define ECJAC ECCAT(EC, jac_t)

typedef struct { EFT x, y; int inf; } ECAFF;
typedef struct { EFT x, y, z; } ECJAC;

static inline void ECF(jac_set_inf)(ECJAC *p) { EFF(one)(&p->x); EFF(one)(&p->y); EFF(zero)(&p->z); }
static inline int ECF(jac_is_inf)(const ECJAC *p) { return EFF(is_zero)(&p->z); }
static inline void ECF(jac_from_aff)(ECJAC *r, const ECAFF *a) {
    if (a->inf) { ECF(jac_set_inf)(r); return; }
    r->x = a->x; r->y = a->y; EFF(one)(&r->z);
}
static inline void ECF(jac_neg)(ECJAC *r, const ECJAC *a) { r->x = a->x; EFF(neg)(&r->y, &a->y); r->z = a->z; }
static inline void ECF(aff_neg)(ECAFF *r, const ECAFF *a) { r->x = a->x; EFF(neg)(&r->y, &a->y); r->inf = a->inf; }

static void ECF(jac_dbl)(ECJAC *r, const ECJAC *p) {
    if (ECF(jac_is_inf)(p)) { *r = *p; return; }
    EFT A, B, C, D, E, F, t, z3;
    EFF(sqr)(&A, &p->x);
    EFF(sqr)(&B, &p->y);
    EFF(sqr)(&C, &B);
    EFF(add)(&t, &p->x, &B); EFF(sqr)(&t, &t); EFF(sub)(&t, &t, &A); EFF(sub)(&t, &t, &C);
    EFF(dbl)(&D, &t);
    EFF(dbl)(&E, &A); EFF(add)(&E, &E, &A);
    EFF(sqr)(&F, &E);
    EFF(mul)(&z3, &p->y, &p->z); EFF(dbl)(&z3, &z3);
    EFF(sub)(&r->x, &F, &D); EFF(sub)(&r->x, &r->x, &D);
    EFF(sub)(&t, &D, &r->x); EFF(mul)(&t, &E, &t);
    EFF(dbl)(&C, &C); EFF(dbl)(&C, &C); EFF(dbl)(&C, &C);
    EFF(sub)(&r->y, &t, &C);
    r->z = z3;
}

/* r = p + q (q affine) */
static void ECF(jac_add_mixed)(ECJAC *r, const ECJAC *p, const ECAFF *q) {
    if (q->inf) { *r = *p; return; }
    if (ECF(jac_is_inf)(p)) { ECF(jac_from_aff)(r, q); return; }
    EFT Z1Z1, U2, S2, H, HH, I, J, rr, V, t;
    EFF(sqr)(&Z1Z1, &p->z);
    EFF(mul)(&U2, &q->x, &Z1Z1);
    EFF(mul)(&S2, &q->y, &p->z); EFF(mul)(&S2, &S2, &Z1Z1);
    EFF(sub)(&H, &U2, &p->x);
    EFF(sub)(&rr, &S2, &p->y);
    if (EFF(is_zero)(&H)) {
        if (EFF(is_zero)(&rr)) { ECJAC qq; ECF(jac_from_aff)(&qq, q); ECF(jac_dbl)(r, &qq); }
        else ECF(jac_set_inf)(r);
        return;
    }
    EFF(dbl)(&rr, &rr);
    EFF(sqr)(&HH, &H);
    EFF(dbl)(&I, &HH); EFF(dbl)(&I, &I);
    EFF(mul)(&J, &H, &I);
    EFF(mul)(&V, &p->x, &I);
    ECJAC o;
    EFF(sqr)(&o.x, &rr); EFF(sub)(&o.x, &o.x, &J); EFF(sub)(&o.x, &o.x, &V); EFF(sub)(&o.x, &o.x, &V);
    EFF(sub)(&t, &V, &o.x); EFF(mul)(&t, &rr, &t);
    EFF(mul)(&J, &p->y, &J); EFF(dbl)(&J, &J);
    EFF(sub)(&o.y, &t, &J);
    EFF(add)(&o.z, &p->z, &H); EFF(sqr)(&o.z, &o.z); EFF(sub)(&o.z, &o.z, &Z1Z1); EFF(sub)(&o.z, &o.z, &HH);
    *r = o;
}

static void ECF(jac_add)(ECJAC *r, const ECJAC *p, const ECJAC *q) {
    if (ECF(jac_is_inf)(p)) { *r = *q; return; }
    if (ECF(jac_is_inf)(q)) { *r = *p; return; }
    EFT Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t;
    EFF(sqr)(&Z1Z1, &p->z);
    EFF(sqr)(&Z2Z2, &q->z);
    EFF(mul)(&U1, &p->x, &Z2Z2);
    EFF(mul)(&U2, &q->x, &Z1Z1);
    EFF(mul)(&S1, &p->y, &q->z); EFF(mul)(&S1, &S1, &Z2Z2);
    EFF(mul)(&S2, &q->y, &p->z); EFF(mul)(&S2, &S2, &Z1Z1);
    EFF(sub)(&H, &U2, &U1);
    EFF(sub)(&rr, &S2, &S1);
    if (EFF(is_zero)(&H)) {
        if (EFF(is_zero)(&rr)) ECF(jac_dbl)(r, p);
        else ECF(jac_set_inf)(r);
        return;
    }
    EFF(dbl)(&rr, &rr);
    EFF(dbl)(&I, &H); EFF(sqr)(&I, &I);
    EFF(mul)(&J, &H, &I);
    EFF(mul)(&V, &U1, &I);
    ECJAC o;
    EFF(sqr)(&o.x, &rr); EFF(sub)(&o.x, &o.x, &J); EFF(sub)(&o.x, &o.x, &V); EFF(sub)(&o.x, &o.x, &V);
    EFF(sub)(&t, &V, &o.x); EFF(mul)(&t, &rr, &t);
    EFF(mul)(&J, &S1, &J); EFF(dbl)(&J, &J);
    EFF(sub)(&o.y, &t, &J);
    EFF(add)(&o.z, &p->z, &q->z); EFF(sqr)(&o.z, &o.z); EFF(sub)(&o.z, &o.z, &Z1Z1); EFF(sub)(&o.z, &o.z, &Z2Z2);
    EFF(mul)(&o.z, &o.z, &H);
    *r = o;
}

static void ECF(jac_to_aff)(ECAFF *r, const ECJAC *p) {
    if (ECF(jac_is_inf)(p)) { EFF(zero)(&r->x); EFF(zero)(&r->y); r->inf = 1; return; }
    EFT zi, zi2, zi3;
    EFF(inv)(&zi, &p->z);
    EFF(sqr)(&zi2, &zi);
    EFF(mul)(&zi3, &zi2, &zi);
    EFF(mul)(&r->x, &p->x, &zi2);
    EFF(mul)(&r->y, &p->y, &zi3);
    r->inf = 0;
}

/* r = [k]p, k = 4 canonical limbs (255-bit), plain double-and-add (MSB first). */
static void ECF(jac_mul)(ECJAC *r, const ECJAC *p, const u64 k[4]) {
    ECJAC acc;
    ECF(jac_set_inf)(&acc);
    for (int i = 255; i >= 0; i--) {
        ECF(jac_dbl)(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) ECF(jac_add)(&acc, &acc, p);
    }
    *r = acc;
}

/* Batch normalisation with one inversion (Montgomery's trick). */
static void ECF(batch_to_aff)(ECAFF *out, const ECJAC *in, size_t n) {
    if (n == 0) return;
    EFT *pre = (EFT *)malloc(sizeof(EFT) * n);
    EFT acc;
    EFF(one)(&acc);
    for (size_t i = 0; i < n; i++) {
        pre[i] = acc;
        if (!ECF(jac_is_inf)(&in[i])) EFF(mul)(&acc, &acc, &in[i].z);
    }
    EFT inv;
    EFF(inv)(&inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (ECF(jac_is_inf)(&in[i])) { EFF(zero)(&out[i].x); EFF(zero)(&out[i].y); out[i].inf = 1; continue; }
        EFT zi, zi2, zi3;
        EFF(mul)(&zi, &inv, &pre[i]);
        EFF(mul)(&inv, &inv, &in[i].z);
        EFF(sqr)(&zi2, &zi);
        EFF(mul)(&zi3, &zi2, &zi);
        EFF(mul)(&out[i].x, &in[i].x, &zi2);
        EFF(mul)(&out[i].y, &in[i].y, &zi3);
        out[i].inf = 0;
    }
    free(pre);
}

/* Pippenger MSM following ark-ec 0.4.2 `VariableBaseMSM::msm_bigint` (signed-digit variant;
 * ark-ec src/scalar_mul/variable_base/mod.rs — SURVEY.md A.5):
 *   c = 3 if n < 32 else ln_without_floats(n) + 2,  ln_without_floats(n) = log2(n) * 69 / 100
 *   signed radix-2^c digits, one bucket array of 2^(c-1) per window, running-sum reduction,
 *   Horner over windows from the top.  One OpenMP task per window (= ark `parallel` feature).
 * scalars: n x 4 canonical limbs. */
static void ECF(msm)(ECJAC *out, const ECAFF *bases, const u64 *scalars, size_t n) {
    ECF(jac_set_inf)(out);
    if (n == 0) return;
    unsigned c;
    if (n < 32) c = 3;
    else {
        unsigned lg = 0;
        while (((size_t)1 << lg) < n) lg++;          /* ceil(log2 n) == ark_std::log2 */
        c = lg * 69 / 100 + 2;
    }
    const unsigned num_bits = 255;
    const unsigned nwin = (num_bits + c - 1) / c;    /* div_ceil(num_bits, c) */
    /* signed digits: digit in [-2^(c-1), 2^(c-1)) with carry into the next window */
    int32_t *digits = (int32_t *)malloc(sizeof(int32_t) * n * nwin);
    for (size_t i = 0; i < n; i++) {
        const u64 *s = scalars + 4 * i;
        u64 carry = 0;
        for (unsigned w = 0; w < nwin; w++) {
            unsigned bit = w * c;
            unsigned limb = bit / 64, off = bit % 64;
            u64 v = s[limb] >> off;
            if (off + c > 64 && limb + 1 < 4) v |= s[limb + 1] << (64 - off);
            v &= (((u64)1 << c) - 1);
            v += carry;
            carry = 0;
            int64_t d = (int64_t)v;
            if (w != nwin - 1 && v >= ((u64)1 << (c - 1))) { d -= (int64_t)((u64)1 << c); carry = 1; }
            digits[i * nwin + w] = (int32_t)d;
        }
    }
    ECJAC *wsum = (ECJAC *)malloc(sizeof(ECJAC) * nwin);
    const size_t nb = (size_t)1 << (c - 1);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (unsigned w = 0; w < nwin; w++) {
        /* the top window may hold an unsigned digit up to 2^c - 1 + carry: size its bucket array for that */
        size_t nbw = (w == nwin - 1) ? ((size_t)1 << c) + 1 : nb;
        ECJAC *buckets = (ECJAC *)malloc(sizeof(ECJAC) * nbw);
        for (size_t b = 0; b < nbw; b++) ECF(jac_set_inf)(&buckets[b]);
        for (size_t i = 0; i < n; i++) {
            int32_t d = digits[i * nwin + w];
            if (d > 0) ECF(jac_add_mixed)(&buckets[d - 1], &buckets[d - 1], &bases[i]);
            else if (d < 0) { ECAFF nq; ECF(aff_neg)(&nq, &bases[i]); ECF(jac_add_mixed)(&buckets[-d - 1], &buckets[-d - 1], &nq); }
        }
        ECJAC run, acc;
        ECF(jac_set_inf)(&run); ECF(jac_set_inf)(&acc);
        for (size_t b = nbw; b-- > 0;) {
            ECF(jac_add)(&run, &run, &buckets[b]);
            ECF(jac_add)(&acc, &acc, &run);
        }
        wsum[w] = acc;
        free(buckets);
    }
    ECJAC total = wsum[nwin - 1];
    for (int w = (int)nwin - 2; w >= 0; w--) {
        for (unsigned k = 0; k < c; k++) ECF(jac_dbl)(&total, &total);
        ECF(jac_add)(&total, &total, &wsum[w]);
    }
    *out = total;
    free(wsum);
    free(digits);
}

/* Fixed-base batch multiplication [k_i]g for the known-trapdoor setup (ark-ec FixedBase::msm
 * semantics: a per-window table of multiples of g, then one mixed add per window). */
static void ECF(fixed_base_batch)(ECAFF *out, const ECJAC *g, const u64 *scalars, size_t n) {
    const unsigned wbits = 8, nwin = 32, tsz = 255;     /* 32 windows x 255 non-zero multiples */
    ECJAC *tj = (ECJAC *)malloc(sizeof(ECJAC) * nwin * tsz);
    ECJAC base = *g;
    for (unsigned w = 0; w < nwin; w++) {
        ECJAC acc = base;
        for (unsigned k = 0; k < tsz; k++) {
            tj[w * tsz + k] = acc;
            ECF(jac_add)(&acc, &acc, &base);
        }
        base = acc;                                      /* = 2^8 * previous base */
    }
    ECAFF *ta = (ECAFF *)malloc(sizeof(ECAFF) * nwin * tsz);
    ECF(batch_to_aff)(ta, tj, (size_t)nwin * tsz);
    free(tj);
    const size_t chunk = 4096;
    ECJAC *tmp = (ECJAC *)malloc(sizeof(ECJAC) * chunk);
    for (size_t s0 = 0; s0 < n; s0 += chunk) {
        size_t m = n - s0 < chunk ? n - s0 : chunk;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
        for (size_t i = 0; i < m; i++) {
            const u64 *k = scalars + 4 * (s0 + i);
            ECJAC acc;
            ECF(jac_set_inf)(&acc);
            for (unsigned w = 0; w < nwin; w++) {
                unsigned d = (unsigned)((k[(w * wbits) / 64] >> ((w * wbits) % 64)) & 0xff);
                if (d) ECF(jac_add_mixed)(&acc, &acc, &ta[w * tsz + d - 1]);
            }
            tmp[i] = acc;
        }
        ECF(batch_to_aff)(out + s0, tmp, m);
    }
    free(tmp);
    free(ta);
}

#undef EFT
#undef EFF
#undef ECF
#undef ECAFF
#undef ECJAC
#undef ECCAT
#undef ECCAT_
