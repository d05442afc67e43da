/* =====================================================================================
 * TEST INFRASTRUCTURE — CPU ORACLE.  Not part of the product path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this
 * library (oracle/build/libg16oracle.so).  The product (libzkg16.so) never links or calls it.
 *
 * What it is: a from-scratch C restatement of the upstream arithmetic that the reference's
 * `Groth16::<Bls12_381>::prove` call executes (call sites:
 *   /root/reference/src/arkworks/backend/matrix_proof.rs:139-140,
 *   /root/reference/src/arkworks/backend/fibbonaci_handler.rs:110,
 *   /root/reference/src/arkworks/backend/prime_snark.rs:119).
 * The arithmetic itself lives in crates that are NOT vendored in /root/reference and cannot be
 * fetched or built here (no cargo/rustc, no network):
 *   ark-groth16 ^0.4.0 (prover.rs, r1cs_to_qap.rs, generator.rs), ark-poly ^0.4.2 (radix-2 domain),
 *   ark-ec ^0.4.2 (VariableBaseMSM, FixedBase, short_weierstrass), ark-ff ^0.4 (Montgomery Fp),
 *   ark-bls12-381 ^0.4.0 (constants).
 * The published algorithms are restated from SURVEY.md Appendix A.
 *
 * PARITY PINNING: the reference holds no golden vectors / KATs for this path, and
 * arkworks cannot run here, so byte-level parity with arkworks is **unpinned**.  What pins this
 * oracle: the tests/golden JSON fixtures produced by tests/golden/gen_golden.py from an independent
 * pure-Python big-integer implementation (tests/golden/pyref.py): field/curve KATs, O(N^2) DFTs,
 * naive MSMs, and whole Groth16 proofs computed in the exponent with a known trapdoor.
 * Since Fr/Fq values and affine points are canonical, any mathematically correct prover is
 * bit-identical to arkworks on the same (pk, r, s, R1CS, z).
 * ===================================================================================== */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ constants */
static const u64 FR_MOD_[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static const u64 FR_R_[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
static const u64 FR_R2_[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
#define FR_INV_ 0xfffffffeffffffffULL
static const u64 FQ_MOD_[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const u64 FQ_R_[6] = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};
static const u64 FQ_R2_[6] = {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL, 0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};
#define FQ_INV_ 0x89f3fffcfffcfffdULL
static const u64 FR_ROOT32_MONT[4] = {0xb9b58d8c5f0e466aULL, 0x5b1b4c801819d7ecULL, 0x0af53ae352a31e64ULL, 0x5bf3adda19e9b27bULL};
static const u64 FR_GEN_MONT[4] = {0x0000000efffffff1ULL, 0x17e363d300189c0fULL, 0xff9c57876f8457b0ULL, 0x351332208fc5a8c4ULL};

/* ------------------------------------------------------------------ Fr, Fq */
#define FP fr
#define FP_NL 4
#define FP_MOD FR_MOD_
#define FP_INV FR_INV_
#define FP_R FR_R_
#define FP_R2 FR_R2_
#include "fp_tmpl.h"
#undef FP
#undef FP_NL
#undef FP_MOD
#undef FP_INV
#undef FP_R
#undef FP_R2

#define FP fq
#define FP_NL 6
#define FP_MOD FQ_MOD_
#define FP_INV FQ_INV_
#define FP_R FQ_R_
#define FP_R2 FQ_R2_
#include "fp_tmpl.h"
#undef FP
#undef FP_NL
#undef FP_MOD
#undef FP_INV
#undef FP_R
#undef FP_R2

/* ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2+1)
 * (ark-bls12-381 src/fields/fq2.rs: NONRESIDUE = -1) */
typedef struct { fq_t c0, c1; } fq2_t;
static inline int fq2_is_zero(const fq2_t *a) { return fq_is_zero(&a->c0) && fq_is_zero(&a->c1); }
static inline int fq2_eq(const fq2_t *a, const fq2_t *b) { return fq_eq(&a->c0, &b->c0) && fq_eq(&a->c1, &b->c1); }
static inline void fq2_zero(fq2_t *a) { fq_zero(&a->c0); fq_zero(&a->c1); }
static inline void fq2_one(fq2_t *a) { fq_one(&a->c0); fq_zero(&a->c1); }
static inline void fq2_add(fq2_t *r, const fq2_t *a, const fq2_t *b) { fq_add(&r->c0, &a->c0, &b->c0); fq_add(&r->c1, &a->c1, &b->c1); }
static inline void fq2_sub(fq2_t *r, const fq2_t *a, const fq2_t *b) { fq_sub(&r->c0, &a->c0, &b->c0); fq_sub(&r->c1, &a->c1, &b->c1); }
static inline void fq2_neg(fq2_t *r, const fq2_t *a) { fq_neg(&r->c0, &a->c0); fq_neg(&r->c1, &a->c1); }
static inline void fq2_dbl(fq2_t *r, const fq2_t *a) { fq2_add(r, a, a); }
static inline void fq2_mul(fq2_t *r, const fq2_t *a, const fq2_t *b) {
    fq_t v0, v1, s, t, o0, o1;
    fq_mul(&v0, &a->c0, &b->c0);
    fq_mul(&v1, &a->c1, &b->c1);
    fq_add(&s, &a->c0, &a->c1);
    fq_add(&t, &b->c0, &b->c1);
    fq_mul(&o1, &s, &t);
    fq_sub(&o1, &o1, &v0);
    fq_sub(&o1, &o1, &v1);
    fq_sub(&o0, &v0, &v1);
    r->c0 = o0; r->c1 = o1;
}
static inline void fq2_sqr(fq2_t *r, const fq2_t *a) {
    fq_t s, d, p, o0;
    fq_add(&s, &a->c0, &a->c1);
    fq_sub(&d, &a->c0, &a->c1);
    fq_mul(&p, &a->c0, &a->c1);
    fq_mul(&o0, &s, &d);
    r->c0 = o0;
    fq_dbl(&r->c1, &p);
}
static inline void fq2_inv(fq2_t *r, const fq2_t *a) {
    fq_t n0, n1, n;
    fq_sqr(&n0, &a->c0);
    fq_sqr(&n1, &a->c1);
    fq_add(&n, &n0, &n1);
    fq_inv(&n, &n);
    fq_mul(&r->c0, &a->c0, &n);
    fq_mul(&n1, &a->c1, &n);
    fq_neg(&r->c1, &n1);
}

/* ------------------------------------------------------------------ G1, G2 */
#define EC g1
#define EF fq
#include "ec_tmpl.h"
#undef EC
#undef EF
#define EC g2
#define EF fq2
#include "ec_tmpl.h"
#undef EC
#undef EF

/* ABI <-> struct converters.  ABI layout (include/zkg16.h): G1 affine = 12 u64 (x, y), G2 affine =
 * 24 u64 (x.c0, x.c1, y.c0, y.c1), Montgomery limbs, infinity in a separate byte array. */
static inline void g1_load(g1_aff_t *p, const u64 *l, int inf) {
    memcpy(p->x.l, l, 48); memcpy(p->y.l, l + 6, 48); p->inf = inf;
    if (inf) { fq_zero(&p->x); fq_zero(&p->y); }
}
static inline void g1_store(u64 *l, uint8_t *inf, const g1_aff_t *p) {
    memcpy(l, p->x.l, 48); memcpy(l + 6, p->y.l, 48); if (inf) *inf = (uint8_t)p->inf;
}
static inline void g2_load(g2_aff_t *p, const u64 *l, int inf) {
    memcpy(p->x.c0.l, l, 48); memcpy(p->x.c1.l, l + 6, 48);
    memcpy(p->y.c0.l, l + 12, 48); memcpy(p->y.c1.l, l + 18, 48); p->inf = inf;
    if (inf) { fq2_zero(&p->x); fq2_zero(&p->y); }
}
static inline void g2_store(u64 *l, uint8_t *inf, const g2_aff_t *p) {
    memcpy(l, p->x.c0.l, 48); memcpy(l + 6, p->x.c1.l, 48);
    memcpy(l + 12, p->y.c0.l, 48); memcpy(l + 18, p->y.c1.l, 48); if (inf) *inf = (uint8_t)p->inf;
}

/* ==================================================================================== exported API */
#define ORC_API __attribute__((visibility("default")))

ORC_API const char *orc_banner(void) { return "g16 oracle (TEST INFRASTRUCTURE; parity vs arkworks unpinned, pinned by tests/golden)"; }
#ifdef _OPENMP
#include <omp.h>
#endif
ORC_API int orc_omp_enabled(void) {
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}
/* number of worker threads for the OpenMP regions (one task per MSM window, NTT stages, SpMV rows):
 * 1 reproduces the single-threaded arkworks build the reference's plots are consistent with. */
ORC_API int orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ---- field KAT helpers (Montgomery in, Montgomery out unless named otherwise) */
ORC_API void orc_fr_add(const u64 *a, const u64 *b, u64 *o) { fr_add((fr_t *)o, (const fr_t *)a, (const fr_t *)b); }
ORC_API void orc_fr_sub(const u64 *a, const u64 *b, u64 *o) { fr_sub((fr_t *)o, (const fr_t *)a, (const fr_t *)b); }
ORC_API void orc_fr_mul(const u64 *a, const u64 *b, u64 *o) { fr_mul((fr_t *)o, (const fr_t *)a, (const fr_t *)b); }
ORC_API void orc_fr_inv(const u64 *a, u64 *o) { fr_inv((fr_t *)o, (const fr_t *)a); }
ORC_API void orc_fr_from_canonical(const u64 *a, u64 *o, size_t n) { for (size_t i = 0; i < n; i++) fr_from_canonical((fr_t *)(o + 4 * i), a + 4 * i); }
ORC_API void orc_fr_to_canonical(const u64 *a, u64 *o, size_t n) { for (size_t i = 0; i < n; i++) fr_to_canonical(o + 4 * i, (const fr_t *)(a + 4 * i)); }
ORC_API void orc_fq_add(const u64 *a, const u64 *b, u64 *o) { fq_add((fq_t *)o, (const fq_t *)a, (const fq_t *)b); }
ORC_API void orc_fq_sub(const u64 *a, const u64 *b, u64 *o) { fq_sub((fq_t *)o, (const fq_t *)a, (const fq_t *)b); }
ORC_API void orc_fq_mul(const u64 *a, const u64 *b, u64 *o) { fq_mul((fq_t *)o, (const fq_t *)a, (const fq_t *)b); }
ORC_API void orc_fq_inv(const u64 *a, u64 *o) { fq_inv((fq_t *)o, (const fq_t *)a); }
ORC_API void orc_fq_from_canonical(const u64 *a, u64 *o, size_t n) { for (size_t i = 0; i < n; i++) fq_from_canonical((fq_t *)(o + 6 * i), a + 6 * i); }
ORC_API void orc_fq_to_canonical(const u64 *a, u64 *o, size_t n) { for (size_t i = 0; i < n; i++) fq_to_canonical(o + 6 * i, (const fq_t *)(a + 6 * i)); }
ORC_API void orc_fq2_mul(const u64 *a, const u64 *b, u64 *o) { fq2_t r; fq2_mul(&r, (const fq2_t *)a, (const fq2_t *)b); memcpy(o, &r, 96); }
ORC_API void orc_fq2_sqr(const u64 *a, u64 *o) { fq2_t r; fq2_sqr(&r, (const fq2_t *)a); memcpy(o, &r, 96); }
ORC_API void orc_fq2_inv(const u64 *a, u64 *o) { fq2_t r; fq2_inv(&r, (const fq2_t *)a); memcpy(o, &r, 96); }

/* ---- curve helpers */
ORC_API void orc_g1_add(const u64 *p, int pinf, const u64 *q, int qinf, u64 *o, uint8_t *oinf) {
    g1_aff_t a, b, r; g1_jac_t j;
    g1_load(&a, p, pinf); g1_load(&b, q, qinf);
    g1_jac_from_aff(&j, &a); g1_jac_add_mixed(&j, &j, &b); g1_jac_to_aff(&r, &j); g1_store(o, oinf, &r);
}
ORC_API void orc_g1_mul(const u64 *p, int pinf, const u64 *k_canonical, u64 *o, uint8_t *oinf) {
    g1_aff_t a, r; g1_jac_t j;
    g1_load(&a, p, pinf); g1_jac_from_aff(&j, &a); g1_jac_mul(&j, &j, k_canonical); g1_jac_to_aff(&r, &j); g1_store(o, oinf, &r);
}
ORC_API void orc_g2_add(const u64 *p, int pinf, const u64 *q, int qinf, u64 *o, uint8_t *oinf) {
    g2_aff_t a, b, r; g2_jac_t j;
    g2_load(&a, p, pinf); g2_load(&b, q, qinf);
    g2_jac_from_aff(&j, &a); g2_jac_add_mixed(&j, &j, &b); g2_jac_to_aff(&r, &j); g2_store(o, oinf, &r);
}
ORC_API void orc_g2_mul(const u64 *p, int pinf, const u64 *k_canonical, u64 *o, uint8_t *oinf) {
    g2_aff_t a, r; g2_jac_t j;
    g2_load(&a, p, pinf); g2_jac_from_aff(&j, &a); g2_jac_mul(&j, &j, k_canonical); g2_jac_to_aff(&r, &j); g2_store(o, oinf, &r);
}
ORC_API int orc_g1_on_curve(const u64 *p) {
    g1_aff_t a; g1_load(&a, p, 0);
    fq_t l, r, b4; fq_sqr(&l, &a.y); fq_sqr(&r, &a.x); fq_mul(&r, &r, &a.x); fq_from_u64(&b4, 4); fq_add(&r, &r, &b4);
    return fq_eq(&l, &r);
}
ORC_API int orc_g2_on_curve(const u64 *p) {
    g2_aff_t a; g2_load(&a, p, 0);
    fq2_t l, r, b; fq2_sqr(&l, &a.y); fq2_sqr(&r, &a.x); fq2_mul(&r, &r, &a.x);
    fq_from_u64(&b.c0, 4); fq_from_u64(&b.c1, 4); fq2_add(&r, &r, &b);
    return fq2_eq(&l, &r);
}

/* ---- MSM (ark-ec VariableBaseMSM::msm_bigint) */
ORC_API int orc_msm_g1(const u64 *bases, const uint8_t *inf, const u64 *scalars_canonical, size_t n, u64 *out, uint8_t *oinf) {
    g1_aff_t *b = (g1_aff_t *)malloc(sizeof(g1_aff_t) * (n ? n : 1));
    if (!b) return 1;
    for (size_t i = 0; i < n; i++) g1_load(&b[i], bases + 12 * i, inf ? inf[i] : 0);
    g1_jac_t acc; g1_aff_t r;
    g1_msm(&acc, b, scalars_canonical, n);
    g1_jac_to_aff(&r, &acc); g1_store(out, oinf, &r);
    free(b);
    return 0;
}
ORC_API int orc_msm_g2(const u64 *bases, const uint8_t *inf, const u64 *scalars_canonical, size_t n, u64 *out, uint8_t *oinf) {
    g2_aff_t *b = (g2_aff_t *)malloc(sizeof(g2_aff_t) * (n ? n : 1));
    if (!b) return 1;
    for (size_t i = 0; i < n; i++) g2_load(&b[i], bases + 24 * i, inf ? inf[i] : 0);
    g2_jac_t acc; g2_aff_t r;
    g2_msm(&acc, b, scalars_canonical, n);
    g2_jac_to_aff(&r, &acc); g2_store(out, oinf, &r);
    free(b);
    return 0;
}

/* ---- fixed-base batch [k_i]g (ark-ec FixedBase::msm semantics; used by the known-trapdoor setup) */
ORC_API int orc_fixed_base_g1(const u64 *g, const u64 *scalars_canonical, size_t n, u64 *out, uint8_t *oinf) {
    g1_aff_t ga; g1_jac_t gj;
    g1_load(&ga, g, 0); g1_jac_from_aff(&gj, &ga);
    g1_aff_t *o = (g1_aff_t *)malloc(sizeof(g1_aff_t) * (n ? n : 1));
    if (!o) return 1;
    g1_fixed_base_batch(o, &gj, scalars_canonical, n);
    for (size_t i = 0; i < n; i++) g1_store(out + 12 * i, oinf ? oinf + i : NULL, &o[i]);
    free(o);
    return 0;
}
ORC_API int orc_fixed_base_g2(const u64 *g, const u64 *scalars_canonical, size_t n, u64 *out, uint8_t *oinf) {
    g2_aff_t ga; g2_jac_t gj;
    g2_load(&ga, g, 0); g2_jac_from_aff(&gj, &ga);
    g2_aff_t *o = (g2_aff_t *)malloc(sizeof(g2_aff_t) * (n ? n : 1));
    if (!o) return 1;
    g2_fixed_base_batch(o, &gj, scalars_canonical, n);
    for (size_t i = 0; i < n; i++) g2_store(out + 24 * i, oinf ? oinf + i : NULL, &o[i]);
    free(o);
    return 0;
}

/* ------------------------------------------------------------------ NTT
 * ark-poly 0.4.2 Radix2EvaluationDomain<Fr> semantics (src/domain/radix2/{mod,fft}.rs — SURVEY.md A.4):
 * natural order in / out; group_gen = ROOT_2^32 ^ (2^(32-log_n)); coset offset g = 7;
 * ifft multiplies by size_inv; coset fft = distribute_powers(g) then fft; coset ifft = ifft then
 * distribute_powers(g^-1). */
static void fr_root_of_unity(fr_t *w, unsigned log_n) {
    memcpy(w->l, FR_ROOT32_MONT, 32);
    for (unsigned i = log_n; i < 32; i++) fr_sqr(w, w);
}
static void bitrev_permute(fr_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        size_t r = 0;
        for (unsigned b = 0; b < log_n; b++) r |= ((i >> b) & 1) << (log_n - 1 - b);
        if (i < r) { fr_t t = a[i]; a[i] = a[r]; a[r] = t; }
    }
}
static void ntt_core(fr_t *a, unsigned log_n, const fr_t *root) {
    size_t n = (size_t)1 << log_n;
    if (n == 1) return;
    fr_t *tw = (fr_t *)malloc(sizeof(fr_t) * (n / 2));
    fr_one(&tw[0]);
    for (size_t i = 1; i < n / 2; i++) fr_mul(&tw[i], &tw[i - 1], root);
    bitrev_permute(a, log_n);
    for (unsigned s = 1; s <= log_n; s++) {
        size_t half = (size_t)1 << (s - 1), stride = n >> s;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (n >= 4096)
#endif
        for (size_t i = 0; i < n / 2; i++) {
            size_t k = (i / half) * (half * 2), j = i % half;
            fr_t t, u = a[k + j];
            fr_mul(&t, &tw[j * stride], &a[k + j + half]);
            fr_add(&a[k + j], &u, &t);
            fr_sub(&a[k + j + half], &u, &t);
        }
    }
    free(tw);
}
static void distribute_powers(fr_t *a, size_t n, const fr_t *g) {
    fr_t p; fr_one(&p);
    for (size_t i = 0; i < n; i++) { fr_mul(&a[i], &a[i], &p); fr_mul(&p, &p, g); }
}
static int ntt_dispatch(fr_t *a, unsigned log_n, int inverse, int coset) {
    if (log_n > 32) return 2;
    size_t n = (size_t)1 << log_n;
    fr_t w, g;
    fr_root_of_unity(&w, log_n);
    memcpy(g.l, FR_GEN_MONT, 32);
    if (!inverse) {
        if (coset) distribute_powers(a, n, &g);
        ntt_core(a, log_n, &w);
    } else {
        fr_t wi, ninv, nn;
        fr_inv(&wi, &w);
        ntt_core(a, log_n, &wi);
        fr_from_u64(&nn, (u64)n);
        fr_inv(&ninv, &nn);
        for (size_t i = 0; i < n; i++) fr_mul(&a[i], &a[i], &ninv);
        if (coset) { fr_t gi; fr_inv(&gi, &g); distribute_powers(a, n, &gi); }
    }
    return 0;
}
ORC_API int orc_ntt(u64 *data, unsigned log_n, int inverse, int coset) { return ntt_dispatch((fr_t *)data, log_n, inverse, coset); }

/* ------------------------------------------------------------------ R1CS -> QAP witness map
 * ark-groth16 0.4 `LibsnarkReduction::witness_map_from_matrices` (src/r1cs_to_qap.rs — SURVEY.md A.4). */
typedef struct { const u64 *row_ptr; const uint32_t *col; const fr_t *coeff; } csr_t;

static void csr_row_dot(fr_t *out, const csr_t *m, size_t row, const fr_t *z) {
    fr_t acc, t;
    fr_zero(&acc);
    for (u64 k = m->row_ptr[row]; k < m->row_ptr[row + 1]; k++) {
        fr_mul(&t, &m->coeff[k], &z[m->col[k]]);
        fr_add(&acc, &acc, &t);
    }
    *out = acc;
}
static unsigned log2_ceil(size_t n) { unsigned l = 0; while (((size_t)1 << l) < n) l++; return l; }

static int witness_map(const csr_t m[3], size_t num_inputs, size_t nc, const fr_t *z, fr_t *h, unsigned log_n) {
    size_t N = (size_t)1 << log_n;
    fr_t *a = (fr_t *)calloc(N, sizeof(fr_t)), *b = (fr_t *)calloc(N, sizeof(fr_t)), *c = (fr_t *)calloc(N, sizeof(fr_t));
    if (!a || !b || !c) return 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (size_t i = 0; i < nc; i++) { csr_row_dot(&a[i], &m[0], i, z); csr_row_dot(&b[i], &m[1], i, z); csr_row_dot(&c[i], &m[2], i, z); }
    for (size_t i = 0; i < num_inputs; i++) a[nc + i] = z[i];
    ntt_dispatch(a, log_n, 1, 0); ntt_dispatch(b, log_n, 1, 0);
    ntt_dispatch(a, log_n, 0, 1); ntt_dispatch(b, log_n, 0, 1);
    for (size_t i = 0; i < N; i++) fr_mul(&a[i], &a[i], &b[i]);
    ntt_dispatch(c, log_n, 1, 0); ntt_dispatch(c, log_n, 0, 1);
    /* Z(g) = g^N - 1 on the whole coset */
    fr_t g, zg, one, zinv;
    memcpy(g.l, FR_GEN_MONT, 32);
    zg = g;
    for (unsigned i = 0; i < log_n; i++) fr_sqr(&zg, &zg);
    fr_one(&one); fr_sub(&zg, &zg, &one); fr_inv(&zinv, &zg);
    for (size_t i = 0; i < N; i++) { fr_sub(&a[i], &a[i], &c[i]); fr_mul(&a[i], &a[i], &zinv); }
    ntt_dispatch(a, log_n, 1, 1);
    memcpy(h, a, sizeof(fr_t) * N);
    free(a); free(b); free(c);
    return 0;
}

ORC_API int orc_witness_map(const u64 *a_rp, const uint32_t *a_col, const u64 *a_cf,
                            const u64 *b_rp, const uint32_t *b_col, const u64 *b_cf,
                            const u64 *c_rp, const uint32_t *c_col, const u64 *c_cf,
                            size_t num_inputs, size_t num_constraints, const u64 *z, u64 *h_out, unsigned log_n) {
    csr_t m[3] = {{a_rp, a_col, (const fr_t *)a_cf}, {b_rp, b_col, (const fr_t *)b_cf}, {c_rp, c_col, (const fr_t *)c_cf}};
    if (log_n != log2_ceil(num_constraints + num_inputs)) return 2;
    return witness_map(m, num_inputs, num_constraints, (const fr_t *)z, (fr_t *)h_out, log_n);
}

/* ------------------------------------------------------------------ known-trapdoor setup: discrete logs
 * ark-groth16 0.4 `generate_parameters_with_qap` (src/generator.rs — SURVEY.md A.7) evaluated in the
 * exponent: returns the scalars k such that pk element = [k]g.  trap = tau, alpha, beta, gamma, delta
 * (5 x 4 limbs, Montgomery).  Outputs Montgomery Fr:
 *   a_log[num_vars] = u_k(tau), b_log[num_vars] = v_k(tau), l_log[num_vars-num_inputs],
 *   h_log[N-1], gabc_log[num_inputs]. */
ORC_API int orc_setup_logs(const u64 *a_rp, const uint32_t *a_col, const u64 *a_cf,
                           const u64 *b_rp, const uint32_t *b_col, const u64 *b_cf,
                           const u64 *c_rp, const uint32_t *c_col, const u64 *c_cf,
                           size_t num_inputs, size_t nc, size_t num_vars, const u64 *trap,
                           u64 *a_log, u64 *b_log, u64 *l_log, u64 *h_log, u64 *gabc_log) {
    csr_t m[3] = {{a_rp, a_col, (const fr_t *)a_cf}, {b_rp, b_col, (const fr_t *)b_cf}, {c_rp, c_col, (const fr_t *)c_cf}};
    const fr_t *tau = (const fr_t *)trap, *alpha = tau + 1, *beta = tau + 2, *gamma = tau + 3, *delta = tau + 4;
    unsigned log_n = log2_ceil(nc + num_inputs);
    size_t N = (size_t)1 << log_n;
    fr_t one, w, zt, ninv, nn;
    fr_one(&one);
    fr_root_of_unity(&w, log_n);
    zt = *tau;
    for (unsigned i = 0; i < log_n; i++) fr_sqr(&zt, &zt);
    fr_sub(&zt, &zt, &one);                                   /* Z(tau) = tau^N - 1 */
    fr_from_u64(&nn, (u64)N); fr_inv(&ninv, &nn);
    /* L_i(tau) = Z(tau)/N * w^i / (tau - w^i): batch inversion of (tau - w^i) */
    fr_t *L = (fr_t *)malloc(sizeof(fr_t) * N), *den = (fr_t *)malloc(sizeof(fr_t) * N), *pre = (fr_t *)malloc(sizeof(fr_t) * N);
    if (!L || !den || !pre) return 1;
    fr_t wi = one, acc = one;
    for (size_t i = 0; i < N; i++) {
        fr_sub(&den[i], tau, &wi);
        if (fr_is_zero(&den[i])) return 3;                    /* tau in the domain: not a valid trapdoor */
        L[i] = wi;
        pre[i] = acc;
        fr_mul(&acc, &acc, &den[i]);
        fr_mul(&wi, &wi, &w);
    }
    fr_t inv, scale;
    fr_inv(&inv, &acc);
    fr_mul(&scale, &zt, &ninv);
    for (size_t i = N; i-- > 0;) {
        fr_t di;
        fr_mul(&di, &inv, &pre[i]);
        fr_mul(&inv, &inv, &den[i]);
        fr_mul(&L[i], &L[i], &di);
        fr_mul(&L[i], &L[i], &scale);
    }
    free(den); free(pre);
    fr_t *u = (fr_t *)a_log, *v = (fr_t *)b_log;
    fr_t *wv = (fr_t *)calloc(num_vars, sizeof(fr_t));
    memset(u, 0, sizeof(fr_t) * num_vars);
    memset(v, 0, sizeof(fr_t) * num_vars);
    fr_t t;
    for (size_t i = 0; i < nc; i++) {
        for (u64 k = m[0].row_ptr[i]; k < m[0].row_ptr[i + 1]; k++) { fr_mul(&t, &m[0].coeff[k], &L[i]); fr_add(&u[m[0].col[k]], &u[m[0].col[k]], &t); }
        for (u64 k = m[1].row_ptr[i]; k < m[1].row_ptr[i + 1]; k++) { fr_mul(&t, &m[1].coeff[k], &L[i]); fr_add(&v[m[1].col[k]], &v[m[1].col[k]], &t); }
        for (u64 k = m[2].row_ptr[i]; k < m[2].row_ptr[i + 1]; k++) { fr_mul(&t, &m[2].coeff[k], &L[i]); fr_add(&wv[m[2].col[k]], &wv[m[2].col[k]], &t); }
    }
    for (size_t k = 0; k < num_inputs; k++) fr_add(&u[k], &u[k], &L[nc + k]);
    fr_t dinv, ginv;
    fr_inv(&dinv, delta); fr_inv(&ginv, gamma);
    for (size_t k = 0; k < num_vars; k++) {
        fr_t x, y;
        fr_mul(&x, beta, &u[k]); fr_mul(&y, alpha, &v[k]); fr_add(&x, &x, &y); fr_add(&x, &x, &wv[k]);
        if (k < num_inputs) fr_mul((fr_t *)gabc_log + k, &x, &ginv);
        else fr_mul((fr_t *)l_log + (k - num_inputs), &x, &dinv);
    }
    fr_t p;
    fr_mul(&p, &zt, &dinv);
    for (size_t i = 0; i + 1 < N; i++) { ((fr_t *)h_log)[i] = p; fr_mul(&p, &p, tau); }
    free(L); free(wv);
    return 0;
}

/* ------------------------------------------------------------------ prove
 * ark-groth16 0.4 `create_proof_with_assignment` (src/prover.rs — SURVEY.md A.3, A.6), with r, s
 * given (the reference draws them from its rng inside `Groth16::prove`, matrix_proof.rs:139-140).
 * pk layout = include/zkg16.h.  proof_out = A(12) || B(24) || C(12) affine Montgomery; inf_out[3]. */
typedef struct {
    const u64 *a_query, *b_g1_query, *b_g2_query, *h_query, *l_query;
    const uint8_t *a_inf, *b1_inf, *b2_inf, *h_inf, *l_inf;     /* nullable */
    size_t n_a, n_b1, n_b2, n_h, n_l;
    const u64 *alpha_g1, *beta_g1, *beta_g2, *delta_g1, *delta_g2;
} orc_pk_t;

static void msm_g1_raw(g1_jac_t *out, const u64 *bases, const uint8_t *inf, const u64 *sc, size_t n) {
    g1_aff_t *b = (g1_aff_t *)malloc(sizeof(g1_aff_t) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) g1_load(&b[i], bases + 12 * i, inf ? inf[i] : 0);
    g1_msm(out, b, sc, n);
    free(b);
}
static void msm_g2_raw(g2_jac_t *out, const u64 *bases, const uint8_t *inf, const u64 *sc, size_t n) {
    g2_aff_t *b = (g2_aff_t *)malloc(sizeof(g2_aff_t) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) g2_load(&b[i], bases + 24 * i, inf ? inf[i] : 0);
    g2_msm(out, b, sc, n);
    free(b);
}

ORC_API int orc_prove(const orc_pk_t *pk, const u64 *r_mont, const u64 *s_mont,
                      const u64 *a_rp, const uint32_t *a_col, const u64 *a_cf,
                      const u64 *b_rp, const uint32_t *b_col, const u64 *b_cf,
                      const u64 *c_rp, const uint32_t *c_col, const u64 *c_cf,
                      size_t num_inputs, size_t nc, const u64 *z, size_t n_assign,
                      u64 *proof_out, uint8_t *inf_out, double *stage_ms /* nullable, 4 doubles */) {
    (void)stage_ms;
    csr_t m[3] = {{a_rp, a_col, (const fr_t *)a_cf}, {b_rp, b_col, (const fr_t *)b_cf}, {c_rp, c_col, (const fr_t *)c_cf}};
    unsigned log_n = log2_ceil(nc + num_inputs);
    size_t N = (size_t)1 << log_n;
    size_t w = n_assign - num_inputs;
    if (pk->n_a != n_assign || pk->n_b1 != n_assign || pk->n_b2 != n_assign || pk->n_l != w || pk->n_h != N - 1) return 2;
    fr_t *h = (fr_t *)malloc(sizeof(fr_t) * N);
    if (!h) return 1;
    int rc = witness_map(m, num_inputs, nc, (const fr_t *)z, h, log_n);
    if (rc) { free(h); return rc; }
    u64 *hc = (u64 *)malloc(32 * N), *zc = (u64 *)malloc(32 * n_assign);
    for (size_t i = 0; i < N; i++) fr_to_canonical(hc + 4 * i, &h[i]);
    for (size_t i = 0; i < n_assign; i++) fr_to_canonical(zc + 4 * i, (const fr_t *)z + i);
    free(h);
    u64 rc4[4], sc4[4], rs4[4];
    fr_t rs;
    fr_to_canonical(rc4, (const fr_t *)r_mont);
    fr_to_canonical(sc4, (const fr_t *)s_mont);
    fr_mul(&rs, (const fr_t *)r_mont, (const fr_t *)s_mont);
    fr_to_canonical(rs4, &rs);

    g1_jac_t h_acc, l_acc, t1;
    msm_g1_raw(&h_acc, pk->h_query, pk->h_inf, hc, N - 1);
    msm_g1_raw(&l_acc, pk->l_query, pk->l_inf, zc + 4 * num_inputs, w);

    g1_aff_t alpha, beta1, delta1, q0;
    g1_load(&alpha, pk->alpha_g1, 0); g1_load(&beta1, pk->beta_g1, 0); g1_load(&delta1, pk->delta_g1, 0);
    g1_jac_t dj, rdelta, sdelta;
    g1_jac_from_aff(&dj, &delta1);
    g1_jac_mul(&rdelta, &dj, rc4);
    g1_jac_mul(&sdelta, &dj, sc4);

    /* calculate_coeff(initial, query, vk_param, assignment) = initial + query[0] + msm(query[1..], assignment) + vk_param */
    g1_jac_t g_a, g1_b;
    msm_g1_raw(&t1, pk->a_query + 12, pk->a_inf ? pk->a_inf + 1 : NULL, zc + 4, n_assign - 1);
    g1_load(&q0, pk->a_query, pk->a_inf ? pk->a_inf[0] : 0);
    g_a = rdelta; g1_jac_add_mixed(&g_a, &g_a, &q0); g1_jac_add(&g_a, &g_a, &t1); g1_jac_add_mixed(&g_a, &g_a, &alpha);

    msm_g1_raw(&t1, pk->b_g1_query + 12, pk->b1_inf ? pk->b1_inf + 1 : NULL, zc + 4, n_assign - 1);
    g1_load(&q0, pk->b_g1_query, pk->b1_inf ? pk->b1_inf[0] : 0);
    g1_b = sdelta; g1_jac_add_mixed(&g1_b, &g1_b, &q0); g1_jac_add(&g1_b, &g1_b, &t1); g1_jac_add_mixed(&g1_b, &g1_b, &beta1);

    g2_aff_t beta2, delta2, q02;
    g2_load(&beta2, pk->beta_g2, 0); g2_load(&delta2, pk->delta_g2, 0);
    g2_jac_t d2j, sdelta2, t2, g2_b;
    g2_jac_from_aff(&d2j, &delta2);
    g2_jac_mul(&sdelta2, &d2j, sc4);
    msm_g2_raw(&t2, pk->b_g2_query + 24, pk->b2_inf ? pk->b2_inf + 1 : NULL, zc + 4, n_assign - 1);
    g2_load(&q02, pk->b_g2_query, pk->b2_inf ? pk->b2_inf[0] : 0);
    g2_b = sdelta2; g2_jac_add_mixed(&g2_b, &g2_b, &q02); g2_jac_add(&g2_b, &g2_b, &t2); g2_jac_add_mixed(&g2_b, &g2_b, &beta2);

    /* g_c = s*g_a + r*g1_b - (r*s)*delta + l_acc + h_acc */
    g1_jac_t g_c, x;
    g1_jac_mul(&g_c, &g_a, sc4);
    g1_jac_mul(&x, &g1_b, rc4); g1_jac_add(&g_c, &g_c, &x);
    g1_jac_mul(&x, &dj, rs4); g1_jac_neg(&x, &x); g1_jac_add(&g_c, &g_c, &x);
    g1_jac_add(&g_c, &g_c, &l_acc);
    g1_jac_add(&g_c, &g_c, &h_acc);

    g1_aff_t A, C; g2_aff_t B;
    g1_jac_to_aff(&A, &g_a); g2_jac_to_aff(&B, &g2_b); g1_jac_to_aff(&C, &g_c);
    g1_store(proof_out, inf_out, &A);
    g2_store(proof_out + 12, inf_out + 1, &B);
    g1_store(proof_out + 36, inf_out + 2, &C);
    free(hc); free(zc);
    return 0;
}
