// TEST-ONLY: exposes the __host__ side of csrc/ff.cuh + csrc/ec.cuh (the exact templates the kernels
// instantiate) so the limb arithmetic and curve formulas can be checked against the golden vectors on
// a machine without a GPU.  Built by tests/test_ff_host.py with `hipcc --offload-host-only`.
#include "ff.cuh"
#include "ec.cuh"
#include <string.h>
using namespace zk;

template <class F> static F ld(const uint32_t *p) { F r; memcpy(&r, p, sizeof(F)); return r; }
template <class F> static void st(uint32_t *p, const F &v) { memcpy(p, &v, sizeof(F)); }

template <class F> static void field_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *o) {
    F x = ld<F>(a), y = ld<F>(b), r;
    switch (op) {
        case 0: r = f_add(x, y); break;
        case 1: r = f_sub(x, y); break;
        case 2: r = f_mul(x, y); break;
        case 3: r = f_sqr(x); break;
        case 4: r = f_inv(x); break;
        case 5: r = f_neg(x); break;
        default: r = x;
    }
    st(o, r);
}
template <class F> static void point_op(int op, const uint32_t *acc_in, const uint32_t *q, const uint32_t *k, int neg, uint32_t *out_xyzz, uint32_t *out_aff) {
    XYZZ<F> acc = ld<XYZZ<F>>(acc_in);
    switch (op) {
        case 0: xyzz_madd(acc, ld<Affine<F>>(q), neg != 0); break;
        case 1: xyzz_add(acc, ld<XYZZ<F>>(q)); break;
        case 2: acc = xyzz_dbl(acc); break;
        case 3: acc = xyzz_mul(acc, k); break;
        case 4: acc = XYZZ<F>::from_affine(ld<Affine<F>>(q)); break;
    }
    st(out_xyzz, acc);
    st(out_aff, xyzz_to_affine(acc));
}
extern "C" {
void ht_fr_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *o) { field_op<Fr>(op, a, b, o); }
void ht_fq_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *o) { field_op<Fq>(op, a, b, o); }
void ht_fq2_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *o) { field_op<Fq2>(op, a, b, o); }
void ht_fr_to_mont(const uint32_t *a, uint32_t *o) { st(o, fp_to_mont(ld<Fr>(a))); }
void ht_fr_from_mont(const uint32_t *a, uint32_t *o) { st(o, fp_from_mont(ld<Fr>(a))); }
void ht_g1_op(int op, const uint32_t *acc, const uint32_t *q, const uint32_t *k, int neg, uint32_t *ox, uint32_t *oa) { point_op<Fq>(op, acc, q, k, neg, ox, oa); }
void ht_g2_op(int op, const uint32_t *acc, const uint32_t *q, const uint32_t *k, int neg, uint32_t *ox, uint32_t *oa) { point_op<Fq2>(op, acc, q, k, neg, ox, oa); }
}
