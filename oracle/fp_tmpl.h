/* TEST INFRASTRUCTURE (oracle) — prime-field template, included once per field.
 *
 * CPU restatement of ark-ff 0.4 `Fp<MontBackend<_, N>, N>` semantics
 * (ark-ff `src/fields/models/fp/montgomery_backend.rs`; crate not vendored in
 * /root/reference — see SURVEY.md §8c): N little-endian u64 limbs, Montgomery form
 * with R = 2^(64N), CIOS multiplication, values always fully reduced (< p).
 * Reference call sites that reach this arithmetic: src/arkworks/backend/matrix_proof.rs:139-140.
 *
 * Parameters (all #defined by the includer):
 *   FP        type/function prefix (fr / fq)
 *   FP_NL     limb count
 *   FP_MOD    const u64[FP_NL] modulus
 *   FP_INV    -p^-1 mod 2^64
 *   FP_R      const u64[FP_NL]  R mod p   (Montgomery one)
 *   FP_R2     const u64[FP_NL]  R^2 mod p
 */
#define FPCAT_(a, b) a##_##b
#define FPCAT(a, b) FPCAT_(a, b)
#define FPT FPCAT(FP, t)
#define FPF(name) FPCAT(FP, name)

typedef struct { u64 l[FP_NL]; } FPT;

static inline int FPF(is_zero)(const FPT *a) {
    u64 acc = 0;
    for (int i = 0; i < FP_NL; i++) acc |= a->l[i];
    return acc == 0;
}
static inline int FPF(eq)(const FPT *a, const FPT *b) {
    u64 acc = 0;
    for (int i = 0; i < FP_NL; i++) acc |= a->l[i] ^ b->l[i];
    return acc == 0;
}
static inline void FPF(zero)(FPT *a) { memset(a, 0, sizeof *a); }
static inline void FPF(one)(FPT *a) { memcpy(a->l, FP_R, sizeof a->l); }

/* a >= p ? */
static inline int FPF(geq_mod)(const u64 *a) {
    for (int i = FP_NL - 1; i >= 0; i--) {
        if (a[i] > FP_MOD[i]) return 1;
        if (a[i] < FP_MOD[i]) return 0;
    }
    return 1;
}
static inline void FPF(sub_mod_raw)(u64 *a) {
    u64 borrow = 0;
    for (int i = 0; i < FP_NL; i++) {
        u128 d = (u128)a[i] - FP_MOD[i] - borrow;
        a[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
}
static inline void FPF(add)(FPT *r, const FPT *a, const FPT *b) {
    u64 carry = 0;
    for (int i = 0; i < FP_NL; i++) {
        u128 s = (u128)a->l[i] + b->l[i] + carry;
        r->l[i] = (u64)s;
        carry = (u64)(s >> 64);
    }
    /* both moduli leave the top bit(s) free, so carry is always 0 */
    if (carry || FPF(geq_mod)(r->l)) FPF(sub_mod_raw)(r->l);
}
static inline void FPF(sub)(FPT *r, const FPT *a, const FPT *b) {
    u64 borrow = 0;
    for (int i = 0; i < FP_NL; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        r->l[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
    if (borrow) {
        u64 carry = 0;
        for (int i = 0; i < FP_NL; i++) {
            u128 s = (u128)r->l[i] + FP_MOD[i] + carry;
            r->l[i] = (u64)s;
            carry = (u64)(s >> 64);
        }
    }
}
static inline void FPF(neg)(FPT *r, const FPT *a) {
    if (FPF(is_zero)(a)) { *r = *a; return; }
    u64 borrow = 0;
    for (int i = 0; i < FP_NL; i++) {
        u128 d = (u128)FP_MOD[i] - a->l[i] - borrow;
        r->l[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
}
static inline void FPF(dbl)(FPT *r, const FPT *a) { FPF(add)(r, a, a); }

/* Montgomery product a*b*R^-1 mod p, CIOS. */
static inline void FPF(mul)(FPT *r, const FPT *a, const FPT *b) {
    u64 t[FP_NL + 2];
    memset(t, 0, sizeof t);
    for (int i = 0; i < FP_NL; i++) {
        u128 carry = 0;
        for (int j = 0; j < FP_NL; j++) {
            u128 cur = (u128)a->l[j] * b->l[i] + t[j] + carry;
            t[j] = (u64)cur;
            carry = cur >> 64;
        }
        u128 cur = (u128)t[FP_NL] + carry;
        t[FP_NL] = (u64)cur;
        t[FP_NL + 1] = (u64)(cur >> 64);
        u64 m = t[0] * FP_INV;
        carry = ((u128)m * FP_MOD[0] + t[0]) >> 64;
        for (int j = 1; j < FP_NL; j++) {
            cur = (u128)m * FP_MOD[j] + t[j] + carry;
            t[j - 1] = (u64)cur;
            carry = cur >> 64;
        }
        cur = (u128)t[FP_NL] + carry;
        t[FP_NL - 1] = (u64)cur;
        t[FP_NL] = t[FP_NL + 1] + (u64)(cur >> 64);
    }
    if (t[FP_NL] || FPF(geq_mod)(t)) FPF(sub_mod_raw)(t);
    memcpy(r->l, t, sizeof r->l);
}
static inline void FPF(sqr)(FPT *r, const FPT *a) { FPF(mul)(r, a, a); }

/* canonical (plain residue limbs) <-> Montgomery */
static inline void FPF(from_canonical)(FPT *r, const u64 *canon) {
    FPT c, r2;
    memcpy(c.l, canon, sizeof c.l);
    memcpy(r2.l, FP_R2, sizeof r2.l);
    FPF(mul)(r, &c, &r2);
}
static inline void FPF(to_canonical)(u64 *canon, const FPT *a) {
    FPT one;
    memset(&one, 0, sizeof one);
    one.l[0] = 1;
    FPT o;
    FPF(mul)(&o, a, &one);
    memcpy(canon, o.l, sizeof o.l);
}
static inline void FPF(from_u64)(FPT *r, u64 v) {
    u64 c[FP_NL];
    memset(c, 0, sizeof c);
    c[0] = v;
    FPF(from_canonical)(r, c);
}
/* a^e, e given as nl canonical limbs */
static inline void FPF(pow)(FPT *r, const FPT *a, const u64 *e, int nl) {
    FPT acc, base = *a;
    FPF(one)(&acc);
    int started = 0;
    for (int i = nl * 64 - 1; i >= 0; i--) {
        if (started) FPF(sqr)(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) {
            if (started) FPF(mul)(&acc, &acc, &base);
            else { acc = base; started = 1; }
        }
    }
    *r = acc;
}
/* a^-1 = a^(p-2); inverse of 0 is 0 */
static inline void FPF(inv)(FPT *r, const FPT *a) {
    u64 e[FP_NL];
    memcpy(e, FP_MOD, sizeof e);
    e[0] -= 2; /* both moduli end in ...01 / ...ab: no borrow */
    FPF(pow)(r, a, e, FP_NL);
}

#undef FPT
#undef FPF
#undef FPCAT
#undef FPCAT_
