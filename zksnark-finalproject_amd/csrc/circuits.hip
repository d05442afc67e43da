// Host-side mirror of the reference's circuits (row a2 of the scope table: synthesis stays on the host) — pure host
// code, no GPU involved.  The reference is Rust (ark-relations / ark-r1cs-std); no Rust toolchain exists in this image,
// so the same constraint systems are produced here in C++ with the same allocation order, so that variable k of the
// reference is variable k here:
//   FibonacciCircuit   /root/reference/src/arkworks/constraints/fibbonaci.rs:22-48
//   MatrixCircuit      /root/reference/src/arkworks/matrix_proof_of_work/constraints.rs:78-128  (+ alloc.rs:43-49)
//   Poseidon sponge    /root/reference/src/arkworks/matrix_proof_of_work/hasher.rs:17-40 and the gadget copy at
//                      hashing/hashing_utils.rs:737-877 (ark-crypto-primitives 0.4 PoseidonSponge / PoseidonSpongeVar);
//                      parameters = hashing_utils.rs:15-716 (data: poseidon_bls381_params.json -> poseidon_params.inc)
// Semantics restated from ark-r1cs-std 0.4 `FpVar` (constants fold; Var*Var allocates a product witness and one
// constraint; addition and scaling are symbolic linear combinations, inlined at `finalize()`), `pow_by_constant`
// (square-and-multiply from the MSB starting at the constant 1: x^17 = 4 squarings + 1 product = 5 constraints),
// `enforce_equal` ((a - b) * 1 = 0) and `mul_equals` (a * b = c).  Matrices are exported as CSR with the instance
// variables first (ark `ConstraintMatrices`).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/zkg16.h"
#include "ff.cuh"
#include "matrix_plan.hpp"

using namespace zk;

namespace {

#include "poseidon_params.inc"

Fr fr_from_canon(const uint64_t l[4]) {
    Fr c;
    for (int i = 0; i < 4; i++) { c.l[2 * i] = (uint32_t)l[i]; c.l[2 * i + 1] = (uint32_t)(l[i] >> 32); }
    return fp_to_mont(c);
}
Fr fr_from_u64(uint64_t v) {
    const uint64_t l[4] = {v, 0, 0, 0};
    return fr_from_canon(l);
}

// variable id: instance variables (incl. the constant One = instance 0) and witness variables are numbered separately
// during synthesis; bit 31 marks a witness.  Final column = instance index, or num_instance + witness index.
typedef uint32_t VarId;
const VarId WIT = 0x80000000u;

struct Term { VarId v; Fr c; };

// false while a Poseidon permutation is replayed from a template (permute_gadget): values, witness allocation and the
// is_const flags follow the normal code path, term vectors and constraint rows are filled in from the template afterwards
static thread_local bool g_terms = true;

// An FpVar: a linear combination over variables (sorted by id) together with its value.  Constants are LCs over the
// One variable only (`is_const`), mirroring FpVar::Constant.
struct Lc {
    std::vector<Term> t;
    Fr val = Fr::zero();
    bool is_const = true;    // only the One variable (or nothing) appears
};

struct Circuit {
    std::vector<Fr> instance;              // instance[0] = 1
    std::vector<Fr> witness;
    // constraint rows of A, B, C in flat CSR form (one allocation per matrix, not one per row)
    struct Rows {
        std::vector<uint64_t> ptr{0};
        std::vector<Term> t;
        size_t size() const { return ptr.size() - 1; }
        void push(const std::vector<Term> &row) { t.insert(t.end(), row.begin(), row.end()); ptr.push_back(t.size()); }
        void push(const Term *row, size_t n) { t.insert(t.end(), row, row + n); ptr.push_back(t.size()); }
    } rows[3];

    // A circuit may be built as several segments (zkg16_circuit): a segment's own witnesses are numbered from wit_base in
    // the whole circuit's numbering; variables of earlier segments are referenced by id only (their values travel in Lc::val).
    size_t wit_base = 0;
    size_t next_wit() const { return wit_base + witness.size(); }

    Circuit() { instance.push_back(Fr::one()); }

    Lc constant(const Fr &c) const {
        Lc r;
        if (!c.is_zero()) r.t.push_back(Term{0, c});
        r.val = c;
        r.is_const = true;
        return r;
    }
    Lc new_input(const Fr &v) {
        instance.push_back(v);
        Lc r;
        r.t.push_back(Term{(VarId)(instance.size() - 1), Fr::one()});
        r.val = v;
        r.is_const = false;
        return r;
    }
    Lc new_witness(const Fr &v) {
        witness.push_back(v);
        Lc r;
        if (g_terms) r.t.push_back(Term{WIT | (VarId)(next_wit() - 1), Fr::one()});
        r.val = v;
        r.is_const = false;
        return r;
    }
    VarId new_witness_id(const Fr &v) {      // new_witness without the one-term Lc (a heap allocation per call)
        witness.push_back(v);
        return WIT | (VarId)(next_wit() - 1);
    }
    VarId new_input_id(const Fr &v) {
        instance.push_back(v);
        return (VarId)(instance.size() - 1);
    }
    static Lc add(const Lc &a, const Lc &b) {
        Lc r;
        r.val = fp_add(a.val, b.val);
        r.is_const = a.is_const && b.is_const;
        if (!g_terms) return r;
        r.t.reserve(a.t.size() + b.t.size());
        size_t i = 0, j = 0;
        while (i < a.t.size() || j < b.t.size()) {
            if (j == b.t.size() || (i < a.t.size() && a.t[i].v < b.t[j].v)) r.t.push_back(a.t[i++]);
            else if (i == a.t.size() || b.t[j].v < a.t[i].v) r.t.push_back(b.t[j++]);
            else {
                Fr c = fp_add(a.t[i].c, b.t[j].c);
                if (!c.is_zero()) r.t.push_back(Term{a.t[i].v, c});   // ark's make_row drops zero coefficients
                i++; j++;
            }
        }
        return r;
    }
    // a += c * One, in place (the One variable has id 0, so its term is always first)
    static void add_const(Lc &a, const Fr &c) {
        if (c.is_zero()) return;
        a.val = fp_add(a.val, c);
        if (!g_terms) return;
        if (!a.t.empty() && a.t[0].v == 0) {
            a.t[0].c = fp_add(a.t[0].c, c);
            if (a.t[0].c.is_zero()) a.t.erase(a.t.begin());
        } else {
            a.t.insert(a.t.begin(), Term{0, c});
        }
    }
    // out[i] = sum_j m[i][j] * in[j], i, j < 3: one three-way merge instead of nine scales and nine pairwise merges
    static void mix3(const Lc in[3], const Fr m[3][3], Lc out[3]) {
        size_t k[3] = {0, 0, 0};
        const size_t cap = in[0].t.size() + in[1].t.size() + in[2].t.size();
        for (int i = 0; i < 3; i++) {
            out[i].t.clear();
            out[i].t.reserve(cap);
            out[i].val = fp_add(fp_add(fp_mul(m[i][0], in[0].val), fp_mul(m[i][1], in[1].val)), fp_mul(m[i][2], in[2].val));
            out[i].is_const = in[0].is_const && in[1].is_const && in[2].is_const;
        }
        if (!g_terms) return;
        for (;;) {
            VarId v = 0xffffffffu;
            bool any = false;
            for (int j = 0; j < 3; j++)
                if (k[j] < in[j].t.size()) {
                    any = true;
                    if (in[j].t[k[j]].v < v) v = in[j].t[k[j]].v;
                }
            if (!any) break;
            Fr acc[3] = {Fr::zero(), Fr::zero(), Fr::zero()};
            for (int j = 0; j < 3; j++)
                if (k[j] < in[j].t.size() && in[j].t[k[j]].v == v) {
                    const Fr &c = in[j].t[k[j]].c;
                    for (int i = 0; i < 3; i++) acc[i] = fp_add(acc[i], fp_mul(m[i][j], c));
                    k[j]++;
                }
            for (int i = 0; i < 3; i++)
                if (!acc[i].is_zero()) out[i].t.push_back(Term{v, acc[i]});
        }
    }
    static Lc scale(const Lc &a, const Fr &c) {
        Lc r;
        if (c.is_zero()) { r.is_const = true; return r; }
        r.val = fp_mul(a.val, c);
        r.is_const = a.is_const;
        if (!g_terms) return r;
        r.t.reserve(a.t.size());
        for (const Term &x : a.t) r.t.push_back(Term{x.v, fp_mul(x.c, c)});
        return r;
    }
    static Lc sub(const Lc &a, const Lc &b) { return add(a, scale(b, fp_neg(Fr::one()))); }
    void enforce(const Lc &a, const Lc &b, const Lc &c) {
        if (!g_terms) return;
        rows[0].push(a.t);
        rows[1].push(b.t);
        rows[2].push(c.t);
    }
    // FpVar * FpVar
    Lc mul(const Lc &a, const Lc &b) {
        if (a.is_const) return scale(b, a.val);
        if (b.is_const) return scale(a, b.val);
        Lc p = new_witness(fp_mul(a.val, b.val));
        enforce(a, b, p);
        return p;
    }
    Lc square(const Lc &a) { return mul(a, a); }
    void mul_equals(const Lc &a, const Lc &b, const Lc &c) { enforce(a, b, c); }
    void enforce_equal(const Lc &a, const Lc &b) { enforce(sub(a, b), constant(Fr::one()), Lc()); }
    // FieldVar::pow_by_constant (BitIteratorBE::without_leading_zeros)
    Lc pow_by_constant(const Lc &x, unsigned e) {
        Lc res = constant(Fr::one());
        int top = 31;
        while (top > 0 && !((e >> top) & 1)) top--;
        for (int i = top; i >= 0; i--) {
            res = square(res);
            if ((e >> i) & 1) res = mul(res, x);
        }
        return res;
    }
    // rows of this segment under the whole circuit's assignment
    bool satisfied(const std::vector<Fr> &instance, const std::vector<Fr> &witness) const {
        const size_t ni = instance.size();
        auto eval = [&](const Rows &m, size_t i) {
            Fr acc = Fr::zero();
            for (uint64_t k = m.ptr[i]; k < m.ptr[i + 1]; k++) {
                const Term &x = m.t[k];
                const Fr &v = (x.v & WIT) ? witness[x.v & ~WIT] : instance[x.v];
                acc = fp_add(acc, fp_mul(x.c, v));
            }
            return acc;
        };
        (void)ni;
        for (size_t i = 0; i < rows[0].size(); i++)
            if (fp_mul(eval(rows[0], i), eval(rows[1], i)) != eval(rows[2], i)) return false;
        return true;
    }
};

// ------------------------------------------------------------------------------------------------ Poseidon
struct PoseidonParams {
    Fr mds[3][3], ark[37][3];
    PoseidonParams() {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) mds[i][j] = fr_from_canon(POSEIDON_MDS[i][j]);
        for (int r = 0; r < 37; r++)
            for (int j = 0; j < 3; j++) ark[r][j] = fr_from_canon(POSEIDON_ARK[r][j]);
    }
};
const PoseidonParams &pparams() {
    static PoseidonParams p;
    return p;
}
const int P_ROUNDS = POSEIDON_FULL + POSEIDON_PARTIAL, P_HALF = POSEIDON_FULL / 2;

Fr pow17(const Fr &x) {
    Fr r = x;
    for (int i = 0; i < 4; i++) r = fp_sqr(r);
    return fp_mul(r, x);
}
// native permutation (ark-crypto-primitives PoseidonSponge::permute)
void permute_native(Fr st[3]) {
    const PoseidonParams &p = pparams();
    for (int r = 0; r < P_ROUNDS; r++) {
        for (int i = 0; i < 3; i++) st[i] = fp_add(st[i], p.ark[r][i]);
        const bool full = r < P_HALF || r >= P_HALF + POSEIDON_PARTIAL;
        if (full) for (int i = 0; i < 3; i++) st[i] = pow17(st[i]);
        else st[0] = pow17(st[0]);
        Fr n[3];
        for (int i = 0; i < 3; i++) {
            Fr acc = Fr::zero();
            for (int j = 0; j < 3; j++) acc = fp_add(acc, fp_mul(st[j], p.mds[i][j]));
            n[i] = acc;
        }
        for (int i = 0; i < 3; i++) st[i] = n[i];
    }
}
// sponge.absorb(&elems); sponge.squeeze_native_field_elements(1)[0]  (hasher.rs:17-27)
Fr poseidon_hash_native(const Fr *elems, size_t n) {
    Fr st[3] = {Fr::zero(), Fr::zero(), Fr::zero()};
    size_t idx = 0, pos = 0;             // pos = next_absorb_index
    while (idx < n) {
        if (pos == POSEIDON_RATE) { permute_native(st); pos = 0; }
        st[POSEIDON_CAP + pos] = fp_add(st[POSEIDON_CAP + pos], elems[idx]);
        pos++; idx++;
    }
    permute_native(st);                  // squeeze from Absorbing mode permutes first
    return st[POSEIDON_CAP];
}
// the same through the gadget (PoseidonSpongeVar): returns state[1] as an FpVar
void permute_gadget_generic(Circuit &cs, Lc st[3]) {
    const PoseidonParams &p = pparams();
    for (int r = 0; r < P_ROUNDS; r++) {
        for (int i = 0; i < 3; i++) Circuit::add_const(st[i], p.ark[r][i]);
        const bool full = r < P_HALF || r >= P_HALF + POSEIDON_PARTIAL;
        if (full) for (int i = 0; i < 3; i++) st[i] = cs.pow_by_constant(st[i], POSEIDON_ALPHA);
        else st[0] = cs.pow_by_constant(st[0], POSEIDON_ALPHA);
        Lc n[3];
        Circuit::mix3(st, p.mds, n);
        for (int i = 0; i < 3; i++) st[i].t.swap(n[i].t), st[i].val = n[i].val, st[i].is_const = n[i].is_const;
    }
}

// Every permutation of a sponge after the first sees the same *shape* of input state (the same coefficients on a
// different set of variables), so its ~265 constraint rows are the first one's with the variable ids renamed.  The rows of
// the first permutation of each shape are kept as a template over "slots" (the distinct input variables in id order, then
// the witnesses the permutation allocates, in allocation order); later permutations run the generic code for values and
// witness allocation only and copy the rows.  Renaming is order-preserving (slots are ranked by id and new witnesses are
// larger than everything before them), so rows stay sorted by column exactly as the generic path leaves them.
struct PermTemplate {
    std::vector<uint32_t> shape;    // per state LC: #terms, is_const, then the slot of every term
    std::vector<Fr> coeff;          // the coefficients in the same order
    bool slot0_is_one = false;
    uint32_t n_slots = 0, n_new = 0;
    std::vector<uint32_t> ptr[3];   // rows as CSR over template ids (slot, or n_slots + new-witness index)
    std::vector<Term> t[3];
    std::vector<Term> out[3];
    bool out_const[3] = {false, false, false};
};
struct PermTemplates {
    std::vector<PermTemplate> list;
    bool enabled = true;
};

static void state_signature(const Lc st[3], std::vector<VarId> &ids, std::vector<uint32_t> &shape, std::vector<Fr> &coeff) {
    ids.clear(); shape.clear(); coeff.clear();
    ids.push_back(0);       // the One variable is always slot 0: round constants put it into rows whether or not the state has it
    for (int i = 0; i < 3; i++)
        for (const Term &x : st[i].t) ids.push_back(x.v);
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    for (int i = 0; i < 3; i++) {
        shape.push_back((uint32_t)st[i].t.size());
        shape.push_back(st[i].is_const ? 1u : 0u);
        for (const Term &x : st[i].t) {
            shape.push_back((uint32_t)(std::lower_bound(ids.begin(), ids.end(), x.v) - ids.begin()));
            coeff.push_back(x.c);
        }
    }
}

void permute_gadget(Circuit &cs, PermTemplates &tpls, Lc st[3]) {
    if (!tpls.enabled) { permute_gadget_generic(cs, st); return; }
    std::vector<VarId> ids;
    std::vector<uint32_t> shape;
    std::vector<Fr> coeff;
    state_signature(st, ids, shape, coeff);
    const bool one0 = true;     // see state_signature
    const PermTemplate *hit = nullptr;
    for (const PermTemplate &t : tpls.list)
        if (t.slot0_is_one == one0 && t.shape == shape && t.coeff.size() == coeff.size() &&
            memcmp(t.coeff.data(), coeff.data(), coeff.size() * sizeof(Fr)) == 0) { hit = &t; break; }
    const size_t w0 = cs.next_wit();
    if (!hit) {
        size_t r0[3], k0[3];
        for (int m = 0; m < 3; m++) { r0[m] = cs.rows[m].size(); k0[m] = cs.rows[m].t.size(); }
        permute_gadget_generic(cs, st);
        PermTemplate t;
        t.shape = shape; t.coeff = coeff; t.slot0_is_one = one0;
        t.n_slots = (uint32_t)ids.size();
        t.n_new = (uint32_t)(cs.next_wit() - w0);
        bool ok = true;
        auto to_tpl = [&](VarId v) -> VarId {
            if ((v & WIT) && (v & ~WIT) >= w0) return t.n_slots + (VarId)((v & ~WIT) - w0);
            auto it = std::lower_bound(ids.begin(), ids.end(), v);
            if (it == ids.end() || *it != v) { ok = false; return 0; }     // a variable from outside the state: not replayable
            return (VarId)(it - ids.begin());
        };
        for (int m = 0; m < 3; m++) {
            const Circuit::Rows &R = cs.rows[m];
            t.ptr[m].push_back(0);
            for (size_t r = r0[m]; r < R.size(); r++) {
                for (uint64_t k = R.ptr[r]; k < R.ptr[r + 1]; k++) t.t[m].push_back(Term{to_tpl(R.t[k].v), R.t[k].c});
                t.ptr[m].push_back((uint32_t)t.t[m].size());
            }
            (void)k0;
        }
        for (int i = 0; i < 3; i++) {
            for (const Term &x : st[i].t) t.out[i].push_back(Term{to_tpl(x.v), x.c});
            t.out_const[i] = st[i].is_const;
        }
        if (ok && tpls.list.size() < 16) tpls.list.push_back(std::move(t));
        return;
    }
    g_terms = false;
    permute_gadget_generic(cs, st);
    g_terms = true;
    if (cs.next_wit() - w0 != hit->n_new) throw std::logic_error("poseidon template: witness count mismatch");
    auto from_tpl = [&](VarId v) -> VarId { return v < hit->n_slots ? ids[v] : (WIT | (VarId)(w0 + (v - hit->n_slots))); };
    for (int m = 0; m < 3; m++) {
        Circuit::Rows &R = cs.rows[m];
        const uint64_t base = R.t.size();
        for (const Term &x : hit->t[m]) R.t.push_back(Term{from_tpl(x.v), x.c});
        for (size_t r = 1; r < hit->ptr[m].size(); r++) R.ptr.push_back(base + hit->ptr[m][r]);
    }
    for (int i = 0; i < 3; i++) {
        st[i].t.clear();
        for (const Term &x : hit->out[i]) st[i].t.push_back(Term{from_tpl(x.v), x.c});
        if (st[i].is_const != hit->out_const[i]) throw std::logic_error("poseidon template: state flag mismatch");
    }
}
Lc poseidon_hash_gadget(Circuit &cs, const std::vector<Lc> &elems) {
    PermTemplates tpls;
    const char *env = getenv("ZKG16_SYNTH_GENERIC");       // tests: force the plain path to compare against
    tpls.enabled = !(env && env[0] == '1');
    Lc st[3] = {cs.constant(Fr::zero()), cs.constant(Fr::zero()), cs.constant(Fr::zero())};
    size_t pos = 0;
    for (const Lc &e : elems) {
        if (pos == POSEIDON_RATE) { permute_gadget(cs, tpls, st); pos = 0; }
        st[POSEIDON_CAP + pos] = Circuit::add(st[POSEIDON_CAP + pos], e);
        pos++;
    }
    permute_gadget(cs, tpls, st);
    return st[POSEIDON_CAP];
}

// ---- one sponge over several segments.  Permutation p absorbs elements [RATE p, RATE p + RATE) and allocates 265 witnesses
// (the first one 260), so the witness offset of every permutation is known up front, and the state a permutation leaves is a
// fixed linear form over ITS OWN last three S-box witnesses whatever went in.  A chunk that starts at permutation p_lo >= 2
// therefore needs only the native state values in front of permutation p_lo - 1: it replays that one permutation on a
// scratch circuit (numbered where the real one is) to obtain the entering state as FpVars, then continues exactly as the
// sequential gadget would.  Chunks of one hash run on their own threads (build_matrix_circuit).
static constexpr size_t PERM_WITNESSES = 265;
struct SpongeChunk { Circuit *seg; size_t p_lo, p_hi; Fr pre[3]; };
template <class AddSegment>
static std::vector<SpongeChunk> plan_sponge_chunks(AddSegment &&add_segment, size_t wit_base, size_t count, size_t want) {
    const size_t perms = (count + POSEIDON_RATE - 1) / POSEIDON_RATE;
    size_t k = want;
    while (k > 1 && perms / k < 64) k--;
    std::vector<SpongeChunk> ch(k);
    for (size_t j = 0; j < k; j++) {
        ch[j].p_lo = j == 0 ? 0 : std::max<size_t>(2, perms * j / k);
        ch[j].p_hi = j + 1 == k ? perms : std::max<size_t>(2, perms * (j + 1) / k);
        ch[j].seg = &add_segment(wit_base + (ch[j].p_lo == 0 ? 0 : ch[j].p_lo * PERM_WITNESSES - 5));
        for (int i = 0; i < 3; i++) ch[j].pre[i] = Fr::zero();
    }
    return ch;
}
// the native sponge, recording for every chunk after the first the state in front of permutation p_lo - 1; returns the hash
static Fr poseidon_native_with_cuts(const Fr *elems, size_t count, std::vector<SpongeChunk> &ch) {
    Fr st[3] = {Fr::zero(), Fr::zero(), Fr::zero()};
    const size_t perms = (count + POSEIDON_RATE - 1) / POSEIDON_RATE;
    size_t next = 1;
    for (size_t p = 0; p < perms; p++) {
        for (size_t pos = 0; pos < POSEIDON_RATE && p * POSEIDON_RATE + pos < count; pos++)
            st[POSEIDON_CAP + pos] = fp_add(st[POSEIDON_CAP + pos], elems[p * POSEIDON_RATE + pos]);
        if (next < ch.size() && ch[next].p_lo == p + 1) {
            for (int i = 0; i < 3; i++) ch[next].pre[i] = st[i];
            next++;
        }
        permute_native(st);
    }
    return st[POSEIDON_CAP];
}
static Lc poseidon_hash_chunk(const SpongeChunk &ck, const std::vector<Lc> &elems) {
    Circuit &cs = *ck.seg;
    PermTemplates tpls;
    const char *env = getenv("ZKG16_SYNTH_GENERIC");
    tpls.enabled = !(env && env[0] == '1');
    Lc st[3];
    if (ck.p_lo == 0) {
        for (int i = 0; i < 3; i++) st[i] = cs.constant(Fr::zero());
    } else {
        Circuit scratch;
        scratch.wit_base = cs.wit_base - PERM_WITNESSES;        // where permutation p_lo - 1 numbers its witnesses
        for (int i = 0; i < 3; i++) {
            st[i].t.assign(1, Term{WIT | (VarId)0, Fr::one()});   // any non-constant form: the state that comes out does not depend on it
            st[i].val = ck.pre[i];
            st[i].is_const = false;
        }
        permute_gadget_generic(scratch, st);
        if (scratch.witness.size() != PERM_WITNESSES) throw std::logic_error("sponge chunk: witness count of a permutation");
        for (int i = 0; i < 3; i++)
            for (const Term &x : st[i].t)
                if (!(x.v & WIT) || (x.v & ~WIT) < scratch.wit_base) throw std::logic_error("sponge chunk: leaving state refers to its input");
    }
    for (size_t p = ck.p_lo; p < ck.p_hi; p++) {
        for (size_t pos = 0; pos < POSEIDON_RATE && p * POSEIDON_RATE + pos < elems.size(); pos++)
            st[POSEIDON_CAP + pos] = Circuit::add(st[POSEIDON_CAP + pos], elems[p * POSEIDON_RATE + pos]);
        permute_gadget(cs, tpls, st);
    }
    return st[POSEIDON_CAP];
}
// all chunks of one hash: the native pass first (it also yields the hash value), then chunk 0 here and the others on threads
static Lc poseidon_hash_chunked(std::vector<SpongeChunk> &ch, const std::vector<Lc> &elems, const Fr *vals, Fr &hash_out) {
    hash_out = poseidon_native_with_cuts(vals, elems.size(), ch);
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(ch.size());
    std::vector<Lc> res(ch.size());
    auto run = [&](size_t j) {
        try { res[j] = poseidon_hash_chunk(ch[j], elems); } catch (...) { err[j] = std::current_exception(); g_terms = true; }
    };
    for (size_t j = 1; j < ch.size(); j++) th.emplace_back(run, j);
    run(0);
    for (auto &t : th) t.join();
    for (const std::exception_ptr &e : err)
        if (e) std::rethrow_exception(e);
    return res.back();
}

// helper threads of one build: joined on every exit path; a thread that cannot be started runs inline (never std::terminate)
class ThreadGroupLocal {
    std::vector<std::thread> th_;
  public:
    ThreadGroupLocal() = default;
    ThreadGroupLocal(const ThreadGroupLocal &) = delete;
    ThreadGroupLocal &operator=(const ThreadGroupLocal &) = delete;
    template <class Fn>
    void run(Fn fn) {
        try {
            th_.emplace_back(fn);
        } catch (const std::system_error &) {
            fn();
        }
    }
    ~ThreadGroupLocal() {
        for (auto &t : th_)
            if (t.joinable()) t.join();
    }
};

#include "prime_circuit.inc"

}  // namespace

struct zkg16_circuit {
    // segments in row order; segs[0] holds the instance variables.  Witness values: each segment's own, at its wit_base.
    std::vector<std::unique_ptr<Circuit>> segs;
    bool pooled_storage = false;       // the PrimeCircuit's per-part segments: their vectors go back to prime_store() when it is freed
    Circuit &head() { return *segs[0]; }
    const Circuit &head() const { return *segs[0]; }
    Circuit &add_segment(size_t wit_base) {
        segs.emplace_back(new Circuit());
        segs.back()->wit_base = wit_base;
        return *segs.back();
    }
    size_t num_witness() const {
        size_t n = 0;
        for (const auto &sg : segs) n += sg->witness.size();
        return n;
    }
    std::vector<Fr> flat_witness() const {
        std::vector<Fr> w(num_witness());
        for (const auto &sg : segs)
            if (!sg->witness.empty()) memcpy(w.data() + sg->wit_base, sg->witness.data(), sg->witness.size() * sizeof(Fr));
        return w;
    }
};

namespace {
// witnesses one PoseidonSpongeVar hash of `count` (>= 2) variable elements allocates: every permutation 53 S-boxes x 5
// products, minus the capacity element's S-box of the first permutation (a constant).  Used only to place the segments of
// the matrix circuit before they are built concurrently; checked against what the builders actually allocated.
size_t poseidon_hash_witnesses(size_t count) { return (count + POSEIDON_RATE - 1) / POSEIDON_RATE * 265 - 5; }
}  // namespace

// ------------------------------------------------------------------------------------------------ the MatrixCircuit's R1CS as a plan
// (matrix_plan.hpp).  The templates come out of the same gadget code the full synthesis runs, on a scratch circuit with
// at most nine absorbed elements: permutations 0, 1, 2 (and 3, to check that a later one reuses 2's template) and, for an odd
// element count, the last one with its single element.
namespace {

bool build_hash_plan(size_t n_elems_real, uint32_t elem_vars, MatrixPlanHash &out, std::vector<Term> *state1_last /* state[1] after the last permutation, real witness-relative ids */) {
    const size_t E = n_elems_real <= 9 ? n_elems_real : (n_elems_real % 2 == 0 ? 8 : 9);
    Circuit cs;
    std::vector<Lc> elems(E);
    for (size_t e = 0; e < E; e++) {
        elems[e] = cs.new_witness(Fr::zero());
        for (uint32_t k = 1; k < elem_vars; k++) elems[e] = Circuit::add(elems[e], cs.new_witness(Fr::zero()));
    }
    const size_t elem_wits = E * elem_vars;
    PermTemplates tpls;
    Lc st[3] = {cs.constant(Fr::zero()), cs.constant(Fr::zero()), cs.constant(Fr::zero())};
    const size_t perms_s = (E + POSEIDON_RATE - 1) / POSEIDON_RATE;
    std::vector<size_t> w0(perms_s), n_new(perms_s);
    std::vector<int> used(perms_s, -1);
    std::vector<std::vector<VarId>> slot_ids(perms_s);
    for (size_t q = 0; q < perms_s; q++) {
        for (size_t pos = 0; pos < (size_t)POSEIDON_RATE && q * POSEIDON_RATE + pos < E; pos++)
            st[POSEIDON_CAP + pos] = Circuit::add(st[POSEIDON_CAP + pos], elems[q * POSEIDON_RATE + pos]);
        std::vector<uint32_t> shape;
        std::vector<Fr> coeff;
        state_signature(st, slot_ids[q], shape, coeff);
        w0[q] = cs.next_wit();
        permute_gadget(cs, tpls, st);
        n_new[q] = cs.next_wit() - w0[q];
        for (size_t t = 0; t < tpls.list.size(); t++)
            if (tpls.list[t].shape == shape && tpls.list[t].coeff.size() == coeff.size() &&
                memcmp(tpls.list[t].coeff.data(), coeff.data(), coeff.size() * sizeof(Fr)) == 0) { used[q] = (int)t; break; }
        if (used[q] < 0) return false;                  // the permutation could not be turned into a template
    }
    const bool odd = n_elems_real % 2 == 1;
    // every scratch permutation from the third on (except an odd tail) must replay permutation 2's template
    for (size_t q = 3; q < perms_s; q++)
        if (!(odd && q + 1 == perms_s) && used[q] != used[2]) return false;
    out.perms = (uint32_t)((n_elems_real + POSEIDON_RATE - 1) / POSEIDON_RATE);
    out.odd_tail = odd;
    out.elem_vars = elem_vars;
    auto export_tpl = [&](size_t q, MatrixPlanTemplate &T) -> bool {
        const PermTemplate &P = tpls.list[used[q]];
        T.n_slots = P.n_slots;
        T.n_new = P.n_new;
        T.n_rows = (uint32_t)(P.ptr[0].size() - 1);
        T.slots.resize(P.n_slots);
        if (slot_ids[q].size() != P.n_slots) return false;
        for (uint32_t k = 0; k < P.n_slots; k++) {
            const VarId v = slot_ids[q][k];
            if (!(v & WIT)) {
                if (v != 0) return false;               // only the constant One among the instance variables
                T.slots[k] = MatrixPlanSlot{0, 0, 0};
                continue;
            }
            const size_t w = v & ~WIT;
            if (w < elem_wits) {
                const size_t e = w / elem_vars;
                if (e < q * POSEIDON_RATE || e >= q * POSEIDON_RATE + POSEIDON_RATE) return false;
                T.slots[k] = MatrixPlanSlot{1, (uint32_t)(e - q * POSEIDON_RATE), (uint32_t)(w % elem_vars)};
            } else {
                if (q == 0 || w < w0[q - 1] || w >= w0[q - 1] + n_new[q - 1]) return false;
                T.slots[k] = MatrixPlanSlot{2, (uint32_t)(w - w0[q - 1]), 0};
            }
        }
        for (int m = 0; m < 3; m++) {
            if (P.ptr[m].size() != T.n_rows + 1) return false;
            T.ptr[m] = P.ptr[m];
            T.id[m].resize(P.t[m].size());
            T.coeff[m].resize(P.t[m].size());
            for (size_t k = 0; k < P.t[m].size(); k++) { T.id[m][k] = P.t[m][k].v; T.coeff[m][k] = P.t[m][k].c; }
        }
        return true;
    };
    for (size_t q = 0; q < perms_s && q < 3; q++) {
        if (odd && q + 1 == perms_s && q > 0) break;     // that one is the tail class
        if (!export_tpl(q, out.tpl[q])) return false;
        out.has[q] = true;
    }
    if (odd && perms_s > 1) {
        if (!export_tpl(perms_s - 1, out.tpl[3])) return false;
        out.has[3] = true;
    }
    // classes the real circuit uses must exist
    for (uint32_t p : {0u, 1u, 2u, out.perms - 1})
        if (p < out.perms && !out.has[out.cls(p)]) return false;
    // state[1] after the last permutation: the last scratch permutation's leaving state is a form over ITS OWN witnesses, at the same
    // offsets in the real last permutation (same template class)
    if (state1_last) {
        state1_last->clear();
        for (const Term &x : st[POSEIDON_CAP].t) {
            if (!(x.v & WIT)) return false;
            const size_t w = x.v & ~WIT;
            if (w < w0[perms_s - 1]) return false;
            state1_last->push_back(Term{(VarId)(w - w0[perms_s - 1]), x.c});
        }
    }
    return true;
}

}  // namespace

namespace zk {

bool matrix_plan_build(size_t n, MatrixPlan &P) {
    if (n < 2 || n > 1024) return false;
    const size_t nn = n * n, ni = 4;
    P.n = n; P.nn = nn; P.num_instance = ni;
    std::vector<Term> s1[3];
    if (!build_hash_plan(nn, 1, P.hash[0], &s1[0])) return false;
    if (!build_hash_plan(nn, 1, P.hash[1], &s1[1])) return false;
    if (!build_hash_plan(nn, (uint32_t)(n + 1), P.hash[2], &s1[2])) return false;
    // witnesses: a | b | gadget a | gadget b | n^2 zeros | per (i, j): seed + n products | gadget c
    uint64_t hw[3];
    for (int h = 0; h < 3; h++) {
        uint64_t rows, nz[3];
        (void)rows; (void)nz;
        hw[h] = 0;
        for (uint32_t p = 0; p < P.hash[h].perms; p++) hw[h] += P.hash[h].tpl[P.hash[h].cls(p)].n_new;
    }
    P.col_a0 = ni;
    P.col_b0 = ni + nn;
    P.hash[0].elem_col0 = P.col_a0;
    P.hash[1].elem_col0 = P.col_b0;
    P.hash[0].wit_col0 = ni + 2 * nn;
    P.hash[1].wit_col0 = P.hash[0].wit_col0 + hw[0];
    const uint64_t mm_w0 = P.hash[1].wit_col0 + hw[1];           // the n^2 pre-allocated zeros
    P.col_prod0 = mm_w0 + nn;                                    // block of (0, 0): [seed, n products]
    P.hash[2].elem_col0 = P.col_prod0;
    P.hash[2].wit_col0 = P.col_prod0 + (uint64_t)nn * (n + 1);
    P.num_witness = (size_t)(P.hash[2].wit_col0 + hw[2] - ni);
    // rows and non-zeros: hash a | hash b | eq a, eq b | matrix_mul | hash c | eq c
    uint64_t row = 0, nz[3] = {0, 0, 0};
    auto place_hash = [&](int h) {
        P.hash[h].row0 = row;
        for (int m = 0; m < 3; m++) P.hash[h].nnz0[m] = nz[m];
        uint64_t r, k[3];
        matrix_plan_prefix(P.hash[h], P.hash[h].perms, r, k);
        row += r;
        for (int m = 0; m < 3; m++) nz[m] += k[m];
    };
    const Fr one = Fr::one(), minus_one = fp_neg(Fr::one());
    auto place_eq = [&](int which) {
        // enforce_equal(hash, input): (state[1] - input) * 1 = 0, the instance variable first (columns are sorted)
        const MatrixPlanHash &H = P.hash[which];
        uint64_t last_w0 = H.wit_col0;
        for (uint32_t p = 0; p + 1 < H.perms; p++) last_w0 += H.tpl[H.cls(p)].n_new;
        MatrixPlanRow &A = P.eq[which][0], &B = P.eq[which][1], &C = P.eq[which][2];
        A.col.clear(); A.coeff.clear(); B.col.clear(); B.coeff.clear(); C.col.clear(); C.coeff.clear();
        A.col.push_back((uint32_t)(which + 1));          // hash_a, hash_b, hash_c are instance variables 1, 2, 3
        A.coeff.push_back(minus_one);
        for (const Term &x : s1[which]) { A.col.push_back((uint32_t)(last_w0 + x.v)); A.coeff.push_back(x.c); }
        B.col.push_back(0);
        B.coeff.push_back(one);
        P.eq_row[which] = row;
        for (int m = 0; m < 3; m++) { P.eq_nnz0[which][m] = nz[m]; nz[m] += P.eq[which][m].col.size(); }
        row += 1;
    };
    place_hash(0);
    place_hash(1);
    place_eq(0);
    place_eq(1);
    P.mm_row0 = row;
    for (int m = 0; m < 3; m++) { P.mm_nnz0[m] = nz[m]; nz[m] += 2 * (uint64_t)nn * n; }
    row += 2 * (uint64_t)nn * n;
    place_hash(2);
    place_eq(2);
    P.num_constraints = (size_t)row;
    for (int m = 0; m < 3; m++) P.nnz[m] = nz[m];
    return P.num_instance + P.num_witness < ((uint64_t)1 << 32);
}

// the plan written out by plain loops: the reference the device kernel (matrix_r1cs.hip) is tested against, itself tested against
// the full synthesis (zkg16_circuit_matrix + zkg16_circuit_export)
void matrix_plan_instantiate_host(const MatrixPlan &P, uint64_t *const rp[3], uint32_t *const col[3], Fr *const cf[3]) {
    const Fr one = Fr::one();
    for (int m = 0; m < 3; m++) rp[m][0] = 0;
    for (int h = 0; h < 3; h++) {
        const MatrixPlanHash &H = P.hash[h];
        uint64_t w0_prev = 0, w0 = H.wit_col0;
        for (uint32_t p = 0; p < H.perms; p++) {
            const MatrixPlanTemplate &T = H.tpl[H.cls(p)];
            uint64_t r0, k0[3];
            matrix_plan_prefix(H, p, r0, k0);
            auto real = [&](uint32_t id) -> uint32_t {
                if (id >= T.n_slots) return (uint32_t)(w0 + (id - T.n_slots));
                const MatrixPlanSlot &sl = T.slots[id];
                if (sl.kind == 0) return 0;
                if (sl.kind == 1) return (uint32_t)(H.elem_col0 + ((uint64_t)POSEIDON_RATE * p + sl.a) * H.elem_vars + sl.b);
                return (uint32_t)(w0_prev + sl.a);
            };
            for (int m = 0; m < 3; m++) {
                const uint64_t kb = H.nnz0[m] + k0[m];
                for (uint32_t r = 0; r < T.n_rows; r++) rp[m][H.row0 + r0 + r + 1] = kb + T.ptr[m][r + 1];
                for (size_t k = 0; k < T.id[m].size(); k++) { col[m][kb + k] = real(T.id[m][k]); cf[m][kb + k] = T.coeff[m][k]; }
            }
            w0_prev = w0;
            w0 += T.n_new;
        }
    }
    for (int w = 0; w < 3; w++)
        for (int m = 0; m < 3; m++) {
            const MatrixPlanRow &R = P.eq[w][m];
            for (size_t k = 0; k < R.col.size(); k++) { col[m][P.eq_nnz0[w][m] + k] = R.col[k]; cf[m][P.eq_nnz0[w][m] + k] = R.coeff[k]; }
            rp[m][P.eq_row[w] + 1] = P.eq_nnz0[w][m] + R.col.size();
        }
    const size_t n = P.n;
    for (uint64_t t = 0; t < 2 * (uint64_t)P.nn * n; t++) {
        const uint64_t pr = t >> 1, cell = pr / n, k = pr % n, i = cell / n, j = cell % n;
        const uint32_t c[3] = {(uint32_t)(P.col_a0 + i * n + k), (uint32_t)(P.col_b0 + k * n + j), (uint32_t)(P.col_prod0 + cell * (n + 1) + 1 + k)};
        for (int m = 0; m < 3; m++) {
            col[m][P.mm_nnz0[m] + t] = c[m];
            cf[m][P.mm_nnz0[m] + t] = one;
            rp[m][P.mm_row0 + t + 1] = P.mm_nnz0[m] + t + 1;
        }
    }
}

}  // namespace zk

extern "C" {

// The MatrixCircuit's R1CS of size n from its plan (host loops): dims, then the CSR arrays as zkg16_circuit_export lays them out.
int zkg16_matrix_r1cs_dims(size_t n, size_t *num_constraints, size_t *num_witness, size_t nnz[3]) {
    try {
        MatrixPlan P;
        if (!matrix_plan_build(n, P)) return ZKG16_ERR_UNSUPPORTED;
        if (num_constraints) *num_constraints = P.num_constraints;
        if (num_witness) *num_witness = P.num_witness;
        if (nnz) for (int m = 0; m < 3; m++) nnz[m] = (size_t)P.nnz[m];
    } catch (const std::bad_alloc &) {
        return ZKG16_ERR_OOM;
    } catch (const std::exception &) {
        g_terms = true;
        return ZKG16_ERR_UNSUPPORTED;
    }
    return ZKG16_OK;
}
int zkg16_matrix_r1cs_host(size_t n, uint64_t *const row_ptr[3], uint32_t *const col[3], uint64_t *const coeff[3]) {
    if (!row_ptr || !col || !coeff) return ZKG16_ERR_BAD_ARG;
    try {
        MatrixPlan P;
        if (!matrix_plan_build(n, P)) return ZKG16_ERR_UNSUPPORTED;
        Fr *cf[3] = {reinterpret_cast<Fr *>(coeff[0]), reinterpret_cast<Fr *>(coeff[1]), reinterpret_cast<Fr *>(coeff[2])};
        matrix_plan_instantiate_host(P, row_ptr, col, cf);
    } catch (const std::bad_alloc &) {
        return ZKG16_ERR_OOM;
    } catch (const std::exception &) {
        g_terms = true;
        return ZKG16_ERR_UNSUPPORTED;
    }
    return ZKG16_OK;
}

// FibonacciCircuit { a, b, num_of_steps, result } with result computed in Fr (the reference's u128 helper overflows
// above 186 rounds: SURVEY.md F8).
int zkg16_circuit_fibonacci(uint64_t a, uint64_t b, size_t steps, zkg16_circuit **out) {
    if (!out) return ZKG16_ERR_BAD_ARG;
    auto c = new (std::nothrow) zkg16_circuit();
    if (!c) return ZKG16_ERR_OOM;
    Circuit &cs = c->add_segment(0);
    const Fr fa = fr_from_u64(a), fb = fr_from_u64(b);
    // fibbonaci_handler.rs:13-27 semantics: after `steps` rounds the result is f_{steps} of the (a, b) sequence
    Fr x = fa, y = fb, res = steps == 0 ? Fr::zero() : fb;
    for (size_t i = 0; i < steps; i++) { res = fp_add(x, y); x = y; y = res; }
    Lc f2 = cs.new_input(fa), f1 = cs.new_input(fb), saved = cs.new_input(res);
    Lc fi = cs.new_witness(Fr::zero());
    for (size_t i = 0; i < steps; i++) {
        fi = Circuit::add(f1, f2);
        cs.enforce_equal(fi, Circuit::add(f1, f2));
        f2 = f1;
        f1 = fi;
    }
    cs.enforce_equal(fi, saved);
    *out = c;
    return ZKG16_OK;
}

// The native search of the prime handler (backend/prime_snark.rs:57-70): the first j in [0, i_max] for which
// hash(x + j) mod 2^20 passes the Fermat test with the three hashed bases (check_if_next_is_prime).
int zkg16_prime_search(uint64_t x, uint64_t i_max, uint64_t *j_out, uint32_t *prime_out, uint8_t digest_out[32], int *found) {
    if (!found) return ZKG16_ERR_BAD_ARG;
    *found = 0;
    for (uint64_t j = 0; j <= i_max; j++) {
        const PrimeCandidate c = prime_candidate(x, j);
        if (c.is_prime) {
            *found = 1;
            if (j_out) *j_out = j;
            if (prime_out) *prime_out = c.n;
            if (digest_out) memcpy(digest_out, c.digest, 32);
            return ZKG16_OK;
        }
        if (j == UINT64_MAX) break;
    }
    return ZKG16_OK;
}
// one candidate natively: digest = SHA-256(le32(x + j)), n = digest mod 2^20, the three bases hash(r || k) mod n and the
// Fermat verdict (tests compare these with hashlib / pow)
int zkg16_prime_candidate(uint64_t x, uint64_t j, uint8_t digest_out[32], uint32_t *n_out, uint32_t bases_out[3], uint8_t r_bytes_out[32],
                          int *is_prime) {
    const PrimeCandidate c = prime_candidate(x, j);
    if (digest_out) memcpy(digest_out, c.digest, 32);
    if (n_out) *n_out = c.n;
    if (bases_out) for (int k = 0; k < PRIME_K; k++) bases_out[k] = c.n >= 2 ? c.base[k] : 0;
    if (r_bytes_out) memcpy(r_bytes_out, c.r_bytes, 32);
    if (is_prime) *is_prime = c.is_prime ? 1 : 0;
    return ZKG16_OK;
}
// PrimeCircuit::new(...) for the candidate at index j, as prove_prime builds it for the j it found and verify_prime
// rebuilds it to recover the public inputs (prime_snark.rs:98-105, 170-193).  Instance: 1, x, the 256 digest bits.
// The PrimeCircuit's public inputs for candidate j of x — x, then the 256 bits of SHA-256(x + j) (bit t of byte k at position
// 8 k + t) — without building the circuit: what verify_prime (prime_snark.rs:165-206) re-synthesizes the whole circuit for.
// out: 257 x 4 limbs, Montgomery.  (Checked against the instance of zkg16_circuit_prime in tests/test_prime_circuit.py.)
int zkg16_prime_public_inputs(uint64_t x, uint64_t j, uint64_t *out) {
    if (!out) return ZKG16_ERR_BAD_ARG;
    const PrimeCandidate cand = prime_candidate(x, j);
    Fr *o = reinterpret_cast<Fr *>(out);
    o[0] = fr_from_u64(x);
    for (int k = 0; k < 256; k++) o[1 + k] = ((cand.digest[k >> 3] >> (k & 7)) & 1) ? Fr::one() : Fr::zero();
    return ZKG16_OK;
}

// The PrimeCircuit's storage (75 MB of rows and witness values, the same size for every request) kept between requests: fresh
// vectors of that size come from mmap and the first build of a process pays ~18 k page faults (90 ms on the GPU box's host against
// the 18 ms of the synthesis itself).  One set of per-part vectors; a second concurrent build simply allocates.
namespace {
struct PrimeStore {
    std::mutex mu;
    struct Part {
        bool full = false;
        std::vector<Fr> witness;
        std::vector<Term> t[3];
        std::vector<uint64_t> ptr[3];
    } part[PRIME_PARTS];
    PrimeLayout layout;        // recorded by the first (sequential) build
};
PrimeStore &prime_store() {
    static PrimeStore s;
    return s;
}
void prime_store_take(Circuit &seg, int p) {
    PrimeStore &st = prime_store();
    std::lock_guard<std::mutex> lk(st.mu);
    PrimeStore::Part &pt = st.part[p];
    if (!pt.full) return;
    pt.full = false;
    seg.witness.swap(pt.witness);
    seg.witness.clear();
    for (int m = 0; m < 3; m++) {
        seg.rows[m].t.swap(pt.t[m]);
        seg.rows[m].ptr.swap(pt.ptr[m]);
        seg.rows[m].t.clear();
        seg.rows[m].ptr.assign(1, 0);
    }
}
void prime_store_give(Circuit &seg, int p) {
    PrimeStore &st = prime_store();
    std::lock_guard<std::mutex> lk(st.mu);
    PrimeStore::Part &pt = st.part[p];
    if (pt.full) return;
    pt.witness.swap(seg.witness);
    for (int m = 0; m < 3; m++) {
        pt.t[m].swap(seg.rows[m].t);
        pt.ptr[m].swap(seg.rows[m].ptr);
    }
    pt.full = true;
}
}  // namespace

// The first build of a process runs the seven parts of the circuit (prime_circuit.inc) one after another in a single segment and
// records how many witnesses precede each part; every later build gives each part its own segment and thread.  The counts do not
// depend on (x, j); should a part ever allocate a different number the build is repeated sequentially
// (tests/test_prime_circuit.py compares both forms array by array).  ZKG16_SYNTH_THREADS=0: always sequential.
int zkg16_circuit_prime(uint64_t x, uint64_t j, zkg16_circuit **out) {
    if (!out) return ZKG16_ERR_BAD_ARG;
    *out = nullptr;
    const PrimeCandidate cand = prime_candidate(x, j);
    if (cand.n < 2) return ZKG16_ERR_UNSUPPORTED;          // the reference's BigUint modpow panics on a modulus below 2
    for (int k = 0; k < PRIME_K; k++)
        if (cand.base[k] == 0) return ZKG16_ERR_UNSUPPORTED;  // base.inverse().unwrap() panics upstream
    try {
        const bool trace = getenv("ZKG16_TRACE_HOST") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        const char *env = getenv("ZKG16_SYNTH_THREADS");
        PrimeLayout lay;
        {
            PrimeStore &st = prime_store();
            std::lock_guard<std::mutex> lk(st.mu);
            lay = st.layout;
        }
        std::unique_ptr<zkg16_circuit> c;
        bool parallel = lay.known && !(env && env[0] == '0') && std::thread::hardware_concurrency() > 1;
        if (parallel) {
            c.reset(new zkg16_circuit());
            c->pooled_storage = true;
            for (int p = 0; p < PRIME_PARTS; p++) {
                Circuit &seg = c->add_segment(lay.wit_base[p]);
                prime_store_take(seg, p);
                seg.witness.reserve(lay.wit_base[p + 1] - lay.wit_base[p] + 16);
                for (int m = 0; m < 3; m++) { seg.rows[m].t.reserve(lay.nnz_hint[p][m] + 64); seg.rows[m].ptr.reserve(lay.rows_hint[p] + 16); }
            }
            std::atomic<bool> failed{false};
            {
                ThreadGroupLocal tg;
                for (int p = 1; p < PRIME_PARTS; p++)
                    tg.run([&, p] {
                        try {
                            PrimeShared sh = lay.sh;
                            prime_part(*c->segs[p], p, x, j, cand, sh);
                        } catch (...) {
                            failed = true;
                        }
                    });
                try {
                    PrimeShared sh = lay.sh;
                    prime_part(*c->segs[0], 0, x, j, cand, sh);
                } catch (...) {
                    failed = true;
                }
            }
            bool ok = !failed;
            for (int p = 0; ok && p < PRIME_PARTS; p++) ok = c->segs[p]->witness.size() == lay.wit_base[p + 1] - lay.wit_base[p];
            if (!ok) {
                std::lock_guard<std::mutex> lk(prime_store().mu);
                prime_store().layout.known = false;
                parallel = false;
            }
        }
        if (!parallel) {
            c.reset(new zkg16_circuit());
            Circuit &seg = c->add_segment(0);
            PrimeLayout fresh;
            build_prime_circuit(seg, x, j, cand, &fresh);
            std::lock_guard<std::mutex> lk(prime_store().mu);
            prime_store().layout = fresh;
        }
        if (trace) fprintf(stderr, "zkg16_circuit_prime: built in %.2f ms (%s)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                           parallel ? "seven threads" : "one thread");
        *out = c.release();
    } catch (const std::bad_alloc &) {
        return ZKG16_ERR_OOM;
    } catch (const std::exception &) {
        return ZKG16_ERR_UNSUPPORTED;
    }
    return ZKG16_OK;
}

// MatrixCircuit::new(matrix_a, matrix_b, hash_a, hash_b, hash_c) with the hashes computed natively as the handler does
// (matrix_proof.rs:104-125).  a, b: n*n u64 entries, row-major.
// The three hash gadgets do not depend on each other, so the circuit is built as segments in arkworks' allocation order —
// [inputs a, b + matrix witnesses] [hash_a gadget] [hash_b gadget] [two equalities] [matrix_mul + hash_c gadget] [input c +
// equality] — with the hash_a and hash_b segments on their own threads; each segment numbers its witnesses from the offset
// the earlier ones will have used (poseidon_hash_witnesses).  ZKG16_SYNTH_THREADS=0 builds them one after another.
static void build_matrix_circuit(zkg16_circuit *c, size_t n, const uint64_t *a, const uint64_t *b, bool threaded) {
    const size_t nn = n * n;
    std::vector<Fr> av(nn), bv(nn), cv(nn, Fr::zero());
    for (size_t i = 0; i < nn; i++) { av[i] = fr_from_u64(a[i]); bv[i] = fr_from_u64(b[i]); }
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) {
            Fr s = Fr::zero();
            for (size_t k = 0; k < n; k++) s = fp_add(s, fp_mul(av[i * n + k], bv[k * n + j]));
            cv[i * n + j] = s;
        }
    // generate_constraints (constraints.rs:101-128)
    Circuit &head = c->add_segment(0);
    std::vector<Lc> ma(nn), mb(nn);
    Fr hash_a, hash_b, hash_c;
    Lc ha, hb, hc;
    const size_t hw = threaded ? poseidon_hash_witnesses(nn) : 0;
    // every hash is cut into chunks of permutations with a thread each (plan_sponge_chunks): 3 x 4 + the matrix_mul thread fit
    // the 16 host cores a GPU comes with; in-order building keeps one segment per hash
    size_t chunks = 1;
    if (threaded) {
        const char *ce = getenv("ZKG16_SYNTH_CHUNKS");
        chunks = ce ? (size_t)atoi(ce) : 4;
        if (chunks < 1 || chunks > 16) chunks = 1;
    }
    auto add_seg = [c](size_t base) -> Circuit & { return c->add_segment(base); };
    std::vector<SpongeChunk> ch_a = plan_sponge_chunks(add_seg, 2 * nn, nn, chunks);
    std::vector<SpongeChunk> ch_b = plan_sponge_chunks(add_seg, 2 * nn + hw, nn, chunks);
    Circuit &seg_a = *ch_a[0].seg, &seg_b = *ch_b[0].seg;
    Circuit &seg_mid = c->add_segment(0);
    // matrix_mul: rows [n t / parts, n (t + 1) / parts) of the product per segment (each (i, j) allocates 1 + n witnesses)
    const size_t mm_parts = (threaded && n >= 32) ? 3 : 1;
    std::vector<Circuit *> seg_mm;
    for (size_t t = 0; t < mm_parts; t++)
        seg_mm.push_back(&c->add_segment(2 * nn + 2 * hw + (t == 0 ? 0 : nn + (n * t / mm_parts) * n * (n + 1))));
    Circuit &seg_c = *seg_mm[0];
    std::vector<SpongeChunk> ch_d = plan_sponge_chunks(add_seg, 2 * nn + 2 * hw + nn + nn * (n + 1), nn, chunks);      // hash of the product: starts after matrix_mul's witnesses
    Circuit &seg_d = *ch_d[0].seg;
    Circuit &seg_tail = c->add_segment(0);
    std::exception_ptr err_a, err_b, err_c, err_d;
    const bool synth_trace = getenv("ZKG16_SYNTH_TRACE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (synth_trace) fprintf(stderr, "synthesis: %s done at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    };
    auto build_a = [&]() {
        try {
            if (ch_a.size() > 1) {
                ha = poseidon_hash_chunked(ch_a, ma, av.data(), hash_a);
            } else {
                hash_a = poseidon_hash_native(av.data(), nn);
                ha = poseidon_hash_gadget(seg_a, ma);
            }
            lap("hash_a");
        } catch (...) { err_a = std::current_exception(); g_terms = true; }
    };
    auto build_b = [&]() {
        try {
            if (ch_b.size() > 1) {
                hb = poseidon_hash_chunked(ch_b, mb, bv.data(), hash_b);
            } else {
                hash_b = poseidon_hash_native(bv.data(), nn);
                hb = poseidon_hash_gadget(seg_b, mb);
            }
            lap("hash_b");
        } catch (...) { err_b = std::current_exception(); g_terms = true; }
    };
    // hash of C: entry (i, j) of the product is the symbolic sum  sum_ij + sum_k product_ijk  (constraints.rs:87-92), whose
    // variables matrix_mul allocates in a fixed order — [n*n pre-allocated entries][per (i, j): sum, then n products] — so the
    // linear combinations can be written down without waiting for matrix_mul to run
    auto build_d = [&]() {
        try {
            std::vector<Lc> mc(nn);
            const Fr one = Fr::one();
            for (size_t e = 0; e < nn; e++) {
                const size_t first = seg_c.wit_base + nn + e * (n + 1);
                mc[e].t.reserve(n + 1);
                for (size_t k = 0; k <= n; k++) mc[e].t.push_back(Term{WIT | (VarId)(first + k), one});
                mc[e].val = cv[e];
                mc[e].is_const = false;
            }
            if (ch_d.size() > 1) {
                hc = poseidon_hash_chunked(ch_d, mc, cv.data(), hash_c);
            } else {
                hash_c = poseidon_hash_native(cv.data(), nn);
                hc = poseidon_hash_gadget(seg_d, mc);
            }
            lap("hash_c");
        } catch (...) { err_d = std::current_exception(); g_terms = true; }
    };
    // the inputs come first in arkworks' order, but their values (the native hashes) are only needed at the end: the
    // instance slots are reserved now and filled after the builders have produced them
    Lc in_a = head.new_input(Fr::zero()), in_b = head.new_input(Fr::zero());
    for (size_t i = 0; i < nn; i++) ma[i] = head.new_witness(av[i]);
    for (size_t i = 0; i < nn; i++) mb[i] = head.new_witness(bv[i]);
    std::thread ta, tb, td;
    if (threaded) {
        ta = std::thread(build_a);
        tb = std::thread(build_b);
        td = std::thread(build_d);
    } else {            // one after another: every segment starts where the previous one actually ended
        build_a();
        seg_b.wit_base = seg_a.next_wit();
        build_b();
        seg_c.wit_base = seg_b.next_wit();
        seg_d.wit_base = seg_c.wit_base + nn + nn * (n + 1);
    }
    // matrix_mul (constraints.rs:78-99): part 0 on this thread, the others on their own
    std::vector<std::exception_ptr> err_mm(mm_parts);
    auto build_mm = [&](size_t t) {
        try {
            Circuit &sg = *seg_mm[t];
            if (t == 0)
                for (size_t i = 0; i < nn; i++) sg.new_witness(Fr::zero());     // pre-allocated, never constrained (:84)
            for (size_t i = n * t / mm_parts; i < n * (t + 1) / mm_parts; i++)
                for (size_t j = 0; j < n; j++) {
                    sg.new_witness(Fr::zero());                  // the sum's seed (:87); the running sum itself is symbolic
                    for (size_t k = 0; k < n; k++) {
                        const Lc &ij = ma[i * n + k], &jk = mb[k * n + j];
                        Lc product = sg.mul(ij, jk);            // `*`: product witness + constraint (:91)
                        sg.mul_equals(ij, jk, product);         // second constraint on the same triple (:93)
                    }
                }
        } catch (...) { err_mm[t] = std::current_exception(); g_terms = true; }
    };
    std::vector<std::thread> tmm;
    for (size_t t = 1; t < mm_parts; t++) tmm.emplace_back(build_mm, t);
    build_mm(0);
    for (auto &t : tmm) t.join();
    lap("matrix_mul");
    for (const std::exception_ptr &e : err_mm)
        if (e && !err_c) err_c = e;
    if (threaded) { ta.join(); tb.join(); td.join(); }
    else build_d();
    for (const std::exception_ptr &e : {err_a, err_b, err_c, err_d})
        if (e) std::rethrow_exception(e);
    {       // every witness-allocating segment must start where the one before it ended
        std::vector<const Circuit *> order;
        for (const SpongeChunk &k : ch_a) order.push_back(k.seg);
        for (const SpongeChunk &k : ch_b) order.push_back(k.seg);
        for (const Circuit *sg : seg_mm) order.push_back(sg);
        for (const SpongeChunk &k : ch_d) order.push_back(k.seg);
        for (size_t i = 0; i + 1 < order.size(); i++)
            if (order[i + 1]->wit_base != order[i]->next_wit()) throw std::logic_error("matrix circuit: segment offsets do not line up");
    }
    head.instance[1] = hash_a;
    head.instance[2] = hash_b;
    in_a.val = hash_a;
    in_b.val = hash_b;
    seg_mid.enforce_equal(ha, in_a);
    seg_mid.enforce_equal(hb, in_b);
    Lc in_c = head.new_input(hash_c);
    seg_tail.enforce_equal(hc, in_c);
}

int zkg16_circuit_matrix(size_t n, const uint64_t *a, const uint64_t *b, zkg16_circuit **out) {
    if (!out || !a || !b || n == 0 || n > 1024) return ZKG16_ERR_BAD_ARG;
    const char *env = getenv("ZKG16_SYNTH_THREADS");
    bool threaded = n >= 2 && !(env && env[0] == '0');
    for (;;) {
        auto c = new (std::nothrow) zkg16_circuit();
        if (!c) return ZKG16_ERR_OOM;
        try {
            build_matrix_circuit(c, n, a, b, threaded);
            *out = c;
            return ZKG16_OK;
        } catch (const std::bad_alloc &) {
            g_terms = true;
            delete c;
            return ZKG16_ERR_OOM;
        } catch (const std::exception &) {
            g_terms = true;
            delete c;
            if (!threaded) return ZKG16_ERR_UNSUPPORTED;      // a template replay that did not line up: a bug, never a property of the input
            if (getenv("ZKG16_SYNTH_STRICT")) return ZKG16_ERR_UNSUPPORTED;      // tests: the concurrent build itself must succeed
            threaded = false;                                  // predicted segment offsets were off: build in order instead
        }
    }
}

// Releasing a large circuit is ~20 ms of unmapping (160 MB in three arenas at n = 32): it is handed to a detached thread so
// that the caller's request path does not wait for it.
// Only the full assignment z = instance || witness of the MatrixCircuit for these inputs (z: (4 + 2 n^2 + ...) x 4 limbs as
// zkg16_circuit_export would give it; n_assign must equal that circuit's variable count): the R1CS matrices of this circuit
// depend on n alone, so a server keeps them (and their device copy) per size and asks only for this per request.
// The sponge's witnesses without any gadget objects: the values the PoseidonSpongeVar allocates, in its order — per
// permutation and round the five products x^2, x^4, x^8, x^16, x^17 of every S-box input that is not a constant (the only
// constant one is the capacity element in the first round of the first permutation).  Returns the hash; appends to out.
static Fr poseidon_hash_witness_values(const Fr *elems, size_t count, Fr *out, size_t &n_out) {
    const PoseidonParams &p = pparams();
    Fr st[3] = {Fr::zero(), Fr::zero(), Fr::zero()};
    size_t pos = 0;
    bool first = true;
    auto permute = [&]() {
        for (int r = 0; r < P_ROUNDS; r++) {
            for (int i = 0; i < 3; i++) st[i] = fp_add(st[i], p.ark[r][i]);
            const bool full = r < P_HALF || r >= P_HALF + POSEIDON_PARTIAL;
            for (int i = 0; i < (full ? 3 : 1); i++) {
                const Fr x = st[i];
                const Fr x2 = fp_sqr(x), x4 = fp_sqr(x2), x8 = fp_sqr(x4), x16 = fp_sqr(x8), x17 = fp_mul(x16, x);
                if (!(first && r == 0 && i == 0)) {
                    out[n_out] = x2; out[n_out + 1] = x4; out[n_out + 2] = x8; out[n_out + 3] = x16; out[n_out + 4] = x17;
                    n_out += 5;
                }
                st[i] = x17;
            }
            Fr nst[3];
            for (int i = 0; i < 3; i++) {
                Fr acc = Fr::zero();
                for (int j = 0; j < 3; j++) acc = fp_add(acc, fp_mul(st[j], p.mds[i][j]));
                nst[i] = acc;
            }
            for (int i = 0; i < 3; i++) st[i] = nst[i];
        }
        first = false;
    };
    for (size_t idx = 0; idx < count; idx++) {
        if (pos == POSEIDON_RATE) { permute(); pos = 0; }
        st[POSEIDON_CAP + pos] = fp_add(st[POSEIDON_CAP + pos], elems[idx]);
        pos++;
    }
    permute();
    return st[POSEIDON_CAP];
}

int zkg16_circuit_matrix_witness(size_t n, const uint64_t *a, const uint64_t *b, uint64_t *z, size_t n_assign) {
    if (!a || !b || !z || n < 2 || n > 1024) return ZKG16_ERR_BAD_ARG;
    const size_t nn = n * n, hw = poseidon_hash_witnesses(nn), ni = 4;
    // [1, hash_a, hash_b, hash_c] | a | b | hash_a gadget | hash_b gadget | n^2 zeros | per (i, j): 0, n products | hash_c gadget
    const size_t off_a = ni, off_b = off_a + nn, off_ha = off_b + nn, off_hb = off_ha + hw, off_mc = off_hb + hw, off_mm = off_mc + nn,
                 off_hc = off_mm + nn * (n + 1), total = off_hc + hw;
    if (total != n_assign) return ZKG16_ERR_BAD_ARG;
    try {
        Fr *Z = reinterpret_cast<Fr *>(z);
        std::vector<Fr> cv(nn);
        for (size_t i = 0; i < nn; i++) { Z[off_a + i] = fr_from_u64(a[i]); Z[off_b + i] = fr_from_u64(b[i]); }
        const Fr *av = Z + off_a, *bv = Z + off_b;
        Fr ha, hb, hc;
        size_t na = 0, nb = 0, nc = 0;
        const char *env = getenv("ZKG16_SYNTH_THREADS");
        const bool threaded = !(env && env[0] == '0');
        auto do_a = [&]() { ha = poseidon_hash_witness_values(av, nn, Z + off_ha, na); };
        auto do_b = [&]() { hb = poseidon_hash_witness_values(bv, nn, Z + off_hb, nb); };
        std::thread ta, tb;
        if (threaded) { ta = std::thread(do_a); tb = std::thread(do_b); } else { do_a(); do_b(); }
        // matrix_mul: the pre-allocated entries and the sums' seeds are zero, the products are a_ik * b_kj in allocation order
        for (size_t i = 0; i < nn; i++) Z[off_mc + i] = Fr::zero();
        // the products sit on the critical path in front of the (sequential) sponge over C: rows of the product in parallel
        auto rows = [&](size_t i_lo, size_t i_hi) {
            for (size_t i = i_lo; i < i_hi; i++)
                for (size_t j = 0; j < n; j++) {
                    Fr *blk = Z + off_mm + (i * n + j) * (n + 1);
                    blk[0] = Fr::zero();
                    Fr sum = Fr::zero();
                    for (size_t k = 0; k < n; k++) {
                        blk[1 + k] = fp_mul(av[i * n + k], bv[k * n + j]);
                        sum = fp_add(sum, blk[1 + k]);
                    }
                    cv[i * n + j] = sum;
                }
        };
        const size_t helpers = (threaded && n >= 32) ? 5 : 0;
        std::vector<std::thread> mm;
        for (size_t t = 0; t < helpers; t++) mm.emplace_back(rows, n * t / (helpers + 1), n * (t + 1) / (helpers + 1));
        rows(n * helpers / (helpers + 1), n);
        for (auto &t : mm) t.join();
        hc = poseidon_hash_witness_values(cv.data(), nn, Z + off_hc, nc);
        if (threaded) { ta.join(); tb.join(); }
        if (na != hw || nb != hw || nc != hw) return ZKG16_ERR_UNSUPPORTED;      // the closed form of the sponge's witness count is off: a bug
        Z[0] = Fr::one();
        Z[1] = ha; Z[2] = hb; Z[3] = hc;
    } catch (const std::bad_alloc &) {
        return ZKG16_ERR_OOM;
    } catch (const std::exception &) {
        return ZKG16_ERR_UNSUPPORTED;
    }
    return ZKG16_OK;
}

void zkg16_circuit_free(zkg16_circuit *c) {
    if (!c) return;
    if (c->pooled_storage && c->segs.size() == (size_t)PRIME_PARTS)
        for (int p = 0; p < PRIME_PARTS; p++) prime_store_give(*c->segs[p], p);
    size_t terms = 0;
    for (const auto &sg : c->segs) terms += sg->rows[0].t.size() + sg->rows[1].t.size() + sg->rows[2].t.size();
    if (terms < (1u << 16)) {
        delete c;
        return;
    }
    try {
        std::thread([c] { delete c; }).detach();
    } catch (...) {
        delete c;
    }
}

int zkg16_circuit_dims(const zkg16_circuit *c, size_t *num_instance, size_t *num_witness, size_t *num_constraints, size_t nnz[3]) {
    if (!c || c->segs.empty()) return ZKG16_ERR_BAD_ARG;
    if (num_instance) *num_instance = c->head().instance.size();
    if (num_witness) *num_witness = c->num_witness();
    size_t rows = 0, k[3] = {0, 0, 0};
    for (const auto &sg : c->segs) {
        rows += sg->rows[0].size();
        for (int m = 0; m < 3; m++) k[m] += sg->rows[m].t.size();
    }
    if (num_constraints) *num_constraints = rows;
    if (nnz)
        for (int m = 0; m < 3; m++) nnz[m] = k[m];
    return ZKG16_OK;
}

// the instance assignment without the leading one (= the public inputs a verifier is given): (num_instance - 1) x 4 limbs
int zkg16_circuit_public_inputs(const zkg16_circuit *c, uint64_t *out, size_t cap) {
    if (!c || c->segs.empty() || !out) return ZKG16_ERR_BAD_ARG;
    const std::vector<Fr> &inst = c->head().instance;
    if (cap + 1 < inst.size()) return ZKG16_ERR_BAD_ARG;
    if (inst.size() > 1) memcpy(out, inst.data() + 1, (inst.size() - 1) * sizeof(Fr));
    return ZKG16_OK;
}

int zkg16_circuit_is_satisfied(const zkg16_circuit *c) {
    if (!c || c->segs.empty()) return 0;
    const std::vector<Fr> w = c->flat_witness();
    for (const auto &sg : c->segs)
        if (!sg->satisfied(c->head().instance, w)) return 0;
    return 1;
}

// ConstraintMatrices as CSR (caller-allocated: row_ptr[m] has num_constraints + 1 entries, col/coeff nnz[m]) and the
// full assignment z = instance || witness (Montgomery limbs).
int zkg16_circuit_export(const zkg16_circuit *c, uint64_t *const row_ptr[3], uint32_t *const col[3], uint64_t *const coeff[3], uint64_t *z) {
    if (!c || c->segs.empty() || !row_ptr || !col || !coeff || !z) return ZKG16_ERR_BAD_ARG;
    const size_t ni = c->head().instance.size();
    // (matrix, segment) pieces with their row / term offsets; the copies are plain streaming work (3.7 GB at 128x128), cut into
    // tasks of ~2^18 terms and spread over the host threads (ZKG16_SYNTH_THREADS=0: this thread only)
    struct Task { int m; const Circuit::Rows *r; size_t row0, k0, r_lo, r_hi; };
    std::vector<Task> tasks;
    size_t witness_terms = 0;
    for (int m = 0; m < 3; m++) {
        size_t row = 0, k = 0;
        row_ptr[m][0] = 0;
        for (const auto &sg : c->segs) {
            const Circuit::Rows &r = sg->rows[m];
            const size_t nrows = r.size();
            size_t lo = 0;
            while (lo < nrows) {
                size_t hi = lo;
                while (hi < nrows && r.ptr[hi] - r.ptr[lo] < ((size_t)1 << 18)) hi++;
                if (hi == lo) hi = lo + 1;
                tasks.push_back(Task{m, &r, row, k, lo, hi});
                lo = hi;
            }
            row += nrows;
            k += r.t.size();
        }
        witness_terms += k;
    }
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t id = next.fetch_add(1);
            if (id >= tasks.size()) break;
            const Task &t = tasks[id];
            const Circuit::Rows &r = *t.r;
            for (size_t i = t.r_lo; i < t.r_hi; i++) row_ptr[t.m][t.row0 + i + 1] = t.k0 + r.ptr[i + 1];
            // instance columns first, then witnesses (ids are already ordered that way within a row)
            for (size_t j = r.ptr[t.r_lo]; j < r.ptr[t.r_hi]; j++) {
                const Term &x = r.t[j];
                col[t.m][t.k0 + j] = (x.v & WIT) ? (uint32_t)(ni + (x.v & ~WIT)) : x.v;
                memcpy(coeff[t.m] + 4 * (t.k0 + j), x.c.l, 32);
            }
        }
    };
    const char *env = getenv("ZKG16_SYNTH_THREADS");
    unsigned nthreads = (env && env[0] == '0') ? 1u : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (witness_terms < ((size_t)1 << 20)) nthreads = 1;
    try {
        std::vector<std::thread> pool;
        for (unsigned i = 1; i < nthreads; i++) pool.emplace_back(work);
        work();
        for (auto &th : pool) th.join();
    } catch (const std::exception &) {
        return ZKG16_ERR_OOM;
    }
    memcpy(z, c->head().instance.data(), ni * 32);
    for (const auto &sg : c->segs)
        if (!sg->witness.empty()) memcpy(z + 4 * (ni + sg->wit_base), sg->witness.data(), sg->witness.size() * 32);
    return ZKG16_OK;
}

// native sponge hash of n Montgomery Fr elements (hasher.rs:17-27)
int zkg16_poseidon_hash(const uint64_t *elems, size_t n, uint64_t out[4]) {
    if ((!elems && n) || !out) return ZKG16_ERR_BAD_ARG;
    const Fr h = poseidon_hash_native(reinterpret_cast<const Fr *>(elems), n);
    memcpy(out, h.l, 32);
    return ZKG16_OK;
}

}  // extern "C"
