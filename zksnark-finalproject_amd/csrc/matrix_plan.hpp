// The MatrixCircuit's R1CS as a PLAN: what a kernel (matrix_r1cs.hip) or a plain loop (matrix_plan_instantiate_host, the
// test reference) needs to write the three CSR matrices of size n without synthesising 10.7 M constraints one by one.
//
// /root/reference/src/arkworks/matrix_proof_of_work/constraints.rs:101-128 produces, in this order: the gadget rows of the
// sponge over a, of the sponge over b, two `enforce_equal` rows, the 2 n^3 rows of matrix_mul (:78-99), the gadget rows of the
// sponge over c, one `enforce_equal` row.  The sponge rows are, per Poseidon permutation, a copy of ONE template per
// permutation class — first (the capacity lane is still a constant), second, any later one, and the last one of an odd-sized
// input (a single absorbed element) — with its variables renamed: circuits.hip already builds and replays such templates on the
// host (permute_gadget); here they are exported with their slots described symbolically, so that the renaming is arithmetic.
#pragma once
#include <stdint.h>

#include <vector>

#include "ff.cuh"

namespace zk {

struct MatrixPlanSlot {      // one template id below n_slots: which real column it stands for in permutation p
    uint32_t kind;           // 0: the constant One (column 0); 1: a variable of an absorbed element; 2: a witness of permutation p - 1
    uint32_t a;              // kind 1: element 2 p + a (a = 0, 1); kind 2: offset from that permutation's first witness
    uint32_t b;              // kind 1: offset inside the element's variable block
};

struct MatrixPlanTemplate {
    uint32_t n_slots = 0, n_new = 0, n_rows = 0;
    std::vector<MatrixPlanSlot> slots;               // n_slots
    std::vector<uint32_t> ptr[3];                    // n_rows + 1 per matrix
    std::vector<uint32_t> id[3];                     // template ids: < n_slots = slot, else n_slots + (new witness index)
    std::vector<Fr> coeff[3];
};

struct MatrixPlanHash {
    // permutation p uses tpl[cls(p)]: 0 = first, 1 = second, 2 = later, 3 = the last one when it absorbs a single element
    MatrixPlanTemplate tpl[4];
    bool has[4] = {false, false, false, false};
    uint32_t perms = 0;
    bool odd_tail = false;              // the last permutation absorbs one element
    uint64_t elem_col0 = 0;             // column of the first variable of element 0
    uint32_t elem_vars = 1;             // variables per absorbed element (1 for a and b; n + 1 for c: the sum's seed and the n products)
    uint64_t wit_col0 = 0;              // column of the gadget's first witness
    uint64_t row0 = 0, nnz0[3] = {0, 0, 0};      // where this hash's rows / non-zeros start in the whole system
    int cls(uint32_t p) const { return (odd_tail && p + 1 == perms && p > 0) ? 3 : p < 2 ? (int)p : 2; }
};

struct MatrixPlanRow { std::vector<uint32_t> col; std::vector<Fr> coeff; };      // the three enforce_equal rows, whole

struct MatrixPlan {
    size_t n = 0, nn = 0, num_instance = 4, num_witness = 0, num_constraints = 0;
    uint64_t nnz[3] = {0, 0, 0};
    MatrixPlanHash hash[3];                          // a, b, c
    uint64_t eq_row[3] = {0, 0, 0};                  // row indices of the three equality rows (a, b, c)
    uint64_t eq_nnz0[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};      // [which][matrix]
    MatrixPlanRow eq[3][3];                          // [which][matrix]
    uint64_t mm_row0 = 0, mm_nnz0[3] = {0, 0, 0};    // matrix_mul: 2 n^3 rows with one non-zero per matrix
    uint64_t col_a0 = 0, col_b0 = 0, col_prod0 = 0;  // columns of a[0], b[0] and of the (i, j) = (0, 0) block [seed, n products]
};

// rows / non-zeros of the first q permutations of a hash (prefix over the permutation classes)
inline void matrix_plan_prefix(const MatrixPlanHash &h, uint32_t q, uint64_t &rows, uint64_t nnz[3]) {
    rows = 0;
    nnz[0] = nnz[1] = nnz[2] = 0;
    auto add = [&](int c, uint64_t count) {
        if (!count) return;
        rows += count * h.tpl[c].n_rows;
        for (int m = 0; m < 3; m++) nnz[m] += count * h.tpl[c].ptr[m][h.tpl[c].n_rows];
    };
    const uint32_t tail = h.odd_tail && h.perms > 1 ? 1u : 0u;          // class-3 permutations (at the very end)
    const uint32_t body_end = h.perms - tail;                            // permutations [0, body_end) are classes 0, 1, 2
    const uint32_t qb = q < body_end ? q : body_end;
    add(0, qb > 0 ? 1 : 0);
    add(1, qb > 1 ? 1 : 0);
    add(2, qb > 2 ? qb - 2 : 0);
    add(3, q > body_end ? 1 : 0);
}

}  // namespace zk
