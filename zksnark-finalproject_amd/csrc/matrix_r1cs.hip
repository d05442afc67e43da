// The MatrixCircuit's R1CS written ON THE DEVICE (the step in front of the path, "in C++ / on GPU").
//
// The reference's `Groth16::setup` and `Groth16::prove` each synthesise the circuit (matrix_proof.rs:129, :139-140): 10.7 M
// constraints at 128x128, 86.6 M non-zeros, 3.7 GB of CSR arrays that round 2 built on the host (0.8 s on 16 threads) and
// uploaded (0.2 s of PCIe).  The system has almost no information in it: matrix_mul's 2 n^3 rows are (a_ik, b_kj, product)
// triples in closed form (constraints.rs:78-99), and the three sponges' rows are, per Poseidon permutation, a copy of one of
// four templates with renamed variables (circuits.hip / matrix_plan.hpp).  Here the host builds the plan (2 ms: the gadget code
// run on nine elements), uploads the templates (a few hundred KB) and two kernels write the arrays where the witness map and
// the setup read them: HBM streaming at the arrays' size, nothing else crosses PCIe.
#include "common.hpp"
#include "matrix_plan.hpp"

using namespace zk;

namespace {

struct DevTpl {
    const uint32_t *ptr[3], *id[3];
    const Fr *coeff[3];
    const MatrixPlanSlot *slots;
    uint32_t n_slots, n_new, n_rows, nnz[3];
};
struct DevHash {
    DevTpl tpl[4];
    uint32_t perms, odd_tail, elem_vars, pad;
    uint64_t elem_col0, wit_col0, row0, nnz0[3];
};
struct GenArgs {
    DevHash hash[3];
    uint64_t *rp[3];
    uint32_t *col[3];
    Fr *cf[3];
    uint64_t mm_row0, mm_nnz0[3], col_a0, col_b0, col_prod0;
    uint32_t n, rate;
};

__device__ __forceinline__ int cls_of(const DevHash &h, uint32_t p) { return (h.odd_tail && p + 1 == h.perms && p > 0) ? 3 : p < 2 ? (int)p : 2; }
// rows / non-zeros / witnesses of the first q permutations (matrix_plan_prefix)
__device__ __forceinline__ void prefix(const DevHash &h, uint32_t q, uint64_t &rows, uint64_t nnz[3], uint64_t &wits) {
    rows = wits = 0;
    nnz[0] = nnz[1] = nnz[2] = 0;
    const uint32_t tail = (h.odd_tail && h.perms > 1) ? 1u : 0u, body_end = h.perms - tail, qb = q < body_end ? q : body_end;
    const uint64_t cnt[4] = {qb > 0 ? 1u : 0u, qb > 1 ? 1u : 0u, qb > 2 ? (uint64_t)(qb - 2) : 0u, q > body_end ? 1u : 0u};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        rows += cnt[c] * h.tpl[c].n_rows;
        wits += cnt[c] * h.tpl[c].n_new;
        for (int m = 0; m < 3; m++) nnz[m] += cnt[c] * h.tpl[c].nnz[m];
    }
}

__device__ __forceinline__ void st_fr(Fr *p, const Fr &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
__device__ __forceinline__ Fr ld_fr(const Fr *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 lo = q[0], hi = q[1];
    Fr v;
    v.l[0] = lo.x; v.l[1] = lo.y; v.l[2] = lo.z; v.l[3] = lo.w; v.l[4] = hi.x; v.l[5] = hi.y; v.l[6] = hi.z; v.l[7] = hi.w;
    return v;
}

// one workgroup per (permutation, hash x matrix): the template's non-zeros with renamed columns, and its row pointers
__global__ void __launch_bounds__(256) r1cs_sponge_rows_kernel(GenArgs g) {
    const int h = blockIdx.y / 3, m = blockIdx.y % 3;
    const DevHash &H = g.hash[h];
    const uint32_t p = blockIdx.x;
    if (p >= H.perms) return;
    const DevTpl &T = H.tpl[cls_of(H, p)];
    uint64_t r0, k0[3], wits;
    prefix(H, p, r0, k0, wits);
    const uint64_t w0 = H.wit_col0 + wits;
    uint64_t w0_prev = 0;
    if (p > 0) {
        uint64_t r1, k1[3], w1;
        prefix(H, p - 1, r1, k1, w1);
        w0_prev = H.wit_col0 + w1;
    }
    const uint64_t kb = H.nnz0[m] + k0[m];
    uint64_t *rp = g.rp[m] + H.row0 + r0;
    for (uint32_t r = threadIdx.x; r < T.n_rows; r += blockDim.x) rp[r + 1] = kb + T.ptr[m][r + 1];
    uint32_t *col = g.col[m] + kb;
    Fr *cf = g.cf[m] + kb;
    for (uint32_t k = threadIdx.x; k < T.nnz[m]; k += blockDim.x) {
        const uint32_t id = T.id[m][k];
        uint32_t c;
        if (id >= T.n_slots) {
            c = (uint32_t)(w0 + (id - T.n_slots));
        } else {
            const MatrixPlanSlot sl = T.slots[id];
            c = sl.kind == 0 ? 0u : sl.kind == 1 ? (uint32_t)(H.elem_col0 + ((uint64_t)g.rate * p + sl.a) * H.elem_vars + sl.b) : (uint32_t)(w0_prev + sl.a);
        }
        col[k] = c;
        st_fr(cf + k, ld_fr(T.coeff[m] + k));
    }
}

// matrix_mul: row t of the 2 n^3 is (a_ik) * (b_kj) = product_ijk, twice per (i, j, k) (`*` and `mul_equals`: constraints.rs:91, :93)
__global__ void __launch_bounds__(256) r1cs_matmul_rows_kernel(GenArgs g, uint64_t total) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint64_t n = g.n, pr = t >> 1, cell = pr / n, k = pr % n, i = cell / n, j = cell % n;
    const uint32_t c[3] = {(uint32_t)(g.col_a0 + i * n + k), (uint32_t)(g.col_b0 + k * n + j), (uint32_t)(g.col_prod0 + cell * (n + 1) + 1 + k)};
    const Fr one = Fr::one();
#pragma unroll
    for (int m = 0; m < 3; m++) {
        g.col[m][g.mm_nnz0[m] + t] = c[m];
        st_fr(g.cf[m] + g.mm_nnz0[m] + t, one);
        g.rp[m][g.mm_row0 + t + 1] = g.mm_nnz0[m] + t + 1;
    }
}

}  // namespace

namespace zk {

// the R1csDev of the MatrixCircuit of size n, arrays written by the kernels above on ctx->stream (synchronised before return)
std::shared_ptr<R1csDev> matrix_r1cs_on_device(zkg16_ctx *ctx, size_t n, int *status) {
    *status = ZKG16_OK;
    MatrixPlan P;
    if (!matrix_plan_build(n, P)) { *status = ZKG16_ERR_UNSUPPORTED; return nullptr; }
    const size_t nc = P.num_constraints, nvars = P.num_instance + P.num_witness;
    int log_n = 0;
    while (((size_t)1 << log_n) < nc + P.num_instance) log_n++;
    if (log_n > 28) { *status = ZKG16_ERR_DOMAIN_TOO_LARGE; return nullptr; }
    auto r = std::make_shared<R1csDev>();
    r->num_instance = P.num_instance;
    r->num_constraints = nc;
    r->num_variables = nvars;
    r->log_n = log_n;
    for (int m = 0; m < 3; m++) {
        r->nnz[m] = (size_t)P.nnz[m];
        r->rp[m].alloc((nc + 1) * sizeof(uint64_t));
        r->col[m].alloc((size_t)P.nnz[m] * sizeof(uint32_t));
        r->cf[m].alloc((size_t)P.nnz[m] * sizeof(Fr));
    }
    // ---- the templates as one blob: [slots | per matrix: ptr, id, coeff] per (hash, class)
    std::vector<unsigned char> blob;
    auto put = [&](const void *src, size_t bytes) {
        const size_t off = (blob.size() + 31) & ~(size_t)31;
        blob.resize(off + bytes);
        if (bytes) memcpy(blob.data() + off, src, bytes);
        return off;
    };
    struct Off { size_t slots, ptr[3], id[3], coeff[3]; } off[3][4];
    for (int h = 0; h < 3; h++)
        for (int c = 0; c < 4; c++) {
            if (!P.hash[h].has[c]) continue;
            const MatrixPlanTemplate &T = P.hash[h].tpl[c];
            off[h][c].slots = put(T.slots.data(), T.slots.size() * sizeof(MatrixPlanSlot));
            for (int m = 0; m < 3; m++) {
                off[h][c].ptr[m] = put(T.ptr[m].data(), T.ptr[m].size() * sizeof(uint32_t));
                off[h][c].id[m] = put(T.id[m].data(), T.id[m].size() * sizeof(uint32_t));
                off[h][c].coeff[m] = put(T.coeff[m].data(), T.coeff[m].size() * sizeof(Fr));
            }
        }
    DevBuf d_blob(blob.size() + 64);
    ZK_HIP(hipMemcpyAsync(d_blob.p, blob.data(), blob.size(), hipMemcpyHostToDevice, ctx->stream));
    const unsigned char *base = d_blob.as<unsigned char>();
    GenArgs g;
    memset(&g, 0, sizeof g);
    uint32_t max_perms = 0;
    for (int h = 0; h < 3; h++) {
        const MatrixPlanHash &H = P.hash[h];
        DevHash &D = g.hash[h];
        D.perms = H.perms; D.odd_tail = H.odd_tail ? 1 : 0; D.elem_vars = H.elem_vars;
        D.elem_col0 = H.elem_col0; D.wit_col0 = H.wit_col0; D.row0 = H.row0;
        for (int m = 0; m < 3; m++) D.nnz0[m] = H.nnz0[m];
        max_perms = H.perms > max_perms ? H.perms : max_perms;
        for (int c = 0; c < 4; c++) {
            if (!H.has[c]) continue;
            const MatrixPlanTemplate &T = H.tpl[c];
            DevTpl &t = D.tpl[c];
            t.slots = reinterpret_cast<const MatrixPlanSlot *>(base + off[h][c].slots);
            t.n_slots = T.n_slots; t.n_new = T.n_new; t.n_rows = T.n_rows;
            for (int m = 0; m < 3; m++) {
                t.ptr[m] = reinterpret_cast<const uint32_t *>(base + off[h][c].ptr[m]);
                t.id[m] = reinterpret_cast<const uint32_t *>(base + off[h][c].id[m]);
                t.coeff[m] = reinterpret_cast<const Fr *>(base + off[h][c].coeff[m]);
                t.nnz[m] = (uint32_t)T.id[m].size();
            }
        }
    }
    for (int m = 0; m < 3; m++) {
        g.rp[m] = r->rp[m].as<uint64_t>();
        g.col[m] = r->col[m].as<uint32_t>();
        g.cf[m] = r->cf[m].as<Fr>();
        g.mm_nnz0[m] = P.mm_nnz0[m];
    }
    g.mm_row0 = P.mm_row0; g.col_a0 = P.col_a0; g.col_b0 = P.col_b0; g.col_prod0 = P.col_prod0;
    g.n = (uint32_t)n; g.rate = 2;
    static_assert(sizeof(GenArgs) <= 3584, "kernel arguments");
    const uint64_t zero = 0;
    for (int m = 0; m < 3; m++) ZK_HIP(hipMemcpyAsync(g.rp[m], &zero, sizeof zero, hipMemcpyHostToDevice, ctx->stream));
    {
        ScopedKernelTimer kt(ctx, "r1cs_sponge_rows_kernel", (double)(P.nnz[0] + P.nnz[1] + P.nnz[2]));
        hipLaunchKernelGGL(r1cs_sponge_rows_kernel, dim3(max_perms, 9), dim3(256), 0, ctx->stream, g);
    }
    {
        const uint64_t total = 2 * (uint64_t)P.nn * n;
        ScopedKernelTimer kt(ctx, "r1cs_matmul_rows_kernel", (double)total);
        hipLaunchKernelGGL(r1cs_matmul_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, g, total);
    }
    ZK_HIP(hipGetLastError());
    // the three enforce_equal rows: whole rows from the plan
    uint64_t ends[3][3];
    for (int w = 0; w < 3; w++)
        for (int m = 0; m < 3; m++) {
            const MatrixPlanRow &R = P.eq[w][m];
            ends[w][m] = P.eq_nnz0[w][m] + R.col.size();
            ZK_HIP(hipMemcpyAsync(g.rp[m] + P.eq_row[w] + 1, &ends[w][m], sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
            if (R.col.empty()) continue;
            ZK_HIP(hipMemcpyAsync(g.col[m] + P.eq_nnz0[w][m], R.col.data(), R.col.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            ZK_HIP(hipMemcpyAsync(g.cf[m] + P.eq_nnz0[w][m], R.coeff.data(), R.coeff.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        }
    ZK_HIP(hipStreamSynchronize(ctx->stream));          // the plan, the blob and `end` are read by the copies above
    return r;
}

}  // namespace zk
