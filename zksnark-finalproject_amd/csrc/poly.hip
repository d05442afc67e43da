// R1CS -> QAP witness map on device (everything of ark-groth16 0.4 `LibsnarkReduction::witness_map_from_matrices`
// — src/r1cs_to_qap.rs, SURVEY.md A.4 — that is not an NTT): the three CSR SpMVs <A_i,z>, <B_i,z>, <C_i,z>,
// the pointwise (a*b - c) / Z on the coset, and the Montgomery -> canonical conversion that feeds the MSMs.
// Called from /root/reference/src/arkworks/backend/matrix_proof.rs:139-140 via Groth16::prove.
// All kernels are HBM-streaming (32 B per Fr, 16-B vector loads); SpMV gathers z through L2/MALL.
#include "common.hpp"

namespace zk {

__device__ __forceinline__ Fr gld_fr(const Fr *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fr v;
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
    v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
    return v;
}
__device__ __forceinline__ void gst_fr(Fr *p, const Fr &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

struct SpmvArgs {
    const uint64_t *rp[3];
    const uint32_t *col[3];
    const Fr *cf[3];
    Fr *out[3];
    const Fr *z;
    size_t nc, num_instance, n;   // n = domain size (outputs are zero-padded to n)
};

// One thread per (matrix, row): the reference's circuits have short rows (matmul rows: 1 nnz per side;
// Poseidon rows: a few tens), so a row per lane keeps all 64 lanes busy.  Rows >= nc are the padding:
// a[nc + i] = z[i] for i < num_instance (the "input consistency" rows of LibsnarkReduction), else 0.
__global__ void __launch_bounds__(256) spmv_kernel(SpmvArgs a) {
    const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (row >= a.n) return;
    Fr acc = Fr::zero();
    if (row < a.nc) {
        const uint64_t lo = a.rp[m][row], hi = a.rp[m][row + 1];
        for (uint64_t k = lo; k < hi; k++) {
            Fr c = gld_fr(a.cf[m] + k);
            Fr v = gld_fr(a.z + a.col[m][k]);
            acc = fp_add(acc, fp_mul(c, v));
        }
    } else if (m == 0 && row < a.nc + a.num_instance) {
        acc = gld_fr(a.z + (row - a.nc));
    }
    gst_fr(a.out[m] + row, acc);
}

// ab[i] = (a[i]*b[i] - c[i]) * zinv
__global__ void __launch_bounds__(256) pointwise_h_kernel(Fr *a, const Fr *b, const Fr *c, Fr zinv, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = fp_mul(gld_fr(a + i), gld_fr(b + i));
    x = fp_sub(x, gld_fr(c + i));
    gst_fr(a + i, fp_mul(x, zinv));
}

__global__ void __launch_bounds__(256) fr_from_mont_kernel(const Fr *in, Fr *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    gst_fr(out + i, fp_from_mont(gld_fr(in + i)));
}

void spmv_run(zkg16_ctx *ctx, const R1csDev &m, const Fr *z, Fr *a, Fr *b, Fr *c) {
    SpmvArgs s;
    for (int i = 0; i < 3; i++) {
        s.rp[i] = m.rp[i].as<uint64_t>();
        s.col[i] = m.col[i].as<uint32_t>();
        s.cf[i] = m.cf[i].as<Fr>();
    }
    s.out[0] = a; s.out[1] = b; s.out[2] = c;
    s.z = z;
    s.nc = m.num_constraints;
    s.num_instance = m.num_instance;
    s.n = (size_t)1 << m.log_n;
    const unsigned grid = (unsigned)((s.n + 255) / 256);
    ScopedKernelTimer kt(ctx, "spmv_kernel", (double)(m.nnz[0] + m.nnz[1] + m.nnz[2]));
    hipLaunchKernelGGL(spmv_kernel, dim3(grid, 3), dim3(256), 0, ctx->stream, s);
    ZK_HIP(hipGetLastError());
}

void pointwise_h_run(zkg16_ctx *ctx, Fr *ab_a, const Fr *b, const Fr *c, const Fr &zinv, size_t n) {
    ScopedKernelTimer kt(ctx, "pointwise_h_kernel", (double)n);
    hipLaunchKernelGGL(pointwise_h_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ab_a, b, c, zinv, n);
    ZK_HIP(hipGetLastError());
}

void fr_from_mont_run(zkg16_ctx *ctx, const Fr *in, Fr *out, size_t n) {
    if (n == 0) return;
    ScopedKernelTimer kt(ctx, "fr_from_mont_kernel", (double)n);
    hipLaunchKernelGGL(fr_from_mont_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, in, out, n);
    ZK_HIP(hipGetLastError());
}

// h = coset_ifft( (coset_fft(ifft a) * coset_fft(ifft b) - coset_fft(ifft c)) / Z ), N Montgomery coefficients.
// The transforms ping-pong between each vector and the one scratch buffer (ifft: x -> tmp, coset fft: tmp -> x), and the
// point-wise (ab - c)/Z rides on the load of the seventh transform: no device-to-device copy and no separate point-wise
// pass (round 1 had both: 8 extra passes over N x 32 B per proof).  Result pointer = ctx->poly[3].
void witness_map_run(zkg16_ctx *ctx, const R1csDev &m, const Fr *z, Fr **h_out) {
    const size_t n = (size_t)1 << m.log_n;
    for (int i = 0; i < 4; i++) ctx->poly[i].ensure(n * sizeof(Fr));
    Fr *a = ctx->poly[0].as<Fr>(), *b = ctx->poly[1].as<Fr>(), *c = ctx->poly[2].as<Fr>(), *tmp = ctx->poly[3].as<Fr>();
    spmv_run(ctx, m, z, a, b, c);
    ntt_run(ctx, a, tmp, m.log_n, true, false);
    ntt_run(ctx, tmp, a, m.log_n, false, true);
    ntt_run(ctx, b, tmp, m.log_n, true, false);
    ntt_run(ctx, tmp, b, m.log_n, false, true);
    ntt_run(ctx, c, tmp, m.log_n, true, false);
    ntt_run(ctx, tmp, c, m.log_n, false, true);
    NttTables *t = ntt_get_tables(ctx, m.log_n);
    if (ctx->opt_fuse_pointwise) {
        const NttPointwise pw{b, c, t->zinv};
        *h_out = ntt_run(ctx, a, tmp, m.log_n, true, true, &pw);
    } else {
        pointwise_h_run(ctx, a, b, c, t->zinv, n);
        *h_out = ntt_run(ctx, a, tmp, m.log_n, true, true);
    }
}

}  // namespace zk
