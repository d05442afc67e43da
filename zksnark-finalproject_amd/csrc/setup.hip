// Groth16 circuit-specific setup from a caller-supplied trapdoor, on the device — the step the reference runs on every
// request right before the hot path (`Groth16::<Bls12_381>::setup` at /root/reference/src/arkworks/backend/matrix_proof.rs:129,
// `circuit_specific_setup` at fibbonaci_handler.rs:107 and prime_snark.rs:112-113; upstream ark-groth16 0.4
// `generate_parameters_with_qap`, src/generator.rs — SURVEY.md A.7, scope row f-1).
//   L_i(tau)                    one inverse NTT of (tau^j)_j                      [ntt.hip]
//   u_k, v_k, w_k               column sums of A, B, C weighted by L  (CSR -> column order by a radix sort of (col, entry))
//   query scalars               a = u, b = v, l = (beta u + alpha v + w)/delta, gamma_abc = (...)/gamma, h_i = tau^i Z(tau)/delta
//   query points                fixed-base batch multiplication [scalar] g      [msm.hip fixed_base_*]
// The caller draws tau, alpha, beta, gamma, delta and the two generators exactly as upstream does (from its rng).
#include <algorithm>

#include "common.hpp"

namespace zk {

__device__ __forceinline__ Fr ld_fr(const Fr *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fr v;
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w; v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
    return v;
}
__device__ __forceinline__ void st_fr(Fr *p, const Fr &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// keys[e] = col[e] << 32 | e, and the row of every CSR entry
__global__ void __launch_bounds__(256) setup_expand_kernel(const uint64_t *row_ptr, const uint32_t *col, size_t nrows, uint64_t *keys, uint32_t *rowid) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    for (uint64_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
        keys[e] = ((uint64_t)col[e] << 32) | (uint64_t)(uint32_t)e;
        rowid[e] = (uint32_t)r;
    }
}
// first sorted position whose column is >= k, k = 0..ncols
__global__ void __launch_bounds__(256) setup_col_offsets_kernel(const uint2 *sorted, size_t nnz, uint32_t *offsets, size_t ncols) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > ncols) return;
    size_t lo = 0, hi = nnz;
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if (sorted[mid].y < (uint32_t)k) lo = mid + 1; else hi = mid;
    }
    offsets[k] = (uint32_t)lo;
}
// out[k] = sum over the entries of column k of coeff[e] * L[row[e]]   (+ L[nc + k] for k < num_instance when add_inputs)
// Columns longer than COL_HEAVY entries are not walked by one lane: they are queued as slices of COL_SLICE entries (the
// constant-one column of the 128x128 MatrixCircuit holds millions of entries — one block walking it alone took 15.7 ms per
// matrix, 47 ms of a 250 ms setup), one block per slice (setup_col_sum_heavy_kernel), and the slices of a column are added up by
// setup_col_sum_gather_kernel.  Queue: heavy[0] = number of slices, then (column, slice index, slices of that column) triples;
// the slices of one column are contiguous (one atomicAdd reserves them).
constexpr uint32_t COL_HEAVY = 1024;          // a lane walks up to this many entries itself (~1 ms)
constexpr uint32_t COL_SLICE = 8192;          // entries per queued slice: 32 per lane of a 256-lane block
static size_t col_queue_slots(size_t nnz) { return nnz / COL_HEAVY + nnz / COL_SLICE + 2; }      // heavy columns + full slices, rounded up
__global__ void __launch_bounds__(256) setup_col_sum_kernel(const uint2 *sorted, const uint32_t *offsets, const uint32_t *rowid, const Fr *coeff,
                                                            const Fr *L, size_t ncols, size_t nc, size_t num_instance, int add_inputs, Fr *out,
                                                            uint32_t *heavy) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncols) return;
    const uint32_t lo = offsets[k], hi = offsets[k + 1];
    if (hi - lo > COL_HEAVY) {
        const uint32_t slices = (hi - lo + COL_SLICE - 1) / COL_SLICE;
        const uint32_t at = atomicAdd(heavy, slices);
        for (uint32_t j = 0; j < slices; j++) {
            heavy[1 + 3 * (at + j)] = (uint32_t)k;
            heavy[2 + 3 * (at + j)] = j;
            heavy[3 + 3 * (at + j)] = slices;
        }
        return;
    }
    Fr acc = Fr::zero();
    for (uint32_t p = lo; p < hi; p++) {
        const uint32_t e = sorted[p].x;
        acc = fp_add(acc, fp_mul(ld_fr(coeff + e), ld_fr(L + rowid[e])));
    }
    if (add_inputs && k < num_instance) acc = fp_add(acc, ld_fr(L + nc + k));
    st_fr(out + k, acc);
}
// one 256-lane block per queued slice: strided partial sums, then an LDS tree; partial[q] = the slice's sum
__global__ void __launch_bounds__(256) setup_col_sum_heavy_kernel(const uint2 *sorted, const uint32_t *offsets, const uint32_t *rowid, const Fr *coeff,
                                                                  const Fr *L, const uint32_t *heavy, Fr *partial) {
    __shared__ uint32_t part[8][256];
    const uint32_t cnt = heavy[0];
    for (uint32_t q = blockIdx.x; q < cnt; q += gridDim.x) {
        const uint32_t k = heavy[1 + 3 * q], j = heavy[2 + 3 * q];
        const uint32_t lo = offsets[k] + j * COL_SLICE, end = offsets[k + 1], hi = end - lo > COL_SLICE ? lo + COL_SLICE : end;
        Fr acc = Fr::zero();
        for (uint32_t p = lo + threadIdx.x; p < hi; p += 256) {
            const uint32_t e = sorted[p].x;
            acc = fp_add(acc, fp_mul(ld_fr(coeff + e), ld_fr(L + rowid[e])));
        }
        for (int i = 0; i < 8; i++) part[i][threadIdx.x] = acc.l[i];
        __syncthreads();
        for (int step = 128; step >= 1; step >>= 1) {
            if ((int)threadIdx.x < step) {
                Fr o;
                for (int i = 0; i < 8; i++) o.l[i] = part[i][threadIdx.x + step];
                acc = fp_add(acc, o);
                for (int i = 0; i < 8; i++) part[i][threadIdx.x] = acc.l[i];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) st_fr(partial + q, acc);
        __syncthreads();
    }
}
// one lane per queued slice; the lane of a column's slice 0 adds the column's partials (at most nnz / COL_SLICE of them)
__global__ void __launch_bounds__(256) setup_col_sum_gather_kernel(const uint32_t *heavy, const Fr *partial, const Fr *L, size_t nc, size_t num_instance,
                                                                   int add_inputs, Fr *out) {
    const uint32_t cnt = heavy[0];
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < cnt; q += gridDim.x * blockDim.x) {
        if (heavy[2 + 3 * q] != 0) continue;
        const uint32_t k = heavy[1 + 3 * q], slices = heavy[3 + 3 * q];
        Fr acc = ld_fr(partial + q);
        for (uint32_t j = 1; j < slices; j++) acc = fp_add(acc, ld_fr(partial + q + j));
        if (add_inputs && k < num_instance) acc = fp_add(acc, ld_fr(L + nc + k));
        st_fr(out + k, acc);
    }
}
// lg[k] = (beta u_k + alpha v_k + w_k) * (k < num_instance ? gamma^-1 : delta^-1)
__global__ void __launch_bounds__(256) setup_lg_kernel(const Fr *u, const Fr *v, const Fr *w, Fr alpha, Fr beta, Fr ginv, Fr dinv, size_t ncols,
                                                       size_t num_instance, Fr *lg) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncols) return;
    Fr x = fp_add(fp_add(fp_mul(beta, ld_fr(u + k)), fp_mul(alpha, ld_fr(v + k))), ld_fr(w + k));
    st_fr(lg + k, fp_mul(x, k < num_instance ? ginv : dinv));
}

void setup_run(zkg16_ctx *ctx, const R1csDev &m, const Fr trap[5], const G1Affine &g1, const G2Affine &g2, const SetupOut &out, PkDev *res) {
    const Fr &tau = trap[0], &alpha = trap[1], &beta = trap[2], &gamma = trap[3], &delta = trap[4];
    const size_t N = (size_t)1 << m.log_n, nc = m.num_constraints, ni = m.num_instance, nv = m.num_variables;
    Fr zt = tau;
    for (int i = 0; i < m.log_n; i++) zt = fp_sqr(zt);
    zt = fp_sub(zt, Fr::one());                                  // Z(tau) = tau^N - 1
    if (zt.is_zero()) throw HipError{hipErrorInvalidValue, "setup: tau lies in the evaluation domain", __FILE__, __LINE__};
    const Fr dinv = fp_inv(delta), ginv = fp_inv(gamma);

    // L = ifft((tau^j)_j)
    DevBuf Lsrc(N * sizeof(Fr)), L(N * sizeof(Fr));
    fr_powers_run(ctx, Lsrc.as<Fr>(), tau, Fr::one(), N);
    ntt_run(ctx, Lsrc.as<Fr>(), L.as<Fr>(), m.log_n, true, false);      // out of place: the transform ends in L

    // u, v, w: per-matrix column sums
    DevBuf uvw[3];
    const size_t queue_slots = col_queue_slots(std::max(m.nnz[0], std::max(m.nnz[1], m.nnz[2])));
    DevBuf keys, sorted, rowid, coloff, sort_temp, heavy((1 + 3 * queue_slots) * sizeof(uint32_t)), partial(queue_slots * sizeof(Fr));
    unsigned key_bits = 1;
    while (((size_t)1 << key_bits) <= nv) key_bits++;
    for (int k = 0; k < 3; k++) {
        uvw[k].alloc(nv * sizeof(Fr));
        const size_t nnz = m.nnz[k];
        keys.ensure((nnz ? nnz : 1) * sizeof(uint64_t));
        sorted.ensure((nnz ? nnz : 1) * sizeof(uint64_t));
        rowid.ensure((nnz ? nnz : 1) * sizeof(uint32_t));
        coloff.ensure((nv + 1) * sizeof(uint32_t));
        if (nnz) {
            hipLaunchKernelGGL(setup_expand_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, ctx->stream, m.rp[k].as<uint64_t>(),
                               m.col[k].as<uint32_t>(), nc, keys.as<uint64_t>(), rowid.as<uint32_t>());
            radix_sort_hi32(ctx, keys.as<uint64_t>(), sorted.as<uint64_t>(), nnz, key_bits, sort_temp, "setup_radix_sort");
        }
        hipLaunchKernelGGL(setup_col_offsets_kernel, dim3((unsigned)((nv + 1 + 255) / 256)), dim3(256), 0, ctx->stream, sorted.as<uint2>(), nnz,
                           coloff.as<uint32_t>(), nv);
        ZK_HIP(hipMemsetAsync(heavy.p, 0, sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(setup_col_sum_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, ctx->stream, sorted.as<uint2>(),
                           coloff.as<uint32_t>(), rowid.as<uint32_t>(), m.cf[k].as<Fr>(), L.as<Fr>(), nv, nc, ni, k == 0 ? 1 : 0, uvw[k].as<Fr>(),
                           heavy.as<uint32_t>());
        hipLaunchKernelGGL(setup_col_sum_heavy_kernel, dim3(2048), dim3(256), 0, ctx->stream, sorted.as<uint2>(), coloff.as<uint32_t>(),
                           rowid.as<uint32_t>(), m.cf[k].as<Fr>(), L.as<Fr>(), heavy.as<uint32_t>(), partial.as<Fr>());
        hipLaunchKernelGGL(setup_col_sum_gather_kernel, dim3(64), dim3(256), 0, ctx->stream, heavy.as<uint32_t>(), partial.as<Fr>(), L.as<Fr>(), nc, ni,
                           k == 0 ? 1 : 0, uvw[k].as<Fr>());
        ZK_HIP(hipGetLastError());
    }
    DevBuf lg(nv * sizeof(Fr));
    hipLaunchKernelGGL(setup_lg_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, ctx->stream, uvw[0].as<Fr>(), uvw[1].as<Fr>(),
                       uvw[2].as<Fr>(), alpha, beta, ginv, dinv, nv, ni, lg.as<Fr>());
    ZK_HIP(hipGetLastError());
    // h_i = tau^i * Z(tau) / delta, i < N - 1   (reuses L's storage)
    fr_powers_run(ctx, L.as<Fr>(), tau, fp_mul(zt, dinv), N - 1);

    DevBuf canon, pts;
    if (res) {      // resident key (whole key on this device): layouts as in zkg16_pk_load with shard 0 of 1
        res->num_instance = ni; res->m_total = nv; res->n_h_total = N - 1;
        res->z_lo = 0; res->z_hi = nv; res->h_lo = 0; res->h_hi = N - 1;
        res->blinding = true; res->full = true;
        res->a.alloc((nv + 3) * sizeof(G1AffineU)); res->b1.alloc((nv + 3) * sizeof(G1AffineU)); res->l.alloc((nv + 3) * sizeof(G1AffineU));
        res->b2.alloc((nv + 3) * sizeof(G2AffineU)); res->h.alloc((N > 1 ? N - 1 : 1) * sizeof(G1AffineU));
        ZK_HIP(hipMemsetAsync(res->a.p, 0, res->a.bytes, ctx->stream));
        ZK_HIP(hipMemsetAsync(res->b1.p, 0, res->b1.bytes, ctx->stream));
        ZK_HIP(hipMemsetAsync(res->l.p, 0, res->l.bytes, ctx->stream));
        ZK_HIP(hipMemsetAsync(res->b2.p, 0, res->b2.bytes, ctx->stream));
    }
    // alpha, beta, delta (G1) ; beta, delta, gamma (G2)
    Fr singles[4] = {alpha, beta, delta, gamma};
    DevBuf d_s(4 * sizeof(Fr));
    ZK_HIP(hipMemcpyAsync(d_s.p, singles, sizeof singles, hipMemcpyHostToDevice, ctx->stream));
    uint64_t o1[4 * 12], o2[4 * 24];
    // One fixed-base pass per group for all of its queries (every pass pays two latency chains whatever its size: the 32
    // dependent mixed additions per point and the one inversion per thread of the batched to-affine; seven G1 passes and
    // three G2 passes cost a 1,594-constraint setup 9 of its 15 ms).
    //   G1: a | b1 | h | l | gamma_abc | alpha, beta, delta, (gamma)        G2: b2 | beta, delta, gamma via (alpha, beta, delta, gamma)
    // The G2 pass is queued on the second stream first, the G1 pass on the main stream: their latency chains overlap.
    DevBuf canon2, pts2;
    G2Affine *g2_sat[2] = {nullptr, nullptr};
    hipEvent_t ev_in = nullptr;
    ZK_HIP(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
    ZK_HIP(hipEventRecord(ev_in, ctx->stream));                  // uvw, lg, L, d_s are ready on the main stream
    ZK_HIP(hipStreamWaitEvent(ctx->wm_stream, ev_in, 0));
    {
        const size_t len[2] = {nv, 4};
        const Fr *src[2] = {uvw[1].as<Fr>(), d_s.as<Fr>()};
        const size_t start[3] = {0, nv, nv + 4};
        canon2.ensure((nv + 4) * sizeof(Fr));
        pts2.ensure((res ? 4 : nv + 4) * sizeof(G2Affine));
        G2Affine *sat = pts2.as<G2Affine>();
        G2AffineU *ou[2] = {res ? res->b2.as<G2AffineU>() : nullptr, nullptr};
        g2_sat[0] = res ? nullptr : sat;
        g2_sat[1] = res ? sat : sat + nv;
        std::swap(ctx->stream, ctx->wm_stream);                  // every launch helper targets ctx->stream
        try {
            for (int k = 0; k < 2; k++) fr_from_mont_run(ctx, src[k], canon2.as<Fr>() + start[k], len[k]);
            fixed_base_g2_multi(ctx, g2, canon2.as<Fr>(), nv + 4, 2, start, ou, g2_sat, false);
        } catch (...) {
            std::swap(ctx->stream, ctx->wm_stream);
            (void)hipEventDestroy(ev_in);
            throw;
        }
        std::swap(ctx->stream, ctx->wm_stream);
    }
    (void)hipEventDestroy(ev_in);
    {
        const size_t len[6] = {nv, nv, N - 1, nv - ni, ni, 4};
        const Fr *src[6] = {uvw[0].as<Fr>(), uvw[1].as<Fr>(), L.as<Fr>(), lg.as<Fr>() + ni, lg.as<Fr>(), d_s.as<Fr>()};
        size_t start[7] = {0};
        for (int k = 0; k < 6; k++) start[k + 1] = start[k] + len[k];
        const size_t total = start[6];
        canon.ensure(total * sizeof(Fr));
        for (int k = 0; k < 6; k++)
            if (len[k]) fr_from_mont_run(ctx, src[k], canon.as<Fr>() + start[k], len[k]);
        // saturated outputs (host-bound): everything for zkg16_setup, only gamma_abc and the singles for the resident key
        pts.ensure((res ? ni + 4 : total) * sizeof(G1Affine));
        G1Affine *sat = pts.as<G1Affine>();
        G1AffineU *ou[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        G1Affine *os[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (res) {
            ou[0] = res->a.as<G1AffineU>(); ou[1] = res->b1.as<G1AffineU>(); ou[2] = res->h.as<G1AffineU>(); ou[3] = res->l.as<G1AffineU>() + ni;
            os[4] = sat; os[5] = sat + ni;
        } else {
            for (int k = 0; k < 6; k++) os[k] = sat + start[k];
        }
        fixed_base_g1_multi(ctx, g1, canon.as<Fr>(), total, 6, start, ou, os);
        auto fetch = [&](const G1Affine *dsrc, size_t n, uint64_t *dst, uint8_t *inf) {
            if (!n) return;
            ZK_HIP(hipMemcpyAsync(dst, dsrc, n * sizeof(G1Affine), hipMemcpyDeviceToHost, ctx->stream));
            ZK_HIP(hipStreamSynchronize(ctx->stream));
            if (inf) {
                const G1Affine *o = reinterpret_cast<const G1Affine *>(dst);
                for (size_t i = 0; i < n; i++) inf[i] = o[i].is_inf() ? 1 : 0;
            }
        };
        if (!res) {
            fetch(os[0], nv, out.a_query, out.a_inf);
            fetch(os[1], nv, out.b_g1_query, out.b_g1_inf);
            fetch(os[2], N - 1, out.h_query, nullptr);
            fetch(os[3], nv - ni, out.l_query, out.l_inf);
        }
        fetch(os[4], ni, out.gamma_abc_g1, nullptr);
        fetch(os[5], 4, o1, nullptr);
    }
    ZK_HIP(hipStreamSynchronize(ctx->wm_stream));               // the G2 pass
    if (!res) {
        ZK_HIP(hipMemcpy(out.b_g2_query, g2_sat[0], nv * sizeof(G2Affine), hipMemcpyDeviceToHost));
        const G2Affine *o = reinterpret_cast<const G2Affine *>(out.b_g2_query);
        for (size_t i = 0; i < nv; i++) out.b_g2_inf[i] = o[i].is_inf() ? 1 : 0;
    }
    ZK_HIP(hipMemcpy(o2, g2_sat[1], 4 * sizeof(G2Affine), hipMemcpyDeviceToHost));
    if (res) {
        memcpy(&res->alpha_g1, o1, 96); memcpy(&res->beta_g1, o1 + 12, 96); memcpy(&res->delta_g1, o1 + 24, 96);
        memcpy(&res->beta_g2, o2 + 24, 192); memcpy(&res->delta_g2, o2 + 48, 192);
        // extra slots (scalars r, s, -rs): a += r*delta1 ; b1 += s*delta1 ; b2 += s*delta2 ; l += (-rs)*delta1
        const G1AffineU d1{to_u(res->delta_g1.x), to_u(res->delta_g1.y)};
        const G2AffineU d2{to_u(res->delta_g2.x), to_u(res->delta_g2.y)};
        ZK_HIP(hipMemcpyAsync(res->a.as<G1AffineU>() + nv + 0, &d1, sizeof d1, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipMemcpyAsync(res->b1.as<G1AffineU>() + nv + 1, &d1, sizeof d1, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipMemcpyAsync(res->b2.as<G2AffineU>() + nv + 1, &d2, sizeof d2, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipMemcpyAsync(res->l.as<G1AffineU>() + nv + 2, &d1, sizeof d1, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
        res->b_mask.alloc(nv + 3);
        res->b_skipped = b_density_mask_run(ctx, res->b1.as<G1AffineU>(), res->b2.as<G2AffineU>(), nv + 3, res->b_mask.as<uint8_t>());
    }
    memcpy(out.alpha_g1, o1, 96);
    memcpy(out.beta_g1, o1 + 12, 96);
    memcpy(out.delta_g1, o1 + 24, 96);
    memcpy(out.beta_g2, o2 + 24, 192);
    memcpy(out.delta_g2, o2 + 48, 192);
    memcpy(out.gamma_g2, o2 + 72, 192);
}

}  // namespace zk
