/* zkg16 — MI355X-native Groth16 (BLS12-381) prove hot path, C ABI.
 *
 * Drop-in boundary for the inside of `Groth16::<Bls12_381>::prove(&pk, circuit, &mut rng)` as called by
 * ArielElb/zkSnark-FinalProject:
 *     src/arkworks/backend/matrix_proof.rs:139-140      (matrix-mul + Poseidon circuit)
 *     src/arkworks/backend/fibbonaci_handler.rs:110     (Fibonacci circuit)
 *     src/arkworks/backend/prime_snark.rs:119           (Fermat-prime circuit)
 * The reference has no FFI layer of its own; the narrowest upstream seam is
 *     ark_groth16::Groth16::create_proof_with_reduction_and_matrices(pk, r, s, &matrices, num_inputs,
 *                                                                    num_constraints, &full_assignment)
 * (ark-groth16 0.4 src/prover.rs, reached from the call sites above).  `zkg16_prove` is that function; the
 * reference-side binding (Rust `extern "C"` + a `GpuGroth16::prove` wrapper) is shown in INTEGRATION.md.
 *
 * Data conventions (identical to arkworks' in-memory representation, so a Rust caller passes slices as-is):
 *   Fr  = 4 little-endian u64 limbs, Montgomery form (R = 2^256)            (ark_bls12_381::Fr)
 *   Fq  = 6 little-endian u64 limbs, Montgomery form (R = 2^384)            (ark_bls12_381::Fq)
 *   G1 affine = 12 u64: x | y                                                (G1Affine.x, .y)
 *   G2 affine = 24 u64: x.c0 | x.c1 | y.c0 | y.c1                             (G2Affine over Fq2)
 *   point at infinity: separate flag byte per point (G?Affine.infinity); coordinates then ignored
 *   "canonical" scalars = 4 LE u64 limbs of the plain residue (what `into_bigint()` yields)
 * All inputs are caller-owned and copied before the call returns; outputs are caller-allocated.
 * Every entry point returns a zkg16_status; nothing aborts or unwinds across this boundary.
 * A ctx is bound to ONE GPU (one process per GPU; multi-GPU = one ctx per rank + zkg16_prove_partial /
 * zkg16_prove_finish around a single all-gather of 5 partial points).
 * A ctx is re-entrant (actix workers call prove concurrently: src/main.rs:37-43): the proving entry points run on "lanes" — up to
 * option "lanes" (default 2) proofs at a time, each with its own streams and workspaces, all on the SAME resident key, window tables,
 * matrices and NTT tables (callers beyond that wait); a single caller always gets lane 0 and sees no difference.  Everything else
 * is serialised by the ctx mutex.  pk / r1cs / witness handles are immutable after load; freeing one while a proof on another lane
 * still uses it is safe (the memory goes back when that proof ends).
 */
#ifndef ZKG16_H
#define ZKG16_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define ZKG16_API __attribute__((visibility("default")))

typedef enum {
    ZKG16_OK = 0,
    ZKG16_ERR_BAD_ARG = 1,          /* null pointer, inconsistent lengths */
    ZKG16_ERR_DOMAIN_TOO_LARGE = 2, /* = ark SynthesisError::PolynomialDegreeTooLarge (domain > 2^32), or > build limit */
    ZKG16_ERR_HIP = 3,              /* a HIP runtime call failed; see zkg16_last_error */
    ZKG16_ERR_OOM = 4,
    ZKG16_ERR_NO_DEVICE = 5,        /* no gfx950 device visible: the product has NO CPU fallback */
    ZKG16_ERR_BAD_HANDLE = 6,
    ZKG16_ERR_UNSUPPORTED = 7
} zkg16_status;

typedef struct zkg16_ctx zkg16_ctx;

/* ---- lifecycle.  device_ids/n_devices: exactly one device per ctx (n_devices == 1). */
ZKG16_API int zkg16_init(const int *device_ids, int n_devices, zkg16_ctx **out);
ZKG16_API void zkg16_destroy(zkg16_ctx *ctx);
ZKG16_API const char *zkg16_strerror(int status);
ZKG16_API const char *zkg16_last_error(zkg16_ctx *ctx); /* text of the last HIP failure on this ctx */
ZKG16_API const char *zkg16_version(void);

/* ---- proving key residency  (replaces holding `ProvingKey<Bls12_381>` on the host: matrix_proof.rs:129-140).
 * Query lengths follow ark-groth16: n_a == n_b1 == n_b2 == num_instance + num_witness, n_h == N - 1,
 * n_l == num_witness.  *_inf may be NULL (= no infinity points).  shard_index/shard_count: this ctx keeps
 * only the contiguous index range [i*n/count, (i+1)*n/count) of every query vector (1 GPU: 0, 1). */
ZKG16_API int zkg16_pk_load(zkg16_ctx *ctx,
                  const uint64_t *a_query, const uint8_t *a_inf, size_t n_a,
                  const uint64_t *b_g1_query, const uint8_t *b_g1_inf, size_t n_b1,
                  const uint64_t *b_g2_query, const uint8_t *b_g2_inf, size_t n_b2,
                  const uint64_t *h_query, const uint8_t *h_inf, size_t n_h,
                  const uint64_t *l_query, const uint8_t *l_inf, size_t n_l,
                  const uint64_t alpha_g1[12], const uint64_t beta_g1[12], const uint64_t beta_g2[24],
                  const uint64_t delta_g1[12], const uint64_t delta_g2[24],
                  size_t num_instance, int shard_index, int shard_count, uint64_t *pk_handle);
/* The same with explicit index ranges instead of an equal split: this ctx keeps [z_lo, z_hi) of a_query / b_g1_query /
 * b_g2_query (and the l_query entries of those variables) and [h_lo, h_hi) of h_query; `blinding` != 0 on exactly one rank of a
 * proof (its MSMs carry the r*delta, s*delta, -rs*delta terms).  A rank with an empty h range skips the witness map and the H
 * MSM, one with an empty z range (and blinding == 0) skips the four z-side MSMs: ranks may take different roles (zkg16_shard_plan). */
ZKG16_API int zkg16_pk_load_range(zkg16_ctx *ctx,
                  const uint64_t *a_query, const uint8_t *a_inf, size_t n_a,
                  const uint64_t *b_g1_query, const uint8_t *b_g1_inf, size_t n_b1,
                  const uint64_t *b_g2_query, const uint8_t *b_g2_inf, size_t n_b2,
                  const uint64_t *h_query, const uint8_t *h_inf, size_t n_h,
                  const uint64_t *l_query, const uint8_t *l_inf, size_t n_l,
                  const uint64_t alpha_g1[12], const uint64_t beta_g1[12], const uint64_t beta_g2[24],
                  const uint64_t delta_g1[12], const uint64_t delta_g2[24],
                  size_t num_instance, size_t z_lo, size_t z_hi, size_t h_lo, size_t h_hi, int blinding, uint64_t *pk_handle);
/* A shard cut out of a whole key that is already resident on this ctx (zkg16_setup_resident, or zkg16_pk_load with shard 0 of 1):
 * device-to-device copies only.  The source handle stays valid (free it with zkg16_pk_free when the shard is all a rank needs). */
ZKG16_API int zkg16_pk_slice(zkg16_ctx *ctx, uint64_t pk_handle, size_t z_lo, size_t z_hi, size_t h_lo, size_t h_hi, int blinding,
                   uint64_t *shard_handle);
/* Window tables for a key (or shard) that stays resident: next to every base P_i of the five queries its multiples 2^(c w) P_i
 * for the windows w = 1 .. 254/c, built on the device.  Proofs on this handle then put all digits of a scalar into ONE bucket set
 * per MSM (fewer bucket additions at the same bucket count, one reduction per MSM) — same proofs bit for bit, 6-13 % less time per
 * proof (128x128: 181 -> 171 ms, 46x46: 22.6 -> 19.7 ms), for (254/c) x the key's HBM and c * (254/c) doublings per base once.  window_bits_z covers a / b_g1 / b_g2 / l,
 * window_bits_h h_query: 0 = chosen from the query length (none under 393,216 terms), 4..24 = as given, < 0 = leave that side
 * without a table.
 * table_bytes (nullable): HBM added.  Not part of the reference's flow (its key is rebuilt per request): for servers that keep a
 * circuit's key resident.  A second call for a side that already has its table is ZKG16_ERR_BAD_ARG. */
ZKG16_API int zkg16_pk_precompute(zkg16_ctx *ctx, uint64_t pk_handle, int window_bits_z, int window_bits_h, uint64_t *table_bytes);
/* The widths of a handle's window tables (0 = that side has none). */
ZKG16_API int zkg16_pk_table_bits(zkg16_ctx *ctx, uint64_t pk_handle, int *window_bits_z, int *window_bits_h);
/* Rank roles for one proof over n_ranks GPUs (host-only, no ctx, no GPU).  ranges: n_ranks x 4 = z_lo, z_hi, h_lo, h_hi per rank;
 * blinding: n_ranks flags (exactly one set).  The first *h_ranks_out ranks run the witness map and share h_query by index range;
 * every rank gets the share of the z ranges that makes all ranks finish together under a cost model in G1 mixed additions
 * (b_density = fraction of variables present in the B queries, <= 0 for the default 0.8; h_ranks = 0 lets the model choose,
 * h_ranks = n_ranks gives the equal split of zkg16_pk_load's shard_index / shard_count).  z_cost (nullable): m_total per-index
 * costs of the z-side terms in G1 mixed additions — (0, 1 or window-count entries of the scalar) x (queries whose base is not
 * at infinity, the G2 one counted 2.8x); with it the z ranges are cut by cumulative cost instead of by index count. */
ZKG16_API int zkg16_shard_plan(int n_ranks, size_t m_total, size_t n_h, double b_density, int h_ranks, const float *z_cost,
                     uint64_t *ranges /* n_ranks x 4 */, uint8_t *blinding /* n_ranks */, int *h_ranks_out);
/* The same plan with the cost factors of shards that carry window tables (window_tables != 0; zkg16_pk_precompute on every shard). */
ZKG16_API int zkg16_shard_plan_tables(int n_ranks, size_t m_total, size_t n_h, double b_density, int h_ranks, const float *z_cost,
                            int window_tables, uint64_t *ranges, uint8_t *blinding, int *h_ranks_out);
ZKG16_API void zkg16_pk_free(zkg16_ctx *ctx, uint64_t pk_handle);

/* ---- R1CS residency (ark_relations `ConstraintMatrices<Fr>` as CSR; matrices are per-circuit constants).
 * row_ptr: num_constraints + 1 entries; col: nnz u32; coeff: nnz x 4 limbs (Montgomery). */
ZKG16_API int zkg16_r1cs_load(zkg16_ctx *ctx,
                    const uint64_t *a_row_ptr, const uint32_t *a_col, const uint64_t *a_coeff,
                    const uint64_t *b_row_ptr, const uint32_t *b_col, const uint64_t *b_coeff,
                    const uint64_t *c_row_ptr, const uint32_t *c_col, const uint64_t *c_coeff,
                    size_t num_instance, size_t num_constraints, size_t num_variables, uint64_t *r1cs_handle);
ZKG16_API void zkg16_r1cs_free(zkg16_ctx *ctx, uint64_t r1cs_handle);

/* ---- full assignment z = instance || witness (Montgomery), n_assign x 4 limbs, uploaded once per proof. */
ZKG16_API int zkg16_witness_load(zkg16_ctx *ctx, const uint64_t *full_assignment, size_t n_assign, uint64_t *witness_handle);
ZKG16_API void zkg16_witness_free(zkg16_ctx *ctx, uint64_t witness_handle);
/* the assignment behind a handle, copied back (n_assign must match the handle's length) */
ZKG16_API int zkg16_witness_read(zkg16_ctx *ctx, uint64_t witness_handle, uint64_t *full_assignment_out, size_t n_assign);

/* ---- the hot path.  r, s: Montgomery Fr (drawn by the caller exactly as ark-groth16 does: r then s).
 * proof_out = A (12) | B (24) | C (12) affine Montgomery limbs; inf_out[3] = infinity flags of A, B, C. */
ZKG16_API int zkg16_prove_resident(zkg16_ctx *ctx, uint64_t pk_handle, uint64_t r1cs_handle, uint64_t witness_handle,
                         const uint64_t r[4], const uint64_t s[4], uint64_t proof_out[48], uint8_t inf_out[3]);

/* One-shot form mirroring create_proof_with_reduction_and_matrices: host pointers in, proof out
 * (uploads matrices + assignment, proves, frees). */
ZKG16_API int zkg16_prove(zkg16_ctx *ctx, uint64_t pk_handle, const uint64_t r[4], const uint64_t s[4],
                const uint64_t *a_row_ptr, const uint32_t *a_col, const uint64_t *a_coeff,
                const uint64_t *b_row_ptr, const uint32_t *b_col, const uint64_t *b_coeff,
                const uint64_t *c_row_ptr, const uint32_t *c_col, const uint64_t *c_coeff,
                size_t num_instance, size_t num_constraints,
                const uint64_t *full_assignment, size_t n_assign,
                uint64_t proof_out[48], uint8_t inf_out[3]);

/* Multi-GPU (index-range sharded pk): each rank computes h redundantly and the MSM partial sums over its shard.
 * partial_out = 4 G1 affine (H, L, A, B1 partials: 4 x 12) | 1 G2 affine (B2 partial: 24) = 72 u64;
 * partial_inf[5].  The caller all-gathers the 72-u64 records (RCCL) and any rank finishes. */
ZKG16_API int zkg16_prove_partial(zkg16_ctx *ctx, uint64_t pk_handle, uint64_t r1cs_handle, uint64_t witness_handle,
                        const uint64_t r[4], const uint64_t s[4], uint64_t partial_out[72], uint8_t partial_inf[5]);
ZKG16_API int zkg16_prove_finish(zkg16_ctx *ctx, uint64_t pk_handle, const uint64_t r[4], const uint64_t s[4],
                       const uint64_t *partials /* n_ranks x 72 */, const uint8_t *partial_inf /* n_ranks x 5 */,
                       int n_ranks, uint64_t proof_out[48], uint8_t inf_out[3]);

/* Host-only form of the finish step (no ctx, no GPU): sums the per-rank partial records and applies the proof
 * tail  A = alpha + sum A_k,  B = beta2 + sum B2_k,  C = s*A + r*(beta1 + sum B1_k) + sum L_k + sum H_k.
 * (r*delta1, s*delta1, s*delta2 and -rs*delta1 already ride inside rank 0's partials as extra MSM terms.) */
ZKG16_API int zkg16_combine_partials(const uint64_t alpha_g1[12], const uint64_t beta_g1[12], const uint64_t beta_g2[24],
                           const uint64_t r[4], const uint64_t s[4],
                           const uint64_t *partials /* n_ranks x 72 */, const uint8_t *partial_inf /* n_ranks x 5 */,
                           int n_ranks, uint64_t proof_out[48], uint8_t inf_out[3]);

/* ---- Groth16 circuit-specific setup on the device from a caller-supplied trapdoor (scope row f-1).  Replaces
 * `Groth16::<Bls12_381>::setup(circuit, &mut rng)` (matrix_proof.rs:129, fibbonaci_handler.rs:107, prime_snark.rs:112-113): the caller
 * draws trapdoor = tau | alpha | beta | gamma | delta (5 x 4 limbs, Montgomery) and the generators g1, g2 from its rng as upstream's
 * generate_random_parameters_with_reduction does.  Outputs are caller-allocated with the lengths of zkg16_pk_load
 * (a/b_g1/b_g2: num_variables, h: N-1, l: num_witness) plus the verifying-key elements. */
ZKG16_API int zkg16_setup(zkg16_ctx *ctx, uint64_t r1cs_handle, const uint64_t trapdoor[20], const uint64_t g1_gen[12], const uint64_t g2_gen[24],
                uint64_t *a_query, uint8_t *a_inf, uint64_t *b_g1_query, uint8_t *b_g1_inf, uint64_t *b_g2_query, uint8_t *b_g2_inf,
                uint64_t *h_query, uint64_t *l_query, uint8_t *l_inf,
                uint64_t alpha_g1[12], uint64_t beta_g1[12], uint64_t beta_g2[24], uint64_t delta_g1[12], uint64_t delta_g2[24],
                uint64_t gamma_g2[24], uint64_t *gamma_abc_g1 /* num_instance x 12 */);

/* Same, but the proving key never leaves the device: returns a pk handle (as zkg16_pk_load would) plus the verifying-key
 * elements.  This is the per-request flow of the reference's handlers (setup then prove) without the host round trip. */
ZKG16_API int zkg16_setup_resident(zkg16_ctx *ctx, uint64_t r1cs_handle, const uint64_t trapdoor[20], const uint64_t g1_gen[12],
                         const uint64_t g2_gen[24], uint64_t *pk_handle,
                         uint64_t alpha_g1[12], uint64_t beta_g2[24], uint64_t gamma_g2[24], uint64_t delta_g2[24],
                         uint64_t *gamma_abc_g1 /* num_instance x 12 */);

/* ---- Groth16 verification on the host (scope row f-3; no ctx, no GPU): `verify_with_processed_vk` of the handlers
 * (matrix_proof.rs:200-205).  gamma_abc_g1: num_instance points; public_inputs: num_instance - 1 Montgomery Fr; *ok = 1 iff
 * e(A,B) = e(alpha,beta) e(sum z_i gamma_abc_i, gamma) e(C,delta). */
ZKG16_API int zkg16_verify(const uint64_t alpha_g1[12], const uint64_t beta_g2[24], const uint64_t gamma_g2[24], const uint64_t delta_g2[24],
                 const uint64_t *gamma_abc_g1, size_t num_instance, const uint64_t *public_inputs,
                 const uint64_t proof[48], const uint8_t inf[3], int *ok);

/* prepare_verifying_key (what the reference's handlers return as `pvk`: matrix_proof.rs:134-136, io.rs:62-77) and
 * verify_with_processed_vk on such a key, both host-only.  alpha_beta = e(alpha, beta) as ark's Fq12 (six Fq2 in tower order,
 * 72 u64 Montgomery limbs); *_neg_coeffs = the G2Prepared line coefficients of -gamma / -delta (*n_coeffs = 68 triples of Fq2 =
 * 68 x 36 u64 each, caller-allocated).  ark-ec's formulas are restated (un-vendored crate): self-consistent and cross-checked
 * against the plain verifier, byte parity with upstream unpinned.  Both verifiers reject proof points that are not on the curve
 * or not in the prime-order subgroup (zkg16_point_check: group 1 = G1 / 12 limbs, 2 = G2 / 24 limbs). */
ZKG16_API int zkg16_pvk_prepare(const uint64_t alpha_g1[12], const uint64_t beta_g2[24], const uint64_t gamma_g2[24], const uint64_t delta_g2[24],
                      uint64_t alpha_beta[72], uint64_t *gamma_neg_coeffs, uint64_t *delta_neg_coeffs, size_t *n_coeffs);
ZKG16_API int zkg16_verify_prepared(const uint64_t *gamma_abc_g1, size_t num_instance, const uint64_t *public_inputs, const uint64_t alpha_beta[72],
                          const uint64_t *gamma_neg_coeffs, const uint64_t *delta_neg_coeffs, size_t n_coeffs,
                          const uint64_t proof[48], const uint8_t inf[3], int *ok);
/* n compressed G1 points (48 bytes each, the ark-serialize / zcash encoding the reference's keys and proofs travel in) -> affine
 * Montgomery limbs (n x 12) + infinity flags, strict as G1Affine::deserialize_compressed (+ the subgroup check when validate).
 * status (nullable, n ints): 0 ok, 1 not compressed, 2 non-canonical infinity, 3 x not reduced, 4 not on the curve, 5 not in the
 * subgroup; returns ZKG16_ERR_BAD_ARG if any point failed. */
ZKG16_API int zkg16_g1_decompress(const uint8_t *bytes, size_t n, uint64_t *out, uint8_t *inf, int validate, int *status);
/* The same for G2 (96 bytes = x.c1 || x.c0; out n x 24 limbs), the inverse for both groups (group 1 / 2; inf nullable), and the
 * 48-byte little-endian canonical form of Fq values (the Fq12 and line coefficients of a prepared verifying key; from-bytes
 * refuses values >= q). */
ZKG16_API int zkg16_g2_decompress(const uint8_t *bytes, size_t n, uint64_t *out, uint8_t *inf, int validate, int *status);
ZKG16_API int zkg16_points_compress(int group, const uint64_t *points, const uint8_t *inf, size_t n, uint8_t *out);
ZKG16_API int zkg16_fq_to_le_bytes(const uint64_t *limbs, size_t n, uint8_t *out);
ZKG16_API int zkg16_fq_from_le_bytes(const uint8_t *bytes, size_t n, uint64_t *out);
ZKG16_API int zkg16_point_check(int group, const uint64_t *point, int *ok);

/* prod_i e(P_i, Q_i) == 1 ?  Host-only (no ctx, no GPU).  g1: n x 12 limbs, g2: n x 24 limbs, flag bytes nullable.
 * flags: ZKG16_PAIRING_PLAIN_FINAL_EXP = final exponentiation as one plain power by (q^12-1)/r (slow cross-check of the
 * default Frobenius + |z|-chain path).  The building block of zkg16_verify; mirrors ark-ec's `Pairing::multi_pairing`
 * followed by the `== one` test of verifier.rs. */
#define ZKG16_PAIRING_PLAIN_FINAL_EXP 1
ZKG16_API int zkg16_pairing_check(const uint64_t *g1, const uint8_t *g1_inf, const uint64_t *g2, const uint8_t *g2_inf, size_t n,
                        int flags, int *ok);

/* ---- host-side circuit synthesis (row a2: stays on the host; no ctx, no GPU).  C++ mirrors of the reference's circuits with
 * the same allocation order (variable k here = variable k in arkworks):
 *   MatrixCircuit     src/arkworks/matrix_proof_of_work/constraints.rs:78-128 (+ Poseidon hasher.rs:17-40, hashing_utils.rs:15-877)
 *   FibonacciCircuit  src/arkworks/constraints/fibbonaci.rs:22-48
 * zkg16_circuit_export yields exactly the arguments of zkg16_r1cs_load / zkg16_witness_load. */
typedef struct zkg16_circuit zkg16_circuit;
ZKG16_API int zkg16_circuit_matrix(size_t n, const uint64_t *a /* n*n, row-major */, const uint64_t *b, zkg16_circuit **out);
/* Only the assignment z (instance || witness, n_assign x 4 limbs) of that circuit for these inputs: the matrices of the
 * MatrixCircuit depend on n alone, so a caller that kept them from one zkg16_circuit_matrix call needs only this per request. */
ZKG16_API int zkg16_circuit_matrix_witness(size_t n, const uint64_t *a, const uint64_t *b, uint64_t *z, size_t n_assign);
ZKG16_API int zkg16_circuit_fibonacci(uint64_t a, uint64_t b, size_t steps, zkg16_circuit **out);
/*   PrimeCircuit      src/arkworks/prime_snark/prime_circut.rs:92-146 (+ fermat_circut.rs, utils/hasher.rs, utils/modulo.rs): SHA-256 of
 *                     x + j, the digest mod 2^20, three hashed Fermat bases, 20-step square-and-multiply with witnessed quotients.
 *                     The SHA-256 / Boolean / comparison gadgets come from un-vendored crates upstream and are restated here:
 *                     values (digests, modpows, satisfaction, verification) are checked, the constraint LAYOUT is unpinned.
 * zkg16_prime_search = the handler's loop over check_if_next_is_prime (backend/prime_snark.rs:57-70). */
ZKG16_API int zkg16_prime_search(uint64_t x, uint64_t i_max, uint64_t *j_out, uint32_t *prime_out, uint8_t digest_out[32], int *found);
ZKG16_API int zkg16_prime_candidate(uint64_t x, uint64_t j, uint8_t digest_out[32], uint32_t *n_out, uint32_t bases_out[3],
                          uint8_t r_bytes_out[32], int *is_prime);
ZKG16_API int zkg16_circuit_prime(uint64_t x, uint64_t j, zkg16_circuit **out);
/* The PrimeCircuit's 257 public inputs (x, then the digest bits) without building the circuit: out = 257 x 4 limbs, Montgomery. */
ZKG16_API int zkg16_prime_public_inputs(uint64_t x, uint64_t j, uint64_t *out);
ZKG16_API void zkg16_circuit_free(zkg16_circuit *c);
ZKG16_API int zkg16_circuit_dims(const zkg16_circuit *c, size_t *num_instance, size_t *num_witness, size_t *num_constraints, size_t nnz[3]);
ZKG16_API int zkg16_circuit_is_satisfied(const zkg16_circuit *c);
ZKG16_API int zkg16_circuit_export(const zkg16_circuit *c, uint64_t *const row_ptr[3], uint32_t *const col[3], uint64_t *const coeff[3],
                         uint64_t *full_assignment /* (num_instance + num_witness) x 4 */);
/* The public inputs (instance assignment without the leading one): (num_instance - 1) x 4 limbs; cap = room in elements. */
ZKG16_API int zkg16_circuit_public_inputs(const zkg16_circuit *c, uint64_t *out, size_t cap);
/* Load a synthesized circuit straight onto the device: *r1cs_handle as from zkg16_r1cs_load and *witness_handle as from
 * zkg16_witness_load of zkg16_circuit_export's arrays (same bytes on the device), staged through pinned memory the ctx keeps — the
 * request path of the handlers whose circuits are re-synthesized per request (PrimeCircuit, Fibonacci). */
ZKG16_API int zkg16_circuit_load(zkg16_ctx *ctx, const zkg16_circuit *c, uint64_t *r1cs_handle, uint64_t *witness_handle);
/* ---- the MatrixCircuit's assignment built ON THE DEVICE (scope row f-4 "on GPU"; csrc/witness.hip).  The reference re-synthesises
 * the circuit inside its timed `Groth16::prove` (matrix_proof.rs:138-145, constraints.rs:78-128): per request only the
 * assignment changes, 75 % of which are the S-box products of 3 ceil(n^2/2) Poseidon permutations (hashing_utils.rs:737-802).
 * The host runs the three native sponges (sequential by construction: hasher.rs:17-27) and keeps the state in front of every
 * permutation; kernels write z in place — a, b as Fr, the n^3 products, 265 values per permutation — into a buffer
 * zkg16_prove_resident reads.  *witness_handle: as from zkg16_witness_load of zkg16_circuit_matrix_witness's output (same bytes).
 * public_inputs (nullable): hash_a | hash_b | hash_c, Montgomery.  timings_ms (nullable, 3 floats): host sponges, device
 * (upload + kernels), whole call. */
ZKG16_API int zkg16_witness_matrix(zkg16_ctx *ctx, size_t n, const uint64_t *a, const uint64_t *b, uint64_t *witness_handle,
                         uint64_t public_inputs[12], float *timings_ms);
/* One request of the matrix handler on a key and matrices that are resident: assignment + proof, overlapped — the host sponges feed
 * the device in parts (option "matrix_parts", default five growing slices) and the z-side MSMs run on the parts that exist while the sponges
 * still compute the rest; same proof bytes as zkg16_witness_matrix + zkg16_prove_resident.  r1cs_handle / pk_handle must be the
 * MatrixCircuit of size n (else ZKG16_ERR_BAD_ARG).  public_inputs (nullable): hash_a | hash_b | hash_c.  timings_ms (nullable, 3):
 * host sponges (wall, overlapped with the device), parts used, whole call. */
ZKG16_API int zkg16_prove_matrix(zkg16_ctx *ctx, uint64_t pk_handle, uint64_t r1cs_handle, size_t n, const uint64_t *a, const uint64_t *b,
                       const uint64_t r[4], const uint64_t s[4], uint64_t proof_out[48], uint8_t inf_out[3], uint64_t public_inputs[12],
                       float *timings_ms);
/* Its host-only half (no ctx, no GPU): states (nullable) = 3 hashes x ceil(n^2/2) permutations x 3 Fr, the sponge state in
 * front of each permutation (after its two elements were absorbed); hashes = hash_a | hash_b | hash_c. */
ZKG16_API int zkg16_matrix_sponge_states(size_t n, const uint64_t *a, const uint64_t *b, uint64_t *states, uint64_t hashes[12]);
/* ---- the MatrixCircuit's R1CS of size n WITHOUT synthesising it constraint by constraint (csrc/matrix_plan.hpp): the sponge rows are
 * copies of one template per permutation class with renamed variables, matrix_mul's rows have a closed form.  _dims / _host: host
 * loops (the reference the device kernel is tested against; same arrays as zkg16_circuit_matrix + zkg16_circuit_export);
 * zkg16_r1cs_matrix: the same arrays written by a kernel straight into HBM -> an r1cs handle as from zkg16_r1cs_load, with nothing
 * but the templates (a few hundred KB) crossing PCIe instead of 3.7 GB at 128x128.  zkg16_r1cs_read copies a handle's arrays back. */
ZKG16_API int zkg16_matrix_r1cs_dims(size_t n, size_t *num_constraints, size_t *num_witness, size_t nnz[3]);
ZKG16_API int zkg16_matrix_r1cs_host(size_t n, uint64_t *const row_ptr[3], uint32_t *const col[3], uint64_t *const coeff[3]);
ZKG16_API int zkg16_r1cs_matrix(zkg16_ctx *ctx, size_t n, uint64_t *r1cs_handle);
ZKG16_API int zkg16_r1cs_read(zkg16_ctx *ctx, uint64_t r1cs_handle, uint64_t *const row_ptr[3], uint32_t *const col[3], uint64_t *const coeff[3],
                    size_t *num_instance, size_t *num_constraints, size_t *num_variables, size_t nnz[3]);
/* native Poseidon sponge hash of n Fr elements (Montgomery) — the public inputs hash_a/b/c of the matrix handler */
ZKG16_API int zkg16_poseidon_hash(const uint64_t *elems, size_t n, uint64_t out[4]);

/* ---- stage entry points (tests / bench; host buffers) ----------------------------------------- */
/* ark-poly Radix2EvaluationDomain<Fr>: in-place, natural order; inverse => ifft (incl. 1/N);
 * coset => offset g = 7 (coset_fft = g^i then fft; coset_ifft = ifft then g^-i). */
ZKG16_API int zkg16_ntt(zkg16_ctx *ctx, uint64_t *data, size_t log_n, int inverse, int coset);
/* ark-ec VariableBaseMSM::msm_bigint: sum scalars[i] * bases[i]; result affine. */
ZKG16_API int zkg16_msm_g1(zkg16_ctx *ctx, const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars_canonical,
                 size_t n, uint64_t out_affine[12], uint8_t *out_inf);
ZKG16_API int zkg16_msm_g2(zkg16_ctx *ctx, const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars_canonical,
                 size_t n, uint64_t out_affine[24], uint8_t *out_inf);
/* LibsnarkReduction::witness_map_from_matrices: h (N x 4 limbs, Montgomery), N = 2^log_n. */
ZKG16_API int zkg16_witness_map(zkg16_ctx *ctx, uint64_t r1cs_handle, uint64_t witness_handle, uint64_t *h_out, size_t *log_n_out);
/* ark-ec FixedBase::msm: out[i] = [scalars[i]] base (used by the known-trapdoor setup in tests/bench). */
ZKG16_API int zkg16_fixed_base_g1(zkg16_ctx *ctx, const uint64_t base[12], const uint64_t *scalars_canonical, size_t n,
                        uint64_t *out_affine /* n x 12 */, uint8_t *out_inf /* n */);
ZKG16_API int zkg16_fixed_base_g2(zkg16_ctx *ctx, const uint64_t base[24], const uint64_t *scalars_canonical, size_t n,
                        uint64_t *out_affine /* n x 24 */, uint8_t *out_inf /* n */);
/* One scalar multiplication on the host (no ctx, no GPU): [k] base, k canonical.  What a caller uses to derive its own
 * generators (upstream: `E::G1::rand`), where a device pass would be all latency. */
ZKG16_API int zkg16_scalar_mul_g1(const uint64_t base[12], const uint64_t k_canonical[4], uint64_t out_affine[12], uint8_t *out_inf);
ZKG16_API int zkg16_scalar_mul_g2(const uint64_t base[24], const uint64_t k_canonical[4], uint64_t out_affine[24], uint8_t *out_inf);

/* ---- device-resident stage benches (inputs uploaded once, op repeated on device) ---------------- */
ZKG16_API int zkg16_bench_ntt(zkg16_ctx *ctx, size_t log_n, int inverse, int coset, int iters, float *ms_per_iter);
ZKG16_API int zkg16_bench_witness_map(zkg16_ctx *ctx, uint64_t r1cs_handle, uint64_t witness_handle, int iters, float *ms_per_iter);
ZKG16_API int zkg16_bench_msm(zkg16_ctx *ctx, int group /*1|2*/, const uint64_t *bases, const uint8_t *inf,
                    const uint64_t *scalars_canonical, size_t n, int iters, float *ms_per_iter,
                    uint64_t *out_affine, uint8_t *out_inf);

/* ---- instrumentation --------------------------------------------------------------------------- */
/* Times of the last prove on this ctx, in ms (returns the number of entries written, up to 22):
 * [0] unused, [1] witness map (device; upstream span "R1CS to QAP witness map"), [2] scalar digits + bucket scatter of both scalar
 * vectors (device), [3..7] host-observed completion gaps of H, L, A, B1, B2 in collection order (NOT a breakdown: the first gap
 * holds most of the device time), [8] host tail ("Finish C"), [9] total wall;
 * [10..14] device time of the bucket accumulation + fix-ups of H, L, A, B1, B2 and [15..19] of their bucket reductions, from event
 * pairs on the streams they ran on: upstream's "Compute C" = H + L, "Compute A", "Compute B in G1", "Compute B in G2"
 * (ark-groth16 prover.rs).  MSMs overlap each other and the witness map, so the device times sum to more than [9];
 * [20] host time of combining H's window sums (after the proof's last device event), [21] of the other four MSMs' (under H's device work). */
ZKG16_API int zkg16_last_timings(zkg16_ctx *ctx, float *ms, int cap);
/* Lengths of the sorted (scalar, window) term lists of the last proof on this ctx = mixed additions of each MSM that walks the list:
 * [0] the z list (A and L), [1] the B list (B1 and B2; 0 = they used the z list), [2] the h list.  Synchronises the ctx. */
ZKG16_API int zkg16_last_term_counts(zkg16_ctx *ctx, uint64_t counts[3]);
/* G1 accumulation waves per SIMD (2 or 4) the last proof's term lists ran at: [0] z list, [1] B list, [2] h list; 0 = not built. */
ZKG16_API int zkg16_last_acc_waves(zkg16_ctx *ctx, int waves[3]);
/* Waves of the bucket-accumulation kernels one SIMD holds at once (their register use decides): [0] G1, [1] G2.  The grid of an
 * accumulation may be sized for more waves per SIMD than that (zkg16_last_acc_waves): the extra ones run as a second round. */
ZKG16_API int zkg16_acc_resident_waves(zkg16_ctx *ctx, int waves[2]);
/* The lanes of the most recent proofs: rows of (lane, start ms, end ms) on the host's steady clock, oldest first; returns the number of
 * rows written (<= cap_rows).  Two rows with different lanes and intersecting intervals = two proofs in flight at once. */
ZKG16_API int zkg16_lane_log(zkg16_ctx *ctx, double *rows, int cap_rows);
/* Live HIP-event timing of individual kernels (bench.py's roofline leg).  enable: 0 off, 1 every kernel family, 2 only the
 * bucket-accumulation launches (five event pairs per proof instead of ~60).  stats are accumulated per kernel name since
 * the last reset. */
ZKG16_API int zkg16_kernel_timing(zkg16_ctx *ctx, int enable);
ZKG16_API int zkg16_kernel_stats(zkg16_ctx *ctx, const char *kernel_name, uint64_t *launches, double *total_ms,
                       double *units /* kernel-specific work units, e.g. bucket additions */);
ZKG16_API void zkg16_kernel_stats_reset(zkg16_ctx *ctx);
/* Tuning / A-B switches; none of them changes a result (tests/test_gpu_parity.py toggles every one and compares proofs).
 *   "window_bits"    MSM window bits c for every plan (0 = by size: 13 / 15 / 16 / 17; <= 20)      "window_bits_h"  the H MSM's plan only
 *   "reduce_chunk"   buckets per lane in the bucket reduction (0 = 8)                   "reduce_mode"    1 = work-efficient two-level form,
 *                    2 = that form except for the proof's last MSM, 4 = classic everywhere, 5 = bit-sliced wherever it applies,
 *                    6 = the default without the bit-sliced form, 0 = default (bit-sliced for bucket sets up to 2^19, else 2 from 16-bit windows on)
 *   "sort_mode"      0 = hand-written wave-ballot bucket scatter (default), 1 = rocPRIM radix sort
 *   "acc_pipeline"   bit 0 / 1: G1 / G2 accumulation gathers the next base behind the last (inlined) product; 4 = off, 0 = default (both)
 *   "wm_concurrent"  0 = witness map in order on the main stream (default: own stream)  "fixup_aux"      1 = fix-ups on the reduction stream
 *   "g1_waves"       G1 accumulation waves per SIMD in the resident round (0 = 2)       "min_seg"        shortest per-lane run (0 = adaptive)
 *   "ntt_mode"       0 = saturated-limb butterflies (first version), 1 = unsaturated (default)
 *   "ntt_radix"      1 (default; also 0) = the last seven butterfly stages of a tile by lane exchanges, 3 = the top seven as well,
 *                    2 = every stage through the LDS, 4 = two stages per LDS trip
 *   "fuse_pointwise" 1 (default) = (ab - c)/Z fused into the load of the seventh transform, 0 = its own pass
 *   "collect_threads" host combination of the z-side MSMs' window sums: 0 = own threads for plain keys, 1 = always, 2 = never
 *   "fixed_base_bits" window width of the setup's fixed-base multiplications (0 = by batch size; 16 / 18 / 20 = two-level tables)
 *   "g2_lazy"        G2 accumulation's Fq2 products: 0 / 1 (default) two fused two-product reductions with operands parked in LDS, 2 = Karatsuba
 *   "lanes"          proofs this ctx runs at a time (1..8, default 2): see the note on re-entrancy at the top
 *   "matrix_parts"   zkg16_prove_matrix: slices of the host sponges the proof is fed in (0 = five growing slices, k = k equal ones; 1 = assignment first, then the proof)
 * Unknown names return ZKG16_ERR_UNSUPPORTED. */
ZKG16_API int zkg16_set_option(zkg16_ctx *ctx, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* ZKG16_H */
