// Instruction-throughput microbenchmark for the integer paths a 381-bit Montgomery product can be built from
// on gfx950.  Prints ops/s per instruction kind at full-chip occupancy.  (Development tool, not product.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../zksnark-finalproject_amd/csrc/ec.cuh"
using namespace zk;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;

#define DEF_KERNEL(name, decl, body)                                          \
    __global__ void __launch_bounds__(256) name(uint32_t *out, uint32_t seed) { \
        decl;                                                                 \
        for (int it = 0; it < ITERS; it++) { body; }                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = sink;                    \
    }

// 8 independent accumulators per lane to cover the pipeline latency
__global__ void __launch_bounds__(256) k_mad_u64_u32(uint32_t *out, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t x = seed * 2654435761u + threadIdx.x, y = x ^ 0x9e3779b9u;
    for (int it = 0; it < ITERS; it++) {
#define M(a) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y) : "vcc");
        M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
#define K32(name, INS)                                                                                          \
    __global__ void __launch_bounds__(256) name(uint32_t *out, uint32_t seed) {                                 \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t x = seed * 2654435761u + threadIdx.x;                                                          \
        for (int it = 0; it < ITERS; it++) {                                                                    \
            asm volatile(INS : "+v"(a0) : "v"(x)); asm volatile(INS : "+v"(a1) : "v"(x));                       \
            asm volatile(INS : "+v"(a2) : "v"(x)); asm volatile(INS : "+v"(a3) : "v"(x));                       \
            asm volatile(INS : "+v"(a4) : "v"(x)); asm volatile(INS : "+v"(a5) : "v"(x));                       \
            asm volatile(INS : "+v"(a6) : "v"(x)); asm volatile(INS : "+v"(a7) : "v"(x));                       \
        }                                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                      \
    }
K32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
K32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
K32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %0")
K32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
K32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
K32(k_add_u32, "v_add_u32 %0, %0, %1")
K32(k_add3_u32, "v_add3_u32 %0, %0, %1, %0")
K32(k_mov_b32, "v_mov_b32 %0, %1")
K32(k_mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %0")
K32(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")

__global__ void __launch_bounds__(256) k_lshl_add_u64(uint32_t *out, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint64_t x = seed * 2654435761u + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#define M(a) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a) : "v"(x));
        M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void __launch_bounds__(256) k_fma_f64(uint32_t *out, uint32_t seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double x = 1.0000001, y = 1e-9;
    for (int it = 0; it < ITERS; it++) {
#define M(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
        M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void __launch_bounds__(256) k_fma_f32(uint32_t *out, uint32_t seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float x = 1.0000001f, y = 1e-9f;
    for (int it = 0; it < ITERS; it++) {
#define M(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
        M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
// the product's own Fq / Fr Montgomery multiplication, dependent chain per lane (throughput comes from occupancy)
__global__ void __launch_bounds__(256) k_fq_mul(uint32_t *out, uint32_t seed) {
    Fq a = Fq::one(), b = Fq::r2();
    a.l[0] += threadIdx.x; b.l[1] ^= seed;
    for (int it = 0; it < ITERS / 16; it++) { a = fp_mul(a, b); b = fp_mul(b, a); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.l[0] ^ b.l[3];
}
__global__ void __launch_bounds__(256) k_fq_mul_inline(uint32_t *out, uint32_t seed) {
    Fq a = Fq::one(), b = Fq::r2();
    a.l[0] += threadIdx.x; b.l[1] ^= seed;
    for (int it = 0; it < ITERS / 16; it++) { a = fp_mul_inline(a, b); b = fp_mul_inline(b, a); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.l[0] ^ b.l[3];
}
__global__ void __launch_bounds__(256) k_fr_mul(uint32_t *out, uint32_t seed) {
    Fr a = Fr::one(), b = Fr::r2();
    a.l[0] += threadIdx.x; b.l[1] ^= seed;
    for (int it = 0; it < ITERS / 16; it++) { a = fp_mul(a, b); b = fp_mul(b, a); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.l[0] ^ b.l[3];
}

// unsaturated 29-bit product variants
template <int NACC>
__device__ __forceinline__ FqU fqu_mul_var(const FqU &a, const FqU &b) {
    constexpr int N = 14;
    uint32_t m[N];
    FqU r;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; k++) {
        const int lo = k < N ? 0 : k - N + 1;
        const int hi = k < N ? k : N - 1;
        uint64_t acc[4] = {carry, 0, 0, 0};
        int t = 0;
#pragma unroll
        for (int i = lo; i <= hi; i++) { acc[t % NACC] += (uint64_t)a.l[i] * b.l[k - i]; t++; }
        if (k < N) {
#pragma unroll
            for (int i = 0; i < k; i++) { acc[t % NACC] += (uint64_t)m[i] * FqUP::mod(k - i); t++; }
            uint64_t s = acc[0];
#pragma unroll
            for (int j = 1; j < NACC; j++) s += acc[j];
            m[k] = ((uint32_t)s * FqUP::INV) & FqU::MASK;
            s += (uint64_t)m[k] * FqUP::mod(0);
            carry = s >> 29;
        } else {
#pragma unroll
            for (int i = lo; i <= hi; i++) { acc[t % NACC] += (uint64_t)m[i] * FqUP::mod(k - i); t++; }
            uint64_t s = acc[0];
#pragma unroll
            for (int j = 1; j < NACC; j++) s += acc[j];
            r.l[k - N] = (uint32_t)s & FqU::MASK;
            carry = s >> 29;
        }
    }
    r.l[N - 1] = (uint32_t)carry;
    return r;
}
template <int NACC>
__global__ void __launch_bounds__(256) k_fqu_mul_var(uint32_t *out, uint32_t seed) {
    FqU a = FqU::one(), b = FqU::one();
    a.l[0] += threadIdx.x & 0xff; b.l[1] ^= seed & 0xfff;
    for (int it = 0; it < ITERS / 16; it++) { a = fqu_mul_var<NACC>(a, b); b = fqu_mul_var<NACC>(b, a); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.l[0] ^ b.l[3];
}
__global__ void __launch_bounds__(256) k_fqu_mul_call(uint32_t *out, uint32_t seed) {
    FqU a = FqU::one(), b = FqU::one();
    a.l[0] += threadIdx.x & 0xff; b.l[1] ^= seed & 0xfff;
    for (int it = 0; it < ITERS / 16; it++) { a = fqu_mul(a, b); b = fqu_mul(b, a); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.l[0] ^ b.l[3];
}
__global__ void __launch_bounds__(256) k_fqu_sqr_call(uint32_t *out, uint32_t seed) {
    FqU a = FqU::one();
    a.l[0] += threadIdx.x & 0xff; a.l[1] ^= seed & 0xfff;
    for (int it = 0; it < ITERS / 8; it++) { a = fqu_sqr(a); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.l[0];
}
// one XYZZ mixed addition per iteration (the accumulate kernel's inner body), G1 and G2
template <class F>
__global__ void __launch_bounds__(64) k_madd(uint32_t *out, uint32_t seed) {
    XYZZ<F> acc = XYZZ<F>{F::one(), F::one(), F::one(), F::one()};
    Affine<F> q{F::one(), F::one()};
    for (int it = 0; it < ITERS / 64; it++) { xyzz_madd(acc, q, (it & 1) != 0); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((uint32_t *)&acc)[0] ^ seed;
}

// ---- batched-affine bucket addition, the alternative to the XYZZ mixed addition: a lane adds B
// independent pairs P_i + Q_i in AFFINE coordinates with ONE shared inversion (Montgomery's trick):
//   forward   d_i = x2_i - x1_i,  pref_i = pref_{i-1} * d_i          (1 product; pref kept in global scratch)
//   inversion inv = pref_B^-1                                          (Fermat: 380 squarings + ~190 products)
//   backward  dinv_i = inv * pref_{i-1}, inv *= d_i, lambda = (y2 - y1) dinv_i, x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1
//             (2 + 2 products + 1 squaring)
// = 5 products + 1 squaring per addition + 570 / B for the inversion, against 8 + 2 for the XYZZ mixed addition — but the
// operands and prefix products of a batch do not fit in registers: 112 B (x1, x2) + 56 B (pref) in the forward sweep, 224 + 56 B
// read and 112 B written in the backward sweep = 560 B of memory traffic per addition, laid out [i][lane] (coalesced).
// Values need not be curve points for timing; they are distinct so no exceptional case is hit.
template <int B>
__global__ void __launch_bounds__(64, 2) k_batch_affine_g1(const FqU *pts /* [4][B][threads]: x1, y1, x2, y2 */, FqU *pref, FqU *out, uint32_t *sink) {
    const size_t T = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const FqU *x1 = pts, *y1 = pts + (size_t)B * T, *x2 = pts + 2 * (size_t)B * T, *y2 = pts + 3 * (size_t)B * T;
    FqU acc = FqU::one();
    for (int i = 0; i < B; i++) {
        const FqU d = fqu_sub<32>(x2[(size_t)i * T + t], x1[(size_t)i * T + t]);
        pref[(size_t)i * T + t] = acc;
        acc = fqu_mul(acc, d);
    }
    FqU inv = fqu_inv(acc);
    for (int i = B - 1; i >= 0; i--) {
        const FqU a = x1[(size_t)i * T + t], b = y1[(size_t)i * T + t], c = x2[(size_t)i * T + t], e = y2[(size_t)i * T + t];
        const FqU d = fqu_sub<32>(c, a);
        const FqU dinv = fqu_mul(inv, pref[(size_t)i * T + t]);
        inv = fqu_mul(inv, d);
        const FqU lam = fqu_mul(fqu_sub<32>(e, b), dinv);
        const FqU x3 = fqu_sub<32>(fqu_sqr(lam), fqu_add(a, c));
        const FqU y3 = fqu_sub<32>(fqu_mul(lam, fqu_sub<64>(a, x3)), b);
        out[(size_t)(2 * i) * T + t] = x3;
        out[(size_t)(2 * i + 1) * T + t] = y3;
    }
    sink[t] = inv.l[0];
}
__global__ void k_fill_fqu(FqU *p, size_t n, uint32_t seed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s = seed + (uint32_t)i * 2654435761u;
    FqU v;
    for (int k = 0; k < 14; k++) { s = s * 1664525u + 1013904223u; v.l[k] = (s >> 3) & (k == 13 ? 0xffu : FqU::MASK); }
    p[i] = v;
}
template <int B>
static int run_batch_affine() {
    const size_t T = (size_t)256 * 4 * 2 * 64;      // one resident round at 2 waves per SIMD, as the accumulation kernel
    FqU *pts, *pref, *out;
    uint32_t *sink;
    CHK(hipMalloc(&pts, 4 * (size_t)B * T * sizeof(FqU)));
    CHK(hipMalloc(&pref, (size_t)B * T * sizeof(FqU)));
    CHK(hipMalloc(&out, 2 * (size_t)B * T * sizeof(FqU)));
    CHK(hipMalloc(&sink, T * sizeof(uint32_t)));
    const size_t np = 4 * (size_t)B * T;
    hipLaunchKernelGGL(k_fill_fqu, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, 0, pts, np, 12345u);
    hipLaunchKernelGGL(k_batch_affine_g1<B>, dim3((unsigned)(T / 64)), dim3(64), 0, 0, pts, pref, out, sink);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, 0));
    const int reps = 3;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_batch_affine_g1<B>, dim3((unsigned)(T / 64)), dim3(64), 0, 0, pts, pref, out, sink);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double adds = (double)B * T * reps;
    printf("batched affine G1  B=%-5d lanes=%zu  %8.3f ms  %7.3f G additions/s   (%.0f B of traffic per addition -> %.2f TB/s)\n", B, T, ms / reps,
           adds / (ms * 1e-3) * 1e-9, 560.0, adds * 560.0 / (ms * 1e-3) * 1e-12);
    hipFree(pts); hipFree(pref); hipFree(out); hipFree(sink);
    return 0;
}

template <class K>
static int run(const char *name, K kernel, double ops_per_thread, int blocks_per_cu, uint32_t *d_out) {
    const int cus = 256;
    const int threads = strstr(name, "64thr") ? 64 : 256;
    const int grid = cus * blocks_per_cu;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), 0, 0, d_out, 12345u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    const int reps = 5;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), 0, 0, d_out, 12345u + i);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double total_ops = ops_per_thread * (double)grid * threads * reps;
    const double rate = total_ops / (ms * 1e-3);
    // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
    const double wave_instr_per_s = rate / 64.0;
    const double cyc = 2.4e9 * 1024.0 / wave_instr_per_s;
    printf("%-22s blocks/CU=%d  %8.3f ms  %10.3f Gop/s   ~%6.2f cyc/wave-instr/SIMD @2.4GHz\n", name, blocks_per_cu, ms / reps, rate * 1e-9, cyc);
    return 0;
}

int main() {
    uint32_t *d_out;
    CHK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(uint32_t)));
    const double per = 8.0 * ITERS;
    for (int b : {4, 8}) {
        run("v_mad_u64_u32", k_mad_u64_u32, per, b, d_out);
        run("v_mul_lo_u32", k_mul_lo_u32, per, b, d_out);
        run("v_mul_hi_u32", k_mul_hi_u32, per, b, d_out);
        run("v_mad_u32_u24", k_mad_u32_u24, per, b, d_out);
        run("v_mul_u32_u24", k_mul_u32_u24, per, b, d_out);
        run("v_mul_hi_u32_u24", k_mul_hi_u32_u24, per, b, d_out);
        run("v_mad_u32_u16", k_mad_u32_u16, per, b, d_out);
        run("v_add_u32", k_add_u32, per, b, d_out);
        run("v_add3_u32", k_add3_u32, per, b, d_out);
        run("v_addc_co_u32", k_addc, per, b, d_out);
        run("v_mov_b32", k_mov_b32, per, b, d_out);
        run("v_lshl_add_u64", k_lshl_add_u64, per, b, d_out);
        run("v_fma_f32", k_fma_f32, per, b, d_out);
        run("v_fma_f64", k_fma_f64, per, b, d_out);
    }
    const double muls = 2.0 * (ITERS / 16);
    for (int b : {1, 2, 4}) {
        run("FqU mul (call)", k_fqu_mul_call, muls, b, d_out);
        run("FqU sqr (call)", k_fqu_sqr_call, (double)(ITERS / 8), b, d_out);
        run("FqU mul inline 1acc", k_fqu_mul_var<1>, muls, b, d_out);
        run("FqU mul inline 2acc", k_fqu_mul_var<2>, muls, b, d_out);
        run("FqU mul inline 4acc", k_fqu_mul_var<4>, muls, b, d_out);
    }
    for (int b : {4, 8, 16}) {   // 64-thread blocks: b blocks/CU = b/4 waves per SIMD
        run("madd G1 (64thr blk)", k_madd<FqU>, (double)(ITERS / 64), b, d_out);
        run("madd G2 (64thr blk)", k_madd<Fq2U>, (double)(ITERS / 64), b, d_out);
    }
    run_batch_affine<16>();
    run_batch_affine<64>();
    run_batch_affine<256>();
    run_batch_affine<1024>();
    for (int b : {1, 2, 4, 8}) {
        run("Fq mul (call)", k_fq_mul, muls, b, d_out);
        run("Fq mul (inline)", k_fq_mul_inline, muls, b, d_out);
        run("Fr mul (inline)", k_fr_mul, muls, b, d_out);
    }
    hipFree(d_out);
    return 0;
}
