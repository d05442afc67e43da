// G2 mixed addition, one lane per point (Karatsuba and one-reduction forms) against the pair-split form (g2split.cuh): the split
// form's results are checked against the one-lane form (as group elements), then the bare loops are timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zksnark-finalproject_amd/csrc tools/experiments/microbench_g2.hip -o tools/bin/microbench_g2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "g2split.cuh"
using namespace zk;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
constexpr int ITERS = 64;

__host__ __device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__host__ __device__ inline FqU rnd_fqu(uint32_t seed) {
    FqU r;
    for (int i = 0; i < 14; i++) r.l[i] = mix(seed * 31u + i) & FqU::MASK;
    r.l[13] = mix(seed) % 12u;
    return r;
}
__host__ __device__ inline Affine<Fq2U> rnd_pt(uint32_t seed) { return Affine<Fq2U>{Fq2U{rnd_fqu(seed), rnd_fqu(seed + 7777u)}, Fq2U{rnd_fqu(seed + 99u), rnd_fqu(seed + 12345u)}}; }
// the point the t-th addition of sequence s uses: repeats (doubling), then its negative (infinity), then distinct ones
__host__ __device__ inline uint32_t pt_seed(uint32_t s, int t) { return s * 64u + (t == 1 ? 0u : t == 2 ? 0u : t == 3 ? 0u : (uint32_t)t); }
__host__ __device__ inline bool pt_neg(int t) { return t == 2 || t == 3 || (t % 5) == 4; }

__global__ void __launch_bounds__(64, 1) k_check_one(XYZZ<Fq2U> *out, int steps) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ<Fq2U> acc = XYZZ<Fq2U>::inf();
    for (int t = 0; t < steps; t++) xyzz_madd_lazy(acc, rnd_pt(pt_seed(s, t)), pt_neg(t));
    uint4 *o = reinterpret_cast<uint4 *>(out + s);
    const uint4 *v = reinterpret_cast<const uint4 *>(&acc);
    for (unsigned i = 0; i < sizeof(acc) / 16; i++) o[i] = v[i];
}
__global__ void __launch_bounds__(64, 2) k_check_split(XYZZ<Fq2U> *out, int steps) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, s = tid >> 1;
    const int h = tid & 1;
    H2XYZZ acc = h2_inf();
    for (int t = 0; t < steps; t++) {
        const Affine<Fq2U> p = rnd_pt(pt_seed(s, t));
        h2_madd(acc, H2Affine{h ? p.x.c1 : p.x.c0, h ? p.y.c1 : p.y.c0}, pt_neg(t), h);
    }
    h2_st_xyzz(out + s, acc, h);
}
template <bool LAZY>
__global__ void __launch_bounds__(64, 1) k_madd_one(uint32_t *out, uint32_t seed) {
    XYZZ<Fq2U> acc = XYZZ<Fq2U>{Fq2U::one(), Fq2U::one(), Fq2U::one(), Fq2U::one()};
    Affine<Fq2U> q = rnd_pt(seed);
    for (int it = 0; it < ITERS; it++) { if (LAZY) xyzz_madd_lazy(acc, q, (it & 1) != 0); else xyzz_madd(acc, q, (it & 1) != 0); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((uint32_t *)&acc)[0] ^ seed;
}
template <int MINB>
__global__ void __launch_bounds__(64, MINB) k_madd_split(uint32_t *out, uint32_t seed) {
    const int h = threadIdx.x & 1;
    const Affine<Fq2U> p = rnd_pt(seed);
    H2XYZZ acc{FqU::one(), FqU::one(), h2_one(h), h2_one(h)};
    const H2Affine q{h ? p.x.c1 : p.x.c0, h ? p.y.c1 : p.y.c0};
    for (int it = 0; it < ITERS; it++) h2_madd(acc, q, (it & 1) != 0, h);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x.l[0] ^ seed;
}

template <class K>
static int run(const char *name, K kernel, double adds_per_thread, int blocks_per_cu, uint32_t *d_out) {
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, 0, d_out, 12345u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    const int reps = 5;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, 0, d_out, 12345u + i);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s waves/SIMD launched=%2d  %8.3f ms  %7.3f G additions/s\n", name, blocks_per_cu / 4, ms / reps, adds_per_thread * grid * 64.0 * reps / (ms * 1e-3) * 1e-9);
    return 0;
}

static bool same_point(const XYZZ<Fq2U> &a, const XYZZ<Fq2U> &b) {
    const XYZZ<Fq2> A{to_sat(a.x), to_sat(a.y), to_sat(a.zz), to_sat(a.zzz)}, B{to_sat(b.x), to_sat(b.y), to_sat(b.zz), to_sat(b.zzz)};
    const bool ia = A.zz.is_zero(), ib = B.zz.is_zero();
    if (ia || ib) return ia == ib;
    auto eq = [](const Fq2 &x, const Fq2 &y) { return memcmp(&x, &y, sizeof x) == 0; };
    return eq(f_mul(A.x, B.zz), f_mul(B.x, A.zz)) && eq(f_mul(A.y, B.zzz), f_mul(B.y, A.zzz));
}

int main() {
    const int n = 4096;
    XYZZ<Fq2U> *d_a, *d_b;
    CHK(hipMalloc(&d_a, n * sizeof(XYZZ<Fq2U>)));
    CHK(hipMalloc(&d_b, n * sizeof(XYZZ<Fq2U>)));
    std::vector<XYZZ<Fq2U>> ha(n), hb(n);
    int bad = 0, infs = 0;
    for (int steps : {1, 2, 3, 4, 5, 12, 23}) {
        hipLaunchKernelGGL(k_check_one, dim3(n / 64), dim3(64), 0, 0, d_a, steps);
        hipLaunchKernelGGL(k_check_split, dim3(2 * n / 64), dim3(64), 0, 0, d_b, steps);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(ha.data(), d_a, n * sizeof(XYZZ<Fq2U>), hipMemcpyDeviceToHost));
        CHK(hipMemcpy(hb.data(), d_b, n * sizeof(XYZZ<Fq2U>), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) {
            if (!same_point(ha[i], hb[i])) bad++;
            if (ha[i].zz.is_zero()) infs++;
        }
        printf("steps %2d: mismatches so far %d (infinities seen %d)\n", steps, bad, infs);
    }
    printf("split == one-lane on %d sequences x 7 lengths: %s\n", n, bad ? "FAILED" : "ok");
    uint32_t *d_out;
    CHK(hipMalloc(&d_out, 256 * 16 * 64 * sizeof(uint32_t)));
    for (int b : {4, 8}) {
        run("G2 madd one lane, Karatsuba", k_madd_one<false>, (double)ITERS, b, d_out);
        run("G2 madd one lane, one-reduction", k_madd_one<true>, (double)ITERS, b, d_out);
    }
    for (int b : {4, 8, 12, 16}) {
        run("G2 madd pair-split (bounds 64,2)", k_madd_split<2>, ITERS / 2.0, b, d_out);
        run("G2 madd pair-split (bounds 64,3)", k_madd_split<3>, ITERS / 2.0, b, d_out);
        run("G2 madd pair-split (bounds 64,4)", k_madd_split<4>, ITERS / 2.0, b, d_out);
    }
    return bad ? 1 : 0;
}
