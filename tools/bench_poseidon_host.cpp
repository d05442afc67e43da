#include <atomic>
#include <chrono>
#include <cstdio>
#include <vector>
#include "hostff.hpp"
using namespace zk;
using zk::h64::Fr64;
namespace {
#include "poseidon_params.inc"
constexpr int P_ROUNDS = POSEIDON_FULL + POSEIDON_PARTIAL, P_HALF = POSEIDON_FULL / 2;
typedef unsigned __int128 u128;
#include "poseidon_h64.inc"
}
int main() {
    const PoseidonH &pp = pparams();
    Fr64 st[3] = {Fr64::one(), Fr64::r2(), Fr64::one()};
    const int N = 20000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) permute64(st, pp);
    auto t1 = std::chrono::steady_clock::now();
    printf("%016lx  %.2f us per permutation\n", st[0].l[0], std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    // pieces
    Fr64 x = st[1];
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N * 50; i++) x = pow17(x);
    t1 = std::chrono::steady_clock::now();
    printf("%016lx  %.1f ns per pow17 (dependent)\n", x.l[0], std::chrono::duration<double, std::nano>(t1 - t0).count() / (N * 50));
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N * 50; i++) { Fr64 n0 = dot3(pp.mds[0], st), n1 = dot3(pp.mds[1], st), n2 = dot3(pp.mds[2], st); st[0] = n0; st[1] = n1; st[2] = n2; }
    t1 = std::chrono::steady_clock::now();
    printf("%016lx  %.1f ns per 3x dot3\n", st[0].l[0], std::chrono::duration<double, std::nano>(t1 - t0).count() / (N * 50));
}
struct Extra { Extra() {
    const PoseidonH &pp = pparams();
    printf("sparse_ok=%d\n", (int)pp.sparse_ok);
    Fr64 st[3] = {Fr64::one(), Fr64::r2(), Fr64::one()};
    const int N = 400000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) full_round64(st, pp, i & 3);
    auto t1 = std::chrono::steady_clock::now();
    printf("%016lx  %.1f ns per full round\n", st[0].l[0], std::chrono::duration<double, std::nano>(t1 - t0).count() / N);
    t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < N / 29; k++)
    for (int i = 0; i < POSEIDON_PARTIAL; i++) {
        Fr64 s[3];
        s[1] = h64::add(st[1], pp.pc[i][1]);
        s[2] = h64::add(st[2], pp.pc[i][2]);
        s[0] = pow17(h64::add(st[0], pp.pc[i][0]));
        st[0] = dot3(pp.prow[i], s);
        st[1] = h64::add(h64::mul(pp.pv[i][0], s[0]), s[1]);
        st[2] = h64::add(h64::mul(pp.pv[i][1], s[0]), s[2]);
    }
    t1 = std::chrono::steady_clock::now();
    printf("%016lx  %.1f ns per sparse partial round (dot3 after)\n", st[0].l[0], std::chrono::duration<double, std::nano>(t1 - t0).count() / (N / 29 * 29));
    t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < N / 29; k++)
    for (int i = 0; i < POSEIDON_PARTIAL; i++) {
        Fr64 s[3];
        s[1] = h64::add(st[1], pp.pc[i][1]);
        s[2] = h64::add(st[2], pp.pc[i][2]);
        const Wide part = wide_dot<2>(pp.prow[i] + 1, s + 1);
        s[0] = pow17(h64::add(st[0], pp.pc[i][0]));
        st[0] = wide_reduce(wide_add(part, wide_dot<1>(pp.prow[i], s)));
        st[1] = h64::add(h64::mul(pp.pv[i][0], s[0]), s[1]);
        st[2] = h64::add(h64::mul(pp.pv[i][1], s[0]), s[2]);
    }
    t1 = std::chrono::steady_clock::now();
    printf("%016lx  %.1f ns per sparse partial round (partial sums first)\n", st[0].l[0], std::chrono::duration<double, std::nano>(t1 - t0).count() / (N / 29 * 29));
} } extra;
