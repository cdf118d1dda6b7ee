// topp_bench.hip -- the three launches of the device top-p sampler (csrc/topp_sort.hpp) one by one:
// microseconds per launch (HIP events around back-to-back launches) and where the time goes inside
// each (100 MHz time stamps of workgroup 0).  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRAMA_TOPP_STAMPS -o build/topp_bench tools/topp_bench.hip
#include "../rama_amd/csrc/topp_sort.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace rama;

template <class F>
static double time_us(hipStream_t st, int reps, F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.0 / reps;
}

// [r4] small blocks: statistics once (topp_stats_kernel), BS-entry sorts, the pairs against OB blocks at a time, scatter, pick
template <int BS, int OB, class K3>
static void small_blocks(hipStream_t st, int reps, const ToppSortParams& sq, int n, K3 k3, int* result) {
    ToppStats* stt; CK(hipMalloc(&stt, 64 * sizeof(ToppStats)));
    ToppSortParams s5 = sq; s5.nblk = (n + BS - 1) / BS;
    const int nstat = (n + 1023) / 1024;
    auto j0 = [&] { hipLaunchKernelGGL(topp_stats_kernel, dim3(nstat), dim3(1024), 0, st, s5, stt); };
    auto j1 = [&] { hipLaunchKernelGGL(topp_blocksort_bs_kernel<BS>, dim3(s5.nblk), dim3(BS / 2), 0, st, s5, (const ToppStats*)stt, nstat); };
    auto j2 = [&] { hipLaunchKernelGGL((topp_rank_pairs_bs_kernel<BS, OB>), dim3(s5.nblk, (s5.nblk + OB - 1) / OB), dim3(BS / 2), 0, st, s5); };
    auto j3 = [&] { hipLaunchKernelGGL(topp_rank_scatter_bs_kernel<BS>, dim3((s5.nblk * BS + 1023) / 1024), dim3(1024), 0, st, s5); };
    j0(); j1(); j2(); j3(); k3(); CK(hipStreamSynchronize(st));
    int hr3; CK(hipMemcpy(&hr3, result, 4, hipMemcpyDeviceToHost));
    const double t0_ = time_us(st, reps, j0), t01 = time_us(st, reps, [&] { j0(); j1(); }), t012 = time_us(st, reps, [&] { j0(); j1(); j2(); }),
                 t0123 = time_us(st, reps, [&] { j0(); j1(); j2(); j3(); }), t5 = time_us(st, reps, [&] { j0(); j1(); j2(); j3(); k3(); });
    printf("   blocks of %4d x %d: stats %.1f, + sort %.1f, + pairs %.1f, + scatter %.1f, all five %.1f us (token %d)\n", BS, OB, t0_, t01 - t0_, t012 - t01, t0123 - t012, t5, hr3);
    CK(hipFree(stt));
}

int main(int argc, char** argv) {
    const int n = 32000, reps = argc > 1 ? atoi(argv[1]) : 100;
    hipStream_t st; CK(hipStreamCreate(&st));
    float *logits, *bp, *keys, *prefix; int *bi, *bcount, *vals, *m, *result; unsigned* err;
    CK(hipMalloc(&logits, n * 4)); CK(hipMalloc(&bp, 32768 * 4)); CK(hipMalloc(&bi, 32768 * 4)); CK(hipMalloc(&bcount, 64));
    CK(hipMalloc(&keys, 32768 * 4)); CK(hipMalloc(&vals, 32768 * 4)); CK(hipMalloc(&prefix, 32768 * 4)); CK(hipMalloc(&m, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&result, 4));
    CK(hipMemset(err, 0, 4));
    const double scales[] = {0.05, 1.0, 3.0, 8.0};
    for (double sc : scales) {
        std::mt19937 g(1); std::normal_distribution<float> nd(0.f, (float)sc);
        std::vector<float> h(n); for (auto& v : h) v = nd(g);
        CK(hipMemcpy(logits, h.data(), n * 4, hipMemcpyHostToDevice));
        ToppSortParams sp{}; sp.logits = logits; sp.n = n; sp.temperature = 1.0f; sp.topp = 0.9f; sp.bp = bp; sp.bi = bi; sp.bcount = bcount;
        sp.keys = keys; sp.vals = vals; sp.m = m; sp.err = err; sp.nblk = (n + kToppBlock - 1) / kToppBlock;
        ToppParams tp{}; tp.logits = logits; tp.n = n; tp.temperature = 1.0f; tp.topp = 0.9f; tp.u = 0.27211744f; tp.keys = keys; tp.vals = vals; tp.prefix = nullptr; tp.m = m; tp.err = err;
        ArgmaxParams fin{}; fin.logits = logits; fin.n = n; fin.result = result;
        auto k1 = [&] { hipLaunchKernelGGL(topp_blocksort_kernel<false>, dim3(sp.nblk), dim3(1024), 0, st, sp); };
        auto k2 = [&] { hipLaunchKernelGGL(topp_rank_kernel<kToppMaxBlocks>, dim3(sp.nblk * (kToppBlock / kRankThreads)), dim3(kRankThreads), 0, st, sp); };
        auto k3 = [&] { hipLaunchKernelGGL(topp_pick_scan_kernel, dim3(1), dim3(1024), 0, st, tp, fin); };
        k1(); k2(); k3(); CK(hipStreamSynchronize(st));
        const double t1 = time_us(st, reps, k1), t2 = time_us(st, reps, k2), t3 = time_us(st, reps, k3);
        const double tall = time_us(st, reps, [&] { k1(); k2(); k3(); });
        unsigned long long s[64]; CK(hipMemcpyFromSymbol(s, HIP_SYMBOL(g_topp_stamps), sizeof s));
        int hm, hr; CK(hipMemcpy(&hm, m, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hr, result, 4, hipMemcpyDeviceToHost));
        auto d = [&](int a, int b) { return (double)(long long)(s[b] - s[a]) / 100.0; };
        printf("== std %.2f: kept %d, token %d | blocksort %.1f us, rank %.1f us, pick %.1f us, all three %.1f us\n", sc, hm, hr, t1, t2, t3, tall);
        printf("   blocksort wg0: load+max %.1f  exp+sum %.1f  slice %.1f  sort %.1f\n", d(0, 1), d(1, 2), d(2, 3), d(3, 4));
        printf("   rank wg0:      stage %.1f  search %.1f\n", d(8, 9), d(9, 10));
        printf("   pick:          load %.1f  ripple %.1f  rounds %.1f  tail %.1f  | per round:", d(16, 17), d(17, 18), d(18, 19), d(19, 20));
        unsigned long long prev = s[18];
        for (int r = 0; r < 31 && s[24 + r] > s[18] && s[24 + r] <= s[19]; r++) { printf(" %.1f", (double)(long long)(s[24 + r] - prev) / 100.0); prev = s[24 + r]; }
        printf("\n");
        {   // round 4: the ranking as (block, block) pairs over the whole chip + a scatter launch
            int* racc; CK(hipMalloc(&racc, 32768 * 4));
            ToppSortParams sq = sp; sq.racc = racc;
            auto k1p = [&] { hipLaunchKernelGGL(topp_blocksort_kernel<false>, dim3(sp.nblk), dim3(1024), 0, st, sq); };
            auto k2a = [&] { hipLaunchKernelGGL(topp_rank_pairs_kernel, dim3(sp.nblk, sp.nblk), dim3(1024), 0, st, sq); };
            auto k2b = [&] { hipLaunchKernelGGL(topp_rank_scatter_kernel, dim3(sp.nblk * 2), dim3(1024), 0, st, sq); };
            k1p(); k2a(); k2b(); k3(); CK(hipStreamSynchronize(st));
            int hr2; CK(hipMemcpy(&hr2, result, 4, hipMemcpyDeviceToHost));
            const double ta = time_us(st, reps, [&] { k1p(); k2a(); }) - time_us(st, reps, k1p), tb = time_us(st, reps, k2b), tall4 = time_us(st, reps, [&] { k1p(); k2a(); k2b(); k3(); });
            printf("   ranking by pairs: pairs %.1f us (behind the block sort), scatter %.1f us, all four %.1f us (token %d)\n", ta, tb, tall4, hr2);
            small_blocks<512, 8>(st, reps, sq, n, k3, result);
            small_blocks<512, 4>(st, reps, sq, n, k3, result);
            small_blocks<1024, 4>(st, reps, sq, n, k3, result);
            small_blocks<1024, 2>(st, reps, sq, n, k3, result);
            small_blocks<1024, 1>(st, reps, sq, n, k3, result);
            CK(hipFree(racc));
        }
    }
    return 0;
}
