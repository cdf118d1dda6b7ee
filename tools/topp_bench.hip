// topp_bench.hip -- the three launches of the device top-p sampler (csrc/topp_sort.hpp) one by one:
// microseconds per launch (HIP events around back-to-back launches) and where the time goes inside
// each (100 MHz time stamps of workgroup 0).  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRAMA_TOPP_STAMPS -o build/topp_bench tools/topp_bench.hip
#include "../rama_amd/csrc/topp_pick.hpp"
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

// [r4] small blocks: statistics once (topp_stats_kernel), BS-entry sorts, the pairs against OB blocks at a time, scatter, then the
// running sums by one workgroup's scan rounds (topp_pick_scan_kernel) or by 32 workgroups in one launch (topp_pick_dist_kernel)
template <int BS, int OB, class K3>
static void small_blocks(hipStream_t st, int reps, const ToppSortParams& sq, const ToppParams& tp, const ArgmaxParams& fin, int n, K3 k3, int* result) {
    ToppStats* stt; CK(hipMalloc(&stt, 64 * sizeof(ToppStats)));
    ToppSortParams s5 = sq; s5.nblk = (n + BS - 1) / BS;
    CK(hipMalloc(&s5.rk, 32768 * 8)); CK(hipMalloc(&s5.bm, 32768 * 8)); CK(hipMalloc(&s5.approx, 32768 * 4));
    const size_t items = sizeof(PickItem) * kPickChunk * kPickMaxChunks, bytes = items + 8 * kPickMaxChunks + 32;
    char* blob; CK(hipMalloc(&blob, bytes)); CK(hipMemset(blob, 0, bytes));
    ToppDistParams d{}; d.approx = s5.approx; d.items = (PickItem*)blob; d.hdr = (unsigned long long*)(blob + items); d.cross = d.hdr + kPickMaxChunks;
    d.epoch = (const unsigned*)(d.cross + 2); d.bad = (unsigned*)(d.cross + 2) + 1;
    s5.epoch = (unsigned*)d.epoch;
    const int nstat = (n + 1023) / 1024;
    auto j0 = [&] { hipLaunchKernelGGL(topp_stats_kernel, dim3(nstat), dim3(1024), 0, st, s5, stt); };
    auto j1 = [&] { hipLaunchKernelGGL(topp_blocksort_bs_kernel<BS>, dim3(s5.nblk), dim3(BS), 0, st, s5, (const ToppStats*)stt, nstat); };
    auto j2 = [&] { hipLaunchKernelGGL((topp_rank_pairs_bs_kernel<BS, OB>), dim3(s5.nblk, (s5.nblk + OB - 1) / OB), dim3(BS / 2), 0, st, s5); };
    auto j3 = [&] { hipLaunchKernelGGL(topp_rank_scatter_bs_kernel<BS>, dim3((s5.nblk * BS + 1023) / 1024), dim3(1024), 0, st, s5); };
    auto j4 = [&] { hipLaunchKernelGGL(topp_pick_dist_kernel, dim3(kPickMaxChunks), dim3(1024), 0, st, tp, d, fin); };
    j0(); j1(); j2(); j3(); k3(); CK(hipStreamSynchronize(st));
    int hr3; CK(hipMemcpy(&hr3, result, 4, hipMemcpyDeviceToHost));
    j0(); j1(); j2(); j3(); j4(); CK(hipStreamSynchronize(st));
    int hr4; CK(hipMemcpy(&hr4, result, 4, hipMemcpyDeviceToHost));
    const double t0_ = time_us(st, reps, j0), t01 = time_us(st, reps, [&] { j0(); j1(); }), t012 = time_us(st, reps, [&] { j0(); j1(); j2(); }),
                 t0123 = time_us(st, reps, [&] { j0(); j1(); j2(); j3(); }), t5 = time_us(st, reps, [&] { j0(); j1(); j2(); j3(); k3(); }),
                 t6 = time_us(st, reps, [&] { j0(); j1(); j2(); j3(); j4(); });
    unsigned bad; CK(hipMemcpy(&bad, d.bad, 4, hipMemcpyDeviceToHost));
    printf("   blocks of %4d x %d: stats %.1f, + sort %.1f, + pairs %.1f, + scatter %.1f | + scan pick: %.1f us (token %d) | + dist pick %.1f: %.1f us (token %d, bad %u)\n",
           BS, OB, t0_, t01 - t0_, t012 - t01, t0123 - t012, t5, hr3, t6 - t0123, t6, hr4, bad);
    unsigned long long ps[kPickMaxChunks][12]; CK(hipMemcpyFromSymbol(ps, HIP_SYMBOL(g_pick_stamps), sizeof ps));
    unsigned hdr[2 * kPickMaxChunks]; CK(hipMemcpy(hdr, d.hdr, sizeof hdr, hipMemcpyDeviceToHost));
    int ni = 0; for (int c = 0; c < kPickMaxChunks; c++) ni += (int)hdr[2 * c];
    printf("      items per chunk:"); for (int c = 0; c < kPickMaxChunks; c++) printf(" %u", hdr[2 * c]); printf("\n");
    {   // kinds of chunk 0's and chunk 16's items
        std::vector<unsigned long long> it(2 * kPickChunk * kPickMaxChunks); CK(hipMemcpy(it.data(), d.items, it.size() * 8, hipMemcpyDeviceToHost));
        for (int c : {0, 15, 16}) { printf("      chunk %d kinds:", c); for (unsigned q = 0; q < hdr[2 * c] && q < 80; q++) printf(" %llu", (it[2 * ((size_t)c * kPickChunk + q)] >> 32) & 0xFF); printf("\n"); }
    }
    printf("      dist pick, items %d; workgroup: load classify+scan+publish headers items+walk own-walk sums+cross finish (us)\n", ni);
    for (int c : {0, 1, 8, 16, 24, 28, 31}) {
        printf("      wg %2d:", c);
        for (int q = 1; q < 7; q++) printf(" %5.1f", ps[c][q] > ps[c][q - 1] ? (double)(long long)(ps[c][q] - ps[c][q - 1]) / 100.0 : 0.0);
        printf("   (ripples %.2f;", ps[c][9] > ps[c][8] ? (double)(long long)(ps[c][9] - ps[c][8]) / 100.0 : 0.0);
        printf(" items fetched %.1f after the headers; from wg 0's start %.1f)\n", ps[c][7] > ps[c][2] ? (double)(long long)(ps[c][7] - ps[c][2]) / 100.0 : 0.0, (double)(long long)(ps[c][0] - ps[0][0]) / 100.0);
    }
    CK(hipFree(stt)); CK(hipFree(s5.rk)); CK(hipFree(s5.bm)); CK(hipFree(s5.approx)); CK(hipFree(blob));
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
            small_blocks<1024, 2>(st, reps, sq, tp, fin, n, k3, result);
            if (getenv("TOPP_BENCH_SWEEP")) { small_blocks<1024, 1>(st, reps, sq, tp, fin, n, k3, result); small_blocks<1024, 4>(st, reps, sq, tp, fin, n, k3, result); small_blocks<512, 1>(st, reps, sq, tp, fin, n, k3, result); small_blocks<512, 2>(st, reps, sq, tp, fin, n, k3, result); }
            CK(hipFree(racc));
        }
    }
    return 0;
}
