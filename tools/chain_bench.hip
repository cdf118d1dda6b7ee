// chain_bench.hip -- times gemv_chain_kernel (csrc/chain.hpp) on the llama2-7B shapes and prints one wave's timeline.
// Not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -DRAMA_CHAIN_STAMPS -DRAMA_CHAIN_STAMP_BLOCK=17 -o chain_bench chain_bench.hip
#include "../rama_amd/csrc/chain.hpp"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace rama;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int W, int D>
static int run(const char* name, int rows, int K, int nbuf, std::vector<float*>& Wb, float* x, float* o, float* resid) {
    const size_t lds = (size_t)(K + chain_pad_floats(W, D, 4)) * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ChainParams p{}; p.x = x; p.o[0] = o; p.resid = resid; p.K = K; p.rows = rows; p.nmat = 1;
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 24; i++) { p.w[0] = Wb[i % nbuf]; hipLaunchKernelGGL((gemv_chain_kernel<W, D, 4, CEPI_RESID>), dim3((rows + 15) / 16), dim3(W * 64), lds, 0, p); }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    const double us = best * 1e3 / 24, gb = (double)rows * K * 4 / 1e9;
    printf("%s %dx%d W%d D%d: %.2f us per launch, %.0f GB/s\n", name, rows, K, W, D, us, gb / (us * 1e-6));
    unsigned long long st[64]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(rama::g_chain_stamps), sizeof st));
    auto us_of = [&](int i) { return (double)(st[i] - st[0]) * 0.01; };
    printf("   loads issued %.2f | staged+barrier %.2f | first products %.2f | loop done %.2f us;  turns (start-end):", us_of(1), us_of(2), us_of(3), us_of(4));
    for (int c = 0; c < 24; c++) if (st[8 + 2 * c] > st[0] && st[9 + 2 * c] >= st[8 + 2 * c]) printf(" [%.2f-%.2f]", us_of(8 + 2 * c), us_of(9 + 2 * c));
    printf("\n");
    {   // how evenly the row groups start and finish (the last launch): per workgroup, relative to the earliest start
        const int groups = (rows + 15) / 16;
        std::vector<unsigned long long> all(3 * 4096);
        CK(hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(rama::g_chain_all), all.size() * sizeof(unsigned long long)));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < groups && b < 4096; b++) t0 = all[3 * b] < t0 ? all[3 * b] : t0;
        std::vector<double> st_, sg_, dn_;
        for (int b = 0; b < groups && b < 4096; b++) { st_.push_back((all[3 * b] - t0) * 0.01); sg_.push_back((all[3 * b + 1] - t0) * 0.01); dn_.push_back((all[3 * b + 2] - t0) * 0.01); }
        auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
        printf("   all %d groups, us after the first start: start min/med/max %.2f %.2f %.2f | staged %.2f %.2f %.2f | loop done %.2f %.2f %.2f (p90 %.2f, p99 %.2f)\n", groups,
               q(st_, 0), q(st_, .5), q(st_, 1), q(sg_, 0), q(sg_, .5), q(sg_, 1), q(dn_, 0), q(dn_, .5), q(dn_, 1), q(dn_, .9), q(dn_, .99));
    }
    unsigned long long z[64] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(rama::g_chain_stamps), z, sizeof z));
    return 0;
}
int main() {
    const int nbuf = 6;
    const size_t bytes = (size_t)4096 * 11008 * 4;
    std::vector<float*> Wb(nbuf);
    for (auto& q : Wb) { CK(hipMalloc(&q, bytes)); CK(hipMemset(q, 0x3c, bytes)); }
    float *x, *o, *r; CK(hipMalloc(&x, 11008 * 4)); CK(hipMemset(x, 0, 11008 * 4)); CK(hipMalloc(&o, 4096 * 4)); CK(hipMalloc(&r, 4096 * 4)); CK(hipMemset(r, 0, 4096 * 4));
    if (run<4, 32>("wo", 4096, 4096, nbuf, Wb, x, o, r)) return 1;
    if (run<4, 16>("wo", 4096, 4096, nbuf, Wb, x, o, r)) return 1;
    if (run<2, 16>("wo", 4096, 4096, nbuf, Wb, x, o, r)) return 1;
    if (run<4, 32>("w2", 4096, 11008, nbuf, Wb, x, o, r)) return 1;
    if (run<4, 16>("w2", 4096, 11008, nbuf, Wb, x, o, r)) return 1;
    return 0;
}
