// fastsum_stamps.hip -- where the time of seqsum_fast.hpp's exact sum goes (100 MHz stamps of thread 0).  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -DRAMA_FS_STAMPS -o tools/bin/fastsum_stamps tools/fastsum_stamps.hip
#include "../rama_amd/csrc/seqsum_fast.hpp"
#include <cstdio>
#include <vector>
#include <random>
using namespace rama;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int NW, int R> int run(const float* da, int n, float* dout) {
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((seqsum_fast_test_kernel<NW, R>), dim3(1), dim3(NW * 64), 0, 0, da, n, dout);
    CK(hipDeviceSynchronize());
    unsigned long long st[16]; float out[4];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(rama::g_fs_stamps), sizeof st));
    CK(hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost));
    printf("waves %d R %d: total %.2f us (items %d, held %g): loads->0 ?, estimate %.2f | groups %.2f | places+list %.2f | walk %.2f | check %.2f\n", NW, R, out[3] * 0.01, (int)out[2], out[1],
           (st[1] - st[0]) * 0.01, (st[2] - st[1]) * 0.01, (st[3] - st[2]) * 0.01, (st[4] - st[3]) * 0.01, (st[5] - st[4]) * 0.01);
    return 0;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> x(n);
    for (auto& v : x) { v = nd(rng); v = v * v; }
    float *da, *dout;
    CK(hipMalloc(&da, n * 4)); CK(hipMalloc(&dout, 64));
    CK(hipMemcpy(da, x.data(), n * 4, hipMemcpyHostToDevice));
    if (n <= 4096) { if (run<1, 64>(da, n, dout)) return 1; if (run<2, 32>(da, n, dout)) return 1; if (run<4, 16>(da, n, dout)) return 1; }
    return 0;
}
