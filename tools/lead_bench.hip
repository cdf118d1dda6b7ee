// lead_bench.hip -- gemv_chain_kernel with the leader-workgroup norm (CNORM_LEAD) against the plain launch + rmsnorm_chain_kernel, on the
// llama2-7B shapes, with the leader's and one consumer's timeline.  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -DRAMA_CHAIN_STAMPS -DRAMA_CHAIN_STAMP_BLOCK=700 -DRAMA_FS_STAMPS -o tools/bin/lead_bench tools/lead_bench.hip
#include "../rama_amd/csrc/chain.hpp"
#include <cstdio>
#include <vector>
#include <random>
using namespace rama;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void bump(unsigned* e) { *e = *e + 1u; }
template <int W, int LR, int EPI>
static int run(const char* name, int rows, int nmat, int K, int nbuf, std::vector<float*>& Wb, float* x, float* gain, float* o, unsigned long long* slot, unsigned* epoch, unsigned long long* err) {
    const size_t lds0 = (size_t)(K + chain_pad_floats(W, 16, 4)) * 4;
    const size_t lds = lds0 > sizeof(FastSumShared<W>) ? lds0 : sizeof(FastSumShared<W>);
    const int groups = nmat * ((rows + 15) / 16);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ChainParams p{}; p.x = x; p.nw = gain; p.o[0] = o; p.o[1] = o + 16384; p.o[2] = o + 32768; p.K = K; p.rows = rows; p.nmat = nmat;
    p.fr = gain; p.fi = gain; p.head_size = 128; p.kc = o + 65536; p.vc = o + 131072; p.pos_val = 0;
    p.lead = slot; p.epoch = epoch; p.err = err;
    float best = 1e9, best0 = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 24; i++) {
            for (int mm = 0; mm < nmat; mm++) p.w[mm] = Wb[(i + mm) % nbuf];
            hipLaunchKernelGGL((gemv_chain_kernel<W, 16, 4, EPI, CNORM_LEAD, LR>), dim3(groups + 1), dim3(W * 64), lds, 0, p);
            hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, 0, epoch);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        ChainParams q = p; q.nw = nullptr;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 24; i++) {
            for (int mm = 0; mm < nmat; mm++) q.w[mm] = Wb[(i + mm) % nbuf];
            hipLaunchKernelGGL(rmsnorm_chain_kernel, dim3(1), dim3(kNormThreads), ((size_t)K + (K >> 5) + 2) * 4, 0, o + 200000, x, gain, K, (float*)nullptr, 0);
            hipLaunchKernelGGL((gemv_chain_kernel<W, 16, 4, EPI>), dim3(groups), dim3(W * 64), lds0, 0, q);
            hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, 0, epoch);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); best0 = ms < best0 ? ms : best0;
    }
    printf("%s %dx%dx%d W%d: leader norm %.2f us per (launch + bump), norm launch + plain launch + bump %.2f us\n", name, nmat, rows, K, W, best * 1e3 / 24, best0 * 1e3 / 24);
    unsigned long long st[64]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(rama::g_chain_stamps), sizeof st));
    unsigned long long fsst[16]; CK(hipMemcpyFromSymbol(fsst, HIP_SYMBOL(rama::g_fs_stamps), sizeof fsst));
    auto us = [&](unsigned long long a, unsigned long long b) { return (double)((long long)(a - b)) * 0.01; };
    printf("   leader (after its start): x squared %.2f | estimate %.2f | groups %.2f | list %.2f | walk %.2f | check %.2f | published %.2f\n", us(st[41], st[40]), us(fsst[1], st[40]), us(fsst[2], st[40]),
           us(fsst[3], st[40]), us(fsst[4], st[40]), us(fsst[5], st[40]), us(st[42], st[40]));
    printf("   shader clock over the leader's sum: %.0f MHz (walk: %.0f cycles)\n", (double)(fsst[13] - fsst[8]) / ((double)(fsst[5] - fsst[0]) * 0.01), (double)(fsst[12] - fsst[11]));
    printf("   consumer block %d (after the LEADER's start): start %.2f | loads issued %.2f | v seen %.2f | staged %.2f | loop done %.2f\n", RAMA_CHAIN_STAMP_BLOCK, us(st[0], st[40]), us(st[1], st[40]), us(st[5], st[40]),
           us(st[2], st[40]), us(st[4], st[40]));
    unsigned long long e; CK(hipMemcpy(&e, err, 8, hipMemcpyDeviceToHost)); if (e) printf("   ERROR WORD %llx\n", e);
    return 0;
}
int main() {
    const int nbuf = 6;
    const size_t bytes = (size_t)4096 * 11008 * 4 * 2;
    std::vector<float*> Wb(nbuf);
    for (auto& q : Wb) { CK(hipMalloc(&q, bytes)); CK(hipMemset(q, 0x3c, bytes)); }
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hx(4096), hg(4096 * 64, 1.0f); for (auto& v : hx) v = nd(rng);
    float *x, *g, *o; CK(hipMalloc(&x, 4096 * 4)); CK(hipMalloc(&g, hg.size() * 4)); CK(hipMalloc(&o, 400000 * 4));
    CK(hipMemcpy(x, hx.data(), 4096 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    unsigned long long *slot, *err; unsigned* epoch;
    CK(hipMalloc(&slot, 256)); CK(hipMemset(slot, 0, 256)); CK(hipMalloc(&err, 8)); CK(hipMemset(err, 0, 8)); CK(hipMalloc(&epoch, 4));
    { unsigned one = 1; CK(hipMemcpy(epoch, &one, 4, hipMemcpyHostToDevice)); }
    if (run<1, 64, CEPI_SWIGLU>("w1|w3", 22016, 1, 4096, nbuf, Wb, x, g, o, slot, epoch, err)) return 1;
    if (run<2, 32, CEPI_QKV>("wq|wk|wv", 4096, 3, 4096, nbuf, Wb, x, g, o, slot, epoch, err)) return 1;
    return 0;
}
