// seqsum_bench.hip -- times rmsnorm_chain_kernel (csrc/chain.hpp: the exact sequential sum of squares) and prints where
// its time goes.  Not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -DRAMA_SEQ_STAMPS -o seqsum_bench seqsum_bench.hip
#include "../rama_amd/csrc/chain.hpp"
#include <cstdio>
#include <vector>
#include <random>
using namespace rama;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> x(n), w(n, 1.0f);
    for (auto& v : x) v = nd(rng);
    float *dx, *dw, *dout;
    CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dw, n * 4)); CK(hipMalloc(&dout, n * 4));
    CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), n * 4, hipMemcpyHostToDevice));
    const size_t lds = ((size_t)n + (n >> 5) + 2) * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(rmsnorm_chain_kernel, dim3(1), dim3(kNormThreads), lds, 0, dout, dx, dw, n, (float*)nullptr, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("n=%d rmsnorm_chain_kernel: %.2f us per launch (back to back)\n", n, ms * 1e3 / 200);
    }
    unsigned long long st[16];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(rama::g_seq_stamps), sizeof st));
    const char* names[] = {"load+squares+barrier", "approx prefix", "classify", "maps", "seg scan + barrier", "walk + barrier", "scale + store"};
    for (int i = 0; i < 6; i++) printf("  %-22s %6.2f us\n", names[i], (double)(st[i + 1] - st[i]) * 0.01);
    printf("  debug: nseq %llu smask %016llx %016llx %016llx %016llx head %g\n", st[8], st[9], st[10], st[11], st[12], *(float*)&st[15]);
    {   // the parity-mode attention at a long context: 32 heads x 128, position argv[2] (default 1900)
        const int pos = argc > 2 ? atoi(argv[2]) : 1900, H = 32, hs = 128, dim = H * hs, seq = 2048;
        float *q, *kc, *vc, *xb;
        CK(hipMalloc(&q, dim * 4)); CK(hipMalloc(&xb, dim * 4)); CK(hipMalloc(&kc, (size_t)seq * dim * 4)); CK(hipMalloc(&vc, (size_t)seq * dim * 4));
        std::vector<float> h((size_t)seq * dim);
        for (auto& v : h) v = nd(rng);
        CK(hipMemcpy(kc, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(vc, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(q, h.data(), dim * 4, hipMemcpyHostToDevice));
        RefAttnParams ap{}; ap.q = q; ap.kc = kc; ap.vc = vc; ap.xb = xb; ap.pos_val = pos; ap.dim = dim; ap.head_size = hs; ap.seq_len = seq;
        CK(hipFuncSetAttribute((const void*)attention_chain_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
        CK(hipFuncSetAttribute((const void*)attention_chain_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
        for (int nw : {4, 8}) {
            const size_t alds = attn_chain_lds_floats(hs, seq, nw) * 4 + 16;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 20; i++) {
                    if (nw == 4) hipLaunchKernelGGL((attention_chain_kernel<4>), dim3(H), dim3(256), alds, 0, ap);
                    else hipLaunchKernelGGL((attention_chain_kernel<8>), dim3(H), dim3(512), alds, 0, ap);
                }
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("attention_chain_kernel<%d> pos %d: %.2f us per launch\n", nw, pos, ms * 1e3 / 20);
            }
            CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(rama::g_seq_stamps), sizeof st));
            printf("   q %.2f | scores %.2f | max+exp %.2f | sum %.2f | divide %.2f | values %.2f us\n", (st[9] - st[8]) * 0.01 * 0 + 0.0, (st[9] - st[8]) * 0.01, (st[10] - st[9]) * 0.01,
                   (st[11] - st[10]) * 0.01, (st[12] - st[11]) * 0.01, (st[13] - st[12]) * 0.01);
        }
        {   // the spread form of long contexts: the value chains (heads x 32-column slices), stamps of workgroup 0
            float* att; CK(hipMalloc(&att, (size_t)H * seq * 4)); CK(hipMemset(att, 0, (size_t)H * seq * 4));
            ap.att = att;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 20; i++) hipLaunchKernelGGL(attn_values_chain_kernel, dim3(H, hs / kValCols), dim3(kValWaves * 64), 0, 0, ap);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("attn_values_chain_kernel pos %d: %.2f us per launch\n", pos, ms * 1e3 / 20);
            }
            unsigned long long sv[40];
            CK(hipMemcpyFromSymbol(sv, HIP_SYMBOL(rama::g_seq_stamps), sizeof sv));
            printf("   loads issued at 0; the chain wave is through tile k at:");
            for (int t = 0; t < 8; t++) printf(" %.2f", (sv[21 + t] - sv[20]) * 0.01);
            printf(" end %.2f us\n", (sv[33] - sv[20]) * 0.01);
        }
    }
    unsigned ps[2]; CK(hipMemcpyFromSymbol(ps, HIP_SYMBOL(rama::g_pred_stats), sizeof ps));
    printf("  held %u fell back %u\n", ps[0], ps[1]);
    return 0;
}
