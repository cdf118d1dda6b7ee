// fv_bench.hip -- parity mode's attention at a long context (the spread form: attn_scores_chain_kernel + attn_softmax_values_chain_kernel), llama2-7B's
// 32 heads x 128 at position argv[1] (default 1900): time per launch, back to back and alternating, and where workgroup 0 of the second launch spends
// its time.  Not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -DRAMA_SEQ_STAMPS -Irama_amd/csrc -Iinclude -o tools/bin/fv_bench tools/fv_bench.hip
#include "../rama_amd/csrc/chain.hpp"
#include <cstdio>
#include <vector>
#include <cstring>
#include <random>
using namespace rama;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int pos = argc > 1 ? atoi(argv[1]) : 1900, H = 32, hs = 128, dim = H * hs, seq = 2048;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    float *q, *kc, *vc, *xb, *att, *sc;
    CK(hipMalloc(&q, dim * 4)); CK(hipMalloc(&xb, dim * 4)); CK(hipMalloc(&kc, (size_t)seq * dim * 4)); CK(hipMalloc(&vc, (size_t)seq * dim * 4));
    CK(hipMalloc(&att, (size_t)H * seq * 4)); CK(hipMalloc(&sc, (size_t)H * seq * 4));
    std::vector<float> h((size_t)seq * dim);
    for (auto& v : h) v = nd(rng) * 0.3f;
    CK(hipMemcpy(kc, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(vc, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, h.data() + 777, dim * 4, hipMemcpyHostToDevice));
    RefAttnParams ap{}; ap.q = q; ap.kc = kc; ap.vc = vc; ap.xb = xb; ap.att = att; ap.sc = sc; ap.pos_val = pos; ap.dim = dim; ap.head_size = hs; ap.seq_len = seq;
    const size_t fv_lds = attn_fused_values_lds_floats(seq) * 4;
    const int ngroups = (seq + 63) / 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto scores = [&] { hipLaunchKernelGGL(attn_scores_chain_kernel, dim3(H, ngroups), dim3(64), 0, 0, ap); };
    auto fv = [&] { hipLaunchKernelGGL(attn_softmax_values_chain_kernel, dim3(H, hs / kValCols), dim3(kFvSoftWaves * 64), fv_lds, 0, ap); };
    for (int what = 0; what < 3; what++) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 50; i++) { if (what != 1) scores(); if (what != 0) fv(); }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("pos %d %s: %.2f us\n", pos, what == 0 ? "scores" : (what == 1 ? "softmax + values" : "scores, softmax + values"), best * 1e3 / 50);
    }
    unsigned long long sv[64];
    CK(hipMemcpyFromSymbol(sv, HIP_SYMBOL(rama::g_seq_stamps), sizeof sv));
    printf("   softmax + values, workgroup 0 after its start: loads + scores staged %.2f | max %.2f | exp %.2f | sum %.2f | divide %.2f | chain through tile k:",
           (sv[41] - sv[40]) * 0.01, (sv[42] - sv[40]) * 0.01, (sv[43] - sv[40]) * 0.01, (sv[44] - sv[40]) * 0.01, (sv[45] - sv[40]) * 0.01);
    for (int t = 0; t * kFvRows <= pos && t < 12; t++) printf(" %.2f", (sv[46 + t] - sv[40]) * 0.01);
    printf(" | end %.2f us\n", (sv[59] - sv[40]) * 0.01);
    return 0;
}
