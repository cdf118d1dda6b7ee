// fused_stamps.hip -- the one-launch stage (csrc/layer_fused.hpp) at the stories15M / stories110M shapes on zero weights:
// time per launch and the timeline of one layer's phases (first workgroup of each).  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -DRAMA_FUSED_STAMPS -o tools/bin/fused_stamps tools/fused_stamps.hip
#include "../rama_amd/csrc/layer_fused.hpp"
#include <cstdio>
#include <cstdlib>
using namespace rama;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <class T> static T* zalloc(size_t n) { T* p = nullptr; if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) { printf("alloc failed\n"); exit(1); } hipMemset(p, 0, n * sizeof(T)); return p; }
int main(int argc, char** argv) {
    const bool big = argc > 1 && atoi(argv[1]) == 110;
    const int dim = big ? 768 : 288, hidden = big ? 2048 : 768, L = big ? 12 : 6, H = big ? 12 : 6, V = 32000, seq = big ? 1024 : 256, pos = argc > 2 ? atoi(argv[2]) : 100;
    const size_t dd = (size_t)dim * dim, hd = (size_t)hidden * dim;
    FusedParams a{};
    a.dim = dim; a.hidden = hidden; a.n_heads = H; a.seq_len = seq; a.vocab = V; a.n_layers = L; a.do_cls = 1;
    a.wq = zalloc<float>(L * dd); a.wk = zalloc<float>(L * dd); a.wv = zalloc<float>(L * dd); a.wo = zalloc<float>(L * dd);
    a.w1 = zalloc<float>(L * hd); a.w3 = zalloc<float>(L * hd); a.w2 = zalloc<float>(L * hd);
    a.g_att = zalloc<float>((size_t)L * dim); a.g_ffn = zalloc<float>((size_t)L * dim); a.g_final = zalloc<float>(dim); a.wcls = zalloc<float>((size_t)V * dim);
    a.x = zalloc<float>(dim); a.q = zalloc<float>(dim); a.k = zalloc<float>(dim); a.v = zalloc<float>(dim); a.xb = zalloc<float>(dim); a.hb = zalloc<float>(hidden);
    a.logits = zalloc<float>(V); a.kc = zalloc<float>((size_t)L * seq * dim); a.vc = zalloc<float>((size_t)L * seq * dim);
    a.fr = zalloc<float>((size_t)seq * dim); a.fi = zalloc<float>((size_t)seq * dim);
    Ctl* ctl = zalloc<Ctl>(1); Ctl h{}; h.pos = pos; h.token = 1; CK(hipMemcpy(ctl, &h, sizeof h, hipMemcpyHostToDevice)); a.ctl = ctl;
    a.hand = zalloc<tagged_t>((size_t)L * fused_hand_words(dim, hidden)); unsigned* epoch = zalloc<unsigned>(1); { const unsigned one = 1; CK(hipMemcpy(epoch, &one, 4, hipMemcpyHostToDevice)); } a.epoch = epoch; a.err = zalloc<unsigned long long>(1);
    auto wgs = [](int u) { return (u + kPWaves - 1) / kPWaves; };
    a.nA = wgs(3 * (dim / 4)); a.nC = wgs(dim / 4); a.nD = wgs(hidden / 2); a.nE = big ? wgs(dim / 2) : a.nC;
    const int per_layer = a.nA + H + a.nC + a.nD + a.nE, grid = L * per_layer + (big ? wgs(V / 4) : wgs(V / 8));
    size_t lds = (size_t)fused_lds_floats(16, seq, dim, hidden) * 4;
    if (argc > 3 && atoi(argv[3]) > 0) {        // pad the LDS request: fewer workgroups per CU
        lds = (size_t)atoi(argv[3]) * 1024;
        CK(hipFuncSetAttribute((const void*)stage_fused_kernel<16, 4, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void*)stage_fused_kernel<16, 2, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        printf("LDS per workgroup padded to %zu bytes\n", lds);
    }
    float* table = zalloc<float>((size_t)V * dim); a.emb = table;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("dim %d hidden %d layers %d: %d workgroups per layer (A %d, heads %d, C %d, D %d, E %d), grid %d\n", dim, hidden, L, per_layer, a.nA, H, a.nC, a.nD, a.nE, grid);
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        { static unsigned long long z[8][6][8]; CK(hipMemcpyToSymbol(HIP_SYMBOL(rama::g_fused_stamps), z, sizeof z)); }
        for (int i = 0; i < (rep == 2 ? 1 : 50); i++) {
            if (big) hipLaunchKernelGGL((stage_fused_kernel<16, 4, 2, 8>), dim3(grid), dim3(kPThreads), lds, 0, a);
            else hipLaunchKernelGGL((stage_fused_kernel<16, 2, 4, 4>), dim3(grid), dim3(kPThreads), lds, 0, a);
            hipLaunchKernelGGL(fused_epoch_kernel, dim3(1), dim3(1), 0, 0, epoch);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep < 2) printf("stage + epoch: %.2f us per token\n", ms * 1e3 / 50);
    }
    unsigned long long st[8][6][8], err;
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(rama::g_fused_stamps), sizeof st));
    CK(hipMemcpy(&err, a.err, 8, hipMemcpyDeviceToHost));
    printf("error word %llx\n", err);
    const char* ph[] = {"A qkv", "B attn", "C wo", "D w13", "E w2"};
    for (int l = 0; l < 3; l++) {
        const unsigned long long t0 = st[l][0][0];
        printf("layer %d (us since its first workgroup started; first workgroup of each phase)\n", l);
        for (int p = 0; p < 5; p++)
            printf("  %-7s started %8.2f | input in LDS %7.2f | (E: dots done %7.2f) | outputs stored %7.2f | the slowest workgroup's %7.2f\n", ph[p], (double)(long long)(st[l][p][0] - t0) * 0.01,
                   (double)(long long)(st[l][p][1] - t0) * 0.01, (double)(long long)(st[l][p][5] - t0) * 0.01,
                   (double)(long long)(st[l][p][2] - t0) * 0.01, (double)(long long)(st[l][p][3] - t0) * 0.01);
    }
    {
        const unsigned long long t0 = st[0][0][0], tl = st[L - 1 < 8 ? L - 1 : 7][4][3];
        printf("classifier (us since the launch's first workgroup started): first workgroup started %.2f, the last one started %.2f; x in LDS %.2f, its logits stored %.2f, the last workgroup's %.2f",
               (double)(long long)(st[0][5][0] - t0) * 0.01, (double)(long long)(st[0][5][4] - t0) * 0.01, (double)(long long)(st[0][5][1] - t0) * 0.01,
               (double)(long long)(st[0][5][2] - t0) * 0.01, (double)(long long)(st[0][5][3] - t0) * 0.01);
        if (L <= 8) printf("; the last layer's W2 phase ended %.2f", (double)(long long)(tl - t0) * 0.01);
        printf("\n");
    }
    printf("attention of layer 1, head 0 (us after q | k | v were in LDS): timesteps done %.2f, partial results exchanged %.2f, xb stored %.2f\n",
           (double)(long long)(st[1][1][5] - st[1][1][1]) * 0.01, (double)(long long)(st[1][1][6] - st[1][1][1]) * 0.01, (double)(long long)(st[1][1][2] - st[1][1][1]) * 0.01);
    return 0;
}
