// qkv_bench.hip -- why does the chain-order Wq|Wk|Wv launch take 35 us on one box and 45-48 on another, when the fast path's takes 32.5 everywhere?
// Times gemv_chain_kernel at the llama2-7B shape over layouts of the three matrices and epilogues.  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -Irama_amd/csrc -Iinclude -o tools/bin/qkv_bench tools/qkv_bench.hip
#include "../rama_amd/csrc/chain.hpp"
#include <cstdio>
#include <vector>
using namespace rama;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Bufs { float* chain; float *x, *o, *fr, *kc, *vc, *resid, *gain; unsigned long long *slot, *err; unsigned* epoch; };
__global__ void bump(unsigned* e) { *e = *e + 1u; }
constexpr int L = 32, DIM = 4096;
constexpr size_t DD = (size_t)DIM * DIM;

// layout 0: [m][l] (the product's: the copies of wq, wk, wv are L x DD floats apart); 1: [l][m]
template <int W, int D, int EPI, int NORM = CNORM_NONE, int LR = 64>
static int run(const char* what, const Bufs& b, int layout, int nmat, int rows) {
    size_t lds = (size_t)(DIM + chain_pad_floats(W, D, 4)) * 4;
    if (NORM == CNORM_LEAD && lds < sizeof(FastSumShared<W>)) lds = sizeof(FastSumShared<W>);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ChainParams p{}; p.x = b.x; p.o[0] = b.o; p.o[1] = b.o + 16384; p.o[2] = b.o + 32768; p.resid = b.resid; p.K = DIM; p.rows = rows; p.nmat = nmat;
    p.fr = b.fr; p.fi = b.fr; p.head_size = 128; p.kc = b.kc; p.vc = b.vc; p.pos_val = 5;
    if (NORM != CNORM_NONE) p.nw = b.gain;
    p.lead = b.slot; p.epoch = b.epoch; p.err = b.err;
    const int groups = nmat * ((rows + 15) / 16);
    float best = 1e9, worst = 0;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, 0));
        for (int l = 0; l < L; l++) {
            for (int m = 0; m < 3; m++) p.w[m] = layout == 0 ? b.chain + ((size_t)m * L + l) * DD : b.chain + ((size_t)l * 3 + m) * DD;
            hipLaunchKernelGGL((gemv_chain_kernel<W, D, 4, EPI, NORM, LR>), dim3(groups + (NORM == CNORM_LEAD ? 1 : 0)), dim3(W * 64), lds, 0, p);
            if (NORM == CNORM_LEAD) hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, 0, b.epoch);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; worst = ms > worst ? ms : worst;
    }
    const double us = best * 1e3 / L, gb = (double)nmat * rows * DIM * 4 / 1e9;
    printf("%-58s W%d D%d: %.2f us per launch (worst rep %.2f), %.0f GB/s\n", what, W, D, us, worst * 1e3 / L, gb / (us * 1e-6));
    { unsigned long long e; CK(hipMemcpy(&e, b.err, 8, hipMemcpyDeviceToHost)); if (e) printf("   ERROR WORD %llx\n", e); }
    return 0;
}

int main() {
    Bufs b{};
    CK(hipMalloc(&b.chain, 3 * L * DD * 4)); CK(hipMemset(b.chain, 0x3c, 3 * L * DD * 4));
    CK(hipMalloc(&b.x, DIM * 4)); CK(hipMemset(b.x, 0, DIM * 4)); CK(hipMalloc(&b.o, 65536 * 4)); CK(hipMalloc(&b.resid, 16384 * 4)); CK(hipMemset(b.resid, 0, 16384 * 4));
    CK(hipMalloc(&b.fr, 2048 * 64 * 4)); CK(hipMemset(b.fr, 0, 2048 * 64 * 4));
    CK(hipMalloc(&b.kc, (size_t)2048 * DIM * 4)); CK(hipMalloc(&b.vc, (size_t)2048 * DIM * 4));
    { std::vector<float> g(DIM, 1.0f), xx(DIM); for (int i = 0; i < DIM; i++) xx[i] = (float)((i * 37) % 101) * 0.02f - 1.0f;
      CK(hipMalloc(&b.gain, DIM * 4)); CK(hipMemcpy(b.gain, g.data(), DIM * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b.x, xx.data(), DIM * 4, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&b.slot, 256)); CK(hipMemset(b.slot, 0, 256)); CK(hipMalloc(&b.err, 8)); CK(hipMemset(b.err, 0, 8));
    CK(hipMalloc(&b.epoch, 4)); { unsigned one = 1; CK(hipMemcpy(b.epoch, &one, 4, hipMemcpyHostToDevice)); }
    for (int pass = 0; pass < 2; pass++) {
        if (run<2, 16, CEPI_QKV, CNORM_LEAD, 32>("3 matrices, QKV epilogue, leader norm (+ a 1-thread launch per launch)", b, 0, 3, DIM)) return 1;
        if (run<2, 16, CEPI_QKV, CNORM_TREE>("3 matrices, QKV epilogue, tree norm in every workgroup", b, 0, 3, DIM)) return 1;
        if (run<2, 16, CEPI_STORE, CNORM_TREE>("3 matrices, plain store, tree norm in every workgroup", b, 0, 3, DIM)) return 1;
        if (run<1, 16, CEPI_SWIGLU, CNORM_LEAD, 64>("W1|W3 rows = 3 dim, SwiGLU epilogue, leader norm (+ 1-thread launch)", b, 1, 1, 3 * DIM)) return 1;
        if (run<1, 16, CEPI_SWIGLU>("W1|W3 rows = 3 dim, SwiGLU epilogue", b, 1, 1, 3 * DIM)) return 1;
        if (run<2, 16, CEPI_QKV>("3 matrices, copies L x dim x dim apart (the product), QKV epilogue", b, 0, 3, DIM)) return 1;
        if (run<2, 16, CEPI_QKV>("3 matrices, a layer's copies adjacent, QKV epilogue", b, 1, 3, DIM)) return 1;
        if (run<2, 16, CEPI_STORE>("3 matrices, copies L x dim x dim apart, plain store", b, 0, 3, DIM)) return 1;
        if (run<2, 16, CEPI_STORE>("ONE matrix of 3 dim rows (a layer's copies adjacent), plain store", b, 1, 1, 3 * DIM)) return 1;
        if (run<1, 16, CEPI_STORE>("ONE matrix of 3 dim rows, plain store", b, 1, 1, 3 * DIM)) return 1;
        if (run<1, 16, CEPI_QKV>("3 matrices, copies apart, QKV epilogue", b, 0, 3, DIM)) return 1;
        if (run<2, 16, CEPI_RESID>("one matrix of dim rows (Wo), residual epilogue", b, 0, 1, DIM)) return 1;
    }
    return 0;
}
