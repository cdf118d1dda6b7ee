// pf_mfma_bench.hip -- standalone check + microbenchmark of the fp32 MFMA prefill GEMM
// (rama_amd/csrc/prefill_mfma.hpp).  Not part of the product: a tuning aid.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o pf_mfma_bench tools/pf_mfma_bench.hip
// Run:   ./pf_mfma_bench [iters]
// Every variant computes O[p][r] = sum_k W[r][k] X[p][k] for the llama2-7B layer shapes; timing =
// HIP events around `iters` launches rotating over distinct weight buffers (3 GB, so the 256 MiB
// Infinity Cache cannot help); the first launch of each variant is checked against a plain kernel.
#include "../rama_amd/csrc/prefill_mfma.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

using namespace rama;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void ref_gemm(const float* W, const float* X, float* O, int K, int rows, int P) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
    if (r >= rows || p >= P) return;
    double acc = 0.0;
    for (int k = 0; k < K; k++) acc += (double)W[(size_t)r * K + k] * (double)X[(size_t)p * K + k];
    O[(size_t)p * rows + r] = (float)acc;
}
__global__ void fill_kernel(float* d, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned z = (unsigned)i * 2654435761u + seed;
        z ^= z >> 15; z *= 2246822519u; z ^= z >> 13; z *= 3266489917u; z ^= z >> 16;
        d[i] = ((float)(z & 0xFFFF) - 32768.0f) * scale;
    }
}

#define MODE_OF_NAME(n) (strstr((n), "qkv") || strstr((n), "swiglu"))
struct Shape { const char* name; int rows, K; };

enum { MODE_ROWS = 0, MODE_QKV = 1, MODE_SWIGLU = 2 };
template <int PT, int RT, int JN, int LD, int MODE, int STAGGER = 0, int MIX = 0>
static void launch(hipStream_t st, const float* W, const float* XT, float* O, size_t slab, int K, int rows, int P, int cus, int ks) {
    MfParams p{};
    p.x = XT; p.o = O; p.slab_floats = slab; p.o_stride = rows; p.K = K; p.n_tok = P; p.ksplit = ks;
    int groups;
    if (MODE == MODE_ROWS) { p.w[0] = W; p.rows = rows; groups = (rows + 16 * RT - 1) / (16 * RT); }
    else {      // the shape's rows are split evenly over the RT matrices
        p.rows = rows / RT;
        for (int i = 0; i < RT; i++) p.w[i] = W + (size_t)i * p.rows * K;
        groups = (p.rows + 15) / 16;
        p.fr = XT; p.fi = XT; p.head_size = 128; p.kc = O + slab; p.vc = O + 2 * slab; p.pos0 = 0;
    }
    const int total = groups * ks;
    p.nunit = std::max(1, (total + cus - 1) / cus);
    const int grid = (total + p.nunit - 1) / p.nunit;
    constexpr int EPI = MODE == MODE_QKV ? EPI_QKV : (MODE == MODE_SWIGLU ? EPI_SWIGLU : EPI_STORE);
    hipLaunchKernelGGL((gemm_mfma_rows<PT, RT, EPI, JN, LD, STAGGER, MIX>), dim3(grid), dim3(kMfThreads), 0, st, p);
}
typedef void (*LaunchFn)(hipStream_t, const float*, const float*, float*, size_t, int, int, int, int, int);
struct Variant { const char* name; LaunchFn fn; int P; int ks; bool check; };

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    CK(hipSetDevice(0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Shape shapes[] = {{"wo    4096x4096 ", 4096, 4096}, {"w2    4096x11008", 4096, 11008},
                      {"qkv   12288x4096", 12288, 4096}, {"w13   22016x4096", 22016, 4096}};
    const size_t max_floats = (size_t)32000 * 4096;
    const int nbuf = 6;
    std::vector<float*> W(nbuf);
    for (int i = 0; i < nbuf; i++) { CK(hipMalloc(&W[i], max_floats * 4)); hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, W[i], max_floats, 17u + i, 1.0f / 32768.0f * 0.02f); }
    float *X, *XT, *O, *Oref;
    const int PMAX = 128;
    CK(hipMalloc(&X, (size_t)PMAX * 11008 * 4)); CK(hipMalloc(&XT, (size_t)PMAX * 11008 * 4));
    CK(hipMalloc(&O, (size_t)4 * PMAX * 32000 * 4)); CK(hipMalloc(&Oref, (size_t)PMAX * 32000 * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, X, (size_t)PMAX * 11008, 99u, 1.0f / 32768.0f);
    CK(hipStreamSynchronize(st));

    // LD 1 = row-major weights (bit-checked against the host), LD 3 = the same reads from a tile-order copy
    // (timing only here: the harness does not tile W, the product path is checked by tests/test_hip_forward.py)
    std::vector<Variant> vs = {
        {"PT1 qkv3 J2 tile-order     ", launch<1, 3, 2, 3, MODE_QKV>, 16, 1, false},
        {"PT2 qkv3 J2 tile-order     ", launch<2, 3, 2, 3, MODE_QKV>, 32, 1, false},
        {"PT4 qkv3 J2 tile-order     ", launch<4, 3, 2, 3, MODE_QKV>, 64, 1, false},
        {"PT4 swiglu J2 tile-order   ", launch<4, 2, 2, 3, MODE_SWIGLU>, 64, 1, false},
        {"PT4 RT2 J2 ks2 row-major   ", launch<4, 2, 2, 1, MODE_ROWS>, 64, 2, true},
        {"PT4 RT2 J2 ks2 tile-order  ", launch<4, 2, 2, 3, MODE_ROWS>, 64, 2, false},
        {"PT4 RT2 J2 ks4 tile-order  ", launch<4, 2, 2, 3, MODE_ROWS>, 64, 4, false},
        {"PT8 qkv3 J1 tile-order     ", launch<8, 3, 1, 3, MODE_QKV>, 128, 1, false},
        {"PT8 swiglu J1 tile-order   ", launch<8, 2, 1, 3, MODE_SWIGLU>, 128, 1, false},
        {"PT8 RT2 J1 ks2 tile-order  ", launch<8, 2, 1, 3, MODE_ROWS>, 128, 2, false},
        {"PT8 RT2 J1 ks4 tile-order  ", launch<8, 2, 1, 3, MODE_ROWS>, 128, 4, false},
        {"PT4 qkv3 J2 mix2           ", launch<4, 3, 2, 3, MODE_QKV, 0, 2>, 64, 1, false},
        {"PT4 swiglu J2 mix2         ", launch<4, 2, 2, 3, MODE_SWIGLU, 0, 2>, 64, 1, false},
        {"PT4 RT2 J2 ks2 mix2        ", launch<4, 2, 2, 3, MODE_ROWS, 0, 2>, 64, 2, false},
        {"PT8 qkv3 J1 mix1           ", launch<8, 3, 1, 3, MODE_QKV, 0, 1>, 128, 1, false},
        {"PT8 qkv3 J1 mix2           ", launch<8, 3, 1, 3, MODE_QKV, 0, 2>, 128, 1, false},
        {"PT8 qkv3 J1 mix4           ", launch<8, 3, 1, 3, MODE_QKV, 0, 4>, 128, 1, false},
        {"PT8 swiglu J1 mix2         ", launch<8, 2, 1, 3, MODE_SWIGLU, 0, 2>, 128, 1, false},
        {"PT8 RT2 J1 ks2 mix2        ", launch<8, 2, 1, 3, MODE_ROWS, 0, 2>, 128, 2, false},
        {"PT8 qkv3 J1 stagger8       ", launch<8, 3, 1, 3, MODE_QKV, 8, 0>, 128, 1, false},
        {"PT8 qkv3 J1 probe: no loads", launch<8, 3, 1, 6, MODE_QKV>, 128, 1, false},
        {"PT8 qkv3 probe: no X loads ", launch<8, 3, 1, 9, MODE_QKV>, 128, 1, false},
        {"PT8 qkv3 probe: no W loads ", launch<8, 3, 1, 10, MODE_QKV>, 128, 1, false},
        {"PT8 swiglu probe: no X lds ", launch<8, 2, 1, 9, MODE_SWIGLU>, 128, 1, false},
        {"PT8 swiglu probe: no W lds ", launch<8, 2, 1, 10, MODE_SWIGLU>, 128, 1, false},
        {"PT8 swiglu probe: no loads ", launch<8, 2, 1, 6, MODE_SWIGLU>, 128, 1, false},
        {"PT8 qkv3 MFMAs only        ", launch<8, 3, 1, 11, MODE_QKV>, 128, 1, false},
        {"PT8 swiglu MFMAs only      ", launch<8, 2, 1, 11, MODE_SWIGLU>, 128, 1, false},
        {"PT8 qkv3 no loads, B bank+1", launch<8, 3, 1, 8, MODE_QKV>, 128, 1, false},
        {"PT8 qkv3 no loads, prio 4-7", launch<8, 3, 1, 6, MODE_QKV, -1>, 128, 1, false},
        {"PT8 qkv3 no loads, prio alt", launch<8, 3, 1, 6, MODE_QKV, -2>, 128, 1, false},
        {"PT8 qkv3 MFMAs only prio4-7", launch<8, 3, 1, 11, MODE_QKV, -1>, 128, 1, false},
        {"PT8 qkv3 MFMAs only prioalt", launch<8, 3, 1, 11, MODE_QKV, -2>, 128, 1, false},
        {"PT8 qkv3 J1 prio 4-7       ", launch<8, 3, 1, 3, MODE_QKV, -1>, 128, 1, false},
        {"PT8 qkv3 J1 prio alt       ", launch<8, 3, 1, 3, MODE_QKV, -2>, 128, 1, false},
        {"PT8 qkv3 J1 prio alt8      ", launch<8, 3, 1, 3, MODE_QKV, -3>, 128, 1, false},
        {"PT8 swiglu J1 prio 4-7     ", launch<8, 2, 1, 3, MODE_SWIGLU, -1>, 128, 1, false},
        {"PT8 swiglu J1 prio alt     ", launch<8, 2, 1, 3, MODE_SWIGLU, -2>, 128, 1, false},
        {"PT8 qkv3 J1 probe: no MFMA ", launch<8, 3, 1, 7, MODE_QKV>, 128, 1, false},
        {"PT8 RT2 ks2 probe: no loads", launch<8, 2, 1, 6, MODE_ROWS>, 128, 2, false},
        {"PT4 qkv3 J2 probe: no loads", launch<4, 3, 2, 6, MODE_QKV>, 64, 1, false},
    };
    for (const Shape& sh : shapes) {
        printf("== %s\n", sh.name);
        const double wbytes = (double)sh.rows * sh.K * 4;
        for (const Variant& v : vs) {
            const int P = v.P;
            const size_t slab = (size_t)PMAX * 32000;
            CK(hipMemsetAsync(O, 0xff, (size_t)v.ks * slab * 4, st));
            hipLaunchKernelGGL(tile_rows_kernel, dim3(4, P), dim3(256), 0, st, XT, X, P, sh.K);
            v.fn(st, W[0], XT, O, slab, sh.K, sh.rows, P, cus, v.ks);
            double maxerr = -1.0;
            if (v.check) {
                hipLaunchKernelGGL(ref_gemm, dim3((sh.rows + 255) / 256, P), dim3(256), 0, st, W[0], X, Oref, sh.K, sh.rows, P);
                CK(hipStreamSynchronize(st));
                const size_t nt = tile_floats(P, sh.rows);
                std::vector<float> a(nt * v.ks), b((size_t)P * sh.rows);
                for (int k = 0; k < v.ks; k++) CK(hipMemcpy(a.data() + k * nt, O + k * slab, nt * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(b.data(), Oref, b.size() * 4, hipMemcpyDeviceToHost));
                maxerr = 0.0;
                for (int t = 0; t < P; t++)
                    for (int r = 0; r < sh.rows; r++) {
                        float sum = 0.0f;
                        for (int k = 0; k < v.ks; k++) sum += a[k * nt + tile_idx(t, r, sh.rows)];
                        double e = std::fabs((double)sum - (double)b[(size_t)t * sh.rows + r]);
                        if (!(e <= maxerr)) maxerr = e;
                    }
            }
            CK(hipStreamSynchronize(st));
            if (hipGetLastError() != hipSuccess) { printf("  %s launch failed\n", v.name); continue; }
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; i++) v.fn(st, W[i % nbuf], XT, O, slab, sh.K, sh.rows, P, cus, v.ks);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms / iters);
            }
#ifdef RAMA_MF_STAMPS
            {   // one more launch, then the stamps of workgroup RAMA_MF_STAMP_BLOCK: microseconds since its wave 0 started
                v.fn(st, W[0], XT, O, slab, sh.K, sh.rows, P, cus, v.ks);
                CK(hipStreamSynchronize(st));
                unsigned long long h[kMfWaves][8];
                CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mf_stamps), sizeof(h)));
                static unsigned long long hb[1024][4];
                CK(hipMemcpyFromSymbol(hb, HIP_SYMBOL(g_mf_blocks), sizeof(hb)));
                {
                    const int groups = MODE_OF_NAME(v.name) ? (sh.rows / (strstr(v.name, "qkv") ? 3 : 2) + 15) / 16 : (sh.rows + 31) / 32;
                    const int total = groups * v.ks, nunit = std::max(1, (total + cus - 1) / cus), grid = std::min(1024, (total + nunit - 1) / nunit);
                    unsigned long long t0 = ~0ull, t1 = 0; double dmin = 1e9, dmax = 0, dsum = 0;
                    int per_cu[8][64] = {};
                    for (int b = 0; b < grid; b++) {
                        t0 = std::min(t0, hb[b][0]); t1 = std::max(t1, hb[b][1]);
                        const double d = (double)(hb[b][1] - hb[b][0]) / 100.0;
                        dmin = std::min(dmin, d); dmax = std::max(dmax, d); dsum += d;
                        const unsigned hw = (unsigned)hb[b][2], xcc = (unsigned)hb[b][3] & 15u;
                        per_cu[xcc & 7][((hw >> 13) & 3) * 16 + ((hw >> 8) & 15)]++;      // se_id * 16 + cu_id
                    }
                    int used = 0, twice = 0;
                    for (int x = 0; x < 8; x++) for (int q = 0; q < 64; q++) { used += per_cu[x][q] > 0; twice += per_cu[x][q] > 1; }
                    double last_start = 0; for (int b = 0; b < grid; b++) last_start = std::max(last_start, (double)(hb[b][0] - t0) / 100.0);
                    printf("      %d workgroups: first start .. last end %.2f us, last start +%.2f us, duration min / mean / max %.2f / %.2f / %.2f us; %d distinct (xcc, se, cu), %d of them got more than one\n",
                           grid, (double)(t1 - t0) / 100.0, last_start, dmin, dsum / grid, dmax, used, twice);
                }
                for (int w = 0; w < kMfWaves; w++) {
                    printf("      wave %d:", w);
                    for (int k = 0; k < 5; k++) printf(" %7.2f", (double)(h[w][k] - h[0][0]) / 100.0);
                    printf("   (start, loop end, fold written, barrier, epilogue end) us; core clock over the loop %.0f MHz\n",
                           (double)(h[w][6] - h[w][5]) / ((double)(h[w][1] - h[w][0]) / 100.0));
                }
            }
#endif
            printf("  %s P=%2d  %8.1f us  %6.0f GB/s weights  %6.1f TFLOP/s  maxerr %.2e\n", v.name, P, best * 1e3,
                   wbytes / (best * 1e-3) / 1e9, 2.0 * sh.rows * sh.K * P / (best * 1e-3) / 1e12, maxerr);
        }
    }
    return 0;
}
