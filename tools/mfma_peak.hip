// mfma_peak.hip -- what v_mfma_f32_16x16x4_f32 sustains on this part with nothing else going on
// (registers only, NACC independent accumulators per wave, W waves per SIMD).  Sizes the ceiling of the
// fp32 token-batch GEMMs (prefill_mfma.hpp).  Build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_peak tools/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) float acc4;

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
    acc4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = acc4{0.f, 0.f, 0.f, 0.f};
    float a = (float)threadIdx.x * 1e-3f, b = (float)blockIdx.x * 1e-4f + 1.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        a += 1e-7f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
static void run(int wgs, int iters, const char* name) {
    float* out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(wgs), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(wgs), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)wgs * 4 /*waves*/ * iters * NACC * (16.0 * 16 * 4 * 2);
    printf("%-34s %4d workgroups of 4 waves, %d accumulators: %7.1f TFLOP/s (%.2f ms)\n", name, wgs, NACC, flops / ms / 1e9, ms);
    CK(hipFree(out));
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", pr.gcnArchName, cus, pr.clockRate / 1000);
    run<12>(cus, 20000, "1 wave per SIMD");
    run<12>(2 * cus, 20000, "2 waves per SIMD");
    run<12>(4 * cus, 10000, "4 waves per SIMD");
    run<4>(2 * cus, 40000, "2 waves per SIMD");
    run<2>(2 * cus, 40000, "2 waves per SIMD");
    run<1>(2 * cus, 40000, "2 waves per SIMD");
    run<12>(2 * cus, 200000, "2 waves per SIMD, 10x longer");
    return 0;
}
