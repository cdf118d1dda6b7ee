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

// the GEMM's operand pattern: RT x PT accumulators, A[rt] and B[pt] in distinct registers, the 4 k-slots of a
// 16-byte operand in consecutive registers (prefill_mfma.hpp compute()).  LDSF floats of static LDS and PAD extra live
// registers per lane make the kernel's footprint the GEMM's (98 KB, 256 VGPRs); the operands come from `in` (zeros or noise)
template <int RT, int PT, int LDSF, int PAD, int PRIO = 0>
__global__ __launch_bounds__(512) void k_mfma_grid(float* out, const float* in, int iters) {
    __shared__ float lds[LDSF > 0 ? LDSF : 1];
    acc4 acc[RT][PT];
#pragma unroll
    for (int r = 0; r < RT; r++)
#pragma unroll
        for (int q = 0; q < PT; q++) acc[r][q] = acc4{0.f, 0.f, 0.f, 0.f};
    acc4 A[RT], B[PT];
    float pad[PAD > 0 ? PAD : 1];
#pragma unroll
    for (int r = 0; r < RT; r++) A[r] = reinterpret_cast<const acc4*>(in)[threadIdx.x + 512 * r];
#pragma unroll
    for (int q = 0; q < PT; q++) B[q] = reinterpret_cast<const acc4*>(in)[threadIdx.x + 512 * (RT + q)];
#pragma unroll
    for (int k = 0; k < PAD; k++) pad[k] = in[threadIdx.x + 64 * k];
    if (LDSF > 0) lds[threadIdx.x] = A[0][0];
    if (PRIO == 1 && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);      // static priority for the younger half
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (PRIO == 2) { if ((it & 1) == (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256 ? 1 : 0)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int r = 0; r < RT; r++)
#pragma unroll
                for (int q = 0; q < PT; q++) acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r][e], B[q][e], acc[r][q], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < PAD; k++) asm volatile("" : "+v"(pad[k]));      // keeps the padding live across the loop, no instruction
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RT; r++)
#pragma unroll
        for (int q = 0; q < PT; q++) s += acc[r][q].x + acc[r][q].y + acc[r][q].z + acc[r][q].w;
#pragma unroll
    for (int k = 0; k < PAD; k++) s += pad[k];
    if (LDSF > 0) s += lds[(threadIdx.x + 1) & 511];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (float)(c1 - c0); out[2] = (float)(t1 - t0); }
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[4 + (threadIdx.x >> 6)] = (float)(t1 - t0) / 100.0f;      // each wave's loop, us
    if (PRIO == 1 && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(0);
}
__global__ void noise_kernel(float* d, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned z = (unsigned)i * 2654435761u + 12345u; z ^= z >> 15; z *= 2246822519u; z ^= z >> 13;
        d[i] = ((float)(z & 0xFFFF) - 32768.0f) / 32768.0f;
    }
}
template <int RT, int PT, int LDSF, int PAD, int PRIO = 0>
static void run_grid(int wgs, int iters, bool noise, const char* name) {
    const int nin = 512 * 16 * 4;
    float *out, *in; CK(hipMalloc(&out, 64)); CK(hipMalloc(&in, nin * 4)); CK(hipMemset(in, 0, nin * 4));
    if (noise) hipLaunchKernelGGL(noise_kernel, dim3(64), dim3(256), 0, 0, in, nin);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mfma_grid<RT, PT, LDSF, PAD, PRIO>), dim3(wgs), dim3(512), 0, 0, out, in, iters);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mfma_grid<RT, PT, LDSF, PAD, PRIO>), dim3(wgs), dim3(512), 0, 0, out, in, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    float h[16]; CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
    const double flops = (double)wgs * 8 /*waves*/ * iters * 4 * RT * PT * (16.0 * 16 * 4 * 2);
    const double mhz = h[1] / (h[2] / 100.0), cyc = h[1] / ((double)iters * 4 * RT * PT);     // wave 0 of workgroup 0: cycles between its own MFMAs
    printf("%-30s %d x %d accumulators, LDS %3d KB, +%3d registers, %s operands: %7.1f TFLOP/s (%.2f ms); core clock %4.0f MHz, %.1f cycles per MFMA per SIMD\n",
           name, RT, PT, LDSF * 4 / 1024, PAD, noise ? "random" : "zero  ", flops / ms / 1e9, ms, mhz, cyc);
    printf("      priority mode %d; loop of each wave of workgroup 0 (us):", PRIO); for (int w = 0; w < 8; w++) printf(" %.0f", h[4 + w]); printf("\n");
    CK(hipFree(out)); CK(hipFree(in));
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
    run_grid<3, 8, 0, 0>(cus, 2000, false, "8-wave workgroups");
    run_grid<3, 8, 0, 0>(cus, 2000, true, "8-wave workgroups");
    run_grid<3, 8, 0, 30>(cus, 2000, true, "8-wave workgroups");
    run_grid<3, 8, 0, 40>(cus, 2000, true, "8-wave workgroups");
    run_grid<3, 8, 0, 40, 1>(cus, 2000, true, "8-wave workgroups");
    run_grid<3, 8, 0, 40, 2>(cus, 2000, true, "8-wave workgroups");
    run_grid<3, 8, 0, 100>(cus, 2000, true, "8-wave workgroups");
    run_grid<3, 8, 0, 100, 1>(cus, 2000, true, "8-wave workgroups");
    run_grid<3, 8, 0, 100, 2>(cus, 2000, true, "8-wave workgroups");
    run_grid<3, 8, 24576, 100>(cus, 2000, true, "8-wave workgroups");
    run_grid<2, 8, 0, 90>(cus, 3000, true, "8-wave workgroups");
    run_grid<2, 8, 0, 90, 2>(cus, 3000, true, "8-wave workgroups");
    return 0;
}
