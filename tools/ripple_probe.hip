// ripple_probe.hip -- cycles per step of the lane ripples the exact sequential sums use (csrc/chain.hpp, topp_sort.hpp):
// t_i = t_(i-1) + e_i down the lanes, one DPP add per step.  One wave, 4096 dependent steps per variant, s_memtime around them.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ripple_probe tools/ripple_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long* out, float* sink) {
    float pv = 1.0f + threadIdx.x * 1e-3f, sv = pv;
    unsigned long long t[6];
    t[0] = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
#pragma unroll
        for (int k = 0; k < 64; k++) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(sv) : "v"(pv));
    }
    t[1] = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
#pragma unroll
        for (int k = 0; k < 64; k++) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(sv) : "v"(pv));
    }
    t[2] = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
#pragma unroll
        for (int k = 0; k < 64; k++) asm volatile("s_nop 0\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(sv) : "v"(pv));
    }
    t[3] = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
#pragma unroll
        for (int k = 0; k < 64; k++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sv) : "v"(pv));
    }
    t[4] = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
#pragma unroll
        for (int k = 0; k < 64; k++) {
            const float e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv), k));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(sv) : "s"(e));
        }
    }
    t[5] = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) for (int i = 0; i < 6; i++) out[i] = t[i];
    sink[threadIdx.x] = sv;
}
int main() {
    unsigned long long* d; float* s; hipMalloc(&d, 64); hipMalloc(&s, 256);
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, s); hipDeviceSynchronize(); }
    unsigned long long h[6]; hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
    const char* names[] = {"s_nop 1 + v_add_f32_dpp wave_shr:1", "s_nop 1 + v_add_f32_dpp row_shr:1", "s_nop 0 + v_add_f32_dpp row_shr:1", "v_add_f32 (dependent, no DPP)", "v_readlane + v_add_f32 (scalar operand)"};
    for (int i = 0; i < 5; i++) printf("%-42s %.1f cycles per step\n", names[i], (double)(h[i + 1] - h[i]) / 4096.0);
    return 0;
}
