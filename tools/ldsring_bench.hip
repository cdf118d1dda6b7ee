// ldsring_bench.hip -- the residual products of parity mode (Wo: 256 row groups of K = 4096; W2: K = 11008; one wave per group, one group per CU) with
// the ring of weight blocks in LDS, filled by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPRs), against the product's kernel
// (a ring of 32 blocks in registers).  Same products and sums in the same order; results compared.  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -Irama_amd/csrc -Iinclude -o tools/bin/ldsring_bench tools/ldsring_bench.hip
#include "../rama_amd/csrc/chain.hpp"
#include <cstdio>
#include <vector>
#include <cstring>
using namespace rama;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// one wave = 16 rows; R ring slots of 1 KiB in LDS behind the activations
template <int R>
__global__ __launch_bounds__(64) void ring_kernel(const float* __restrict__ Wc, const float* __restrict__ x, float* __restrict__ o, float* __restrict__ resid, int K, int rows) {
    RAMA_NO_CONTRACT
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, g = blockIdx.x, nblk = K >> 4, j = lane & 3, rr = lane >> 2;
    float* xs = sm;                                   // [K] chain order
    const unsigned ring = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)(sm + K);      // R KiB
    const float* wg = Wc + (size_t)g * nblk * 256 + lane * 4;
    const int row = 16 * g + rr;
    const float xold = (j == 0 && row < rows) ? resid[row] : 0.0f;
    // the ring first (HBM), then the activations (L2)
#pragma unroll
    for (int s = 0; s < R; s++) glds16(wg + (size_t)min(s, nblk - 1) * 256, ring + (unsigned)s * 1024u);
    for (int i = lane; i < (K >> 2); i += 64) {       // xs[16 s + 4 j + t] = x[16 s + 4 t + j]
        const f4 v = reinterpret_cast<const f4*>(x)[i];
        const int s = i >> 2, t = i & 3;
        xs[16 * s + 0 + t] = v.x; xs[16 * s + 4 + t] = v.y; xs[16 * s + 8 + t] = v.z; xs[16 * s + 12 + t] = v.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float acc = 0.0f;
    const f4* xq = reinterpret_cast<const f4*>(xs) + j;
    constexpr int U = 4;
    f4 wa[U], xa[U], wb[U], xb[U];
    auto rd = [&](int s0, f4 (&w)[U], f4 (&xv)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) { w[u] = *reinterpret_cast<const f4*>(sm + K + ((s0 + u) % R) * 256 + lane * 4); xv[u] = xq[4 * (s0 + u)]; }
    };
    auto math = [&](const f4 (&w)[U], const f4 (&xv)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) { acc = acc + w[u].x * xv[u].x; acc = acc + w[u].y * xv[u].y; acc = acc + w[u].z * xv[u].z; acc = acc + w[u].w * xv[u].w; }
    };
    auto refill = [&](int s0) {      // the slots of blocks s0 .. s0 + U - 1 take blocks s0 + R ..; behind the row's end the last block again (keeps the count)
#pragma unroll
        for (int u = 0; u < U; u++) glds16(wg + (size_t)min(s0 + u + R, nblk - 1) * 256, ring + (unsigned)((s0 + u) % R) * 1024u);
    };
    // (host: nblk % (2 U) == 0)
    wait_vm<R - U>();
    rd(0, wa, xa);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    refill(0);
    for (int s = 0; s < nblk; s += 2 * U) {
        wait_vm<R - U>();
        rd(s + U, wb, xb);
        math(wa, xa);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        refill(s + U);
        if (s + 2 * U < nblk) {
            wait_vm<R - U>();
            rd(s + 2 * U, wa, xa);
        }
        math(wb, xb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (s + 2 * U < nblk) refill(s + 2 * U);
    }
    wait_vm<0>();
    // (v0 + v1) + (v2 + v3) over the four chains of a row
    const float a1 = __shfl_xor(acc, 1); const float s01 = (j & 1) ? a1 + acc : acc + a1;
    const float a2 = __shfl_xor(s01, 2); const float d = (j & 2) ? a2 + s01 : s01 + a2;
    if (j == 0 && row < rows) { o[row] = d; resid[row] = xold + d; }
}

int main() {
    constexpr int L = 24, DIM = 4096, HID = 11008;
    float *wc, *x, *o, *o2, *res, *res2;
    const size_t per = (size_t)DIM * HID;
    CK(hipMalloc(&wc, L * per * 4));
    { std::vector<float> h(per); for (size_t i = 0; i < per; i++) h[i] = (float)((i * 2654435761u >> 20) & 1023) * (1.0f / 4096.0f) - 0.125f;
      for (int l = 0; l < L; l++) CK(hipMemcpy(wc + l * per, h.data(), per * 4, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&x, HID * 4)); CK(hipMalloc(&o, DIM * 4)); CK(hipMalloc(&o2, DIM * 4)); CK(hipMalloc(&res, DIM * 4)); CK(hipMalloc(&res2, DIM * 4));
    { std::vector<float> xx(HID); for (int i = 0; i < HID; i++) xx[i] = (float)((i * 37) % 101) * 0.02f - 1.0f; CK(hipMemcpy(x, xx.data(), HID * 4, hipMemcpyHostToDevice)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)ring_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)ring_kernel<48>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)ring_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    for (int K : {DIM, HID}) {
        const size_t stride = (size_t)DIM * K;
        for (int what = 0; what < 4; what++) {
            CK(hipMemset(res, 0, DIM * 4)); CK(hipMemset(res2, 0, DIM * 4));
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0, 0));
                for (int l = 0; l < L; l++) {
                    if (what == 0) {
                        ChainParams p{}; p.w[0] = wc + l * stride; p.o[0] = o; p.resid = res; p.x = x; p.K = K; p.rows = DIM; p.nmat = 1;
                        hipLaunchKernelGGL((gemv_chain_kernel<1, 32, 4, CEPI_RESID>), dim3(DIM / 16), dim3(64), (size_t)(K + chain_pad_floats(1, 32, 4)) * 4, 0, p);
                    } else if (what == 1) hipLaunchKernelGGL(ring_kernel<32>, dim3(DIM / 16), dim3(64), (size_t)K * 4 + 32 * 1024, 0, wc + l * stride, x, o2, res2, K, DIM);
                    else if (what == 2) hipLaunchKernelGGL(ring_kernel<48>, dim3(DIM / 16), dim3(64), (size_t)K * 4 + 48 * 1024, 0, wc + l * stride, x, o2, res2, K, DIM);
                    else hipLaunchKernelGGL(ring_kernel<64>, dim3(DIM / 16), dim3(64), (size_t)K * 4 + 64 * 1024, 0, wc + l * stride, x, o2, res2, K, DIM);
                }
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            const double us = best * 1e3 / L, gb = (double)DIM * K * 4 / 1e9;
            const char* names[] = {"product kernel (ring of 32 blocks in registers)", "LDS ring of 32 KiB", "LDS ring of 48 KiB", "LDS ring of 64 KiB"};
            printf("K %5d: %-50s %.2f us per launch, %.0f GB/s\n", K, names[what], us, gb / (us * 1e-6));
        }
        std::vector<float> a(DIM), b(DIM);
        CK(hipMemcpy(a.data(), o, DIM * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o2, DIM * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < DIM; i++) bad += memcmp(&a[i], &b[i], 4) != 0;
        printf("K %5d: rows that differ between the two kernels: %d (o[5] = %g / %g)\n", K, bad, a[5], b[5]);
    }
    return 0;
}
