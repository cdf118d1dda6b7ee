// handoff_bench.hip -- what one inter-workgroup hand-off costs on this chip, in the shape csrc/layer_fused.hpp needs it:
// PH phases of W one-wave workgroups; every wave of phase p needs the whole 4W-float vector phase p-1 produced (4 floats per
// wave).  Protocol 0: sc1 stores, s_waitcnt vmcnt(0), agent-scope arrival counter; consumer polls the counter, then sc1 loads.
// Protocol 1: the data carries its own tag -- 8-byte (value, epoch) stores; the consumer polls one pair, then loads the vector
// and checks every tag (reloads until all match).  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/handoff_bench tools/handoff_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u4;
struct P { int W, PH; unsigned epoch; float* plain; unsigned long long* tagged; unsigned* cnt; unsigned long long* err; unsigned long long* stamps; };

__device__ __forceinline__ float wsum(float v) { for (int m = 32; m; m >>= 1) v += __shfl_xor(v, m); return v; }

template <int PROTO, int SLEEP>
__global__ __launch_bounds__(64) void chain(P a) {
    const int lane = threadIdx.x, ph = blockIdx.x / a.W, w = blockIdx.x - ph * a.W, K = 4 * a.W;
    float sum = 0.0f;
    if (ph > 0) {
        if (PROTO == 0) {
            unsigned* c = a.cnt + (size_t)(ph - 1) * 64;
            if (lane == 0) {
                long spins = 0;
                while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.W) {
                    if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
                    if (++spins > (1L << 22)) { *a.err = 1; break; }
                }
            }
            __builtin_amdgcn_wave_barrier();
            const float* v = a.plain + (size_t)(ph - 1) * K;
            for (int i = lane; i < K; i += 64) sum += __hip_atomic_load(v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned long long* v = a.tagged + (size_t)(ph - 1) * K;
            if (lane == 0) {
                long spins = 0;
                while ((unsigned)(__hip_atomic_load(v + 4 * (a.W - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) != a.epoch) {
                    if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
                    if (++spins > (1L << 22)) { *a.err = 2; break; }
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int tries = 0; tries < (1 << 20); tries++) {
                bool ok = true;
                sum = 0.0f;
                for (int i = lane; i < K; i += 64) {
                    const unsigned long long pr = __hip_atomic_load(v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(pr >> 32) == a.epoch;
                    sum += __uint_as_float((unsigned)pr);
                }
                if (__all(ok)) break;
                if (tries == (1 << 20) - 1) *a.err = 3;
            }
        }
    }
    sum = wsum(sum) * 1e-3f;
    if (w == 0 && lane == 0) a.stamps[ph] = __builtin_amdgcn_s_memrealtime();
    if (lane < 4) {
        const float out = sum + (float)lane;
        if (PROTO == 0) __hip_atomic_store(a.plain + (size_t)ph * K + 4 * w + lane, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(a.tagged + (size_t)ph * K + 4 * w + lane, ((unsigned long long)a.epoch << 32) | __float_as_uint(out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (PROTO == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(a.cnt + (size_t)ph * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    const int PH = 30;
    for (int W : {8, 72, 192, 384}) {
        const int K = 4 * W;
        P a{}; a.W = W; a.PH = PH;
        CK(hipMalloc(&a.plain, (size_t)PH * K * 4)); CK(hipMalloc(&a.tagged, (size_t)PH * K * 8)); CK(hipMalloc(&a.cnt, (size_t)PH * 256));
        CK(hipMalloc(&a.err, 8)); CK(hipMalloc(&a.stamps, PH * 8));
        CK(hipMemset(a.tagged, 0, (size_t)PH * K * 8)); CK(hipMemset(a.err, 0, 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int proto = 0; proto < 2; proto++)
            for (int sleep : {0, 2}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; rep++) {
                    CK(hipMemsetAsync(a.cnt, 0, (size_t)PH * 256, 0));
                    CK(hipEventRecord(e0, 0));
                    for (int i = 0; i < 10; i++) {
                        a.epoch++;
                        if (proto == 0) CK(hipMemsetAsync(a.cnt, 0, (size_t)PH * 256, 0));
                        if (proto == 0 && sleep == 0) hipLaunchKernelGGL((chain<0, 0>), dim3(PH * W), dim3(64), 0, 0, a);
                        else if (proto == 0) hipLaunchKernelGGL((chain<0, 2>), dim3(PH * W), dim3(64), 0, 0, a);
                        else if (sleep == 0) hipLaunchKernelGGL((chain<1, 0>), dim3(PH * W), dim3(64), 0, 0, a);
                        else hipLaunchKernelGGL((chain<1, 2>), dim3(PH * W), dim3(64), 0, 0, a);
                    }
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                unsigned long long st[PH], err;
                CK(hipMemcpy(st, a.stamps, sizeof st, hipMemcpyDeviceToHost)); CK(hipMemcpy(&err, a.err, 8, hipMemcpyDeviceToHost));
                printf("W %3d (vector %4d floats) protocol %s sleep %d: %.2f us per launch of %d phases; per phase by stamps %.2f us (err %llu)\n", W, K,
                       proto ? "tagged " : "counter", sleep, best * 1e3 / 10, PH, (double)(st[PH - 1] - st[4]) * 0.01 / (PH - 5), err);
            }
    }
    return 0;
}
