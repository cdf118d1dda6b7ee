// mall_probe.hip -- does a region read a moment ago come back faster?  Kernel B streams `total` MB with non-temporal 16-byte loads
// (as the decode matvecs do); before it either nothing, or kernel A reads the first `warm` MB of the same region with ordinary
// loads (while a one-workgroup launch such as the parity rmsnorm occupies the stream).  In between runs a 1 GB sweep evicts.
// Not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mall_probe tools/mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) float f4;
template <bool NT>
__global__ __launch_bounds__(512) void sweep(const f4* p, size_t n4, float* sink) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 512 * 4;
    for (size_t i = (size_t)blockIdx.x * 512 * 4 + threadIdx.x; i < n4; i += stride) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const size_t j = i + (size_t)u * 512; v[u] = j < n4 ? (NT ? __builtin_nontemporal_load(p + j) : p[j]) : acc; }
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
int main(int argc, char** argv) {
    const size_t total = (argc > 1 ? atoi(argv[1]) : 201) * (size_t)1 << 20, big = (size_t)1 << 30;
    float *w, *junk, *sink;
    CK(hipMalloc(&w, total)); CK(hipMalloc(&junk, big)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(w, 0, total)); CK(hipMemset(junk, 0, big));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int warm_mb : {0, 16, 32, 48, 96, 201}) {
        float best = 1e9f, sum = 0.f;
        for (int rep = 0; rep < 6; rep++) {
            hipLaunchKernelGGL(sweep<true>, dim3(1024), dim3(512), 0, 0, (const f4*)junk, big / 16, sink);                 // evict
            if (warm_mb) hipLaunchKernelGGL(sweep<false>, dim3(255), dim3(512), 0, 0, (const f4*)w, ((size_t)warm_mb << 20) / 16, sink);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(sweep<true>, dim3(2048), dim3(512), 0, 0, (const f4*)w, total / 16, sink);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) { sum += ms; if (ms < best) best = ms; }
        }
        printf("stream %zu MB after reading its first %3d MB: best %.2f us, mean %.2f us\n", total >> 20, warm_mb, best * 1e3, sum * 1e3 / 5);
    }
    return 0;
}
