// clock_probe.hip -- the core clock a small, latency-bound launch actually runs at (s_memtime cycles per s_memrealtime
// tick, 100 MHz), alone and with a few "heater" workgroups kept busy on another stream.  A tuning aid: decode steps of
// the small shapes are chains of 4-5 us launches on an otherwise idle chip.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/clock_probe tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void small_kernel(unsigned long long* out, int slot, int n_adds, float seed) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    float v = seed + threadIdx.x;
    for (int i = 0; i < n_adds; i++) v = v + 1.0f;      // a dependent chain
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[2 * slot] = c1 - c0; out[2 * slot + 1] = t1 - t0; }
    if (v == 12345.0f) out[0] = 0;
}
__global__ void heater_kernel(volatile int* stop, float* sink) {
    float a = threadIdx.x, b = 1.0001f;
    for (int it = 0; it < 2000000 && !*stop; it++) {      // bounded (about a second) whatever happens to the host
#pragma unroll
        for (int i = 0; i < 256; i++) a = a * b + 0.5f;
    }
    if (a == 1.2345f) *sink = a;
}

int main(int argc, char** argv) {
    const int heaters = argc > 1 ? atoi(argv[1]) : 8;
    const int N = 2000, adds = 400;
    unsigned long long* out; CK(hipMalloc(&out, sizeof(unsigned long long) * 2 * N));
    int* stop; CK(hipHostMalloc(&stop, sizeof(int), hipHostMallocMapped)); *stop = 0;
    int* stop_dev; CK(hipHostGetDevicePointer((void**)&stop_dev, stop, 0));
    float* sink; CK(hipMalloc(&sink, 4));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    auto run = [&](const char* name) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 2; w++) {
            CK(hipEventRecord(e0, s1));
            for (int i = 0; i < N; i++) hipLaunchKernelGGL(small_kernel, dim3(64), dim3(64), 0, s1, out, i, adds, 1.0f);
            CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(2 * N); CK(hipMemcpy(h.data(), out, sizeof(unsigned long long) * 2 * N, hipMemcpyDeviceToHost));
        std::vector<double> mhz, us;
        for (int i = 0; i < N; i++) if (h[2 * i + 1]) { mhz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] / 100.0)); us.push_back(h[2 * i + 1] / 100.0); }
        std::sort(mhz.begin(), mhz.end()); std::sort(us.begin(), us.end());
        printf("%-34s %7.2f us per launch; inside the kernel: clock min / median / max %4.0f / %4.0f / %4.0f MHz, %d dependent adds in %.2f us (median) = %.1f cycles each\n",
               name, ms * 1e3 / N, mhz.front(), mhz[mhz.size() / 2], mhz.back(), adds, us[us.size() / 2], mhz[mhz.size() / 2] * us[us.size() / 2] / adds);
    };
    run("alone");
    hipLaunchKernelGGL(heater_kernel, dim3(heaters), dim3(256), 0, s2, (volatile int*)stop_dev, sink);
    run("with heater workgroups");
    *stop = 1;
    CK(hipStreamSynchronize(s2));
    run("alone again");
    return 0;
}
