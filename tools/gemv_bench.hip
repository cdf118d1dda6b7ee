// gemv_bench.hip -- standalone microbenchmark of fp32 matvec access patterns on gfx950.
// Not part of the product: a tuning aid.  Build: hipcc --offload-arch=gfx950 -O3 -o gemv_bench gemv_bench.hip
// Each variant computes o = W.x for W [rows, K]; timing = HIP events around `iters` launches
// that rotate over `nbuf` distinct weight buffers (so the 256 MiB Infinity Cache cannot help).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <algorithm>
#include <cmath>

typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
constexpr unsigned kOOB = 0x80000000u;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ f4 ldw(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, AUX));
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
    float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float dot4(f4 a, f4 b, float acc) {
    acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
    return acc;
}

// ---- variant A: workgroup of NW waves owns R rows; K split over the waves.
//   ASSIGN 0: wave w takes a contiguous K range; ASSIGN 1: wave w takes chunks c = w (mod NW)
template <int R, int CH, int NW, int ASSIGN, int AUX>
__global__ __launch_bounds__(NW * 64) void gemv_ksplit(const float* W, const float* x, float* o, int K, int rows) {
    __shared__ float part[NW][R];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * R;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(W, (unsigned)rows * (unsigned)K * 4u);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (unsigned)K * 4u);
    const int nch = (K + 255) >> 8;
    const unsigned kbytes = (unsigned)K * 4u;
    unsigned rowoff[R];
#pragma unroll
    for (int s = 0; s < R; s++) rowoff[s] = (r0 + s < rows) ? (unsigned)(r0 + s) * kbytes : kOOB;
    float acc[R];
#pragma unroll
    for (int s = 0; s < R; s++) acc[s] = 0.f;
    int c0, c1, cstride;
    if (ASSIGN == 0) { int cpw = (nch + NW - 1) / NW; c0 = wave * cpw; c1 = min(c0 + cpw, nch); cstride = 1; }
    else { c0 = wave; c1 = nch; cstride = NW; }
    for (int c = c0; c < c1; c += CH * cstride) {
        f4 w[R][CH]; f4 xv[CH]; unsigned kb[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            int ci = c + j * cstride;
            unsigned b = (unsigned)(ci * 1024 + lane * 16);
            kb[j] = (ci < c1 && b < kbytes) ? b : kOOB;
        }
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int s = 0; s < R; s++) w[s][j] = ldw<AUX>(ra, kb[j] == kOOB ? kOOB : rowoff[s] + kb[j]);
#pragma unroll
        for (int j = 0; j < CH; j++) xv[j] = ldw<0>(rx, kb[j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int s = 0; s < R; s++) acc[s] = dot4(w[s][j], xv[j], acc[s]);
    }
#pragma unroll
    for (int s = 0; s < R; s++) acc[s] = wave_sum(acc[s]);
    if (lane == 0) {
#pragma unroll
        for (int s = 0; s < R; s++) part[wave][s] = acc[s];
    }
    __syncthreads();
    if (threadIdx.x < R && r0 + threadIdx.x < rows) {
        float d = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) d += part[w][threadIdx.x];
        o[r0 + threadIdx.x] = d;
    }
}

// ---- variant B: every wave owns R whole rows (no cross-wave reduce); 4 waves per workgroup
template <int R, int CH, int AUX>
__global__ __launch_bounds__(256) void gemv_waverow(const float* W, const float* x, float* o, int K, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * 4 + wave) * R;
    if (r0 >= rows) return;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(W, (unsigned)rows * (unsigned)K * 4u);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (unsigned)K * 4u);
    const int nch = (K + 255) >> 8;
    const unsigned kbytes = (unsigned)K * 4u;
    unsigned rowoff[R];
#pragma unroll
    for (int s = 0; s < R; s++) rowoff[s] = (r0 + s < rows) ? (unsigned)(r0 + s) * kbytes : kOOB;
    float acc[R];
#pragma unroll
    for (int s = 0; s < R; s++) acc[s] = 0.f;
    for (int c = 0; c < nch; c += CH) {
        f4 w[R][CH]; f4 xv[CH]; unsigned kb[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            unsigned b = (unsigned)((c + j) * 1024 + lane * 16);
            kb[j] = ((c + j) < nch && b < kbytes) ? b : kOOB;
        }
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int s = 0; s < R; s++) w[s][j] = ldw<AUX>(ra, kb[j] == kOOB ? kOOB : rowoff[s] + kb[j]);
#pragma unroll
        for (int j = 0; j < CH; j++) xv[j] = ldw<0>(rx, kb[j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int s = 0; s < R; s++) acc[s] = dot4(w[s][j], xv[j], acc[s]);
    }
#pragma unroll
    for (int s = 0; s < R; s++) {
        float d = wave_sum(acc[s]);
        if (lane == 0 && r0 + s < rows) o[r0 + s] = d;
    }
}

// ---- pure read bandwidth probe: every workgroup streams a contiguous slab, sums it
template <int CH, int AUX>
__global__ __launch_bounds__(256) void read_probe(const float* W, float* o, size_t n_f4_per_block) {
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(W + (size_t)blockIdx.x * n_f4_per_block * 4, (unsigned)(n_f4_per_block * 16));
    f4 acc = {0, 0, 0, 0};
    for (unsigned i = threadIdx.x; i < n_f4_per_block; i += 256 * CH) {
        f4 v[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) { unsigned idx = i + j * 256; v[j] = ldw<AUX>(ra, idx < n_f4_per_block ? idx * 16 : kOOB); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CH; j++) acc += v[j];
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) o[blockIdx.x] = s;
}

struct Variant { std::string name; void (*launch)(const float*, const float*, float*, int, int, hipStream_t); };

template <int R, int CH, int NW, int ASSIGN, int AUX>
void launch_ksplit(const float* W, const float* x, float* o, int K, int rows, hipStream_t s) {
    hipLaunchKernelGGL((gemv_ksplit<R, CH, NW, ASSIGN, AUX>), dim3((rows + R - 1) / R), dim3(NW * 64), 0, s, W, x, o, K, rows);
}
template <int R, int CH, int AUX>
void launch_waverow(const float* W, const float* x, float* o, int K, int rows, hipStream_t s) {
    hipLaunchKernelGGL((gemv_waverow<R, CH, AUX>), dim3((rows + 4 * R - 1) / (4 * R)), dim3(256), 0, s, W, x, o, K, rows);
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 40;
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<Variant> vs = {
        {"ksplit R4 CH4 NW4 inter  nt", launch_ksplit<4, 4, 4, 1, 2>},
        {"ksplit R2 CH4 NW4 contig nt", launch_ksplit<2, 4, 4, 0, 2>},
        {"ksplit R2 CH4 NW4 inter  nt", launch_ksplit<2, 4, 4, 1, 2>},
        {"ksplit R2 CH2 NW4 inter  nt", launch_ksplit<2, 2, 4, 1, 2>},
        {"ksplit R2 CH3 NW4 inter  nt", launch_ksplit<2, 3, 4, 1, 2>},
        {"ksplit R2 CH6 NW4 inter  nt", launch_ksplit<2, 6, 4, 1, 2>},
        {"ksplit R1 CH4 NW4 contig nt", launch_ksplit<1, 4, 4, 0, 2>},
        {"ksplit R1 CH4 NW4 inter  nt", launch_ksplit<1, 4, 4, 1, 2>},
        {"ksplit R1 CH12 NW4 inter nt", launch_ksplit<1, 12, 4, 1, 2>},
        {"ksplit R3 CH4 NW4 inter  nt", launch_ksplit<3, 4, 4, 1, 2>},
        {"ksplit R2 CH2 NW8 inter  nt", launch_ksplit<2, 2, 8, 1, 2>},
        {"ksplit R2 CH3 NW8 inter  nt", launch_ksplit<2, 3, 8, 1, 2>},
        {"ksplit R2 CH6 NW8 inter  nt", launch_ksplit<2, 6, 8, 1, 2>},
        {"ksplit R4 CH2 NW8 inter  nt", launch_ksplit<4, 2, 8, 1, 2>},
        {"ksplit R4 CH3 NW8 inter  nt", launch_ksplit<4, 3, 8, 1, 2>},
        {"ksplit R1 CH6 NW8 inter  nt", launch_ksplit<1, 6, 8, 1, 2>},
        {"ksplit R2 CH4 NW2 inter  nt", launch_ksplit<2, 4, 2, 1, 2>},
        {"ksplit R2 CH8 NW2 inter  nt", launch_ksplit<2, 8, 2, 1, 2>},
        {"ksplit R2 CH4 NW16 inter nt", launch_ksplit<2, 4, 16, 1, 2>},
    };
    struct Shape { const char* name; int rows, K; } shapes[] = {
        {"wo    4096x4096 ", 4096, 4096}, {"w2    4096x11008", 4096, 11008},
        {"w1    11008x4096", 11008, 4096}, {"qkv  12288x4096 ", 12288, 4096}, {"cls  32000x4096 ", 32000, 4096},
    };
    const size_t max_bytes = (size_t)32000 * 4096 * 4;
    const int nbuf = 6;   // 6 x 524 MB = 3.1 GB rotation
    std::vector<float*> W(nbuf);
    for (auto& p : W) { CK(hipMalloc(&p, max_bytes)); CK(hipMemset(p, 0x11, max_bytes)); }
    float *x, *o; CK(hipMalloc(&x, 11008 * 4)); CK(hipMemset(x, 0, 11008 * 4)); CK(hipMalloc(&o, 32000 * 4));
    // read-bandwidth probe
    {
        size_t total = max_bytes / 16;   // f4 count
        int blocks = 2048;
        size_t per = total / blocks;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 12; i++) hipLaunchKernelGGL((read_probe<8, 2>), dim3(blocks), dim3(256), 0, st, W[i % nbuf], o, per);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("read_probe nt 2048 blocks: %.1f us per 524MB -> %.0f GB/s\n", ms * 1e3 / 12, (double)per * blocks * 16 * 12 / (ms * 1e-3) / 1e9);
        }
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 12; i++) hipLaunchKernelGGL((read_probe<8, 0>), dim3(blocks), dim3(256), 0, st, W[i % nbuf], o, per);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("read_probe    2048 blocks: %.1f us per 524MB -> %.0f GB/s\n", ms * 1e3 / 12, (double)per * blocks * 16 * 12 / (ms * 1e-3) / 1e9);
        }
    }
    for (auto& sh : shapes) {
        double bytes = (double)sh.rows * sh.K * 4;
        // stride between matrices inside one buffer so consecutive launches never reuse a line
        size_t mat_floats = (size_t)sh.rows * sh.K;
        int per_buf = (int)std::max<size_t>(1, max_bytes / 4 / mat_floats);
        printf("---- %s (%.1f MB)\n", sh.name, bytes / 1e6);
        for (auto& v : vs) {
            float best = 1e9, sum = 0; int cnt = 0;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; i++) {
                    int b = i % nbuf, sub = (i / nbuf) % per_buf;
                    v.launch(W[b] + (size_t)sub * mat_floats, x, o, sh.K, sh.rows, st);
                }
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) { best = std::min(best, ms); sum += ms; cnt++; }
            }
            double us = best * 1e3 / iters;
            printf("  %-30s %8.2f us  %7.0f GB/s  (avg %.2f us)\n", v.name.c_str(), us, bytes / (us * 1e-6) / 1e9, sum / cnt * 1e3 / iters);
        }
    }
    CK(hipGetLastError());
    return 0;
}
