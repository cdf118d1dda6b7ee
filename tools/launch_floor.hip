// launch_floor.hip -- what does ONE dependent kernel cost inside a replayed hipGraph on this box?
// Not part of the product: sizes the floor the small-model decode path (stories15M / 110M: ~26-50
// dependent launches per token) is up against.  Build: hipcc --offload-arch=gfx950 -O3 -o launch_floor tools/launch_floor.hip
#include "../rama_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_empty() {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
// dependent chain: every workgroup reads what the previous kernel wrote, streams `per_wg` floats of
// "weights", reduces, writes
__global__ __launch_bounds__(256) void k_stream(const float* w, const float* x, float* o, int per_wg) {
    __shared__ float red[4];
    const float xv = x[threadIdx.x & 63];
    typedef __attribute__((ext_vector_type(4))) float f4;
    const f4* w4 = reinterpret_cast<const f4*>(w + (size_t)blockIdx.x * per_wg);
    float acc = 0.f;
    for (int i = threadIdx.x; i < per_wg / 4; i += 256) { f4 v = __builtin_nontemporal_load(w4 + i); acc += (v.x + v.y + v.z + v.w) * xv; }
    for (int m = 32; m; m >>= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) o[blockIdx.x & 63] = red[0] + red[1] + red[2] + red[3];
}

// the same kernel as N distinct code objects (does alternating kernels cost more than repeating one?)
template <int ID>
__global__ __launch_bounds__(256) void k_stream_id(const float* w, const float* x, float* o, int per_wg) {
    __shared__ float red[4];
    const float xv = x[threadIdx.x & 63] + (float)ID;
    typedef __attribute__((ext_vector_type(4))) float f4;
    const f4* w4 = reinterpret_cast<const f4*>(w + (size_t)blockIdx.x * per_wg);
    float acc = 0.f;
    for (int i = threadIdx.x; i < per_wg / 4; i += 256) { f4 v = __builtin_nontemporal_load(w4 + i); acc += (v.x + v.y + v.z + v.w) * xv; }
    for (int m = 32; m; m >>= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) o[blockIdx.x & 63] = red[0] + red[1] + red[2] + red[3];
}
struct BigParams { const float* w[3]; const float* x; const float* nw; float* o[3]; int K, rows, nmat; const void* ctl; int pos_val; const float* fr; const float* fi; int hs; float* kc; float* vc; unsigned* z; };
__global__ __launch_bounds__(256) void k_stream_big(BigParams p, int per_wg) {
    __shared__ float red[4];
    const float xv = p.x[threadIdx.x & 63];
    typedef __attribute__((ext_vector_type(4))) float f4;
    const f4* w4 = reinterpret_cast<const f4*>(p.w[0] + (size_t)blockIdx.x * per_wg);
    float acc = 0.f;
    for (int i = threadIdx.x; i < per_wg / 4; i += 256) { f4 v = __builtin_nontemporal_load(w4 + i); acc += (v.x + v.y + v.z + v.w) * xv; }
    for (int m = 32; m; m >>= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p.o[0][blockIdx.x & 63] = red[0] + red[1] + red[2] + red[3];
}

template <class F>
static double graph_us_per_kernel(hipStream_t st, int n, F enqueue) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; i++) enqueue(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const int reps = 20;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3 / (reps * n);
}

int main() {
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *w, *a, *b;
    const size_t wbytes = (size_t)512 << 20;
    CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 0, wbytes)); CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096));
    CK(hipMemset(a, 0, 4096)); CK(hipMemset(b, 0, 4096));
    const int n = 120;
    printf("empty kernel, 1 WG          : %.2f us per kernel\n", graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }));
    printf("empty kernel, 256 WG x 256  : %.2f us per kernel\n", graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st); }));
    printf("empty kernel, 1024 WG x 512 : %.2f us per kernel\n", graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(512), 0, st); }));
    printf("touch one float             : %.2f us per kernel\n", graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, a); }));
    for (int wgs : {64, 256, 1024}) {
        for (int kb : {1, 4, 16, 64}) {           // KiB of weights per workgroup
            const int per = kb * 256;
            size_t total = (size_t)wgs * per * 4;
            auto f = [&](int i) {
                const float* wp = w + ((size_t)i * total % (wbytes - total)) / 4 / 4 * 4;
                hipLaunchKernelGGL(k_stream, dim3(wgs), dim3(256), 0, st, wp, (i & 1) ? a : b, (i & 1) ? b : a, per);
            };
            printf("stream %4d WG x %2d KiB (%6.2f MB): %.2f us per kernel\n", wgs, kb, total / 1e6, graph_us_per_kernel(st, n, f));
        }
    }
    {
        const int wgs = 256, per = 4 * 256; size_t total = (size_t)wgs * per * 4;
        auto wp = [&](int i) { return w + ((size_t)i * total % (wbytes - total)) / 4 / 4 * 4; };
        printf("4 alternating kernels, 256 WG x 4 KiB : %.2f us per kernel\n", graph_us_per_kernel(st, n, [&](int i) {
            const float* x = (i & 1) ? a : b; float* o = (i & 1) ? b : a;
            switch (i & 3) {
                case 0: hipLaunchKernelGGL(k_stream_id<0>, dim3(wgs), dim3(256), 0, st, wp(i), x, o, per); break;
                case 1: hipLaunchKernelGGL(k_stream_id<1>, dim3(wgs), dim3(256), 0, st, wp(i), x, o, per); break;
                case 2: hipLaunchKernelGGL(k_stream_id<2>, dim3(wgs), dim3(256), 0, st, wp(i), x, o, per); break;
                default: hipLaunchKernelGGL(k_stream_id<3>, dim3(wgs), dim3(256), 0, st, wp(i), x, o, per); break;
            } }));
        printf("one kernel, big by-value params        : %.2f us per kernel\n", graph_us_per_kernel(st, n, [&](int i) {
            BigParams p{}; p.w[0] = wp(i); p.x = (i & 1) ? a : b; p.o[0] = (i & 1) ? b : a;
            hipLaunchKernelGGL(k_stream_big, dim3(wgs), dim3(256), 0, st, p, per); }));
        // the same weights every launch (cache resident, like stories15M's 60 MB)
        printf("same 1 MB every launch (cache resident): %.2f us per kernel\n", graph_us_per_kernel(st, n, [&](int i) {
            hipLaunchKernelGGL(k_stream, dim3(wgs), dim3(256), 0, st, w, (i & 1) ? a : b, (i & 1) ? b : a, per); }));
        // 512-thread workgroups, 1024 of them (the decode kernels' geometry)
        printf("1024 WG x 256 threads x 1 KiB          : %.2f us per kernel\n", graph_us_per_kernel(st, n, [&](int i) {
            hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, st, wp(i), (i & 1) ? a : b, (i & 1) ? b : a, 256); }));
    }
    // ---- the product's own decode kernels at the stories15M / 110M shapes, chained in a graph
    {
        using namespace rama;
        float *W2, *xa, *xb2, *hb, *lg, *kc; Ctl* ctl; int* out; int* forced;
        CK(hipMalloc(&W2, (size_t)128 << 20)); CK(hipMemset(W2, 0, (size_t)128 << 20));
        CK(hipMalloc(&xa, 1 << 16)); CK(hipMalloc(&xb2, 1 << 16)); CK(hipMalloc(&hb, 1 << 16)); CK(hipMalloc(&lg, 32000 * 4)); CK(hipMalloc(&kc, (size_t)64 << 20));
        CK(hipMemset(xa, 0, 1 << 16)); CK(hipMemset(xb2, 0, 1 << 16)); CK(hipMemset(hb, 0, 1 << 16)); CK(hipMemset(lg, 0, 32000 * 4)); CK(hipMemset(kc, 0, (size_t)64 << 20));
        CK(hipMalloc(&ctl, sizeof(Ctl))); CK(hipMemset(ctl, 0, sizeof(Ctl))); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&forced, 4096)); CK(hipMemset(forced, 0, 4096));
        struct Sh { const char* name; int dim, hidden, heads, seq; } shs[] = {{"stories15M", 288, 768, 6, 256}, {"stories110M", 768, 2048, 12, 1024}};
        for (auto& sh : shs) {
            const int dim = sh.dim, hid = sh.hidden, hs = dim / sh.heads;
            GemvParams q{}; q.w[0] = W2; q.w[1] = W2 + dim * dim; q.w[2] = W2 + 2 * dim * dim; q.x = xa; q.nw = xb2; q.o[0] = hb; q.o[1] = hb + dim; q.o[2] = hb + 2 * dim;
            q.K = dim; q.rows = dim; q.nmat = 3; q.ctl = ctl; q.fr = xb2; q.fi = xb2; q.head_size = hs; q.kc = kc; q.vc = kc + ((size_t)4 << 20);
            const dim3 gq((3 * ((dim + 3) / 4) + kSoloWaves - 1) / kSoloWaves);
            if (dim <= 512) printf("%s qkv  solo : %.2f us\n", sh.name, graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL((gemv_rows_solo<4, 2, true, EPI_QKV>), gq, dim3(kSoloWaves * 64), 0, st, q); }));
            else printf("%s qkv  solo : %.2f us\n", sh.name, graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL((gemv_rows_solo<4, 4, true, EPI_QKV>), gq, dim3(kSoloWaves * 64), 0, st, q); }));
            printf("%s qkv  8wave: %.2f us\n", sh.name, graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL((gemv_rows<4, 2, 8, true, EPI_QKV>), dim3(3 * (dim / 4)), dim3(512), 0, st, q); }));
            SwigluParams sw{}; sw.w1 = W2; sw.w3 = W2 + (size_t)hid * dim; sw.x = xa; sw.nw = xb2; sw.hb = hb; sw.K = dim; sw.rows = hid;
            const dim3 gs(((hid + 1) / 2 + kSoloWaves - 1) / kSoloWaves);
            if (dim <= 512) printf("%s w13  solo : %.2f us\n", sh.name, graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL((gemv_swiglu_solo<2, 2>), gs, dim3(kSoloWaves * 64), 0, st, sw); }));
            else printf("%s w13  solo : %.2f us\n", sh.name, graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL((gemv_swiglu_solo<2, 4>), gs, dim3(kSoloWaves * 64), 0, st, sw); }));
            GemvParams r{}; r.w[0] = W2; r.x = hb; r.o[0] = xa; r.K = hid; r.rows = dim; r.nmat = 1;
            const dim3 gr(((dim + 3) / 4 + kSoloWaves - 1) / kSoloWaves);
            printf("%s w2   solo : %.2f us\n", sh.name, graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL((gemv_rows_solo<4, 4, false, EPI_RESID>), gr, dim3(kSoloWaves * 64), 0, st, r); }));
            GemvParams o{}; o.w[0] = W2; o.x = xb2; o.o[0] = xa; o.K = dim; o.rows = dim; o.nmat = 1;
            printf("%s wo   solo : %.2f us\n", sh.name, graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL((gemv_rows_solo<4, 4, false, EPI_RESID>), gr, dim3(kSoloWaves * 64), 0, st, o); }));
            GemvParams cl{}; cl.w[0] = W2; cl.x = xa; cl.nw = xb2; cl.o[0] = lg; cl.K = dim; cl.rows = 32000; cl.nmat = 1;
            const dim3 gc((8000 + kSoloWaves - 1) / kSoloWaves);
            if (dim <= 512) printf("%s cls  solo : %.2f us\n", sh.name, graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL((gemv_rows_solo<4, 2, true, EPI_STORE>), gc, dim3(kSoloWaves * 64), 0, st, cl); }));
            else printf("%s cls  solo : %.2f us\n", sh.name, graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL((gemv_rows_solo<4, 4, true, EPI_STORE>), gc, dim3(kSoloWaves * 64), 0, st, cl); }));
            for (int pos : {8, 100, 250}) {
                AttnParams a{}; a.q = hb; a.kc = kc; a.vc = kc + ((size_t)4 << 20); a.att = nullptr; a.xb = xb2; a.ctl = nullptr; a.pos_val = pos; a.dim = dim; a.head_size = hs; a.seq_len = sh.seq;
                size_t shm = (size_t)(attn_scratch_floats(16) + sh.seq) * 4;
                printf("%s attention (separate kernel) pos %3d: %.2f us\n", sh.name, pos, graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL((attention_kernel<16, false>), dim3(sh.heads), dim3(kAttnThreads), shm, st, a); }));
                size_t shm4 = (size_t)(attn_scratch_floats(16, 4) + sh.seq) * 4;
                printf("%s attention 4 waves            pos %3d: %.2f us\n", sh.name, pos, graph_us_per_kernel(st, n, [&](int) { hipLaunchKernelGGL((attention_kernel<16, false, 4>), dim3(sh.heads), dim3(256), shm4, st, a); }));
            }
        }
        ArgmaxParams ap{}; ap.logits = lg; ap.n = 32000; ap.ctl = ctl; ap.forced = forced; ap.out = out; ap.out_cap = 1 << 18; ap.emb = W2; ap.x = xa; ap.dim = 288;
        printf("argmax + cursor + gather: %.2f us\n", graph_us_per_kernel(st, 20, [&](int) { hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, st, ap); }));
    }
    return 0;
}
