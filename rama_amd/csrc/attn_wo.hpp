// attn_wo.hpp -- attention + the Wo matvec as ONE launch (infer.rs:34-37), used where launches are
// latency-bound (dim <= 1024: +4..8 % tokens/s at the stories shapes; rama_set_tuning "merge").
// The first n_heads workgroups run one head's attention each and publish xb with write-through
// (sc1) stores; every workgroup streams its 4 rows of Wo into registers meanwhile, waits for the
// n_heads arrivals on a counter, reads xb with sc1 loads and finishes x += Wo . xb.  Hand-off
// protocol: cdna_hip_programming.md Guideline 16 "R1" (sc1 payload, producer drains vmcnt, relaxed
// agent-scope counter, one polling lane, workgroup barrier, sc1 loads); the whole grid is resident
// (checked with the occupancy API by the host), every spin is bounded.
#pragma once
#include "kernels.hpp"

namespace rama {

constexpr int kPWaves = 8;
constexpr int kPThreads = kPWaves * 64;
constexpr int kPAttnU = 4;    // cache rows in flight per lane in the attention phase

// ---- sc1 (write-through / L1-bypassing) accessors for inter-workgroup data
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f4 ld4_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16));   // cpol sc1
}


struct AttnHeadArgs { int dim, n_heads, seq_len; const float* q; float* xb; };

// ---------------------------------------------------------------- attention for one head, 8 waves
// (same arithmetic as attention_kernel; q, k, v and the cache row of `pos` were written by other
// workgroups in the phase before, so every load of them is sc1)
template <int G>
__device__ __forceinline__ void p_attention(const AttnHeadArgs& p, const float* kc, const float* vc, int h, int pos,
                                         float* lds) {
    float* s_max = lds;
    float* s_sum = lds + kPWaves;
    float* s_acc = lds + 2 * kPWaves;
    float* s_att = lds + 2 * kPWaves + kPWaves * G * 4;
    constexpr int U = kPAttnU;
    constexpr int TPW = 64 / G;
    constexpr int TILE = kPWaves * TPW * U;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hs = p.dim / p.n_heads;
    const int li = lane % G, tg = lane / G;
    const bool lane_ok = li * 4 < hs;
    const unsigned cache_bytes = (unsigned)p.seq_len * (unsigned)p.dim * 4u;
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(kc, cache_bytes);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(vc, cache_bytes);
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(p.q, (unsigned)p.dim * 4u);
    const unsigned col = (unsigned)(h * hs + li * 4) * 4u;
    const unsigned rowb = (unsigned)p.dim * 4u;
    auto t_of = [&](int base, int u) { return base + (u * kPWaves + wave) * TPW + tg; };
    auto off_of = [&](int t) { return (lane_ok && t <= pos) ? (unsigned)t * rowb + col : kOOB; };

    const f4 q4 = ld4_sc1(rq, lane_ok ? col : kOOB);
    const float div = sqrtf((float)hs);
    f4 kt[U], vt[U];
#pragma unroll
    for (int u = 0; u < U; u++) kt[u] = ld4_sc1(rk, off_of(t_of(0, u)));
#pragma unroll
    for (int u = 0; u < U; u++) vt[u] = ld4_sc1(rv, off_of(t_of(0, u)));
    __builtin_amdgcn_sched_barrier(0);
    for (int base = 0; base <= pos; base += TILE) {
        if (base > 0) {
#pragma unroll
            for (int u = 0; u < U; u++) kt[u] = ld4_sc1(rk, off_of(t_of(base, u)));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float d = dot4(q4, kt[u], 0.0f);
            d = row16_sum(d);
            if (G == 32) d += __shfl_xor(d, 16);
            if (G == 64) { d += __shfl_xor(d, 16); d += __shfl_xor(d, 32); }
            const int t = t_of(base, u);
            if (li == 0 && t <= pos) s_att[t] = d / div;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = tid; t <= pos; t += kPThreads) mx = fmaxf(mx, s_att[t]);
    mx = wave_max(mx);
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();
    mx = s_max[0];
#pragma unroll
    for (int w = 1; w < kPWaves; w++) mx = fmaxf(mx, s_max[w]);
    float sum = 0.0f;
    for (int t = tid; t <= pos; t += kPThreads) {
        float e = expf(s_att[t] - mx);
        s_att[t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    {
        float t8[kPWaves];
#pragma unroll
        for (int w = 0; w < kPWaves; w++) t8[w] = s_sum[w];
#pragma unroll
        for (int n = kPWaves; n > 1; n >>= 1)
#pragma unroll
            for (int w = 0; w < n / 2; w++) t8[w] = t8[2 * w] + t8[2 * w + 1];
        sum = t8[0];
    }
    for (int t = tid; t <= pos; t += kPThreads) s_att[t] = s_att[t] / sum;
    __syncthreads();
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base <= pos; base += TILE) {
        if (base > 0) {
#pragma unroll
            for (int u = 0; u < U; u++) vt[u] = ld4_sc1(rv, off_of(t_of(base, u)));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t_of(base, u);
            const float a = (t <= pos) ? s_att[t] : 0.0f;
            acc.x = fmaf(a, vt[u].x, acc.x); acc.y = fmaf(a, vt[u].y, acc.y);
            acc.z = fmaf(a, vt[u].z, acc.z); acc.w = fmaf(a, vt[u].w, acc.w);
        }
    }
#pragma unroll
    for (int m = G; m < 64; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m); acc.y += __shfl_xor(acc.y, m);
        acc.z += __shfl_xor(acc.z, m); acc.w += __shfl_xor(acc.w, m);
    }
    if (lane < G) *reinterpret_cast<f4*>(s_acc + (wave * G + lane) * 4) = acc;
    __syncthreads();
    if (tid < G && tid * 4 < hs) {      // tid < 64: wave 0 does every inter-workgroup store
        f4 t8[kPWaves];
#pragma unroll
        for (int w = 0; w < kPWaves; w++) t8[w] = *reinterpret_cast<f4*>(s_acc + (w * G + tid) * 4);
#pragma unroll
        for (int n = kPWaves; n > 1; n >>= 1)
#pragma unroll
            for (int w = 0; w < n / 2; w++) t8[w] = t8[2 * w] + t8[2 * w + 1];
        float* o = p.xb + (size_t)h * hs + (size_t)tid * 4;
        st_sc1(o, t8[0].x); st_sc1(o + 1, t8[0].y); st_sc1(o + 2, t8[0].z); st_sc1(o + 3, t8[0].w);
    }
}

__host__ __device__ constexpr int p_attn_lds_floats(int G, int seq_len) { return 2 * kPWaves + kPWaves * G * 4 + seq_len; }

// ---------------------------------------------------------------- attention + Wo in one launch
// Decode attention keeps 32 of 256 CUs busy for ~5.6 us while HBM idles, and the Wo matvec that
// follows pays its own ramp.  Here the grid is Wo's grid (dim/4 workgroups of 8 waves): every
// workgroup requests its Wo weight tiles FIRST (weights do not depend on activations), workgroups
// 0..n_heads-1 run attention for one head each (they request their tiles afterwards: they are the
// critical path), and everyone then waits for the n_heads arrivals before reading xb.
//  * Hand-off = the same sc1 protocol as above: xb is written with sc1 stores by wave 0 of the
//    attention workgroups, `s_waitcnt vmcnt(0)`, one agent-scope add; consumers poll from one lane,
//    pass a workgroup barrier and read xb with sc1 loads only.
//  * No deadlock: the host launches this kernel only when the occupancy API says the WHOLE grid is
//    resident at once (else it falls back to the two separate launches), so the attention
//    workgroups always run.  The spin is bounded anyway.
//  * The counter is zeroed by the QKV launch that precedes this one in the stream.
struct AttnWoParams {
    int dim, n_heads, seq_len;
    const float* q; const float* kc; const float* vc;   // this layer's cache slabs
    float* xb; float* x;
    const float* wo;                                    // this layer's [dim, dim]
    const Ctl* ctl;
    unsigned* counter;                                  // arrivals of the attention workgroups
    unsigned long long* err;
};

// launch bound: 8 waves per SIMD = 4 workgroups per CU, so the 1024-workgroup grid of llama2-7B
// (dim 4096 / 4 rows) is resident at once on 256 CUs; it caps the kernel at 64 VGPRs.
template <int G>
__global__ __launch_bounds__(kPThreads, 8) void attn_wo_kernel(AttnWoParams a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int R = 4, CH = 2, NW = kPWaves;
    float (*part)[R] = reinterpret_cast<float (*)[R]>(lds);     // [NW][R]
    float* scratch = lds + NW * R;                              // attention scratch
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.dim, rows = a.dim;
    const int r0 = b * R;
    const int nch = (K + 255) >> 8;
    const unsigned kbytes = (unsigned)K * 4u;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wo, (unsigned)rows * kbytes);
    unsigned rowoff[R];
#pragma unroll
    for (int s = 0; s < R; s++) rowoff[s] = (r0 + s < rows) ? (unsigned)(r0 + s) * kbytes : kOOB;
    unsigned kb[CH];
#pragma unroll
    for (int j = 0; j < CH; j++) {
        const int c = wave + j * NW;
        const unsigned o = (unsigned)(c * 1024 + lane * 16);
        kb[j] = (c < nch && o < kbytes) ? o : kOOB;
    }
    f4 w[R][CH];
    auto issue_first = [&]() {
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int s = 0; s < R; s++) w[s][j] = ld_nt(rw, (kb[j] == kOOB || rowoff[s] == kOOB) ? kOOB : rowoff[s] + kb[j]);
    };
    float resid = 0.0f;
    if (tid < R && r0 + tid < rows) resid = a.x[r0 + tid];      // x: complete since the previous launch

    if (b < a.n_heads) {
        AttnHeadArgs p{a.dim, a.n_heads, a.seq_len, a.q, a.xb};
        p_attention<G>(p, a.kc, a.vc, b, a.ctl->pos, scratch);
        if (tid < 64) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // wave 0's sc1 stores of xb have left
            if (tid == 0) __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        issue_first();
    } else {
        issue_first();
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tid == 0) {
        long spins = 0;
        while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.n_heads) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1L << 22)) {
                __hip_atomic_store(a.err, 0x2000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.xb, kbytes);
    float acc[R] = {0.f, 0.f, 0.f, 0.f};
    {
        f4 xv[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) xv[j] = ld4_sc1(rx, kb[j]);
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int s = 0; s < R; s++) acc[s] = dot4(w[s][j], xv[j], acc[s]);
    }
    for (int c0 = wave + CH * NW; c0 < nch; c0 += CH * NW) {    // rows wider than 16 chunks: the rest, un-prefetched
        f4 ww[R][CH], xv[CH];
        unsigned kk[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int c = c0 + j * NW;
            const unsigned o = (unsigned)(c * 1024 + lane * 16);
            kk[j] = (c < nch && o < kbytes) ? o : kOOB;
        }
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int s = 0; s < R; s++) ww[s][j] = ld_nt(rw, (kk[j] == kOOB || rowoff[s] == kOOB) ? kOOB : rowoff[s] + kk[j]);
#pragma unroll
        for (int j = 0; j < CH; j++) xv[j] = ld4_sc1(rx, kk[j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int s = 0; s < R; s++) acc[s] = dot4(ww[s][j], xv[j], acc[s]);
    }
#pragma unroll
    for (int s = 0; s < R; s++) acc[s] = wave_sum(acc[s]);
    if (lane == 0) {
#pragma unroll
        for (int s = 0; s < R; s++) part[wave][s] = acc[s];
    }
    __syncthreads();
    if (tid < R && r0 + tid < rows) {
        float t8[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) t8[q] = part[q][tid];
#pragma unroll
        for (int n = NW; n > 1; n >>= 1)
#pragma unroll
            for (int q = 0; q < n / 2; q++) t8[q] = t8[2 * q] + t8[2 * q + 1];
        a.x[r0 + tid] = resid + t8[0];      // infer.rs:37
    }
}

}  // namespace rama
