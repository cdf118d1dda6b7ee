// persist.hpp -- the whole decode step as ONE persistent launch (gfx950).
//
// Why: every separately launched matvec costs  bytes / 6.9 TB/s  +  ~2.5 us  of ramp and
// reduce/epilogue tail during which HBM idles (128 launches per llama2-7B token), and decode
// attention (32 workgroups, latency-bound) leaves HBM idle for its whole ~5.6 us.  Weights never
// depend on activations, so a workgroup that has finished phase i can already stream its first
// tiles of phase i+1 while it waits for the other workgroups: the phase boundary is hidden behind
// useful HBM traffic.
//
// Structure
//  * grid = one workgroup of 8 waves per CU (forced by a > 80 KiB LDS request), every workgroup
//    resident for the whole launch, so a counter barrier can never deadlock.
//  * phases, one loop: per layer QKV(+rmsnorm, RoPE, cache append) | attention (workgroups <
//    n_heads) | Wo(+residual) | W1|W3(+rmsnorm, SiLU*gate) | W2(+residual); then the classifier;
//    then argmax + cursor advance + next embedding gather (workgroup 0).  Same arithmetic and
//    the same streaming geometry (4 streams x 8 waves x 2 chunks, chunk c -> wave c mod 8) as
//    kernels.hpp; a workgroup owns row groups g = wg, wg + nwg, ... of every matvec.
//  * software pipeline: NPF weight steps are always in flight per wave (register ring), across
//    row groups AND across the phase barrier.
//  * hand-off protocol (cdna_hip_programming.md Guideline 16, "R1" form; MI355X_MICROARCH.md
//    hand-off table row 1): every byte another workgroup will read is written with an sc1
//    (write-through) store by wave 0, which then drains `s_waitcnt vmcnt(0)` and adds 1 to the
//    barrier counter (relaxed, agent scope); consumers poll the counter with sc1 loads from ONE
//    lane, pass a workgroup barrier, and read the handed-off bytes with sc1 loads only.  No
//    dispatch-order or placement assumption.  Every spin is bounded: on timeout the error word is
//    set and the workgroup falls through (results invalid, host reports the error).
#pragma once
#include "kernels.hpp"

namespace rama {

constexpr int kPWaves = 8;
constexpr int kPThreads = kPWaves * 64;
constexpr int kPS = 4;        // weight streams per row group (4 rows, or 2 (w1,w3) row pairs)
constexpr int kPCH = 2;       // 1-KiB chunks per wave per step
constexpr int kNPF = 3;       // weight steps in flight per wave
constexpr int kPAttnU = 4;    // cache rows in flight per lane in the attention phase

struct PersistParams {
    int dim, hidden, n_heads, vocab, seq_len;
    int n_layers;                      // layers held by this launch (stage-local indexing)
    int do_cls, do_argmax;
    const float *emb, *rms_att, *rms_ffn, *wq, *wk, *wv, *wo, *w1, *w2, *w3, *rms_final, *fr, *fi, *wcls;
    float *x, *xb, *hb, *q, *k, *v, *logits, *kc, *vc;
    Ctl* ctl;
    const int* forced; int* out; int out_cap;
    unsigned long long* bar;           // [0] arrival counter (monotonic over launches), [1] error word
    unsigned long long* stamps;        // diagnostic only (NULL in production): 100 MHz timestamps of workgroup `stamp_wg`
    int stamp_wg;
    unsigned long long* epoch;         // arrivals all completed launches have added to bar[0] (device word);
                                       // = this launch's barrier base, advanced by the last phase
    int nwg;
};

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

// ---- one matvec phase, wave-uniform scalars only (two of these are live at a time)
enum { PE_QKV = 0, PE_RESID = 1, PE_SWIGLU = 2, PE_STORE = 3 };
struct Work {
    const float* m0;      // matrices; QKV: wq|wk|wv, SWIGLU: w1|w3, else m0
    const float* m1;
    const float* m2;
    int K, rows;          // row width, rows per matrix
    int spg;              // steps per row group = ceil(nch / (8 * CH))
    int nch;              // 1-KiB chunks per row
    int epi;
    int steps;            // this workgroup's step count (0 for an attention phase)
};

struct StepBuf { f4 w[kPS][kPCH]; };

// phase index -> kind: 0 QKV, 1 attention, 2 Wo, 3 W1|W3, 4 W2 (per layer), 5 classifier
__device__ __forceinline__ int phase_kind(const PersistParams& p, int ph) { return ph < 5 * p.n_layers ? ph % 5 : 5; }

__device__ __forceinline__ Work work_of(const PersistParams& p, int ph, int wg) {
    Work w;
    w.m0 = w.m1 = w.m2 = nullptr; w.K = 4; w.rows = 0; w.spg = 1; w.nch = 1; w.epi = PE_STORE; w.steps = 0;
    const int kind = phase_kind(p, ph);
    const size_t li = (size_t)(ph / 5);
    const size_t dd = (size_t)p.dim * p.dim, hd = (size_t)p.hidden * p.dim;
    int groups = 0;
    if (kind == 0) {
        w.m0 = p.wq + li * dd; w.m1 = p.wk + li * dd; w.m2 = p.wv + li * dd;
        w.K = p.dim; w.rows = p.dim; groups = 3 * ((p.dim + kPS - 1) / kPS); w.epi = PE_QKV;
    } else if (kind == 2) {
        w.m0 = p.wo + li * dd; w.K = p.dim; w.rows = p.dim; groups = (p.dim + kPS - 1) / kPS; w.epi = PE_RESID;
    } else if (kind == 3) {
        w.m0 = p.w1 + li * hd; w.m1 = p.w3 + li * hd; w.K = p.dim; w.rows = p.hidden; groups = (p.hidden + 1) / 2; w.epi = PE_SWIGLU;
    } else if (kind == 4) {
        w.m0 = p.w2 + li * hd; w.K = p.hidden; w.rows = p.dim; groups = (p.dim + kPS - 1) / kPS; w.epi = PE_RESID;
    } else if (kind == 5) {
        w.m0 = p.wcls; w.K = p.dim; w.rows = p.vocab; groups = (p.vocab + kPS - 1) / kPS; w.epi = PE_STORE;
    } else {
        return w;      // attention: no weights
    }
    w.nch = (w.K + 255) >> 8;
    w.spg = (w.nch + kPWaves * kPCH - 1) / (kPWaves * kPCH);
    const int mine = groups > wg ? (groups - wg + p.nwg - 1) / p.nwg : 0;
    w.steps = mine * w.spg;
    return w;
}

// issue the loads of this workgroup's step (local group gl, step st within the group) of phase
// `w`; steps past the end load nothing (offsets out of range)
__device__ __forceinline__ void p_issue(StepBuf& b, const Work& w, int gl, int st, int wg, int nwg, int wave, int lane) {
    const int g = wg + gl * nwg;
    const bool live = gl * w.spg + st < w.steps;
    const unsigned kbytes = (unsigned)w.K * 4u;
    const unsigned mbytes = (unsigned)w.rows * kbytes;
    const bool pair = w.epi == PE_SWIGLU;
    // 4 consecutive rows of one matrix, or (w1 row, w3 row) x 2
    const int gpm = (w.rows + kPS - 1) / kPS;
    const int mi = pair ? 0 : (g >= 2 * gpm ? 2 : (g >= gpm ? 1 : 0));      // no integer division in the hot loop
    const int r0 = pair ? g * 2 : (g - mi * gpm) * kPS;
    const float* Ma = pair ? w.m0 : (mi == 0 ? w.m0 : (mi == 1 ? w.m1 : w.m2));
    const float* Mb = pair ? w.m1 : Ma;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(Ma, mbytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(Mb, mbytes);
    unsigned rowoff[kPS];
#pragma unroll
    for (int i = 0; i < kPS; i++) {
        const int r = pair ? r0 + (i >> 1) : r0 + i;
        rowoff[i] = (live && r < w.rows) ? (unsigned)r * kbytes : kOOB;
    }
#pragma unroll
    for (int j = 0; j < kPCH; j++) {
        const int c = wave + (st * kPCH + j) * kPWaves;
        const unsigned kb = (unsigned)(c * 1024 + lane * 16);
        const bool ok = c < w.nch && kb < kbytes;
#pragma unroll
        for (int i = 0; i < kPS; i++) {
            const unsigned o = (ok && rowoff[i] != kOOB) ? rowoff[i] + kb : kOOB;
            b.w[i][j] = ld_nt((i & 1) ? rb : ra, o);
        }
    }
}

// the first NPF steps of a phase into the ring (slot u = step u)
__device__ __forceinline__ void p_issue_head(StepBuf (&ring)[kNPF], const Work& w, int wg, int nwg, int wave, int lane) {
    int gl = 0, st = 0;
#pragma unroll
    for (int u = 0; u < kNPF; u++) {
        p_issue(ring[u], w, gl, st, wg, nwg, wave, lane);
        if (++st == w.spg) { st = 0; gl++; }
    }
}

// barrier state carried through the launch
struct Bar {
    unsigned long long* ctr;
    unsigned long long* err;
    unsigned long long base;   // counter value every workgroup had reached before this launch
    int n;                     // barriers passed in this launch
    int nwg;
};

// Arrive: wave 0 (the only wave that stores handed-off data) drains its stores and adds 1.
__device__ __forceinline__ void bar_arrive(Bar& b) {
    if (threadIdx.x < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_fetch_add(b.ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    b.n += 1;
}
// Wait for every workgroup's arrival (bounded), then release the whole workgroup.
__device__ __forceinline__ void bar_wait(Bar& b) {
    if (threadIdx.x == 0) {
        const unsigned long long target = b.base + (unsigned long long)b.n * (unsigned long long)b.nwg;
        long spins = 0;
        while (__hip_atomic_load(b.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1L << 22)) {     // seconds: something is badly wrong -- do not hang the GPU
                __hip_atomic_store(b.err, (unsigned long long)(0x1000 + b.n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// stage the activation vector into LDS (xe = x * gain when a norm is folded in) and return the
// rmsnorm scale (1 when no norm).  All loads of handed-off data are sc1.
__device__ __forceinline__ float stage_x(const float* xin, const float* nw, int K, float* lds_x, float* red,
                                         const float* resid, int nres, float* lds_r) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (resid) {     // x += W.y epilogues read their x[r] from LDS instead of a dependent global load per row group
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(resid, (unsigned)nres * 4u);
        for (int i = tid; i < (nres >> 2); i += kPThreads) *reinterpret_cast<f4*>(lds_r + 4 * i) = ld4_sc1(rr, (unsigned)i * 16u);
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xin, (unsigned)K * 4u);
    const __amdgpu_buffer_rsrc_t rn = make_rsrc(nw ? nw : xin, (unsigned)K * 4u);
    float ss = 0.0f;
    const int n4 = K >> 2;
    for (int i = tid; i < n4; i += kPThreads) {
        f4 xv = ld4_sc1(rx, (unsigned)i * 16u);
        if (nw) {
            const f4 g = ld_c(rn, (unsigned)i * 16u);       // weights: never written, plain load
            ss = dot4(xv, xv, ss);
            xv = xv * g;
        }
        *reinterpret_cast<f4*>(lds_x + 4 * i) = xv;
    }
    float v = 1.0f;
    if (nw) {
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        float t[kPWaves];
#pragma unroll
        for (int i = 0; i < kPWaves; i++) t[i] = red[i];
#pragma unroll
        for (int n = kPWaves; n > 1; n >>= 1)
#pragma unroll
            for (int i = 0; i < n / 2; i++) t[i] = t[2 * i] + t[2 * i + 1];
        v = rms_scale(t[0], K);
    }
    __syncthreads();
    return v;
}

// epilogue of one finished row group (threads 0..3 of wave 0); d[] = the 4 stream sums
__device__ __forceinline__ void p_epilogue(const PersistParams& p, const Work& w, int kind, size_t li, int g, int t,
                                           const float (&d)[kPS], float v, int pos, const float* lds_r, const float* lds_rope) {
    if (w.epi == PE_SWIGLU) {
        if (t < 2) {
            const int r = g * 2 + t;
            if (r < w.rows) {
                float a = d[2 * t] * v;
                const float b = d[2 * t + 1] * v;
                a = a * (1.0f / (1.0f + expf(-a)));        // cpu.rs:56
                st_sc1(p.hb + r, a * b);                   // cpu.rs:59-64
            }
        }
        return;
    }
    const int gpm = (w.rows + kPS - 1) / kPS;
    const int mi = g >= 2 * gpm ? 2 : (g >= gpm ? 1 : 0);
    const int r0 = (g - mi * gpm) * kPS;
    if (w.epi == PE_QKV) {
        if (t < 2) {
            const int hs = p.dim / p.n_heads;
            const int r = r0 + 2 * t;
            float a = d[2 * t] * v, b = d[2 * t + 1] * v;
            if (mi < 2) {   // cpu.rs:87-96
                const int i = (r % hs) >> 1;
                const float c = lds_rope[i];                  // row `pos` of freq_cis_real / _imag, staged once
                const float s = lds_rope[(hs >> 1) + i];
                const float ra_ = a * c - b * s, rb_ = a * s + b * c;
                a = ra_; b = rb_;
            }
            float* o = mi == 0 ? p.q : (mi == 1 ? p.k : p.v);
            st_sc1(o + r, a); st_sc1(o + r + 1, b);
            const size_t crow = (li * p.seq_len + (size_t)pos) * p.dim + (size_t)r;     // infer.rs:31-33
            if (mi == 1) { st_sc1(p.kc + crow, a); st_sc1(p.kc + crow + 1, b); }
            if (mi == 2) { st_sc1(p.vc + crow, a); st_sc1(p.vc + crow + 1, b); }
        }
        return;
    }
    if (t < kPS && r0 + t < w.rows) {
        float val = d[t] * v;
        float* out = kind == 5 ? p.logits : p.x;
        if (w.epi == PE_RESID) val = lds_r[r0 + t] + val;            // infer.rs:37,47 (x staged at phase start)
        st_sc1(out + r0 + t, val);
    }
}

// ---------------------------------------------------------------- attention for one head, 8 waves
// (same arithmetic as attention_kernel; q, k, v and the cache row of `pos` were written by other
// workgroups in the phase before, so every load of them is sc1)
template <int G>
__device__ __forceinline__ void p_attention(const PersistParams& p, const float* kc, const float* vc, int h, int pos,
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

// ---------------------------------------------------------------- the decode-step kernel
// G = lanes per cached row in the attention phase (16 / 32 / 64 for head_size <= 64 / 128 / 256):
// a template parameter so that only one attention body is compiled into each kernel
template <int G>
__global__ __launch_bounds__(kPThreads) void decode_step_kernel(PersistParams p) {
    extern __shared__ float lds[];
    float (*part)[kPWaves][kPS] = reinterpret_cast<float (*)[kPWaves][kPS]>(lds);      // [2][8][4]
    float* red = lds + 2 * kPWaves * kPS;                                              // [16]
    float* lds_rope = red + 16;                                                        // [256] cos | sin of row pos
    float* lds_r = lds_rope + 256;                                                     // [dim] residual x
    float* lds_x = lds_r + ((p.dim + 3) & ~3);                                         // [max(dim, hidden)] / attention scratch
    const int wg = blockIdx.x, nwg = p.nwg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos = p.ctl->pos;
    const int nph = 5 * p.n_layers + (p.do_cls ? 1 : 0);     // phases = barriers per launch
    // launches on one context may differ in their number of phases (different models / stages),
    // so the base is the running total of arrivals, not launches x phases
    Bar bar{p.bar, p.bar + 1, *p.epoch, 0, nwg};

    StepBuf ring[kNPF];
    Work cur = work_of(p, 0, wg);
    p_issue_head(ring, cur, wg, nwg, wave, lane);
    {   // RoPE row of this position (constant tables: plain loads), read by the QKV epilogues
        const int half = (p.dim / p.n_heads) >> 1;
        if (tid < half) {
            lds_rope[tid] = p.fr[(size_t)pos * half + tid];
            lds_rope[half + tid] = p.fi[(size_t)pos * half + tid];
        }
    }

    auto stamp = [&](int ph, int k) {
        if (p.stamps && wg == p.stamp_wg && tid == 0) p.stamps[ph * 8 + k] = __builtin_amdgcn_s_memrealtime();
    };
    for (int ph = 0; ph < nph; ph++) {
        const int kind = phase_kind(p, ph);
        stamp(ph, 0);
        const size_t li = (size_t)(ph / 5);
        // what the ring must hold when this phase ends: the next MATVEC phase's first steps
        // (after QKV comes attention, which has no weights: Wo is requested during attention)
        const bool more = ph + 1 < nph;
        const Work nxt = work_of(p, more ? ph + 1 : ph, wg);
        const bool prefetch_next = more && kind != 0;

        if (kind == 1) {
            // ---- attention (workgroups < n_heads); `nxt` = Wo.  The attention workgroups are
            // the critical path, so they request their Wo tiles only afterwards.
            if (wg < p.n_heads) {
                const float* kc = p.kc + li * p.seq_len * p.dim;
                const float* vc = p.vc + li * p.seq_len * p.dim;
                p_attention<G>(p, kc, vc, wg, pos, lds_x);
                bar_arrive(bar);
                p_issue_head(ring, nxt, wg, nwg, wave, lane);
            } else {
                p_issue_head(ring, nxt, wg, nwg, wave, lane);
                bar_arrive(bar);
            }
            stamp(ph, 3);
            bar_wait(bar);
            stamp(ph, 4);
            cur = nxt;
            continue;
        }

        // ---- matvec phase: ring holds cur's steps 0..NPF-1 (issued before the last barrier)
        const float* xin = kind == 2 ? p.xb : (kind == 4 ? p.hb : p.x);
        const float* nw = kind == 0 ? p.rms_att + li * p.dim : (kind == 3 ? p.rms_ffn + li * p.dim : (kind == 5 ? p.rms_final : nullptr));
        const float v = stage_x(xin, nw, cur.K, lds_x, red, cur.epi == PE_RESID ? p.x : nullptr, p.dim, lds_r);
        stamp(ph, 1);

        float acc[kPS] = {0.f, 0.f, 0.f, 0.f};
        int par = 0;
        // the step count is padded to a multiple of NPF (dead steps load nothing), so ring slot u
        // always holds a step = u (mod NPF) and the hand-over to the next phase stays aligned
        const int psteps = ((cur.steps + kNPF - 1) / kNPF) * kNPF;
        // cursors (local group, step in group): c* = the step being consumed, i* = the step NPF
        // ahead that refills the slot, n* = the next phase's step u -- advanced incrementally, no
        // integer division in the loop
        int cgl = 0, cst = 0, igl = 0, ist = 0, ngl = 0, nst = 0;
#pragma unroll
        for (int u = 0; u < kNPF; u++) { if (++ist == cur.spg) { ist = 0; igl++; } }
        for (int s0 = 0; s0 < psteps; s0 += kNPF) {
#pragma unroll
            for (int u = 0; u < kNPF; u++) {
                const int s = s0 + u;
                const bool live = s < cur.steps;
                if (live) {
#pragma unroll
                    for (int j = 0; j < kPCH; j++) {
                        const int c = wave + (cst * kPCH + j) * kPWaves;
                        f4 xe = {0.f, 0.f, 0.f, 0.f};
                        if (c < cur.nch && c * 256 + lane * 4 < cur.K) xe = *reinterpret_cast<const f4*>(lds_x + c * 256 + lane * 4);
#pragma unroll
                        for (int i = 0; i < kPS; i++) acc[i] = dot4(ring[u].w[i][j], xe, acc[i]);
                    }
                }
                // refill the slot: own step NPF ahead, or -- at the end of the phase -- the next
                // phase's step u (wave 0 defers that until it has signalled the barrier: its
                // `s_waitcnt vmcnt(0)` there must cover only its stores)
                if (s + kNPF < psteps) p_issue(ring[u], cur, igl, ist, wg, nwg, wave, lane);
                else if (prefetch_next && wave != 0) p_issue(ring[u], nxt, ngl, nst, wg, nwg, wave, lane);
                if (s + kNPF >= psteps) { if (++nst == nxt.spg) { nst = 0; ngl++; } }
                if (++ist == cur.spg) { ist = 0; igl++; }
                if (live && cst == cur.spg - 1) {      // row group complete
#pragma unroll
                    for (int i = 0; i < kPS; i++) acc[i] = wave_sum(acc[i]);
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < kPS; i++) part[par][wave][i] = acc[i];
                    }
#pragma unroll
                    for (int i = 0; i < kPS; i++) acc[i] = 0.0f;
                    __syncthreads();
                    if (tid < kPS) {
                        float d[kPS];
#pragma unroll
                        for (int i = 0; i < kPS; i++) {
                            float t8[kPWaves];
#pragma unroll
                            for (int q = 0; q < kPWaves; q++) t8[q] = part[par][q][i];
#pragma unroll
                            for (int n = kPWaves; n > 1; n >>= 1)
#pragma unroll
                                for (int q = 0; q < n / 2; q++) t8[q] = t8[2 * q] + t8[2 * q + 1];
                            d[i] = t8[0];
                        }
                        p_epilogue(p, cur, kind, li, wg + cgl * nwg, tid, d, v, pos, lds_r, lds_rope);
                    }
                    par ^= 1;
                }
                if (++cst == cur.spg) { cst = 0; cgl++; }
            }
        }
        if (psteps == 0 && prefetch_next && wave != 0) p_issue_head(ring, nxt, wg, nwg, wave, lane);   // no row group of this phase here
        stamp(ph, 2);
        bar_arrive(bar);
        if (prefetch_next && wave == 0) p_issue_head(ring, nxt, wg, nwg, wave, lane);
        stamp(ph, 3);
        bar_wait(bar);
        stamp(ph, 4);
        cur = nxt;
    }

    // ---- argmax + cursor advance + next embedding gather (workgroup 0); epoch bump
    if (wg == 0) {
        if (p.do_argmax) {
            float* s_v = red;                                   // [8]
            int* s_i = reinterpret_cast<int*>(red + 8);         // [8]; the block's result goes to s_i[0]
            float bv = -INFINITY; int bi = -1;
            for (int i = tid; i < p.vocab; i += kPThreads) {     // ascending per thread: last max wins
                const float val = ld_sc1(p.logits + i);
                if (!(bv > val)) { bv = val; bi = i; }
            }
            const float wm = wave_max(bv);
            const int wi = wave_max_i(bv == wm ? bi : -1);
            if (lane == 0) { s_v[wave] = wm; s_i[wave] = wi; }
            __syncthreads();
            if (tid == 0) {
                float bestv = s_v[0]; int idx = s_i[0];
                for (int w = 1; w < kPWaves; w++) {
                    const float ov = s_v[w]; const int oi = s_i[w];
                    if (oi >= 0 && (idx < 0 || ov > bestv || (ov == bestv && oi > idx))) { bestv = ov; idx = oi; }
                }
                Ctl* c = p.ctl;
                int next = idx;
                if (pos < c->n_forced) next = p.forced[pos];
                const int n_out = c->n_out;
                if (n_out < p.out_cap) p.out[n_out] = next;
                c->n_out = n_out + 1; c->token = next; c->pos = pos + 1;
                s_i[0] = next;
            }
            __syncthreads();
            const size_t base = (size_t)s_i[0] * p.dim;
            for (int i = tid; i < p.dim; i += kPThreads) p.x[i] = p.emb[base + i];
        }
        // read by the next launch (kernel boundary orders it); every workgroup has read the old
        // value before its first barrier, which precedes this point
        if (tid == 0) *p.epoch = bar.base + (unsigned long long)nph * (unsigned long long)nwg;
    }
}

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
    extern __shared__ float lds[];
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
        PersistParams p{};
        p.dim = a.dim; p.n_heads = a.n_heads; p.seq_len = a.seq_len; p.q = const_cast<float*>(a.q); p.xb = a.xb;
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
