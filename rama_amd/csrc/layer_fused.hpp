// layer_fused.hpp -- a whole stage (its decoder layers, infer.rs:19-47, and the classifier, :49-51) as ONE launch, for
// models whose launches are latency-bound (the stories shapes; rama_set_tuning "fused").  Workgroups of 8 waves, in this
// order per layer, each phase waiting for the vector the phase before it produces:
//   nA  workgroups   q | k | v = rows of Wq | Wk | Wv times rmsnorm(x), RoPE, cache append      reads xe of layer l-1
//   H   workgroups   xb = attention, one head each                                              reads q | k | v
//   nC  workgroups   xc = x + Wo . xb                                                           reads xb
//   nD  workgroups   hb = silu(W1 . xn) * (W3 . xn), xn = rmsnorm(xc)                           reads xc
//   nE  workgroups   xe = xc + W2 . hb                                                          reads hb
// and after the last layer the classifier's workgroups (rmsnorm folded in), which read the last xe.  The first layer reads
// the token's embedding row itself (infer.rs:13-14) when the stage starts there.
// In a matvec workgroup every wave is its own unit -- 4 weight rows, the whole width, as in gemv_rows_solo -- and
// requests its weights BEFORE the workgroup waits: weights do not depend on activations, so while a layer computes, the
// rows of the layers behind it are already on their way.
// Hand-off: THE DATA CARRIES ITS OWN TAG.  Every float that crosses workgroups is one 8-byte (value, epoch) word written
// with a single write-through store; `epoch` is a device counter incremented after every launch of this kernel, and every (layer,
// phase) has its own vector, so a word holds either this token's value or a stale tag.  A consumer polls one word from one
// lane, then the workgroup copies the vector to LDS checking every tag, and repeats the copy until all match.  No counter,
// no store drain, no arrival atomic: tools/handoff_bench.hip measures 1.3 us per hand-off against 2.4 us for the
// counter protocol of attn_wo.hpp (profiles/r03_fused_counter_handoff.txt has that version of this kernel: slower than
// separate launches).
// No deadlock whatever is resident: a workgroup only ever waits for workgroups with LOWER indices, which are dispatched
// before it.  Every spin is bounded all the same, and gives up at once when another workgroup already has (error word).
#pragma once
#include "attn_wo.hpp"

namespace rama {

typedef unsigned long long tagged_t;          // low word: the float's bits, high word: the epoch it was written in
typedef __attribute__((ext_vector_type(4))) unsigned u4;

struct FusedParams {
    int dim, hidden, n_heads, seq_len, vocab, n_layers, do_cls;
    const float *wq, *wk, *wv, *wo, *w1, *w3, *w2;      // the stage's first layer; the others follow at the natural strides
    const float *g_att, *g_ffn, *g_final, *wcls;
    const float* emb;                                    // the embedding table if the stage starts from the token (infer.rs:13-14), else it starts from x
    float *x, *q, *k, *v, *xb, *hb, *logits;             // run-state buffers; all are left as the separate launches leave them
    float *kc, *vc;                                      // the stage's cache slabs [layers, seq, dim]
    const float *fr, *fi;
    const Ctl* ctl;
    tagged_t* hand;                                      // per layer: q|k|v [3 dim], xb [dim], xc [dim], hb [hidden], xe [dim]
    const unsigned* epoch;
    unsigned long long* err;
    int nA, nC, nD, nE;                                  // workgroups per matvec phase
};
constexpr unsigned long long kFusedErr = 0x3000ull;
constexpr int kFusedMaxLayers = 128, kFusedMaxDim = 1024, kFusedMaxHidden = 8192;
constexpr size_t kFusedSoloLds = 88 * 1024;       // more than half a CU's LDS: one workgroup per CU
__host__ __device__ constexpr size_t fused_hand_words(int dim, int hidden) { return (size_t)6 * dim + hidden; }

#ifdef RAMA_FUSED_STAMPS      // tools/fused_stamps.hip: where a layer's time goes (100 MHz clock), first workgroup of every phase
__device__ unsigned long long g_fused_stamps[8][6][8];
#define FUSED_STAMP(first, layer, phase, i) do { if ((first) && threadIdx.x == 0 && (layer) < 8) g_fused_stamps[layer][phase][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define FUSED_STAMP_MAX(layer, phase) do { if (threadIdx.x == 0 && (layer) < 8) atomicMax(&g_fused_stamps[layer][phase][3], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#else
#define FUSED_STAMP(first, layer, phase, i) do { } while (0)
#define FUSED_STAMP_MAX(layer, phase) do { } while (0)
#endif

// Every launch of the stage kernel is FOLLOWED by one increment of the epoch, so the epoch a launch reads was never used
// before: by the sampler's last thread in the chained decode loop (kernels.hpp finish_step), by this kernel otherwise.
__global__ void fused_epoch_kernel(unsigned* epoch) { *epoch = *epoch + 1u; }

__device__ __forceinline__ void put_tagged(tagged_t* p, float v, unsigned epoch) {
    __hip_atomic_store(p, ((tagged_t)epoch << 32) | (tagged_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float get_tagged(const tagged_t* p) {      // a word some phase before the awaited one wrote
    return __uint_as_float((unsigned)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// one lane: wait (sleeping) until the word at p carries `epoch` -- the cheap wait of workgroups far ahead of the data
__device__ __forceinline__ void fused_watch(const tagged_t* p, unsigned epoch, unsigned long long* err) {
    long spins = 0;
    while ((unsigned)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) != epoch) {
        __builtin_amdgcn_s_sleep(2);
        ++spins;
        if ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (spins > (1L << 22)) { __hip_atomic_store(err, kFusedErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
}

// `ok` of every thread of the workgroup, with one barrier: a wave vote, eight LDS words, two slots taken in turn (a wave can
// be one round ahead of the slowest reader, never two).  __syncthreads_and costs three barriers and a DPP reduction.
__device__ __forceinline__ bool fused_all(bool ok, int round, int* s_ok) {
    const bool wave_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
    int* slot = s_ok + (round & 1) * kPWaves;
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = wave_ok ? 1 : 0;
    __syncthreads();
    const int4 a = *reinterpret_cast<const int4*>(slot), b = *reinterpret_cast<const int4*>(slot + 4);
    return (a.x & a.y & a.z & a.w & b.x & b.y & b.z & b.w) != 0;
}

// The whole workgroup: x_s[0..n) = the n floats of `src` (n even, src 16-byte aligned), once every one of them carries
// `epoch`: the vector is copied to LDS, every tag checked, again and again until all match.  `early`: a word of the vector
// the producers of `src` themselves wait for -- one lane sleeps on it first, so that only the workgroups whose input is
// being produced right now poll whole vectors.  src == nullptr: the vector is `plain`, an ordinary float array of an
// earlier launch.
__device__ __forceinline__ void fused_fetch(const tagged_t* src, const float* plain, int n, const tagged_t* early, unsigned epoch, float* x_s,
                                            int* s_ok, unsigned long long* err) {
    const int tid = threadIdx.x;
    if (!src) {
        for (int i = tid * 4; i < n; i += kPThreads * 4) *reinterpret_cast<f4*>(x_s + i) = *reinterpret_cast<const f4*>(plain + i);
        __syncthreads();
        return;
    }
    if (early) {
        if (tid == 0) fused_watch(early, epoch, err);
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, (unsigned)n * 8u);
    for (int tries = 0;; tries++) {
        bool ok = true;
        for (int i = tid * 2; i < n; i += kPThreads * 2) {
            const u4 p = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rs, i * 8, 0, 16));      // sc1: two words
            ok = ok && p.y == epoch && p.w == epoch;
            x_s[i] = __uint_as_float(p.x); x_s[i + 1] = __uint_as_float(p.z);
        }
        if (fused_all(ok, tries, s_ok)) break;
        if ((tries & 63) == 63 &&
            __syncthreads_or(tries >= (1 << 20) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {      // one verdict for the workgroup
            if (tid == 0) __hip_atomic_store(err, kFusedErr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

// v[l] (+ | max) v[l ^ 16] and v[l] (+ | max) v[l ^ 32] as VALU lane swaps (gfx950's v_permlane16_swap / v_permlane32_swap): a
// ds_bpermute shuffle goes through the LDS pipeline and costs its latency every time
template <int MASK>
__device__ __forceinline__ void xor_pair(float v, float& a, float& b) {
    static_assert(MASK == 16 || MASK == 32, "rows or halves");
    const unsigned u = __float_as_uint(v);
    if (MASK == 16) { const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false); a = __uint_as_float(r[0]); b = __uint_as_float(r[1]); }
    else { const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false); a = __uint_as_float(r[0]); b = __uint_as_float(r[1]); }
}
template <int MASK> __device__ __forceinline__ float xor_sum(float v) { float a, b; xor_pair<MASK>(v, a, b); return a + b; }
template <int MASK> __device__ __forceinline__ float xor_max(float v) { float a, b; xor_pair<MASK>(v, a, b); return fmaxf(a, b); }
// over the 64 / G groups of G lanes each
template <int G> __device__ __forceinline__ float groups_sum(float v) {
    if (G <= 16) v = xor_sum<16>(v);
    if (G <= 32) v = xor_sum<32>(v);
    return v;
}
template <int G> __device__ __forceinline__ float groups_max(float v) {
    if (G <= 16) v = xor_max<16>(v);
    if (G <= 32) v = xor_max<32>(v);
    return v;
}

// one wave's weights: R rows, the first CH 256-float chunks of each -- rows r0..r0+R-1 of Wa, or (PAIR, R = 4) rows r0, r0+1 of
// Wa and of Wb interleaved (0/2 = Wa, 1/3 = Wb)
template <int R, int CH, bool NORM, bool PAIR>
struct FusedUnit {
    __amdgpu_buffer_rsrc_t ra, rb, rg;
    unsigned rowoff[R], kbytes;
    int nch, lane;
    f4 w[R][CH], gv[CH];
    __device__ __forceinline__ unsigned kb_of(int c) const { const unsigned o = (unsigned)(c * 1024 + lane * 16); return (c < nch && o < kbytes) ? o : kOOB; }
    __device__ __forceinline__ f4 wload(int s, unsigned kb) const {
        const unsigned o = (kb == kOOB || rowoff[s] == kOOB) ? kOOB : rowoff[s] + kb;
        return ld_nt((PAIR && (s & 1)) ? rb : ra, o);
    }
    // request the weights (and the rmsnorm gain): nothing here depends on this token
    __device__ __forceinline__ void request(const float* Wa, const float* Wb, const float* gain, int rows, int K, int r0, bool valid) {
        lane = threadIdx.x & 63;
        nch = (K + 255) >> 8;
        kbytes = (unsigned)K * 4u;
        ra = make_rsrc(Wa, (unsigned)rows * kbytes);
        rb = make_rsrc(PAIR ? Wb : Wa, (unsigned)rows * kbytes);
        rg = make_rsrc(NORM ? gain : Wa, kbytes);
#pragma unroll
        for (int s = 0; s < R; s++) {
            const int r = PAIR ? r0 + (s >> 1) : r0 + s;
            rowoff[s] = (valid && r < rows) ? (unsigned)r * kbytes : kOOB;
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
#pragma unroll
            for (int s = 0; s < R; s++) w[s][j] = wload(s, kb_of(j));
            if (NORM) gv[j] = ld_c(rg, kb_of(j));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // the R dot products with x_s (LDS, K floats); NORM: x is scaled by the gain here and `scale` = 1 / sqrt(mean(x^2) + eps)
    // comes back (cpu.rs:99-117)
    __device__ __forceinline__ void dots(const float* x_s, int K, float (&acc)[R], float& scale) {
        float ss = 0.0f;
#pragma unroll
        for (int s = 0; s < R; s++) acc[s] = 0.0f;
        auto xload = [&](int c) {      // no branch: a lane past the end reads float 0.. and drops it
            const f4 z = {0.f, 0.f, 0.f, 0.f};
            const bool in = kb_of(c) != kOOB;
            const f4 v = *reinterpret_cast<const f4*>(x_s + (in ? c * 256 + lane * 4 : 0));
            return in ? v : z;
        };
#pragma unroll
        for (int j = 0; j < CH; j++) {
            f4 xv = xload(j);
            if (NORM) { ss = dot4(xv, xv, ss); xv = xv * gv[j]; }
#pragma unroll
            for (int s = 0; s < R; s++) acc[s] = dot4(w[s][j], xv, acc[s]);
        }
        for (int c0 = CH; c0 < nch; c0 += CH) {        // rows wider than the requested part
#pragma unroll
            for (int j = 0; j < CH; j++) {
#pragma unroll
                for (int s = 0; s < R; s++) w[s][j] = wload(s, kb_of(c0 + j));
                if (NORM) gv[j] = ld_c(rg, kb_of(c0 + j));
            }
#pragma unroll
            for (int j = 0; j < CH; j++) {
                f4 xv = xload(c0 + j);
                if (NORM) { ss = dot4(xv, xv, ss); xv = xv * gv[j]; }
#pragma unroll
                for (int s = 0; s < R; s++) acc[s] = dot4(w[s][j], xv, acc[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < R; s++) acc[s] = wave_sum(acc[s]);
        scale = 1.0f;
        if (NORM) scale = rms_scale(wave_sum(ss), K);
    }
};

// attention for head h (infer.rs:34): the cache rows before `pos` are an earlier launch's and are requested before the wait;
// row `pos` and q are this launch's -- cur = q | k | v of this head, fetched into LDS here.  Writes xb as tagged words (and
// plainly, `xb_plain`, for the run state).
template <int G>
__device__ __forceinline__ void fused_attention(int dim, int n_heads, int seq_len, const float* kc, const float* vc, int h, int pos,
                                                const tagged_t* t_qkv, const tagged_t* early, float* cur, float* lds, int* s_ok, tagged_t* xb_t, float* xb_plain,
                                                unsigned epoch, unsigned long long* err, int st_layer = 0) {
    float* s_max = lds;
    float* s_sum = lds + kPWaves;
    float* s_acc = lds + 2 * kPWaves;
    constexpr int U = kPAttnU;
    constexpr int TPW = 64 / G;
    constexpr int TILE = kPWaves * TPW * U;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hs = dim / n_heads;
    const int li = lane % G, tg = lane / G;
    const bool lane_ok = li * 4 < hs;
    const unsigned cache_bytes = (unsigned)seq_len * (unsigned)dim * 4u;
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(kc, cache_bytes);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(vc, cache_bytes);
    const unsigned col = (unsigned)(h * hs + li * 4) * 4u;
    const unsigned rowb = (unsigned)dim * 4u;
    auto t_of = [&](int base, int u) { return base + (u * kPWaves + wave) * TPW + tg; };
    auto off_of = [&](int t) { return (lane_ok && t < pos) ? (unsigned)t * rowb + col : kOOB; };
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    const float inv_div = 1.0f / sqrtf((float)hs);       // one rounding more than cpu.rs:75 (a division): within the fast path's bar
    f4 kt[U], vt[U];
#pragma unroll
    for (int u = 0; u < U; u++) kt[u] = ld_c(rk, off_of(t_of(0, u)));      // rows of earlier launches: on their way before the wait
#pragma unroll
    for (int u = 0; u < U; u++) vt[u] = ld_c(rv, off_of(t_of(0, u)));
    __builtin_amdgcn_sched_barrier(0);
    {   // cur = this head's slices of the q | k | v vector, 3 hs words, once all carry the epoch
        if (early) {
            if (tid == 0) fused_watch(early, epoch, err);
            __syncthreads();
        }
        for (int tries = 0;; tries++) {
            bool ok = true;
            for (int i = tid; i < 3 * hs; i += kPThreads) {
                const int m = i / hs, j = i - m * hs;
                const tagged_t p = __hip_atomic_load(t_qkv + m * dim + h * hs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (unsigned)(p >> 32) == epoch;
                cur[m * hs + j] = __uint_as_float((unsigned)p);
            }
            if (fused_all(ok, tries, s_ok)) break;
            if ((tries & 63) == 63 && __syncthreads_or(tries >= (1 << 20) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                if (tid == 0) __hip_atomic_store(err, kFusedErr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    FUSED_STAMP(h == 0, st_layer, 1, 1);
    const f4 q4 = lane_ok ? *reinterpret_cast<const f4*>(cur + li * 4) : zero;
    const f4 k4 = lane_ok ? *reinterpret_cast<const f4*>(cur + hs + li * 4) : zero;
    const f4 v4 = lane_ok ? *reinterpret_cast<const f4*>(cur + 2 * hs + li * 4) : zero;
    // Every wave runs the softmax of ITS timesteps on its own -- running maximum m, sum l of exp(score - m), acc = the
    // values weighted by them (cpu.rs:64-97 rescaled) -- and the eight partial results meet once, behind one barrier.
    float m = -INFINITY, l = 0.0f;
    f4 acc = zero;
    for (int base = 0; base <= pos; base += TILE) {
        if (base > 0) {
#pragma unroll
            for (int u = 0; u < U; u++) kt[u] = ld_c(rk, off_of(t_of(base, u)));
#pragma unroll
            for (int u = 0; u < U; u++) vt[u] = ld_c(rv, off_of(t_of(base, u)));
            __builtin_amdgcn_sched_barrier(0);
        }
        float d[U], tm = -INFINITY;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t_of(base, u);
            float dd = dot4(q4, t == pos ? k4 : kt[u], 0.0f);
            dd = row16_sum(dd);
            if (G >= 32) dd = xor_sum<16>(dd);
            if (G == 64) dd = xor_sum<32>(dd);
            d[u] = dd * inv_div;
            if (t <= pos) tm = fmaxf(tm, d[u]);
        }
        tm = groups_max<G>(tm);
        const float m_new = fmaxf(m, tm);                       // the same in every lane of the wave
        if (m_new > -INFINITY) {
            const float sc = m > -INFINITY ? __expf(m - m_new) : 0.0f;
            l *= sc; acc.x *= sc; acc.y *= sc; acc.z *= sc; acc.w *= sc;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int t = t_of(base, u);
                const float e = t <= pos ? __expf(d[u] - m_new) : 0.0f;
                const f4 vv = t == pos ? v4 : vt[u];
                l += e;
                acc.x = fmaf(e, vv.x, acc.x); acc.y = fmaf(e, vv.y, acc.y);
                acc.z = fmaf(e, vv.z, acc.z); acc.w = fmaf(e, vv.w, acc.w);
            }
            m = m_new;
        }
    }
    FUSED_STAMP(h == 0, st_layer, 1, 5);
    l = groups_sum<G>(l);                      // the wave's 64 / G timestep groups
    acc.x = groups_sum<G>(acc.x); acc.y = groups_sum<G>(acc.y); acc.z = groups_sum<G>(acc.z); acc.w = groups_sum<G>(acc.w);
    if (lane == 0) { s_max[wave] = m; s_sum[wave] = l; }
    if (lane < G) *reinterpret_cast<f4*>(s_acc + (wave * G + lane) * 4) = acc;
    __syncthreads();
    FUSED_STAMP(h == 0, st_layer, 1, 6);
    if (tid < G && tid * 4 < hs) {
        float M = s_max[0];
#pragma unroll
        for (int w = 1; w < kPWaves; w++) M = fmaxf(M, s_max[w]);
        float L = 0.0f;
        f4 o4 = zero;
#pragma unroll
        for (int w = 0; w < kPWaves; w++) {
            const float f = s_max[w] > -INFINITY ? __expf(s_max[w] - M) : 0.0f;
            const f4 a = *reinterpret_cast<f4*>(s_acc + (w * G + tid) * 4);
            L = fmaf(s_sum[w], f, L);
            o4.x = fmaf(a.x, f, o4.x); o4.y = fmaf(a.y, f, o4.y); o4.z = fmaf(a.z, f, o4.z); o4.w = fmaf(a.w, f, o4.w);
        }
        const float rl = 1.0f / L;
        o4.x *= rl; o4.y *= rl; o4.z *= rl; o4.w *= rl;
        const int o = h * hs + tid * 4;
        put_tagged(xb_t + o, o4.x, epoch); put_tagged(xb_t + o + 1, o4.y, epoch);
        put_tagged(xb_t + o + 2, o4.z, epoch); put_tagged(xb_t + o + 3, o4.w, epoch);
        if (xb_plain) *reinterpret_cast<f4*>(xb_plain + o) = o4;
    }
}

__host__ __device__ constexpr int fused_lds_floats(int G, int seq_len, int dim, int hidden) {
    const int attn = p_attn_lds_floats(G, seq_len) + 3 * 256, vec = hidden > dim ? hidden : dim;
    return attn > vec ? attn : vec;
}

// G: lanes per cache row in the attention phase (head_size / 4 rounded up to 16 | 32 | 64); CD: chunks requested ahead of a row
// as wide as dim; a W2 unit is RE rows, CHH chunks of each requested ahead (4 x 4 up to hidden_dim 1024, 2 x 8 beyond)
template <int G, int CD, int RE, int CHH>
__global__ __launch_bounds__(kPThreads, 4) void stage_fused_kernel(FusedParams a) {
    // (16-byte aligned whatever static LDS sits in front of it: the vectors in it are read 16 bytes at a time and a misaligned
    // ds_read_b128 is split -- round 4 found out when 68 more bytes of static LDS cost 18 % of a stories110M token)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ __attribute__((aligned(16))) int s_ok[2 * kPWaves];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dim = a.dim, hidden = a.hidden, H = a.n_heads, hs = dim / H;
    const int per_layer = a.nA + H + a.nC + a.nD + a.nE;
    const int layer = blockIdx.x / per_layer;
    const size_t hw = fused_hand_words(dim, hidden);
    float acc[4], scale;
    if (layer >= a.n_layers) {                         // ---- infer.rs:49-51: logits = Wcls . rmsnorm(x)
        constexpr int RC = CD == 2 ? 8 : 4;            // rows per unit: as many as fit the registers
        const int r0 = ((blockIdx.x - a.n_layers * per_layer) * kPWaves + wave) * RC;
        FUSED_STAMP(blockIdx.x == a.n_layers * per_layer, 0, 5, 0);
        FUSED_STAMP(blockIdx.x == gridDim.x - 1, 0, 5, 4);
        FusedUnit<RC, CD, true, false> u;
        u.request(a.wcls, nullptr, a.g_final, a.vocab, dim, r0, r0 < a.vocab);
        const unsigned epoch = *a.epoch;
        const tagged_t* hlast = a.hand + (size_t)(a.n_layers ? a.n_layers - 1 : 0) * hw;
        fused_fetch(a.n_layers ? hlast + 5 * dim + hidden : nullptr, a.x, dim, hlast + 5 * dim + hidden - 1, epoch, lds, s_ok, a.err);
        FUSED_STAMP(blockIdx.x == a.n_layers * per_layer, 0, 5, 1);
        float accC[RC];
        u.dots(lds, dim, accC, scale);
        if (lane < RC && r0 + lane < a.vocab) a.logits[r0 + lane] = pick<RC>(accC, lane) * scale;
        FUSED_STAMP(blockIdx.x == a.n_layers * per_layer, 0, 5, 2);
        FUSED_STAMP_MAX(0, 5);
        return;
    }
    int b = blockIdx.x - layer * per_layer;
    const bool last = layer == a.n_layers - 1;
    const size_t dd = (size_t)dim * dim, hd = (size_t)hidden * dim;
    tagged_t* hl = a.hand + (size_t)layer * hw;
    tagged_t *t_qkv = hl, *t_xb = hl + 3 * dim, *t_xc = hl + 4 * dim, *t_hb = hl + 5 * dim, *t_xe = hl + 5 * dim + hidden;
    const tagged_t* t_in = layer ? hl - dim : nullptr;                  // xe of the layer before
    const float* x_in = a.emb ? a.emb + (size_t)a.ctl->token * dim : a.x;   // ... or what the stage starts from
    float* kc = a.kc + (size_t)layer * a.seq_len * dim;
    float* vc = a.vc + (size_t)layer * a.seq_len * dim;
    if (b < a.nA) {                                    // ---- infer.rs:19-33: rmsnorm, Wq | Wk | Wv, RoPE, cache append
        const int upm = (dim + 3) >> 2, un = b * kPWaves + wave, m = un / upm, r0 = (un - m * upm) * 4;
        FUSED_STAMP(b == 0, layer, 0, 0);
        FusedUnit<4, CD, true, false> u;
        u.request((m == 0 ? a.wq : (m == 1 ? a.wk : a.wv)) + layer * dd, nullptr, a.g_att + (size_t)layer * dim, dim, dim, r0, m < 3);
        const unsigned epoch = *a.epoch;
        const int pos = a.ctl->pos;
        const bool mine = lane < 2 && m < 3 && r0 + 2 * lane < dim;
        float rc = 1.0f, rs = 0.0f;                    // cpu.rs:45-62: the rotation of this lane's pair
        if (mine && m < 2) {
            const int i = ((r0 + 2 * lane) % hs) >> 1;
            rc = a.fr[(size_t)pos * (hs >> 1) + i]; rs = a.fi[(size_t)pos * (hs >> 1) + i];
        }
        fused_fetch(t_in, x_in, dim, layer ? hl - dim - 1 : nullptr, epoch, lds, s_ok, a.err);      // early: the last word of hb of the layer before
        FUSED_STAMP(b == 0, layer, 0, 1);
        u.dots(lds, dim, acc, scale);
        if (mine) {
            const int r = r0 + 2 * lane;
            float va = pick<4>(acc, 2 * lane) * scale, vb = pick<4>(acc, 2 * lane + 1) * scale;
            if (m < 2) {
                const float ra = va * rc - vb * rs, rb = va * rs + vb * rc;
                va = ra; vb = rb;
            }
            put_tagged(t_qkv + m * dim + r, va, epoch); put_tagged(t_qkv + m * dim + r + 1, vb, epoch);
            if (m == 1) { kc[(size_t)pos * dim + r] = va; kc[(size_t)pos * dim + r + 1] = vb; }
            else if (m == 2) { vc[(size_t)pos * dim + r] = va; vc[(size_t)pos * dim + r + 1] = vb; }
            if (last) { float* o = m == 0 ? a.q : (m == 1 ? a.k : a.v); o[r] = va; o[r + 1] = vb; }
        }
        FUSED_STAMP(b == 0, layer, 0, 2);
        FUSED_STAMP_MAX(layer, 0);
        return;
    }
    b -= a.nA;
    if (b < H) {                                       // ---- infer.rs:34: one head
        FUSED_STAMP(b == 0, layer, 1, 0);
        const unsigned epoch = *a.epoch;
        float* cur = lds + p_attn_lds_floats(G, a.seq_len);           // q | k | v of this head, hs floats each (hs <= 256)
        fused_attention<G>(dim, H, a.seq_len, kc, vc, b, a.ctl->pos, t_qkv, t_in ? t_in + dim - 1 : nullptr, cur, lds, s_ok, t_xb, last ? a.xb : nullptr, epoch,
                           a.err, layer);
        FUSED_STAMP(b == 0, layer, 1, 2);
        FUSED_STAMP_MAX(layer, 1);
        return;
    }
    b -= H;
    if (b < a.nC) {                                    // ---- infer.rs:35-37: xc = x + Wo . xb
        const int r0 = (b * kPWaves + wave) * 4;
        FUSED_STAMP(b == 0, layer, 2, 0);
        FusedUnit<4, CD, false, false> u;
        u.request(a.wo + layer * dd, nullptr, nullptr, dim, dim, r0, r0 < dim);
        const unsigned epoch = *a.epoch;
        fused_fetch(t_xb, nullptr, dim, t_qkv + 3 * dim - 1, epoch, lds, s_ok, a.err);
        FUSED_STAMP(b == 0, layer, 2, 1);
        float resid = 0.0f;                            // complete since before this layer's first phase
        if (lane < 4 && r0 + lane < dim) resid = t_in ? get_tagged(t_in + r0 + lane) : x_in[r0 + lane];
        __builtin_amdgcn_sched_barrier(0);
        u.dots(lds, dim, acc, scale);
        if (lane < 4 && r0 + lane < dim) put_tagged(t_xc + r0 + lane, resid + pick<4>(acc, lane), epoch);
        FUSED_STAMP(b == 0, layer, 2, 2);
        FUSED_STAMP_MAX(layer, 2);
        return;
    }
    b -= a.nC;
    if (b < a.nD) {                                    // ---- infer.rs:39-45: rmsnorm, W1 | W3, SiLU * gate
        const int r0 = (b * kPWaves + wave) * 2;
        FUSED_STAMP(b == 0, layer, 3, 0);
        FusedUnit<4, CD, true, true> u;
        u.request(a.w1 + layer * hd, a.w3 + layer * hd, a.g_ffn + (size_t)layer * dim, hidden, dim, r0, r0 < hidden);
        const unsigned epoch = *a.epoch;
        fused_fetch(t_xc, nullptr, dim, t_xb + dim - 1, epoch, lds, s_ok, a.err);
        FUSED_STAMP(b == 0, layer, 3, 1);
        u.dots(lds, dim, acc, scale);
        if (lane < 2 && r0 + lane < hidden) {
            const float va = pick<4>(acc, 2 * lane) * scale, vb = pick<4>(acc, 2 * lane + 1) * scale;
            const float g = va * (1.0f / (1.0f + expf(-va))) * vb;
            put_tagged(t_hb + r0 + lane, g, epoch);
            if (last) a.hb[r0 + lane] = g;
        }
        FUSED_STAMP(b == 0, layer, 3, 2);
        FUSED_STAMP_MAX(layer, 3);
        return;
    }
    b -= a.nD;
    {                                                  // ---- infer.rs:46-47: xe = xc + W2 . hb
        const int r0 = (b * kPWaves + wave) * RE;
        FUSED_STAMP(b == 0, layer, 4, 0);
        FusedUnit<RE, CHH, false, false> u;
        u.request(a.w2 + layer * hd, nullptr, nullptr, dim, hidden, r0, r0 < dim);
        const unsigned epoch = *a.epoch;
        fused_fetch(t_hb, nullptr, hidden, t_xc + dim - 1, epoch, lds, s_ok, a.err);
        FUSED_STAMP(b == 0, layer, 4, 1);
        float resid = 0.0f;                            // complete since before the phase before this one
        if (lane < RE && r0 + lane < dim) resid = get_tagged(t_xc + r0 + lane);
        __builtin_amdgcn_sched_barrier(0);
        float accE[RE];
        u.dots(lds, hidden, accE, scale);
        FUSED_STAMP(b == 0, layer, 4, 5);
        if (lane < RE && r0 + lane < dim) {
            const float o = resid + pick<RE>(accE, lane);
            put_tagged(t_xe + r0 + lane, o, epoch);
            if (last) a.x[r0 + lane] = o;
        }
        FUSED_STAMP(b == 0, layer, 4, 2);
        FUSED_STAMP_MAX(layer, 4);
    }
}

}  // namespace rama
