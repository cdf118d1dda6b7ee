// layer_chain_fused.hpp -- [r5] PARITY MODE's whole stage as ONE launch, for models whose launches are latency-bound (dim <= 1024: the
// stories shapes; rama_set_tuning "chain_fused").  layer_fused.hpp's structure -- per layer the phases Wq|Wk|Wv, attention, Wo, W1|W3,
// W2, then the classifier; every phase waits for the vector the phase before it produces, every float that crosses workgroups is one
// tagged (value, epoch) word, weights are requested BEFORE the wait, a workgroup only waits for workgroups with lower indices, every wait is
// bounded -- with chain.hpp's arithmetic: the reference's rounding order in every op (cpu.rs:23-153), results bit-identical to the oracle.
//   * a matvec workgroup is 8 waves, each wave ONE 16-row group of the chain-order weight copy (lane = (row, k mod 4) chain, W = 1: a ring of
//     D blocks, no relay); the workgroup fetches the input vector once, forms the exact norm where infer.rs has one (:19, :39, :50: the
//     sequential sum of squares -- lane ripples up to 320 terms, seqsum_fast.hpp beyond), and stages it in chain order for its eight waves;
//   * attention is chain.hpp's attention_chain_body<8> with in-launch inputs (q | k | v awaited as tagged words, the cache -- row `pos`
//     included: its producers store it write-through and drain before they tag -- read with sc1 loads);
//   * the run state is left as the separate launches leave it (x, xb, xb2, hb, hb2, q, k, v, att of the last layer, logits, caches); every
//     plain buffer is written by ONE phase of the launch (the L2s of the eight XCDs are not coherent with each other).
#pragma once
#include "chain.hpp"

namespace rama {

struct ChainFusedParams {
    int dim, hidden, n_heads, seq_len, vocab, n_layers, do_cls;
    const float *cq, *ck, *cv, *co, *c13, *c2, *ccls;    // chain-order copies, the stage's first layer (the others at the natural strides)
    const float *g_att, *g_ffn, *g_final;
    const float* emb;                                    // the embedding table if the stage starts from the token (infer.rs:13-14), else it starts from x
    float *x, *xb, *xb2, *q, *k, *v, *hb, *hb2, *att, *logits;
    float *kc, *vc;                                      // the stage's cache slabs [layers, seq, dim]
    const float *fr, *fi;
    const Ctl* ctl;
    tagged_t* hand;                                      // per layer (fused_hand_words): q|k|v [3 dim], xb [dim], xc [dim], hb [hidden], xe [dim]
    const unsigned* epoch;
    unsigned long long* err;
    int nA, nC, nD, nE;                                  // workgroups per matvec phase (8 row groups each)
    int lds_seq;                                         // timesteps the attention's LDS arrays are laid out for
};

constexpr int kCfXD = 4;                                              // activation blocks read ahead, per wave
constexpr int kCfDMax = 32;                                           // the deepest ring: 32 KiB of a row group in a wave's registers
__host__ __device__ constexpr int cf_pad_floats() { return chain_pad_floats(1, kCfDMax, kCfXD); }
// the ring a row group of nblk blocks gets: ALL of it when it fits a wave's registers (48 blocks: a row of 768 floats) -- the weights of a phase
// are then there before its input is, and the chain runs out of registers and LDS -- else a rolling ring of 16.  (Measured: the kernel's five
// phases in one function leave hipcc ~20 free registers beside a ring of 32: 260-330 spilled; profiles/r05_experiments.md section 8.)
__host__ __device__ constexpr int cf_ring(int) { return 16; }      // (rings of 32 / 48 blocks were measured -- stories15M 220 -> 191 us -- but spill 260-970 registers)
// dynamic LDS of a matvec phase: the fetched vector | its chain-order copy + the zeros behind it | the squares (scan_slot layout)
__host__ __device__ constexpr size_t cf_matvec_lds_floats(int kmax, int dim) { return (size_t)kmax + 4 + (size_t)kmax + cf_pad_floats() + (size_t)dim + (dim >> 5) + 8; }
__host__ __device__ constexpr size_t cf_attn_lds_floats(int hs, int lds_seq) {
    return (size_t)((hs + 3) & ~3) + (size_t)lds_seq + (size_t)(lds_seq >> 5) + 4 + (size_t)((lds_seq + 3) & ~3) + (size_t)(2 * kAttTile * hs) + 8;
}

// one wave's row group: the ring of its first D blocks is requested by request() (nothing there depends on this token)
template <int D>
struct CfUnit {
    static_assert(D % 16 == 0 && D <= kCfDMax, "whole 16-block stretches");
    const float* Wg;            // the group's stream: nblk blocks of 256 floats
    int nblk;
    unsigned vo[4];
    f4 wr[D];
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t stretch(int q) const {      // 16 blocks at a time; what lies behind the row reads as 0
        const int n16 = (nblk + 15) >> 4;
        const int left = min(max(nblk - q * 16, 0), 16);
        return make_rsrc_uniform(Wg + (size_t)min(q, n16) * 16 * 256, (unsigned)left * 1024u);
    }
    __device__ __forceinline__ void request(const float* W, int group, int nblk_, bool valid) {
        const int lane = threadIdx.x & 63;
        nblk = valid ? nblk_ : 0;
        Wg = W + (size_t)(valid ? group : 0) * (size_t)nblk_ * 256;
#pragma unroll
        for (int k = 0; k < 4; k++) vo[k] = (unsigned)lane * 16u + (unsigned)k * 4096u;
#pragma unroll
        for (int h = 0; h < D / 16; h++) {
            const __amdgpu_buffer_rsrc_t r0 = stretch(h);
#pragma unroll
            for (int u = 0; u < 16; u++) wr[h * 16 + u] = ld_nt(r0, vo[u >> 2] + (unsigned)(u & 3) * 1024u);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // the chain (cpu.rs:141-148 for this lane's (row, k mod 4)): xs = the activations in chain order, zeros behind them
    __device__ __forceinline__ float run(const float* xs) {
        RAMA_NO_CONTRACT
        const int j = threadIdx.x & 3;
        const f4* xq = reinterpret_cast<const f4*>(xs) + j;
        const int nchunk = (nblk + D - 1) / D;
        float v = 0.0f;
        f4 xr[kCfXD];
#pragma unroll
        for (int u = 0; u < kCfXD; u++) xr[u] = xq[4 * u];
        if (nchunk <= 1) {                                            // (uniform) the whole group is in the ring: registers and LDS only
#pragma unroll
            for (int u = 0; u < D; u++) {
                const f4 xv = xr[u % kCfXD];
                xr[u % kCfXD] = xq[4 * (u + kCfXD)];
                const f4 wv = wr[u];
                v = v + wv.x * xv.x;
                v = v + wv.y * xv.y;
                v = v + wv.z * xv.z;
                v = v + wv.w * xv.w;
            }
        } else {
            for (int c = 0; c < nchunk; c++) {
                __amdgpu_buffer_rsrc_t rn[D / 16];
#pragma unroll
                for (int h = 0; h < D / 16; h++) rn[h] = stretch((c + 1) * (D / 16) + h);
                const f4* xc = xq + 4 * c * D;
#pragma unroll
                for (int u = 0; u < D; u++) {
                    const f4 xv = xr[u % kCfXD];
                    xr[u % kCfXD] = xc[4 * (u + kCfXD)];
                    const f4 wv = wr[u];
                    v = v + wv.x * xv.x;
                    v = v + wv.y * xv.y;
                    v = v + wv.z * xv.z;
                    v = v + wv.w * xv.w;
                    wr[u] = ld_nt(rn[u >> 4], vo[(u & 15) >> 2] + (unsigned)(u & 3) * 1024u);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const float t2 = v + dpp_mov<0xB1>(v);                       // (v0 + v1) + (v2 + v3), cpu.rs:150
        return t2 + dpp_mov<0x4E>(t2);
    }
};

// A matvec workgroup WORKS on kCfMW waves (the other waves of the 8 leave at once): a phase then has four times the workgroups -- a wave owns 16 rows
// where the fast path's owns 4, and with 8 row groups per workgroup stories110M's W1|W3 was 32 workgroups on 32 of 256 compute units.
constexpr int kCfMW = 2, kCfMT = kCfMW * 64;
struct CfShared {
    FastSumShared<kCfMW> fs;
    PredShared<kCfMW> ps;
    SeqSumShared<kCfMW> sh;
    float v;
};

// fused_fetch (layer_fused.hpp) for the kCfMT working threads of a matvec workgroup
__device__ __forceinline__ void cf_fetch(const tagged_t* src, const float* plain, int n, const tagged_t* early, unsigned epoch, float* x_s, int* s_ok, unsigned long long* err) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!src) {
        for (int i = tid * 4; i < n; i += kCfMT * 4) *reinterpret_cast<f4*>(x_s + i) = *reinterpret_cast<const f4*>(plain + i);
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
        for (int i = tid * 2; i < n; i += kCfMT * 2) {
            const u4 p = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rs, i * 8, 0, 16));      // sc1: two words
            ok = ok && p.y == epoch && p.w == epoch;
            x_s[i] = __uint_as_float(p.x); x_s[i + 1] = __uint_as_float(p.z);
        }
        const bool wave_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
        int* slot = s_ok + (tries & 1) * kPWaves;
        if (lane == 0) slot[wave] = wave_ok ? 1 : 0;
        if (tid == 0) slot[kCfMW] = ((tries & 63) == 63 && (tries >= (1 << 20) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) ? 1 : 0;      // give up?
        __syncthreads();
        bool all = true;
#pragma unroll
        for (int w = 0; w < kCfMW; w++) all = all && slot[w] != 0;
        if (all) break;
        if (slot[kCfMW]) {
            if (tid == 0) __hip_atomic_store(err, kFusedErr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

// The workgroup's input: raw[0..K) was fetched (fused_fetch).  NORM: v = 1 / sqrt(sum(x^2) / K + 1e-5), the sum in index order (cpu.rs:110-113),
// and xs <- gain * (v * x) (cpu.rs:114-116); else xs <- x.  xs in chain order (xs[16 s + 4 j + t] = x[16 s + 4 t + j]) with zeros behind.
// Returns v (1 without a norm).  All 512 threads.
// FAST: seqsum_fast.hpp for lists of more than 320 terms (its ~900 instructions and ~100 registers are then part of every norm phase -- beside
// a ring of 32 blocks that is more than the register file holds); without it the ripples / scan rounds of seq_sum_predict
template <bool NORM, bool FAST>
__device__ __forceinline__ float cf_stage(const float* raw, const float* gain, int K, float* xs, float* sq, CfShared& cs) {
    RAMA_NO_CONTRACT
    const int tid = threadIdx.x;
    float v = 1.0f;
    if constexpr (NORM) {
        for (int i = tid; i < K; i += kCfMT) { const float a = raw[i]; sq[scan_slot(i)] = a * a; }
        __syncthreads();
        float ss = 0.0f;
        bool ok = false;
        if constexpr (FAST) { if (K > 320) ok = (seq_sum_fast_prepare<kCfMW>(cs.fs), seq_sum_lds_fast_r<kCfMW, 8>(sq, K, cs.fs, &ss)); }      // (K <= 1024 = 8 x 128 terms)
        if (!ok) {      // (the instantiations of seq_sum_predict for long lists hold 64 terms per thread: only what K <= 1024 on 512 threads can reach is compiled in)
            __syncthreads();
            bool held = true;
            if (K <= kRippleMax) ss = seq_sum_ripples<kCfMW>(sq, K, cs.ps);
            else held = seq_sum_predict_r<kCfMW, 16>(sq, K, cs.ps, &ss);
            if (!held) ss = seq_sum_exact<kCfMW>(sq, K, cs.sh);
        }
        v = 1.0f / sqrtf(ss / (float)K + 1e-5f);
    }
    for (int i = tid; i < K; i += kCfMT) {
        const float a = raw[i];
        const int s_ = i >> 4, t = (i >> 2) & 3, j = i & 3;
        xs[16 * s_ + 4 * j + t] = NORM ? gain[i] * (v * a) : a;
    }
    for (int i = K + tid; i < K + cf_pad_floats(); i += kCfMT) xs[i] = 0.0f;
    __syncthreads();
    return v;
}

// DK / DH: the rings of the products over dim floats (Wq | Wk | Wv, Wo, W1 | W3, the classifier) and over hidden floats (W2): cf_ring
template <int DK, int DH, bool FAST>
__global__ __launch_bounds__(kPThreads) void stage_chain_fused_kernel(ChainFusedParams a) {
    RAMA_NO_CONTRACT
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ __attribute__((aligned(16))) int s_ok[2 * kPWaves];
    __shared__ CfShared cs;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dim = a.dim, hidden = a.hidden, H = a.n_heads, hs = dim / H;
    const int j = lane & 3, rr = lane >> 2;
    const int per_layer = a.nA + H + a.nC + a.nD + a.nE;
    const int layer = blockIdx.x / per_layer;
    const size_t hw = fused_hand_words(dim, hidden);
    const int kmax = hidden > dim ? hidden : dim;
    float* raw = lds;                                   // [kmax] the fetched vector
    float* xs = lds + ((kmax + 3) & ~3);                // [kmax + pad] chain order
    float* sq = xs + kmax + cf_pad_floats();            // [dim + dim / 32 + ...] squares
    const int gd = dim >> 4, gh = hidden >> 4;          // row groups of a dim-row / hidden-row matrix (host: dim % 16 == hidden % 16 == 0)
    CfUnit<DK> u;
    if (layer >= a.n_layers) {                          // ---- infer.rs:49-51: xb = x; x = rmsnorm(xb); logits = Wcls . x
        if (wave >= kCfMW) return;
        const int b = blockIdx.x - a.n_layers * per_layer;
        const int g = b * kCfMW + wave, gv = (a.vocab + 15) >> 4;
        u.request(a.ccls, g, gd, g < gv);
        const unsigned epoch = *a.epoch;
        const tagged_t* hlast = a.hand + (size_t)(a.n_layers ? a.n_layers - 1 : 0) * hw;
        cf_fetch(a.n_layers ? hlast + 5 * dim + hidden : nullptr, a.x, dim, hlast + 5 * dim + hidden - 1, epoch, raw, s_ok, a.err);
        const float v = cf_stage<true, FAST>(raw, a.g_final, dim, xs, sq, cs);
        if (b == 0) {                                   // the run state of :49-50, once
            for (int i = tid; i < dim; i += kCfMT) { const float xv = raw[i]; a.xb[i] = xv; a.x[i] = a.g_final[i] * (v * xv); }
        }
        const float d = u.run(xs);
        const int row = 16 * g + rr;
        if (j == 0 && g < gv && row < a.vocab) a.logits[row] = d;
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
    if (b < a.nA) {                                     // ---- infer.rs:19-33: rmsnorm, Wq | Wk | Wv, RoPE, cache append
        if (wave >= kCfMW) return;
        const int g3 = b * kCfMW + wave, m = g3 / gd, g = g3 - m * gd;
        const bool valid = m < 3;
        u.request((m == 0 ? a.cq : (m == 1 ? a.ck : a.cv)) + layer * dd, g, gd, valid);
        const unsigned epoch = *a.epoch;
        const int pos = a.ctl->pos;
        const int row = 16 * g + rr;
        float rc = 1.0f, rs = 0.0f;
        if (valid && m < 2) {
            const int i = ((row & ~1) % hs) >> 1;                       // infer.rs:15-16: table row pos, pair i of the head
            rc = a.fr[(size_t)pos * (hs >> 1) + i]; rs = a.fi[(size_t)pos * (hs >> 1) + i];
        }
        cf_fetch(t_in, x_in, dim, layer ? hl - dim - 1 : nullptr, epoch, raw, s_ok, a.err);      // early: the last word of hb of the layer before
        cf_stage<true, FAST>(raw, a.g_att + (size_t)layer * dim, dim, xs, sq, cs);
        const float d = u.run(xs);
        const float other = __shfl_xor(d, 4);                           // the pair's other row (neighbouring quad)
        const float p0 = (rr & 1) ? other : d, p1 = (rr & 1) ? d : other;
        float out = d;
        if (m < 2) out = (rr & 1) ? p0 * rs + p1 * rc : p0 * rc - p1 * rs;      // cpu.rs:87-96
        const bool mine = j == 0 && valid && row < dim;
        // the cache row first, write-through and drained: whoever sees this row's tags reads the cache with sc1 loads
        if (mine && m == 1) st_sc1(kc + (size_t)pos * dim + row, out);
        if (mine && m == 2) st_sc1(vc + (size_t)pos * dim + row, out);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (mine) {
            put_tagged(t_qkv + (size_t)m * dim + row, out, epoch);
            if (last) { float* o = m == 0 ? a.q : (m == 1 ? a.k : a.v); o[row] = out; }
        }
        return;
    }
    b -= a.nA;
    if (b < H) {                                        // ---- infer.rs:34: one head
        RefAttnParams p{};
        p.q = nullptr; p.kc = kc; p.vc = vc; p.att = last ? a.att : nullptr; p.xb = nullptr; p.ctl = a.ctl; p.pos_val = 0;
        p.dim = dim; p.head_size = hs; p.seq_len = a.seq_len;
        AttnInl inl{t_qkv, t_in ? t_in + dim - 1 : nullptr, t_xb, a.err, *a.epoch, s_ok};
        attention_chain_body<kPWaves, false, true>(p, b, 0, a.lds_seq, nullptr, nullptr, inl);
        return;
    }
    b -= H;
    if (b < a.nC) {                                     // ---- infer.rs:35-37: xb2 = Wo . xb; xc = x + xb2
        if (wave >= kCfMW) return;
        const int g = b * kCfMW + wave;
        u.request(a.co + layer * dd, g, gd, g < gd);
        const unsigned epoch = *a.epoch;
        cf_fetch(t_xb, nullptr, dim, t_qkv + 3 * dim - 1, epoch, raw, s_ok, a.err);
        const int row = 16 * g + rr;
        const bool mine = j == 0 && g < gd && row < dim;
        float resid = 0.0f;                             // complete since before this layer's first phase
        if (mine) resid = t_in ? get_tagged(t_in + row) : x_in[row];
        cf_stage<false, FAST>(raw, nullptr, dim, xs, sq, cs);
        const float d = u.run(xs);
        if (mine) {
            put_tagged(t_xc + row, resid + d, epoch);
            if (last) a.xb2[row] = d;
        }
        return;
    }
    b -= a.nC;
    if (b < a.nD) {                                     // ---- infer.rs:39-45: rmsnorm, W1 | W3 (interleaved rows), SiLU * gate
        if (wave >= kCfMW) return;
        const int g = b * kCfMW + wave, g2 = 2 * gh;
        u.request(a.c13 + layer * 2 * hd, g, gd, g < g2);
        const unsigned epoch = *a.epoch;
        cf_fetch(t_xc, nullptr, dim, t_xb + dim - 1, epoch, raw, s_ok, a.err);
        cf_stage<true, FAST>(raw, a.g_ffn + (size_t)layer * dim, dim, xs, sq, cs);
        const float d = u.run(xs);
        const float h3 = __shfl_xor(d, 4);              // even row = W1 row i, odd row = W3 row i
        const int row = 16 * g + rr;
        if (j == 0 && !(rr & 1) && g < g2 && row < 2 * hidden) {
            const float sl = d * (1.0f / (1.0f + expf_glibc(-d)));     // cpu.rs:56
            const float o = sl * h3;                                   // cpu.rs:59-64
            put_tagged(t_hb + (row >> 1), o, epoch);
            if (last) { a.hb[row >> 1] = o; a.hb2[row >> 1] = h3; }
        }
        return;
    }
    b -= a.nD;
    {                                                   // ---- infer.rs:46-47: xb = W2 . hb; xe = xc + xb
        if (wave >= kCfMW) return;
        const int g = b * kCfMW + wave;
        CfUnit<DH> u2;
        u2.request(a.c2 + layer * hd, g, gh, g < gd);
        const unsigned epoch = *a.epoch;
        cf_fetch(t_hb, nullptr, hidden, t_xc + dim - 1, epoch, raw, s_ok, a.err);
        const int row = 16 * g + rr;
        const bool mine = j == 0 && g < gd && row < dim;
        float resid = 0.0f;                             // complete since before the phase before this one
        if (mine) resid = get_tagged(t_xc + row);
        cf_stage<false, FAST>(raw, nullptr, hidden, xs, sq, cs);
        const float d = u2.run(xs);
        if (mine) {
            const float o = resid + d;
            put_tagged(t_xe + row, o, epoch);
            // (every plain buffer has ONE writer in the launch: two workgroups on different XCDs that store to one line leave two dirty copies in two
            // L2s, written back in any order.  With a classifier in the stage x and xb are its workgroup 0's: infer.rs:49-50)
            if (last && !a.do_cls) { a.xb[row] = d; a.x[row] = o; }
        }
    }
}

}  // namespace rama
