// prefill.hpp -- batched-prompt prefill: PB prompt positions go
// through a layer together, so every weight row is streamed ONCE per PB tokens instead of once
// per token.  The reference has no such path (generate() calls forward() per forced prompt token,
// transformer/mod.rs:187-194); the contract is "same KV-cache rows and same final logits as PB
// sequential forward() calls", checked against the oracle.
//
// Why not MFMA: on gfx950 the f32-input MFMA (v_mfma_f32_32x32x2_f32) runs at exactly the f32 VALU
// FMA rate (MI355X_MICROARCH.md, Matrix cores) and its A-operand layout (lane = row) would turn the
// coalesced 1-KiB weight-row reads into 4-byte strided gathers.  A multi-right-hand-side matvec
// keeps the decode path's streaming core (16-byte coalesced nt loads, chunk c -> wave c mod 8) and
// spends 4*PB FMAs per loaded float4: at PB = 8 that is ~18 % of the VALU issue rate per CU while
// HBM stays the bound, i.e. the prompt runs ~PB x faster than token-by-token.
#pragma once
#include "kernels.hpp"

namespace rama {

constexpr int kPB = 8;          // prompt positions per pass
constexpr int kMtWaves = 8;
constexpr int kMtThreads = kMtWaves * 64;

struct MtParams {
    const float* w[3];     // matrices [rows, K]
    const float* x;        // activations [PB, x_stride] (rows beyond n_tok are not read)
    int x_stride;
    const float* nw;       // rmsnorm gain or NULL
    float* o[3];           // outputs [PB, o_stride] (QKV: q, k-scratch, v-scratch)
    int o_stride;
    int K, rows, nmat, n_tok;
    int epi;               // EPI_STORE / EPI_RESID / EPI_QKV, or 3 = SwiGLU (w[0] = w1, w[1] = w3)
    int pos0;              // position of token 0 (EPI_QKV)
    const float* fr; const float* fi; int head_size;
    float* kc; float* vc;  // this layer's cache slabs [seq, dim]
};

// S streams x PB tokens.  x of token t is read at x + t * x_stride (L2-resident, default policy).
template <int S, int NM, int CH, bool NORM>
__device__ __forceinline__ void stream_dots_mt(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb,
                                               const unsigned (&rowoff)[S], const float* x, int x_stride, int n_tok,
                                               __amdgpu_buffer_rsrc_t rn, int K, float (&acc)[S][kPB], float (&ss)[kPB]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = (K + 255) >> 8;
    const unsigned kbytes = (unsigned)K * 4u;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (unsigned)(n_tok * x_stride) * 4u);
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
        for (int t = 0; t < kPB; t++) acc[s][t] = 0.0f;
#pragma unroll
    for (int t = 0; t < kPB; t++) ss[t] = 0.0f;
    for (int c = wave; c < nch; c += CH * kMtWaves) {
        f4 w[S][CH];
        unsigned kb[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int ci = c + j * kMtWaves;
            const unsigned b = (unsigned)(ci * 1024 + lane * 16);
            kb[j] = (ci < nch && b < kbytes) ? b : kOOB;
        }
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int s = 0; s < S; s++)
                w[s][j] = ld_nt((NM == 2 && (s & 1)) ? rb : ra, (kb[j] == kOOB || rowoff[s] == kOOB) ? kOOB : rowoff[s] + kb[j]);
        f4 nv[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) if (NORM) nv[j] = ld_c(rn, kb[j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < kPB; t++) {
#pragma unroll
            for (int j = 0; j < CH; j++) {
                // token rows beyond n_tok fall outside the descriptor and read as 0
                f4 xe = ld_c(rx, (kb[j] == kOOB) ? kOOB : (unsigned)(t * x_stride) * 4u + kb[j]);
                if (NORM) { ss[t] = dot4(xe, xe, ss[t]); xe = xe * nv[j]; }
#pragma unroll
                for (int s = 0; s < S; s++) acc[s][t] = dot4(w[s][j], xe, acc[s][t]);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
        for (int t = 0; t < kPB; t++) acc[s][t] = wave_sum(acc[s][t]);
    if (NORM) {
#pragma unroll
        for (int t = 0; t < kPB; t++) ss[t] = wave_sum(ss[t]);
    }
}

// R = 4 rows (or 2 (w1,w3) pairs) x PB tokens per workgroup; grid = nmat * ceil(rows / R)
template <bool NORM, int EPI>
__global__ __launch_bounds__(kMtThreads) void gemm_mt_rows(MtParams p) {
    constexpr int S = 4, CH = 2;
    constexpr bool PAIR = EPI == 3;
    __shared__ float part[kMtWaves][S + 1][kPB];
    const int rows_per_wg = PAIR ? 2 : 4;
    const int gpm = (p.rows + rows_per_wg - 1) / rows_per_wg;
    const int m = PAIR ? 0 : blockIdx.x / gpm;
    const int r0 = (blockIdx.x - m * gpm) * rows_per_wg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned kbytes = (unsigned)p.K * 4u, mbytes = (unsigned)p.rows * kbytes;
    const float* Wa = PAIR ? p.w[0] : (m == 0 ? p.w[0] : (m == 1 ? p.w[1] : p.w[2]));
    const float* Wb = PAIR ? p.w[1] : Wa;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(Wa, mbytes), rb = make_rsrc(Wb, mbytes);
    const __amdgpu_buffer_rsrc_t rn = make_rsrc(NORM ? p.nw : p.x, kbytes);
    unsigned rowoff[S];
#pragma unroll
    for (int s = 0; s < S; s++) {
        const int r = PAIR ? r0 + (s >> 1) : r0 + s;
        rowoff[s] = r < p.rows ? (unsigned)r * kbytes : kOOB;
    }
    float acc[S][kPB], ss[kPB];
    stream_dots_mt<S, PAIR ? 2 : 1, CH, NORM>(ra, rb, rowoff, p.x, p.x_stride, p.n_tok, rn, p.K, acc, ss);
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < kPB; t++) {
#pragma unroll
            for (int s = 0; s < S; s++) part[wave][s][t] = acc[s][t];
            part[wave][S][t] = ss[t];
        }
    }
    __syncthreads();
    // one thread per (token, output unit)
    const int t = tid / 4, u = tid % 4;
    if (t >= p.n_tok || t >= kPB) return;
    auto total = [&](int s) {
        float v8[kMtWaves];
#pragma unroll
        for (int q = 0; q < kMtWaves; q++) v8[q] = part[q][s][t];
#pragma unroll
        for (int n = kMtWaves; n > 1; n >>= 1)
#pragma unroll
            for (int q = 0; q < n / 2; q++) v8[q] = v8[2 * q] + v8[2 * q + 1];
        return v8[0];
    };
    const float v = NORM ? rms_scale(total(S), p.K) : 1.0f;
    if (PAIR) {
        if (u < 2 && r0 + u < p.rows) {
            float a = total(2 * u) * v;
            const float b = total(2 * u + 1) * v;
            a = a * (1.0f / (1.0f + expf(-a)));
            p.o[0][(size_t)t * p.o_stride + r0 + u] = a * b;
        }
    } else if (EPI == EPI_QKV) {
        if (u < 2) {
            const int r = r0 + 2 * u;
            float a = total(2 * u) * v, b = total(2 * u + 1) * v;
            const int pos = p.pos0 + t;
            if (m < 2) {
                const int i = (r % p.head_size) >> 1;
                const float c = p.fr[(size_t)pos * (p.head_size >> 1) + i], s = p.fi[(size_t)pos * (p.head_size >> 1) + i];
                const float ra_ = a * c - b * s, rb_ = a * s + b * c;
                a = ra_; b = rb_;
            }
            float* o = (m == 0 ? p.o[0] : (m == 1 ? p.o[1] : p.o[2])) + (size_t)t * p.o_stride;
            o[r] = a; o[r + 1] = b;
            if (m == 1) { p.kc[(size_t)pos * p.rows + r] = a; p.kc[(size_t)pos * p.rows + r + 1] = b; }
            if (m == 2) { p.vc[(size_t)pos * p.rows + r] = a; p.vc[(size_t)pos * p.rows + r + 1] = b; }
        }
    } else {
        if (r0 + u < p.rows) {
            float d = total(u) * v;
            float* o = p.o[0] + (size_t)t * p.o_stride + r0 + u;
            if (EPI == EPI_RESID) d = *o + d;
            *o = d;
        }
    }
}

// X[t] = token_embedding_table[tokens[t]]   (infer.rs:13 per prompt position)
__global__ void embed_mt_kernel(float* X, const float* emb, const int* tokens, int n_tok, int dim) {
    const int t = blockIdx.y;
    if (t >= n_tok) return;
    const size_t base = (size_t)tokens[t] * dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x) X[(size_t)t * dim + i] = emb[base + i];
}

}  // namespace rama
