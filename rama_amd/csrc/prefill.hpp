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
    int nsub;              // sub-groups (4 rows / 2 (w1,w3) pairs each) per workgroup, >= 1
    int epi;               // EPI_STORE / EPI_RESID / EPI_QKV, or 3 = SwiGLU (w[0] = w1, w[1] = w3)
    int pos0;              // position of token 0 (EPI_QKV)
    const float* fr; const float* fi; int head_size;
    float* kc; float* vc;  // this layer's cache slabs [seq, dim]
    // batched independent sequences (rama_decode_batch): token t belongs to sequence t
    const SeqSlot* seqs; size_t layer_off;
};

// Wave reduction of 32 per-lane values at once: each butterfly step folds the upper half of the
// value list onto the lower half across one lane bit (31 fold ops instead of 32 x 6).  On return
// v[0] of lane l holds the wave total of value mt_value_of_lane(l) (both lanes of a pair agree).
typedef __attribute__((ext_vector_type(2))) float f2;
__device__ __forceinline__ f2 dot4_pk(f4 a, f4 b, f2 acc) {
    acc = __builtin_elementwise_fma(a.xy, b.xy, acc);
    return __builtin_elementwise_fma(a.zw, b.zw, acc);
}

// fold across lane bit 5 / 4 with the gfx950 lane-swap instructions: after the swap one register
// holds the halves that stay and the other the halves that move, so a fold is swap + add
template <int HALF, bool ROW16>
__device__ __forceinline__ void fold_swap(float (&v)[32]) {
#pragma unroll
    for (int i = 0; i < HALF; i++) {
        const unsigned a = __float_as_uint(v[i]), b = __float_as_uint(v[i + HALF]);
        const auto r = ROW16 ? __builtin_amdgcn_permlane16_swap(a, b, false, false)
                             : __builtin_amdgcn_permlane32_swap(a, b, false, false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
}
// fold across lane bit 3 / 2 / 1 with DPP: lanes with the bit set keep the upper value
template <int HALF>
__device__ __forceinline__ void fold_dpp(float (&v)[32], int lane) {
    constexpr int O = HALF * 2;            // 8, 4 or 2
    const bool up = lane & O;
#pragma unroll
    for (int i = 0; i < HALF; i++) {
        const float keep = up ? v[i + HALF] : v[i];
        const int send = __float_as_int(up ? v[i] : v[i + HALF]);
        int got;
        if (O == 8) got = __builtin_amdgcn_update_dpp(0, send, 0x128, 0xF, 0xF, false);           // row_ror:8 = lane ^ 8
        else if (O == 4) {
            got = __builtin_amdgcn_update_dpp(0, send, 0x104, 0xF, 0x5, false);                     // row_shl:4 into banks 0,2
            got = __builtin_amdgcn_update_dpp(got, send, 0x114, 0xF, 0xA, false);                   // row_shr:4 into banks 1,3
        } else got = __builtin_amdgcn_update_dpp(0, send, 0x4E, 0xF, 0xF, false);                  // quad_perm [2,3,0,1]
        v[i] = keep + __int_as_float(got);
    }
}
__device__ __forceinline__ void wave_sum32(float (&v)[32]) {
    const int lane = threadIdx.x & 63;
    fold_swap<16, false>(v);
    fold_swap<8, true>(v);
    fold_dpp<4>(v, lane);
    fold_dpp<2>(v, lane);
    fold_dpp<1>(v, lane);
    v[0] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[0]), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ int mt_value_of_lane(int lane) { return (lane >> 1) & 31; }

// One workgroup = p.nsub consecutive sub-groups of 4 rows (2 (w1,w3) pairs) x PB tokens; sub-groups
// are numbered across the launch's matrices (Q, K, V), so the host can cut any launch into one even
// round over the CUs.  When a row fits one step (K <= 16 chunks, ONE) each wave keeps ITS chunks of
// the PB activation vectors in registers across its sub-groups, so they are read from L2 once per
// workgroup instead of once per 4 rows; wider rows (W2: K = hidden) re-read them per step.
template <bool NORM, int EPI, bool ONE>
__global__ __launch_bounds__(kMtThreads) void gemm_mt_rows(MtParams p) {
    constexpr int S = 4, CH = 2;
    constexpr bool PAIR = EPI == 3;
    __shared__ float part[2][kMtWaves][S][kPB];
    __shared__ float part_ss[kMtWaves][kPB];
    const int rows_per_sub = PAIR ? 2 : 4;
    const int spm = (p.rows + rows_per_sub - 1) / rows_per_sub;     // sub-groups per matrix
    const int total = (PAIR ? 1 : p.nmat) * spm;
    const int g0 = blockIdx.x * p.nsub;
    const int nsub = min(p.nsub, total - g0);                        // uniform, >= 1 (host sizes the grid)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = (p.K + 255) >> 8;
    const unsigned kbytes = (unsigned)p.K * 4u, mbytes = (unsigned)p.rows * kbytes;
    const __amdgpu_buffer_rsrc_t rn = make_rsrc(NORM ? p.nw : p.x, kbytes);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (unsigned)(p.n_tok * p.x_stride) * 4u);
    constexpr bool one_step = ONE;          // host guarantees nch <= CH * kMtWaves when set

    // activations of this wave's chunks (one-step rows only), gain folded in, and sum x^2 per token
    f4 xr[kPB][CH];
    float ss[kPB];
#pragma unroll
    for (int t = 0; t < kPB; t++) ss[t] = 0.0f;
    if constexpr (one_step) {
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int ci = wave + j * kMtWaves;
            const unsigned b = (unsigned)(ci * 1024 + lane * 16);
            const unsigned kb = (ci < nch && b < kbytes) ? b : kOOB;
            f4 nv = {1.f, 1.f, 1.f, 1.f};
            if (NORM) nv = ld_c(rn, kb);
#pragma unroll
            for (int t = 0; t < kPB; t++) {
                f4 xe = ld_c(rx, kb == kOOB ? kOOB : (unsigned)(t * p.x_stride) * 4u + kb);   // rows >= n_tok read as 0
                if (NORM) { ss[t] = dot4(xe, xe, ss[t]); xe = xe * nv; }
                xr[t][j] = xe;
            }
        }
    }

    // weight loads of (sub-group, step) are issued one iteration ahead: right after the FMAs that
    // free the w registers and BEFORE that sub-group's reduction / epilogue, which then overlap
    // the HBM latency even with one resident workgroup per CU
    const int nsteps = one_step ? 1 : (nch + CH * kMtWaves - 1) / (CH * kMtWaves);
    f4 w[S][CH];
    auto load_w = [&](int sub, int st) {
        const int g = g0 + sub;
        const int m = PAIR ? 0 : g / spm;
        const int r0 = (g - m * spm) * rows_per_sub;
        const float* Wa = PAIR ? p.w[0] : (m == 0 ? p.w[0] : (m == 1 ? p.w[1] : p.w[2]));
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(Wa, mbytes), rb = make_rsrc(PAIR ? p.w[1] : Wa, mbytes);
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int ci = wave + (st * CH + j) * kMtWaves;
            const unsigned b = (unsigned)(ci * 1024 + lane * 16);
            const unsigned kb = (ci < nch && b < kbytes) ? b : kOOB;
#pragma unroll
            for (int s = 0; s < S; s++) {
                const int r = PAIR ? r0 + (s >> 1) : r0 + s;
                w[s][j] = ld_nt((PAIR && (s & 1)) ? rb : ra, (kb == kOOB || r >= p.rows) ? kOOB : (unsigned)r * kbytes + kb);
            }
        }
    };
    load_w(0, 0);
#pragma unroll 1
    for (int sub = 0; sub < nsub; sub++) {
        const int m = PAIR ? 0 : (g0 + sub) / spm;
        const int r0 = (g0 + sub - m * spm) * rows_per_sub;
        f2 acc[S][kPB];                 // .x sums the even elements of a float4, .y the odd ones (v_pk_fma_f32)
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int t = 0; t < kPB; t++) acc[s][t] = f2{0.0f, 0.0f};
#pragma unroll 1
        for (int st = 0; st < nsteps; st++) {
            if constexpr (one_step) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < kPB; t++)
#pragma unroll
                    for (int j = 0; j < CH; j++)
#pragma unroll
                        for (int s = 0; s < S; s++) acc[s][t] = dot4_pk(w[s][j], xr[t][j], acc[s][t]);
            } else {
                unsigned kb[CH];
                f4 nv[CH];
#pragma unroll
                for (int j = 0; j < CH; j++) {
                    const int ci = wave + (st * CH + j) * kMtWaves;
                    const unsigned b = (unsigned)(ci * 1024 + lane * 16);
                    kb[j] = (ci < nch && b < kbytes) ? b : kOOB;
                    nv[j] = f4{1.f, 1.f, 1.f, 1.f};
                    if (NORM) nv[j] = ld_c(rn, kb[j]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < kPB; t++)
#pragma unroll
                    for (int j = 0; j < CH; j++) {
                        f4 xe = ld_c(rx, (kb[j] == kOOB) ? kOOB : (unsigned)(t * p.x_stride) * 4u + kb[j]);
                        if (NORM) { if (sub == 0) ss[t] = dot4(xe, xe, ss[t]); xe = xe * nv[j]; }
#pragma unroll
                        for (int s = 0; s < S; s++) acc[s][t] = dot4_pk(w[s][j], xe, acc[s][t]);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < nsteps) load_w(sub, st + 1);
            else if (sub + 1 < nsub) load_w(sub + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        static_assert(S * kPB == 32, "wave_sum32 takes S x PB = 32 accumulators");
        float flat[32];
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int t = 0; t < kPB; t++) flat[s * kPB + t] = acc[s][t].x + acc[s][t].y;
        wave_sum32(flat);
        const int par = sub & 1;
        if ((lane & 1) == 0) (&part[par][wave][0][0])[mt_value_of_lane(lane)] = flat[0];
        if (sub == 0 && NORM) {
#pragma unroll
            for (int t = 0; t < kPB; t++) { const float v = wave_sum(ss[t]); if (lane == 0) part_ss[wave][t] = v; }
        }
        __syncthreads();
        // one thread per (token, output unit) of this sub-group
        const int t = tid / 4, u = tid % 4;
        if (t < p.n_tok && t < kPB) {
            auto tree = [&](auto get) {
                float v8[kMtWaves];
#pragma unroll
                for (int q = 0; q < kMtWaves; q++) v8[q] = get(q);
#pragma unroll
                for (int n = kMtWaves; n > 1; n >>= 1)
#pragma unroll
                    for (int q = 0; q < n / 2; q++) v8[q] = v8[2 * q] + v8[2 * q + 1];
                return v8[0];
            };
            auto total = [&](int s) { return tree([&](int q) { return part[par][q][s][t]; }); };
            const float v = NORM ? rms_scale(tree([&](int q) { return part_ss[q][t]; }), p.K) : 1.0f;
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
                    const int pos = p.seqs ? p.seqs[t].pos : p.pos0 + t;
                    float* kc = p.seqs ? p.seqs[t].kc + p.layer_off : p.kc;
                    float* vc = p.seqs ? p.seqs[t].vc + p.layer_off : p.vc;
                    if (m < 2) {
                        const int i = (r % p.head_size) >> 1;
                        const float c = p.fr[(size_t)pos * (p.head_size >> 1) + i], sn = p.fi[(size_t)pos * (p.head_size >> 1) + i];
                        const float ra_ = a * c - b * sn, rb_ = a * sn + b * c;
                        a = ra_; b = rb_;
                    }
                    float* o = (m == 0 ? p.o[0] : (m == 1 ? p.o[1] : p.o[2])) + (size_t)t * p.o_stride;
                    o[r] = a; o[r + 1] = b;
                    if (m == 1) { kc[(size_t)pos * p.rows + r] = a; kc[(size_t)pos * p.rows + r + 1] = b; }
                    if (m == 2) { vc[(size_t)pos * p.rows + r] = a; vc[(size_t)pos * p.rows + r + 1] = b; }
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
        // part[] is double-buffered by sub parity: the next sub-group's publish cannot overtake
        // these reads, because every wave passes the next __syncthreads first
    }
}

constexpr int kMtOneStepK = 2 * kMtWaves * 256;   // widest row (floats) the register-resident-activation variant takes

// X[t] = token_embedding_table[tokens[t]]   (infer.rs:13 per prompt position)
__global__ void embed_mt_kernel(float* X, const float* emb, const int* tokens, int n_tok, int dim) {
    const int t = blockIdx.y;
    if (t >= n_tok) return;
    const size_t base = (size_t)tokens[t] * dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x) X[(size_t)t * dim + i] = emb[base + i];
}

}  // namespace rama
