// kernels.hpp -- hand-written HIP kernels for gfx950 (CDNA4, wave64) of rama's fp32
// decode path.  Arithmetic follows the reference CPU backend (engine/src/device/cpu.rs);
// the op order follows engine/src/transformer/infer.rs:8-53.
//
// Design (DESIGN.md has the numbers):
//  * The W.x matvecs are >99 % of the bytes and HBM-bound (0.5 FLOP/B).  Weight rows are
//    streamed once with 16-byte-per-lane `buffer_load_dwordx4 ... nt` straight into VGPRs
//    (no LDS round trip: nothing is shared between waves), every load of a step issued
//    before the first use.  A workgroup of NW waves owns R consecutive rows and splits K
//    across its waves in 256-float (1 KiB) chunks, chunk c going to wave c mod NW, so the
//    workgroup as a whole sweeps each row front to back in NW-KiB contiguous pieces
//    (measured: +10-14 % over giving each wave a contiguous K range; tools/gemv_bench.hip).
//    Partial sums are reduced with DPP inside a wave and through a few bytes of LDS across
//    the waves.  Shipping geometry: NW = 8, R = 4, CH = 2 (rama_api.hip DISPATCH_GEOM).
//  * Buffer (SRSRC) addressing gives hardware bounds checking: out-of-range lanes/chunks
//    get offset 0x80000000 and return 0 without touching memory, so ragged widths
//    (288 = 256 + 32) and row tails need no branches in the load stream.
//  * rmsnorm is folded into the consuming matvec (sum x^2 rides along the same K split),
//    RoPE + KV-cache append into the QKV epilogue, SiLU*gate into W1|W3, residual adds
//    into Wo / W2: five launches per layer.
//  * (token, pos) live in device memory (`Ctl`) so a decode step can be replayed from a
//    hipGraph and chained to the device-side argmax with no host round trip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rama {

typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

struct Ctl {          // device-resident decode cursor
    int token;        // token fed to the next forward
    int pos;          // its position
    int n_forced;     // forced (prompt) tokens: next = forced[pos] while pos < n_forced
    int n_out;        // tokens written to `out` so far
};

constexpr unsigned kOOB = 0x80000000u;   // >= any num_records we build: load returns 0
constexpr int kWG = 256;                 // workgroup size of the non-matvec kernels

// ---------------------------------------------------------------- small device helpers

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// streamed-once weights: non-temporal (aux bit 1)
__device__ __forceinline__ f4 ld_nt(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 2));
}
// shared activations: default policy (L2 resident)
__device__ __forceinline__ f4 ld_c(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_movi(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ float lane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// sum over the 16 lanes of a DPP row; every lane of the row gets the sum
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return v;
}
// sum over all 64 lanes, fixed order -- (r0 + r1) + (r2 + r3) over the four row sums -- the same in every lane.  The rows meet
// through gfx950's lane swaps (v_permlane16_swap: rows 0|1 and 2|3, v_permlane32_swap: the halves): two VALU instructions
// each instead of four v_readlane and the trip through the scalar registers.
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    {
        const unsigned u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
__device__ __forceinline__ int wave_max_i(int v) {
    v = max(v, dpp_movi<0xB1>(v));
    v = max(v, dpp_movi<0x4E>(v));
    v = max(v, dpp_movi<0x141>(v));
    v = max(v, dpp_movi<0x140>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
// minimum over a workgroup of up to 16 waves (one barrier pair; every thread gets the result)
__device__ __forceinline__ int block_min_i(int v) {
    __shared__ int s_m[16];
    v = wave_min_i(v);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = v;
    __syncthreads();
    int r = s_m[0];
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 1; w < nw; w++) r = min(r, s_m[w]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ int block_sum_i(int v) {
    __shared__ int s_s[16];
    v = wave_sum_i(v);
    if ((threadIdx.x & 63) == 0) s_s[threadIdx.x >> 6] = v;
    __syncthreads();
    int r = 0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; w++) r += s_s[w];
    __syncthreads();
    return r;
}
__device__ __forceinline__ float dot4(f4 a, f4 b, float acc) {
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
    return acc;
}

// ---------------------------------------------------------------- the streaming core
// S weight streams (rows) of width K against one activation vector; the row's 256-float
// chunks are dealt round-robin to the workgroup's NW waves (chunk c -> wave c mod NW) and a
// wave takes CH of its chunks per step.  On return acc[s] / ss hold this WAVE's
// (wave-reduced, wave-uniform) partial sums.
//   NM   : streams alternate over NM matrices (s % NM): 1, or 2 for W1|W3
//   NORM : activations are nw[k]*x[k]; ss accumulates sum x[k]^2 (rmsnorm folded in)
//   SOLO : the calling wave owns its rows alone and sweeps every chunk itself (small-K kernels:
//          a row is a few KiB, several waves per row would mostly idle and need an LDS turn)
template <int S, int NM, int CH, int NW, bool NORM, bool SOLO = false>
__device__ __forceinline__ void stream_dots(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb,
                                            const unsigned (&rowoff)[S],
                                            __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t rn,
                                            int K, float (&acc)[S], float& ss) {
    const int lane = threadIdx.x & 63;
    const int wave = SOLO ? 0 : threadIdx.x >> 6;
    const int nch = (K + 255) >> 8;          // 256-float chunks in a row
    const unsigned kbytes = (unsigned)K * 4u;
    static_assert(!SOLO || NW == 1, "a solo wave deals the chunks to itself");
#pragma unroll
    for (int s = 0; s < S; s++) acc[s] = 0.0f;
    ss = 0.0f;
    for (int c = wave; c < nch; c += CH * NW) {
        f4 w[S][CH];
        f4 xv[CH];
        f4 nv[CH];
        unsigned kb[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int ci = c + j * NW;
            const unsigned b = (unsigned)(ci * 1024 + lane * 16);
            kb[j] = (ci < nch && b < kbytes) ? b : kOOB;
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
#pragma unroll
            for (int s = 0; s < S; s++) {
                const unsigned o = (kb[j] == kOOB) ? kOOB : rowoff[s] + kb[j];
                w[s][j] = ld_nt((NM == 2 && (s & 1)) ? rb : ra, o);
            }
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
            xv[j] = ld_c(rx, kb[j]);
            if (NORM) nv[j] = ld_c(rn, kb[j]);
        }
        // every load of the step is in flight before the first FMA: without this fence
        // hipcc's max-occupancy scheduler sinks loads next to their uses to save VGPRs
        // and leaves only a few outstanding.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CH; j++) {
            f4 xe = xv[j];
            if (NORM) {
                ss = dot4(xv[j], xv[j], ss);
                xe = xv[j] * nv[j];
            }
#pragma unroll
            for (int s = 0; s < S; s++) acc[s] = dot4(w[s][j], xe, acc[s]);
        }
    }
#pragma unroll
    for (int s = 0; s < S; s++) acc[s] = wave_sum(acc[s]);
    if (NORM) ss = wave_sum(ss);
}

// cross-wave combine through LDS: part[wave][0..S-1] = acc, part[wave][S] = ss
template <int S, int NW>
__device__ __forceinline__ void publish_partials(float (*part)[S + 1], const float (&acc)[S], float ss) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int s = 0; s < S; s++) part[wave][s] = acc[s];
        part[wave][S] = ss;
    }
    __syncthreads();
}
template <int S, int NW>
__device__ __forceinline__ float combined(float (*part)[S + 1], int s) {
    float t[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) t[w] = part[w][s];
#pragma unroll
    for (int n = NW; n > 1; n >>= 1)      // fixed pairwise tree
#pragma unroll
        for (int w = 0; w < n / 2; w++) t[w] = t[2 * w] + t[2 * w + 1];
    return t[0];
}
// cpu.rs:110-113: v = 1/sqrt(sum(x^2)/len + 1e-5)
__device__ __forceinline__ float rms_scale(float ss, int n) {
    return 1.0f / sqrtf(ss / (float)n + 1e-5f);
}

// ---------------------------------------------------------------- matvec kernels

enum { EPI_STORE = 0, EPI_RESID = 1, EPI_QKV = 2, EPI_SWIGLU_PAIR = 5 };   // 3, 4: prefill_mfma.hpp

struct GemvParams {
    const float* w[3];     // nmat matrices, each [rows, K] row-major
    const float* x;        // [K]
    const float* nw;       // [K] rmsnorm gain (NORM)
    float* o[3];           // outputs per matrix (EPI_STORE: o[0]; EPI_RESID: o[0] += ; EPI_QKV: q,k,v)
    int K, rows, nmat;
    // EPI_QKV
    const Ctl* ctl;        // device cursor, or NULL -> pos_val
    int pos_val;
    const float* fr;       // freq_cis_real [seq, hs/2]
    const float* fi;
    int head_size;
    float* kc;             // this layer's key cache   [seq, dim]
    float* vc;             // this layer's value cache [seq, dim]
    unsigned* zero_me;     // optional: a counter the NEXT launch uses, cleared here (stream order publishes it)
};

// R rows per workgroup of NW waves; grid = nmat * ceil(rows/R) (tail rows read as 0)
template <int R, int CH, int NW, bool NORM, int EPI>
__global__ __launch_bounds__(NW * 64) void gemv_rows(GemvParams p) {
    __shared__ float part[NW][R + 1];
    const int groups_per_mat = (p.rows + R - 1) / R;
    const int m = blockIdx.x / groups_per_mat;
    const int r0 = (blockIdx.x - m * groups_per_mat) * R;
    const int t = threadIdx.x;
    const float* W = (m == 0) ? p.w[0] : (m == 1 ? p.w[1] : p.w[2]);
    float* o = (m == 0) ? p.o[0] : (m == 1 ? p.o[1] : p.o[2]);
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(W, (unsigned)p.rows * (unsigned)p.K * 4u);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (unsigned)p.K * 4u);
    const __amdgpu_buffer_rsrc_t rn = make_rsrc(NORM ? p.nw : p.x, (unsigned)p.K * 4u);
    unsigned rowoff[R];
#pragma unroll
    for (int s = 0; s < R; s++)
        rowoff[s] = (r0 + s < p.rows) ? (unsigned)(r0 + s) * (unsigned)p.K * 4u : kOOB;

    // epilogue operands are fetched up front so their latency hides behind the weight stream
    float resid = 0.0f, rc = 1.0f, rs = 0.0f;
    int pos = 0;
    if (EPI == EPI_RESID) {
        if (t < R && r0 + t < p.rows) resid = o[r0 + t];
    } else if (EPI == EPI_QKV) {
        if (p.zero_me && blockIdx.x == 0 && t == 0) *p.zero_me = 0u;
        pos = p.ctl ? p.ctl->pos : p.pos_val;
        if (t < R / 2 && m < 2) {   // table row pos, entry (r % hs)/2  (infer.rs:15-16)
            const int i = ((r0 + 2 * t) % p.head_size) >> 1;
            rc = p.fr[(size_t)pos * (p.head_size >> 1) + i];
            rs = p.fi[(size_t)pos * (p.head_size >> 1) + i];
        }
    }

    float acc[R], ss;
    stream_dots<R, 1, CH, NW, NORM>(ra, ra, rowoff, rx, rn, p.K, acc, ss);
    publish_partials<R, NW>(part, acc, ss);

    if (EPI == EPI_STORE || EPI == EPI_RESID) {
        if (t < R && r0 + t < p.rows) {
            float d = combined<R, NW>(part, t);
            if (NORM) d *= rms_scale(combined<R, NW>(part, R), p.K);
            if (EPI == EPI_RESID) d = resid + d;     // infer.rs:37,47  x[i] += y[i]
            o[r0 + t] = d;
        }
    } else if (EPI == EPI_SWIGLU_PAIR) {
        // rows (2i, 2i + 1) = row i of W1 and of W3, stored interleaved (model.hip): infer.rs:41-45
        if (t < R / 2 && r0 + 2 * t < p.rows) {
            const float v = rms_scale(combined<R, NW>(part, R), p.K);
            float a = combined<R, NW>(part, 2 * t) * v;
            const float b = combined<R, NW>(part, 2 * t + 1) * v;
            a = a * (1.0f / (1.0f + expf(-a)));          // cpu.rs:56
            o[(r0 >> 1) + t] = a * b;                    // cpu.rs:59-64
        }
    } else {   // EPI_QKV: rows are (even, odd) pairs of one head (R even, head_size even)
        if (t < R / 2) {
            const int r = r0 + 2 * t;
            float a = combined<R, NW>(part, 2 * t), b = combined<R, NW>(part, 2 * t + 1);
            if (NORM) {
                const float v = rms_scale(combined<R, NW>(part, R), p.K);
                a *= v; b *= v;
            }
            if (m < 2) {   // cpu.rs:87-96 rotate (q, k)
                const float ra_ = a * rc - b * rs;
                const float rb_ = a * rs + b * rc;
                a = ra_; b = rb_;
            }
            o[r] = a; o[r + 1] = b;
            if (m == 1) {          // infer.rs:32  key_cache[lo + pos*dim ..] = k
                p.kc[(size_t)pos * p.rows + r] = a; p.kc[(size_t)pos * p.rows + r + 1] = b;
            } else if (m == 2) {   // infer.rs:33
                p.vc[(size_t)pos * p.rows + r] = a; p.vc[(size_t)pos * p.rows + r + 1] = b;
            }
        }
    }
}

struct SwigluParams {
    const float* w1;   // [rows, K]
    const float* w3;   // [rows, K]
    const float* x;    // [K]
    const float* nw;   // [K]
    float* hb;         // [rows]
    int K, rows;
};

// hb[r] = silu(w1[r].xs) * (w3[r].xs), xs = rmsnorm(x) * nw   (infer.rs:39-45, cpu.rs:54-64)
template <int R2, int CH, int NW>
__device__ __forceinline__ void gemv_swiglu_body(const SwigluParams& p) {
    constexpr int S = 2 * R2;
    __shared__ float part[NW][S + 1];
    const int r0 = blockIdx.x * R2;
    const unsigned mbytes = (unsigned)p.rows * (unsigned)p.K * 4u;
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(p.w1, mbytes);
    const __amdgpu_buffer_rsrc_t r3 = make_rsrc(p.w3, mbytes);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (unsigned)p.K * 4u);
    const __amdgpu_buffer_rsrc_t rn = make_rsrc(p.nw, (unsigned)p.K * 4u);
    unsigned rowoff[S];
#pragma unroll
    for (int s = 0; s < S; s++)
        rowoff[s] = (r0 + (s >> 1) < p.rows) ? (unsigned)(r0 + (s >> 1)) * (unsigned)p.K * 4u : kOOB;
    float acc[S], ss;
    stream_dots<S, 2, CH, NW, true>(r1, r3, rowoff, rx, rn, p.K, acc, ss);
    publish_partials<S, NW>(part, acc, ss);
    const int t = threadIdx.x;
    if (t < R2 && r0 + t < p.rows) {
        const float v = rms_scale(combined<S, NW>(part, S), p.K);
        float a = combined<S, NW>(part, 2 * t) * v;
        const float b = combined<S, NW>(part, 2 * t + 1) * v;
        a = a * (1.0f / (1.0f + expf(-a)));   // cpu.rs:56
        p.hb[r0 + t] = a * b;                 // cpu.rs:59-64
    }
}
template <int R2, int CH, int NW>
__global__ __launch_bounds__(NW * 64) void gemv_swiglu(SwigluParams p) { gemv_swiglu_body<R2, CH, NW>(p); }

// ---------------------------------------------------------------- small-K matvecs (dim <= ~2048)
// At the stories15M / 110M widths a weight row is 1-3 KiB: with K split over 8 waves most of the
// workgroup idles, yet every launch still pays the cross-wave LDS turn and two barriers -- and these
// launches are latency-bound (a dependent launch costs 1.6 us empty, ~2.4 us streaming 1 MB;
// tools/launch_floor.hip).  Here ONE wave owns R rows (2 (w1,w3) pairs) outright: it sweeps the whole
// row, reduces with DPP and runs the epilogue from registers; no LDS, no barrier, waves of a workgroup
// are independent.  Same arithmetic per wave as the big kernels (chunk order front to back).
template <int S>
__device__ __forceinline__ float pick(const float (&a)[S], int i) {
    // the asm keeps each element an opaque register value: otherwise hipcc folds the select chain
    // into ONE indexed load of the array, parks the array in LDS (promoted alloca) and reads the
    // workgroup size from the dispatch packet in host memory to find its slot -- ~25 us per kernel
    float v = a[0];
    asm volatile("" : "+v"(v));
#pragma unroll
    for (int s = 1; s < S; s++) {
        float e = a[s];
        asm volatile("" : "+v"(e));
        v = (i == s) ? e : v;
    }
    return v;
}

constexpr int kSoloWaves = 1;

template <int R, int CH, bool NORM, int EPI>
__global__ __launch_bounds__(kSoloWaves * 64) void gemv_rows_solo(GemvParams p) {
    const int lane = threadIdx.x & 63;
    const int groups_per_mat = (p.rows + R - 1) / R;
    // readfirstlane: the wave index is wave-uniform, but only an SGPR tells hipcc so -- otherwise every
    // buffer descriptor below sits in VGPRs and each load becomes a waterfall loop
    const int g = blockIdx.x * kSoloWaves + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (g >= p.nmat * groups_per_mat) return;
    const int m = g / groups_per_mat;
    const int r0 = (g - m * groups_per_mat) * R;
    const int t = lane;
    const float* W = (m == 0) ? p.w[0] : (m == 1 ? p.w[1] : p.w[2]);
    float* o = (m == 0) ? p.o[0] : (m == 1 ? p.o[1] : p.o[2]);
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(W, (unsigned)p.rows * (unsigned)p.K * 4u);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (unsigned)p.K * 4u);
    const __amdgpu_buffer_rsrc_t rn = make_rsrc(NORM ? p.nw : p.x, (unsigned)p.K * 4u);
    unsigned rowoff[R];
#pragma unroll
    for (int s = 0; s < R; s++)
        rowoff[s] = (r0 + s < p.rows) ? (unsigned)(r0 + s) * (unsigned)p.K * 4u : kOOB;
    float resid = 0.0f, rc = 1.0f, rs = 0.0f;
    int pos = 0;
    if (EPI == EPI_RESID) {
        if (t < R && r0 + t < p.rows) resid = o[r0 + t];
    } else if (EPI == EPI_QKV) {
        if (p.zero_me && g == 0 && t == 0) *p.zero_me = 0u;
        pos = p.ctl ? p.ctl->pos : p.pos_val;
        if (t < R / 2 && m < 2) {
            const int i = ((r0 + 2 * t) % p.head_size) >> 1;
            rc = p.fr[(size_t)pos * (p.head_size >> 1) + i];
            rs = p.fi[(size_t)pos * (p.head_size >> 1) + i];
        }
    }
    float acc[R], ss;
    stream_dots<R, 1, CH, 1, NORM, true>(ra, ra, rowoff, rx, rn, p.K, acc, ss);
    const float v = NORM ? rms_scale(ss, p.K) : 1.0f;
    if (EPI == EPI_STORE || EPI == EPI_RESID) {
        if (t < R && r0 + t < p.rows) {
            float d = pick<R>(acc, t);
            if (NORM) d *= v;
            if (EPI == EPI_RESID) d = resid + d;
            o[r0 + t] = d;
        }
    } else if (EPI == EPI_SWIGLU_PAIR) {
        if (t < R / 2 && r0 + 2 * t < p.rows) {
            float a = pick<R>(acc, 2 * t) * v;
            const float b = pick<R>(acc, 2 * t + 1) * v;
            a = a * (1.0f / (1.0f + expf(-a)));
            o[(r0 >> 1) + t] = a * b;
        }
    } else {
        if (t < R / 2) {
            const int r = r0 + 2 * t;
            float a = pick<R>(acc, 2 * t), b = pick<R>(acc, 2 * t + 1);
            if (NORM) { a *= v; b *= v; }
            if (m < 2) {
                const float ra_ = a * rc - b * rs;
                const float rb_ = a * rs + b * rc;
                a = ra_; b = rb_;
            }
            o[r] = a; o[r + 1] = b;
            if (m == 1) { p.kc[(size_t)pos * p.rows + r] = a; p.kc[(size_t)pos * p.rows + r + 1] = b; }
            else if (m == 2) { p.vc[(size_t)pos * p.rows + r] = a; p.vc[(size_t)pos * p.rows + r + 1] = b; }
        }
    }
}

template <int R2, int CH>
__global__ __launch_bounds__(kSoloWaves * 64) void gemv_swiglu_solo(SwigluParams p) {
    constexpr int S = 2 * R2;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * kSoloWaves + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = g * R2;
    if (r0 >= p.rows) return;
    const unsigned mbytes = (unsigned)p.rows * (unsigned)p.K * 4u;
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(p.w1, mbytes);
    const __amdgpu_buffer_rsrc_t r3 = make_rsrc(p.w3, mbytes);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (unsigned)p.K * 4u);
    const __amdgpu_buffer_rsrc_t rn = make_rsrc(p.nw, (unsigned)p.K * 4u);
    unsigned rowoff[S];
#pragma unroll
    for (int s = 0; s < S; s++)
        rowoff[s] = (r0 + (s >> 1) < p.rows) ? (unsigned)(r0 + (s >> 1)) * (unsigned)p.K * 4u : kOOB;
    float acc[S], ss;
    stream_dots<S, 2, CH, 1, true, true>(r1, r3, rowoff, rx, rn, p.K, acc, ss);
    if (lane < R2 && r0 + lane < p.rows) {
        const float v = rms_scale(ss, p.K);
        float a = pick<S>(acc, 2 * lane) * v;
        const float b = pick<S>(acc, 2 * lane + 1) * v;
        a = a * (1.0f / (1.0f + expf(-a)));   // cpu.rs:56
        p.hb[r0 + lane] = a * b;              // cpu.rs:59-64
    }
}

// generic o_cols > 1 product of the trait signature (never used by forward): one thread per output
__global__ void matmul_generic(float* o, const float* a, const float* b, int width, int o_rows, int o_cols) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= o_rows * o_cols) return;
    int r = idx / o_cols, c = idx % o_cols;
    float acc = 0.0f;
    for (int k = 0; k < width; k++) acc = fmaf(a[(size_t)r * width + k], b[(size_t)k * o_cols + c], acc);
    o[idx] = acc;
}
// unaligned-view fallback of the o_cols == 1 product (16-byte alignment not given): one wave per row
__global__ __launch_bounds__(kWG) void matvec_unaligned(float* o, const float* a, const float* x, int width, int rows) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int lane = threadIdx.x & 63;
    float acc = 0.0f;
    for (int k = lane; k < width; k += 64) acc = fmaf(a[(size_t)row * width + k], x[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) o[row] = acc;
}

// ---------------------------------------------------------------- attention (cpu.rs:23-52)
// One workgroup of 16 waves per head.  G lanes cooperate on one cached row segment (head_size
// floats = G x float4), so a wave covers TPW = 64/G timesteps per load instruction and the
// workgroup a tile of 16*TPW*U timesteps per round (256 at head_size 128), U = 8 independent
// 16-byte loads in flight per lane: decode attention is latency-bound (a few MB of cache per
// layer against 255 idle CUs), so what matters is how few dependent memory round trips the
// chain has, not bandwidth.  Cache rows are read through a buffer descriptor: timesteps beyond
// pos and the padding lanes of head_size 48 get offset 0x80000000 and read as 0 -- no branches
// in the load stream.  The first V tile is requested before the softmax, which does not need
// it, so its latency hides behind the exp/sum phase.
// one independent sequence of a batched decode step (rama_decode_batch): its cache bases and position
struct SeqSlot { float* kc; float* vc; int pos; int pad; };

struct AttnParams {
    const float* q;       // [dim]
    const float* kc;      // this layer's key cache   [seq, dim]
    const float* vc;      // this layer's value cache [seq, dim]
    float* att;           // [n_heads, seq_len] probabilities (reference scratch) or NULL
    float* xb;            // [dim] output
    const Ctl* ctl;
    int pos_val;
    int dim, head_size, seq_len;
    // split-T mode (long contexts): grid (n_heads, nsplit); workgroup (h, s) covers timesteps
    // [s*chunk, (s+1)*chunk) and writes an un-normalised partial {max, sum, acc[hs]} to part
    float* part;          // [n_heads, nsplit, head_size + 4]
    int nsplit;
    // batched queries (prefill): blockIdx.z = query index; query z sits at position pos + z
    int q_stride, xb_stride;
    // tiled = 1: q and xb are in the MFMA tile layout of prefill_mfma.hpp ([tokens, dim] in 16 x 16
    // blocks); the 16 bytes of (token z, floats k .. k + 3) sit at attn_tile_idx(z, k, dim)
    int tiled;
    // batched sequences (rama_decode_batch): query z belongs to sequence z with its own caches and
    // position; layer_off = floats from a cache base to this layer's slab
    const SeqSlot* seqs; size_t layer_off;
};

__device__ __forceinline__ size_t attn_tile_idx(int tk, int k, int K) {      // = tile_idx of prefill_mfma.hpp, k % 4 == 0
    return ((size_t)((tk >> 4) * (K >> 4) + (k >> 4)) * 64 + (size_t)(((k >> 2) & 3) * 16 + (tk & 15))) * 4;
}

constexpr int kAttnWaves = 16;
constexpr int kAttnThreads = kAttnWaves * 64;
// LDS floats: [2 * W] max / sum, [W * G * 4] partial outputs, [seq_len] scores (W waves, default 16)
__host__ __device__ constexpr int attn_scratch_floats(int G, int W = kAttnWaves) { return 2 * W + W * G * 4; }

// W = 4: the token-batch passes' short contexts (a few dozen timesteps per query, thousands of
// (head, query) workgroups): a round of 4 waves already covers 64 timesteps, and 8 such workgroups fit a CU
// NT: cache rows are fetched non-temporally (long contexts: every row is read once per step and the
// 2 GB/token K/V stream of llama2-7B at 2K context should not evict anything)
// UU: cache rows a lane keeps in flight per round (K and V each).  16 at 8 waves makes a round 256
// timesteps (head_size 128): a whole split-T slice of a 2048-token context, so K AND V are requested
// up front and the slice costs one memory round trip instead of three.
template <int G, bool SPLIT, int W = kAttnWaves, bool NT = false, int UU = 8>
__global__ __launch_bounds__(W * 64) void attention_kernel(AttnParams p) {
    constexpr int kAttnWaves = W, kAttnThreads = W * 64;       // shadow the 16-wave defaults
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_max = sm;
    float* s_sum = sm + kAttnWaves;
    float* s_acc = sm + 2 * kAttnWaves;
    float* s_att = sm + attn_scratch_floats(G, W);
    constexpr int U = UU;
    constexpr int TPW = 64 / G;                       // timesteps per wave-instruction
    constexpr int TILE = kAttnWaves * TPW * U;        // timesteps per workgroup round
    const int h = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int zq = (int)blockIdx.z;
    const int pos = p.seqs ? p.seqs[zq].pos : (p.ctl ? p.ctl->pos : p.pos_val) + zq;
    const int hs = p.head_size;
    // timesteps [t0, t1) of this workgroup: everything, or one slice in split-T mode
    int t0 = 0, t1 = pos + 1;
    if (SPLIT) {
        const int chunk = (pos + 1 + p.nsplit - 1) / p.nsplit;
        t0 = (int)blockIdx.y * chunk;
        t1 = min(t0 + chunk, pos + 1);
    }
    const int li = lane % G;                 // my float4 within the head segment
    const int tg = lane / G;                 // my timestep slot within the wave
    const bool lane_ok = li * 4 < hs;
    const unsigned cache_bytes = (unsigned)p.seq_len * (unsigned)p.dim * 4u;
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(p.seqs ? p.seqs[zq].kc + p.layer_off : p.kc, cache_bytes);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(p.seqs ? p.seqs[zq].vc + p.layer_off : p.vc, cache_bytes);
    const unsigned col = (unsigned)(h * hs + li * 4) * 4u;
    const unsigned rowb = (unsigned)p.dim * 4u;
    // timestep of slot u in the round starting at `base`
    auto t_of = [&](int base, int u) { return base + (u * kAttnWaves + wave) * TPW + tg; };
    auto off_of = [&](int t) { return (lane_ok && t < t1) ? (unsigned)t * rowb + col : kOOB; };

    f4 q4 = {0.f, 0.f, 0.f, 0.f};
    if (lane_ok) q4 = *reinterpret_cast<const f4*>(p.tiled ? p.q + attn_tile_idx(zq, h * hs + li * 4, p.dim)
                                                           : p.q + (size_t)zq * p.q_stride + (size_t)h * hs + (size_t)li * 4);
    const float div = sqrtf((float)hs);

    f4 kt[U], vt[U];
#pragma unroll
    for (int u = 0; u < U; u++) kt[u] = NT ? ld_nt(rk, off_of(t_of(t0, u))) : ld_c(rk, off_of(t_of(t0, u)));
#pragma unroll
    for (int u = 0; u < U; u++) vt[u] = NT ? ld_nt(rv, off_of(t_of(t0, u))) : ld_c(rv, off_of(t_of(t0, u)));     // needed only after the softmax
    __builtin_amdgcn_sched_barrier(0);

    // scores: att[t] = (q . k_t) / sqrt(hs)     (cpu.rs:34-41); s_att is indexed from t0
    for (int base = t0; base < t1; base += TILE) {
        if (base > t0) {
#pragma unroll
            for (int u = 0; u < U; u++) kt[u] = NT ? ld_nt(rk, off_of(t_of(base, u))) : ld_c(rk, off_of(t_of(base, u)));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float d = dot4(q4, kt[u], 0.0f);
            d = row16_sum(d);
            if (G == 32) d += __shfl_xor(d, 16);
            if (G == 64) { d += __shfl_xor(d, 16); d += __shfl_xor(d, 32); }
            const int t = t_of(base, u);
            if (li == 0 && t < t1) s_att[t - t0] = d / div;
        }
    }
    __syncthreads();

    // softmax over the slice (cpu.rs:187-192): max, exp(a - max), sum, divide
    const int nt = t1 - t0;
    float mx = -INFINITY;
    for (int t = tid; t < nt; t += kAttnThreads) mx = fmaxf(mx, s_att[t]);
    mx = wave_max(mx);
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();
    mx = s_max[0];
#pragma unroll
    for (int w = 1; w < kAttnWaves; w++) mx = fmaxf(mx, s_max[w]);
    float sum = 0.0f;
    for (int t = tid; t < nt; t += kAttnThreads) {
        float e = expf(s_att[t] - mx);
        s_att[t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    {
        float t8[kAttnWaves];
#pragma unroll
        for (int w = 0; w < kAttnWaves; w++) t8[w] = s_sum[w];
#pragma unroll
        for (int n = kAttnWaves; n > 1; n >>= 1)
#pragma unroll
            for (int w = 0; w < n / 2; w++) t8[w] = t8[2 * w] + t8[2 * w + 1];
        sum = t8[0];
    }
    if (!SPLIT) {       // a slice keeps exp(a - slice max); the combine kernel rescales and divides
        for (int t = tid; t < nt; t += kAttnThreads) {
            float a = s_att[t] / sum;
            s_att[t] = a;
            if (p.att) p.att[(size_t)h * p.seq_len + t] = a;
        }
        __syncthreads();
    }

    // xb[i] = sum_t att[t] * v_t[i]     (cpu.rs:43-49)
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int base = t0; base < t1; base += TILE) {
        if (base > t0) {
#pragma unroll
            for (int u = 0; u < U; u++) vt[u] = NT ? ld_nt(rv, off_of(t_of(base, u))) : ld_c(rv, off_of(t_of(base, u)));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t_of(base, u);
            const float a = (t < t1) ? s_att[t - t0] : 0.0f;
            acc.x = fmaf(a, vt[u].x, acc.x); acc.y = fmaf(a, vt[u].y, acc.y);
            acc.z = fmaf(a, vt[u].z, acc.z); acc.w = fmaf(a, vt[u].w, acc.w);
        }
    }
    // fold the TPW timestep slots of the wave, then the waves (fixed pairwise tree)
#pragma unroll
    for (int m = G; m < 64; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m); acc.y += __shfl_xor(acc.y, m);
        acc.z += __shfl_xor(acc.z, m); acc.w += __shfl_xor(acc.w, m);
    }
    if (lane < G) *reinterpret_cast<f4*>(s_acc + (wave * G + lane) * 4) = acc;
    __syncthreads();
    if (tid < G && tid * 4 < hs) {
        f4 t8[kAttnWaves];
#pragma unroll
        for (int w = 0; w < kAttnWaves; w++) t8[w] = *reinterpret_cast<f4*>(s_acc + (w * G + tid) * 4);
#pragma unroll
        for (int n = kAttnWaves; n > 1; n >>= 1)
#pragma unroll
            for (int w = 0; w < n / 2; w++) t8[w] = t8[2 * w] + t8[2 * w + 1];
        if (SPLIT) {
            float* o = p.part + ((size_t)h * p.nsplit + blockIdx.y) * (size_t)(hs + 4);
            *reinterpret_cast<f4*>(o + 4 + (size_t)tid * 4) = t8[0];
            if (tid == 0) { o[0] = mx; o[1] = sum; }
        } else {
            *reinterpret_cast<f4*>(p.tiled ? p.xb + attn_tile_idx(zq, h * hs + tid * 4, p.dim)
                                           : p.xb + (size_t)zq * p.xb_stride + (size_t)h * hs + (size_t)tid * 4) = t8[0];
        }
    }
}

// split-T combine: xb[h] = sum_s e^(m_s - M) acc_s / sum_s e^(m_s - M) l_s,  M = max_s m_s
// (algebraically the softmax over all timesteps; an empty slice has l = 0 and drops out)
// the plain loop (round 1), kept for A/B (rama_set_tuning "combine_v" = 0)
__global__ void attention_combine_loop_kernel(const float* part, float* xb, int head_size, int nsplit) {
    const int h = blockIdx.x, i = threadIdx.x;
    const size_t ps = (size_t)(head_size + 4);
    const float* ph = part + (size_t)h * nsplit * ps;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; s++) if (ph[s * ps + 1] > 0.0f) M = fmaxf(M, ph[s * ps]);
    float L = 0.0f, o = 0.0f;
    for (int s = 0; s < nsplit; s++) {
        const float l = ph[s * ps + 1];
        if (l > 0.0f) {
            const float sc = expf(ph[s * ps] - M);
            L += sc * l;
            if (i < head_size) o += sc * ph[s * ps + 4 + i];
        }
    }
    if (i < head_size) xb[(size_t)h * head_size + i] = o / L;
}

template <int MAXS>
__global__ void attention_combine_kernel(const float* part, float* xb, int head_size, int nsplit) {
    const int h = blockIdx.x, i = threadIdx.x;
    const size_t ps = (size_t)(head_size + 4);
    const float* ph = part + (size_t)h * nsplit * ps;
    // every slice's (max, sum, acc[i]) is requested before the first use: the launch is one cache
    // round trip, not one per slice (it sits on the critical path of every long-context layer)
    float m[MAXS], l[MAXS], a[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; s++) {
        const bool on = s < nsplit;
        m[s] = on ? ph[s * ps] : -INFINITY;
        l[s] = on ? ph[s * ps + 1] : 0.0f;
        a[s] = (on && i < head_size) ? ph[s * ps + 4 + i] : 0.0f;
    }
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < MAXS; s++) if (l[s] > 0.0f) M = fmaxf(M, m[s]);
    float L = 0.0f, o = 0.0f;
#pragma unroll
    for (int s = 0; s < MAXS; s++) {
        if (l[s] > 0.0f) {
            const float sc = expf(m[s] - M);
            L += sc * l[s];
            o += sc * a[s];
        }
    }
    if (i < head_size) xb[(size_t)h * head_size + i] = o / L;
}

// ---------------------------------------------------------------- small ops

// infer.rs:13: x = token_embedding_table[token*dim .. (token+1)*dim]
__global__ void embed_kernel(float* x, const float* emb, const Ctl* ctl, int token_val, int dim) {
    const int token = ctl ? ctl->token : token_val;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x)
        x[i] = emb[(size_t)token * dim + i];
}

__global__ void set_ctl_kernel(Ctl* ctl, int token, int pos, int n_forced, int n_out) {
    ctl->token = token; ctl->pos = pos; ctl->n_forced = n_forced; ctl->n_out = n_out;
}

// cursor from a token id that lives in device memory (pipeline stages); an id outside the
// vocabulary is clamped so a corrupted hand-off cannot turn into a wild embedding read
__global__ void set_ctl_dev_kernel(Ctl* ctl, const int* token_dev, int pos, int vocab) {
    int t = token_dev ? *token_dev : 0;
    t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
    ctl->token = t; ctl->pos = pos; ctl->n_forced = 0; ctl->n_out = 0;
}

__global__ void array_add_kernel(float* t, const float* s, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) t[i] += s[i];
}
__global__ void array_mult_kernel(float* t, const float* s, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) t[i] *= s[i];
}
__global__ void sinu_kernel(float* o, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float a = o[i];
        o[i] = a * (1.0f / (1.0f + expf(-a)));
    }
}
// two Device::copy_from_slice calls as one launch (infer.rs:32-33: the key row and the value row of the cache)
__global__ void copy2_kernel(float* t1, const float* s1, size_t n1, float* t2, const float* s2, size_t n2) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n1 + n2; i += (size_t)gridDim.x * blockDim.x) {
        if (i < n1) t1[i] = s1[i]; else t2[i - n1] = s2[i - n1];
    }
}
// Device::sinu followed by Device::array_mult on the same vector (infer.rs:44-45) as one launch, each product rounded as in the two
__global__ void sinu_mult_kernel(float* o, const float* s, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float a = o[i];
        a = a * (1.0f / (1.0f + expf(-a)));
        o[i] = a * s[i];
    }
}
__global__ void copy_kernel(float* t, const float* s, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) t[i] = s[i];
}

// block-wide sum / max helpers for the single-workgroup ops (1024 threads = 16 waves)
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.0f;
    for (int i = 0; i < nw; i++) s += red[i];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = red[0];
    for (int i = 1; i < nw; i++) s = fmaxf(s, red[i]);
    return s;
}

// cpu.rs:99-117, one workgroup
__global__ __launch_bounds__(1024) void rmsnorm_kernel(float* o, const float* x, const float* w, int n) {
    __shared__ float red[16];
    float ss = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) ss = fmaf(x[i], x[i], ss);
    ss = block_sum(ss, red);
    const float v = rms_scale(ss, n);
    for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = w[i] * (v * x[i]);
}

// cpu.rs:74-97, one head
__global__ void apply_position_kernel(float* q, float* k, const float* pr, const float* pi, int head_size) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= head_size / 2) return;
    float c = pr[i], s = pi[i];
    float q0 = q[2 * i], q1 = q[2 * i + 1];
    q[2 * i] = q0 * c - q1 * s; q[2 * i + 1] = q0 * s + q1 * c;
    float k0 = k[2 * i], k1 = k[2 * i + 1];
    k[2 * i] = k0 * c - k1 * s; k[2 * i + 1] = k0 * s + k1 * c;
}

// ... for a run of consecutive heads (dim = heads x head_size floats of q and of k; the same table rows for every head: infer.rs:25-29's loop as one launch)
__global__ void apply_position_heads_kernel(float* q, float* k, const float* pr, const float* pi, int head_size, int dim) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= dim / 2) return;
    const int i = j % (head_size / 2);
    float c = pr[i], s = pi[i];
    float q0 = q[2 * j], q1 = q[2 * j + 1];
    q[2 * j] = q0 * c - q1 * s; q[2 * j + 1] = q0 * s + q1 * c;
    float k0 = k[2 * j], k1 = k[2 * j + 1];
    k[2 * j] = k0 * c - k1 * s; k[2 * j + 1] = k0 * s + k1 * c;
}

// cpu.rs:119-125, one workgroup
__global__ __launch_bounds__(1024) void softmax_kernel(float* x, int n) {
    __shared__ float red[16];
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) mx = fmaxf(mx, x[i]);
    mx = block_max(mx, red);
    float sum = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { float e = expf(x[i] - mx); x[i] = e; sum += e; }
    sum = block_sum(sum, red);
    for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] /= sum;
}

// Device::sample T == 0 (cpu.rs:163-167): max value, ties -> LAST index.  One workgroup.
// With `ctl` set it also advances the decode cursor (generate loop, mod.rs:190-203):
//   next = pos < n_forced ? forced[pos] : argmax; out[n_out++] = next; token = next; pos += 1
// and, when `emb` is given, gathers the next step's embedding row into x (infer.rs:13), so a
// chained decode step needs no separate gather launch.
struct ArgmaxParams {
    const float* logits; int n;
    int* result;          // optional device int
    Ctl* ctl;             // optional cursor to advance
    const int* forced; int* out; int out_cap;
    int* ring;            // optional: host-visible (pinned, mapped) copy of `out`, entry = token + 1, 0 = not produced yet
    const float* emb; float* x; int dim;   // optional next-token embedding gather
    unsigned* epoch;      // optional: the one-launch stage's epoch (layer_fused.hpp), advanced once per step
};

// what ends a chained decode step once the sampled index is known (mod.rs:187-201): the result
// word, the forced prompt token overriding the sample, the output list and the cursor
__device__ __forceinline__ int finish_step(const ArgmaxParams& p, int idx, int pos, int n_forced, int n_out, int forced_tok) {
    if (p.result) *p.result = idx;          // -1: the top-p sampler found no candidate
    if (p.epoch) *p.epoch = *p.epoch + 1u;
    int next = idx < 0 ? 0 : idx;
    if (p.ctl) {
        if (forced_tok >= 0) next = forced_tok;
        if (n_out < p.out_cap) {
            p.out[n_out] = next;
            // the host may be polling this word while the loop runs on (rama_decode_stream_poll): one system-scope store
            if (p.ring) __hip_atomic_store(p.ring + n_out, next + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        p.ctl->n_out = n_out + 1;
        p.ctl->token = next;
        p.ctl->pos = pos + 1;
    }
    return next;
}
// x = emb[next] for the following step (infer.rs:13); *s_next is a __shared__ word thread 0 wrote
__device__ __forceinline__ void gather_next_embedding(const ArgmaxParams& p, const int* s_next) {
    if (p.emb) {
        __syncthreads();
        const size_t base = (size_t)*s_next * p.dim;
        for (int i = threadIdx.x; i < p.dim; i += blockDim.x) p.x[i] = p.emb[base + i];
    }
}

__global__ __launch_bounds__(1024) void argmax_kernel(ArgmaxParams p) {
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    __shared__ int s_next;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // cursor reads go out first; their latency overlaps the logits sweep
    int pos = 0, n_forced = 0, n_out = 0, forced_tok = -1;
    if (p.ctl && tid == 0) {
        pos = p.ctl->pos; n_forced = p.ctl->n_forced; n_out = p.ctl->n_out;
        if (pos < n_forced) forced_tok = p.forced[pos];
    }
    float bv = -INFINITY; int bi = -1;
    // 16-byte loads when the view is aligned; every thread visits its indices in ascending
    // order, so "replace unless strictly smaller" keeps the LAST maximum (cpu.rs:165-167)
    const int n4 = (((uintptr_t)p.logits & 15) == 0) ? (p.n >> 2) : 0;
    const f4* l4 = reinterpret_cast<const f4*>(p.logits);
    // 8 loads in flight per thread: one workgroup sweeps 128 KB, and a dependent load per iteration
    // would cost a cache round trip each (this launch is pure latency)
    for (int i0 = tid; i0 < n4; i0 += 8 * 1024) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = l4[min(i0 + u * 1024, n4 - 1)];      // ([r4] clamped, not conditional: `i < n4 ? load : x` is a branch with
                                                                                  // s_waitcnt vmcnt(0) behind it -- the eight loads went out one by one)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u * 1024;
            if (i < n4) {
                if (!(bv > v[u].x)) { bv = v[u].x; bi = 4 * i; }
                if (!(bv > v[u].y)) { bv = v[u].y; bi = 4 * i + 1; }
                if (!(bv > v[u].z)) { bv = v[u].z; bi = 4 * i + 2; }
                if (!(bv > v[u].w)) { bv = v[u].w; bi = 4 * i + 3; }
            }
        }
    }
    for (int i = 4 * n4 + tid; i < p.n; i += 1024) {
        const float v = p.logits[i];
        if (!(bv > v)) { bv = v; bi = i; }
    }
    const float wm = wave_max(bv);
    const int wi = wave_max_i(bv == wm ? bi : -1);
    if (lane == 0) { s_v[wave] = wm; s_i[wave] = wi; }
    __syncthreads();
    if (tid == 0) {
        float v = s_v[0]; int idx = s_i[0];
        for (int w = 1; w < 16; w++) {
            const float ov = s_v[w]; const int oi = s_i[w];
            if (oi >= 0 && (idx < 0 || ov > v || (ov == v && oi > idx))) { v = ov; idx = oi; }
        }
        s_next = finish_step(p, idx, pos, n_forced, n_out, forced_tok);
    }
    gather_next_embedding(p, &s_next);
}

// ---------------------------------------------------------------- top-p sampling on the device
// Device::sample for temperature != 0 (cpu.rs:168-178 + sample_top_q, infer.rs:55-85), without the
// reference GPU path's 128 KB logits download per token (gpu.rs:153).  The ordering (softmax, cutoff,
// stable descending order) is topp_sort.hpp; below: the parameters and the generic pick --
//   topp_pick_kernel      cum += p in sorted order until cum > topp (sequential fp32, one thread,
//                         staged through LDS); r = u * cum; first i < last with r < cum_i, else last
// for candidate lists longer than one workgroup's LDS (topp_sort.hpp's topp_pick_scan_kernel otherwise).
struct ToppParams {
    const float* logits; int n;
    float temperature, topp, u;
    float* keys; int* vals;          // [n] the kept probabilities / their indices in the reference's order
    float* prefix;                   // [n] scratch: running sums of the sorted probabilities
    int* m;                          // number of candidates
    unsigned* err;                   // set to 1 when no probability exceeds the cutoff
};

constexpr int kToppChunk = 4096;
__global__ __launch_bounds__(1024) void topp_pick_kernel(ToppParams p, ArgmaxParams fin) {
    __shared__ float s_p[2][kToppChunk];
    __shared__ int s_last, s_next, s_pick;
    __shared__ float s_cum;
    const int tid = threadIdx.x;
    int pos = 0, n_forced = 0, n_out = 0, forced_tok = -1;
    if (fin.ctl && tid == 0) {
        pos = fin.ctl->pos; n_forced = fin.ctl->n_forced; n_out = fin.ctl->n_out;
        if (pos < n_forced) forced_tok = fin.forced[pos];
    }
    const int m = *p.m;
    if (tid == 0) { s_last = m > 0 ? m - 1 : 0; s_cum = 0.0f; s_pick = 0; }
    // Sequential running sum in sorted order (infer.rs:70-73): the fp32 rounding of cum_i depends
    // on every earlier add, so the chain cannot be split.  Wave 0 walks it 64 values at a time with
    // a lane ripple: lane i holds p_i, lane 0 is seeded with carry + p_0, and 63 identical
    // `v_add_f32_dpp s, s, p wave_shr:1` steps (lane i: s = s[i-1] + p_i; lane 0 has no source lane
    // and keeps its value) leave S_i in lane i -- one 4-cycle VALU op (+2 wait states) per element,
    // no LDS turn inside the chain.  All threads stage the next 4096-value chunk meanwhile (double
    // buffer) and afterwards copy the running sums out.  Padding zeros leave the sum unchanged.
    const int nchunks = (m + kToppChunk - 1) / kToppChunk;
    auto stage = [&](int c) {
        const int base = c * kToppChunk, len = min(kToppChunk, m - base), padded = (len + 63) & ~63;
        for (int i = tid; i < padded; i += 1024) s_p[c & 1][i] = i < len ? p.keys[base + i] : 0.0f;
    };
    if (nchunks > 0) stage(0);
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        const int base = c * kToppChunk, len = min(kToppChunk, m - base);
        if (c + 1 < nchunks) stage(c + 1);
        if (tid < 64) {
            float* q = s_p[c & 1];
            float carry = s_cum;
            const int nblk = (len + 63) >> 6;
            int hit = -1;
            float pv = q[tid];
            for (int blk = 0; blk < nblk; blk++) {
                const float pn = q[min(blk + 1, nblk - 1) * 64 + tid];      // next block's values: in flight during the ripple
                float sv = tid == 0 ? carry + pv : pv;
#pragma unroll
                for (int k = 0; k < 63; k++)
                    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(sv) : "v"(pv));
                q[blk * 64 + tid] = sv;
                carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sv), 63));
                const unsigned long long over = __ballot(sv > p.topp);
                if (over) {                       // sums are non-decreasing: the first set lane is the crossing
                    const int l0 = __ffsll((long long)over) - 1;
                    hit = blk * 64 + l0;
                    carry = __shfl(sv, l0);
                    break;
                }
                pv = pn;
            }
            if (tid == 0) {
                s_cum = carry;
                if (hit >= 0) { s_last = base + hit; s_pick = 1; }
            }
        }
        __syncthreads();
        // running sums of this chunk -> global (read again below); only indices < last matter
        for (int i = tid; i < len; i += 1024) p.prefix[base + i] = s_p[c & 1][i];
        if (s_pick) break;                                   // uniform
        __syncthreads();
    }
    __syncthreads();
    // r = u * cum; the first i < last whose running sum exceeds r wins, else `last` (infer.rs:75-84);
    // the running sums of that loop are exactly the ones stored above
    const int last = s_last;
    const float r = p.u * s_cum;
    // the running sums never decrease, so "first i < last with r < cum_i" = the number of i < last
    // with cum_i <= r: independent loads, no early exit
    int below = 0;
    for (int i = tid; i < last; i += 1024) below += !(r < p.prefix[i]);
    const int best = min(block_sum_i(below), last);
    if (tid == 0) {
        const int idx = m > 0 ? p.vals[best] : -1;
        s_next = finish_step(fin, idx, pos, n_forced, n_out, forced_tok);
    }
    gather_next_embedding(fin, &s_next);
}

// bit-exact twin of oracle_fill_synth (integer hash, Irwin-Hall(4), one multiply, one add)
__global__ void fill_synth_kernel(float* dst, size_t n, uint64_t base, float scale, float bias) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = (uint64_t)i + base;
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
        z ^= z >> 27; z *= 0x94D049BB133111EBULL;
        z ^= z >> 31;
        int sum = (int)(z & 0xFFFF) + (int)((z >> 16) & 0xFFFF) + (int)((z >> 32) & 0xFFFF) + (int)(z >> 48);
        float prod = (float)(sum - 131070) * scale;
        asm volatile("" : "+v"(prod));   // opaque: bias + prod must round twice like the CPU generator, never one FMA
        dst[i] = bias + prod;
    }
}

}  // namespace rama
