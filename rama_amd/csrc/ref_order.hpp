// ref_order.hpp -- the decode step in the REFERENCE'S OWN ARITHMETIC ORDER (opt-in:
// rama_set_tuning(ctx, "ref_order", 1)).
//
// The fast path (kernels.hpp) sums every dot product in a tree with fused multiply-adds; the
// reference's CPU backend sums 4 strided lanes sequentially with separate multiplies and adds
// (engine/src/device/cpu.rs:127-153) and carries ~1.4e-4 of rounding error in the llama2-7B logits
// (profiles/r02_parity_llama2_7b_200pos.json: oracle vs the fp64-accumulated network), so no
// implementation that sums differently can stay within 1e-4 of it over a long generation -- being
// closer to the exact result does not help.  This file reproduces the reference's rounding sequence
// operation by operation instead, so that the HIP path can be compared with it BIT FOR BIT:
//   matmul               cpu.rs:127-153   4 lane sums over k = j (mod 4), product and sum rounded
//                                         separately, (l0 + l1) + (l2 + l3)
//   rmsnorm              cpu.rs:99-117    sequential sum of squares, w * (v * x)
//   apply_position       cpu.rs:74-97     a*c - b*s with three roundings
//   multi_head_attention cpu.rs:23-52     sequential q.k per timestep, / sqrt(hs); softmax: max,
//                                         exp, sequential sum, divide; xb += att * v, t ascending
//   sinu / array_mult    cpu.rs:54-64     a * (1 / (1 + exp(-a))), then * hb2
// exp is glibc's expf (what Rust's f32::exp calls on Linux), restated below from the binary of this
// image's libm (2.35, the FMA ifunc variant every AVX2 host selects): table-driven, evaluated in
// double, with the same four fused multiply-adds.  Everything here is parallel only across
// independent outputs (rows, timesteps, head columns); every sum that the reference orders is one
// thread's sequential loop.  Speed is not the point (~8 ms per llama2-7B token).
#pragma once
#include "kernels.hpp"

namespace rama {

// every function below must round a*b + c twice: hipcc contracts by default
#define RAMA_NO_CONTRACT _Pragma("clang fp contract(off)")

// ---------------------------------------------------------------- glibc 2.35 expf, FMA variant
// sysdeps/ieee754/flt-32/e_expf.c + e_exp2f_data.c (N = 32), as compiled with -mfma:
//   kd = fma(InvLn2N, x, Shift); ki = bits(kd); kd -= Shift; r = fma(InvLn2N, x, -kd)
//   s = double(T[ki % 32] + (ki << 47)); z = fma(C0, r, C1); y = fma(C2, r, 1); y = fma(z, r*r, y)
//   return float(y * s)
__device__ __constant__ const unsigned long long kExp2fTab[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL,
};

// tab: the 32 table words -- kExp2fTab itself, or a workgroup's copy of it in LDS (exp_tab_fill).  A lookup in constant memory is a per-lane load
// that travels to L2 (and past it once a long context's cache rows have pushed the table out): ~0.2 us in front of EVERY exponential of a loop,
// 1.5 of the 19 us of the long-context softmax + values launch at position 1 900 ([r5]: an LDS read instead).
__device__ __forceinline__ float expf_glibc_tab(float x, const unsigned long long* tab) {
    const unsigned ux = __float_as_uint(x);
    const unsigned abstop = (ux >> 20) & 0x7ff;
    if (abstop > 0x42a) {                                   // |x| >= 88 or not finite
        if (ux == 0xff800000u) return 0.0f;                 // exp(-inf)
        if (abstop > 0x7f7) return x + x;                   // inf, nan
        if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);     // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;                // underflow
    }
    const double xd = (double)x;
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0, Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0, C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    double kd = __builtin_fma(InvLn2N, xd, Shift);
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd = __dsub_rn(kd, Shift);
    const double r = __builtin_fma(InvLn2N, xd, -kd);
    const unsigned long long t = tab[ki & 31] + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    const double z = __builtin_fma(C0, r, C1);
    const double r2 = __dmul_rn(r, r);
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = __dmul_rn(y, s);
    return (float)y;
}
__device__ __forceinline__ float expf_glibc(float x) { return expf_glibc_tab(x, kExp2fTab); }
// the table into LDS: threads 0..31 of the workgroup write it; a barrier of the caller's stands between this and the first expf_glibc_tab
__device__ __forceinline__ void exp_tab_fill(unsigned long long* s_tab) { if (threadIdx.x < 32) s_tab[threadIdx.x] = kExp2fTab[threadIdx.x]; }

__global__ void expf_glibc_kernel(float* o, const float* x, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = expf_glibc(x[i]);
}

// ---------------------------------------------------------------- cpu.rs:127-153 matmul, o_cols == 1
// One thread per output row: v_j += a[r][4i + j] * b[4i + j] for i ascending, four chains, then
// (v0 + v1) + (v2 + v3).  Up to 3 matrices per launch (blockIdx.y), all [rows, K].
// lane_reduce: the order of the final sum of the four chains, cpu.rs:148 `v.reduce_add()` -- wide::f32x4 leaves it to the build's target features (an SSE3
// hadd pair, the SSE2 movehl + shuffle idiom, or an array sum), the oracle has the same switch (oracle_set_lane_reduce), "lane_reduce" sets it
enum { LANES_PAIRWISE = 0, LANES_STRIDED = 1, LANES_SEQUENTIAL = 2 };
struct RefMatParams { const float* w[3]; float* o[3]; const float* x; int K, rows; int lane_reduce; };

__global__ __launch_bounds__(64) void matvec_ref_kernel(RefMatParams p) {
    RAMA_NO_CONTRACT
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= p.rows) return;
    const int m = blockIdx.y;
    const float* W = (m == 0 ? p.w[0] : (m == 1 ? p.w[1] : p.w[2])) + (size_t)r * p.K;
    float* o = m == 0 ? p.o[0] : (m == 1 ? p.o[1] : p.o[2]);
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
    const bool vec = (((uintptr_t)W | (uintptr_t)p.x) & 15) == 0;
    if (vec) {
        const f4* w4 = reinterpret_cast<const f4*>(W);
        const f4* x4 = reinterpret_cast<const f4*>(p.x);
        const int n4 = p.K >> 2;
        int i = 0;
        for (; i + 8 <= n4; i += 8) {
            f4 a[8];
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] = w4[i + u];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const f4 b = x4[i + u];
                v0 = v0 + a[u].x * b.x; v1 = v1 + a[u].y * b.y; v2 = v2 + a[u].z * b.z; v3 = v3 + a[u].w * b.w;
            }
        }
        for (; i < n4; i++) {
            const f4 a = w4[i], b = x4[i];
            v0 = v0 + a.x * b.x; v1 = v1 + a.y * b.y; v2 = v2 + a.z * b.z; v3 = v3 + a.w * b.w;
        }
    } else {
        for (int k = 0; k < p.K; k += 4) {
            v0 = v0 + W[k] * p.x[k]; v1 = v1 + W[k + 1] * p.x[k + 1];
            v2 = v2 + W[k + 2] * p.x[k + 2]; v3 = v3 + W[k + 3] * p.x[k + 3];
        }
    }
    o[r] = p.lane_reduce == LANES_STRIDED ? (v0 + v2) + (v1 + v3) : (p.lane_reduce == LANES_SEQUENTIAL ? ((v0 + v1) + v2) + v3 : (v0 + v1) + (v2 + v3));
}

// the trait's o_cols > 1 product (cpu.rs:137-151 as written: output idx = r * o_cols + c, b strided by o_cols; forward() never calls it): one thread per
// output, the same four chains and the same final order
__global__ __launch_bounds__(256) void matmul_cols_ref_kernel(float* o, const float* a, const float* b, int width, int o_rows, int o_cols, int lane_reduce) {
    RAMA_NO_CONTRACT
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)o_rows * (size_t)o_cols) return;
    const size_t r = idx / (size_t)o_cols, cc = idx % (size_t)o_cols;
    const float* ar = a + r * (size_t)width;
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
    for (int k = 0; k < width; k += 4) {
        v0 = v0 + ar[k] * b[(size_t)k * o_cols + cc];
        v1 = v1 + ar[k + 1] * b[(size_t)(k + 1) * o_cols + cc];
        v2 = v2 + ar[k + 2] * b[(size_t)(k + 2) * o_cols + cc];
        v3 = v3 + ar[k + 3] * b[(size_t)(k + 3) * o_cols + cc];
    }
    o[idx] = lane_reduce == LANES_STRIDED ? (v0 + v2) + (v1 + v3) : (lane_reduce == LANES_SEQUENTIAL ? ((v0 + v1) + v2) + v3 : (v0 + v1) + (v2 + v3));
}

// ---------------------------------------------------------------- cpu.rs:99-117 rmsnorm
__global__ __launch_bounds__(1024) void rmsnorm_ref_kernel(float* o, const float* x, const float* w, int n) {
    RAMA_NO_CONTRACT
    extern __shared__ __attribute__((aligned(16))) float s_x[];
    __shared__ float s_vv;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s_x[i] = x[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        float ss = 0.0f;
        for (int i = 0; i < n; i++) ss = ss + s_x[i] * s_x[i];
        s_vv = 1.0f / sqrtf(ss / (float)n + 1e-5f);
    }
    __syncthreads();
    const float v = s_vv;
    for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = w[i] * (v * s_x[i]);
}

// ---------------------------------------------------------------- cpu.rs:74-97 apply_position, all heads
// + the two cache appends of infer.rs:31-33 when kc / vc are given
__global__ void rope_ref_kernel(float* q, float* k, const float* v, const float* pr, const float* pi, int dim, int head_size,
                                float* kc_row, float* vc_row) {
    RAMA_NO_CONTRACT
    const int j = blockIdx.x * blockDim.x + threadIdx.x;      // pair index over the whole vector
    if (j >= dim / 2) return;
    const int i = j % (head_size / 2);
    const float fcr = pr[i], fci = pi[i];
    const float q0 = q[2 * j], q1 = q[2 * j + 1];
    const float a0 = q0 * fcr - q1 * fci, a1 = q0 * fci + q1 * fcr;
    q[2 * j] = a0; q[2 * j + 1] = a1;
    const float k0 = k[2 * j], k1 = k[2 * j + 1];
    const float b0 = k0 * fcr - k1 * fci, b1 = k0 * fci + k1 * fcr;
    k[2 * j] = b0; k[2 * j + 1] = b1;
    if (kc_row) { kc_row[2 * j] = b0; kc_row[2 * j + 1] = b1; }
    if (vc_row) { vc_row[2 * j] = v[2 * j]; vc_row[2 * j + 1] = v[2 * j + 1]; }
}

// the same with the position taken from the device cursor (fused path): table row pos, cache row pos
__global__ void rope_ref_cursor_kernel(float* q, float* k, const float* v, const float* fr, const float* fi, int dim, int head_size,
                                       float* kc, float* vc, const Ctl* ctl) {
    RAMA_NO_CONTRACT
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= dim / 2) return;
    const int pos = ctl->pos;
    const int i = j % (head_size / 2);
    const float fcr = fr[(size_t)pos * (head_size / 2) + i], fci = fi[(size_t)pos * (head_size / 2) + i];
    const float q0 = q[2 * j], q1 = q[2 * j + 1];
    const float a0 = q0 * fcr - q1 * fci, a1 = q0 * fci + q1 * fcr;
    q[2 * j] = a0; q[2 * j + 1] = a1;
    const float k0 = k[2 * j], k1 = k[2 * j + 1];
    const float b0 = k0 * fcr - k1 * fci, b1 = k0 * fci + k1 * fcr;
    k[2 * j] = b0; k[2 * j + 1] = b1;
    kc[(size_t)pos * dim + 2 * j] = b0; kc[(size_t)pos * dim + 2 * j + 1] = b1;           // infer.rs:32
    vc[(size_t)pos * dim + 2 * j] = v[2 * j]; vc[(size_t)pos * dim + 2 * j + 1] = v[2 * j + 1];   // infer.rs:33
}

// ---------------------------------------------------------------- cpu.rs:23-52 multi_head_attention
// One workgroup per head; att scratch in LDS (pos + 1 floats) and, when `att` is given, in the
// reference's att[h * seq_len + t] buffer too.
struct RefAttnParams {
    const float* q; const float* kc; const float* vc;     // kc / vc: this layer's slabs [seq, dim]
    float* att; float* xb;
    const Ctl* ctl; int pos_val;
    int dim, head_size, seq_len;
    float* sc = nullptr;      // spread attention (chain.hpp): where the scores launch leaves the raw scores ([n_heads, seq_len]; == att for the three-launch form)
    // a grid of (heads, tokens) -- the parity-mode prefill pass, chain.hpp: token y sits at position pos + y, its q / xb
    // rows are y * tok_stride floats further on, its att rows y * att_stride.  (0, 0 and gridDim.y = 1: one token.)
    int tok_stride, att_stride;
    // ... or tokens of independent sequences (rama_decode_batch in parity mode): token y reads its position and its cache
    // bases from seqs[y] (+ layer_off floats for this layer); pos / kc / vc above are then unused
    const SeqSlot* seqs; size_t layer_off;
};

__global__ __launch_bounds__(1024) void attention_ref_kernel(RefAttnParams p) {
    RAMA_NO_CONTRACT
    extern __shared__ __attribute__((aligned(16))) float s_att[];
    __shared__ float red[16];
    __shared__ float s_sum;
    const int h = blockIdx.x, tid = threadIdx.x;
    const int pos = p.ctl ? p.ctl->pos : p.pos_val;
    const int hs = p.head_size;
    const float* q = p.q + (size_t)h * hs;
    const float scale_div = sqrtf((float)hs);
    for (int t = tid; t <= pos; t += blockDim.x) {
        const float* k = p.kc + (size_t)t * p.dim + (size_t)h * hs;
        float acc = 0.0f;
        for (int i = 0; i < hs; i++) acc = acc + q[i] * k[i];
        s_att[t] = acc / scale_div;
    }
    __syncthreads();
    // softmax_num (cpu.rs:187-192): max, exp(a - max), sum, divide
    float mx = -INFINITY;
    for (int t = tid; t <= pos; t += blockDim.x) mx = fmaxf(mx, s_att[t]);
    mx = block_max(mx, red);
    __syncthreads();
    for (int t = tid; t <= pos; t += blockDim.x) s_att[t] = expf_glibc(s_att[t] - mx);
    __syncthreads();
    if (tid == 0) {
        float sum = 0.0f;
        for (int t = 0; t <= pos; t++) sum = sum + s_att[t];
        s_sum = sum;
    }
    __syncthreads();
    const float sum = s_sum;
    for (int t = tid; t <= pos; t += blockDim.x) {
        const float a = s_att[t] / sum;
        s_att[t] = a;
        if (p.att) p.att[(size_t)h * p.seq_len + t] = a;
    }
    __syncthreads();
    for (int i = tid; i < hs; i += blockDim.x) {
        float acc = 0.0f;
        for (int t = 0; t <= pos; t++) acc = acc + s_att[t] * p.vc[(size_t)t * p.dim + (size_t)h * hs + i];
        p.xb[(size_t)h * hs + i] = acc;
    }
}

// ---------------------------------------------------------------- cpu.rs:54-64 sinu, then array_mult
__global__ void sinu_ref_kernel(float* o, size_t n) {
    RAMA_NO_CONTRACT
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = o[i];
        o[i] = a * (1.0f / (1.0f + expf_glibc(-a)));
    }
}

// ... followed by Device::array_mult on the same vector (infer.rs:44-45), one launch: the same two roundings per element (cpu.rs:56, :59-64)
__global__ void sinu_mult_ref_kernel(float* o, const float* s, size_t n) {
    RAMA_NO_CONTRACT
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = o[i];
        const float g = a * (1.0f / (1.0f + expf_glibc(-a)));
        o[i] = g * s[i];
    }
}

// cpu.rs:119-125 Device::softmax (whole view)
__global__ __launch_bounds__(1024) void softmax_ref_kernel(float* x, int n) {
    RAMA_NO_CONTRACT
    __shared__ float red[16];
    __shared__ float s_sum;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) mx = fmaxf(mx, x[i]);
    mx = block_max(mx, red);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = expf_glibc(x[i] - mx);
    __syncthreads();
    if (threadIdx.x == 0) {
        float sum = 0.0f;
        for (int i = 0; i < n; i++) sum = sum + x[i];
        s_sum = sum;
    }
    __syncthreads();
    const float sum = s_sum;
    for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = x[i] / sum;
}

}  // namespace rama
