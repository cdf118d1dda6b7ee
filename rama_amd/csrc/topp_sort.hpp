// Ordering step of the device top-p sampler (Device::sample, cpu.rs:168-178 + sample_top_q,
// infer.rs:55-85): any vocabulary size, no library sort.
// [r4] n <= 32768: block sort, then topp_rank_pairs_kernel + topp_rank_scatter_kernel (the ranking spread over the chip as (block, block)
// pairs: 5.8 + 2.6 us where topp_rank_kernel's one-workgroup-holds-all-blocks search took 23.1 us; "topp_pairs" = 0 selects the latter).
//
//   topp_blocksort_kernel   one workgroup per 2048 logits.  Every workgroup repeats the softmax
//                           statistics over the whole vector (max, then sum of exp in the same
//                           per-thread order everywhere, so all of them hold the same
//                           bits), keeps the entries of its slice with p > (1 - topp)/(n - 1)
//                           (infer.rs:56-63) and sorts them in LDS.
//   topp_rank_kernel        a kept entry's place in the whole order = its place in its block + the
//                           number of entries of every other block that precede it (binary searches in
//                           LDS, all blocks' probes of one step in flight together); scatters (p, index).
//
// The sort key is 64 bits: the probability's bit pattern (positive floats order like integers) above
// ~index.  Descending key order = descending probability with equal probabilities in ascending index
// order, which is what the reference's stable sort of the index-ordered candidate list produces
// (infer.rs:64); the keys are distinct, so no sorting network needs to be stable for it.
// topp_pick_scan_kernel (below) then forms the running sums of the sorted probabilities.
#pragma once
#include "kernels.hpp"

namespace rama {

// phase time stamps for tools/topp_bench.hip only (100 MHz counter, thread 0 of workgroup 0)
#ifdef RAMA_TOPP_STAMPS
__device__ unsigned long long g_topp_stamps[64];
#define TOPP_STAMP(id) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_topp_stamps[id] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TOPP_STAMP(id) do { } while (0)
#endif

constexpr int kToppBlock = 2048;        // logits per sorting workgroup
constexpr int kToppMaxBlocks = 16;      // => n <= 32768 on the LDS-rank path; larger vocabularies rank through global memory (topp_rank_global_kernel)

struct ToppSortParams {
    const float* logits; int n;
    float temperature, topp;
    float* bp; int* bi;                 // [nblk * 2048] every block's kept probabilities (descending) and their indices
    int* bcount;                        // [nblk] kept entries per block
    float* keys; int* vals;             // [n] out: probabilities / indices in the order of the whole sort
    int* m;                             // out: number of kept entries
    unsigned* err;                      // set to 1 when nothing is kept
    int nblk;
    int* racc;                          // [nblk * 2048] pair-wise ranking (topp_rank_pairs_kernel): entries of OTHER blocks that precede block b's entry s; zeroed by the block sort
    // small-block path (topp_*_bs_kernel): count and probability mass in one 64-bit accumulator, see topp_fixed()
    unsigned long long* rk;             // [nblk * BS] entries of other blocks that precede the entry (bits 48..63) and their mass (bits 0..47); zeroed by the block sort
    unsigned long long* bm;             // [nblk * BS] mass of a block's entries up to and including each one (the block sort's output)
    float* approx;                      // [n] out: the mass in front of every entry of the whole order (what topp_pick_dist_kernel predicts binades from)
    unsigned* epoch;                    // the pick launch's hand-off tag (topp_pick.hpp), advanced by the scatter launch
};
// Probability mass as a 48-bit fixed-point number, unit 2^-47, truncated: integer sums are exact in any order, so the mass in front of an
// entry -- gathered from 32 blocks by atomics -- is the same number however the adds arrive, and differs from the real-number sum by less
// than 2^-47 per entry.  p <= 1 (the softmax sum contains the maximum's exp(0) = 1), so a sum over the whole list stays below 2^48.
constexpr int kMassBits = 48;
constexpr unsigned long long kMassMask = (1ull << kMassBits) - 1ull;
__device__ __forceinline__ unsigned long long topp_fixed(float p) {
    const unsigned b = __float_as_uint(p);
    const int e = (int)(b >> 23);                                  // p = mant 2^(e - 150): p 2^47 = mant 2^(e - 103)
    const unsigned long long mant = (unsigned long long)((b & 0x7FFFFFu) | 0x800000u);
    const int sh = e - 103;
    return e == 0 ? 0ull : (sh >= 0 ? mant << min(sh, 24) : (sh > -24 ? mant >> (-sh) : 0ull));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {                 // lanes without a source get 0
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROW_MASK, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xF, true);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_scan_u64(unsigned long long v) {           // inclusive, lane order
    v += dpp_u64<0x111, 0xF>(v); v += dpp_u64<0x112, 0xF>(v); v += dpp_u64<0x114, 0xF>(v); v += dpp_u64<0x118, 0xF>(v);
    v += dpp_u64<0x142, 0xA>(v); v += dpp_u64<0x143, 0xC>(v);
    return v;
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, mask), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), mask);
    return ((unsigned long long)hi << 32) | lo;
}

// BIG: n > 32768 -- the softmax statistics loop over the logits instead of holding them in registers (same
// per-thread order: ascending i = tid + 1024 k, then the wave tree, then the 16-wave tree)
template <bool BIG>
__global__ __launch_bounds__(1024) void topp_blocksort_kernel(ToppSortParams p) {
    __shared__ float s_r[16];
    __shared__ unsigned long long s_k[2][kToppBlock];
    __shared__ int s_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool scale = p.temperature < 1.0f;                       // cpu.rs:170-172: T > 1 has no effect
    TOPP_STAMP(0);
    // thread t owns logits t, t + 1024, ... (n <= 32768: at most 32 of them, all requested at once)
    float x[BIG ? 1 : 32];
    float mx = -INFINITY;
    if (!BIG) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            const int i = tid + 1024 * k;
            const float v = i < p.n ? p.logits[i] : -INFINITY;
            x[BIG ? 0 : k] = scale ? v / p.temperature : v;
        }
#pragma unroll
        for (int k = 0; k < 32; k++) mx = fmaxf(mx, x[BIG ? 0 : k]);
    } else {
        for (int i = tid; i < p.n; i += 1024) { const float v = p.logits[i]; mx = fmaxf(mx, scale ? v / p.temperature : v); }
    }
    if (tid == 0) s_n = 0;
    mx = wave_max(mx);
    if (lane == 0) s_r[wave] = mx;
    __syncthreads();
    mx = s_r[0];
#pragma unroll
    for (int w = 1; w < 16; w++) mx = fmaxf(mx, s_r[w]);
    __syncthreads();
    TOPP_STAMP(1);
    // sum of exp: per thread in ascending index order, wave tree, then the 16-wave tree
    float sum = 0.0f;
    if (!BIG) {
#pragma unroll
        for (int k = 0; k < 32; k++)
            if (tid + 1024 * k < p.n) sum += expf(x[BIG ? 0 : k] - mx);
    } else {
        for (int i = tid; i < p.n; i += 1024) { const float v = p.logits[i]; sum += expf((scale ? v / p.temperature : v) - mx); }
    }
    sum = wave_sum(sum);
    if (lane == 0) s_r[wave] = sum;
    __syncthreads();
    {
        float t[16];
#pragma unroll
        for (int w = 0; w < 16; w++) t[w] = s_r[w];
#pragma unroll
        for (int n = 16; n > 1; n >>= 1)
#pragma unroll
            for (int w = 0; w < n / 2; w++) t[w] = t[2 * w] + t[2 * w + 1];
        sum = t[0];
    }
    TOPP_STAMP(2);
    const float cutoff = (1.0f - p.topp) / (float)(p.n - 1);      // infer.rs:56
    // this block's slice: entries 2048 b + tid and 2048 b + 1024 + tid; kept ones are appended to
    // the LDS list in any order (the key carries the index)
    const int base = blockIdx.x * kToppBlock;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int i = base + h * 1024 + tid;
        if (i < p.n) {
            const float v = p.logits[i];
            const float pr = expf((scale ? v / p.temperature : v) - mx) / sum;
            if (pr > cutoff) {
                const int slot = atomicAdd(&s_n, 1);
                s_k[0][slot] = ((unsigned long long)__float_as_uint(pr) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
            }
        }
    }
    __syncthreads();
    const int cnt = s_n;
    TOPP_STAMP(3);
    if (p.racc) { p.racc[(size_t)base + tid] = 0; p.racc[(size_t)base + 1024 + tid] = 0; }      // the pair-wise ranking adds into these
    if (tid == 0) p.bcount[blockIdx.x] = cnt;
    if (cnt == 0) return;                                          // uniform
    int P = 2;
    while (P < cnt) P <<= 1;                                       // uniform
    const int H = P >> 1;
    // bitonic network over P keys, descending.  Thread t < P/2 holds elements t and t + P/2 in
    // registers; a stage's partner is in the same thread (j = P/2), the same wave (j < 64: lane
    // exchange, no barrier) or another wave (LDS, two buffers in turn: one barrier per stage).  The
    // network is bound by instruction issue (~30 per stage and wave, 16 waves): DPP moves instead of
    // the lane permutes changed nothing (20 vs 18 us at P = 2048).
    // (8 keys per thread on 4 waves was tried: 30 of 66 stages become register-only, but the chain
    // of dependent steps then runs on a quarter of the issue slots -- 32 us instead of 18.)
    unsigned long long A = tid < H && tid < cnt ? s_k[0][tid] : 0ull;             // zero padding sorts last
    unsigned long long B = tid < H && tid + H < cnt ? s_k[0][tid + H] : 0ull;
    int buf = 1;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j == H) {                                          // only in the last merge: always descending
                if (A < B) { const unsigned long long t = A; A = B; B = t; }
                continue;
            }
            unsigned long long oa, ob;
            if (j < 64) {
                oa = shfl_xor_u64(A, j); ob = shfl_xor_u64(B, j);
            } else {
                if (tid < H) { s_k[buf][tid] = A; s_k[buf][tid + H] = B; }
                __syncthreads();
                oa = tid < H ? s_k[buf][tid ^ j] : 0ull;
                ob = tid < H ? s_k[buf][(tid ^ j) + H] : 0ull;
                buf ^= 1;
            }
            const bool lower = (tid & j) == 0;
            const bool descA = (tid & k) == 0, descB = ((tid + H) & k) == 0;
            A = (lower == descA) ? (A > oa ? A : oa) : (A < oa ? A : oa);
            B = (lower == descB) ? (B > ob ? B : ob) : (B < ob ? B : ob);
        }
    }
    TOPP_STAMP(4);
    if (tid < H) {
        if (tid < cnt) {
            p.bp[(size_t)base + tid] = __uint_as_float((unsigned)(A >> 32));
            p.bi[(size_t)base + tid] = (int)(0xFFFFFFFFu - (unsigned)(A & 0xFFFFFFFFull));
        }
        if (tid + H < cnt) {
            p.bp[(size_t)base + tid + H] = __uint_as_float((unsigned)(B >> 32));
            p.bi[(size_t)base + tid + H] = (int)(0xFFFFFFFFu - (unsigned)(B & 0xFFFFFFFFull));
        }
    }
}

// The value of lane (l ^ J) for a constant J < 64 without the LDS crossbar: DPP moves inside a row of 16 lanes (two for J = 4: half
// mirror i -> 7 - i, then reversed quads i -> i ^ 3), gfx950's row / half swaps across rows.  ~8 cycles each where ds_bpermute is a
// 100-cycle round trip -- what a 4- or 8-wave sorting workgroup (one or two waves per SIMD, nothing to hide latency behind) is bound by.
template <int J>
__device__ __forceinline__ unsigned lane_xor_u32(unsigned v) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "lane_xor_u32: J");
    if constexpr (J == 1) return (unsigned)dpp_movi<0xB1>((int)v);                         // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return (unsigned)dpp_movi<0x4E>((int)v);                    // quad_perm [2,3,0,1]
    else if constexpr (J == 4) return (unsigned)dpp_movi<0x1B>(dpp_movi<0x141>((int)v));   // row_half_mirror, quad_perm [3,2,1,0]
    else if constexpr (J == 8) return (unsigned)dpp_movi<0x128>((int)v);                   // row_ror:8
    else if constexpr (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);               // r[0] = rows {0,0,2,2}, r[1] = rows {1,1,3,3}
        return (threadIdx.x & 16) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);               // r[0] = the lower half twice, r[1] = the upper half
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
}
// ---------------------------------------------------------------- [r4] small blocks: statistics once, 1024-entry sorts
// topp_blocksort_kernel is 16 workgroups of 16 waves: each repeats the softmax statistics over all n logits (7.6 us) and runs a
// 66-stage bitonic network with ds_bpermute exchanges (18 us).  With the ranking spread over the chip (topp_rank_pairs_bs_kernel)
// smaller sorted blocks cost little, so:
//   topp_stats_kernel          one workgroup per 1024 logits: its maximum and the sum of exp(x - that maximum)
//   topp_blocksort_bs_kernel   BS = 1024 (or 512) entries per workgroup, one key per thread: folds the partial statistics (the same
//                              order in every workgroup: the same bits), keeps and sorts its slice -- 55 stages, 45 of them inside a
//                              wave by DPP moves / lane swaps -- and leaves the block's running mass
struct ToppStats { float mx, sum; };
__global__ __launch_bounds__(1024) void topp_stats_kernel(ToppSortParams p, ToppStats* st) {
    __shared__ float s_r[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool scale = p.temperature < 1.0f;                       // cpu.rs:170-172: T > 1 has no effect
    const int i = blockIdx.x * 1024 + tid;
    const float v = i < p.n ? p.logits[i] : -INFINITY;
    const float x = scale ? v / p.temperature : v;
    float mx = wave_max(x);
    if (lane == 0) s_r[wave] = mx;
    __syncthreads();
    mx = s_r[0];
#pragma unroll
    for (int w = 1; w < 16; w++) mx = fmaxf(mx, s_r[w]);
    __syncthreads();
    float sum = wave_sum((i < p.n && mx != -INFINITY) ? expf(x - mx) : 0.0f);      // (a chunk of -inf logits -- a masked vocabulary -- has no mass, not exp(nan))
    if (lane == 0) s_r[wave] = sum;
    __syncthreads();
    if (tid == 0) {
        float t[16];
#pragma unroll
        for (int w = 0; w < 16; w++) t[w] = s_r[w];
#pragma unroll
        for (int n = 16; n > 1; n >>= 1)
#pragma unroll
            for (int w = 0; w < n / 2; w++) t[w] = t[2 * w] + t[2 * w + 1];
        st[blockIdx.x] = ToppStats{mx, t[0]};
    }
}

// BS entries per workgroup, ONE key per thread (16 waves at BS = 1024, four per SIMD): the bitonic network on 64-bit keys is bound by
// instruction issue -- a stage is a short dependent chain (exchange, 64-bit compare, two selects) -- and four waves per SIMD
// interleave better than two (two keys per thread on BS / 2 threads: 5.9 us for the network of 1 024 keys, sort launch 9.2 us;
// this way 8.2 us, although ten stages instead of nine cross waves through LDS: two buffers in turn, one barrier per stage).
// The 45 in-wave stages (j < 64) are lane_stage<J>.  Then the block's running mass in sorted order as fixed-point integers.
template <int J>
__device__ __forceinline__ void lane_stage(unsigned long long& A, bool desc) {
    const unsigned long long oa = ((unsigned long long)lane_xor_u32<J>((unsigned)(A >> 32)) << 32) | lane_xor_u32<J>((unsigned)A);
    const bool lower = (threadIdx.x & J) == 0;
    A = ((oa > A) == (lower == desc)) ? oa : A;
}
template <int BS>
__global__ __launch_bounds__(BS) void topp_blocksort_bs_kernel(ToppSortParams p, const ToppStats* st, int nstat) {
    __shared__ unsigned long long s_k[2][BS];
    __shared__ unsigned long long s_w[BS / 64];
    __shared__ int s_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool scale = p.temperature < 1.0f;
    if (tid == 0) s_n = 0;
    const int base = blockIdx.x * BS, i = base + tid;
    const float lg = p.logits[min(i, p.n - 1)];                   // (requested with the statistics)
    const ToppStats part = st[min(lane, nstat - 1)];
    const float mx = wave_max(part.mx);                            // (lanes behind the last partial repeat it: the maximum does not mind)
    const float sum = wave_sum((lane < nstat && part.mx != -INFINITY) ? part.sum * expf(part.mx - mx) : 0.0f);
    const float cutoff = (1.0f - p.topp) / (float)(p.n - 1);      // infer.rs:56
    __syncthreads();
    {
        const float pr = expf((scale ? lg / p.temperature : lg) - mx) / sum;
        if (i < p.n && pr > cutoff) {
            const int slot = atomicAdd(&s_n, 1);
            s_k[0][slot] = ((unsigned long long)__float_as_uint(pr) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
    }
    __syncthreads();
    const int cnt = s_n;
    p.rk[(size_t)base + tid] = 0ull;
    if (tid == 0) p.bcount[blockIdx.x] = cnt;
    if (cnt == 0) return;                                          // uniform
    int P = 2;
    while (P < cnt) P <<= 1;                                       // uniform
    unsigned long long A = tid < cnt ? s_k[0][tid] : 0ull;         // zero padding sorts last
    int buf = 1;
    for (int k = 2; k <= P; k <<= 1) {
        const bool desc = (tid & k) == 0;                          // (the last merge, k = P: every thread below P descending)
        int j = k >> 1;
        for (; j >= 64; j >>= 1) {
            s_k[buf][tid] = A;
            __syncthreads();
            const unsigned long long oa = s_k[buf][tid ^ j];
            buf ^= 1;
            const bool lower = (tid & j) == 0;
            A = ((oa > A) == (lower == desc)) ? oa : A;
        }
        if (j >= 32) lane_stage<32>(A, desc);
        if (j >= 16) lane_stage<16>(A, desc);
        if (j >= 8) lane_stage<8>(A, desc);
        if (j >= 4) lane_stage<4>(A, desc);
        if (j >= 2) lane_stage<2>(A, desc);
        if (j >= 1) lane_stage<1>(A, desc);
    }
    // (threads at and above P took part with zeros among themselves: tid ^ j stays on their side of P for every j < P)
    const float pr = __uint_as_float((unsigned)(A >> 32));
    if (tid < cnt) {
        p.bp[(size_t)base + tid] = pr;
        p.bi[(size_t)base + tid] = (int)(0xFFFFFFFFu - (unsigned)(A & 0xFFFFFFFFull));
    }
    // the block's running mass in sorted order (fixed point: exact integer sums)
    unsigned long long run = wave_scan_u64(tid < cnt ? topp_fixed(pr) : 0ull);
    if (lane == 63) s_w[wave] = run;
    __syncthreads();
    for (int w = 0; w < wave; w++) run += s_w[w];
    if (tid < cnt) p.bm[(size_t)base + tid] = run;
}

// the pair-wise ranking for blocks of BS entries: a workgroup of BS / 2 threads takes block b against OB other blocks at once.  Besides the
// COUNT of block o's entries that precede an entry it looks up their MASS (the block's running mass at that place) and adds both to the
// entry's accumulator in one 64-bit integer atomic: count << 48 | mass.
template <int BS, int OB>
__global__ __launch_bounds__(BS / 2) void topp_rank_pairs_bs_kernel(ToppSortParams p) {
    constexpr int NT = BS / 2;
    __shared__ unsigned s_o[OB * BS];
    __shared__ unsigned long long s_m[OB * BS];
    const int tid = threadIdx.x;
    const int b = blockIdx.x, o0 = blockIdx.y * OB;
    const int cb = p.bcount[b];
    if (cb == 0) return;                                           // uniform
    int co[OB], most = 0;
#pragma unroll
    for (int q = 0; q < OB; q++) { const int o = o0 + q; co[q] = (o < p.nblk && o != b) ? p.bcount[o] : 0; most = max(most, co[q]); }
    if (most == 0) return;                                         // uniform
    // every slot of bp / bm exists (nblk x BS): all loads go out unconditionally and what lies behind a block's count is zeroed
    // afterwards -- `cond ? load : 0` is a branch with s_waitcnt vmcnt(0) behind it, one cache round trip per load instruction
    const unsigned k0r = __float_as_uint(p.bp[(size_t)b * BS + tid]), k1r = __float_as_uint(p.bp[(size_t)b * BS + NT + tid]);
    {
        unsigned v[2 * OB];
        unsigned long long f[2 * OB];
#pragma unroll
        for (int q = 0; q < OB; q++) {
            const size_t ob = (size_t)min(o0 + q, p.nblk - 1) * BS;
            v[2 * q] = __float_as_uint(p.bp[ob + tid]); v[2 * q + 1] = __float_as_uint(p.bp[ob + NT + tid]);
            f[2 * q] = p.bm[ob + tid]; f[2 * q + 1] = p.bm[ob + NT + tid];
        }
#pragma unroll
        for (int q = 0; q < OB; q++) {
            s_o[q * BS + tid] = tid < co[q] ? v[2 * q] : 0u; s_o[q * BS + NT + tid] = tid + NT < co[q] ? v[2 * q + 1] : 0u;
            s_m[q * BS + tid] = tid < co[q] ? f[2 * q] : 0ull; s_m[q * BS + NT + tid] = tid + NT < co[q] ? f[2 * q + 1] : 0ull;
        }
    }
    const unsigned k0 = tid < cb ? k0r : 0u, k1 = tid + NT < cb ? k1r : 0u;
    __syncthreads();
    int top = 1;
    while (top <= most) top <<= 1;                                 // uniform
    int p0[OB], p1[OB];
#pragma unroll
    for (int q = 0; q < OB; q++) { p0[q] = 0; p1[q] = 0; }
    for (int step = top >> 1; step >= 1; step >>= 1) {
        unsigned q0[OB], q1[OB];
#pragma unroll
        for (int q = 0; q < OB; q++) { q0[q] = s_o[q * BS + min(p0[q] + step - 1, BS - 1)]; q1[q] = s_o[q * BS + min(p1[q] + step - 1, BS - 1)]; }
#pragma unroll
        for (int q = 0; q < OB; q++) {
            const bool before = o0 + q < b;                        // equal probabilities: the earlier block's entry comes first
            const bool pr0 = before ? q0[q] >= k0 : q0[q] > k0, pr1 = before ? q1[q] >= k1 : q1[q] > k1;
            p0[q] += (p0[q] + step <= co[q] && pr0) ? step : 0;
            p1[q] += (p1[q] + step <= co[q] && pr1) ? step : 0;
        }
    }
    unsigned long long a0 = 0, a1 = 0;
#pragma unroll
    for (int q = 0; q < OB; q++) {
        a0 += ((unsigned long long)p0[q] << kMassBits) + (p0[q] > 0 ? s_m[q * BS + p0[q] - 1] : 0ull);
        a1 += ((unsigned long long)p1[q] << kMassBits) + (p1[q] > 0 ? s_m[q * BS + p1[q] - 1] : 0ull);
    }
    if (tid < cb && a0) atomicAdd(&p.rk[(size_t)b * BS + tid], a0);
    if (tid + NT < cb && a1) atomicAdd(&p.rk[(size_t)b * BS + NT + tid], a1);
}
// place = own place in the block + the count; the mass in front of the entry = the accumulated mass + the own block's mass before it
template <int BS>
__global__ __launch_bounds__(1024) void topp_rank_scatter_bs_kernel(ToppSortParams p) {
    const int g = blockIdx.x * 1024 + threadIdx.x;
    const int b = g / BS, s = g % BS;
    if (g == 0) {
        if (p.epoch) *p.epoch = *p.epoch + 1u;
        int total = 0;
        for (int o = 0; o < p.nblk; o++) total += p.bcount[o];
        *p.m = total;
        if (total == 0 && p.err) *p.err = 1u;
    }
    if (b >= p.nblk) return;
    // (all five loads together: the count, the accumulator, the mass before the entry -- slot s - 1, clamped -- and the entry itself)
    const size_t at = (size_t)b * BS + s;
    const int cnt = p.bcount[b];
    const unsigned long long acc = p.rk[at], before = p.bm[at - (at > 0 ? 1 : 0)];
    const float pr = p.bp[at];
    const int ix = p.bi[at];
    if (s >= cnt) return;
    const int rank = s + (int)(acc >> kMassBits);
    const unsigned long long mass = (acc & kMassMask) + (s > 0 ? before : 0ull);
    p.keys[rank] = pr;
    p.vals[rank] = ix;
    if (p.approx) p.approx[rank] = (float)mass * 0x1p-47f;
}

// Blocks are index ranges, so among equal probabilities an entry of an earlier block comes first:
// the rank needs the other blocks' PROBABILITIES only -- all of them fit in one workgroup's LDS.
constexpr int kRankThreads = 1024;
template <int NB>
__global__ __launch_bounds__(kRankThreads) void topp_rank_kernel(ToppSortParams p) {
    __shared__ unsigned s_p[NB * kToppBlock];                      // 128 KB: every block's sorted probability bits
    const int tid = threadIdx.x;
    TOPP_STAMP(8);
    const int g = blockIdx.x * kRankThreads + tid;
    const int b = g / kToppBlock, s = g % kToppBlock;              // b is uniform over the workgroup
    int cnt[NB];
    int total = 0, most = 0;
#pragma unroll
    for (int o = 0; o < NB; o++) {
        cnt[o] = o < p.nblk ? p.bcount[o] : 0;
        total += cnt[o];
        most = max(most, cnt[o]);
    }
    if (g == 0) {
        *p.m = total;
        if (total == 0 && p.err) *p.err = 1u;
    }
    int mine = 0;
#pragma unroll
    for (int o = 0; o < NB; o++) mine = o == b ? cnt[o] : mine;
    if ((blockIdx.x * kRankThreads) % kToppBlock >= mine) return;  // uniform: nothing of this stretch is kept
    {   // stage every block's kept probabilities: all 2 NB loads of a thread in flight together
        constexpr int per_block = kToppBlock / kRankThreads;
        unsigned v[NB * per_block];
#pragma unroll
        for (int q = 0; q < NB * per_block; q++) {
            const int o = q / per_block, j = (q % per_block) * kRankThreads + tid;
            v[q] = j < cnt[o] ? __float_as_uint(p.bp[(size_t)o * kToppBlock + j]) : 0u;
        }
#pragma unroll
        for (int q = 0; q < NB * per_block; q++) {
            const int o = q / per_block, j = (q % per_block) * kRankThreads + tid;
            if (j < cnt[o]) s_p[o * kToppBlock + j] = v[q];
        }
    }
    __syncthreads();
    TOPP_STAMP(9);
    if (s >= mine) return;
    const unsigned key = s_p[b * kToppBlock + s];
    // pos[o] = entries of block o that precede this one.  "Precedes" is true on a prefix of the
    // sorted block, so pos grows by every power of two whose last covered entry still precedes.
    int pos[NB];
#pragma unroll
    for (int o = 0; o < NB; o++) pos[o] = 0;
    int top = 1;
    while (top <= most) top <<= 1;                                 // uniform: the first power of two above the longest list
    for (int step = top >> 1; step >= 1; step >>= 1) {
        unsigned probe[NB];
#pragma unroll
        for (int o = 0; o < NB; o++) probe[o] = s_p[o * kToppBlock + min(pos[o] + step - 1, kToppBlock - 1)];
#pragma unroll
        for (int o = 0; o < NB; o++) {
            const bool precedes = o < b ? probe[o] >= key : probe[o] > key;
            pos[o] += (pos[o] + step <= cnt[o] && precedes && o != b) ? step : 0;
        }
    }
    int rank = s;
#pragma unroll
    for (int o = 0; o < NB; o++) rank += pos[o];
    TOPP_STAMP(10);
    p.keys[rank] = __uint_as_float(key);
    p.vals[rank] = p.bi[(size_t)b * kToppBlock + s];
}

// The ranking spread over the chip (round 4).  topp_rank_kernel holds ALL sorted blocks in one workgroup's LDS and lets each of its
// 1 024 threads search the 15 other blocks: 169 000 conflicting LDS probes on ONE CU per workgroup, 32 CUs busy, the last wave out at
// 21.8 us (tools/topp_bench.hip).  Here a workgroup is a PAIR (b, o): it stages block o alone (8 KB), every entry of block b does ONE
// binary search in it, and the count goes into the entry's accumulator with an integer atomic -- nblk x (nblk - 1) workgroups of
// 22 000 probes each, all CUs busy; topp_rank_scatter_kernel then puts (p, index) at place = own place in the block + the sum.
__global__ __launch_bounds__(1024) void topp_rank_pairs_kernel(ToppSortParams p) {
    __shared__ unsigned s_o[kToppBlock];
    const int tid = threadIdx.x;
    const int b = blockIdx.x, o = blockIdx.y;
    if (b == o) return;
    const int cb = p.bcount[b], co = p.bcount[o];
    if (cb == 0 || co == 0) return;                                // uniform
    const unsigned k0 = tid < cb ? __float_as_uint(p.bp[(size_t)b * kToppBlock + tid]) : 0u;
    const unsigned k1 = tid + 1024 < cb ? __float_as_uint(p.bp[(size_t)b * kToppBlock + 1024 + tid]) : 0u;
    {
        const unsigned v0 = tid < co ? __float_as_uint(p.bp[(size_t)o * kToppBlock + tid]) : 0u;
        const unsigned v1 = tid + 1024 < co ? __float_as_uint(p.bp[(size_t)o * kToppBlock + 1024 + tid]) : 0u;
        s_o[tid] = v0; s_o[tid + 1024] = v1;
    }
    __syncthreads();
    // entries of block o that precede mine: "precedes" holds on a prefix of the sorted block, so the count grows by every power of
    // two whose last covered entry still precedes (equal probabilities: the earlier block's entry comes first)
    int top = 1;
    while (top <= co) top <<= 1;                                   // uniform
    int p0 = 0, p1 = 0;
    for (int step = top >> 1; step >= 1; step >>= 1) {
        const unsigned q0 = s_o[min(p0 + step - 1, kToppBlock - 1)], q1 = s_o[min(p1 + step - 1, kToppBlock - 1)];
        const bool pr0 = o < b ? q0 >= k0 : q0 > k0, pr1 = o < b ? q1 >= k1 : q1 > k1;
        p0 += (p0 + step <= co && pr0) ? step : 0;
        p1 += (p1 + step <= co && pr1) ? step : 0;
    }
    if (tid < cb && p0) atomicAdd(&p.racc[(size_t)b * kToppBlock + tid], p0);
    if (tid + 1024 < cb && p1) atomicAdd(&p.racc[(size_t)b * kToppBlock + 1024 + tid], p1);
}
__global__ __launch_bounds__(1024) void topp_rank_scatter_kernel(ToppSortParams p) {
    const int g = blockIdx.x * 1024 + threadIdx.x;
    const int b = g / kToppBlock, s = g % kToppBlock;
    if (g == 0) {
        int total = 0;
        for (int o = 0; o < p.nblk; o++) total += p.bcount[o];
        *p.m = total;
        if (total == 0 && p.err) *p.err = 1u;
    }
    if (b >= p.nblk || s >= p.bcount[b]) return;
    const int rank = s + p.racc[(size_t)b * kToppBlock + s];
    p.keys[rank] = p.bp[(size_t)b * kToppBlock + s];
    p.vals[rank] = p.bi[(size_t)b * kToppBlock + s];
}

// The same ranking for ANY number of blocks (vocabularies above 32768 entries; rama_set_tuning "topp_sort" = 0 anywhere):
// the other blocks' sorted probabilities are probed in global memory (L2) instead of LDS, eight blocks' probes of a
// step in flight together.  Replaces the library radix sort of rounds 1-2: no library kernel is left in the product.
__global__ __launch_bounds__(256) void topp_rank_global_kernel(ToppSortParams p) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int b = g / kToppBlock, s = g % kToppBlock;              // b is uniform over the workgroup (2048 % 256 == 0)
    if (g == 0) {
        int total = 0;
        for (int o = 0; o < p.nblk; o++) total += p.bcount[o];
        *p.m = total;
        if (total == 0 && p.err) *p.err = 1u;
    }
    if (b >= p.nblk) return;
    const int mine = p.bcount[b];
    if (s >= mine) return;
    const unsigned key = __float_as_uint(p.bp[(size_t)b * kToppBlock + s]);
    int rank = s;
    for (int o0 = 0; o0 < p.nblk; o0 += 8) {
        int cnt[8], pos[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const int o = o0 + q; cnt[q] = (o < p.nblk && o != b) ? p.bcount[o] : 0; pos[q] = 0; }
        for (int step = kToppBlock; step >= 1; step >>= 1) {         // pos grows by every power of two whose last covered entry still precedes
            unsigned probe[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int o = o0 + q;
                const int j = min(pos[q] + step - 1, kToppBlock - 1);
                probe[q] = cnt[q] > 0 ? __float_as_uint(p.bp[(size_t)min(o, p.nblk - 1) * kToppBlock + j]) : 0u;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int o = o0 + q;
                const bool precedes = o < b ? probe[q] >= key : probe[q] > key;
                pos[q] += (pos[q] + step <= cnt[q] && precedes) ? step : 0;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; q++) rank += pos[q];
    }
    p.keys[rank] = __uint_as_float(key);
    p.vals[rank] = p.bi[(size_t)b * kToppBlock + s];
}

// ---------------------------------------------------------------- the running sum, in parallel and exact
// sample_top_q adds the sorted probabilities one by one in fp32 (infer.rs:70-73): cum_i = fl(cum_{i-1} + p_i).
// The rounding of every add depends on all earlier ones, so the chain cannot simply be re-associated
// -- but between two changes of cum's exponent it is INTEGER arithmetic: with U = ulp(cum) fixed,
// cum = c U for an integer c in [2^23, 2^24), p_i = (a_i + f_i) U with 0 <= f_i < 1, and round-to-
// nearest-even gives c + a_i (f_i < 1/2), c + a_i + 1 (f_i > 1/2) or, on a tie, the even one of the
// two.  Each element is therefore a map c -> c + D[c & 1] (two increments, one per parity of c; they
// differ only on ties), such maps compose into maps of the same form, and composition is associative:
// a workgroup scan over them yields every cum_i of the stretch bit for bit.  The first element whose
// c reaches 2^24 leaves the binade; that one add is done in fp32 and the next round starts behind it
// with the new U (cum grows from > 3e-6 to <= 1: at most ~20 rounds, and the windows grow with the
// position as the binades do).  Sorted input guarantees p_i <= cum_{i-1}, i.e. a shift >= 0 below.
constexpr int kScanClamp = 1 << 26;        // saturation keeps sums past a binade's end from overflowing; they are discarded
constexpr int kScanRun = 32;               // most elements per thread and round
constexpr int kNoEvent = 0x7FFFFFFF;

struct Inc { int d0, d1; };                // increment for even / odd c
__device__ __forceinline__ Inc inc_then(Inc a, Inc b) {            // a first, then b
    Inc r;
    r.d0 = min(a.d0 + ((a.d0 & 1) ? b.d1 : b.d0), kScanClamp);
    r.d1 = min(a.d1 + (((1 + a.d1) & 1) ? b.d1 : b.d0), kScanClamp);
    return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Inc inc_dpp(Inc v) {                    // lanes without a source get the identity map (0, 0)
    return Inc{__builtin_amdgcn_update_dpp(0, v.d0, CTRL, ROW_MASK, 0xF, true), __builtin_amdgcn_update_dpp(0, v.d1, CTRL, ROW_MASK, 0xF, true)};
}
// One element as a packed word: bits 0..27 the increment d0 for an even c, bits 30..31 the correction
// t in {0, +1, -1} an odd c adds to it (non-zero only on ties).  q = p_i / U, formed exactly by a
// power-of-two multiply (capped at 2^25: an element larger than the sum -- unsorted input -- only has
// to trigger the fp32 add).  Round-to-nearest-even of q is the increment for an even c in every case,
// ties included (the even one of a, a+1); an odd c takes the other one on a tie.
__device__ __forceinline__ int elem_of(float p, float invU) {
    const float q = fminf(p * invU, 33554432.0f);
    const float r = rintf(q);
    const bool tie = __builtin_amdgcn_fractf(q) == 0.5f;
    const int t = tie ? (r == floorf(q) ? 0x40000000 : (int)0xC0000000) : 0;        // a even: (a, a+1); a odd: (a+1, a)
    return (int)r | t;
}
__device__ __forceinline__ int elem_apply(int c, int e) { return c + (e & 0x0FFFFFFF) + ((c & 1) ? (e >> 30) : 0); }

// The whole sorted list lives in LDS (n <= 32768), element i at slot i + i/32: a thread's run of R
// consecutive elements and the coalesced sweeps over i both spread over the banks.  A round writes
// its running sums over the inputs it has consumed (only those: the slots behind an event still
// hold probabilities for the next round).  Round state is kept twice and used in turn, so a round
// costs three barriers.
struct ScanState { int pos, done, last; float cum; };
template <int CAP>                        // the longest list (a multiple of 1024)
struct ScanSharedT {
    float x[CAP + CAP / 32];
    Inc w[2][16];
    int ev[2][16];
    ScanState st[2];
    int next;
};
using ScanShared = ScanSharedT<32768>;
__device__ __forceinline__ int scan_slot(int i) { return i + (i >> 5); }

// One round: the sums of elements [pos, pos + 64 NW R) as long as cum stays in its binade.  Short
// windows run on 4 waves -- one per SIMD; the other waves only keep the barriers company.
template <int R, int NW, int CAP>
__device__ __forceinline__ void scan_round(const ToppParams& p, ScanSharedT<CAP>& sh, int m, int par) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const ScanState st = sh.st[par];
    const int pos = st.pos;
    const unsigned cb = __float_as_uint(st.cum);
    const int E = (int)(cb >> 23);
    const int c0 = (int)((cb & 0x7FFFFFu) | 0x800000u);
    const float U = __uint_as_float((unsigned)(E - 23) << 23);
    const float invU = __uint_as_float((unsigned)(277 - E) << 23);                  // 2^(150 - E), exact scaling
    // the largest c that is no event: c < 2^24 and not c U > topp  <=>  c <= floor(topp / U)
    const int climit = min((int)floorf(fminf(p.topp * invU, 16777216.0f)), (1 << 24) - 1);
    const int i0 = pos + tid * R;
    const bool active = wave < NW;                                 // wave-uniform
    int e[R];                                                      // the run's elements; after the replay: c behind each
    Inc before{0, 0};
    if (active) {
        // the run's map as the images of an even and an odd start (0 and 1)
        int x0 = 0, x1 = 1;
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int i = i0 + k;
            const float pk = sh.x[scan_slot(min(i, CAP - 1))];       // no branch around the read
            e[k] = elem_of(i < m ? pk : 0.0f, invU);
            x0 = elem_apply(x0, e[k]);
            x1 = elem_apply(x1, e[k]);
        }
        Inc inc{min(x0, kScanClamp), min(x1 - 1, kScanClamp)};
        // inclusive scan of the threads' maps over the wave (DPP moves only), then over the waves
        inc = inc_then(inc_dpp<0x111, 0xF>(inc), inc);             // row_shr:1
        inc = inc_then(inc_dpp<0x112, 0xF>(inc), inc);             // row_shr:2
        inc = inc_then(inc_dpp<0x114, 0xF>(inc), inc);             // row_shr:4
        inc = inc_then(inc_dpp<0x118, 0xF>(inc), inc);             // row_shr:8
        inc = inc_then(inc_dpp<0x142, 0xA>(inc), inc);             // row_bcast:15 into rows 1 and 3
        inc = inc_then(inc_dpp<0x143, 0xC>(inc), inc);             // row_bcast:31 into rows 2 and 3
        if (lane == 63) sh.w[par][wave] = inc;
        before = inc_dpp<0x138, 0xF>(inc);                         // wave_shr:1: all earlier lanes of this wave
    }
    __syncthreads();
    int ev = kNoEvent, c_before_ev = 0;
    if (active) {
        Inc wt[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) wt[w] = sh.w[par][w];
        Inc pre{0, 0};
#pragma unroll
        for (int w = 0; w < NW - 1; w++) if (w < wave) pre = inc_then(pre, wt[w]);
        pre = inc_then(pre, before);
        // replay the run from its true starting value; the event condition is monotone along the list
        int c = min(c0 + ((c0 & 1) ? pre.d1 : pre.d0), kScanClamp);
        int quiet = 0;
        c_before_ev = c;
#pragma unroll
        for (int k = 0; k < R; k++) {
            c = elem_apply(c, e[k]);
            e[k] = c;
            const bool ok = c <= climit;
            quiet += ok;
            c_before_ev = ok ? c : c_before_ev;
        }
        ev = quiet < R && i0 + quiet < m ? i0 + quiet : kNoEvent;
        const int wev = ~wave_max_i(~ev);                          // minimum (ev >= 0): ~ reverses the order
        if (lane == 0) sh.ev[par][wave] = wev;
    }
    __syncthreads();
    int first = sh.ev[par][0];
#pragma unroll
    for (int w = 1; w < NW; w++) first = min(first, sh.ev[par][w]);
    if (active) {
        const int upto = min(first == kNoEvent ? pos + 64 * NW * R : first, m);     // the sums of [pos, upto) are final
#pragma unroll
        for (int k = 0; k < R; k++)
            if (i0 + k < upto) sh.x[scan_slot(i0 + k)] = (float)e[k] * U;
        ScanState nx = st;
        if (first == kNoEvent) {
            if (tid == 64 * NW - 1) { nx.pos = pos + 64 * NW * R; nx.cum = (float)e[R - 1] * U; sh.st[par ^ 1] = nx; }
        } else if (ev == first) {
            // the event element's own add in fp32 (it leaves the binade, passes topp, or both), by its owner
            const float s = (float)c_before_ev * U + sh.x[scan_slot(first)];
            sh.x[scan_slot(first)] = s;
            nx.cum = s;
            if (s > p.topp) { nx.last = first; nx.done = 1; }
            else nx.pos = first + 1;
            sh.st[par ^ 1] = nx;
        }
    }
    __syncthreads();
}

// the pick of one workgroup for a list of at most CAP entries (the launch below; topp_pick.hpp takes it for short lists)
template <int CAP>
__device__ __forceinline__ void topp_pick_scan_body(const ToppParams& p, const ArgmaxParams& fin, ScanSharedT<CAP>& sh) {
    const int tid = threadIdx.x;
    int cpos = 0, n_forced = 0, n_out = 0, forced_tok = -1;
    if (fin.ctl && tid == 0) {
        cpos = fin.ctl->pos; n_forced = fin.ctl->n_forced; n_out = fin.ctl->n_out;
        if (cpos < n_forced) forced_tok = fin.forced[cpos];
    }
    TOPP_STAMP(16);
    const int m = min(*p.m, CAP);
    {
        float v[CAP / 1024];
#pragma unroll
        for (int j = 0; j < CAP / 1024; j++) { const int i = j * 1024 + tid; v[j] = i < m ? p.keys[i] : 0.0f; }
#pragma unroll
        for (int j = 0; j < CAP / 1024; j++) { const int i = j * 1024 + tid; sh.x[scan_slot(i)] = v[j]; }       // zeros behind the list
    }
    __syncthreads();
    TOPP_STAMP(17);
    // the first 64 sums by the lane ripple of topp_pick_kernel: cum passes through several short binades there
    if (tid < 64) {
        const float pv = tid < m ? sh.x[scan_slot(tid)] : 0.0f;
        float sv = pv;
#pragma unroll
        for (int k = 0; k < 63; k++)
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(sv) : "v"(pv));
        if (tid < m) sh.x[scan_slot(tid)] = sv;
        const unsigned long long over = __ballot(tid < m && sv > p.topp);
        const int l0 = over ? __ffsll((long long)over) - 1 : 63;  // zero padding: lane 63 holds the sum of all
        const float c = __shfl(sv, l0);
        if (tid == 0) {
            ScanState st;
            st.last = over ? l0 : (m > 0 ? m - 1 : 0);
            st.done = over ? 1 : 0;
            st.pos = min(64, m);
            st.cum = c;
            sh.st[0] = st;
        }
    }
    __syncthreads();
    TOPP_STAMP(18);
    int par = 0, rounds = 0;
    while (true) {
        const ScanState st = sh.st[par];
        if (st.done || st.pos >= m) break;                         // uniform
        const int want = min(st.pos, m - st.pos);                  // binades double in length; never beyond the list
        if (want <= 1024) scan_round<4, 4, CAP>(p, sh, m, par);
        else if (want <= 2048) scan_round<8, 4, CAP>(p, sh, m, par);
        else if (want <= 4096) scan_round<4, 16, CAP>(p, sh, m, par);
        else if (want <= 8192) scan_round<8, 16, CAP>(p, sh, m, par);
        else if (want <= 16384) scan_round<16, 16, CAP>(p, sh, m, par);
        else scan_round<kScanRun, 16, CAP>(p, sh, m, par);
        TOPP_STAMP(24 + min(rounds, 30));
        par ^= 1; rounds++;
    }
    TOPP_STAMP(19);
    // r = u * cum; the first i < last whose running sum exceeds r wins, else `last` (infer.rs:75-84):
    // the sums never decrease, so that is the number of i < last with cum_i <= r
    const int last = sh.st[par].last;
    const float r = p.u * sh.st[par].cum;
    int below = 0;
    for (int i = tid; i < last; i += 1024) below += !(r < sh.x[scan_slot(i)]);
    if (p.prefix)                                                  // kept for inspection (tests): the sums up to the crossing
        for (int i = tid; i <= last && i < m; i += 1024) p.prefix[i] = sh.x[scan_slot(i)];
    const int best = min(block_sum_i(below), last);
    if (tid == 0) {
        const int idx = m > 0 ? p.vals[best] : -1;
        sh.next = finish_step(fin, idx, cpos, n_forced, n_out, forced_tok);
    }
    gather_next_embedding(fin, &sh.next);
    TOPP_STAMP(20);
    (void)rounds;
}
__global__ __launch_bounds__(1024) void topp_pick_scan_kernel(ToppParams p, ArgmaxParams fin) {
    __shared__ ScanShared sh;
    topp_pick_scan_body(p, fin, sh);
}

}  // namespace rama
