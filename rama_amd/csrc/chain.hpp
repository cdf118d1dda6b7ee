// chain.hpp -- the reference's arithmetic order AT STREAMING SPEED ("parity mode",
// rama_set_tuning(ctx, "ref_order", 1) on a resident model).
//
// ref_order.hpp reproduces engine/src/device/cpu.rs operation by operation but parallel only across
// whole outputs (one thread per row): bit-identical to the oracle, ~2x slower than the fast path.  This
// file keeps the same rounding sequence and restores the parallelism the sequence allows:
//
//  * matmul (cpu.rs:127-153): o[r] = (v0 + v1) + (v2 + v3), v_j += W[r][4i + j] * x[4i + j] for i
//    ascending, product and sum rounded separately.  The four lane sums of a row are independent
//    chains, so ONE LANE OWNS ONE (row, j) CHAIN: a wave is 16 rows x 4 chains, a quad finishes a row
//    with two DPP adds.  A model keeps a CHAIN-ORDER copy of every matrix (model.hip):
//        [rows / 16][K / 16][lane = (r % 16) * 4 + j][t = 0..3] = W[r][16 s + 4 t + j]
//    so one `buffer_load_dwordx4 nt` gives a lane four consecutive steps of its own chain and the
//    wave reads 1 KiB contiguous; a wave streams its 16 rows front to back as ONE contiguous region
//    (16 K floats) through a ring of D loads in flight.  The activations are staged once per wave in
//    LDS in the same order, so a step's four x values are one broadcast ds_read_b128.
//    A chain is K / 4 dependent adds (~5 cycles each): 1-3 us against 10-80 us of HBM time per launch,
//    so the kernels stay bandwidth-bound as long as the ring keeps the loads coming.
//  * rmsnorm (cpu.rs:99-117) and softmax (cpu.rs:187-192; the oracle's sequential order) sum
//    non-negative terms one by one.  That chain is reproduced bit for bit by the parallel scan of
//    topp_sort.hpp: between two changes of the sum's exponent the fp32 adds are integer increments that
//    depend only on the parity of the running integer, such maps compose associatively, a workgroup
//    scans them, and the one add that leaves the binade is done in fp32 (seq_sum_exact below).
//  * attention (cpu.rs:23-52): one thread per timestep for the sequential q.k dots, cooperative tile
//    loads + one thread per head column for the t-ascending value accumulation.
//  * RoPE, SiLU * gate and the residual adds ride in the matvec epilogues with the reference's
//    roundings (contraction off, glibc's expf restated in ref_order.hpp).
// Results are bit-identical to ref_order.hpp's and to the oracle's (tests/test_hip_ref_order.py,
// tests/test_hip_parity_7b.py).
#pragma once
#include "kernels.hpp"
#include "ref_order.hpp"
#include "topp_sort.hpp"
#include "seqsum_fast.hpp"
#include "layer_fused.hpp"      // put_tagged, the epoch / error-word conventions of in-launch hand-offs
#include "attn_wo.hpp"          // st_sc1 / ld4_sc1: write-through stores and L1-bypassing loads of in-launch hand-offs

namespace rama {

// (the copy itself is made by chain_weights_kernel in model.hip)

// ---------------------------------------------------------------- the sequential sum, exact and parallel
// s = ((a_0 + a_1) + a_2) + ... of n non-negative floats a[scan_slot(i)] in LDS, every add rounded to
// fp32 in that order (Iterator::sum of cpu.rs:112 / the oracle's softmax sum), computed by all NW
// waves of the workgroup.  Same construction as topp_sort.hpp's running sum (elem_of / Inc maps), without
// the stored partial sums: the first 64 elements by a lane ripple (the sum passes through several short
// binades there), then rounds of NW * 64 * R elements -- scan of the increment maps, the first element
// that would leave the binade is added in fp32 by its owner and the next round starts behind it.
// Sums outside the scan's exponent range (zero, subnormal-ish, inf, nan) finish on one thread.
template <int NW>
struct SeqSumShared {
    Inc w[2][NW];
    int ev[2][NW];
    int pos[2];
    float cum[2];
};

template <int R, int NW>
__device__ __forceinline__ void seqsum_round(const float* a, int n, SeqSumShared<NW>& sh, int par) {
    RAMA_NO_CONTRACT
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos = sh.pos[par];
    const unsigned cb = __float_as_uint(sh.cum[par]);
    const int E = (int)(cb >> 23);
    const int c0 = (int)((cb & 0x7FFFFFu) | 0x800000u);
    const float U = __uint_as_float((unsigned)(E - 23) << 23);
    const float invU = __uint_as_float((unsigned)(277 - E) << 23);
    constexpr int climit = (1 << 24) - 1;
    const int i0 = pos + tid * R;
    int e[R];
    int x0 = 0, x1 = 1;
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int i = i0 + k;
        const float pk = a[scan_slot(min(i, n - 1))];
        e[k] = elem_of(i < n ? pk : 0.0f, invU);
        x0 = elem_apply(x0, e[k]);
        x1 = elem_apply(x1, e[k]);
    }
    Inc inc{min(x0, kScanClamp), min(x1 - 1, kScanClamp)};
    inc = inc_then(inc_dpp<0x111, 0xF>(inc), inc);             // row_shr:1
    inc = inc_then(inc_dpp<0x112, 0xF>(inc), inc);             // row_shr:2
    inc = inc_then(inc_dpp<0x114, 0xF>(inc), inc);             // row_shr:4
    inc = inc_then(inc_dpp<0x118, 0xF>(inc), inc);             // row_shr:8
    inc = inc_then(inc_dpp<0x142, 0xA>(inc), inc);             // row_bcast:15 into rows 1 and 3
    inc = inc_then(inc_dpp<0x143, 0xC>(inc), inc);             // row_bcast:31 into rows 2 and 3
    if (lane == 63) sh.w[par][wave] = inc;
    const Inc before = inc_dpp<0x138, 0xF>(inc);               // wave_shr:1: all earlier lanes of this wave
    __syncthreads();
    Inc pre{0, 0};
#pragma unroll
    for (int w = 0; w < NW - 1; w++) if (w < wave) pre = inc_then(pre, sh.w[par][w]);
    pre = inc_then(pre, before);
    int c = min(c0 + ((c0 & 1) ? pre.d1 : pre.d0), kScanClamp);
    int quiet = 0, c_before_ev = c;
#pragma unroll
    for (int k = 0; k < R; k++) {
        c = elem_apply(c, e[k]);
        const bool ok = c <= climit;
        quiet += ok;
        c_before_ev = ok ? c : c_before_ev;
    }
    const int ev = (quiet < R && i0 + quiet < n) ? i0 + quiet : kNoEvent;
    const int wev = ~wave_max_i(~ev);
    if (lane == 0) sh.ev[par][wave] = wev;
    __syncthreads();
    int first = sh.ev[par][0];
#pragma unroll
    for (int w = 1; w < NW; w++) first = min(first, sh.ev[par][w]);
    if (first == kNoEvent) {
        if (tid == 64 * NW - 1) { sh.pos[par ^ 1] = pos + 64 * NW * R; sh.cum[par ^ 1] = (float)c * U; }
    } else if (ev == first) {
        sh.pos[par ^ 1] = first + 1;
        sh.cum[par ^ 1] = (float)c_before_ev * U + a[scan_slot(first)];
    }
    __syncthreads();
}

template <int NW>
__device__ __forceinline__ float seq_sum_exact(const float* a, int n, SeqSumShared<NW>& sh) {
    RAMA_NO_CONTRACT
    const int tid = threadIdx.x;
    if (tid < 64) {
        // the first elements by a lane ripple: lane i holds a_i, 63 steps of s_i = s_{i-1} + a_i leave S_i in lane i
        // (zeros behind the list leave the sum as it is).  Up to kRipple blocks of 64: the sum passes through many
        // short binades here, where a scan round would cost more than the ~0.15 us of a ripple.
        constexpr int kRipple = 4;
        float carry = 0.0f;
        int done = 0;
#pragma unroll 1
        for (int b = 0; b < kRipple && done < n; b++, done += 64) {
            const int i = done + tid;
            const float pv = i < n ? a[scan_slot(i)] : 0.0f;
            float sv = (b > 0 && tid == 0) ? carry + pv : pv;
#pragma unroll
            for (int k = 0; k < 63; k++)
                asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(sv) : "v"(pv));
            carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sv), 63));
        }
        if (tid == 0) { sh.pos[0] = min(done, n); sh.cum[0] = carry; }
    }
    __syncthreads();
    int par = 0;
    while (true) {
        const int pos = sh.pos[par];
        if (pos >= n) break;                                      // uniform
        const int E = (int)(__float_as_uint(sh.cum[par]) >> 23);
        if (E < 24 || E > 253) {                                  // uniform: outside the scan's range
            if (tid == 0) {
                float s = sh.cum[par];
                for (int i = pos; i < n; i++) s = s + a[scan_slot(i)];
                sh.pos[par ^ 1] = n; sh.cum[par ^ 1] = s;
            }
            __syncthreads();
            par ^= 1;
            continue;
        }
        const int want = min(max(pos, 64), n - pos);              // binades double in length
        if (want <= 64 * NW) seqsum_round<1, NW>(a, n, sh, par);
        else if (want <= 128 * NW) seqsum_round<2, NW>(a, n, sh, par);
        else if (want <= 256 * NW) seqsum_round<4, NW>(a, n, sh, par);
        else if (want <= 512 * NW) seqsum_round<8, NW>(a, n, sh, par);
        else seqsum_round<16, NW>(a, n, sh, par);
        par ^= 1;
    }
    return sh.cum[par];
}

// ---------------------------------------------------------------- the same sum in ONE pass: predict, then verify
// seq_sum_exact pays three barriers and a scan per binade the sum passes through (6-10 rounds for 4096 terms).
// But the increment map of a run of elements depends only on the BINADE the sum is in while it crosses the
// run, not on the sum's value -- and the binade can be predicted from an ordinary (tree-ordered, approximate)
// parallel prefix sum, which differs from the sequential fp32 sum by ~1e-6 relative (1.2e-4 is allowed for; a
// wrong prediction is caught in step 4 and costs only time).  So:
//   1. thread t owns the contiguous run [t R, (t + 1) R); approximate prefix sums lo_t (before the run) and hi_t
//      (behind it) come from one block scan;
//   2. if lo_t (1 - 2^-13) and hi_t (1 + 2^-13) lie in one binade e, the run is a MAP item: its elements' increments
//      for U = ulp(2^e) are composed into one map; otherwise (the sum may change binade inside the run, or is
//      still tiny: the first few runs) it is a SEQ item;
//   3. inside a wave, neighbouring MAP items of one binade are composed by a segmented DPP scan;
//   4. ONE lane then walks the items in order with the true sum: a SEQ item's elements are added in fp32, a
//      segment's map is applied as c -> c + D[c & 1] AFTER checking that the true sum really is in the predicted
//      binade, and that it still is behind the segment (c < 2^24: the sum only grows, so every add in between was
//      in that binade too).  Either check failing means the prediction was wrong: the caller falls back to
//      seq_sum_exact.  The result is the sequential sum bit for bit whenever it is returned.
// phase time stamps for tools/seqsum_bench.hip only (100 MHz counter, thread 0)
#ifdef RAMA_SEQ_STAMPS
__device__ unsigned long long g_seq_stamps[64];
#define SEQ_STAMP(id) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_seq_stamps[id] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SEQ_STAMP(id) do { } while (0)
#endif
__device__ unsigned g_pred_stats[2];      // diagnostics: [0] sums the one-pass path returned, [1] sums that fell back
constexpr int kMaxSeq = 12;                // SEQ items the one-pass walk handles (binade crossings behind the first 64 elements)
template <int NW>
struct PredShared {
    float wsum[NW];
    int wf[NW], wd0[NW], wd1[NW], wefirst[NW], welast[NW];
    unsigned long long smask[NW];
    int d0[NW * 64], d1[NW * 64], ebin[NW * 64];
    int seqlist[kMaxSeq];
    float head;
    float result;
    int fail;
};
struct SegInc { int f; Inc m; };
__device__ __forceinline__ SegInc seg_then(SegInc a, SegInc b) {      // a (earlier items) first, then b; a segment start in b hides a
    SegInc r;
    const Inc ab = inc_then(a.m, b.m);
    r.m.d0 = b.f ? b.m.d0 : ab.d0;
    r.m.d1 = b.f ? b.m.d1 : ab.d1;
    r.f = a.f | b.f;
    return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ SegInc seg_dpp(SegInc v) {                 // lanes without a source get (no start, identity)
    return SegInc{__builtin_amdgcn_update_dpp(0, v.f, CTRL, ROW_MASK, 0xF, true), inc_dpp<CTRL, ROW_MASK>(v.m)};
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float fadd_dpp(float v) {                  // v + (value of the source lane, 0 without one)
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}

// Elements 0..63 are summed by the lane ripple of seq_sum_exact (the sum passes through a binade every few
// elements there); the items cover elements 64.. : thread t owns [64 + t R, 64 + (t + 1) R).
// RMAX: the run is fetched from LDS once, all reads in flight together, and kept in registers (R <= RMAX)
template <int NW, int RMAX>
__device__ __forceinline__ bool seq_sum_predict_r(const float* a, int n, PredShared<NW>& ps, float* out) {
    RAMA_NO_CONTRACT
    constexpr int T = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = max((n - 64 + T - 1) / T, 1);
    const int i0 = 64 + tid * R;
    float run[RMAX];
#pragma unroll
    for (int k = 0; k < RMAX; k++) { const int i = i0 + k; const float v_ = a[scan_slot(min(i, n - 1))]; run[k] = (k < R && i < n) ? v_ : 0.0f; }
    SEQ_STAMP(1);
    // 1. approximate prefix sums (wave 0 first ripples the head: exact, and the prefix's base)
    if (wave == 0) {
        const float pv = lane < n ? a[scan_slot(lane)] : 0.0f;
        float sv = pv;
#pragma unroll
        for (int k = 0; k < 63; k++)
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(sv) : "v"(pv));
        if (lane == 63) { ps.head = sv; ps.fail = 0; }
    }
    float loc = 0.0f;
    {   // any order will do for the estimate: four partial sums
        float l4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < RMAX; k++) l4[k & 3] += run[k];
        loc = (l4[0] + l4[1]) + (l4[2] + l4[3]);
    }
    float inc = loc;
    inc = fadd_dpp<0x111, 0xF>(inc); inc = fadd_dpp<0x112, 0xF>(inc); inc = fadd_dpp<0x114, 0xF>(inc); inc = fadd_dpp<0x118, 0xF>(inc);
    inc = fadd_dpp<0x142, 0xA>(inc); inc = fadd_dpp<0x143, 0xC>(inc);
    if (lane == 63) ps.wsum[wave] = inc;
    __syncthreads();
    float base = ps.head;
#pragma unroll
    for (int w = 0; w < NW - 1; w++) base += w < wave ? ps.wsum[w] : 0.0f;
    const float lo = base + (inc - loc), hi = base + inc;
    SEQ_STAMP(2);
    // 2. one binade for the whole run?
    const int el = (int)(__float_as_uint(lo * (1.0f - 0x1p-13f)) >> 23), eh = (int)(__float_as_uint(hi * (1.0f + 0x1p-13f)) >> 23);
    const bool conf = lo > 0.0f && el == eh && el >= 24 && el <= 253;     // (a sign bit or nan makes the exponents differ or leave the range)
    const int e = conf ? el : -1;
    const int eprev = __builtin_amdgcn_update_dpp(-2, e, 0x138, 0xF, 0xF, false);      // wave_shr:1; lane 0 keeps -2
    // a segment start inside the wave; whether lane 0 continues the previous wave's segment is settled in step 3b
    const bool F = !conf || (lane != 0 && eprev != e);
    SegInc sv{F ? 1 : 0, Inc{0, 0}};
    if (conf) {
        const float invU = __uint_as_float((unsigned)(277 - e) << 23);
        int x0 = 0, x1 = 1;
#pragma unroll
        for (int k = 0; k < RMAX; k++) {           // (elements behind the run are zeros: identity increments)
            const int el_ = elem_of(run[k], invU);
            x0 = elem_apply(x0, el_);
            x1 = elem_apply(x1, el_);
            if ((k & 7) == 7) { x0 = min(x0, kScanClamp); x1 = min(x1, kScanClamp); }      // 8 increments of <= 2^25 cannot overflow
        }
        sv.m = Inc{min(x0, kScanClamp), min(x1, kScanClamp) - 1};
    }
    SEQ_STAMP(3);
    // 3a. segmented inclusive scan over the wave
    sv = seg_then(seg_dpp<0x111, 0xF>(sv), sv);
    sv = seg_then(seg_dpp<0x112, 0xF>(sv), sv);
    sv = seg_then(seg_dpp<0x114, 0xF>(sv), sv);
    sv = seg_then(seg_dpp<0x118, 0xF>(sv), sv);
    sv = seg_then(seg_dpp<0x142, 0xA>(sv), sv);
    sv = seg_then(seg_dpp<0x143, 0xC>(sv), sv);
    const unsigned long long sm = __ballot(!conf);
    if (lane == 0) { ps.wefirst[wave] = e; ps.smask[wave] = sm; }
    if (lane == 63) { ps.wf[wave] = sv.f; ps.wd0[wave] = sv.m.d0; ps.wd1[wave] = sv.m.d1; ps.welast[wave] = e; }
    __syncthreads();
    // 3b. across the waves: wave w' continues the segment that ends wave w' - 1 iff both border items are MAP items of one binade
    {
        SegInc carry{1, Inc{0, 0}};
        int prev_last = -3, rank = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            if (w < wave) {
                SegInc tot{ps.wf[w], Inc{ps.wd0[w], ps.wd1[w]}};
                const int ef = ps.wefirst[w];
                if (!(ef >= 0 && ef == prev_last)) tot.f = 1;
                carry = seg_then(carry, tot);
                prev_last = ps.welast[w];
                rank += __popcll(ps.smask[w]);
            }
        }
        const int ef = ps.wefirst[wave];
        const bool cont = ef >= 0 && ef == prev_last;            // uniform over the wave
        if (!sv.f && cont) sv.m = inc_then(carry.m, sv.m);
        ps.d0[tid] = sv.m.d0; ps.d1[tid] = sv.m.d1; ps.ebin[tid] = e;
        if (!conf) {
            rank += __popcll(sm & ((1ull << lane) - 1ull));
            if (rank < kMaxSeq) ps.seqlist[rank] = tid;
        }
    }
    __syncthreads();
    SEQ_STAMP(4);
    // 4. the walk, by wave 0, every operand fetched up front: per SEQ item (in order) the map of the segment in front
    // of it, then its R elements (lane k: element k; zeros behind the run and the list leave the sum as it is); at the
    // end the segment behind the last SEQ item.
    if (wave == 0) {
        int nseq = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) nseq += __popcll(ps.smask[w]);
        int it[kMaxSeq + 1], cd0[kMaxSeq + 1], cd1[kMaxSeq + 1], ceb[kMaxSeq + 1];
        float ev[kMaxSeq];
#pragma unroll
        for (int g = 0; g <= kMaxSeq; g++) {
            it[g] = (g < kMaxSeq && g < nseq) ? ps.seqlist[g] : T;             // behind the last SEQ item: the item "T"
        }
#pragma unroll
        for (int g = 0; g <= kMaxSeq; g++) {
            const int t = max(it[g] - 1, 0);
            cd0[g] = ps.d0[t]; cd1[g] = ps.d1[t]; ceb[g] = ps.ebin[t];
            if (g < kMaxSeq) {
                const int j = 64 + it[g] * R + lane;
                ev[g] = (it[g] < T && lane < R && j < n) ? a[scan_slot(min(j, n - 1))] : 0.0f;
            }
        }
        float s = ps.head;
        int bad = nseq > kMaxSeq ? 1 : 0;
        const int nsteps = __builtin_amdgcn_readfirstlane(min(nseq, kMaxSeq));
#pragma unroll
        for (int g = 0; g <= kMaxSeq; g++) {
            if (g <= nsteps) {                                    // uniform
            // a segment lies in front of SEQ item g iff the item before it is a MAP item (and there is one)
            const int prev = g == 0 ? -1 : it[g - 1];
            const bool seg = it[g] - 1 > prev;
            {
                const unsigned sb = __float_as_uint(s);
                const int c0 = (int)((sb & 0x7FFFFFu) | 0x800000u);
                const int c = c0 + ((c0 & 1) ? cd1[g] : cd0[g]);
                const bool wrong = (int)(sb >> 23) != ceb[g] || c > (1 << 24) - 1;
                const float sn = (float)c * __uint_as_float((unsigned)(max(ceb[g], 23) - 23) << 23);
                bad |= (seg && wrong) ? 1 : 0;
                s = seg ? sn : s;
            }
            if (g < kMaxSeq && g < nsteps) {
                // the run's elements sit in lanes 0..R-1: lane 0 is seeded with s + e_0 and a ripple of >= R - 1 steps
                // t_i = t_(i-1) + e_i leaves the sum in lane R - 1 (further steps recompute the same values)
                float t_ = lane == 0 ? s + ev[g] : ev[g];
                for (int k = 1; k < R; k += 8) {
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(t_) : "v"(ev[g]));
                }
                s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_), min(R, 64) - 1));
            }
            }
        }
        if (lane == 0) { ps.result = s; ps.fail = bad; atomicAdd(&g_pred_stats[bad ? 1 : 0], 1u); }
    }
    __syncthreads();
    SEQ_STAMP(5);
    *out = ps.result;
    return ps.fail == 0;
}

// short lists: wave 0 ripples through them 64 elements at a time (~0.3 us per block), nothing else to set up
constexpr int kRippleMax = 512;
template <int NW>
__device__ __forceinline__ float seq_sum_ripples(const float* a, int n, PredShared<NW>& ps) {
    RAMA_NO_CONTRACT
    const int tid = threadIdx.x;
    if (tid < 64) {
        float carry = 0.0f;
        for (int done = 0; done < n; done += 64) {
            const int i = done + tid;
            const float pv = i < n ? a[scan_slot(i)] : 0.0f;
            float sv = (done > 0 && tid == 0) ? carry + pv : pv;
#pragma unroll
            for (int k = 0; k < 63; k++)
                asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(sv) : "v"(pv));
            carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sv), 63));
        }
        if (tid == 0) ps.result = carry;
    }
    __syncthreads();
    return ps.result;
}

template <int NW>
__device__ __forceinline__ bool seq_sum_predict(const float* a, int n, PredShared<NW>& ps, float* out) {
    if (n <= kRippleMax) { *out = seq_sum_ripples<NW>(a, n, ps); return true; }     // uniform
    const int R = (n - 64 + NW * 64 - 1) / (NW * 64);             // uniform
    if (R <= 4) return seq_sum_predict_r<NW, 4>(a, n, ps, out);
    if (R <= 16) return seq_sum_predict_r<NW, 16>(a, n, ps, out);
    if (R <= 32) return seq_sum_predict_r<NW, 32>(a, n, ps, out);
    return seq_sum_predict_r<NW, 64>(a, n, ps, out);
}

// ---------------------------------------------------------------- cpu.rs:99-117 rmsnorm, one workgroup
// o[i] = w[i] * (v * x[i]), v = 1 / sqrt(sum(x^2) / n + 1e-5) with the sum in index order.  copy_to, when
// given, receives x unchanged first (infer.rs:49: xb = x before the final norm writes x in place).
constexpr int kNormMax = 16384;
#ifndef RAMA_NORM_WAVES
#define RAMA_NORM_WAVES 4
#endif
constexpr int kNormWaves = RAMA_NORM_WAVES, kNormThreads = kNormWaves * 64;     // one wave per SIMD: the scan rounds are bound by instruction issue

// [r5] seqsum_fast.hpp's sum (the leader workgroups') for a list that sits in LDS in scan_slot layout: thread t takes terms t R .. t R + R - 1 into
// registers.  Lists of >= 256 terms only (below that the ripples of seq_sum_predict are as fast; at 600 terms they take 4.2 us against 2.2 here); false: the list is too short or too long,
// or a prediction did not hold -- the caller then takes seq_sum_predict / seq_sum_exact as before.  All NW waves call it.
#ifndef RAMA_FAST_SUM_MIN
#define RAMA_FAST_SUM_MIN 256
#endif
constexpr int kFastSumMin = RAMA_FAST_SUM_MIN;
template <int NW, int R>
__device__ __forceinline__ bool seq_sum_lds_fast_r(const float* a, int n, FastSumShared<NW>& fs, float* out) {
    float v[R];
#pragma unroll
    for (int k = 0; k < R; k++) { const int i = (int)threadIdx.x * R + k; const float t = a[scan_slot(min(i, n - 1))]; v[k] = i < n ? t : 0.0f; }
    return seq_sum_fast<NW, R>(v, fs, out);
}
template <int NW>
__device__ __forceinline__ bool seq_sum_lds_fast(const float* a, int n, FastSumShared<NW>& fs, float* out) {
    const int per = (n + 64 * NW - 1) / (64 * NW);                 // uniform
    if (n < kFastSumMin || per > 64) return false;
    seq_sum_fast_prepare<NW>(fs);
    if (per <= 8) return seq_sum_lds_fast_r<NW, 8>(a, n, fs, out);
    if (per <= 16) return seq_sum_lds_fast_r<NW, 16>(a, n, fs, out);
    if (per <= 32) return seq_sum_lds_fast_r<NW, 32>(a, n, fs, out);
    return seq_sum_lds_fast_r<NW, 64>(a, n, fs, out);
}
// [r5] the exact sequential sum of a list in LDS (scan_slot layout) by whichever form applies to its length; every thread gets it.  From kFastSumMin terms on
// seqsum_fast.hpp's form; below it, and wherever a prediction fails, the round-4 forms.  (One thread simply adding a short list in order, sixteen LDS reads
// ahead of the adds, was measured: 0.35 us + 12 ns a term -- 1.20 us at 71 terms, 1.36 at 121, against 1.12-1.16 here whatever the length: not kept.)
// The caller has a barrier behind the writes of the list.
template <int NW>
__device__ __forceinline__ float seq_sum_cascade(const float* a, int n, FastSumShared<NW>& fsn, PredShared<NW>& ps, SeqSumShared<NW>& sh) {
    float sum;
    if (!seq_sum_lds_fast<NW>(a, n, fsn, &sum)) { __syncthreads(); if (!seq_sum_predict<NW>(a, n, ps, &sum)) sum = seq_sum_exact<NW>(a, n, sh); }
    return sum;
}
// (a grid of several workgroups: vector b of a token batch, `stride` floats after the one before it)
__global__ __launch_bounds__(kNormThreads) void rmsnorm_chain_kernel(float* o, const float* x, const float* w, int n, float* copy_to, int stride = 0) {
    RAMA_NO_CONTRACT
    o += (size_t)blockIdx.x * stride; x += (size_t)blockIdx.x * stride;
    extern __shared__ __attribute__((aligned(16))) float s_sq[];
    __shared__ SeqSumShared<kNormWaves> sh;
    __shared__ PredShared<kNormWaves> ps;
    __shared__ FastSumShared<kNormWaves> fsn;
    const int tid = threadIdx.x;
    SEQ_STAMP(0);
    const bool vec = n % 4 == 0 && ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)o | (uintptr_t)copy_to) & 15) == 0) && n <= 16 * kNormThreads;
    if (vec) {     // up to 4 x 16 bytes per thread, everything requested up front
        // ([r4] really up front: the index is clamped instead of the load being conditional -- `i < n4 ? load : 0` is a branch with
        // s_waitcnt vmcnt(0) behind it, eight cache round trips one after the other; what a lane past the end reads is never used)
        const int n4 = n >> 2;
        f4 xv[4], wv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) xv[k] = reinterpret_cast<const f4*>(x)[min(tid + kNormThreads * k, n4 - 1)];
#pragma unroll
        for (int k = 0; k < 4; k++) wv[k] = reinterpret_cast<const f4*>(w)[min(tid + kNormThreads * k, n4 - 1)];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = tid + kNormThreads * k;
            if (i < n4) {
                float* d = s_sq + scan_slot(4 * i);               // 4 i .. 4 i + 3 share a 32-element stretch of slots
                d[0] = xv[k].x * xv[k].x; d[1] = xv[k].y * xv[k].y; d[2] = xv[k].z * xv[k].z; d[3] = xv[k].w * xv[k].w;
            }
        }
        __syncthreads();
        float ss;
        if (!seq_sum_lds_fast<kNormWaves>(s_sq, n, fsn, &ss)) { __syncthreads(); if (!seq_sum_predict<kNormWaves>(s_sq, n, ps, &ss)) ss = seq_sum_exact<kNormWaves>(s_sq, n, sh); }
        const float v = 1.0f / sqrtf(ss / (float)n + 1e-5f);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = tid + kNormThreads * k;
            if (i < n4) {
                if (copy_to) reinterpret_cast<f4*>(copy_to)[i] = xv[k];
                f4 r;
                r.x = wv[k].x * (v * xv[k].x); r.y = wv[k].y * (v * xv[k].y); r.z = wv[k].z * (v * xv[k].z); r.w = wv[k].w * (v * xv[k].w);
                reinterpret_cast<f4*>(o)[i] = r;
            }
        }
        SEQ_STAMP(6);
        return;
    }
    for (int i = tid; i < n; i += kNormThreads) { const float a = x[i]; s_sq[scan_slot(i)] = a * a; }
    __syncthreads();
    float ss;
    if (!seq_sum_lds_fast<kNormWaves>(s_sq, n, fsn, &ss)) { __syncthreads(); if (!seq_sum_predict<kNormWaves>(s_sq, n, ps, &ss)) ss = seq_sum_exact<kNormWaves>(s_sq, n, sh); }
    const float v = 1.0f / sqrtf(ss / (float)n + 1e-5f);
    // every thread rewrites only the indices it reads (o may alias x: infer.rs:49-50)
    for (int i = tid; i < n; i += kNormThreads) {
        const float a = x[i];
        if (copy_to) copy_to[i] = a;
        o[i] = w[i] * (v * a);
    }
}

// cpu.rs:119-125 Device::softmax over a whole view (n <= kNormMax), the oracle's order: max, exp, sequential sum, divide
__global__ __launch_bounds__(kNormThreads) void softmax_chain_kernel(float* x, int n) {
    RAMA_NO_CONTRACT
    extern __shared__ __attribute__((aligned(16))) float s_e[];
    __shared__ SeqSumShared<kNormWaves> sh;
    __shared__ PredShared<kNormWaves> ps;
    __shared__ FastSumShared<kNormWaves> fsn;
    __shared__ float red[16];
    __shared__ unsigned long long s_tab[32];
    const int tid = threadIdx.x;
    exp_tab_fill(s_tab);
    float mx = -INFINITY;
    for (int i = tid; i < n; i += kNormThreads) mx = fmaxf(mx, x[i]);
    mx = block_max(mx, red);
    for (int i = tid; i < n; i += kNormThreads) s_e[scan_slot(i)] = expf_glibc_tab(x[i] - mx, s_tab);
    __syncthreads();
    const float sum = seq_sum_cascade<kNormWaves>(s_e, n, fsn, ps, sh);
    for (int i = tid; i < n; i += kNormThreads) x[i] = s_e[scan_slot(i)] / sum;
}

// ---------------------------------------------------------------- cpu.rs:127-153 matmul on chain-order weights
enum { CEPI_STORE = 0, CEPI_RESID = 1, CEPI_QKV = 2, CEPI_SWIGLU = 3 };

struct ChainParams {
    const float* w[3];     // chain-order matrices (nmat <= 3), each [ceil(rows/16)*16, K]
    float* o[3];           // STORE: o[0]; RESID: o[0] = the product (xb2 / xb); QKV: q, k, v; SWIGLU: o[0] = hb, o[1] = hb2
    const float* x;        // [K] activations, row-major
    const float* nw;       // NORM: rmsnorm gain [K]; x is then the residual stream and the kernel normalises it itself
    float* resid;          // RESID: resid[r] += product (infer.rs:37,47)
    int K, rows, nmat;     // SWIGLU: rows = 2 * hidden (interleaved W1 | W3)
    const Ctl* ctl; int pos_val;
    const float* fr; const float* fi; int head_size;
    float* kc; float* vc;  // this layer's cache slabs [seq, dim]
    // CNORM_LEAD: workgroup 0 of the launch forms v = 1 / sqrt(sum(x^2) / K + 1e-5) (the sum in index order) and publishes it as ONE tagged word;
    // the others request their weights, then wait for it (layer_fused.hpp's put_tagged / epoch / error word)
    // ([r5] measured and not kept, profiles/r05_experiments.md: 16 copies of the word on different channels -1.8 %; the leader's compute unit kept free of
    // other workgroups and its waves at priority 3: the word is there 1.1-2 us earlier, the launch no shorter)
    unsigned long long* lead; const unsigned* epoch; unsigned long long* err;
    // half_from > 0 ([r5] one wave per row group): the row groups from this index on are walked as TWO workgroups of 8 rows each (lanes 0..31; the
    // chain-order block of 16 rows is two halves of 512 bytes).  A compute unit streams ~28 GB/s whatever the rest of the chip does, so a launch whose
    // row groups do not divide by the compute units ends when the CUs with one group more are through: llama2-7B's W1|W3 is 1 376 groups on 256 CUs,
    // six on 96 of them and five on the rest; with the last 96 groups as 192 halves it is five and a half at most.
    int half_from;
    float* xout;           // CNORM_LEAD: the leader also stores the normalised vector here (a Device::rmsnorm recorded in front of the run, rama_api.hip flush_mm)
    int lane_reduce;       // [r6] the order of cpu.rs:148 `v.reduce_add()`: LANES_PAIRWISE | LANES_STRIDED | LANES_SEQUENTIAL (ref_order.hpp; "lane_reduce")
};

// a descriptor whose inputs the compiler must take as wave-uniform (they are: kernel arguments and blockIdx)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_uniform(const void* p, unsigned bytes) {
    const unsigned long long b = (unsigned long long)(uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return make_rsrc(reinterpret_cast<const void*>((uintptr_t)(((unsigned long long)hi << 32) | lo)), (unsigned)__builtin_amdgcn_readfirstlane((int)bytes));
}
__device__ __forceinline__ f4 ld_nt_s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 2));
}

// A workgroup of W waves owns 16 rows; grid = nmat * ceil(rows / 16).  The chain of a row group is cut
// into chunks of D blocks (1 KiB of weights each); chunk c is summed by wave c mod W, which receives the 64
// running sums through LDS from the wave before it and meanwhile keeps its next chunk's D loads in flight:
// W x D KiB per group on the way at any time, one wave adding.  (W = 1: a rolling ring of D loads.)
// Measured (tools/chain_bench.hip, tools/chain_sweep.py): a CU takes ~64 KiB of loads in flight -- issuing
// 4 x 48 loads per CU lasts 4.8 us, 2 x 32 lasts 0.8 -- so deeper rings only delay the first turn; in steady state
// every geometry streams at 6.2-6.9 TB/s and what separates a launch from bytes / 6.9 TB/s is ~2.3 us of start
// (the first HBM round trip under the burst of every CU's ring) and ~2 us of tail and launch gap.
// XD = activation blocks read ahead from LDS.  Dynamic LDS: K + chain_pad_floats(W, D, XD) floats.
__host__ __device__ constexpr int chain_pad_floats(int W, int D, int XD) { return 16 * ((W + 1) * D + XD); }   // everything the x ring can touch
// time stamps of one wave for tools/chain_bench.hip only (100 MHz counter; lane 0 of wave 0 of block RAMA_CHAIN_STAMP_BLOCK)
#ifdef RAMA_CHAIN_STAMPS
__device__ unsigned long long g_chain_stamps[64];
// ... and, per workgroup, [start, staged, loop done] of its wave 0 (how evenly the groups finish: g_chain_all[3 b + 0 | 1 | 2])
__device__ unsigned long long g_chain_all[3 * 4096];
#define CHAIN_STAMP(id) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
        if (blockIdx.x == RAMA_CHAIN_STAMP_BLOCK) g_chain_stamps[id] = t_; \
        if (blockIdx.x < 4096 && ((id) == 0 || (id) == 2 || (id) == 4)) g_chain_all[3 * blockIdx.x + (id) / 2] = t_; } } while (0)
#else
#define CHAIN_STAMP(id) do { } while (0)
#endif
// NORM: cpu.rs:99-117's rmsnorm folded in -- every workgroup forms the sum of squares itself and stages w * (v * x).
//   CNORM_EXACT (parity mode, narrow models, K <= 1024): the exact sequential sum -- wave 0 ripples through the list, 0.3 us
//     per 64 elements: shorter than the launch it replaces up to ~1024 -- the bits of cpu.rs:112.
//   CNORM_TREE (tolerance mode, K <= 64 x threads): every thread sums the squares of the <= 64 elements it holds, a wave sum,
//     the waves through LDS: a fixed tree, the same bits in every workgroup, ~1e-7 relative from the sequential sum -- the
//     matvec behind it keeps the reference's rounding sequence.  ~0.3 us per workgroup while its first weights are on their
//     way, instead of a launch of its own (8 us for the exact sum at dim 4096).
// CNORM_NONE: the activations come normalised (parity mode at dim > 512: rmsnorm_chain_kernel in front).
// (Round 4 also folded the EXACT sum in for K <= 4096 -- one wave, 64 consecutive squares per lane, predicted binades, a scalar
// walk; bit-exact -- and measured it slower than the launch it replaced: every workgroup of a CU repeats ~2000 vector
// instructions, +8..10 us per matvec at llama2-7B against 9.4 us for the norm launch; profiles/r04_norm_fold_experiments.json.)
//   CNORM_LEAD ([r5] parity mode, K <= 64 x threads x ... : the host picks LR with 64 W LR >= K): ONE workgroup of the launch -- block 0, a grid
//     of groups + 1 -- forms the exact sum (seqsum_fast.hpp: ~2 us on one wave for 4096 squares) and publishes v as a tagged word; every other
//     workgroup requests x, the gain and its first ring of weights, waits for the word and stages w * (v * x) itself.  What a norm launch of its
//     own costs (its 7.8 us + a launch boundary on either side + the consumer's cold first round trip behind it) shrinks to the leader's
//     x round trip + the sum + one hand-off, with the consumers' first weights already in registers when v arrives.
enum { CNORM_NONE = 0, CNORM_EXACT = 1, CNORM_TREE = 2, CNORM_LEAD = 3 };

constexpr unsigned long long kLeadErr = 0x3100ull;                // error word: a wait for the leader's word gave up
// (the body of gemv_chain_kernel; `bid_in` = the workgroup's index.  [r5] As a PHASE of merged launches -- a consumer waiting for tagged words of
// other workgroups of its launch, a producer leaving such words, two row groups per workgroup -- it was bit-identical and slower or equal:
// profiles/r05_experiments.md, profiles/r06_lost_experiments.patch)
template <int W, int D, int XD, int EPI, int NORM = CNORM_NONE, int LR = 64>
__device__ __forceinline__ void gemv_chain_body(const ChainParams p, int bid_in) {
    RAMA_NO_CONTRACT
    CHAIN_STAMP(0);
    static_assert(D % XD == 0, "the x ring must divide the weight ring");
    extern __shared__ __attribute__((aligned(16))) float xs[];
    __shared__ float relay[64];
    constexpr int WL = W;                                         // waves of the workgroup
    const int lane = threadIdx.x & 63;
    const int wave_all = WL == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave_all;
    const int j = lane & 3, rr = lane >> 2;
    const int groups = (p.rows + 15) >> 4;
    if constexpr (NORM == CNORM_LEAD) {
        if (bid_in == 0) {      // the leader: thread t holds x[t LD .. t LD + LD) (zeros behind K: the descriptor's range check)
            // [r5] on ONE wave wherever a wave holds the whole vector (LR WL <= 64: llama2-7B's Wq|Wk|Wv launch, two waves per row group): the sum's phases
            // then need no barrier, and the word is out ~2 us earlier (8.3 -> 6 us after the leader's start)
#ifndef RAMA_LEAD_SOLO
#define RAMA_LEAD_SOLO 1
#endif
            constexpr bool SOLO = RAMA_LEAD_SOLO && WL > 1 && LR * WL <= 64;
            constexpr int WD = SOLO ? 1 : WL, LD = SOLO ? LR * WL : LR;
            if (SOLO && wave_all != 0) return;
            FastSumShared<WD>& fs = *reinterpret_cast<FastSumShared<WD>*>(xs);        // (host: the dynamic LDS holds it)
#ifdef RAMA_CHAIN_STAMPS
            if (threadIdx.x == 0) g_chain_stamps[40] = __builtin_amdgcn_s_memrealtime();
#endif
            const unsigned ep = *p.epoch;
            const __amdgpu_buffer_rsrc_t rxl = make_rsrc_uniform(p.x, (unsigned)p.K * 4u);
            seq_sum_fast_prepare<WD>(fs);
            float a[LD];
#pragma unroll
            for (int u = 0; u < LD / 4; u++) {
                const f4 xv = ld_c(rxl, ((unsigned)threadIdx.x * (unsigned)LD + 4u * (unsigned)u) * 4u);
                a[4 * u] = xv.x * xv.x; a[4 * u + 1] = xv.y * xv.y; a[4 * u + 2] = xv.z * xv.z; a[4 * u + 3] = xv.w * xv.w;
            }
#ifdef RAMA_CHAIN_STAMPS
            { float t_ = 0.0f;
#pragma unroll
              for (int k = 0; k < LD; k++) t_ += a[k];
              asm volatile("" :: "v"(t_)); }
            if (threadIdx.x == 0) g_chain_stamps[41] = __builtin_amdgcn_s_memrealtime();
#endif
            float ss;
            if (!seq_sum_fast<WD, LD>(a, fs, &ss)) {      // (uniform) the prediction did not hold: the plain loop over the squares
                __syncthreads();
#pragma unroll
                for (int k = 0; k < LD; k++) { const int i = (int)threadIdx.x * LD + k; if (i < p.K) xs[i] = a[k]; }
                __syncthreads();
                if (threadIdx.x == 0) { float s_ = 0.0f; for (int i = 0; i < p.K; i++) s_ = s_ + xs[i]; xs[p.K] = s_; }
                __syncthreads();
                ss = xs[p.K];
            }
            const float v = 1.0f / sqrtf(ss / (float)p.K + 1e-5f);
            if (threadIdx.x == 0) put_tagged(p.lead, v, ep);
#ifdef RAMA_CHAIN_STAMPS
            if (threadIdx.x == 0) g_chain_stamps[42] = __builtin_amdgcn_s_memrealtime();
#endif
            if (p.xout) {      // cpu.rs:113-116 o[i] = w[i] * (v * x[i]), as the row groups form it for themselves
                for (int i = threadIdx.x; i < p.K; i += WD * 64) p.xout[i] = p.nw[i] * (v * p.x[i]);
            }
            return;
        }
    }
    const int bid = NORM == CNORM_LEAD ? bid_in - 1 : bid_in;
    // (the quotient comes out of the vector ALU: without readfirstlane everything derived from it -- the buffer
    // descriptors above all -- counts as divergent and every load turns into a waterfall loop)
    int half = -1, bid_g = bid;
    // ([r6] measured and removed: EVERY group of Wo as two half groups -- two one-wave workgroups per compute unit on different SIMDs, a ring of 32 half
    // blocks each: 14.10 against 13.89 us per launch, 209.0 against 209.8 tok/s; profiles/r06_experiments.md 1)
    if (W == 1 && p.half_from > 0 && bid >= p.half_from) { half = (bid - p.half_from) & 1; bid_g = p.half_from + ((bid - p.half_from) >> 1); }      // (uniform)
    const int m = __builtin_amdgcn_readfirstlane(bid_g / groups), g = __builtin_amdgcn_readfirstlane(bid_g - m * groups);
    const float* Wm = m == 0 ? p.w[0] : (m == 1 ? p.w[1] : p.w[2]);
    const int nblk = p.K >> 4;
    const int nchunk = (nblk + D - 1) / D;
    const bool lane_ok = half < 0 || lane < 32;                   // a half group: 8 rows x 4 chains
    const unsigned lane16 = half < 0 ? (unsigned)lane * 16u : (lane < 32 ? (unsigned)half * 512u + (unsigned)lane * 16u : kOOB);
    static_assert(D % 16 == 0, "a chunk is made of whole 16-block stretches");
    const int n16 = (nblk + 15) >> 4;
    // the descriptor of stretch q (16 blocks = 16 KiB of the group's stream): as many bytes as the row still has there,
    // so that blocks behind the end of the row are dropped by the range check with no select in the loop
    auto stretch = [&](int q) {
        const int left = min(max(nblk - q * 16, 0), 16);
        return make_rsrc_uniform(Wm + ((size_t)g * (size_t)nblk + (size_t)min(q, n16) * 16) * 256, (unsigned)left * 1024u);
    };
    unsigned vo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) vo[k] = lane16 + (unsigned)k * 4096u;
    // The activations are requested FIRST (they come from L2), then this wave's first chunk of weights (from HBM):
    // a wave's loads complete in order, so x asked for behind the weights would wait for the whole first HBM round
    // trip before the staging below could even begin.  ([r5] behind a leader's norm the other order -- ring first, so that ~1000 workgroups asking
    // for the same 32 KiB of x and gain do not keep the leader's own read of x waiting -- measured equal: 39.1 us either way.)
    const __amdgpu_buffer_rsrc_t rx = make_rsrc_uniform(p.x, (unsigned)p.K * 4u);
    const int n4 = p.K >> 2;
    // (whatever of x does not fit the registers follows behind the weights.  [r5] 24 x 16 bytes per thread up front for
    // the residual products -- llama2-7B's W2 reads 11 008 floats on 128 threads, the last 2 816 in a second round trip -- measured: 33.70 against 33.77 us)
    constexpr int T = WL * 64, XU = 16;                            // (host: K <= 16 XU T floats where a norm is folded in)
    f4 xa[XU];
#pragma unroll
    for (int u = 0; u < XU; u++) xa[u] = ld_c(rx, ((int)threadIdx.x + T * u) < n4 ? (unsigned)((int)threadIdx.x + T * u) * 16u : kOOB);
    f4 ga[NORM != CNORM_NONE ? XU : 1];
    if constexpr (NORM != CNORM_NONE) {
        const __amdgpu_buffer_rsrc_t rg = make_rsrc_uniform(p.nw, (unsigned)p.K * 4u);
#pragma unroll
        for (int u = 0; u < XU; u++) ga[u] = ld_c(rg, ((int)threadIdx.x + T * u) < n4 ? (unsigned)((int)threadIdx.x + T * u) * 16u : kOOB);
    }
    f4 wr[D];
#pragma unroll
    for (int h = 0; h < D / 16; h++) {
        const __amdgpu_buffer_rsrc_t r0 = stretch(wave * (D / 16) + h);
#pragma unroll
        for (int u = 0; u < 16; u++) wr[h * 16 + u] = ld_nt(r0, vo[u >> 2] + (unsigned)(u & 3) * 1024u);
    }
    const int row = lane_ok ? 16 * g + (half > 0 ? 8 : 0) + rr : p.rows + 16;      // (lanes a half group does not use: behind every matrix)
    // epilogue operands
    float xold = 0.0f, rc = 1.0f, rs = 0.0f;
    unsigned long long tabv = 0;                                  // SWIGLU: the exponential's table word lane & 31 (requested with the ring, see the epilogue)
    int pos = 0;
    if (EPI == CEPI_RESID) {
        if (j == 0 && row < p.rows) xold = p.resid[row];
    } else if (EPI == CEPI_SWIGLU) {
        tabv = kExp2fTab[lane & 31];
    } else if (EPI == CEPI_QKV) {
        pos = p.ctl ? p.ctl->pos : p.pos_val;
        if (m < 2) {
            const int i = ((row & ~1) % p.head_size) >> 1;       // infer.rs:15-16: table row pos, pair i of the head
            rc = p.fr[(size_t)pos * (p.head_size >> 1) + i];
            rs = p.fi[(size_t)pos * (p.head_size >> 1) + i];
        }
    }
    CHAIN_STAMP(1);
    if constexpr (NORM == CNORM_TREE) {     // x <- w * (v * x) with the sum of squares as a fixed tree (host: K <= 64 T floats, all of x is in xa)
        float ssl = 0.0f;
#pragma unroll
        for (int u = 0; u < XU; u++) ssl = ssl + ((xa[u].x * xa[u].x + xa[u].y * xa[u].y) + (xa[u].z * xa[u].z + xa[u].w * xa[u].w));
        ssl = wave_sum(ssl);
        if (W > 1) {
            __shared__ float wss[W];
            if (lane == 0) wss[wave] = ssl;
            __syncthreads();
            ssl = wss[0];
#pragma unroll
            for (int w_ = 1; w_ < W; w_++) ssl = ssl + wss[w_];
        }
        const float v = 1.0f / sqrtf(ssl / (float)p.K + 1e-5f);
#pragma unroll
        for (int u = 0; u < XU; u++) {
            xa[u].x = ga[u].x * (v * xa[u].x); xa[u].y = ga[u].y * (v * xa[u].y);
            xa[u].z = ga[u].z * (v * xa[u].z); xa[u].w = ga[u].w * (v * xa[u].w);
        }
    }
    if constexpr (NORM == CNORM_LEAD) {     // x <- w * (v * x) with the leader's v: every wave watches the word itself (no barrier), bounded
        float v = 0.0f;
        {
            const unsigned ep = *p.epoch;
            unsigned long long word = 0;
            long spins = 0;
            while (true) {
                if (lane == 0) word = __hip_atomic_load(p.lead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                word = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(word >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)word);
                if ((unsigned)(word >> 32) == ep) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if ((spins & 255) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (spins > (1L << 22)) { if (lane == 0) __hip_atomic_store(p.err, kLeadErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            v = __uint_as_float((unsigned)word);
        }
        CHAIN_STAMP(5);
#pragma unroll
        for (int u = 0; u < XU; u++) {
            xa[u].x = ga[u].x * (v * xa[u].x); xa[u].y = ga[u].y * (v * xa[u].y);
            xa[u].z = ga[u].z * (v * xa[u].z); xa[u].w = ga[u].w * (v * xa[u].w);
        }
    }
    if constexpr (NORM == CNORM_EXACT) {     // x <- w * (v * x), v = 1 / sqrt(sum(x^2) / K + 1e-5) with the sum in index order (host: K <= 64 T floats, all of x is in xa)
        __shared__ PredShared<W> nps;
        float* sq = xs + p.K + chain_pad_floats(W, D, XD);       // squares, scan_slot layout
#pragma unroll
        for (int u = 0; u < XU; u++) {
            const int i = (int)threadIdx.x + T * u;
            if (i < n4) {
                float* d = sq + scan_slot(4 * i);                 // 4 i .. 4 i + 3 share a 32-element stretch of slots
                d[0] = xa[u].x * xa[u].x; d[1] = xa[u].y * xa[u].y; d[2] = xa[u].z * xa[u].z; d[3] = xa[u].w * xa[u].w;
            }
        }
        __syncthreads();
        const float ss = seq_sum_ripples<W>(sq, p.K, nps);
        const float v = 1.0f / sqrtf(ss / (float)p.K + 1e-5f);
#pragma unroll
        for (int u = 0; u < XU; u++) {
            xa[u].x = ga[u].x * (v * xa[u].x); xa[u].y = ga[u].y * (v * xa[u].y);
            xa[u].z = ga[u].z * (v * xa[u].z); xa[u].w = ga[u].w * (v * xa[u].w);
        }
    }
    // activations -> LDS in chain order: xs[16 s + 4 j + t] = x[16 s + 4 t + j]
    {
#pragma unroll
        for (int u = 0; u < XU; u++) {
            // no branch around the stores (it would serialise the loads): a lane past the end holds zeros and
            // drops them onto the zero padding behind x
            const int i = (int)threadIdx.x + T * u;
            float* d = xs + (i < n4 ? 16 * (i >> 2) + (i & 3) : p.K);
            d[0] = xa[u].x; d[4] = xa[u].y; d[8] = xa[u].z; d[12] = xa[u].w;
        }
        for (int i0 = (int)threadIdx.x + T * XU; i0 < n4; i0 += T * 8) {      // rows longer than 64 T floats: the rest, behind the weights
            f4 a[8];
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] = ld_c(rx, (i0 + T * u) < n4 ? (unsigned)(i0 + T * u) * 16u : kOOB);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + T * u;
                float* d = xs + (i < n4 ? 16 * (i >> 2) + (i & 3) : p.K);
                d[0] = a[u].x; d[4] = a[u].y; d[8] = a[u].z; d[12] = a[u].w;
            }
        }
    }
    // zeros behind x: D blocks that may be multiplied (by weights the range check zeroed) + the x ring's read-ahead
    for (int i = p.K + (int)threadIdx.x; i < p.K + chain_pad_floats(W, D, XD); i += T) xs[i] = 0.0f;
    __syncthreads();
    // A lone wave issues one instruction per several cycles whatever its kind, and only ONE wave of the group can
    // be adding at any time, so the adding wave's instruction count per block is what bounds a launch with one
    // group per CU.  Hence: no guards and no selects in the loop (blocks behind the end of the row are dropped by
    // the descriptor's range check and read as 0; LDS holds zeros behind x, so such a block adds 0 * 0 = +0 to a
    // sum that can never be -0: the bits stay), chunk-relative immediate offsets, and -- with W > 1 -- the
    // PRODUCTS are formed outside the turn: a wave multiplies its next chunk in place (wr *= x, rounded: the
    // reference's separate multiply) while the wave before it is adding, so a turn is 4 dependent adds and one
    // reload per block.
    CHAIN_STAMP(2);
    const f4* xq = reinterpret_cast<const f4*>(xs) + j;          // block s: xq[4 s]
    float v = 0.0f;
    if (W == 1) {
        f4 xr[XD];
#pragma unroll
        for (int u = 0; u < XD; u++) xr[u] = xq[4 * u];
        for (int c = 0; c < nchunk; c++) {
            __amdgpu_buffer_rsrc_t rn[D / 16];                    // the chunk reloaded behind this one, stretch by stretch
#pragma unroll
            for (int h = 0; h < D / 16; h++) rn[h] = stretch((c + 1) * (D / 16) + h);
            const f4* xc = xq + 4 * c * D;
#pragma unroll
            for (int u = 0; u < D; u++) {
                const f4 xv = xr[u % XD];
                xr[u % XD] = xc[4 * (u + XD)];
                const f4 wv = wr[u];
                v = v + wv.x * xv.x;
                v = v + wv.y * xv.y;
                v = v + wv.z * xv.z;
                v = v + wv.w * xv.w;
                wr[u] = ld_nt(rn[u >> 4], vo[(u & 15) >> 2] + (unsigned)(u & 3) * 1024u);
                __builtin_amdgcn_sched_barrier(0);                // keep the ring rolling: one reload per consumed block
            }
        }
    } else {
        auto premultiply = [&](int c) {                           // wr <- wr * x for chunk c (this wave's next turn)
            const f4* xc = xq + 4 * c * D;
#pragma unroll
            for (int u = 0; u < D; u++) {
                const f4 xv = xc[4 * u];
                wr[u].x = wr[u].x * xv.x; wr[u].y = wr[u].y * xv.y; wr[u].z = wr[u].z * xv.z; wr[u].w = wr[u].w * xv.w;
                // (deep rings: the LDS reads of eight blocks at a time -- all 32 at once are 128 registers beside the ring's 128)
                if constexpr (D > 16) { if ((u & 7) == 7) __builtin_amdgcn_sched_barrier(0); }
            }
        };
        if (wave == 0) premultiply(0);
        CHAIN_STAMP(3);
        for (int c = 0; c < nchunk; c++) {
            if ((c % W) == wave) {                                // uniform: my turn
                if (c < 24) CHAIN_STAMP(8 + 2 * c);
                if (c > 0) v = relay[lane];
                __amdgpu_buffer_rsrc_t rn[D / 16];
#pragma unroll
                for (int h = 0; h < D / 16; h++) rn[h] = stretch((c + W) * (D / 16) + h);
#pragma unroll
                for (int u = 0; u < D; u++) {
                    const f4 pv = wr[u];
                    v = v + pv.x;
                    v = v + pv.y;
                    v = v + pv.z;
                    v = v + pv.w;
                    wr[u] = ld_nt(rn[u >> 4], vo[(u & 15) >> 2] + (unsigned)(u & 3) * 1024u);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (c + 1 < nchunk) relay[lane] = v;
                if (c < 24) CHAIN_STAMP(9 + 2 * c);
            } else if (((c + 1) % W) == wave && c + 1 < nchunk) { // my turn is next: the products, while the wave before me adds
                premultiply(c + 1);
            }
            __syncthreads();
        }
    }
    CHAIN_STAMP(4);
    if (W > 1 && ((nchunk - 1) % W) != wave) return;             // the wave that summed the last chunk finishes the rows
    // cpu.rs:148 reduce_add of the quad's four chains, in the order the host selected: every lane of the quad ends with the row's bits
    // (a uniform branch at the very end of the launch)
    float d;
    if (p.lane_reduce == LANES_STRIDED) {                         // (v0 + v2) + (v1 + v3): both adds are commutative
        const float t2 = v + dpp_mov<0x4E>(v);                    // quad_perm [2,3,0,1]
        d = t2 + dpp_mov<0xB1>(t2);                               // quad_perm [1,0,3,2]
    } else if (p.lane_reduce == LANES_SEQUENTIAL) {               // ((v0 + v1) + v2) + v3: every lane walks the quad front to back
        d = ((dpp_mov<0x00>(v) + dpp_mov<0x55>(v)) + dpp_mov<0xAA>(v)) + dpp_mov<0xFF>(v);      // quad_perm [0,0,0,0] .. [3,3,3,3]
    } else {                                                      // (v0 + v1) + (v2 + v3)
        const float t2 = v + dpp_mov<0xB1>(v);                    // quad_perm [1,0,3,2]
        d = t2 + dpp_mov<0x4E>(t2);                               // quad_perm [2,3,0,1]
    }
    if (EPI == CEPI_STORE) {
        if (j == 0 && row < p.rows) { float* o = m == 0 ? p.o[0] : (m == 1 ? p.o[1] : p.o[2]); o[row] = d; }      // (nmat > 1: a run of Device::matmul calls as one launch)
    } else if (EPI == CEPI_RESID) {
        if (j == 0 && row < p.rows) { p.o[0][row] = d; p.resid[row] = xold + d; }
    } else if (EPI == CEPI_QKV) {
        const float other = __shfl_xor(d, 4);                    // the pair's other row (neighbouring quad)
        const float a = (rr & 1) ? other : d, b = (rr & 1) ? d : other;
        float out = d;
        if (m < 2) out = (rr & 1) ? a * rs + b * rc : a * rc - b * rs;      // cpu.rs:87-96
        if (j == 0 && row < p.rows) {
            float* o = m == 0 ? p.o[0] : (m == 1 ? p.o[1] : p.o[2]);
            o[row] = out;
            if (m == 1) p.kc[(size_t)pos * p.rows + row] = out;              // infer.rs:32
            else if (m == 2) p.vc[(size_t)pos * p.rows + row] = out;         // infer.rs:33
        }
    } else {   // CEPI_SWIGLU: even row = W1 row i, odd row = W3 row i
        const float h3 = __shfl_xor(d, 4);
        // the table of expf in LDS, written by every wave for itself from words it asked for in front of the ring (all waves write the same 256 bytes;
        // a wave's LDS operations keep their order): a lookup in constant memory HERE is a trip to L2 or further at the very end of the workgroup
        __shared__ unsigned long long s_tab[32];
        s_tab[lane & 31] = tabv;
        __builtin_amdgcn_wave_barrier();
        if (j == 0 && !(rr & 1) && row < p.rows) {
            const float sl = d * (1.0f / (1.0f + expf_glibc_tab(-d, s_tab)));       // cpu.rs:56
            p.o[0][row >> 1] = sl * h3;                                        // cpu.rs:59-64
            p.o[1][row >> 1] = h3;
        }
    }
}

// TAG: nothing but a name -- the same body under another symbol, so that a kernel trace tells two uses of one geometry apart ([r6] TAG 1 = the SQUARE
// residual product, Wo; round 5's traces showed Wo and W2 as one row with their mean)
template <int W, int D, int XD, int EPI, int NORM = CNORM_NONE, int LR = 64, int TAG = 0>
__global__ __launch_bounds__(W * 64) void gemv_chain_kernel(ChainParams p) {
    gemv_chain_body<W, D, XD, EPI, NORM, LR>(p, (int)blockIdx.x);
}

// ---------------------------------------------------------------- token batches in the reference's order (parity-mode prefill)
// O[t][r] = the chain-order product of row r with activation vector t, for T = 4 TPW tokens at once (TPW = 1, 2, 4, 8): the same
// per-output sequence of roundings as gemv_chain_kernel (k ascending in each of the four lane chains k = j mod 4, separate
// multiply and add, (v0 + v1) + (v2 + v3)), with every weight read ONCE for the T tokens.
//   A workgroup of 4 waves owns 16 rows; wave w carries the sums of tokens w TPW .. w TPW + TPW - 1, lane = (chain j =
//   lane / 16, row = lane % 16).  The chain is walked in chunks of 8 blocks (128 floats of K): 8 KiB of weights (chain
//   order: a straight copy) and 512 bytes of activations per token (transposed into chain order on the way) go
//   HBM / L2 -> registers -> LDS, two chunks in flight in registers, two LDS slots, ONE barrier per chunk and no branch
//   in the loop; from LDS a lane reads a block's weights (16 bytes) and ONE activation per token; the multiplies take
//   the activation of step i from lane i of the 16-lane row through DPP.
//   Measured at llama2-7B, 32 positions per pass (8 per wave; rocprofv3): QKV 111 us, Wo / W2 93 (mean), W1|W3 204 per
//   layer.  Bound by the latency of the dependent adds (four independent pair-chains per wave), not by HBM (3.5 TB/s)
//   and not by the instruction count.
//   Blocks behind the end of a row: weights dropped by the descriptor's range check, activations stored as zeros: + (0 * 0).
struct GemmChainParams {
    const float* w[3];      // chain-order matrices (nmat <= 3), each [ceil(rows / 16) * 16, K]
    int nmat, K, rows;      // SWIGLU: rows = 2 * hidden (interleaved W1 | W3)
    const float* x; int xstride;          // activations [n_tok][xstride], row-major
    float* o[3]; int ostride;             // STORE / RESID: o[0][t][row] (RESID: may be NULL); QKV: q = o[0]; SWIGLU: hb = o[0][t][row / 2]
    float* resid; int rstride;            // RESID: resid[t][row] += product (infer.rs:37,47)
    int n_tok;
    int pos0; const float* fr; const float* fi; int head_size; float* kc; float* vc;      // QKV: token t sits at pos0 + t; this layer's cache slabs
    const SeqSlot* seqs; size_t layer_off;   // QKV, independent sequences: token t at seqs[t].pos, its caches at seqs[t].kc / .vc + layer_off
    int lane_reduce;        // [r6] as ChainParams::lane_reduce
};
constexpr int kGcWaves = 4, kGcThreads = kGcWaves * 64;
constexpr int kGcBlocks = 8;             // blocks of 16 floats per chunk (8 KiB of weights; 32 KiB of LDS per workgroup at 16 tokens: four workgroups per CU)
constexpr int kGcSets = 2;               // chunks in flight in registers
__host__ __device__ constexpr size_t gemm_chain_lds_bytes(int tpw) { return 2 * ((size_t)kGcBlocks * 1024 + (size_t)(kGcWaves * tpw) * kGcBlocks * 64); }

// a * w with a taken from lane I of the caller's DPP row (16 lanes): one v_mul_f32 with the row broadcast on its source
// operand -- the rounded product of the reference's separate multiply.  (As assembly: hipcc would pack the multiplies of two
// tokens into v_pk_mul_f32, which takes no DPP operand, and pay for it with a broadcast move and shuffles per product.)
template <int I>
__device__ __forceinline__ float mul_row(float a, float w) {
    float r;
    if (I == 0) asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(a), "v"(w));
    else if (I == 1) asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(a), "v"(w));
    else if (I == 2) asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(a), "v"(w));
    else asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(a), "v"(w));
    return r;
}

template <int TPW, int EPI>
__global__ __launch_bounds__(kGcThreads) void gemm_chain_kernel(GemmChainParams p) {
    RAMA_NO_CONTRACT
    constexpr int T = kGcWaves * TPW;                            // tokens of a pass
    constexpr int WN = kGcBlocks * 64 / kGcThreads;               // 16-byte pieces of weights per thread and chunk
    constexpr int PPT = kGcBlocks * 4;                            // 16-byte pieces of one token's activations per chunk
    constexpr int XF = (T * PPT + kGcThreads - 1) / kGcThreads;   // ... of activations per thread and chunk
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    f4* wl = reinterpret_cast<f4*>(gsm);                         // [2][kGcBlocks * 64] weights, chain order
    float* xl = gsm + 2 * kGcBlocks * 256;                       // [2][T][kGcBlocks * 16] activations, chain order per token
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int groups = (p.rows + 15) >> 4;
    const int m = __builtin_amdgcn_readfirstlane(blockIdx.x / groups), g = blockIdx.x - m * groups;
    const float* Wm = m == 0 ? p.w[0] : (m == 1 ? p.w[1] : p.w[2]);
    const int nblk = p.K >> 4;
    const int nchunk = (nblk + kGcBlocks - 1) / kGcBlocks;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc_uniform(p.x, (unsigned)((size_t)p.n_tok * (size_t)p.xstride * 4u));
    f4 wreg[kGcSets][WN], xreg[kGcSets][XF];

    // loads of chunk c into register set S (c >= nchunk: nothing but zeros, no memory touched)
    auto issue = [&](int c, f4 (&wr)[WN], f4 (&xr)[XF]) {
        const int left = min(max(nblk - c * kGcBlocks, 0), kGcBlocks);
        const __amdgpu_buffer_rsrc_t rw = make_rsrc_uniform(Wm + ((size_t)g * (size_t)nblk + (size_t)min(c, nchunk) * kGcBlocks) * 256, (unsigned)left * 1024u);
#pragma unroll
        for (int n = 0; n < WN; n++) wr[n] = ld_nt(rw, (unsigned)(tid + kGcThreads * n) * 16u);
#pragma unroll
        for (int n = 0; n < XF; n++) {
            const int q = tid + kGcThreads * n, t = q / PPT, i = q % PPT;    // token, 16-byte piece of its chunk
            const int k = c * (kGcBlocks * 16) + i * 4;
            xr[n] = ld_c(rx, (t < p.n_tok && k < p.K) ? (unsigned)(((size_t)t * (size_t)p.xstride + (size_t)k) * 4u) : kOOB);
        }
    };
    // registers -> LDS slot: weights as they are, activations xs[16 s + 4 j + i] = x[16 s + 4 i + j] (gemv_chain_kernel's order)
    auto stage = [&](int slot, const f4 (&wr)[WN], const f4 (&xr)[XF]) {
#pragma unroll
        for (int n = 0; n < WN; n++) wl[slot * (kGcBlocks * 64) + tid + kGcThreads * n] = wr[n];
#pragma unroll
        for (int n = 0; n < XF; n++) {
            const int q = tid + kGcThreads * n, t = q / PPT, i = q % PPT;
            if (T * PPT % kGcThreads == 0 || q < T * PPT) {
                float* d = xl + ((size_t)slot * T + t) * (kGcBlocks * 16) + 16 * (i >> 2) + (i & 3);
                d[0] = xr[n].x; d[4] = xr[n].y; d[8] = xr[n].z; d[12] = xr[n].w;
            }
        }
    };
    // In the sums a lane is (chain j = lane / 16, row = lane % 16): the 16 lanes of a DPP row share a chain, hence the
    // four activations of a block and token.  With every lane reading all four (16 bytes) the LDS array bounded the
    // kernel (1 KiB per token, block and wave for 64 bytes of data); so a lane reads ONE of them -- step i = row % 4 --
    // and the multiply of step i takes it from lane i of the row through DPP (row_newbcast:i on the source operand).
    const int jj = lane >> 4, r16 = lane & 15;
    typedef float f2 __attribute__((ext_vector_type(2)));
    float acc1 = 0.0f;                                           // TPW == 1
    f2 acc2[TPW >= 2 ? TPW / 2 : 1];                             // TPW >= 2: tokens (2 tp, 2 tp + 1)
#pragma unroll
    for (int tp = 0; tp < (TPW >= 2 ? TPW / 2 : 1); tp++) acc2[tp] = f2{0.0f, 0.0f};
    // (one basic block: with a branch in here hipcc sinks the adds of a whole chunk behind it)
    auto compute = [&](int slot) {
        const f4* wq = wl + slot * (kGcBlocks * 64) + (r16 * 4 + jj);          // the chain-order copy keeps lane = row * 4 + chain
        const float* xb = xl + ((size_t)slot * T + (size_t)wave * TPW) * (kGcBlocks * 16) + 4 * jj + (r16 & 3);
        // the operands of block u + 1 are read while block u is summed; no further ahead (registers), two sets in turn
        f4 w0, w1;
        float xa[TPW], xc[TPW];
        auto fetch = [&](int u, f4& wv, float (&xv)[TPW]) {
            wv = wq[u * 64];
#pragma unroll
            for (int tt = 0; tt < TPW; tt++) xv[tt] = xb[tt * (kGcBlocks * 16) + 16 * u];
        };
        auto sums = [&](const f4& wv, const float (&xs1)[TPW]) {
            if constexpr (TPW == 1) {
                float v = acc1;
                v = v + mul_row<0>(xs1[0], wv.x);
                v = v + mul_row<1>(xs1[0], wv.y);
                v = v + mul_row<2>(xs1[0], wv.z);
                v = v + mul_row<3>(xs1[0], wv.w);
                acc1 = v;
            } else {
#pragma unroll
                for (int tp = 0; tp < TPW / 2; tp++) {
                    f2 v = acc2[tp];
                    v = v + f2{mul_row<0>(xs1[2 * tp], wv.x), mul_row<0>(xs1[2 * tp + 1], wv.x)};
                    v = v + f2{mul_row<1>(xs1[2 * tp], wv.y), mul_row<1>(xs1[2 * tp + 1], wv.y)};
                    v = v + f2{mul_row<2>(xs1[2 * tp], wv.z), mul_row<2>(xs1[2 * tp + 1], wv.z)};
                    v = v + f2{mul_row<3>(xs1[2 * tp], wv.w), mul_row<3>(xs1[2 * tp + 1], wv.w)};
                    acc2[tp] = v;
                }
            }
            // the sums are pinned HERE: left alone, hipcc sinks a whole chunk's adds behind the barrier that ends the chunk
            // (the loop body has exec-masked blocks from the guarded loads) -- 128 products wait in registers and the
            // adds run as one dependent burst at the top of the next step
            if constexpr (TPW == 1) asm volatile("" : "+v"(acc1));
            else {
#pragma unroll
                for (int tp = 0; tp < TPW / 2; tp++) asm volatile("" : "+v"(acc2[tp]));
            }
        };
        fetch(0, w0, xa);
#pragma unroll
        for (int u = 0; u < kGcBlocks; u += 2) {
            fetch(u + 1, w1, xc);
            __builtin_amdgcn_sched_barrier(0);
            sums(w0, xa);
            __builtin_amdgcn_sched_barrier(0);
            if (u + 2 < kGcBlocks) fetch(u + 2, w0, xa);
            __builtin_amdgcn_sched_barrier(0);
            sums(w1, xc);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // chunk c is computed from slot c % 2; before that, chunk c + 1 (in registers since step c - 1) goes to the other
    // slot and chunk c + 3 is requested into the registers it leaves.  No branch in the loop: the steps behind the last
    // chunk (the count is rounded up to the unroll) load nothing, stage zeros and add + (0 * 0) -- with a condition
    // around a step hipcc's waitcnt pass forgets what is in flight at the join and drains the loads just issued
    // (vmcnt(0): measured 5 us per chunk instead of 1).
    issue(0, wreg[0], xreg[0]);
    issue(1, wreg[1], xreg[1]);
    stage(0, wreg[0], xreg[0]);
    issue(2, wreg[0], xreg[0]);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < nchunk; c += 2) {
        stage(1, wreg[1], xreg[1]);                              // chunk c + 1
        __builtin_amdgcn_sched_barrier(0);
        issue(c + 3, wreg[1], xreg[1]);
        __builtin_amdgcn_sched_barrier(0);
        compute(0);                                              // chunk c
        __syncthreads();
        stage(0, wreg[0], xreg[0]);                              // chunk c + 2
        __builtin_amdgcn_sched_barrier(0);
        issue(c + 4, wreg[0], xreg[0]);
        __builtin_amdgcn_sched_barrier(0);
        compute(1);                                              // chunk c + 1
        __syncthreads();
    }
    // epilogues of gemv_chain_kernel, per token (the four chains of a row sit 16 lanes apart here, its neighbour row one lane)
    const int row = 16 * g + r16;
#pragma unroll
    for (int tt = 0; tt < TPW; tt++) {
        const int t = wave * TPW + tt;
        const float v = TPW == 1 ? acc1 : acc2[tt / 2][tt & 1];
        // cpu.rs:148 reduce_add: the four chains of a row sit 16 lanes apart (chain j on lanes 16 j ..)
        float d;
        if (p.lane_reduce == LANES_STRIDED) {                     // (v0 + v2) + (v1 + v3)
            const float t2 = v + __shfl_xor(v, 32);
            d = t2 + __shfl_xor(t2, 16);
        } else if (p.lane_reduce == LANES_SEQUENTIAL) {           // ((v0 + v1) + v2) + v3
            const int l16 = (int)(threadIdx.x & 15);
            d = ((__shfl(v, l16) + __shfl(v, l16 + 16)) + __shfl(v, l16 + 32)) + __shfl(v, l16 + 48);
        } else {                                                  // (v0 + v1) + (v2 + v3)
            const float t2 = v + __shfl_xor(v, 16);
            d = t2 + __shfl_xor(t2, 32);
        }
        const float other = __shfl_xor(d, 1);                    // the neighbouring row (RoPE pair / W3 row)
        if (t >= p.n_tok) continue;                              // uniform per wave
        if (EPI == CEPI_STORE) {
            if (jj == 0 && row < p.rows) p.o[0][(size_t)t * p.ostride + row] = d;
        } else if (EPI == CEPI_RESID) {
            if (jj == 0 && row < p.rows) {
                if (p.o[0]) p.o[0][(size_t)t * p.ostride + row] = d;
                float* xr_ = p.resid + (size_t)t * p.rstride + row;
                *xr_ = *xr_ + d;
            }
        } else if (EPI == CEPI_QKV) {
            const int pos = p.seqs ? p.seqs[t].pos : p.pos0 + t;
            float* kcl = p.seqs ? p.seqs[t].kc + p.layer_off : p.kc;
            float* vcl = p.seqs ? p.seqs[t].vc + p.layer_off : p.vc;
            float out = d;
            if (m < 2) {
                const int i = ((row & ~1) % p.head_size) >> 1;   // infer.rs:15-16
                const float rc = p.fr[(size_t)pos * (p.head_size >> 1) + i], rs = p.fi[(size_t)pos * (p.head_size >> 1) + i];
                const float a = (r16 & 1) ? other : d, b = (r16 & 1) ? d : other;
                out = (r16 & 1) ? a * rs + b * rc : a * rc - b * rs;         // cpu.rs:87-96
            }
            if (jj == 0 && row < p.rows) {
                if (m == 0) p.o[0][(size_t)t * p.ostride + row] = out;
                else if (m == 1) kcl[(size_t)pos * p.rows + row] = out;      // infer.rs:32
                else vcl[(size_t)pos * p.rows + row] = out;                  // infer.rs:33
            }
        } else {   // CEPI_SWIGLU: even row = W1 row i, odd row = W3 row i
            if (jj == 0 && !(r16 & 1) && row < p.rows) {
                const float sl = d * (1.0f / (1.0f + expf_glibc(-d)));       // cpu.rs:56
                p.o[0][(size_t)t * p.ostride + (row >> 1)] = sl * other;      // cpu.rs:59-64
            }
        }
    }
}

// X[t] = token_embedding_table[tokens[t]] (infer.rs:13), rows of a token batch
__global__ void embed_rows_kernel(float* X, const float* emb, const int* tokens, int n_tok, int dim) {
    const int t = blockIdx.y;
    if (t >= n_tok) return;
    const size_t base = (size_t)tokens[t] * dim;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < dim; k += gridDim.x * blockDim.x) X[(size_t)t * dim + k] = emb[base + k];
}

// ---------------------------------------------------------------- cpu.rs:23-52 multi_head_attention
// One workgroup of NW waves per head.
//  * Scores: thread t owns timestep t; its q.k dot runs over the head in index order.  A lane reading its own
//    512-byte cache row 16 bytes at a time is 64 different lines per wave instruction (70 us per layer at 1900
//    timesteps), so a wave stages 64 rows x 32 floats at a time through a PRIVATE piece of LDS: coalesced 16-byte
//    loads (8 lanes per row), stored with rows 36 floats apart, then every lane reads its own row back
//    (conflict-free) and continues its chain.  No workgroup barrier inside; the next piece's loads are in flight
//    while this one is summed.
//  * Softmax: max, glibc expf, the sequential sum (seq_sum_predict / seq_sum_exact), divide.
//  * Values: xb[i] = sum_t att[t] * v[t][i], t ascending, is one chain per head column: pos + 1 dependent adds that
//    nothing can shorten.  So everything else leaves the chain: all threads load tiles of kAttTile cache rows and
//    store the PRODUCTS att[t] * v[t][i] (rounded, as the reference's `a * vi`) in LDS, two tiles in turn (one
//    barrier per tile), and thread i < head_size only adds them up in row order.
constexpr int kAttTile = 64;
constexpr int kAttPiece = 32;                 // floats of a row per staged piece
constexpr int kAttStride = kAttPiece + 4;     // row pitch of a staged piece (16-byte aligned, conflict-free row reads)
__host__ __device__ constexpr size_t attn_chain_region_floats(int head_size, int nw) {
    return (size_t)(nw * 64 * kAttStride) > (size_t)(2 * kAttTile * head_size) ? (size_t)(nw * 64 * kAttStride) : (size_t)(2 * kAttTile * head_size);
}
__host__ __device__ constexpr size_t attn_chain_lds_floats(int head_size, int seq_len, int nw) {
    return (size_t)((head_size + 3) & ~3) + (size_t)seq_len + (size_t)(seq_len >> 5) + 4 + (size_t)((seq_len + 3) & ~3) + attn_chain_region_floats(head_size, nw);
}
// the value tiles are loaded with 8 x 16 bytes per thread
__host__ __device__ constexpr bool attn_chain_fits(int head_size, int nw) { return kAttTile * (head_size / 4) <= (nw >= 16 ? 4 : 8) * 64 * nw; }

// (the body of attention_chain_kernel: head h of token y.  lds_seq = the timesteps the LDS arrays are laid out for.
// [r5] forms of this body that ran inside merged launches -- xb handed to Wo groups of the same launch, q | k | v taken as tagged words from
// Wq|Wk|Wv groups of the same launch, the whole parity-mode stage as one launch -- were built, are bit-identical and measured slower or equal;
// they left the product in round 6: profiles/r06_lost_experiments.patch puts them back, profiles/r05_experiments.md holds the measurements)
// (a cache base taken from a SeqSlot in memory is a pointer of unknown address space to the compiler: its loads become FLAT loads, which count in lgkmcnt
// as well -- every wait for an LDS operation then also waits for the cache rows on their way.  The cache is global memory: say so.)
typedef const __attribute__((address_space(1))) f4* gf4p;
__device__ __forceinline__ gf4p gptr4(const float* p) { return (gf4p)reinterpret_cast<const f4*>(p); }
template <int NW>
__device__ __forceinline__ void attention_chain_body(RefAttnParams p, int h, int y, int lds_seq) {
    RAMA_NO_CONTRACT
    constexpr int T = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ SeqSumShared<NW> sh;
    __shared__ PredShared<NW> ps;
    __shared__ FastSumShared<NW> fsn;
    __shared__ float red[16];
    __shared__ unsigned long long s_tab[32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    exp_tab_fill(s_tab);
    const int pos = p.seqs ? p.seqs[y].pos : (p.ctl ? p.ctl->pos : p.pos_val) + y;
    if (p.seqs) { p.kc = p.seqs[y].kc + p.layer_off; p.vc = p.seqs[y].vc + p.layer_off; }
    p.q += (size_t)y * p.tok_stride; p.xb += (size_t)y * p.tok_stride;
    if (p.att) p.att += (size_t)y * p.att_stride;
    const int hs = p.head_size, hs4 = hs >> 2;
    float* s_q = sm;                                              // [hs]
    float* s_p = sm + ((hs + 3) & ~3);                            // [lds_seq] the probabilities, unskewed (read 4 at a time)
    float* s_att = s_p + ((lds_seq + 3) & ~3);                    // [scan_slot(lds_seq)] scores -> exponentials
    // score staging, then the product tiles: the next 16-byte boundary, as an OFFSET into the LDS array (rounding the pointer through an integer made
    // every access behind it a FLAT load or store -- slower than a ds_ operation, and counted in vmcnt as well: [r5])
    const int region_off = (((hs + 3) & ~3) + ((lds_seq + 3) & ~3) + lds_seq + (lds_seq >> 5) + 4 + 3) & ~3;
    float* region = sm + region_off;
    const size_t col = (size_t)h * hs;
    SEQ_STAMP(8);
    // ([r5] measured and removed: q, a thread's first key row and the first two value tiles requested in one go at the top of the PLAIN launch, straight-line
    // so that the waits are counted (q: vmcnt(48); the scores: all but the value tiles).  At position 70 (tools/seqsum_bench): values 2.84 -> 2.40 us, but
    // scores 2.20 -> 3.20 -- every thread then requests a row and 48 requests per thread stand in front of the first product; the launch 8.72 -> 9.36 us.)
    for (int i = tid; i < hs; i += T) s_q[i] = p.q[col + i];
    // xb[i] = sum_t att[t] * v[t][i], t ascending (cpu.rs:43-49).
    // (16 waves: 1 024 threads cover a tile of head size 256 with four loads each -- and have 128 registers, which eight loads per tile buffer
    // and a 32-load key batch overran by 56: [r5])
    constexpr int U = NW >= 16 ? 4 : 8;
    const int tile4 = kAttTile * hs4;                             // f4 elements of a tile
    int er[U], ec[U];                                             // tile element e = tid + u T: its row and 16-byte column (one division each, here)
#pragma unroll
    for (int u = 0; u < U; u++) { const int e = tid + u * T; er[u] = e / hs4; ec[u] = e - er[u] * hs4; }
    auto vissue = [&](int t0, f4 (&vr)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            // (rows behind pos are clamped: their products are written, never added -- a load under a condition is a branch with
            // `s_waitcnt vmcnt(0)` behind it, one cache round trip per load instruction)
            const int tr = min(t0 + er[u], pos), c4 = ec[u];
            vr[u] = __builtin_nontemporal_load(gptr4(p.vc + (size_t)tr * p.dim + col) + c4);
        }
    };
    f4 va[U], vb[U];                                              // two tiles on their way while a third is added up
    __syncthreads();
    const float scale_div = sqrtf((float)hs);
    const f4* q4 = reinterpret_cast<const f4*>(s_q);
    if (hs % kAttPiece == 0 && pos >= 1024) {       // long contexts: staged pieces, two of them in flight per wave
        float* stage = region + wave * (64 * kAttStride);
        const int npiece = hs / kAttPiece;
        const int lrow = lane >> 3, lc4 = lane & 7;               // loader role: row (+ 8 u) and 16-byte column of the piece
        const int ngroup = (pos + 64) >> 6;                       // groups of 64 timesteps
        const int nstep = ((ngroup - wave + NW - 1) / NW) * npiece;          // this wave's (group, piece) steps, in order
        auto load_step = [&](int st, f4 (&d)[8]) {                // step st = piece st % npiece of group wave + (st / npiece) NW
            // (steps behind the wave's last and rows behind pos are clamped -- their scores are never stored: a load under a condition
            // is a branch with `s_waitcnt vmcnt(0)` behind it, one cache round trip per load instruction)
            const int sc = min(st, max(nstep - 1, 0));
            const int g = wave + (sc / npiece) * NW, pc = sc % npiece;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int t = min(g * 64 + u * 8 + lrow, pos);
                d[u] = __builtin_nontemporal_load(gptr4(p.kc + (size_t)t * p.dim + col + pc * kAttPiece) + lc4);
            }
        };
        constexpr int NB = NW >= 16 ? 1 : 2;                      // staged pieces in flight per wave (16 waves: 128 registers -- one)
        f4 na[8], nb[NB == 2 ? 8 : 1];
        load_step(0, na);
        if constexpr (NB == 2) load_step(1, nb);
        float acc = 0.0f;
        auto consume = [&](int st, f4 (&d)[8]) {
#pragma unroll
            for (int u = 0; u < 8; u++) *reinterpret_cast<f4*>(stage + (u * 8 + lrow) * kAttStride + 4 * lc4) = d[u];
            load_step(st + NB, d);                                // this buffer's next use
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the piece is in LDS (a wave's LDS operations finish in order)
            const int pc = st % npiece;
            const f4* row = reinterpret_cast<const f4*>(stage + lane * kAttStride);
            f4 kk[8];
#pragma unroll
            for (int i = 0; i < 8; i++) kk[i] = row[i];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const f4 qq = q4[pc * 8 + i];
                acc = acc + qq.x * kk[i].x; acc = acc + qq.y * kk[i].y; acc = acc + qq.z * kk[i].z; acc = acc + qq.w * kk[i].w;
            }
            if (pc + 1 == npiece) {
                const int t = (wave + (st / npiece) * NW) * 64 + lane;
                if (t <= pos) s_att[scan_slot(t)] = acc / scale_div;
                acc = 0.0f;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // rows read before the next piece overwrites them
            __builtin_amdgcn_wave_barrier();
        };
        for (int st = 0; st < nstep; st += NB) {                  // uniform per wave
            consume(st, na);
            if constexpr (NB == 2) { if (st + 1 < nstep) consume(st + 1, nb); }
        }
    } else {
        for (int t = tid; t <= pos; t += T) {                     // a few rounds of timesteps: straight from the cache, a row's loads together
            const gf4p k4 = gptr4(p.kc + (size_t)t * p.dim + col);
            float acc = 0.0f;
            int i = 0;
            // batches of 32, 16, 8, 4 x 16 bytes, each requested at once ([r4]: 16 and 8 are new -- head size 64 took four round
            // trips of 4, head size 48 three)
            auto batch = [&](auto nb) {
                constexpr int NB = decltype(nb)::value;
                for (; i + NB <= hs4; i += NB) {
                    f4 kk[NB];
#pragma unroll
                    for (int u = 0; u < NB; u++) kk[u] = k4[i + u];
#pragma unroll
                    for (int u = 0; u < NB; u++) {
                        const f4 qq = q4[i + u];
                        acc = acc + qq.x * kk[u].x; acc = acc + qq.y * kk[u].y; acc = acc + qq.z * kk[u].z; acc = acc + qq.w * kk[u].w;
                    }
                }
            };
            if constexpr (NW < 16) batch(std::integral_constant<int, 32>{});
            batch(std::integral_constant<int, 16>{});
            batch(std::integral_constant<int, 8>{});
            batch(std::integral_constant<int, 4>{});
            for (; i < hs4; i++) {
                const f4 kk = k4[i], qq = q4[i];
                acc = acc + qq.x * kk.x; acc = acc + qq.y * kk.y; acc = acc + qq.z * kk.z; acc = acc + qq.w * kk.w;
            }
            s_att[scan_slot(t)] = acc / scale_div;
        }
    }
    __syncthreads();
    SEQ_STAMP(9);
    // softmax_num (cpu.rs:187-192)
    float mx = -INFINITY;
    for (int t = tid; t <= pos; t += T) mx = fmaxf(mx, s_att[scan_slot(t)]);
    mx = block_max(mx, red);
    for (int t = tid; t <= pos; t += T) s_att[scan_slot(t)] = expf_glibc_tab(s_att[scan_slot(t)] - mx, s_tab);
    __syncthreads();
    SEQ_STAMP(10);
    const float sum = seq_sum_cascade<NW>(s_att, pos + 1, fsn, ps, sh);
    SEQ_STAMP(11);
    for (int t = tid; t <= pos; t += T) {
        const float a = s_att[scan_slot(t)] / sum;
        s_p[t] = a;
        if (p.att) p.att[(size_t)h * p.seq_len + t] = a;
    }
    float acc = 0.0f;
    SEQ_STAMP(12);
    vissue(0, va);
    if (pos >= kAttTile) vissue(kAttTile, vb);                   // (uniform: 64 KiB of requests a tile cost the CU ~0.25 us wherever they stand)
    __syncthreads();                                              // the probabilities are final; the staging region is free
    auto vtile = [&](int t0, int buf, f4 (&vr)[U]) {
        float* tile = region + buf * (kAttTile * hs);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + u * T;
            if (e < tile4) {
                const float a = s_p[min(t0 + er[u], pos)];
                const f4 vv = vr[u];
                f4 pr;
                pr.x = a * vv.x; pr.y = a * vv.y; pr.z = a * vv.z; pr.w = a * vv.w;     // cpu.rs:48 `a * vi`, rounded
                *reinterpret_cast<f4*>(tile + 4 * e) = pr;
            }
        }
        if (t0 + 2 * kAttTile <= pos) vissue(t0 + 2 * kAttTile, vr);      // (uniform)
        __syncthreads();                                          // this tile is written; the other one (read last round) is free again
        if (tid < hs) {
            const int nt = min(kAttTile, pos + 1 - t0);
            int r = 0;
            for (; r + 16 <= nt; r += 16) {                       // 16 reads in flight, then the adds in row order
                float v16[16];
#pragma unroll
                for (int u = 0; u < 16; u++) v16[u] = tile[(r + u) * hs + tid];
#pragma unroll
                for (int u = 0; u < 16; u++) acc = acc + v16[u];
            }
            for (; r < nt; r++) acc = acc + tile[r * hs + tid];
        }
    };
    for (int t0 = 0; t0 <= pos; t0 += 2 * kAttTile) {
        vtile(t0, 0, va);
        if (t0 + kAttTile <= pos) vtile(t0 + kAttTile, 1, vb);    // uniform
    }
    if (tid < hs) p.xb[col + tid] = acc;
    SEQ_STAMP(13);
}
template <int NW>
__global__ __launch_bounds__(NW * 64) void attention_chain_kernel(RefAttnParams p) {
    attention_chain_body<NW>(p, (int)blockIdx.x, (int)blockIdx.y, p.seq_len);
}

// ---------------------------------------------------------------- the same attention spread over the chip (long contexts)
// One workgroup per head leaves 224 CUs idle, and a CU takes ~64 KiB of loads in flight: at 1 900 timesteps K and V
// arrive at ~1 TB/s and the launch lasts 81 us.  From position 128 on ([r4]; round 3: 1 024) the phases are launches of their own
// -- the last two as ONE since round 4 (attn_softmax_values_chain_kernel below; "attn_fv" = 0 keeps them apart):
//   attn_scores_chain_kernel   grid (heads x groups of 64 timesteps), one wave each: the staged q.k chains -> att (scores)
//   attn_softmax_chain_kernel  grid heads: max, glibc expf, the exact sequential sum, divide -> att (probabilities)
//   attn_values_chain_kernel   grid (heads x slices of 32 head columns): product tiles of 256 rows x 32 columns, the 32
//                              chains of the slice add them in row order
// Same operations in the same order per output as attention_chain_kernel: the same bits.
// one group of 64 timesteps of one head on ONE wave: stage = 64 x kAttStride floats, s_q = head_size floats of this wave's own; emit(t, score) for t <= pos
template <bool ALL4 = true, class Emit>
__device__ __forceinline__ void attn_scores_group(const RefAttnParams& p, int h, int g, int lane, int pos, float* stage, float* s_q, Emit emit) {
    RAMA_NO_CONTRACT
    const int hs = p.head_size, npiece = hs / kAttPiece;
    const size_t col = (size_t)h * hs;
    const int lrow = lane >> 3, lc4 = lane & 7;
    auto load_piece = [&](int pc, f4 (&d)[8]) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            // (rows behind pos and pieces behind the head are clamped, their scores never stored: a load under a condition is a branch
            // with `s_waitcnt vmcnt(0)` behind it)
            const int t = min(g * 64 + u * 8 + lrow, pos), pcc = min(pc, npiece - 1);
            d[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.kc + (size_t)t * p.dim + col + pcc * kAttPiece) + lc4);
        }
    };
    // [r4] head sizes up to 128: all four pieces (32 KB a wave) are requested at once -- with two in flight the third piece's round
    // trip began only when the first had been consumed, and a launch is little more than round trips (8.9 -> 6 us at 1 900 timesteps)
    f4 na[8], nb[8], nc[ALL4 ? 8 : 1], nd[ALL4 ? 8 : 1];          // (ALL4 = false: two pieces in flight -- 64 registers less, for a kernel that is short of them)
    load_piece(0, na);
    load_piece(1, nb);
    const bool all4 = ALL4 && npiece <= 4;                        // uniform
    if constexpr (ALL4) { if (all4) { load_piece(2, nc); load_piece(3, nd); } }
    for (int i = lane; i < hs; i += 64) s_q[i] = p.q[col + i];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const f4* q4 = reinterpret_cast<const f4*>(s_q);
    const float scale_div = sqrtf((float)hs);
    float acc = 0.0f;
    auto consume = [&](int pc, f4 (&d)[8], bool more) {
#pragma unroll
        for (int u = 0; u < 8; u++) *reinterpret_cast<f4*>(stage + (u * 8 + lrow) * kAttStride + 4 * lc4) = d[u];
        if (more) load_piece(pc + 2, d);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const f4* row = reinterpret_cast<const f4*>(stage + lane * kAttStride);
        f4 kk[8];
#pragma unroll
        for (int i = 0; i < 8; i++) kk[i] = row[i];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const f4 qq = q4[pc * 8 + i];
            acc = acc + qq.x * kk[i].x; acc = acc + qq.y * kk[i].y; acc = acc + qq.z * kk[i].z; acc = acc + qq.w * kk[i].w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };
    if (all4) {
        if constexpr (ALL4) {
            consume(0, na, false);
            if (npiece > 1) consume(1, nb, false);
            if (npiece > 2) consume(2, nc, false);
            if (npiece > 3) consume(3, nd, false);
        }
    } else {
        for (int pc = 0; pc < npiece; pc += 2) {
            consume(pc, na, true);
            if (pc + 1 < npiece) consume(pc + 1, nb, true);
        }
    }
    const int t = g * 64 + lane;
    if (t <= pos) emit(t, acc / scale_div);
}
__global__ __launch_bounds__(64) void attn_scores_chain_kernel(RefAttnParams p) {
    __shared__ __attribute__((aligned(16))) float stage[64 * kAttStride];
    __shared__ __attribute__((aligned(16))) float s_q[256];
    const int h = blockIdx.x, g = blockIdx.y, lane = threadIdx.x;
    const int pos = p.ctl ? p.ctl->pos : p.pos_val;
    if (g * 64 > pos) return;                                     // uniform
    float* out = p.sc + (size_t)h * p.seq_len;
    attn_scores_group(p, h, g, lane, pos, stage, s_q, [&](int t, float v) { out[t] = v; });
}

constexpr int kSoftWaves = 4;
__global__ __launch_bounds__(kSoftWaves * 64) void attn_softmax_chain_kernel(RefAttnParams p) {
    RAMA_NO_CONTRACT
    constexpr int T = kSoftWaves * 64;
    extern __shared__ __attribute__((aligned(16))) float s_att[];      // [scan_slot(seq_len)]
    __shared__ SeqSumShared<kSoftWaves> sh;
    __shared__ PredShared<kSoftWaves> ps;
    __shared__ FastSumShared<kSoftWaves> fsn;
    __shared__ float red[16];
    __shared__ unsigned long long s_tab[32];
    const int h = blockIdx.x, tid = threadIdx.x;
    exp_tab_fill(s_tab);
    const int pos = p.ctl ? p.ctl->pos : p.pos_val;
    float* att = p.att + (size_t)h * p.seq_len;
    float mx = -INFINITY;
    for (int t = tid; t <= pos; t += T) { const float a = att[t]; s_att[scan_slot(t)] = a; mx = fmaxf(mx, a); }
    mx = block_max(mx, red);
    for (int t = tid; t <= pos; t += T) s_att[scan_slot(t)] = expf_glibc_tab(s_att[scan_slot(t)] - mx, s_tab);
    __syncthreads();
    const float sum = seq_sum_cascade<kSoftWaves>(s_att, pos + 1, fsn, ps, sh);
    for (int t = tid; t <= pos; t += T) att[t] = s_att[scan_slot(t)] / sum;
}

constexpr int kValCols = 16, kValRows = 256, kValWaves = 5;            // a slice's tile: 256 rows x 16 columns = 16 KiB of products; 4 loading waves + the chain wave
constexpr int kValStride = kValRows + 4;                               // a column's 256 products in a row (+4: the 16-byte reads of 8 columns cover all banks)
// [r4] 20.8 -> 13 us per launch at 1 900 timesteps, the same operations in the same order per output:
//  * every cache load is unconditional (rows behind pos are clamped, their weight set to 0 when the product is formed): a load under
//    a condition compiles to a branch with `s_waitcnt vmcnt(0)` behind it -- one cache round trip per load INSTRUCTION;
//  * four tiles' cache rows are in flight instead of two (a slice reads 64-byte pieces 16 KiB apart: 2 us until they are there);
//  * the product tile is stored COLUMN by column (a column's 256 rows contiguous), so the chain lane of column c reads four rows'
//    products with one 16-byte LDS read: one add in ~10 cycles (its own latency: 8.3) instead of a 4-byte read + an add in 16;
//  * wave 0 only adds (lanes 0..15: the slice's 16 chains); waves 1..4 load, multiply and store -- the chain of tile k + 1 starts the
//    moment tile k's is through.
__global__ __launch_bounds__(kValWaves * 64) void attn_values_chain_kernel(RefAttnParams p) {
    RAMA_NO_CONTRACT
    constexpr int T = (kValWaves - 1) * 64, U = kValRows * (kValCols / 4) / T;      // 4 x 16 bytes per loading thread and tile
    __shared__ __attribute__((aligned(16))) float tile[2][kValCols * kValStride];
    const int h = blockIdx.x, sl = blockIdx.y, tid = (int)threadIdx.x - 64;        // loading threads 0..255; the chain wave: -64..-1
    const bool chain = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) == 0;
    const int pos = p.ctl ? p.ctl->pos : p.pos_val;
    const size_t col = (size_t)h * p.head_size + (size_t)sl * kValCols;
    const float* att = p.att + (size_t)h * p.seq_len;
    SEQ_STAMP(20);
    if (chain) {
        const int lane = threadIdx.x;
        float acc = 0.0f;
        int buf = 0;
        for (int t0 = 0; t0 <= pos; t0 += kValRows, buf ^= 1) {
            __syncthreads();                                      // tile t0 is written (and the other buffer, read last round, is free again)
            if (lane < kValCols) {
                // the chain itself: one dependent add per row.  The next 16 products (four 16-byte reads) are fetched from LDS while
                // the current 16 are added (two register sets in turn)
                const int nt = min(kValRows, pos + 1 - t0);
                const f4* tb = reinterpret_cast<const f4*>(&tile[buf][lane * kValStride]);
                f4 w0[4], w1[4];
                auto rd = [&](int r, f4 (&w)[4]) {                // (rows behind the tile are clamped: read, never added)
#pragma unroll
                    for (int u = 0; u < 4; u++) w[u] = tb[min(r / 4 + u, kValRows / 4 - 1)];
                };
                auto ad = [&](const f4 (&w)[4]) {
#pragma unroll
                    for (int u = 0; u < 4; u++) { acc = acc + w[u].x; acc = acc + w[u].y; acc = acc + w[u].z; acc = acc + w[u].w; }
                };
                int r = 0;
                rd(0, w0);
                for (; r + 32 <= nt; r += 32) {
                    rd(r + 16, w1);
                    __builtin_amdgcn_sched_barrier(0);
                    ad(w0);
                    __builtin_amdgcn_sched_barrier(0);
                    rd(r + 32, w0);
                    __builtin_amdgcn_sched_barrier(0);
                    ad(w1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float* ts = &tile[buf][lane * kValStride];
                for (; r < nt; r++) acc = acc + ts[r];
            }
            SEQ_STAMP(21 + min(t0 / kValRows, 11));
        }
        if (lane < kValCols) p.xb[col + lane] = acc;
        SEQ_STAMP(33);
        return;
    }
    // tile element e = tid + u T: row e / (kValCols / 4), 16-byte column e % (kValCols / 4)
    f4 v0[U], v1[U], v2[U], v3[U];
    float a0[U], a1[U], a2[U], a3[U];
    auto vissue = [&](int t0, f4 (&vr)[U], float (&ar)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + u * T, r = e / (kValCols / 4), c4 = e % (kValCols / 4);
            const int tr = min(t0 + r, pos);                      // rows behind pos: row `pos`, weight 0 (vtile)
            vr[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.vc + (size_t)tr * p.dim + col) + c4);
            ar[u] = att[tr];
        }
    };
    vissue(0, v0, a0);
    vissue(kValRows, v1, a1);
    vissue(2 * kValRows, v2, a2);
    vissue(3 * kValRows, v3, a3);
    auto vtile = [&](int t0, int buf, f4 (&vr)[U], float (&ar)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + u * T, r = e / (kValCols / 4), c4 = e % (kValCols / 4);
            const float a = t0 + r <= pos ? ar[u] : 0.0f;
            float* d = &tile[buf][(4 * c4) * kValStride + r];
            d[0] = a * vr[u].x; d[kValStride] = a * vr[u].y; d[2 * kValStride] = a * vr[u].z; d[3 * kValStride] = a * vr[u].w;     // cpu.rs:48 `a * vi`, rounded
        }
        vissue(t0 + 4 * kValRows, vr, ar);
        __syncthreads();                                          // this tile is written: the chain wave takes it; the other one is free once it arrives here again
    };
    // (the loaders run one tile ahead of the chain wave at most: tile k + 2 overwrites the buffer of tile k, which the chain wave has
    // left when it arrives at the barrier that releases tile k + 1)
    for (int t0 = 0; t0 <= pos; t0 += 4 * kValRows) {
        vtile(t0, 0, v0, a0);
        if (t0 + kValRows <= pos) vtile(t0 + kValRows, 1, v1, a1);              // uniform
        if (t0 + 2 * kValRows <= pos) vtile(t0 + 2 * kValRows, 0, v2, a2);
        if (t0 + 3 * kValRows <= pos) vtile(t0 + 3 * kValRows, 1, v3, a3);
    }
}

// [r4] The softmax and the value chains of long contexts as ONE launch: grid (heads x slices of 16 columns), 4 waves.  Every slice
// workgroup of a head repeats the head's softmax (max, glibc expf, the exact sequential sum: ~5 us of LDS work, the same bits in all 8
// of them) while its first four tiles of value rows are on their way -- the launch between (3 us), the round trip of the
// probabilities through memory and the value launch's 2.8 us until its first tile is there fall away; the weights are read from
// LDS.  Then as attn_values_chain_kernel: wave 0 adds (lanes 0..15), waves 1..3 load, multiply and store tiles of 192 rows.
constexpr int kFvRows = 192, kFvStride = kFvRows + 4, kFvWaves = 4;
#ifndef RAMA_FV_SOFT
#define RAMA_FV_SOFT 8
#endif
constexpr int kFvSoftWaves = RAMA_FV_SOFT;        // [r5] waves of the softmax phase (waves kFvWaves.. leave after it): its loops are a few elements per thread, each with its latency in full
__host__ __device__ constexpr size_t attn_fused_values_lds_floats(int seq_len) { return (size_t)seq_len + ((size_t)seq_len >> 5) + 4 + (((size_t)seq_len + 3) & ~(size_t)3); }
// ([r5] the scores formed by extra waves of the same launch -- ONE launch for the whole spread attention -- is bit-identical and 0.3-1.3 us slower per
// layer: profiles/r05_experiments.md section 12, profiles/r06_lost_experiments.patch)
static_assert(kFvSoftWaves >= kFvWaves && kFvSoftWaves <= 8, "the value chain's assembly names v200..v247: at most 8 waves per workgroup (256 VGPRs a wave)");
__device__ __forceinline__ void attn_softmax_values_chain_body(RefAttnParams p) {
    RAMA_NO_CONTRACT
    constexpr int TS = kFvSoftWaves * 64;                          // softmax: all waves
    constexpr int T = (kFvWaves - 1) * 64, U = kFvRows * (kValCols / 4) / T;      // products: 4 x 16 bytes per loading thread and tile
    extern __shared__ __attribute__((aligned(16))) float fv_sm[];
    __shared__ __attribute__((aligned(16))) float tile[2][kValCols * kFvStride];
    __shared__ SeqSumShared<kFvSoftWaves> sh;
    __shared__ PredShared<kFvSoftWaves> ps;
    __shared__ FastSumShared<kFvSoftWaves> fsn;
    __shared__ float red[16];
    __shared__ unsigned long long s_tab[32];
    exp_tab_fill(s_tab);
    float* s_att = fv_sm;                                          // [scan_slot(seq_len)] scores -> exponentials
    float* s_p = fv_sm + p.seq_len + (p.seq_len >> 5) + 4;         // [seq_len] the probabilities, unskewed
    const int h = blockIdx.x, sl = blockIdx.y, tid0 = threadIdx.x, tid = (int)threadIdx.x - 64;      // loading threads 0..191; the chain wave: -64..-1
    const bool chain = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) == 0;
    const int pos = p.ctl ? p.ctl->pos : p.pos_val;
    const size_t col = (size_t)h * p.head_size + (size_t)sl * kValCols;
    // the scores come from p.sc, the probabilities go to p.att: DIFFERENT buffers (the host sees to it) -- the eight slice workgroups of a
    // head all read the scores, and nothing orders the workgroups of one launch, so none of them may write what the others read
    const float* scores = p.sc + (size_t)h * p.seq_len;
    float* att = p.att + (size_t)h * p.seq_len;
    // the scores first, staged and their maximum taken BEFORE the value rows are asked for ([r5]; round 4 asked for both at once: behind the join of
    // `if (loading wave)` the compiler no longer knows how many loads follow the scores in the queue -- 16 or none -- and waited for everything, value
    // rows included: 2.1 us in front of the first barrier.  The scores are one round trip of 0.4 us, and the rows are not needed for 5 us)
    constexpr int kSc = 2048 / TS;                                 // scores per thread in flight at once (2 048 positions; a loop for the rest)
    float sc[kSc];
    f4 v0[U], v1[U], v2[U], v3[U];
    auto vissue = [&](int t0, f4 (&vr)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = max(tid, 0) + u * T, r = e / (kValCols / 4), c4 = e % (kValCols / 4);
            const int tr = min(t0 + r, pos);                      // rows behind pos: row `pos`, weight 0 (vtile)
            vr[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.vc + (size_t)tr * p.dim + col) + c4);
        }
    };
#pragma unroll
    for (int k = 0; k < kSc; k++) sc[k] = scores[min(tid0 + k * TS, pos)];
    SEQ_STAMP(40);
    // softmax_num (cpu.rs:187-192), as attn_softmax_chain_kernel
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < kSc; k++) { const int t = tid0 + k * TS; if (t <= pos) { s_att[scan_slot(t)] = sc[k]; mx = fmaxf(mx, sc[k]); } }
    for (int t = tid0 + kSc * TS; t <= pos; t += TS) { const float a = scores[t]; s_att[scan_slot(t)] = a; mx = fmaxf(mx, a); }
    // (tiles behind pos are not asked for -- uniform branches: a tile's requests cost the CU's address unit ~0.2 us wherever they stand)
    if (!chain && tid < T) { vissue(0, v0); if (kFvRows <= pos) vissue(kFvRows, v1); if (2 * kFvRows <= pos) vissue(2 * kFvRows, v2); if (3 * kFvRows <= pos) vissue(3 * kFvRows, v3); }     // uniform per wave
    SEQ_STAMP(41);
    mx = block_max(mx, red);
    SEQ_STAMP(42);
    for (int t = tid0; t <= pos; t += TS) s_att[scan_slot(t)] = expf_glibc_tab(s_att[scan_slot(t)] - mx, s_tab);
    __syncthreads();
    SEQ_STAMP(43);
    const float sum = seq_sum_cascade<kFvSoftWaves>(s_att, pos + 1, fsn, ps, sh);
    SEQ_STAMP(44);
    for (int t = tid0; t <= pos; t += TS) {
        const float a = s_att[scan_slot(t)] / sum;
        s_p[t] = a;
        if (sl == 0) att[t] = a;                                   // the probabilities as the launches before this one left them (tests read them)
    }
    __syncthreads();
    SEQ_STAMP(45);
    if (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) >= kFvWaves) return;      // (a barrier counts the waves that are still there)
    if (chain) {
        const int lane = threadIdx.x;
        float acc = 0.0f;
        int buf = 0;
        for (int t0 = 0; t0 <= pos; t0 += kFvRows, buf ^= 1) {
            __syncthreads();                                      // tile t0 is written (and the other buffer, read last round, is free again)
            if (lane < kValCols) {
                // [r5] the chain itself as one block of assembly: a dependent add every ~8 cycles, with the LDS reads of the rows 32 ahead (four
                // 16-byte reads per 16 rows, three register sets in turn) in the adds' shadows -- compiled C++ keeps reads and adds apart (10.6
                // cycles an add).  Whole groups of 16 rows: the rows behind pos hold +0 (vtile), which leaves a sum as it is; the reads that run
                // ahead of the tile's end fetch words that are never added.
                const int nt = min(kFvRows, pos + 1 - t0);
                unsigned a = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)&tile[buf][lane * kFvStride];
                int nb = (nt + 15) >> 4;
                asm volatile(
                    "ds_read_b128 v[200:203], %[a] offset:0\n"
                    "ds_read_b128 v[204:207], %[a] offset:16\n"
                    "ds_read_b128 v[208:211], %[a] offset:32\n"
                    "ds_read_b128 v[212:215], %[a] offset:48\n"
                    "ds_read_b128 v[216:219], %[a] offset:64\n"
                    "ds_read_b128 v[220:223], %[a] offset:80\n"
                    "ds_read_b128 v[224:227], %[a] offset:96\n"
                    "ds_read_b128 v[228:231], %[a] offset:112\n"
                    "1:\n"
                    "s_waitcnt lgkmcnt(4)\n"
                    "v_add_f32 %[acc], %[acc], v200\n"
                    "ds_read_b128 v[232:235], %[a] offset:128\n"
                    "v_add_f32 %[acc], %[acc], v201\n"
                    "v_add_f32 %[acc], %[acc], v202\n"
                    "s_sub_u32 %[nb], %[nb], 1\n"
                    "v_add_f32 %[acc], %[acc], v203\n"
                    "v_add_f32 %[acc], %[acc], v204\n"
                    "ds_read_b128 v[236:239], %[a] offset:144\n"
                    "v_add_f32 %[acc], %[acc], v205\n"
                    "v_add_f32 %[acc], %[acc], v206\n"
                    "s_cmp_eq_u32 %[nb], 0\n"
                    "v_add_f32 %[acc], %[acc], v207\n"
                    "v_add_f32 %[acc], %[acc], v208\n"
                    "ds_read_b128 v[240:243], %[a] offset:160\n"
                    "v_add_f32 %[acc], %[acc], v209\n"
                    "v_add_f32 %[acc], %[acc], v210\n"
                    "v_add_f32 %[acc], %[acc], v211\n"
                    "v_add_f32 %[acc], %[acc], v212\n"
                    "ds_read_b128 v[244:247], %[a] offset:176\n"
                    "v_add_f32 %[acc], %[acc], v213\n"
                    "v_add_f32 %[acc], %[acc], v214\n"
                    "v_add_f32 %[acc], %[acc], v215\n"
                    "s_cbranch_scc1 2f\n"
                    "s_waitcnt lgkmcnt(4)\n"
                    "v_add_f32 %[acc], %[acc], v216\n"
                    "ds_read_b128 v[200:203], %[a] offset:192\n"
                    "v_add_f32 %[acc], %[acc], v217\n"
                    "v_add_f32 %[acc], %[acc], v218\n"
                    "s_sub_u32 %[nb], %[nb], 1\n"
                    "v_add_f32 %[acc], %[acc], v219\n"
                    "v_add_f32 %[acc], %[acc], v220\n"
                    "ds_read_b128 v[204:207], %[a] offset:208\n"
                    "v_add_f32 %[acc], %[acc], v221\n"
                    "v_add_f32 %[acc], %[acc], v222\n"
                    "s_cmp_eq_u32 %[nb], 0\n"
                    "v_add_f32 %[acc], %[acc], v223\n"
                    "v_add_f32 %[acc], %[acc], v224\n"
                    "ds_read_b128 v[208:211], %[a] offset:224\n"
                    "v_add_f32 %[acc], %[acc], v225\n"
                    "v_add_f32 %[acc], %[acc], v226\n"
                    "v_add_f32 %[acc], %[acc], v227\n"
                    "v_add_f32 %[acc], %[acc], v228\n"
                    "ds_read_b128 v[212:215], %[a] offset:240\n"
                    "v_add_f32 %[acc], %[acc], v229\n"
                    "v_add_f32 %[acc], %[acc], v230\n"
                    "v_add_f32 %[acc], %[acc], v231\n"
                    "s_cbranch_scc1 2f\n"
                    "s_waitcnt lgkmcnt(4)\n"
                    "v_add_f32 %[acc], %[acc], v232\n"
                    "ds_read_b128 v[216:219], %[a] offset:256\n"
                    "v_add_f32 %[acc], %[acc], v233\n"
                    "v_add_f32 %[acc], %[acc], v234\n"
                    "s_sub_u32 %[nb], %[nb], 1\n"
                    "v_add_f32 %[acc], %[acc], v235\n"
                    "v_add_f32 %[acc], %[acc], v236\n"
                    "ds_read_b128 v[220:223], %[a] offset:272\n"
                    "v_add_f32 %[acc], %[acc], v237\n"
                    "v_add_f32 %[acc], %[acc], v238\n"
                    "s_cmp_eq_u32 %[nb], 0\n"
                    "v_add_f32 %[acc], %[acc], v239\n"
                    "v_add_f32 %[acc], %[acc], v240\n"
                    "ds_read_b128 v[224:227], %[a] offset:288\n"
                    "v_add_f32 %[acc], %[acc], v241\n"
                    "v_add_f32 %[acc], %[acc], v242\n"
                    "v_add_f32 %[acc], %[acc], v243\n"
                    "v_add_f32 %[acc], %[acc], v244\n"
                    "ds_read_b128 v[228:231], %[a] offset:304\n"
                    "v_add_f32 %[acc], %[acc], v245\n"
                    "v_add_u32 %[a], 0xc0, %[a]\n"
                    "v_add_f32 %[acc], %[acc], v246\n"
                    "v_add_f32 %[acc], %[acc], v247\n"
                    "s_cbranch_scc0 1b\n"
                    "2:\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    : [acc] "+v"(acc), [a] "+v"(a), [nb] "+s"(nb)
                    :
                    : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "scc", "memory");
            }
            SEQ_STAMP(46 + min(t0 / kFvRows, 12));
        }
        if (lane < kValCols) p.xb[col + lane] = acc;
        SEQ_STAMP(59);
        return;
    }
    auto vtile = [&](int t0, int buf, f4 (&vr)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + u * T, r = e / (kValCols / 4), c4 = e % (kValCols / 4);
            const bool in = t0 + r <= pos;                         // (rows behind pos: +0, whatever the clamped row holds -- the chain adds whole batches of 32 rows)
            const float a = s_p[min(t0 + r, pos)];
            float* d = &tile[buf][(4 * c4) * kFvStride + r];
            d[0] = in ? a * vr[u].x : 0.0f; d[kFvStride] = in ? a * vr[u].y : 0.0f;      // cpu.rs:48 `a * vi`, rounded
            d[2 * kFvStride] = in ? a * vr[u].z : 0.0f; d[3 * kFvStride] = in ? a * vr[u].w : 0.0f;
        }
        if (t0 + 4 * kFvRows <= pos) vissue(t0 + 4 * kFvRows, vr);      // (uniform)
        __syncthreads();                                          // this tile is written: the chain wave takes it
    };
    for (int t0 = 0; t0 <= pos; t0 += 4 * kFvRows) {
        vtile(t0, 0, v0);
        if (t0 + kFvRows <= pos) vtile(t0 + kFvRows, 1, v1);                    // uniform
        if (t0 + 2 * kFvRows <= pos) vtile(t0 + 2 * kFvRows, 0, v2);
        if (t0 + 3 * kFvRows <= pos) vtile(t0 + 3 * kFvRows, 1, v3);
    }
}

__global__ __launch_bounds__(kFvSoftWaves * 64) void attn_softmax_values_chain_kernel(RefAttnParams p) { attn_softmax_values_chain_body(p); }

}  // namespace rama
