// seqsum_fast.hpp -- the sequential fp32 sum s = ((a_0 + a_1) + a_2) + ... of non-negative terms (cpu.rs:110-113: rmsnorm's
// Iterator::sum over x * x), bit for bit, as ONE pass over registers and a short walk.  [r5]
//
// chain.hpp's seq_sum_predict composes integer increment maps (ties included) with segmented scans and walks <= 12 sixteen-
// element runs: 4.7 us of dependent work on four waves for 4096 terms, 22 us on one.  This version keeps the maps out of it:
//   * thread t holds R consecutive terms in registers, in groups of 8.  An approximate prefix sum (any order) predicts the
//     binade E the true sum is in across a group; while the sum stays in one binade, fl(s + a) = s + RN_U(a) with
//     U = ulp(2^E) -- and RN_U(a) = (2^E + a) - 2^E, two adds, no scaling -- EXCEPT when a lies exactly between two multiples
//     of U (a tie: the result then depends on the parity of s / U).
//   * A group is a MAP group when the predicted sums at its two ends lie in one binade with a margin of 2^-12 on either
//     side AND none of its terms is a tie; its contribution is the plain fp32 sum of its eight RN_U(a) (multiples of U below
//     2^(E+1): exact in any order).  Any other group (a binade crossing -- ~1.3 groups per crossing --, a tie, the first
//     terms where the sum changes binade every few terms, zeros / infinities / subnormal sums) is a SEQ group.
//   * The threads write a list of ITEMS in index order: one float per run of MAP groups of one binade (within a thread),
//     eight floats -- the terms themselves -- per SEQ group.  One wave then WALKS the list with the true sum: s += item, one
//     dependent add each (~8 cycles), ~200 items for 4096 squares, and leaves the sum behind every item in LDS.
//   * Afterwards every thread checks its own MAP items: the true sum in front of the run and behind it must lie in the
//     predicted binade (the sum only grows, so every add in between was in that binade too, and with no tie among them
//     s + sum(RN_U(a)) is what the sequential adds give).  A failed check -- the prediction was off by more than the margin --
//     makes the function return false and the caller takes the plain loop; the result is the sequential sum whenever it
//     is returned as good.
// ~900 vector instructions for 4096 terms on one wave (R = 64) + the walk: ~2 us; ~1.4 us on two waves (R = 32).
#pragma once
#include "ref_order.hpp"
#include <type_traits>

namespace rama {

constexpr int kFsCap = 1024;          // items the walk list holds (a list that would be longer: return false)
template <int NW>
struct FastSumShared {
    __attribute__((aligned(16))) float items[kFsCap + 64];
    float after[8];                                                 // [4]: the total, for the other waves
    __attribute__((aligned(16))) int tags[kFsCap + 64];             // item i: the binade (biased exponent) a MAP item was formed for, 0 for a term
    float wsum[NW];
    int wcnt[NW];
    int wbad[NW];
    int n_items;      // (diagnostics)
};

template <int CTRL, int RM> __device__ __forceinline__ float fs_fadd(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RM, 0xF, true));
}
template <int CTRL, int RM> __device__ __forceinline__ int fs_iadd(int v) { return v + __builtin_amdgcn_update_dpp(0, v, CTRL, RM, 0xF, true); }

#ifdef RAMA_FS_STAMPS
__device__ unsigned long long g_fs_stamps[16];
#define FS_STAMP(id) do { if (threadIdx.x == 0) { g_fs_stamps[id] = __builtin_amdgcn_s_memrealtime(); g_fs_stamps[8 + (id)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define FS_STAMP(id) do { } while (0)
#endif

// to be called before the terms are there (it overlaps their round trip): the tags of a fresh list
template <int NW>
__device__ __forceinline__ void seq_sum_fast_prepare(FastSumShared<NW>& fs) {
    typedef __attribute__((ext_vector_type(4))) int i4;
    for (int i = threadIdx.x; i < (kFsCap + 64) / 4; i += NW * 64) reinterpret_cast<i4*>(fs.tags)[i] = i4{0, 0, 0, 0};
}

// a[k] = term tid * R + k of the list (terms behind its end: 0.0f -- they change nothing); all NW waves of the workgroup call it, after
// seq_sum_fast_prepare.  *out = the sequential sum when true is returned (the same value and verdict in every thread).
template <int NW, int R>
__device__ __forceinline__ bool seq_sum_fast(const float (&a)[R], FastSumShared<NW>& fs, float* out) {
    RAMA_NO_CONTRACT
    static_assert(R % 8 == 0 && R >= 8 && R <= 64, "whole groups of 8 terms, a 32-bit group mask");
    constexpr int G = R / 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
    FS_STAMP(0);
    // the estimate: group sums as trees, an ordinary scan over the threads
    float gsum[G];
#pragma unroll
    for (int g = 0; g < G; g++)
        gsum[g] = ((a[8 * g] + a[8 * g + 1]) + (a[8 * g + 2] + a[8 * g + 3])) + ((a[8 * g + 4] + a[8 * g + 5]) + (a[8 * g + 6] + a[8 * g + 7]));
    float loc = 0.0f;
#pragma unroll
    for (int g = 0; g < G; g++) loc = loc + gsum[g];
    float inc = loc;
    inc = fs_fadd<0x111, 0xF>(inc); inc = fs_fadd<0x112, 0xF>(inc); inc = fs_fadd<0x114, 0xF>(inc); inc = fs_fadd<0x118, 0xF>(inc);
    inc = fs_fadd<0x142, 0xA>(inc); inc = fs_fadd<0x143, 0xC>(inc);
    float base = 0.0f;
    if (NW > 1) {
        if (lane == 63) fs.wsum[wave] = inc;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW - 1; w++) base += w < wave ? fs.wsum[w] : 0.0f;
    }
    float P = base + (inc - loc);                                 // ~ the sum in front of this thread's terms
    FS_STAMP(1);
    // the groups: SEQ or MAP, the MAP groups' sums, the runs of MAP groups of one binade; their places in the list count up on the way
    unsigned m = 0, cl = 0;                                       // bit g: group g is SEQ / a run of MAP groups ends with group g
    float run[G];                                                 // run[g]: the sum of the run up to and including group g
    int eg[G];
    int cnt = 0;
    int offc[G];                                                  // items of this thread in front of group g's own (its terms, or the run item it closes)
    {
        int eprev = -1;
        bool open = false;
        float acc = 0.0f;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const unsigned lob = __float_as_uint(P);
            const float Phi = P + gsum[g];
            const unsigned hib = __float_as_uint(Phi);
            const unsigned Mb = lob & 0xFF800000u;
            const float M = __uint_as_float(Mb);                  // 2^E
            const float hU = __uint_as_float(Mb - (24u << 23));   // ulp(2^E) / 2 (E >= 32 below)
            const int E = (int)(lob >> 23);
            float rmax = 0.0f, seg = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float a0 = a[8 * g + e], a1 = a[8 * g + e + 1];
                const float d0 = (M + a0) - M, d1 = (M + a1) - M;  // the term rounded to a multiple of ulp(2^E) (meaningless when it is >= 2^E: the group is then not safe)
                const float r0 = a0 - d0, r1 = a1 - d1;            // exact, |r| <= ulp / 2
                rmax = __builtin_fmaxf(__builtin_fmaxf(rmax, __builtin_fabsf(r0)), __builtin_fabsf(r1));        // v_max3_f32
                seg = (seg + d0) + d1;
            }
            const bool tie = rmax == hU;                           // some term lies exactly between two multiples
            // both ends in binade E with 2^-12 to spare: lob - 2048 keeps the exponent iff the sum is >= 2^E (1 + 2^-12), hib + 2048 iff
            // it is < 2^(E+1) (1 - 2^-13).  (A zero estimate wraps to an exponent of 511: not safe; nan / inf: E = 255.)
            const bool safe = (((lob - 2048u) ^ (hib + 2048u)) >> 23) == 0u && E >= 32 && E <= 253;
            const bool sp = !safe || tie;
            const bool cont = open && !sp && E == eprev;           // this group continues the run in front of it
            const bool closes_prev = open && !cont;                // ... or ends it (a SEQ group, or a MAP group of another binade)
            if (g > 0) { cl |= closes_prev ? (1u << (g - 1)) : 0u; cnt += closes_prev ? 1 : 0; }
            offc[g] = cnt;
            cnt += sp ? 8 : 0;
            acc = cont ? acc + seg : seg;
            run[g] = acc; eg[g] = E;
            m |= sp ? (1u << g) : 0u;
            open = !sp; eprev = E; P = Phi;
        }
        if (open) { cl |= 1u << (G - 1); }
        // (the place of a run item: behind everything counted up to the group AFTER its last one -- offc[g + 1] - 1, or cnt for the last run)
        cnt += open ? 1 : 0;
    }
    FS_STAMP(2);
    // Threads that are ONE run (no SEQ group, one binade) and follow one another with the same binade are one run: the last of them
    // writes the item, with the sum of all of them (multiples of one ulp below 2^(E+1): exact in any order).
    const bool pure = m == 0u && cnt == 1 && (cl & ~(1u << (G - 1))) == 0u;
    const int epure = pure ? eg[G - 1] : -1 - lane;               // (no two impure lanes compare equal)
    float chain = pure ? run[G - 1] : 0.0f;
    bool last_of_chain = false;
    {
        const int eleft = __builtin_amdgcn_update_dpp(-1000, epure, 0x138, 0xF, 0xF, false);       // wave_shr:1 (lane 0 keeps -1000)
        const int eright = __builtin_amdgcn_update_dpp(-1000, epure, 0x130, 0xF, 0xF, false);      // wave_shl:1 (lane 63 keeps -1000)
        int head = (pure && eleft == epure) ? 0 : 1;              // 1: a segment starts here (every impure lane is its own segment)
        last_of_chain = pure && eright != epure;
        // segmented inclusive scan of `chain` over the lanes (Hillis-Steele inside the rows, then the row totals)
        auto step = [&](auto ctrl, auto rmask) {
            constexpr int CTRL = decltype(ctrl)::value, RM = decltype(rmask)::value;
            const float cv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, chain), CTRL, RM, 0xF, true));
            const int hv = __builtin_amdgcn_update_dpp(0, head, CTRL, RM, 0xF, true);              // (lanes without a source: nothing to add, nothing changes)
            chain = head ? chain : chain + cv;
            head = head | hv;
        };
        step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xF>{});
        step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xF>{});
        step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xF>{});
        step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xF>{});
        step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});
        step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});
    }
    const int cnt2 = pure ? (last_of_chain ? 1 : 0) : cnt;
    // places in the list
    int ic = cnt2;
    ic = fs_iadd<0x111, 0xF>(ic); ic = fs_iadd<0x112, 0xF>(ic); ic = fs_iadd<0x114, 0xF>(ic); ic = fs_iadd<0x118, 0xF>(ic);
    ic = fs_iadd<0x142, 0xA>(ic); ic = fs_iadd<0x143, 0xC>(ic);
    int off0 = ic - cnt2, nitems;
    if (NW > 1) {
        if (lane == 63) fs.wcnt[wave] = ic;
        __syncthreads();
        nitems = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) { const int c = fs.wcnt[w]; off0 += w < wave ? c : 0; nitems += c; }
    } else {
        nitems = __builtin_amdgcn_readlane(ic, 63);
    }
    if (nitems > kFsCap) return false;                            // uniform
    if (tid < 64) fs.items[nitems + tid] = 0.0f;                  // the walk takes 64 items at a time
    if (tid == 0) fs.n_items = nitems;
    if (pure) {
        if (last_of_chain) { fs.items[off0] = chain; fs.tags[off0] = epure; }
    } else {
#pragma unroll
        for (int g = 0; g < G; g++) {
            if ((cl >> g) & 1u) {                                 // the run item: behind the items counted up to the next group, minus itself
                const int o = off0 + (g + 1 < G ? offc[g + 1] - 1 : cnt - 1);
                fs.items[o] = run[g];
                fs.tags[o] = eg[g];
            }
            if ((m >> g) & 1u) {
                float* d = fs.items + off0 + offc[g];
#pragma unroll
                for (int e = 0; e < 8; e++) d[e] = a[8 * g + e];
            }
        }
    }
    __syncthreads();
    FS_STAMP(3);
    // The walk, by wave 0: the true sum, one dependent add per item -- v_readlane (the item into a scalar register) + v_add_f32, 8.9 cycles
    // a step, nothing else in the chain.  Lane l holds item 64 k + l.
    // The check needs no exact partial sums: a MAP item formed for binade E is good iff the true sum in front of it and behind it lies
    // in [2^E, 2^(E+1)), and the true (sequential) partial sum s_i differs from ANY-order fp32 sum A_i of the same <= 1024 non-negative
    // items by less than (i + 16) 2^-24 relative (both are within gamma_k of the exact sum, k the number of adds on a term's way).  So an
    // ordinary scan gives A, and A_(i-1) (1 - 2^-13) >= 2^E together with A_i (1 + 2^-13) < 2^(E+1) PROVES the item good; anything else
    // counts as bad (the caller then takes the plain loop) -- the prediction had a margin of 2^-12, so that is rare.
    bool bad = false;
    float total = 0.0f;
    if (wave == 0) {
        float s = 0.0f, carryA = 0.0f;
        for (int i0 = 0; i0 < nitems; i0 += 64) {
            const float item = fs.items[i0 + lane];
            const int tag = fs.tags[i0 + lane];
            const int left = nitems - i0;                          // uniform
#define RAMA_FS_WALK_8(B) \
            if (left > B) asm volatile( \
                "v_readlane_b32 s22, %1, " #B "\n\t" \
                "v_readlane_b32 s23, %1, " #B " + 1\n\t" \
                "v_add_f32 %0, s22, %0\n\t" \
                "v_readlane_b32 s22, %1, " #B " + 2\n\t" \
                "v_add_f32 %0, s23, %0\n\t" \
                "v_readlane_b32 s23, %1, " #B " + 3\n\t" \
                "v_add_f32 %0, s22, %0\n\t" \
                "v_readlane_b32 s22, %1, " #B " + 4\n\t" \
                "v_add_f32 %0, s23, %0\n\t" \
                "v_readlane_b32 s23, %1, " #B " + 5\n\t" \
                "v_add_f32 %0, s22, %0\n\t" \
                "v_readlane_b32 s22, %1, " #B " + 6\n\t" \
                "v_add_f32 %0, s23, %0\n\t" \
                "v_readlane_b32 s23, %1, " #B " + 7\n\t" \
                "v_add_f32 %0, s22, %0\n\t" \
                "s_nop 0\n\t" \
                "v_add_f32 %0, s23, %0\n\t" \
                : "+v"(s) : "v"(item) : "s22", "s23")
            RAMA_FS_WALK_8(0); RAMA_FS_WALK_8(8); RAMA_FS_WALK_8(16); RAMA_FS_WALK_8(24);
            RAMA_FS_WALK_8(32); RAMA_FS_WALK_8(40); RAMA_FS_WALK_8(48); RAMA_FS_WALK_8(56);
#undef RAMA_FS_WALK_8
            // (items behind the list are zeros and carry no tag)
            float A = item;
            A = fs_fadd<0x111, 0xF>(A); A = fs_fadd<0x112, 0xF>(A); A = fs_fadd<0x114, 0xF>(A); A = fs_fadd<0x118, 0xF>(A);
            A = fs_fadd<0x142, 0xA>(A); A = fs_fadd<0x143, 0xC>(A);
            A = A + carryA;
            const float Ab = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, carryA), __builtin_bit_cast(int, A), 0x138, 0xF, 0xF, false));     // wave_shr:1; lane 0: the carry
            const int elo = (int)(__float_as_uint(Ab * (1.0f - 0x1p-13f)) >> 23), ehi = (int)(__float_as_uint(A * (1.0f + 0x1p-13f)) >> 23);
            bad = bad || (tag != 0 && (elo != tag || ehi != tag));
            carryA = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, A), 63));
        }
        total = s;
        if (NW > 1 && lane == 0) fs.after[4] = total;
    }
    FS_STAMP(4);
    const bool wbad = __builtin_amdgcn_ballot_w64(bad) != 0;
    bool anybad = wbad;
    if (NW > 1) {
        if (lane == 0) fs.wbad[wave] = wbad ? 1 : 0;
        __syncthreads();
        anybad = false;
#pragma unroll
        for (int w = 0; w < NW; w++) anybad = anybad || fs.wbad[w] != 0;
    }
    *out = NW > 1 ? fs.after[4] : total;
    FS_STAMP(5);
    return !anybad;
}

// test entry: the sum of a[0..n) with NW waves (n <= 64 NW R); out[0] = sum, out[1] = 1.0 when the fast path held (else the
// plain loop's sum is in out[0]), out[2] = items walked, out[3] = 100 MHz ticks from the first instruction to the last
template <int NW, int R>
__global__ __launch_bounds__(NW * 64) void seqsum_fast_test_kernel(const float* a, int n, float* out) {
    RAMA_NO_CONTRACT
    __shared__ FastSumShared<NW> fs;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float v[R];
#pragma unroll
    for (int k = 0; k < R; k++) { const int i = (int)threadIdx.x * R + k; v[k] = a[min(i, n - 1)]; if (i >= n) v[k] = 0.0f; }
    seq_sum_fast_prepare<NW>(fs);
    float s;
    const bool ok = seq_sum_fast<NW, R>(v, fs, &s);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        if (!ok) { s = 0.0f; for (int i = 0; i < n; i++) s = s + a[i]; }
        out[0] = s; out[1] = ok ? 1.0f : 0.0f; out[2] = (float)fs.n_items; out[3] = (float)(t1 - t0);
    }
}

}  // namespace rama
