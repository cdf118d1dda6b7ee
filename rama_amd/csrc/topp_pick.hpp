// The last step of the device top-p sampler spread over the chip (round 4): the running sums of the sorted probabilities
// (sample_top_q, infer.rs:70-73: cum_i = fl(cum_{i-1} + p_i) in fp32, one by one), the crossing of topp and the draw
// (infer.rs:75-84), bit for bit, by up to 32 workgroups in ONE launch.
//
// topp_pick_scan_kernel (topp_sort.hpp) forms the sums with one scan round per binade the sum passes through, on ONE compute
// unit: 29 us of the sampler's 51 when all 32 000 logits are candidates, bound by that unit's instruction issue.  Between two
// changes of the sum's exponent an add is integer arithmetic (topp_sort.hpp: an element is a map c -> c + D[c & 1] on the sum's
// 24-bit significand, maps compose) -- and WHICH binade the sum is in when element i is added can be told in advance from
// the mass in front of i, which the ranking launches deliver as an exact fixed-point sum (ToppSortParams::approx):
//
//   * the sequential fp32 sum s_{i-1} differs from the real-number sum by at most (i - 1) 2^-24 relative (i - 1 positive adds), the
//     fixed-point mass A_i by less than i 2^-47 absolute plus one rounding; with eps_i = (i + 128) 2^-23 (twice that) element i is
//     SAFE for binade E when  2^E <= A_i (1 - eps_i) - i 2^-47  and  (A_i (1 + eps_i) + i 2^-47 + p_i)(1 + 2^-20) < 2^(E+1):
//     then s_{i-1} is in binade E and s_i still is.  Everything else -- the first element, the elements around a power of two -- is SEQ;
//   * workgroup g takes elements [1024 g, 1024 g + 1024), one per thread: SAFE elements become integer maps for their binade's ulp,
//     neighbouring ones are composed by a segmented scan, and the chunk becomes a short list of ITEMS (a MAP segment: E, d0, d1;
//     a SEQ element: p), published through tagged words (no fence: every 8-byte word carries the launch's epoch);
//   * an item is two floats (f0, f1) and a step of the true sum is  cum += (significand of cum odd) ? f1 : f0  -- for a SEQ element
//     f0 = f1 = p, the fp32 add itself; for a MAP segment f = d U, and c U + d U = (c + d) U is exact (c + d < 2^24), the integer map
//     carried out by the adder.  Wave 0 of workgroup g walks the items of all chunks in front of its own like that (four dependent
//     instructions an item, no branch), then its own items, keeping the sum in front of each: every thread now has its element's
//     exact running sum (segment start + its composed map, or the walk's value for a SEQ element);
//   * the chunk that holds the crossing of topp (or the list's end) publishes (last, cum_last); the chunk that holds the first
//     i with u cum_last < cum_i finishes the step (token, cursor, next embedding: kernels.hpp finish_step).
//
// A flat 32 000-entry list is ~45 MAP items and ~90 SEQ elements in front of the crossing (61 of them around 1/2, where eps is
// 2e-3): the launch takes 12.5 us where the scan rounds took 29.  Lists up to kPickScanMax entries are left to the scan rounds.
// Every wait is bounded (kPickSpinLimit): a timeout raises ToppDistParams::bad and the launch drains.
#pragma once
#include "chain.hpp"

namespace rama {

constexpr int kPickChunk = 1024;           // elements per workgroup, one per thread
constexpr int kPickMaxChunks = 32;         // n <= 32768
constexpr long kPickSpinLimit = 1L << 22;
constexpr int kPickScanMax = 8192;         // lists up to this length are left to ONE workgroup's scan rounds (topp_sort.hpp): 5.4 us for 79 entries,
                                           // 8.5 for 7 000, where the phases below cost 7.4 and 9.5 (barriers of 16 waves, two hand-offs)

// 16 bytes as two tagged words: w0 = epoch (24 bits) << 40 | kind << 32 | bits of f0,  w1 = epoch (32 bits) << 32 | bits of f1.
// kind (statistics only) = the binade's biased exponent E (24..253) for a MAP segment (f = d0 U, d1 U); 0 for one SEQ element (f0 = f1 = p)
struct PickItem { unsigned long long w0, w1; };

struct ToppDistParams {
    const float* approx;                   // [m] the mass in front of element i (topp_rank_scatter_bs_kernel)
    PickItem* items;                       // [kPickMaxChunks][kPickChunk]
    unsigned long long* hdr;               // [kPickMaxChunks] epoch << 32 | items of the chunk
    unsigned long long* cross;             // [2] epoch << 32 | last,  epoch << 32 | bits of cum_last
    const unsigned* epoch;                 // advanced by the scatter launch: never seen before by this launch's readers
    unsigned* bad;                         // diagnostics: bit 0 a wait timed out, bit 1 a prediction did not hold (neither may ever happen)
};

__device__ __forceinline__ unsigned long long pick_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pick_store(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifdef RAMA_TOPP_STAMPS
__device__ unsigned long long g_pick_stamps[kPickMaxChunks][12];
#define PICK_STAMP(id) do { if (threadIdx.x == 0) g_pick_stamps[blockIdx.x][id] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PICK_STAMP(id) do { } while (0)
#endif

struct PickShared {
    unsigned long long it0[kPickChunk], it1[kPickChunk];      // a batch of other chunks' items (w0, w1)
    float own_a[kPickChunk], own_b[kPickChunk];
    float cum[kPickChunk + 1];             // the sum in front of own item k; [n_items]: behind the last
    int kind[kPickChunk];                  // per element: E, 0 (SEQ) or -1 (behind the list)
    int wf[16], wd0[16], wd1[16], wends[16];
    int cf[16], cd0[16], cd1[16], coff[16], n_items;       // per wave: the composed map of the waves in front, the items in front
    int off[kPickMaxChunks + 1];           // items in front of chunk c's
    float start;                           // the sum in front of the chunk
    int state;                             // 0 go on, 1 the chunk lies behind the crossing, 2 a wait timed out
    int last; float cum_last;
    int next;
};

// Up to 63 items, item l in lane l + 1 (f0, f1; lanes behind the list hold zeros and change nothing; lane 0 stands for the incoming
// sum), applied to the running sum as a lane ripple: every step all lanes form  c[l-1] + f0, c[l-1] + f1  and keep the one the parity
// of c[l-1] selects; lane l is final after step l (40 cycles a step; a loop that broadcasts item after item to the whole wave through
// v_readlane issues ten instructions an item: 72 cycles).  A window without a tie (f0 == f1 everywhere) is a plain ripple of adds, 16
// cycles a step.  Returns the sum behind the last item; `c` = the sum behind the item of every lane (lane 0: the incoming sum).
constexpr int kRippleItems = 63;
__device__ __forceinline__ float pick_ripple(float cum, float f0, float f1, int nl, float& c) {
    c = cum;
    if (__ballot(f0 != f1) == 0ull) {
        // no item of the window depends on the parity (no tie in any segment: the rule, not the exception): one add a step
        for (int k = 0; k < nl; k += 8) {                          // (steps beyond nl repeat finished lanes: same inputs, same values)
#pragma unroll
            for (int q = 0; q < 8; q++)
                asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(f0));      // lane 0 has no source lane: it keeps the incoming sum
        }
    } else {
        // lane 0 has no source lane: the DPP instructions leave its x0 / x1 / par as they are -- the incoming sum, selected for ever
        float x0 = cum, x1 = cum; unsigned par = 0u;
        const unsigned one = 1u;
        for (int k = 0; k < nl; k += 4) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                asm volatile("s_nop 1\n\t"
                             "v_and_b32_dpp %2, %3, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %0, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %1, %3, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             : "+&v"(x0), "+&v"(x1), "+&v"(par) : "v"(c), "v"(f0), "v"(f1), "v"(one));      // (early-clobber: %3 is read after %0 and %2 are written)
                c = par ? x1 : x0;
            }
        }
    }
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), nl));
}

__global__ __launch_bounds__(1024) void topp_pick_dist_kernel(ToppParams p, ToppDistParams d, ArgmaxParams fin) {
    __shared__ union { PickShared pick; ScanSharedT<kPickScanMax> scan; } shu;
    PickShared& sh = shu.pick;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x;
    int cpos = 0, n_forced = 0, n_out = 0, forced_tok = -1;
    const int i = g * kPickChunk + tid;
    // (the element is requested before m is known: the list and its length arrive together)
    const int m_raw = *p.m;
    if (m_raw <= kPickScanMax) {                                   // uniform over the launch
        if (g == 0) topp_pick_scan_body(p, fin, shu.scan);
        return;
    }
    const unsigned epoch = *d.epoch;
    const float pv_ = p.keys[min(i, p.n - 1)], av_ = d.approx[min(i, p.n - 1)];
    if (fin.ctl && tid == 0) {
        cpos = fin.ctl->pos; n_forced = fin.ctl->n_forced; n_out = fin.ctl->n_out;
        if (cpos < n_forced) forced_tok = fin.forced[cpos];
    }
    const int m = min(m_raw, kPickChunk * kPickMaxChunks);
    if (g * kPickChunk >= m && !(m == 0 && g == 0)) return;       // uniform: nothing of this chunk is kept
    if (m == 0) {                                                  // no candidate: -1 (kernels.hpp finish_step)
        if (tid == 0) sh.next = finish_step(fin, -1, cpos, n_forced, n_out, forced_tok);
        gather_next_embedding(fin, &sh.next);
        return;
    }
    const bool real = i < m;
    const float pv = real ? pv_ : 0.0f, av = real ? av_ : 0.0f;
    PICK_STAMP(0);
    // ---- 1. SAFE for one binade, or SEQ
    const float eps = (float)(i + 128) * 0x1p-23f, absm = (float)i * 0x1p-47f;
    const float lo = av * (1.0f - eps) - absm, hi = (av * (1.0f + eps) + absm + pv) * (1.0f + 0x1p-20f);
    const int el = (int)(__float_as_uint(lo) >> 23), eh = (int)(__float_as_uint(hi) >> 23);       // (a sign bit makes el > 253)
    const bool safe = real && lo > 0.0f && el == eh && el >= 24 && el <= 253;
    const int E = !real ? -1 : (safe ? el : 0);
    sh.kind[tid] = E;
    if (tid == 0) { sh.state = 0; sh.last = -1; }
    __syncthreads();
    const int Eprev = tid > 0 ? sh.kind[tid - 1] : 0, Enext = tid < kPickChunk - 1 ? sh.kind[tid + 1] : 0;
    // a segment starts at the chunk's first element, at a SEQ element, behind one, and (never, by the bound above) where E changes
    const bool F = real && (tid == 0 || E == 0 || Eprev != E);
    const bool end = real && (tid == kPickChunk - 1 || Enext != E || E == 0);
    SegInc sv{F ? 1 : 0, Inc{0, 0}};
    if (safe) {
        const int e = elem_of(pv, __uint_as_float((unsigned)(277 - E) << 23));
        const int d0 = e & 0x0FFFFFFF;
        sv.m = Inc{d0, d0 + (e >> 30)};
    }
    // ---- 2. segmented inclusive scan: over the wave, then over the waves
    sv = seg_then(seg_dpp<0x111, 0xF>(sv), sv);
    sv = seg_then(seg_dpp<0x112, 0xF>(sv), sv);
    sv = seg_then(seg_dpp<0x114, 0xF>(sv), sv);
    sv = seg_then(seg_dpp<0x118, 0xF>(sv), sv);
    sv = seg_then(seg_dpp<0x142, 0xA>(sv), sv);
    sv = seg_then(seg_dpp<0x143, 0xC>(sv), sv);
    const unsigned long long endmask = __ballot(end);
    if (lane == 63) { sh.wf[wave] = sv.f; sh.wd0[wave] = sv.m.d0; sh.wd1[wave] = sv.m.d1; }
    if (lane == 0) sh.wends[wave] = __popcll(endmask);
    __syncthreads();
    // ... the 16 waves' totals by ONE wave (a row of 16 lanes: four DPP steps) instead of a 16-step loop in every thread: that loop was
    // ~320 of this phase's ~500 instructions per thread, on four waves per SIMD
    if (wave == 0) {
        SegInc t{lane < 16 ? sh.wf[lane & 15] : 0, Inc{lane < 16 ? sh.wd0[lane & 15] : 0, lane < 16 ? sh.wd1[lane & 15] : 0}};
        const int we = lane < 16 ? sh.wends[lane & 15] : 0;
        int ie = we;
        t = seg_then(seg_dpp<0x111, 0xF>(t), t); ie += __builtin_amdgcn_update_dpp(0, ie, 0x111, 0xF, 0xF, true);
        t = seg_then(seg_dpp<0x112, 0xF>(t), t); ie += __builtin_amdgcn_update_dpp(0, ie, 0x112, 0xF, 0xF, true);
        t = seg_then(seg_dpp<0x114, 0xF>(t), t); ie += __builtin_amdgcn_update_dpp(0, ie, 0x114, 0xF, 0xF, true);
        t = seg_then(seg_dpp<0x118, 0xF>(t), t); ie += __builtin_amdgcn_update_dpp(0, ie, 0x118, 0xF, 0xF, true);
        const SegInc ex = seg_dpp<0x111, 0xF>(t);                  // exclusive: the waves in front (lane 0: identity)
        if (lane < 16) { sh.cf[lane] = ex.f; sh.cd0[lane] = ex.m.d0; sh.cd1[lane] = ex.m.d1; sh.coff[lane] = ie - we; }
        if (lane == 15) sh.n_items = ie;
    }
    __syncthreads();
    const int my_item = __popcll(endmask & ((1ull << lane) - 1ull)) + sh.coff[wave], n_items = sh.n_items;
    if (!sv.f) sv.m = inc_then(Inc{sh.cd0[wave], sh.cd1[wave]}, sv.m);
    // ---- 3. the chunk's items, to the other workgroups and to this one's walk
    if (end) {
        const float U = __uint_as_float((unsigned)(max(E, 24) - 23) << 23);
        const float a = safe ? (float)sv.m.d0 * U : pv, b = safe ? (float)sv.m.d1 * U : pv;
        PickItem* dst = d.items + (size_t)g * kPickChunk + my_item;
        pick_store(&dst->w0, ((unsigned long long)(epoch & 0xFFFFFFu) << 40) | ((unsigned long long)(unsigned)E << 32) | __float_as_uint(a));
        pick_store(&dst->w1, ((unsigned long long)epoch << 32) | __float_as_uint(b));
        sh.own_a[my_item] = a; sh.own_b[my_item] = b;
    }
    if (tid == 0) pick_store(d.hdr + g, ((unsigned long long)epoch << 32) | (unsigned)n_items);
    PICK_STAMP(1);
    // ---- 4. the walk.  First the other chunks' items, 1024 at a time: how many there are ...
    if (wave == 0) {
        int cnt = 0;
        if (lane < g) {
            long spins = 0;
            unsigned long long h;
            while ((unsigned)((h = pick_load(d.hdr + lane)) >> 32) != epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kPickSpinLimit || ((spins & 1023) == 0 && __hip_atomic_load(d.bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u)) { sh.state = 2; break; }
            }
            cnt = (int)(unsigned)h;
        }
        int incl = cnt;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { const int o = __shfl_up(incl, s); incl += lane >= s ? o : 0; }
        if (lane <= kPickMaxChunks) sh.off[lane] = incl - cnt;     // off[c] for c <= g (lanes >= g hold the total)
    }
    __syncthreads();
    PICK_STAMP(2);
    int bad = 0;
    const int T = sh.state == 0 ? sh.off[g] : 0;                   // uniform
    float cum = 0.0f;                                              // wave 0: the true running sum (uniform)
    for (int base = 0; base < T && sh.state == 0; base += kPickChunk) {
        const int k = base + tid;
        if (k < T) {
            int c = 0;
#pragma unroll
            for (int s = 16; s >= 1; s >>= 1) c += (c + s < g && sh.off[c + s] <= k) ? s : 0;       // the last chunk whose items start at or before k
            const PickItem* src = d.items + (size_t)c * kPickChunk + (k - sh.off[c]);
            long spins = 0;
            unsigned long long w0, w1;
            while (true) {
                w0 = pick_load(&src->w0); w1 = pick_load(&src->w1);
                if ((unsigned)(w0 >> 40) == (epoch & 0xFFFFFFu) && (unsigned)(w1 >> 32) == epoch) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kPickSpinLimit) { sh.state = 2; break; }
            }
            sh.it0[tid] = w0; sh.it1[tid] = w1;
        }
        __syncthreads();
        PICK_STAMP(7);
        if (wave == 0 && sh.state == 0) {
            const int nb = __builtin_amdgcn_readfirstlane(min(T - base, kPickChunk));
            bool over = false;
            PICK_STAMP(8);
            for (int j0 = 0; j0 < nb && !over; j0 += kRippleItems) {
                const int j = j0 + lane - 1;                       // lane l + 1 holds item j0 + l
                const bool have = lane > 0 && j < nb;
                const float f0 = have ? __uint_as_float((unsigned)sh.it0[j]) : 0.0f, f1 = have ? __uint_as_float((unsigned)sh.it1[j]) : 0.0f;
                float c;
                cum = pick_ripple(cum, f0, f1, min(kRippleItems, nb - j0), c);
                over = cum > p.topp;                               // the crossing lies in front of this chunk: nothing left to do here
            }
            PICK_STAMP(9);
            if (over && lane == 0) sh.state = 1;
        }
        __syncthreads();
    }
    if (sh.state != 0) {                                           // uniform
        if (sh.state == 2 && tid == 0) atomicOr(d.bad, 1u);
        return;
    }
    PICK_STAMP(3);
    // ... then the own items, keeping the sum in front of each
    if (wave == 0) {
        if (lane == 0) sh.start = cum;
        const int ni = __builtin_amdgcn_readfirstlane(n_items);
        for (int j0 = 0; j0 < ni; j0 += kRippleItems) {
            const int j = j0 + lane - 1;
            const bool have = lane > 0 && j < ni;
            const float f0 = have ? sh.own_a[j] : 0.0f, f1 = have ? sh.own_b[j] : 0.0f;
            float c;
            cum = pick_ripple(cum, f0, f1, min(kRippleItems, ni - j0), c);
            const float before = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x138, 0xF, 0xF, false));     // wave_shr:1
            if (have) sh.cum[j] = before;                          // the sum in front of item j = behind the item of the lane below
        }
        if (lane == 0) sh.cum[ni] = cum;
    }
    __syncthreads();
    PICK_STAMP(4);
    // ---- 5. every element's running sum; the crossing of topp
    float ci = 0.0f;
    bool beyond = false;
    if (real) {
        const float seg0 = sh.cum[my_item];
        beyond = seg0 > p.topp;
        if (safe) {
            const unsigned cb = __float_as_uint(seg0);
            const int c0 = (int)((cb & 0x7FFFFFu) | 0x800000u);
            const int c = c0 + ((c0 & 1) ? sv.m.d1 : sv.m.d0);
            if (!beyond) bad |= ((int)(cb >> 23) != E || c > (1 << 24) - 1) ? 1 : 0;
            ci = (float)c * __uint_as_float((unsigned)(E - 23) << 23);
        } else ci = sh.cum[my_item + 1];
    }
    const bool over = real && (beyond || ci > p.topp);
    const int first_over = block_min_i(over ? i : kNoEvent);
    const bool have_last = first_over != kNoEvent || (m - 1) / kPickChunk == g;      // uniform
    const int last_here = first_over != kNoEvent ? first_over : m - 1;
    if (have_last && i == last_here) {
        sh.last = last_here; sh.cum_last = ci;
        pick_store(d.cross, ((unsigned long long)epoch << 32) | (unsigned)last_here);
        pick_store(d.cross + 1, ((unsigned long long)epoch << 32) | __float_as_uint(ci));
    }
    if (!have_last && tid == 0) {
        long spins = 0;
        unsigned long long w0, w1;
        while (true) {
            w0 = pick_load(d.cross); w1 = pick_load(d.cross + 1);
            if ((unsigned)(w0 >> 32) == epoch && (unsigned)(w1 >> 32) == epoch) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kPickSpinLimit || ((spins & 1023) == 0 && __hip_atomic_load(d.bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u)) { sh.state = 2; break; }
        }
        sh.last = (int)(unsigned)w0; sh.cum_last = __uint_as_float((unsigned)w1);
    }
    __syncthreads();
    PICK_STAMP(5);
    if (bad) atomicOr(d.bad, 2u);
    if (sh.state != 0) {                                           // uniform
        if (tid == 0) atomicOr(d.bad, 1u);
        return;
    }
    // ---- 6. r = u * cum; the first i < last whose running sum exceeds r wins, else `last` (infer.rs:75-84): the sums never
    // decrease, so exactly one chunk holds the first i <= last with r < cum_i or i == last and has no such sum in front of it
    const int last = sh.last;
    const float r = p.u * sh.cum_last;
    if (p.prefix && real && i <= last) p.prefix[i] = ci;           // kept for inspection (tests): the sums up to the crossing
    const bool cand = real && i <= last && (r < ci || i == last);
    const int best = block_min_i(cand ? i : kNoEvent);
    if (best == kNoEvent || (g > 0 && r < sh.start)) return;      // uniform
    if (tid == 0) sh.next = finish_step(fin, p.vals[best], cpos, n_forced, n_out, forced_tok);
    gather_next_embedding(fin, &sh.next);
    PICK_STAMP(6);
}

}  // namespace rama
