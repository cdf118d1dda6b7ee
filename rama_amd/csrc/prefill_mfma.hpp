// prefill_mfma.hpp -- token-batch passes as a true dense fp32 GEMM on the matrix cores
// (BASELINE north_star "MFMA only if a batched-prompt prefill path is
// added as a true dense GEMM").  P = 16*PT tokens (prompt positions of one sequence, or one token of
// each of several sequences) go through a layer together:
//
//     O[p][r] = sum_k W[r][k] * X[p][k]          W [rows, K] row-major fp32 (the checkpoint layout)
//
// computed with v_mfma_f32_16x16x4_f32 (exact fp32: bit-for-bit an fmaf chain, no reduced
// precision), the weight tile as the A operand and the activations as B, with the decode kernels'
// fused epilogues per token (RoPE + KV-cache append, SiLU*gate, residual add).  The contract is the
// reference's: the same KV-cache rows and final logits as P sequential forward() calls
// (transformer/mod.rs:187-194 feeds the forced prompt tokens one forward() at a time).
//
// What bounds it (measured, tools/pf_mfma_bench.hip): a CU sustains ~24 GB/s from HBM and ~70 GB/s
// from its XCD's L2, and the two streams share one budget of bytes in flight -- time per CU =
// W_bytes / 24 + X_bytes / 70.  Every 16-row weight tile needs the whole activation block, so
// X_bytes = W_bytes * P / (16 RT): the activation re-reads, not the matrix cores (P = 64 needs 55 %
// of the 155 TFLOP/s fp32 MFMA peak at the HBM rate), decide how far above the weight-stream time a
// pass lands, and a larger P amortises both.  Hence:
//   * activations live in TILE layout between the kernels of a pass: a 16-token x 16-float block
//     is stored as the 64 lanes x 16 bytes of the MFMA B operand (lane l = token l&15, k-slot l>>4,
//     its 4 floats = k-slot's floats 4 kq .. 4 kq + 3).  A wave reads a block with ONE fully
//     coalesced 1-KiB buffer_load_dwordx4 and feeds register e to MFMA e -- the sum over k is
//     order-free, so MFMA e's four k-slots are floats 4 kq + e.  The D tile of a 16x16x4 MFMA (lane
//     = token, 4 registers = 4 consecutive rows) IS that layout for the next GEMM's K = this GEMM's
//     rows, so every epilogue store is one coalesced 16-byte store per lane, no transpose anywhere.
//   * weights stay in the reference's row-major layout: a wave reads a 16-row x 64-float block with 4
//     buffer_load_dwordx4 nt -- lane l takes 16 bytes at row l>>2, float 16 j + 4 (l&3), so every
//     quad of lanes covers 64 contiguous bytes and the 4 loads 256 contiguous bytes per row -- and
//     one ds_bpermute_b32 per register rotates the lane index by two bits into the MFMA A layout
//     (lane = row l&15, k-slot l>>4), matching the B operand's k assignment.
//   * rmsnorm is a small kernel of its own on the tile layout (exactly cpu.rs:99-117's
//     w * (v * x)), not folded into the GEMM: the fold would re-read the gain vector per row tile
//     through the same L2 budget.
// K is split over the workgroup's 8 waves in 64-float chunks (chunk c -> wave c mod 8: the
// workgroup sweeps each row front to back like the decode kernels); partial tiles meet in LDS.
#pragma once
#include "kernels.hpp"
#include <type_traits>

namespace rama {

constexpr int kMfWaves = 8;
constexpr int kMfThreads = kMfWaves * 64;
constexpr int kMfMaxTok = 128;               // tokens per pass (PT <= 8; PT = 5 .. 8 need the tile-order weight copy)
constexpr int kMfMaxTokRows = 64;            // ... with row-major weights (PT <= 4)

enum { EPI_SWIGLU = 3, EPI_STORE_ROWS = 4 };

// ---- tile layout of an activation matrix [n_tok, K], K % 16 == 0
// float index of element (token tk, float k): block (tk / 16, k / 16) of 256 floats, inside it
// lane (tk % 16) + 16 * ((k / 4) % 4), register k % 4
__host__ __device__ __forceinline__ size_t tile_idx(int tk, int k, int K) {
    return ((size_t)((tk >> 4) * (K >> 4) + (k >> 4)) * 64 + (size_t)(((k >> 2) & 3) * 16 + (tk & 15))) * 4 + (size_t)(k & 3);
}
__host__ __device__ __forceinline__ size_t tile_floats(int n_tok, int K) { return (size_t)((n_tok + 15) >> 4) * 16 * (size_t)K; }

struct MfParams {
    const float* w[3];     // matrices [rows, K] row-major (EPI_QKV: wq, wk, wv; EPI_SWIGLU: w1, w3), or their tile-order copies
    int tiled;             // 1: w[] are tile-order copies (host picks the LD = 3 instantiation)
    const float* ssp;      // rmsnorm folded in: x holds X * gain and every output is scaled per token by
                           // 1 / sqrt(mean(X^2) + eps) from these partial sums (rms_fold_kernel); NULL: x is used as is
    const float* x;        // activations, tile layout [ceil(n_tok / 16) * 16, K]
    float* o;              // output, tile layout [.., rows] (EPI_QKV: q; EPI_STORE_ROWS: row-major [n_tok, o_stride])
    size_t slab_floats;    // split-K: partial sums of K-slice ks go to o + ks * slab_floats
    int o_stride;
    int K, rows, n_tok;
    int ksplit;            // K-slices (workgroups per row group), >= 1
    int nunit;             // (row group, K-slice) units per workgroup, >= 1
    int pos0;              // position of token 0 (EPI_QKV)
    const float* fr; const float* fi; int head_size;
    float* kc; float* vc;  // this layer's cache slabs [seq, dim]
    const SeqSlot* seqs; size_t layer_off;    // batched independent sequences (rama_decode_batch)
};

typedef __attribute__((ext_vector_type(4))) float acc4;

// in-kernel time stamps of workgroup RAMA_MF_STAMP_BLOCK (tools/pf_mfma_bench.hip -DRAMA_MF_STAMPS): 100 MHz ticks per wave
#ifdef RAMA_MF_STAMPS
#ifndef RAMA_MF_STAMP_BLOCK
#define RAMA_MF_STAMP_BLOCK 0
#endif
__device__ unsigned long long g_mf_stamps[kMfWaves][8];
__device__ unsigned long long g_mf_blocks[1024][4];      // per workgroup: start, end, HW_ID, XCC_ID
#define MF_BLOCK_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) { g_mf_blocks[blockIdx.x][k] = __builtin_amdgcn_s_memrealtime(); \
    if (k == 0) { g_mf_blocks[blockIdx.x][2] = __builtin_amdgcn_s_getreg(63492); g_mf_blocks[blockIdx.x][3] = __builtin_amdgcn_s_getreg(63508); } } } while (0)
#define MF_STAMP(id) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == RAMA_MF_STAMP_BLOCK) { g_mf_stamps[threadIdx.x >> 6][id] = __builtin_amdgcn_s_memrealtime(); \
    if (id < 2) g_mf_stamps[threadIdx.x >> 6][5 + id] = __builtin_amdgcn_s_memtime(); } } while (0)      /* [5], [6]: core-clock cycles at stamps 0 and 1 */
#else
#define MF_STAMP(id) do { } while (0)
#define MF_BLOCK_STAMP(k) do { } while (0)
#endif

// PT  token tiles of 16 (P = 16 PT tokens per pass)
// RT  row tiles of 16 per group.  EPI_QKV (RT = 3) and EPI_SWIGLU (RT = 2) take tile rt from matrix
//     rt, all at the same rows; the others take RT consecutive tiles of w[0]
// JN  16-float blocks of K per wave per step (2 or 4): the prefetch unit.  Registers per lane =
//     2 * 4 JN (RT + PT) double-buffered operands + 4 RT PT accumulators.
// LD  1: row-major weights as described above.  3: weights in tile order (the model's second copy,
//     model.hip make_tiled): the A operand is one contiguous 1-KiB read per wave, no lane permute.
//     4, 5 (row-major) and 6, 7 (tile order) are TIMING PROBES of the microbenchmark with wrong results:
//     5, 6 load in the first two steps only (8: + rotated B registers; 9 / 10: only the activation / weight loads go;
//     11: the whole issue section goes), 4, 7 replace the MFMAs by one vector add per operand.
// MIX 1: the loads of the next step and their scalar bookkeeping are scheduled INTO the current step's
//     MFMA stream (one MFMA, then a few scalar / vector / memory instructions, ...) instead of in front of it: a wave
//     issues them in the shadow of its own MFMAs, whatever the SIMD's other wave is doing
template <int PT, int RT, int EPI, int JN = 2, int LD = 1, int STAGGER = 0, int MIX = 0>
__global__ __launch_bounds__(kMfThreads) void gemm_mfma_rows(MfParams p) {
    constexpr bool ACROSS = EPI == EPI_QKV || EPI == EPI_SWIGLU;
    constexpr bool PAIR = EPI == EPI_SWIGLU;
    static_assert(!PAIR || RT == 2, "SwiGLU groups are one w1 tile + one w3 tile");
    static_assert(EPI != EPI_QKV || RT == 3, "QKV groups are one tile of each of wq, wk, wv");
    constexpr int CHUNK = 16 * JN;                       // floats of K per wave per step
    constexpr bool TILED = LD == 3 || (LD >= 6 && LD <= 11);
    constexpr int PTC = PT > 4 ? 4 : PT;                 // token tiles per round of the cross-wave fold (LDS: 8 KiB per tile)
    constexpr int NTC = RT * PTC;
    __shared__ float part[kMfWaves][NTC][4][64];
    __shared__ float s_scale[PT * 16 > 64 ? PT * 16 : 64];      // per token of the pass: the rmsnorm scale (p.ssp)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows_per_grp = ACROSS ? 16 : 16 * RT;
    const int ngroups = (p.rows + rows_per_grp - 1) / rows_per_grp;
    const int total = ngroups * p.ksplit;                              // units of the launch
    const int u0 = blockIdx.x * p.nunit;
    const int nunit = min(p.nunit, total - u0);                        // uniform, >= 1 (host sizes the grid)
    const unsigned kbytes = (unsigned)p.K * 4u, mbytes = (unsigned)p.rows * kbytes;
    const int nblk = p.K >> 4;                                         // 16-float blocks along K
    const int nch = (p.K + CHUNK - 1) / CHUNK;
    const int cps = (nch + p.ksplit - 1) / p.ksplit;                   // chunks per K-slice
    const int S = (cps + kMfWaves - 1) / kMfWaves;                     // steps per unit (same for every wave)
    const int ntile = (p.n_tok + 15) >> 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (unsigned)ntile * 16u * kbytes);

    // weight reads: my (row-in-tile, k-slot) while loading, and the lane I pull from afterwards
    const int ld_row = lane >> 2, ld_kq = lane & 3;
    const int pull = (((lane & 15) << 2) | (lane >> 4)) << 2;

    f4 A0[RT][JN], A1[RT][JN], B0[PT][JN], B1[PT][JN];

    // loads of step s of local unit ul.  A step beyond the unit's chunks (odd step counts are
    // padded to pairs) or beyond the last unit is issued all the same with every offset out of
    // range: the loads return 0 without touching memory, and the number of loads in flight stays a
    // compile-time constant at every point of the loop, which lets hipcc emit counted vmcnt waits
    // instead of draining the prefetch (its waitcnt pass merges conservatively at joins).
    // wave-uniform quantities in scalar registers: with tile-order weights every address of a step is
    // (uniform offset) + 16 lane, so the loads take their per-step part through the instruction's scalar
    // offset and "out of range" through a zero-sized buffer descriptor -- no per-lane address
    // arithmetic and no exec-mask juggling between the MFMA blocks (it cost ~15 % of the issue slots)
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue = [&](f4 (&A)[RT][JN], f4 (&B)[PT][JN], int ul, int s) {
        if (LD == 11 && !(ul == 0 && s < 2)) return;                             // probe: nothing but MFMAs after the first two steps
        const int u = u0 + ul;
        const int g = p.ksplit == 1 ? u : u / p.ksplit, ks = u - g * p.ksplit;
        const int r0 = g * rows_per_grp;
        const int cl = (TILED ? wv : wave) + s * kMfWaves;                   // chunk within the K-slice
        int c = (ul < nunit && s < S && cl < cps) ? ks * cps + cl : nch;
        if ((LD == 5 || LD == 6 || LD == 8) && !(ul == 0 && s < 2)) c = nch;       // probe: only the first two steps load
        if (TILED) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const float* Wm = ACROSS ? p.w[rt] : p.w[0];
                const int rtile0 = r0 + (ACROSS ? 0 : rt * 16);
#pragma unroll
                for (int j = 0; j < JN; j++) {      // tile (rtile0 / 16), block jb of K / 16: 1 KiB, lane l at 16 l
                    const int jb = c * JN + j;
                    const bool ok = c < nch && jb < nblk && rtile0 < p.rows && !(LD == 10 && !(ul == 0 && s < 2));      // LD 10: probe without weight reads
                    const __amdgpu_buffer_rsrc_t ra = make_rsrc(Wm, ok ? mbytes : 0u);
                    A[rt][j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)lane16, (int)((unsigned)rtile0 * kbytes + (unsigned)jb * 1024u), 2));
                }
            }
#pragma unroll
            for (int pt = 0; pt < PT; pt++)
#pragma unroll
                for (int j = 0; j < JN; j++) {
                    const int jb = c * JN + j;
                    const bool ok = c < nch && jb < nblk && pt < ntile && !(LD == 9 && !(ul == 0 && s < 2));           // LD 9: probe without activation reads
                    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.x, ok ? (unsigned)ntile * 16u * kbytes : 0u);
                    B[pt][j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)lane16, (int)((unsigned)(pt * nblk + jb) * 1024u), 0));
                }
            return;
        }
        const unsigned kb0 = (unsigned)(c * CHUNK + ld_kq * 4) * 4u;           // byte offset of my float4 in load 0
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const float* Wm = ACROSS ? p.w[rt] : p.w[0];
            const __amdgpu_buffer_rsrc_t ra = make_rsrc(Wm, mbytes);
            const int r = r0 + (ACROSS ? 0 : rt * 16) + ld_row;
#pragma unroll
            for (int j = 0; j < JN; j++) {
                const unsigned kb = kb0 + (unsigned)j * 64u;
                A[rt][j] = ld_nt(ra, (c < nch && kb < kbytes && r < p.rows) ? (unsigned)r * kbytes + kb : kOOB);
            }
        }
#pragma unroll
        for (int pt = 0; pt < PT; pt++)
#pragma unroll
            for (int j = 0; j < JN; j++) {
                const int jb = c * JN + j;
                B[pt][j] = ld_c(rx, (c < nch && jb < nblk && pt < ntile) ? (unsigned)((pt * nblk + jb) * 1024 + lane * 16) : kOOB);
            }
    };

    acc4 acc[RT][PT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int pt = 0; pt < PT; pt++) acc[rt][pt] = acc4{0.f, 0.f, 0.f, 0.f};
    };

    auto compute = [&](f4 (&A)[RT][JN], f4 (&B)[PT][JN]) {
        if (LD == 4 || LD == 7) {      // probe: consume the loads with one VALU op each, no MFMA
#pragma unroll
            for (int j = 0; j < JN; j++) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++) acc[rt][0] += A[rt][j];
#pragma unroll
                for (int pt = 0; pt < PT; pt++) acc[0][pt] += B[pt][j];
            }
            return;
        }
        if (LD == 1 || LD == 5) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int j = 0; j < JN; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        A[rt][j][e] = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(A[rt][j][e])));
            // all permutes of the step are in flight before the first MFMA: left alone, hipcc
            // permutes in place one register at a time and waits for each right before its use
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < JN; j++)
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int pt = 0; pt < PT; pt++)
                        acc[rt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rt][j][e], B[pt][j][LD == 8 ? (e + 1) & 3 : e], acc[rt][pt], 0, 0, 0);      // LD 8: probe of operand register banks
    };

    // cross-wave sum of the unit's tiles + the fused epilogue.  D layout of a 16x16 tile: lane l
    // holds token (l & 15), rows 4 (l >> 4) + e -- one 16-byte slot of the output's tile layout.
    auto finish_unit = [&](int ul) {
        const int u = u0 + ul;
        const int g = u / p.ksplit, ks = u - g * p.ksplit;
        const int r0 = g * rows_per_grp;
        if (p.ssp && tid < PT * 16) {      // cpu.rs:66-79's scale per token, from the kRmsParts partial sums of squares
            float t[16];
#pragma unroll
            for (int q = 0; q < 16; q++) t[q] = p.ssp[((size_t)(tid >> 4) * 16 + q) * 16 + (tid & 15)];
#pragma unroll
            for (int n = 16; n > 1; n >>= 1)
#pragma unroll
                for (int q = 0; q < n / 2; q++) t[q] = t[2 * q] + t[2 * q + 1];
            s_scale[tid] = rms_scale(t[0], p.K);
        }
        auto total4 = [&](int tile, int ln) {
            acc4 r;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float v8[kMfWaves];
#pragma unroll
                for (int q = 0; q < kMfWaves; q++) v8[q] = part[q][tile][e][ln];
#pragma unroll
                for (int n = kMfWaves; n > 1; n >>= 1)      // fixed pairwise tree
#pragma unroll
                    for (int q = 0; q < n / 2; q++) v8[q] = v8[2 * q] + v8[2 * q + 1];
                r[e] = v8[0];
            }
            return r;
        };
        // the tiles meet in LDS PTC token tiles at a time (one round for PT <= 4)
#pragma unroll
        for (int c0 = 0; c0 < PT; c0 += PTC) {
            if (c0 > 0) __syncthreads();       // the previous round's reads are done
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int pl = 0; pl < PTC; pl++)
#pragma unroll
                    for (int e = 0; e < 4; e++) part[wave][rt * PTC + pl][e][lane] = acc[rt][c0 + pl < PT ? c0 + pl : PT - 1][e];      // (5-7 tiles: the last round's spare slots repeat a tile; their tokens are behind n_tok)
            constexpr int UNITS = (PAIR ? PTC : NTC) * 64;      // one unit = 4 consecutive rows of one token
            constexpr int ITER = (UNITS + kMfThreads - 1) / kMfThreads;
            // EPI_QKV: the rotation coefficients and positions of my units are on their way before the barrier
            // ([r5] with 7 or 8 token tiles the 84 / 96 accumulator registers are still live here and these ten more were spilled (6 / 14 registers of
            // scratch): those instantiations fetch the coefficients behind the barrier instead)
            constexpr bool EARLY_ROT = PT < 7;
            float rot[ITER][4];
            int posv[ITER];
            if (EPI == EPI_QKV && EARLY_ROT) {
#pragma unroll
                for (int it = 0; it < ITER; it++) {
                    const int v = tid + it * kMfThreads, ln = v & 63, tile = v >> 6;
                    const int rt = tile / PTC, tk = (c0 + tile - rt * PTC) * 16 + (ln & 15), r = r0 + (ln >> 4) * 4;
                    rot[it][0] = rot[it][1] = rot[it][2] = rot[it][3] = 0.0f; posv[it] = 0;
                    if (v < UNITS && tk < p.n_tok && r < p.rows) {
                        posv[it] = p.seqs ? p.seqs[tk].pos : p.pos0 + tk;
                        if (rt < 2) {
                            const size_t fo = (size_t)posv[it] * (p.head_size >> 1) + ((r % p.head_size) >> 1);
                            rot[it][0] = p.fr[fo]; rot[it][1] = p.fi[fo]; rot[it][2] = p.fr[fo + 1]; rot[it][3] = p.fi[fo + 1];
                        }
                    }
                }
            }
            if (c0 == 0) MF_STAMP(2);
            __syncthreads();
            if (c0 == 0) MF_STAMP(3);
#pragma unroll
            for (int it = 0; it < ITER; it++) {
                const int v = tid + it * kMfThreads;
                if (v >= UNITS) break;
                const int ln = v & 63, tile = v >> 6;
                const int rt = PAIR ? 0 : tile / PTC, pl = PAIR ? tile : tile - rt * PTC, pt = c0 + pl;
                const int tk = pt * 16 + (ln & 15), r = r0 + (ACROSS ? 0 : rt * 16) + (ln >> 4) * 4;     // rows r .. r + 3
                if (tk >= p.n_tok || r >= p.rows) continue;     // rows % 4 == 0: a unit is all in or all out
                acc4 a = total4(PAIR ? pl : tile, ln);
                const float nv = p.ssp ? s_scale[pt * 16 + (ln & 15)] : 1.0f;
                if (p.ssp) { a[0] *= nv; a[1] *= nv; a[2] *= nv; a[3] *= nv; }
                if (PAIR) {
                    acc4 b = total4(PTC + pl, ln);
                    if (p.ssp) { b[0] *= nv; b[1] *= nv; b[2] *= nv; b[3] *= nv; }
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float sg = a[e] * (1.0f / (1.0f + expf(-a[e])));      // cpu.rs:56
                        a[e] = sg * b[e];                                              // cpu.rs:59-64
                    }
                    *reinterpret_cast<acc4*>(p.o + tile_idx(tk, r, p.rows)) = a;
                } else if (EPI == EPI_QKV) {
                    if (!EARLY_ROT) {
                        posv[it] = p.seqs ? p.seqs[tk].pos : p.pos0 + tk;
                        rot[it][0] = rot[it][1] = rot[it][2] = rot[it][3] = 0.0f;
                        if (rt < 2) {
                            const size_t fo = (size_t)posv[it] * (p.head_size >> 1) + ((r % p.head_size) >> 1);
                            rot[it][0] = p.fr[fo]; rot[it][1] = p.fi[fo]; rot[it][2] = p.fr[fo + 1]; rot[it][3] = p.fi[fo + 1];
                        }
                    }
                    const int pos = posv[it];
                    if (rt < 2) {                                                      // cpu.rs:87-96 rotate (q, k)
                        const float c0_ = rot[it][0], s0 = rot[it][1], c1 = rot[it][2], s1 = rot[it][3];
                        const acc4 t = a;
                        a[0] = t[0] * c0_ - t[1] * s0; a[1] = t[0] * s0 + t[1] * c0_;
                        a[2] = t[2] * c1 - t[3] * s1; a[3] = t[2] * s1 + t[3] * c1;
                    }
                    if (rt == 0) *reinterpret_cast<acc4*>(p.o + tile_idx(tk, r, p.rows)) = a;
                    else {                                                             // infer.rs:32-33
                        float* cache = (rt == 1 ? (p.seqs ? p.seqs[tk].kc + p.layer_off : p.kc) : (p.seqs ? p.seqs[tk].vc + p.layer_off : p.vc));
                        *reinterpret_cast<acc4*>(cache + (size_t)pos * p.rows + r) = a;
                    }
                } else if (EPI == EPI_STORE_ROWS) {
                    *reinterpret_cast<acc4*>(p.o + (size_t)tk * p.o_stride + r) = a;
                } else {                                                               // K-slice ks of the product, tile layout
                    *reinterpret_cast<acc4*>(p.o + (size_t)ks * p.slab_floats + tile_idx(tk, r, p.rows)) = a;
                }
            }
        }
        __syncthreads();       // part[] is rewritten by the next unit
    };

    // MIX: the order of one step's region -- every load within the first MFMAs, the scalar work spread under all of them
    auto mix = [&]() {
        constexpr int NMF = 4 * JN * RT * PT, NLD = JN * (RT + PT);
#pragma unroll
        for (int i = 0; i < NMF; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x004, MIX, 0);                  // scalar ALU
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                    // vector ALU
            if (i < NLD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // one load
        }
    };

    // steps of a unit in pairs (buffer 0, buffer 1); the loads of the step after next are issued
    // before the current step's MFMAs, the first loads of the next unit before this unit's epilogue
    const int S2 = (S + 1) >> 1;
    MF_STAMP(0);
    MF_BLOCK_STAMP(0);
    issue(A0, B0, 0, 0);
#pragma unroll 1
    for (int ul = 0; ul < nunit; ul++) {
        zero_acc();
        // the two waves of a SIMD run the same program: half a step of delay for waves 4-7 lets one
        // wave's loads / permutes fall under the other's MFMAs instead of both stalling together
        if (STAGGER > 0 && wave >= kMfWaves / 2) __builtin_amdgcn_s_sleep(STAGGER);
        if (STAGGER == -1 && wv >= kMfWaves / 2) __builtin_amdgcn_s_setprio(1);      // the SIMD's younger wave first (wv: wave-uniform, or the branch is an exec mask around an unconditional s_setprio)
#pragma unroll 1
        for (int i = 0; i < S2; i++) {
            __builtin_amdgcn_sched_barrier(0);
            if (STAGGER == -2) { if ((i & 1) == (wv >= kMfWaves / 2 ? 1 : 0)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }      // probe: the two waves of a SIMD take turns per pair of steps
            if (STAGGER == -3) { if (((i >> 2) & 1) == (wv >= kMfWaves / 2 ? 1 : 0)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }      // ... per 8 steps
            issue(A1, B1, ul, 2 * i + 1);
            if (MIX == 0) __builtin_amdgcn_sched_barrier(0);
            compute(A0, B0);
            if (MIX != 0) mix();
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < S2) issue(A0, B0, ul, 2 * i + 2); else issue(A0, B0, ul + 1, 0);
            if (MIX == 0) __builtin_amdgcn_sched_barrier(0);
            compute(A1, B1);
            if (MIX != 0) mix();
        }
        MF_STAMP(1);
        finish_unit(ul);
        MF_STAMP(4);
    }
    MF_BLOCK_STAMP(1);
}

// ---- small kernels on the tile layout

// X[t] = token_embedding_table[tokens[t]]   (infer.rs:13 per token), tile layout
__global__ void embed_tile_kernel(float* X, const float* emb, const int* tokens, int n_tok, int dim) {
    const int t = blockIdx.y;
    if (t >= n_tok) return;
    const size_t base = (size_t)tokens[t] * dim;
    for (int k = (blockIdx.x * blockDim.x + threadIdx.x) * 4; k < dim; k += gridDim.x * blockDim.x * 4)
        *reinterpret_cast<f4*>(X + tile_idx(t, k, dim)) = *reinterpret_cast<const f4*>(emb + base + k);
}

// row-major [n_tok, K] <-> tile layout (tests, the microbenchmark, and the last prompt position's
// residual stream on its way to the classifier)
__global__ void tile_rows_kernel(float* T, const float* rows, int n_tok, int K) {
    const int t = blockIdx.y;
    for (int k = (blockIdx.x * blockDim.x + threadIdx.x) * 4; k < K; k += gridDim.x * blockDim.x * 4)
        *reinterpret_cast<f4*>(T + tile_idx(t, k, K)) = *reinterpret_cast<const f4*>(rows + (size_t)t * K + k);
}
__global__ void untile_rows_kernel(float* rows, const float* T, int tk0, int K) {
    const int t = blockIdx.y;
    for (int k = (blockIdx.x * blockDim.x + threadIdx.x) * 4; k < K; k += gridDim.x * blockDim.x * 4)
        *reinterpret_cast<f4*>(rows + (size_t)t * K + k) = *reinterpret_cast<const f4*>(T + tile_idx(tk0 + t, k, K));
}

// x += (s0 + s1 + ..): the K-slices of the preceding Wo / W2 product (its sum first, then the residual
// add, infer.rs:35-37 / 46-47), on the tile layout; any element order -- both sides use the same one
__device__ __forceinline__ f4 fold_slabs(f4 x, const float* slabs, int nslab, size_t slab_floats, size_t o) {
    if (nslab > 0) {
        f4 t = *reinterpret_cast<const f4*>(slabs + o);
        for (int k = 1; k < nslab; k++) t = t + *reinterpret_cast<const f4*>(slabs + (size_t)k * slab_floats + o);
        x = x + t;
    }
    return x;
}

// cpu.rs:99-117 per token on the tile layout, as two small launches over (token tile, column part)
// workgroups so that the ~1 MB of activations is spread over 64 CUs instead of 4:
//   rms_fold_kernel   x += pending K-slices (written back); partial sum x^2 per (part, token)
//   rms_scale_kernel  v = 1 / sqrt(sum of the parts (fixed order) / n + 1e-5); o = w * (v * x)
constexpr int kRmsParts = 16;

// X += pending K-slices; partial sums of squares per (token tile, part, token) -> ssp; and, with a gain
// vector, XN = X * gain (infer.rs:19/39/50's rmsnorm without its per-token scale, which the consuming
// GEMM applies to its outputs: W (v g x) = v W (g x), as the decode matvecs do)
constexpr int kRmsFoldWaves = 16;      // a part of llama2-7B's rows is 16 column blocks: one per wave, every load of the launch in flight at once
__global__ __launch_bounds__(kRmsFoldWaves * 64) void rms_fold_kernel(float* X, float* ssp, int dim, const float* slabs, int nslab, size_t slab_floats,
                                                                      float* XN = nullptr, const float* gain = nullptr) {
    __shared__ float red[kRmsFoldWaves][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = dim >> 4, per = (nblk + kRmsParts - 1) / kRmsParts;
    const int jb0 = blockIdx.y * per, jb1 = min(jb0 + per, nblk);
    const size_t tbase = (size_t)blockIdx.x * nblk * 256;
    float ss = 0.0f;
    for (int jb = jb0 + wave; jb < jb1; jb += kRmsFoldWaves) {
        const size_t o = tbase + (size_t)jb * 256 + lane * 4;
        f4 x = *reinterpret_cast<const f4*>(X + o);
        if (nslab > 0) {
            x = fold_slabs(x, slabs, nslab, slab_floats, o);
            *reinterpret_cast<f4*>(X + o) = x;
        }
        if (XN) {
            const f4 g = *reinterpret_cast<const f4*>(gain + jb * 16 + (lane >> 4) * 4);
            f4 r;
            r.x = g.x * x.x; r.y = g.y * x.y; r.z = g.z * x.z; r.w = g.w * x.w;
            *reinterpret_cast<f4*>(XN + o) = r;
        }
        ss = dot4(x, x, ss);
    }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    if (lane < 16) red[wave][lane] = ss;
    __syncthreads();
    if (tid < 16) {
        float t[kRmsFoldWaves];
#pragma unroll
        for (int q = 0; q < kRmsFoldWaves; q++) t[q] = red[q][tid];
#pragma unroll
        for (int n = kRmsFoldWaves; n > 1; n >>= 1)      // fixed pairwise tree
#pragma unroll
            for (int q = 0; q < n / 2; q++) t[q] = t[2 * q] + t[2 * q + 1];
        ssp[((size_t)blockIdx.x * kRmsParts + blockIdx.y) * 16 + tid] = t[0];
    }
}

__global__ __launch_bounds__(256) void rms_scale_kernel(float* O, const float* X, const float* w, const float* ssp, int dim) {
    __shared__ float s_v[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = dim >> 4, per = (nblk + kRmsParts - 1) / kRmsParts;
    const int jb0 = blockIdx.y * per, jb1 = min(jb0 + per, nblk);
    const size_t tbase = (size_t)blockIdx.x * nblk * 256;
    if (tid < 16) {
        float t[kRmsParts];
#pragma unroll
        for (int q = 0; q < kRmsParts; q++) t[q] = ssp[((size_t)blockIdx.x * kRmsParts + q) * 16 + tid];
#pragma unroll
        for (int n = kRmsParts; n > 1; n >>= 1)
#pragma unroll
            for (int q = 0; q < n / 2; q++) t[q] = t[2 * q] + t[2 * q + 1];
        s_v[tid] = rms_scale(t[0], dim);
    }
    __syncthreads();
    const float v = s_v[lane & 15];
    for (int jb = jb0 + wave; jb < jb1; jb += 4) {
        const size_t o = tbase + (size_t)jb * 256 + lane * 4;
        const f4 x = *reinterpret_cast<const f4*>(X + o);
        const f4 g = *reinterpret_cast<const f4*>(w + jb * 16 + (lane >> 4) * 4);
        f4 r;
        r.x = g.x * (v * x.x); r.y = g.y * (v * x.y); r.z = g.z * (v * x.z); r.w = g.w * (v * x.w);
        *reinterpret_cast<f4*>(O + o) = r;
    }
}

// rows[t] = X[tk0 + t] (+ pending K-slices): the residual stream of tokens on its way to the
// row-major classifier path
__global__ void untile_fold_kernel(float* rows, const float* T, int tk0, int K, const float* slabs, int nslab, size_t slab_floats) {
    const int t = blockIdx.y;
    for (int k = (blockIdx.x * blockDim.x + threadIdx.x) * 4; k < K; k += gridDim.x * blockDim.x * 4) {
        const size_t o = tile_idx(tk0 + t, k, K);
        *reinterpret_cast<f4*>(rows + (size_t)t * K + k) = fold_slabs(*reinterpret_cast<const f4*>(T + o), slabs, nslab, slab_floats, o);
    }
}

}  // namespace rama
