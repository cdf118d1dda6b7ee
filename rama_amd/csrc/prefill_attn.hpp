// Attention for the queries of one prefill pass (infer.rs:34, cpu.rs:23-52 for P consecutive positions
// of ONE sequence) as fp32 MFMA tiles: a workgroup owns 16 queries of one head and walks the key
// cache once for all of them, instead of once per query as the decode kernel does (64 queries of a
// pass share every K/V row: 2.1 GB of cache reads per layer at 1024 tokens of context become 0.26).
//
//   S^T = K Q^T     16 keys x 16 queries per v_mfma_f32_16x16x4_f32 chain over head_size;
//                   A = cache rows (lane -> key l&15, 4 floats at k-slot l>>4), B = the queries straight
//                   from the GEMMs' tile layout (lane -> query l&15): register e of both loads feeds
//                   MFMA e, the sum over k is order-free.
//   softmax         online (running max m, running sum l, O rescaled when m grows), per query = per
//                   lane column: the D tile holds S^T[key 4(l>>4)+r][query l&15], so a query's 16 keys
//                   are 4 registers x 4 lane groups -- and that D tile IS the B operand of the next
//                   product (k-slot l>>4, register r <-> key 4(l>>4)+r).
//   O^T = V^T P^T   A = V[t][d] (lane -> d l&15 of a 16-float block, key 4(l>>4)+r for MFMA r); the D
//                   tile (query l&15, d = 4(l>>4)+r) is the tile layout of xb: one 16-byte store per lane.
//
// The 4 or 8 waves of a workgroup take key tiles in turn and fold (m, l, O) through LDS at the end.
// Scores are divided by sqrt(head_size) like cpu.rs:40; exp is the device expf of the decode kernel.
#pragma once
#include "kernels.hpp"
#include "prefill_mfma.hpp"

namespace rama {

template <int NB, int kTileAttnWaves>      // head_size / 16; waves per workgroup (4: short contexts, 8: >= 512 keys)
__global__ __launch_bounds__(kTileAttnWaves * 64) void attention_tile_mfma_kernel(AttnParams p, int nt) {
    __shared__ float s_m[kTileAttnWaves][16], s_l[kTileAttnWaves][16];
    __shared__ acc4 s_o[kTileAttnWaves][NB][64];
    const int h = blockIdx.x, qt = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: scalar loop control and tile addresses
    const int i = lane & 15, g = lane >> 4;
    const int hs = NB * 16, dim = p.dim;
    const int q_tok = qt * 16 + i;                         // my query (column of every tile)
    const int pos_i = p.pos_val + q_tok;
    const int last_pos = p.pos_val + min(qt * 16 + 15, nt - 1);
    const int nkt = last_pos / 16 + 1;                     // key tiles any query of this tile attends to
    const float div = sqrtf((float)hs);
    const float* kc = p.kc + (size_t)h * hs;
    const float* vc = p.vc + (size_t)h * hs;

    f4 qb[NB];
#pragma unroll
    for (int kb = 0; kb < NB; kb++) qb[kb] = *reinterpret_cast<const f4*>(p.q + attn_tile_idx(q_tok, h * hs + 16 * kb + 4 * g, dim));

    acc4 o[NB];
#pragma unroll
    for (int db = 0; db < NB; db++) o[db] = acc4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.0f;

    for (int kt = wave; kt < nkt; kt += kTileAttnWaves) {
        const int t0 = kt * 16;
        // K tile: my row is key t0 + i; V tile: my rows are keys t0 + 4g + r, my column d = 16 db + i
        const size_t krow = (size_t)min(t0 + i, p.seq_len - 1) * dim;
        f4 ka[NB];
#pragma unroll
        for (int kb = 0; kb < NB; kb++) ka[kb] = *reinterpret_cast<const f4*>(kc + krow + 16 * kb + 4 * g);
        float va[NB][4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const size_t vrow = (size_t)min(t0 + 4 * g + r, p.seq_len - 1) * dim;
#pragma unroll
            for (int db = 0; db < NB; db++) va[db][r] = vc[vrow + 16 * db + i];
        }
        acc4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NB; kb++) {
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[kb].x, qb[kb].x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[kb].y, qb[kb].y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[kb].z, qb[kb].z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[kb].w, qb[kb].w, s, 0, 0, 0);
        }
        // scores of keys t0 + 4g + r for query i; keys behind the query's position do not exist for it
        float sv[4];
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            sv[r] = (t0 + 4 * g + r <= pos_i) ? s[r] / div : -INFINITY;
            mt = fmaxf(mt, sv[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m, mt);
        // no branch around the MFMAs: a query that sees nothing yet (m_new = -inf) gets p = 0, scale 0
        const float m_ref = m_new == -INFINITY ? 0.0f : m_new;
        const float sc = m == -INFINITY ? 0.0f : expf(m - m_ref);
        float pr[4];
        float ps = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) { pr[r] = sv[r] == -INFINITY ? 0.0f : expf(sv[r] - m_ref); ps += pr[r]; }
        l = l * sc + ps;
        m = m_new;
#pragma unroll
        for (int db = 0; db < NB; db++) {
            acc4 a = o[db];
            a.x *= sc; a.y *= sc; a.z *= sc; a.w *= sc;
#pragma unroll
            for (int r = 0; r < 4; r++) a = __builtin_amdgcn_mfma_f32_16x16x4f32(va[db][r], pr[r], a, 0, 0, 0);
            o[db] = a;
        }
    }
    // my lane group's share of the sum -> the query's sum
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (g == 0) { s_m[wave][i] = m; s_l[wave][i] = l; }
#pragma unroll
    for (int db = 0; db < NB; db++) s_o[wave][db][lane] = o[db];
    __syncthreads();
    // fold the waves: wave w finishes the d-blocks db = w, w + W, ...
    float M = s_m[0][i];
#pragma unroll
    for (int w = 1; w < kTileAttnWaves; w++) M = fmaxf(M, s_m[w][i]);
    float wgt[kTileAttnWaves];
    float L = 0.0f;
#pragma unroll
    for (int w = 0; w < kTileAttnWaves; w++) {
        const float mw = s_m[w][i];
        wgt[w] = mw == -INFINITY ? 0.0f : expf(mw - M);
        L += wgt[w] * s_l[w][i];
    }
    {
#pragma unroll
        for (int db = 0; db < NB; db++) {
            if (db % kTileAttnWaves != wave) continue;
            acc4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < kTileAttnWaves; w++) {
                const acc4 ow = s_o[w][db][lane];
                a.x += wgt[w] * ow.x; a.y += wgt[w] * ow.y; a.z += wgt[w] * ow.z; a.w += wgt[w] * ow.w;
            }
            a.x /= L; a.y /= L; a.z /= L; a.w /= L;
            *reinterpret_cast<acc4*>(p.xb + attn_tile_idx(q_tok, h * hs + 16 * db + 4 * g, dim)) = a;
        }
    }
}

}  // namespace rama
