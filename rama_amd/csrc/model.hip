// model.hip -- weights in HBM: llama2.c v0 checkpoint loader (mmap + one staged H2D copy)
// and the synthetic-weight generator.  Replaces engine/src/transformer/ram.rs:27-52
// (TransformerWeights::from_file, 4-byte read_exact loop of utils/read.rs:25-33) and
// hbm.rs:55-90 (14 htod_sync_copy calls).
#include "../../include/rama_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <mutex>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

extern "C" int rama_fill_synth(rama_ctx*, float*, size_t, uint64_t, uint64_t, uint64_t, float, float);

struct rama_model {
    rama_config cfg{};
    rama_weights w{};
    float* w13i = nullptr;      // W1 | W3 row-interleaved per layer: [n_local_layers, hidden, 2, dim] (see below)
    float* tiled = nullptr;     // every matrix once more in MFMA tile order, for the token-batch GEMMs (see below); made on first use
    float* chain = nullptr;     // every matrix once more in chain order, for parity mode (chain.hpp); made on first use
    bool tiled_tried = false, chain_tried = false;
    std::mutex build_mu;        // held while a derived copy is being made: a second caller waits for the copy instead of missing it
    float* blob = nullptr;      // one allocation holding every tensor (NULL: an ADOPTED model -- tensors uploaded one by one, owned by the caller)
    size_t blob_floats = 0;
    rama_stage stage{};
    bool adopted = false;
};

namespace {

thread_local std::string g_model_err;

struct TensorSpec { const char* name; size_t n; size_t per_layer; int tag; double std; float bias; };

// tensor order of the v0 file = struct-literal order in ram.rs:31-49
std::vector<TensorSpec> v0_layout(const rama_config& c) {
    const size_t L = c.n_layers, d = c.dim, h = c.hidden_dim, V = c.vocab_size, S = c.seq_len;
    const size_t hs = d / c.n_heads;
    const double res = 0.02 / std::sqrt(2.0 * (double)L);   // model.py:232-236
    std::vector<TensorSpec> t = {
        {"token_embedding_table", V * d, 0, 1, 0.02, 0.f},
        {"rms_att_weight", L * d, d, 2, 0.05, 1.f},
        {"wq", L * d * d, d * d, 3, 0.02, 0.f},
        {"wk", L * d * d, d * d, 4, 0.02, 0.f},
        {"wv", L * d * d, d * d, 5, 0.02, 0.f},
        {"wo", L * d * d, d * d, 6, res, 0.f},
        {"rms_ffn_weight", L * d, d, 7, 0.05, 1.f},
        {"w1", L * h * d, h * d, 8, 0.02, 0.f},
        {"w2", L * d * h, d * h, 9, 0.02, 0.f},
        {"w3", L * h * d, h * d, 10, res, 0.f},
        {"rms_final_weight", d, 0, 11, 0.05, 1.f},
        {"freq_cis_real", S * (hs / 2), 0, -1, 0, 0.f},
        {"freq_cis_imag", S * (hs / 2), 0, -1, 0, 0.f},
    };
    if (!c.shared_weight) t.push_back({"wcls", V * d, 0, 12, 0.02, 0.f});
    return t;
}

const float** field(rama_weights& w, const char* name) {
#define F(n) if (!strcmp(name, #n)) return &w.n;
    F(token_embedding_table) F(rms_att_weight) F(rms_ffn_weight) F(wq) F(wk) F(wv) F(wo) F(w1) F(w2) F(w3)
    F(rms_final_weight) F(freq_cis_real) F(freq_cis_imag) F(wcls)
#undef F
    return nullptr;
}

// ---- W1 | W3 interleaved.  infer.rs:41-45 streams W1 and W3 against the same activations; in the
// checkpoint they are two tensors gigabytes apart, and where their pages fall onto HBM channels made
// the fused kernel's time differ by 6 % from allocation to allocation.  A model therefore keeps one
// extra copy with row i of W1 followed by row i of W3 -- the fused launch then streams ONE contiguous
// 4-row block per workgroup like every other matvec (+11.5 GB at llama2-7B of 288).  w1 / w3 in
// rama_weights stay the checkpoint's tensors (the 1:1 trait ops use them); the fused path finds the
// interleaved copy through this registry, keyed by the two tensor addresses.
struct W13Entry { const float* w1; const float* w3; const float* w13i; };
std::vector<W13Entry> g_w13;
std::mutex g_w13_mu;

__global__ void interleave_rows_kernel(float* dst, const float* w1, const float* w3, size_t rows, int K) {
    const size_t n4 = rows * (size_t)(K / 4);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (K / 4), k4 = i - r * (K / 4);
        const float4 a = reinterpret_cast<const float4*>(w1)[i], b = reinterpret_cast<const float4*>(w3)[i];
        reinterpret_cast<float4*>(dst)[(2 * r) * (K / 4) + k4] = a;
        reinterpret_cast<float4*>(dst)[(2 * r + 1) * (K / 4) + k4] = b;
    }
}

int bad(int code, const char* msg) { fprintf(stderr, "rama_model: %s\n", msg); return code; }

size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }

// build + register the interleaved W1 | W3 copy of a freshly created model (no-op without FFN layers)
int make_w13i(rama_ctx* ctx, rama_model* m) {
    const size_t nl = (size_t)(m->stage.layer_end - m->stage.layer_begin);
    if (!nl || !m->w.w1 || !m->w.w3 || m->cfg.dim % 4) return 0;
    const size_t rows = nl * (size_t)m->cfg.hidden_dim;
    int rc = rama_alloc_f32(ctx, 2 * rows * (size_t)m->cfg.dim, &m->w13i);
    if (rc) { m->w13i = nullptr; return 0; }      // no room for the copy: the two-tensor kernel still works
    rama_sync(ctx);
    hipLaunchKernelGGL(interleave_rows_kernel, dim3(4096), dim3(256), 0, 0, m->w13i, m->w.w1, m->w.w3, rows, (int)m->cfg.dim);
    if (hipDeviceSynchronize() != hipSuccess) { rama_free(ctx, m->w13i); m->w13i = nullptr; return bad(RAMA_EIO, "interleaving W1 | W3 failed"); }
    std::lock_guard<std::mutex> lk(g_w13_mu);
    g_w13.push_back({m->w.w1, m->w.w3, m->w13i});
    return 0;
}

// ---- the matrices in MFMA tile order.  The token-batch GEMMs (prefill_mfma.hpp) want the weight tile of
// a v_mfma_f32_16x16x4_f32 chain -- 16 rows x 16 floats, lane = row + 16 (k / 4 % 4), register = k % 4 --
// and a wave that collects it from row-major memory reads 16 rows x 64 bytes per instruction: at 16
// tokens per pass that access pattern alone holds the launch at 4.8 TB/s where the decode matvecs
// stream 6.4-6.8.  A model therefore keeps a second copy of wq, wk, wv, wo, w1, w3, w2 (and wcls) in
// exactly the activations' tile layout (tile_idx): the A operand is then one fully coalesced 1-KiB read
// per wave with no lane permute (+27 GB at llama2-7B of 288; skipped when memory is short or
// RAMA_NO_TILED is set -- the row-major kernels remain).  Found through a registry keyed by the
// row-major tensor's base address.
struct TiledEntry { const float* src; const float* tiled; };
std::vector<TiledEntry> g_tiled;
std::mutex g_tiled_mu;

__global__ void tile_weights_kernel(float* dst, const float* src, size_t nmat, int rows, int K) {
    const size_t per = (size_t)rows * K, n4 = nmat * per / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4, l = e / per, in = e - l * per;
        const int r = (int)(in / K), k = (int)(in - (size_t)r * K);
        // tile_idx(r, k, K) of prefill_mfma.hpp
        const size_t t = ((size_t)((r >> 4) * (K >> 4) + (k >> 4)) * 64 + (size_t)(((k >> 2) & 3) * 16 + (r & 15))) * 4;
        *reinterpret_cast<float4*>(dst + l * per + t) = *reinterpret_cast<const float4*>(src + e);
    }
}

int make_tiled(rama_ctx* ctx, rama_model* m) {
    if (getenv("RAMA_NO_TILED")) return 0;
    const size_t nl = (size_t)(m->stage.layer_end - m->stage.layer_begin);
    const int dim = m->cfg.dim, hidden = m->cfg.hidden_dim, V = m->cfg.vocab_size;
    if (dim % 16 || hidden % 16) return 0;
    const size_t dd = (size_t)dim * dim, hd = (size_t)hidden * dim;
    const bool cls = m->stage.do_cls && m->w.wcls && V % 16 == 0;
    const size_t total = nl * (4 * dd + 3 * hd) + (cls ? (size_t)V * dim : 0);
    if (!total) return 0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < total * 4 + ((size_t)16 << 30)) return 0;   // keep 16 GiB for states and scratch
    if (rama_alloc_f32(ctx, total, &m->tiled)) { m->tiled = nullptr; return 0; }
    rama_sync(ctx);
    struct T { const float* src; size_t nmat; int rows, K; };
    std::vector<T> ts;
    if (nl) {
        ts.push_back({m->w.wq, nl, dim, dim}); ts.push_back({m->w.wk, nl, dim, dim}); ts.push_back({m->w.wv, nl, dim, dim});
        ts.push_back({m->w.wo, nl, dim, dim}); ts.push_back({m->w.w1, nl, hidden, dim}); ts.push_back({m->w.w3, nl, hidden, dim});
        ts.push_back({m->w.w2, nl, dim, hidden});
    }
    if (cls) ts.push_back({m->w.wcls, 1, V, dim});
    float* dst = m->tiled;
    std::vector<TiledEntry> mine;
    for (const T& t : ts) {
        if (!t.src) { rama_free(ctx, m->tiled); m->tiled = nullptr; return 0; }
        hipLaunchKernelGGL(tile_weights_kernel, dim3(4096), dim3(256), 0, 0, dst, t.src, t.nmat, t.rows, t.K);
        mine.push_back({t.src, dst});
        dst += t.nmat * (size_t)t.rows * t.K;
    }
    if (hipDeviceSynchronize() != hipSuccess) { rama_free(ctx, m->tiled); m->tiled = nullptr; return bad(RAMA_EIO, "tiling the weights failed"); }
    std::lock_guard<std::mutex> lk(g_tiled_mu);
    for (auto& e : mine) g_tiled.push_back(e);
    return 0;
}

// ---------------------------------------------------------------- chain-order weight copy
// dst[(g * (K/16) + s) * 256 + lane * 4 + t] = src_row(16 g + lane / 4)[16 s + 4 t + lane % 4]; rows beyond
// `rows` read as zero.  INTER: source row r is row r / 2 of w1 (r even) or of w3 (r odd) -- the
// (W1 row i, W3 row i) pairs infer.rs:41-45 consumes together land in neighbouring quads.
template <bool INTER>
__global__ void chain_weights_kernel(float* dst, const float* src, const float* src2, size_t nmat, int rows, int K) {
    const int nblk = K >> 4, groups = (rows + 15) >> 4;
    const size_t per_dst = (size_t)groups * 16 * K, per_src = (size_t)(INTER ? rows / 2 : rows) * K;
    const size_t n4 = nmat * per_dst / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t l = i / (per_dst / 4), in = i - l * (per_dst / 4);
        const int lane = (int)(in & 63);
        const size_t blk = in >> 6;
        const int g = (int)(blk / nblk), s = (int)(blk - (size_t)g * nblk);
        const int r = 16 * g + (lane >> 2), j = lane & 3;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows) {
            const float* row = INTER ? ((r & 1) ? src2 : src) + l * per_src + (size_t)(r >> 1) * K
                                     : src + l * per_src + (size_t)r * K;
            v.x = row[16 * s + j]; v.y = row[16 * s + 4 + j]; v.z = row[16 * s + 8 + j]; v.w = row[16 * s + 12 + j];
        }
        *reinterpret_cast<float4*>(dst + i * 4) = v;
    }
}


// registry of the chain-order copies: a row-major tensor [nmat][rows][K] (or the (w1, w3) pair, rows = 2 hidden) -> its copy
// (own: a copy made for a VIEW -- rama_internal_chain_view, weights that belong to no model -- is an allocation of its own)
struct ChainEntry { const float* src; const float* src2; size_t per_src; size_t nmat; int rows, K; const float* chain; float* own; };
std::vector<ChainEntry> g_chain;
std::mutex g_chain_mu;

// every live rama_alloc_f32 allocation (base -> floats): lets a view of a tensor that was uploaded by itself (hbm.rs:14-16) be traced to its tensor
std::map<const float*, size_t> g_allocs;
std::mutex g_allocs_mu;
bool alloc_of(const float* p, const float** base, size_t* n) {
    std::lock_guard<std::mutex> lk(g_allocs_mu);
    auto it = g_allocs.upper_bound(p);
    if (it == g_allocs.begin()) return false;
    --it;
    if (p >= it->first + it->second) return false;
    *base = it->first; *n = it->second;
    return true;
}

// advances whenever a derived copy (chain order, tile order, a view's) is freed: captured graphs hold the copies' addresses, and the context that
// frees them (rama_model_release_copies, rama_model_free, rama_free of an adopted tensor) drops only its OWN graphs -- every other context
// compares its graphs' generation with this one before a replay (rama_api.hip same_capture) and captures again
std::atomic<unsigned long long> g_copies_gen{1};

std::vector<rama_model*> g_models;     // every live model: lets the lazily made copies be found from a rama_weights
std::mutex g_models_mu;

int make_chain(rama_ctx* ctx, rama_model* m) {
    if (getenv("RAMA_NO_CHAIN")) return 0;
    const size_t nl = (size_t)(m->stage.layer_end - m->stage.layer_begin);
    const int dim = m->cfg.dim, hidden = m->cfg.hidden_dim, V = m->cfg.vocab_size;
    if (dim % 16 || hidden % 16) return 0;          // the reference-order fallback kernels take it
    auto r16 = [](int r) { return (size_t)((r + 15) / 16) * 16; };
    const bool cls = m->stage.do_cls && m->w.wcls;
    struct T { const float* src; const float* src2; size_t nmat; int rows, K; };
    std::vector<T> ts;
    if (nl) {
        if (!m->w.wq || !m->w.wk || !m->w.wv || !m->w.wo || !m->w.w1 || !m->w.w2 || !m->w.w3) return 0;
        ts.push_back({m->w.wq, nullptr, nl, dim, dim}); ts.push_back({m->w.wk, nullptr, nl, dim, dim}); ts.push_back({m->w.wv, nullptr, nl, dim, dim});
        ts.push_back({m->w.wo, nullptr, nl, dim, dim}); ts.push_back({m->w.w1, m->w.w3, nl, 2 * hidden, dim});
        ts.push_back({m->w.w2, nullptr, nl, dim, hidden});
    }
    if (cls) ts.push_back({m->w.wcls, nullptr, 1, V, dim});
    size_t total = 0;
    for (const T& t : ts) total += t.nmat * r16(t.rows) * (size_t)t.K;
    if (!total) return 0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < total * 4 + ((size_t)16 << 30)) return 0;   // keep 16 GiB for states and scratch
    if (rama_alloc_f32(ctx, total, &m->chain)) { m->chain = nullptr; return 0; }
    rama_sync(ctx);
    float* dst = m->chain;
    std::vector<ChainEntry> mine;
    for (const T& t : ts) {
        if (t.src2) hipLaunchKernelGGL(chain_weights_kernel<true>, dim3(4096), dim3(256), 0, 0, dst, t.src, t.src2, t.nmat, t.rows, t.K);
        else hipLaunchKernelGGL(chain_weights_kernel<false>, dim3(4096), dim3(256), 0, 0, dst, t.src, t.src2, t.nmat, t.rows, t.K);
        mine.push_back({t.src, t.src2, (size_t)(t.src2 ? t.rows / 2 : t.rows) * t.K, t.nmat, t.rows, t.K, dst, nullptr});
        dst += t.nmat * r16(t.rows) * (size_t)t.K;
    }
    if (hipDeviceSynchronize() != hipSuccess) { rama_free(ctx, m->chain); m->chain = nullptr; return bad(RAMA_EIO, "chain-ordering the weights failed"); }
    std::lock_guard<std::mutex> lk(g_chain_mu);
    for (auto& e : mine) g_chain.push_back(e);
    return 0;
}

void drop_chain(rama_ctx* ctx, rama_model* m) {
    if (!m->chain) return;
    g_copies_gen++;
    {
        std::lock_guard<std::mutex> lk(g_chain_mu);
        const float* lo = m->chain;
        for (size_t i = g_chain.size(); i-- > 0;)
            if (!g_chain[i].own && g_chain[i].chain >= lo && (g_chain[i].src == m->w.wq || g_chain[i].src == m->w.wk || g_chain[i].src == m->w.wv || g_chain[i].src == m->w.wo ||
                                           g_chain[i].src == m->w.w1 || g_chain[i].src == m->w.w2 || g_chain[i].src == m->w.wcls))
                g_chain.erase(g_chain.begin() + (long)i);
    }
    rama_free(ctx, m->chain);
    m->chain = nullptr;
}

void drop_tiled(rama_ctx* ctx, rama_model* m) {
    if (!m->tiled) return;
    g_copies_gen++;
    {
        std::lock_guard<std::mutex> lk(g_tiled_mu);
        const float* lo = m->tiled;
        for (size_t i = g_tiled.size(); i-- > 0;)      // every entry of this model points into its one allocation
            if (g_tiled[i].src == m->w.wq || g_tiled[i].src == m->w.wk || g_tiled[i].src == m->w.wv || g_tiled[i].src == m->w.wo ||
                g_tiled[i].src == m->w.w1 || g_tiled[i].src == m->w.w3 || g_tiled[i].src == m->w.w2 || g_tiled[i].src == m->w.wcls)
                if (g_tiled[i].tiled >= lo) g_tiled.erase(g_tiled.begin() + (long)i);
    }
    rama_free(ctx, m->tiled);
    m->tiled = nullptr;
}

// the live model that owns these weights (matched on its first layer tensor, or the classifier of a layerless stage)
rama_model* model_of(const rama_weights* w) {
    for (rama_model* m : g_models) {
        // (an ADOPTED model is the caller's tensors: another rama_weights that shares wq but not the rest is another model)
        if (m->adopted && (m->w.wk != w->wk || m->w.wv != w->wv || m->w.wo != w->wo || m->w.w1 != w->w1 || m->w.w2 != w->w2 || m->w.w3 != w->w3)) continue;
        if ((w->wq && m->w.wq == w->wq) || (!w->wq && w->wcls && m->w.wcls == w->wcls && !m->w.wq)) return m;
    }
    return nullptr;
}

// the matrices of a model as (tensor, floats): what an adopted model must keep alive, what a freed allocation may take away
std::vector<std::pair<const float*, size_t>> model_tensors(const rama_model* m) {
    const size_t nl = (size_t)(m->stage.layer_end - m->stage.layer_begin), d = (size_t)m->cfg.dim, h = (size_t)m->cfg.hidden_dim, V = (size_t)m->cfg.vocab_size;
    std::vector<std::pair<const float*, size_t>> t;
    if (nl) { t = {{m->w.wq, nl * d * d}, {m->w.wk, nl * d * d}, {m->w.wv, nl * d * d}, {m->w.wo, nl * d * d}, {m->w.w1, nl * h * d}, {m->w.w2, nl * d * h}, {m->w.w3, nl * h * d}}; }
    if (m->stage.do_cls && m->w.wcls) t.push_back({m->w.wcls, V * d});
    return t;
}
// does [p, p + n) touch one of the model's matrices?
bool model_holds(const rama_model* m, const float* p, size_t n) {
    for (auto& t : model_tensors(m)) if (t.first && p < t.first + t.second && t.first < p + n) return true;
    return false;
}

}  // namespace

extern "C" unsigned long long rama_internal_copies_generation() { return g_copies_gen.load(); }

// internal: the tile-order copy of a row-major weight tensor (by its base address) a model registered, or NULL
extern "C" const float* rama_internal_tiled_lookup(const float* src) {
    std::lock_guard<std::mutex> lk(g_tiled_mu);
    for (auto& e : g_tiled) if (e.src == src) return e.tiled;
    return nullptr;
}

// internal (not in the C ABI header): the interleaved copy of the (w1, w3) pair a model registered, or NULL
extern "C" const float* rama_internal_w13_lookup(const float* w1, const float* w3) {
    std::lock_guard<std::mutex> lk(g_w13_mu);
    for (auto& e : g_w13) if (e.w1 == w1 && e.w3 == w3) return e.w13i;
    return nullptr;
}

// internal: the chain-order copy of layer-aligned `a` = (a row-major [rows, K] matrix inside a registered tensor), or NULL.
// The (w1, w3) pair is looked up by w1 with rows = 2 * hidden.
extern "C" const float* rama_internal_chain_lookup(const float* a, int rows, int K) {
    std::lock_guard<std::mutex> lk(g_chain_mu);
    for (auto& e : g_chain) {
        if (e.rows != rows || e.K != K || a < e.src) continue;
        const size_t off = (size_t)(a - e.src);
        if (off >= e.nmat * e.per_src || off % e.per_src) continue;
        return e.chain + (off / e.per_src) * ((size_t)((rows + 15) / 16) * 16) * (size_t)K;
    }
    return nullptr;
}

// internal: make the lazily built copies of the model these weights belong to (what: 1 = chain order, 2 = tile order).
// Never called inside a stream capture (it allocates and synchronises).  No model, or no room: nothing happens and
// the callers' row-major kernels run.
extern "C" int rama_internal_model_ensure(rama_ctx* ctx, const rama_weights* w, int what) {
    if (!ctx || !w) return 0;
    rama_model* m;
    {
        std::lock_guard<std::mutex> lk(g_models_mu);
        m = model_of(w);
        if (!m) return 0;
    }
    std::lock_guard<std::mutex> bl(m->build_mu);
    bool& tried = what == 1 ? m->chain_tried : m->tiled_tried;
    if (tried) return 0;
    const int rc = what == 1 ? make_chain(ctx, m) : make_tiled(ctx, m);
    tried = true;       // set once the copy exists (or cannot): nobody sees "tried" and then misses the registry entry
    return rc;
}
// the same for the model whose weight blob contains `p` (the 1:1 trait ops see views, not a rama_weights)
extern "C" int rama_internal_model_ensure_ptr(rama_ctx* ctx, const float* p, int what) {
    if (!ctx || !p) return 0;
    rama_weights w{};
    {
        std::lock_guard<std::mutex> lk(g_models_mu);
        const rama_model* hit = nullptr;
        for (rama_model* m : g_models) if (m->adopted ? model_holds(m, p, 1) : (p >= m->blob && p < m->blob + m->blob_floats)) { hit = m; break; }
        if (!hit) return 0;
        w = hit->w;
    }
    return rama_internal_model_ensure(ctx, &w, what);
}

// ---------------------------------------------------------------- weights that were uploaded tensor by tensor (hbm.rs:55-90)
// internal (rama_api.hip's rama_alloc_f32 / rama_free): the allocation table
extern "C" void rama_internal_note_alloc(const float* base, size_t n) {
    std::lock_guard<std::mutex> lk(g_allocs_mu);
    g_allocs[base] = n;
}
extern "C" void rama_internal_drop_graphs(rama_ctx* ctx);      // rama_api.hip: captured graphs hold the copies' addresses
// [r6] the union interval of every CALLER-OWNED range a copy was derived from (an adopted model's tensors, a view's tensor); it only grows.  A library entry
// that writes device memory asks rama_internal_note_write first: outside the interval (the run state of a host whose weights lie elsewhere; every write while
// nothing was derived) that is two atomic loads, inside it the registries are searched like rama_copy_h2d_f32 does.
std::atomic<uintptr_t> g_derived_lo{UINTPTR_MAX}, g_derived_hi{0};
static void note_derived(const float* p, size_t n) {
    const uintptr_t a = (uintptr_t)p, b = a + n * sizeof(float);
    uintptr_t lo = g_derived_lo.load(std::memory_order_relaxed);
    while (a < lo && !g_derived_lo.compare_exchange_weak(lo, a, std::memory_order_relaxed)) { }
    uintptr_t hi = g_derived_hi.load(std::memory_order_relaxed);
    while (b > hi && !g_derived_hi.compare_exchange_weak(hi, b, std::memory_order_relaxed)) { }
}
// internal: [base, base + n) is about to be freed or overwritten -- every derived copy made from it goes (an adopted model that holds a tensor
// in the range is dissolved, a view's chain-order copy is freed); `freed`: the allocation itself leaves the table.  Cheap when nothing was derived.
extern "C" int rama_internal_forget_range(rama_ctx* ctx, const float* base, size_t n, int freed) {
    if (freed) {      // (the caller knows the base only: the table knows how far the allocation reaches)
        std::lock_guard<std::mutex> lk(g_allocs_mu);
        auto it = g_allocs.find(base);
        if (it != g_allocs.end()) { n = it->second; g_allocs.erase(it); }
    }
    std::vector<rama_model*> gone;
    {
        std::lock_guard<std::mutex> lk(g_models_mu);
        for (size_t i = g_models.size(); i-- > 0;)
            if (g_models[i]->adopted && model_holds(g_models[i], base, n)) { gone.push_back(g_models[i]); g_models.erase(g_models.begin() + (long)i); }
    }
    std::vector<float*> views;
    {
        std::lock_guard<std::mutex> lk(g_chain_mu);
        for (size_t i = g_chain.size(); i-- > 0;) {
            const ChainEntry& e = g_chain[i];
            if (e.own && e.src < base + n && base < e.src + e.nmat * e.per_src) { views.push_back(e.own); g_chain.erase(g_chain.begin() + (long)i); }
        }
    }
    if (gone.empty() && views.empty()) return 0;
    g_copies_gen++;
    int rc = ctx ? rama_sync(ctx) : 0;
    if (ctx) rama_internal_drop_graphs(ctx);
    for (rama_model* m : gone) {
        { std::lock_guard<std::mutex> bl(m->build_mu); }      // whoever found it before it left the list and is making a copy right now has finished (as rama_model_free)
        drop_chain(ctx, m); drop_tiled(ctx, m); delete m;
    }
    for (float* v : views) { std::lock_guard<std::mutex> lk(g_allocs_mu); g_allocs.erase(v); }
    for (float* v : views) if (hipFree(v) != hipSuccess && !rc) rc = RAMA_EIO;
    return rc;
}

extern "C" int rama_internal_note_write(rama_ctx* ctx, const float* dst, size_t n) {
    const uintptr_t a = (uintptr_t)dst, b = a + n * sizeof(float);
    if (b <= g_derived_lo.load(std::memory_order_relaxed) || a >= g_derived_hi.load(std::memory_order_relaxed)) return 0;
    return rama_internal_forget_range(ctx, dst, n, 0);
}

// internal: weights that belong to no model (uploaded tensor by tensor and passed to rama_forward* as a rama_weights) are ADOPTED -- a model
// record that owns nothing but the derived copies it will make (the chain-order copy of parity mode, the tile-order copy of the token-batch
// passes), so that the reference's own upload path (hbm.rs:55-90, integration/rust/hbm_hip.rs) runs the same kernels as a resident model.
// Every matrix must lie inside a live rama_alloc_f32 / rama_upload_f32 allocation that is large enough; freeing or overwriting (rama_free,
// rama_copy_h2d_f32) any of them dissolves the record.  Nothing happens when the weights already belong to a model.
extern "C" int rama_internal_adopt(rama_ctx* ctx, const rama_config* cfg, const rama_stage* st, const rama_weights* w) {
    if (!ctx || !cfg || !w || !st) return 0;
    std::lock_guard<std::mutex> lk(g_models_mu);
    if (model_of(w)) return 0;
    rama_model* m = new rama_model();
    m->cfg = *cfg; m->w = *w; m->stage = *st; m->adopted = true;
    if (cfg->shared_weight && st->do_cls && !m->w.wcls) m->w.wcls = m->w.token_embedding_table;
    for (auto& t : model_tensors(m)) {
        const float* base; size_t n;
        if (!t.first || !alloc_of(t.first, &base, &n) || t.first + t.second > base + n) { delete m; return 0; }
    }
    for (auto& t : model_tensors(m)) note_derived(t.first, t.second);
    g_models.push_back(m);
    return 0;
}

// internal: the chain-order copy of the row-major [rows, K] matrix at `a` -- a model's (resident or adopted), or one made for this VIEW on first
// use: the whole tensor when `a` sits a whole number of matrices into an allocation that holds whole matrices (wq of hbm.rs: every layer at
// once), else the one matrix.  NULL: no room (16 GiB are kept free), `a` is in no known allocation, or a stream capture is running and the
// copy does not exist yet.
extern "C" const float* rama_internal_chain_view(rama_ctx* ctx, const float* a, int rows, int K, int capturing) {
    if (const float* hit = rama_internal_chain_lookup(a, rows, K)) return hit;
    if (capturing || getenv("RAMA_NO_CHAIN") || K % 16) return nullptr;
    static std::mutex build_mu;      // one copy per tensor: a second caller waits for the first one's copy instead of making its own
    std::lock_guard<std::mutex> bl(build_mu);
    if (const float* hit = rama_internal_chain_lookup(a, rows, K)) return hit;
    const float* base; size_t n;
    if (!alloc_of(a, &base, &n)) return nullptr;
    const size_t per = (size_t)rows * K, off = (size_t)(a - base);
    const float* src = a; size_t nmat = 1;
    if (off % per == 0 && n % per == 0) { src = base; nmat = n / per; }
    else if (off + per > n) return nullptr;
    const size_t per_dst = (size_t)((rows + 15) / 16) * 16 * K, total = nmat * per_dst;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < total * 4 + ((size_t)16 << 30)) return nullptr;
    float* dst = nullptr;
    if (rama_alloc_f32(ctx, total, &dst)) return nullptr;
    rama_sync(ctx);
    hipLaunchKernelGGL(chain_weights_kernel<false>, dim3(4096), dim3(256), 0, 0, dst, src, (const float*)nullptr, nmat, rows, K);
    if (hipDeviceSynchronize() != hipSuccess) { rama_free(ctx, dst); return nullptr; }
    note_derived(src, nmat * per);
    { std::lock_guard<std::mutex> lk(g_chain_mu); g_chain.push_back({src, nullptr, per, nmat, rows, K, dst, dst}); }
    return rama_internal_chain_lookup(a, rows, K);
}

extern "C" int rama_model_load_stage(rama_ctx* ctx, const char* path, const rama_stage* stage, rama_model** out) {
    if (!ctx || !path || !out) return bad(RAMA_EINVAL, "rama_model_load: NULL argument");
    int fd = open(path, O_RDONLY);
    if (fd < 0) return bad(RAMA_EIO, "cannot open checkpoint");
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 28) { close(fd); return bad(RAMA_EIO, "checkpoint too small"); }
    void* map = mmap(nullptr, sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return bad(RAMA_EIO, "mmap failed");
    int32_t hdr[7];
    memcpy(hdr, map, 28);
    rama_config c{};
    // mod.rs:141-166: six fields as-is; the sign of vocab_size carries the shared flag
    c.dim = hdr[0]; c.hidden_dim = hdr[1]; c.n_layers = hdr[2]; c.n_heads = hdr[3]; c.n_kv_heads = hdr[4];
    c.vocab_size = hdr[5] > 0 ? hdr[5] : -hdr[5];
    c.shared_weight = hdr[5] > 0;
    c.seq_len = hdr[6];
    if (hdr[0] == 0x616b3432) {   // "ak42": llama2.c v1/v2 header (export.py:132-260), which the engine cannot read
        munmap(map, sb.st_size);
        return bad(RAMA_EUNSUP, "checkpoint is llama2.c v1/v2 (ak42 header); rama reads the v0 legacy format only");
    }
    if (c.dim <= 0 || c.hidden_dim <= 0 || c.n_layers <= 0 || c.n_heads <= 0 || c.vocab_size <= 0 || c.seq_len <= 0 ||
        c.dim % c.n_heads != 0) {
        munmap(map, sb.st_size);
        return bad(RAMA_EIO, "implausible v0 header");
    }
    auto layout = v0_layout(c);
    size_t total = 0;
    for (auto& t : layout) total += t.n;
    if ((size_t)sb.st_size != 28 + total * 4) {
        munmap(map, sb.st_size);
        return bad(RAMA_EIO, "checkpoint size does not match its header (ram.rs:28-51 layout)");
    }
    const rama_stage st = stage ? *stage : rama_stage{0, c.n_layers, 1, 1};
    if (st.layer_begin < 0 || st.layer_begin > st.layer_end || st.layer_end > c.n_layers) {
        munmap(map, sb.st_size);
        return bad(RAMA_EINVAL, "rama_model_load_stage: bad layer range");
    }
    const bool whole = st.layer_begin == 0 && st.layer_end == c.n_layers && st.do_embed && st.do_cls;
    const size_t nl = (size_t)(st.layer_end - st.layer_begin);
    const bool need_emb = st.do_embed || (st.do_cls && c.shared_weight);
    // which floats of each tensor this stage owns (a whole model: the file's blob as it is, one copy)
    struct Piece { const char* name; size_t n, file_off; };
    std::vector<Piece> pieces;
    size_t need = 0, foff = 0;
    for (auto& t : layout) {
        size_t n = t.n, src = 0;
        if (!whole) {
            if (t.per_layer) { n = nl * t.per_layer; src = (size_t)st.layer_begin * t.per_layer; }
            else if (!strcmp(t.name, "token_embedding_table")) n = need_emb ? t.n : 0;
            else if (!strcmp(t.name, "rms_final_weight") || !strcmp(t.name, "wcls")) n = st.do_cls ? t.n : 0;
        }
        if (n) { pieces.push_back({t.name, n, foff + src}); need += whole ? n : align64(n); }
        foff += t.n;
    }
    rama_model* m = new rama_model();
    m->cfg = c;
    m->stage = st;
    m->blob_floats = need;
    int rc = rama_alloc_f32(ctx, need, &m->blob);
    if (rc) { munmap(map, sb.st_size); delete m; return rc; }
    // 28-byte header => the tensor blob starts 12 bytes off a 16-byte boundary in the file,
    // but lands 256-byte aligned in HBM; every tensor size is a multiple of 4 floats.
    const float* file = (const float*)((const char*)map + 28);
    size_t off = 0;
    if (whole) rc = rama_copy_h2d_f32(ctx, m->blob, file, total);
    for (auto& pc : pieces) {
        *field(m->w, pc.name) = m->blob + off;
        if (!whole && !rc) rc = rama_copy_h2d_f32(ctx, m->blob + off, file + pc.file_off, pc.n);
        off += whole ? pc.n : align64(pc.n);
    }
    munmap(map, sb.st_size);
    if (rc) { rama_free(ctx, m->blob); delete m; return rc; }
    if (c.shared_weight && st.do_cls) m->w.wcls = m->w.token_embedding_table;   // state.rs:111-117
    if (!st.do_embed && !(st.do_cls && c.shared_weight)) m->w.token_embedding_table = nullptr;
    rc = make_w13i(ctx, m);
    if (rc) { rama_free(ctx, m->blob); delete m; return rc; }
    if (getenv("RAMA_EAGER_COPIES")) {      // the derived copies are otherwise made on first use (rama_internal_model_ensure)
        m->tiled_tried = m->chain_tried = true;
        rc = make_tiled(ctx, m);
        if (!rc) rc = make_chain(ctx, m);
        if (rc) { rama_model_free(ctx, m); return rc; }
    }
    { std::lock_guard<std::mutex> lk(g_models_mu); g_models.push_back(m); }
    *out = m;
    return 0;
}

extern "C" int rama_model_load(rama_ctx* ctx, const char* path, rama_model** out) {
    return rama_model_load_stage(ctx, path, nullptr, out);
}

extern "C" int rama_model_synth(rama_ctx* ctx, const rama_config* cfg, uint64_t seed, const rama_stage* stage,
                                const float* rope_real_host, const float* rope_imag_host, rama_model** out) {
    if (!ctx || !cfg || !out) return bad(RAMA_EINVAL, "rama_model_synth: NULL argument");
    rama_stage st = stage ? *stage : rama_stage{0, cfg->n_layers, 1, 1};
    if (st.layer_begin < 0 || st.layer_begin > st.layer_end || st.layer_end > cfg->n_layers)
        return bad(RAMA_EINVAL, "rama_model_synth: bad layer range");
    if (cfg->dim <= 0 || cfg->n_heads <= 0 || cfg->dim % cfg->n_heads) return bad(RAMA_EINVAL, "rama_model_synth: bad config");
    const size_t nl = (size_t)(st.layer_end - st.layer_begin);
    auto layout = v0_layout(*cfg);
    const double ih4_std = std::sqrt(4.0 * (65536.0 * 65536.0 - 1.0) / 12.0);
    const bool need_emb = st.do_embed || (st.do_cls && cfg->shared_weight);

    // which tensors this stage owns, and how many floats of each
    struct Piece { TensorSpec t; size_t n; size_t src_off; };
    std::vector<Piece> pieces;
    size_t total = 0;
    for (auto& t : layout) {
        size_t n = 0, src = 0;
        if (t.per_layer) { n = nl * t.per_layer; src = (size_t)st.layer_begin * t.per_layer; }
        else if (!strcmp(t.name, "token_embedding_table")) n = need_emb ? t.n : 0;
        else if (!strcmp(t.name, "rms_final_weight") || !strcmp(t.name, "wcls")) n = st.do_cls ? t.n : 0;
        else n = t.n;   // RoPE tables: replicated, they are KBs
        if (n) { pieces.push_back({t, n, src}); total += align64(n); }
    }
    rama_model* m = new rama_model();
    m->cfg = *cfg; m->stage = st; m->blob_floats = total;
    int rc = rama_alloc_f32(ctx, total, &m->blob);
    if (rc) { delete m; return rc; }
    size_t off = 0;
    const size_t hs = cfg->dim / cfg->n_heads;
    for (auto& p : pieces) {
        float* dst = m->blob + off;
        *field(m->w, p.t.name) = dst;
        if (p.t.tag >= 0) {
            rc = rama_fill_synth(ctx, dst, p.n, seed, (uint64_t)p.t.tag, (uint64_t)p.src_off,
                                 (float)(p.t.std / ih4_std), p.t.bias);
        } else {
            const bool real = !strcmp(p.t.name, "freq_cis_real");
            const float* given = real ? rope_real_host : rope_imag_host;
            std::vector<float> tab;
            if (!given) {   // model.py:41-47: cos/sin(t * 10000^(-2i/hs))
                tab.resize(p.n);
                for (size_t t = 0; t < (size_t)cfg->seq_len; t++)
                    for (size_t i = 0; i < hs / 2; i++) {
                        double f = 1.0 / std::pow(10000.0, (double)(2 * i) / (double)hs);
                        tab[t * (hs / 2) + i] = (float)(real ? std::cos((double)t * f) : std::sin((double)t * f));
                    }
                given = tab.data();
            }
            rc = rama_copy_h2d_f32(ctx, dst, given, p.n);
        }
        if (rc) { rama_free(ctx, m->blob); delete m; return rc; }
        off += align64(p.n);
    }
    if (cfg->shared_weight && st.do_cls) m->w.wcls = m->w.token_embedding_table;
    if (!st.do_embed && !(st.do_cls && cfg->shared_weight)) m->w.token_embedding_table = nullptr;
    rc = make_w13i(ctx, m);
    if (rc) { rama_free(ctx, m->blob); delete m; return rc; }
    if (getenv("RAMA_EAGER_COPIES")) {      // the derived copies are otherwise made on first use (rama_internal_model_ensure)
        m->tiled_tried = m->chain_tried = true;
        rc = make_tiled(ctx, m);
        if (!rc) rc = make_chain(ctx, m);
        if (rc) { rama_model_free(ctx, m); return rc; }
    }
    { std::lock_guard<std::mutex> lk(g_models_mu); g_models.push_back(m); }
    *out = m;
    return 0;
}

// Write the model as a llama2.c v0 file (export.py:75-127 legacy_export = the layout ram.rs:28-51 reads):
// 7 x i32 header with vocab_size negated when the classifier is not shared, then the tensors in
// v0 order.  Only a whole model (every layer, embedding and classifier on this device) can be saved.
extern "C" int rama_model_save(rama_ctx* ctx, const rama_model* m, const char* path) {
    if (!ctx || !m || !path) return bad(RAMA_EINVAL, "rama_model_save: NULL argument");
    const rama_config& c = m->cfg;
    if (m->stage.layer_begin != 0 || m->stage.layer_end != c.n_layers || !m->stage.do_embed || !m->stage.do_cls)
        return bad(RAMA_EINVAL, "rama_model_save: the model holds only a pipeline stage");
    FILE* f = fopen(path, "wb");
    if (!f) return bad(RAMA_EIO, "rama_model_save: cannot open the output file");
    const int32_t hdr[7] = {c.dim, c.hidden_dim, c.n_layers, c.n_heads, c.n_kv_heads,
                            c.shared_weight ? c.vocab_size : -c.vocab_size, c.seq_len};
    bool ok = fwrite(hdr, sizeof(hdr), 1, f) == 1;
    rama_weights w = m->w;
    std::vector<float> buf;
    const size_t chunk = (size_t)16 << 20;     // floats per D2H piece
    for (auto& t : v0_layout(c)) {
        const float* src = *field(w, t.name);
        if (!src) { ok = false; break; }
        for (size_t off = 0; ok && off < t.n; off += chunk) {
            const size_t n = std::min(chunk, t.n - off);
            buf.resize(n);
            if (rama_download_f32(ctx, src + off, n, buf.data()) != 0) { ok = false; break; }
            ok = fwrite(buf.data(), sizeof(float), n, f) == n;
        }
        if (!ok) break;
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { remove(path); return bad(RAMA_EIO, "rama_model_save: write failed"); }
    return 0;
}

extern "C" int rama_model_config(const rama_model* m, rama_config* cfg) {
    if (!m || !cfg) return RAMA_EINVAL;
    *cfg = m->cfg;
    return 0;
}
extern "C" int rama_model_weights(const rama_model* m, rama_weights* w) {
    if (!m || !w) return RAMA_EINVAL;
    *w = m->w;
    return 0;
}
extern "C" size_t rama_model_bytes(const rama_model* m) { return m ? m->blob_floats * sizeof(float) : 0; }

extern "C" void rama_internal_drop_graphs(rama_ctx* ctx);      // rama_api.hip: captured graphs hold the copies' addresses

// Give the derived weight copies back: mask bit 0 (1) = the chain-order copy (parity mode), bit 1 (2) = the tile-order copy (token-batch
// passes).  A server that only decodes in fast mode holds 38 GB for llama2-7B instead of up to 92.  They are made again on the
// next call that wants them.
extern "C" int rama_model_release_copies(rama_ctx* ctx, rama_model* m, int mask) {
    if (!ctx || !m) return bad(RAMA_EINVAL, "rama_model_release_copies: NULL argument");
    if (mask & ~3) return bad(RAMA_EINVAL, "rama_model_release_copies: mask must be a combination of 1 (chain order) and 2 (tile order)");
    std::lock_guard<std::mutex> bl(m->build_mu);
    int rc = rama_sync(ctx);
    if (rc) return rc;
    rama_internal_drop_graphs(ctx);
    if (mask & 1) { drop_chain(ctx, m); m->chain_tried = false; }
    if (mask & 2) { drop_tiled(ctx, m); m->tiled_tried = false; }
    return 0;
}
extern "C" int rama_model_free(rama_ctx* ctx, rama_model* m) {
    if (!m) return 0;
    { std::lock_guard<std::mutex> lk(g_models_mu); g_models.erase(std::remove(g_models.begin(), g_models.end(), m), g_models.end()); }      // nobody finds it any more
    { std::lock_guard<std::mutex> bl(m->build_mu); }      // ... and whoever found it before and is making a copy right now has finished
    if (ctx) { rama_sync(ctx); rama_internal_drop_graphs(ctx); }
    drop_chain(ctx, m);
    if (m->w13i) {
        {
            std::lock_guard<std::mutex> lk(g_w13_mu);
            for (size_t i = 0; i < g_w13.size(); i++) if (g_w13[i].w13i == m->w13i) { g_w13.erase(g_w13.begin() + i); break; }
        }
        rama_free(ctx, m->w13i);
    }
    drop_tiled(ctx, m);
    int rc = m->blob ? rama_free(ctx, m->blob) : 0;
    delete m;
    return rc;
}
