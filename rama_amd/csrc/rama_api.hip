// rama_api.hip -- the C ABI of include/rama_hip.h: context, memory, the 1:1 Device<T>
// ops, the fused decode path, measurement.  Kernels are in kernels.hpp.
#include "../../include/rama_hip.h"
#include "kernels.hpp"
#include "attn_wo.hpp"
#include "layer_fused.hpp"
#include "topp_sort.hpp"
#include "prefill_attn.hpp"
#include "prefill_mfma.hpp"
#include "ref_order.hpp"
#include "chain.hpp"
#include "topp_pick.hpp"

#include <hip/hip_ext.h>   // hipExtLaunchKernelGGL: start/stop events carried by the dispatch itself

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace rama;

extern "C" const float* rama_internal_w13_lookup(const float* w1, const float* w3);   // model.hip
extern "C" const float* rama_internal_tiled_lookup(const float* src);                    // model.hip
extern "C" const float* rama_internal_chain_lookup(const float* a, int rows, int K);     // model.hip
extern "C" int rama_internal_model_ensure(rama_ctx* ctx, const rama_weights* w, int what);       // model.hip: 1 = chain order, 2 = tile order
extern "C" int rama_internal_model_ensure_ptr(rama_ctx* ctx, const float* p, int what);
extern "C" unsigned long long rama_internal_copies_generation();                                           // model.hip: advances whenever a derived weight copy is freed
extern "C" void rama_internal_note_alloc(const float* base, size_t n);                                   // model.hip: the allocation table
extern "C" int rama_internal_forget_range(rama_ctx* ctx, const float* base, size_t n, int freed);      // ... the copies derived from a range go
extern "C" int rama_internal_adopt(rama_ctx* ctx, const rama_config* cfg, const rama_stage* st, const rama_weights* w);
extern "C" const float* rama_internal_chain_view(rama_ctx* ctx, const float* a, int rows, int K, int capturing);
extern "C" int rama_internal_note_write(rama_ctx* ctx, const float* dst, size_t n);                    // model.hip: an entry is about to write [dst, dst + n) on the device
// every entry that writes device memory the caller names says so first: a chain-order copy DERIVED from a tensor uploaded by the caller (an adopted model's, a
// view's) must not outlive a device-side write into that tensor (rama_fill_synth re-seeding it, an op's output landing in it).  Two atomic loads when the
// range lies outside everything copies were derived from.
#define RAMA_WRITES(c, p, n) do { if ((p) && (n)) { const int rw_ = rama_internal_note_write((c), (p), (size_t)(n)); if (rw_) return rw_; } } while (0)

// ---------------------------------------------------------------- error plumbing

static thread_local std::string g_err;

const char* rama_last_error(void) { return g_err.c_str(); }

static int fail(int code, const char* what, const char* file, int line) {
    char buf[512];
    if (code > 0)
        snprintf(buf, sizeof buf, "%s: %s (hipError %d) at %s:%d", what, hipGetErrorString((hipError_t)code), code, file, line);
    else
        snprintf(buf, sizeof buf, "%s (rama error %d) at %s:%d", what, code, file, line);
    g_err = buf;
    return code;
}
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail((int)e_, #expr, __FILE__, __LINE__); } while (0)
#define REQUIRE(cond, code, msg) do { if (!(cond)) return fail((code), msg, __FILE__, __LINE__); } while (0)
#define LAUNCHCHK() HIPCHK(hipGetLastError())

// ---------------------------------------------------------------- context

constexpr int kSmallAttnPosDefault = 256;
constexpr size_t kAttnChainMaxLds = 136 * 1024;      // dynamic LDS attention_chain_kernel may ask for (allowed once per device in rama_ctx_create)
constexpr int kSpreadAttnPos = 128;        // parity mode: from this position on the attention is two launches spread over the chip (chain.hpp; "spread_pos": 187 against 184 tok/s at positions 124..179, 178 against 152 at 800)
constexpr int kLeadSlots = 2 * 256 + 2;       // tagged words of the leader-workgroup norms: two per layer of a stage (<= 256 layers), one for the final norm
constexpr int kLongAttnPos = 256;          // parity mode: attention_chain_kernel runs 16 waves per head from this position on

struct KProf {
    int kernel_id = -1;
    int max_records = 0;
    int used = 0;
    std::vector<hipEvent_t> ev;   // 2 per record
};

struct GraphCache {
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph = nullptr;
    // identity of what was captured
    rama_config cfg{};
    rama_weights w{};
    rama_run_state s{};
    bool valid = false;
    int steps = 1;                     // decode steps in the captured graph
    unsigned long long copies_gen = 0; // model.hip's generation of derived weight copies at capture: a graph holds their addresses, and ANOTHER context may free them
    bool handoff = false;              // [r6] the captured launches hand data over inside a kernel (a leader norm, attention+Wo, the one-launch stage): a REPLAY must mark
                                       // the error word as worth reading too (rama_ctx::handoff_dirty is otherwise only set where such a launch is enqueued)
};

struct rama_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    Ctl* ctl = nullptr;          // device cursor
    int* forced = nullptr;       // device forced-token list
    int forced_cap = 0;
    int* out = nullptr;          // device produced-token list
    int out_cap = 0;
    int* ring = nullptr;                    // host-pinned, device-mapped: ring[i] = token i of the chained loop + 1 (0: not produced yet)
    int* ring_dev = nullptr;
    int ring_hi = 0;                        // entries that may be non-zero
    int* argmax_result = nullptr;   // device int for rama_sample_argmax
    int* pinned_int = nullptr;      // host pinned
    int* pinned_tok = nullptr;      // host pinned staging: token ids + a SeqSlot table of a token-batch pass
    bool graph_mode = false;
    struct StageGraph { GraphCache g; rama_stage st{}; int variant = 0; unsigned long long used = 0; };
    std::vector<StageGraph> sg;        // rama_forward / rama_forward_stage* in graph mode: one graph per (state, stage, attention variant)
    unsigned long long sg_clock = 0;
    GraphCache gc[8];                  // [0..3] one step per graph, [4..7] tune_graph_steps steps per graph; by attention variant (attn_variant)
    KProf kp;
    int cu_count = 0;
    hipEvent_t cur_start = nullptr, cur_stop = nullptr;   // events the next profiled launch carries
    int tune_geom = 3;
    int tune_w13i = 1;                     // 1: the fused W1|W3 launch streams the model's row-interleaved copy when there is one
    int tune_solo = -1;                    // small-K matvecs, one wave per row group: 1 on, 0 off, -1 = rows of <= 2048 floats
    int tune_ref_order = 0;                // 1: every op in the reference's own rounding order: bit-comparable with the CPU path ("parity mode")
    int tune_tol = 0;                      // "ref_order" = 2, the tolerance-mode experiment: the chain-order matvecs (the reference's rounding sequence) with the layer
                                           // norms folded into them as tree-shaped sums and the fast attention -- 0.72 of the roofline, but 1.4e-4 from the CPU path at
                                           // llama2-7B x 200 positions, no closer than the fast path; kept as the per-op A/B instrument
    int tune_bar = 0;                      // [r6] "ref_order" = 3, BAR mode: parity mode's launches (chain-order matvecs, exact norms, the exact attention) up to position
                                           // "bar_pos", the FAST path's attention from there on -- not bit-identical, but measured <= 1e-4 from the CPU path over the WHOLE
                                           // 2 048-position context at llama2-7B depth (profiles/r06_tolerance_sweep_7b_2048pos.jsonl: fast attention at every position 9.75e-5;
                                           // every other single swap >= 1.3e-4 already at 200 positions), where the exact attention costs 25 us a layer at position 1 900
    int tune_bar_pos = kSpreadAttnPos;     // ... the first position that takes the fast attention (clamped to the spread attention's switch and to 256: below it one exact variant)
    bool bar_fast = false;                 // ... the steps being enqueued / captured are at or behind it
    int tune_lane_reduce = 0;              // [r6] parity mode: the order of the final 4-lane sum of cpu.rs:148 `v.reduce_add()` (wide::f32x4 leaves it to the build's target
                                           // features): 0 pairwise (l0+l1)+(l2+l3), 1 strided (l0+l2)+(l1+l3), 2 sequential ((l0+l1)+l2)+l3 -- the oracle's switch of the same name
    int tune_tol_mask = 0;                 // tolerance mode, A/B: ops swapped for the fast path's (1 qkv, 2 wo, 4 w13, 8 w2, 16 cls) or parity mode's (32 attention, 64 norms)
    int tune_chain = 1;                    // parity mode streams the model's chain-order weight copy (chain.hpp); 0: ref_order.hpp's one-thread-per-row kernels
    int tune_chain_d = 0;                  // chain-order matvec geometry: 0 = by row groups per CU, else 100 W + D (waves per group, blocks per wave in flight)
    // device top-p sampler (Device::sample for temperature != 0); temperature 0 = argmax
    float samp_T = 0.0f, samp_topp = 0.9f, samp_u = 0.0f;
    float* topp_keys[2] = {nullptr, nullptr}; int* topp_vals[2] = {nullptr, nullptr};
    float* topp_prefix = nullptr; int* topp_m = nullptr; unsigned* topp_err = nullptr;
    int topp_cap = 0;
    float* topp_bp = nullptr; int* topp_bi = nullptr; int* topp_bcount = nullptr;       // topp_sort.hpp
    int* topp_racc = nullptr;               // pair-wise ranking: the accumulators, one per block slot
    unsigned long long* topp_rk = nullptr;  // small-block path: count << 48 | mass accumulators, one per block slot
    unsigned long long* topp_bm = nullptr;  // ... the blocks' running masses
    float* topp_approx = nullptr;           // ... the mass in front of every entry of the whole order
    void* topp_dist = nullptr;              // topp_pick_dist_kernel's hand-off words: items | hdr | cross | epoch | bad
    bool topp_dist_dirty = false;           // ... a distributed pick has been enqueued since its error word was last read (else a synchronising exit need not read it)
    int tune_spread_pos = kSpreadAttnPos;   // parity mode: from this position on the attention is spread over the chip (scores | softmax + values)
    int tune_attn_fv = 1;                   // parity mode, long contexts: softmax + value chains as one launch (0: two launches)
    int tune_topp_dist = 1;                 // 1: the running sums by up to 32 workgroups in one launch (topp_pick.hpp); 0: one workgroup's scan rounds
    ToppStats* topp_stats = nullptr;        // small-block path: partial softmax statistics, one per 1024 logits
    int tune_topp_block = 1024;             // entries per sorted block on the pair-ranking path: 1024 or 512 (statistics once + 8- / 4-wave sorts) or 2048 (round 3's block sort)
    int tune_topp_pairs = 1;                // 1: the ranking as (block, block) pairs spread over the chip + a scatter launch; 0: one workgroup searches all blocks in its LDS
    int tune_norm_in_gemm = 1;              // token-batch passes: the rmsnorm's per-token scale is applied by the consuming GEMM (one launch per norm instead of two)
    int tune_tiled = 1;                     // token-batch GEMMs read the model's tile-order weight copy when it exists
    int tune_prefill_tok = kMfMaxTok;       // prompt positions per weight pass of rama_prefill: 128 (needs the tile-order copies) or 64
    int tune_prefill_attn = 1;              // 1: prefill passes run attention as MFMA tiles, 16 queries per workgroup (prefill_attn.hpp)
    int tune_graph_steps = -1;              // decode steps captured per hipGraph (the cursor lives on the device, so steps are identical); -1: 4 for dim <= 1024, else 1
    int tune_attn_u = 8;                    // cache rows per lane and round in the split-T attention (8 | 16; 16 measured no faster)
    int tune_topp_sort = 1;                 // 0: ranks through global memory (topp_rank_global_kernel) for every vocabulary size
    int tune_topp_keep_sums = 0;            // 1: the scan sampler also writes its running sums to global memory (tests)
    int tune_split_pos = -1;               // attention runs split-T (+ combine launch) from this position on; -1 = by model size
    int tune_resid_r2 = 2;                 // Wo / W2 under geometry 3: 0 = 4-row workgroups, 1 = 2 rows x 8 waves (+0.45 %),
                                           // 2 = additionally 16 waves for rows wider than 8192 floats (W2: +1.15 % more), 3 = 16 waves x 4 chunks
    int tune_prefill = 1;                  // 1: rama_generate_greedy runs the forced prompt positions through rama_prefill
    int tune_merge = -1;                   // attention + Wo in one launch: 1 on, 0 off, -1 by model size (on for dim <= 1024:
                                           // +4..8 % at the stories shapes; at llama2-7B +0.9 % short / -2.5 % long contexts)
    int tune_fused = -1;                   // a stage's layers (+ classifier) as one launch (layer_fused.hpp): 1 on, 0 off, -1 on for dim <= 1024
                                           // (stories15M +4 %, stories110M +25 % tokens/s over the separate launches)
    int tune_fused_solo = -1;              // its workgroups alone on their CU (LDS request padded): 1, 0, -1 = for dim > 512 (stories110M: 216 -> 200 us
                                           // per token, a consumer's polls do not queue behind a neighbour's weight requests; stories15M: 89 -> 92)
    tagged_t* fused_hand = nullptr;        // device: its hand-off vectors (tagged words), room for the largest shape it takes
    unsigned* fused_epoch = nullptr;       // device: the tag of the current token, advanced after every launch of the stage kernel
    bool fused_chained = false;            // the step being enqueued ends in a sampler launch, which advances the epoch
    bool fused_epoch_owed = false;         // ... and the stage launch just enqueued relies on that
    int merge_blocks_per_cu[3] = {-1, -1, -1};   // occupancy of attn_wo_kernel<16|32|64> at the LDS size below
    size_t merge_lds[3] = {0, 0, 0};
    unsigned* attn_counter = nullptr;      // device: arrivals of the attention workgroups
    float* attn_part = nullptr;            // split-T partials [n_heads, nsplit, head_size + 4]
    size_t attn_part_floats = 0;
    float* attn_scores = nullptr;          // parity mode, spread attention: the raw scores [n_heads, seq_len] (the softmax+values launch reads them here and
    size_t attn_scores_floats = 0;         // writes the probabilities to the caller's att: no workgroup reads a buffer another one of the launch writes)
    float* pf_blob = nullptr;              // token-batch scratch (tile layout): see BatchScratch
    float* pc_blob = nullptr;              // parity-mode prefill scratch (row-major token batches): see prefill_chain
    size_t pc_floats = 0;
    int tune_chain_lead = 1;               // parity mode, dim > 512: the layer norms' exact sums by a leader workgroup INSIDE the consuming matvec's launch (chain.hpp CNORM_LEAD)
    unsigned long long* lead_slots = nullptr;   // device: one tagged word per (layer, norm), 256 bytes apart
    int tune_chain_norm = 1;               // parity mode, dim <= 512: the layer norms folded into the matvecs that consume them
    int tune_chain_split = 1;              // parity mode: the row groups that do not divide by the compute units walked as half groups (chain.hpp half_from)
    int tune_chain_lead_w = 0;             // parity mode: waves per row group of the launches with a leader norm (0: by the number of row groups)
    int tune_chain_resid_d = -1;           // parity mode: 100 W + D for the residual products (Wo, W2) only; 0: by the number of row groups like the others; -1: W = 1, D = 32 when a CU holds one group
    // [r5] a run of Device::apply_position calls on consecutive heads (infer.rs:25-29: n_heads calls per layer, 1 024 per llama2-7B token, each a launch of
    // its own) is ISSUED AS ONE LAUNCH: a call only records (q, k, table rows, head size); the next call extends the run when it continues it, and
    // whatever enters the library next issues it first (RAMA_ENTER).  Only on a stream the context owns ("rope_batch" = 0: every call a launch).
    struct { float* q = nullptr; float* k = nullptr; const float* pr = nullptr; const float* pi = nullptr; int hs = 0, count = 0; } rope;
    int tune_rope_batch = 1;
    // ... and so is a run of up to three parity-mode Device::matmul calls with the same activations and shape on chain-order copies (infer.rs:20-23: Wq, Wk,
    // Wv; :41-42: W1, W3): one launch over all their row groups ("matmul_batch")
    struct { const float* w[3]; float* o[3]; const float* x = nullptr; int K = 0, rows = 0, count = 0; bool norm = false; } mm;
    int tune_matmul_batch = 1;
    // ... and a parity-mode Device::rmsnorm waits for the run of matmuls on its output (infer.rs:19-23, :40-42): the run's launch then carries the norm as
    // its leader workgroup (chain.hpp CNORM_LEAD: the exact sum of squares while the row groups' weights are already on their way), and the leader also
    // stores the normalised vector the call was asked for.  Anything else entering the library issues the norm as its own launch first ("norm_fold").
    // The leader's tagged words rotate through a range of their own; the epoch advances when the range wraps.
    struct { float* o = nullptr; const float* x = nullptr; const float* w = nullptr; int n = 0; bool on = false; } nrm;
    int tune_norm_fold = 1;
    int tune_qkv_fold = 1;                 // ... and a run of three matmuls, the apply_position calls over all heads of its first two outputs and the copies of its last two into cache
                                           // rows (infer.rs:20-33) are ONE launch with the Wq|Wk|Wv epilogue (rotation, cache rows)
    int tune_resid_fold = 1;               // ... and a Device::array_add of a recorded matmul's output becomes that launch's residual epilogue
    int op_lead_next = 0;
    // ... and Device::sinu waits for the Device::array_mult on the same vector (infer.rs:44-45), one Device::copy_from_slice for the next (:32-33): one
    // launch per pair ("ew_batch").  At most ONE of the three records is pending at any time: whoever records flushes the others first.
    struct { int kind = 0; float* t = nullptr; const float* s = nullptr; size_t n = 0; } ew;      // 1: sinu(t, n); 2: copy(t, s, n)
    int tune_ew_batch = 1;
    int tune_chain_views = 1;              // parity mode, Device::matmul on a matrix of no model: a chain-order copy of the tensor is made on first use
    int tune_prefill_chain = 1;            // parity mode: prompt positions go through the chain-order token-batch kernels (32 per weight pass); 0: one forward() each
    size_t pf_floats = 0;
    int host_pos = -1;                     // position of the next chained decode step (mirrors the device cursor)
    bool split_attn = false;               // variant the steps being enqueued / captured use
    bool long_attn = false;                // parity mode: the position is >= 256 (16 waves per head in attention_chain_kernel)
    bool spread_attn = false;              // parity mode: the position is >= tune_spread_pos (the exact attention as launches spread over the whole chip)
    int variant = 0;                       // attn_variant() of the steps being enqueued / captured
    bool small_attn = false;               // 4-wave attention workgroups (contexts of <= kSmallAttnPos timesteps)
    int tune_small_waves = 8, tune_small_pos = kSmallAttnPosDefault;   // waves per head and position limit of the small-attention variant
    int tune_combine_v = 1;                // split-T combine: 1 = all slice loads up front, 0 = round 1's loop
    int tune_attn_nsplit = 0;              // split-T slices per head: 0 = #CUs / n_heads (<= 16), else 1..32
    int tune_attn_waves = 8;               // waves per split-T workgroup (16, 8 or 4); 8 measured best at llama2-7B, 1000-1900 tokens
    int tune_attn_nt = 1;                  // 1: split-T attention reads the cache rows non-temporally (+2.7 % tokens/s at 1900 tokens)
    int tune_small_attn = -1;              // fewer-wave attention in the decode step: -1 (default) below tune_small_pos where attention
                                           // is not merged with Wo, 0 never, 1 always (below the split threshold)
    unsigned long long* pbar = nullptr;    // device: [1] = error word of the merged attention+Wo launch's bounded spin
    bool handoff_dirty = false;            // a launch with an in-kernel hand-off (attention+Wo, the one-launch stage) has been enqueued since the error word was last read
    const float* embedded_x = nullptr;   // run-state x that already holds emb[ctl.token] (chained decode)
    // rama_decode_batch_begin / _steps: the sequences' cursors live on the device
    struct BatchChain {
        int n_seq = 0, pos_max = 0, out_cap = 0, steps_done = 0;
        int* toks = nullptr;               // [kMfMaxTok] the token each sequence feeds next
        SeqSlot* seqs = nullptr;           // [kMfMaxTok] cache bases + position of every sequence
        int* out = nullptr;                // [kMfMaxTok, out_cap] the tokens produced
        int* ring = nullptr;               // the same, host-pinned and device-mapped: token + 1, 0 = not produced yet (rama_decode_batch_stream_poll)
        int* ring_dev = nullptr;
        rama_config cfg{}; rama_weights w{};
        hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; int graph_bucket = -1;
    } bc;
};

static int set_device(rama_ctx* c) { HIPCHK(hipSetDevice(c->device)); return 0; }
// the pending run of apply_position calls (rama_ctx::rope) is issued by whatever enters the library next: first statement of every entry point
// that enqueues, synchronises or changes a setting
static int flush_rope(rama_ctx* c);
static int flush_mm(rama_ctx* c);
static int flush_ew(rama_ctx* c);
// (in the order they were recorded: a norm and the run of matmuls on it, the rotations of that run's q and k, a copy -- see rama_copy_from_slice)
static int flush_pending(rama_ctx* c) { int rf = flush_mm(c); if (!rf) rf = flush_rope(c); if (!rf) rf = flush_ew(c); return rf; }      // (flush_mm issues a recorded norm too)
#define RAMA_PENDING(c) ((c)->rope.count | (c)->mm.count | (c)->ew.kind | (int)(c)->nrm.on)
#define RAMA_ENTER(c) do { if ((c) && RAMA_PENDING(c)) { const int rf_ = flush_pending(c); if (rf_) return rf_; } } while (0)

// internal accessors for the library's other translation units (pipe.hip); not in the C ABI header
extern "C" void* rama_internal_stream(rama_ctx* c) { if (c && RAMA_PENDING(c)) (void)flush_pending(c); return c ? (void*)c->stream : nullptr; }
extern "C" int rama_internal_device(rama_ctx* c) { return c ? c->device : 0; }
// the sampler's device scratch after a rama_sample_topp* call (tests compare the running sums with a
// sequential fp32 cumsum): sorted probabilities, sorted indices, running sums, candidate count
extern "C" void rama_internal_topp_scratch(rama_ctx* c, float** keys, int** vals, float** prefix, int** m) {
    if (c && RAMA_PENDING(c)) (void)flush_pending(c);
    if (keys) *keys = c->topp_keys[1];
    if (vals) *vals = c->topp_vals[1];
    if (prefix) *prefix = c->topp_prefix;
    if (m) *m = c->topp_m;
}

// diagnostics (not in the C ABI header): how often the one-pass exact sum (chain.hpp seq_sum_predict) held / fell back
extern "C" int rama_internal_pred_stats(rama_ctx* c, unsigned* held, unsigned* fell_back, int reset) {
    RAMA_ENTER(c);
    unsigned h[2] = {0, 0};
    hipStreamSynchronize(c->stream);
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(rama::g_pred_stats), sizeof h) != hipSuccess) return 1;
    if (held) *held = h[0];
    if (fell_back) *fell_back = h[1];
    if (reset) { const unsigned z[2] = {0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rama::g_pred_stats), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}

// test entry (not in the C ABI header): seqsum_fast.hpp's sum of a_dev[0..n) on nw waves; out_dev[0..3] = sum, held, items, 100 MHz ticks
extern "C" int rama_internal_seqsum_fast(rama_ctx* c, const float* a_dev, int n, int nw, float* out_dev) {
    RAMA_ENTER(c);
    REQUIRE(c && a_dev && out_dev && n > 0, RAMA_EINVAL, "seqsum_fast: bad argument");
    const int per = (n + 64 * nw - 1) / (64 * nw);
#define RAMA_FS(NW_, R_) hipLaunchKernelGGL((seqsum_fast_test_kernel<NW_, R_>), dim3(1), dim3(NW_ * 64), 0, c->stream, a_dev, n, out_dev)
#define RAMA_FS_R(NW_) do { if (per <= 8) RAMA_FS(NW_, 8); else if (per <= 16) RAMA_FS(NW_, 16); else if (per <= 32) RAMA_FS(NW_, 32); else if (per <= 64) RAMA_FS(NW_, 64); \
                            else return fail(RAMA_EINVAL, "seqsum_fast: list too long for this many waves", __FILE__, __LINE__); } while (0)
    if (nw == 1) RAMA_FS_R(1); else if (nw == 2) RAMA_FS_R(2); else if (nw == 4) RAMA_FS_R(4);
    else return fail(RAMA_EINVAL, "seqsum_fast: nw must be 1, 2 or 4", __FILE__, __LINE__);
#undef RAMA_FS_R
#undef RAMA_FS
    LAUNCHCHK();
    return 0;
}

int rama_ctx_create(int device, void* hip_stream, rama_ctx** out) {
    REQUIRE(out, RAMA_EINVAL, "rama_ctx_create: out is NULL");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    REQUIRE(device >= 0 && device < n, RAMA_EINVAL, "rama_ctx_create: no such device");
    HIPCHK(hipSetDevice(device));
    rama_ctx* c = new rama_ctx();
    c->device = device;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    HIPCHK(hipEventCreate(&c->t0));
    HIPCHK(hipEventCreate(&c->t1));
    HIPCHK(hipMalloc(&c->ctl, sizeof(Ctl)));
    HIPCHK(hipMemset(c->ctl, 0, sizeof(Ctl)));
    c->out_cap = 1 << 16;
    HIPCHK(hipMalloc(&c->out, sizeof(int) * c->out_cap));
    HIPCHK(hipHostMalloc(&c->ring, sizeof(int) * c->out_cap, hipHostMallocMapped));
    memset(c->ring, 0, sizeof(int) * c->out_cap);
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->ring_dev), c->ring, 0));
    c->forced_cap = 1 << 16;
    HIPCHK(hipMalloc(&c->forced, sizeof(int) * c->forced_cap));
    HIPCHK(hipMalloc(&c->argmax_result, sizeof(int)));
    HIPCHK(hipMalloc(&c->attn_counter, sizeof(unsigned)));
    HIPCHK(hipMemset(c->attn_counter, 0, sizeof(unsigned)));
    HIPCHK(hipMalloc(&c->fused_hand, (size_t)kFusedMaxLayers * fused_hand_words(kFusedMaxDim, kFusedMaxHidden) * sizeof(tagged_t)));
    HIPCHK(hipMemset(c->fused_hand, 0, (size_t)kFusedMaxLayers * fused_hand_words(kFusedMaxDim, kFusedMaxHidden) * sizeof(tagged_t)));
    HIPCHK(hipMalloc(&c->fused_epoch, sizeof(unsigned)));
    { const unsigned one = 1; HIPCHK(hipMemcpy(c->fused_epoch, &one, sizeof one, hipMemcpyHostToDevice)); }      // the zeroed vectors carry tag 0
    HIPCHK(hipMalloc(&c->lead_slots, 2 * kLeadSlots * 32 * sizeof(unsigned long long)));      // (the second half: the recorded norms of the 1:1 op path)
    HIPCHK(hipMemset(c->lead_slots, 0, 2 * kLeadSlots * 32 * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&c->pbar, 4 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(c->pbar, 0, 4 * sizeof(unsigned long long)));
    HIPCHK(hipHostMalloc(&c->pinned_int, sizeof(int) * 4));
    HIPCHK(hipHostMalloc(&c->pinned_tok, sizeof(int) * kMfMaxTok + sizeof(SeqSlot) * kMfMaxTok));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    c->cu_count = prop.multiProcessorCount;
    // parity mode's attention keeps two product tiles + the scores of a whole context in LDS: more than the 64 KiB a kernel gets by default
#define RAMA_FUSED_ATTR(G_, CD_) \
    HIPCHK(hipFuncSetAttribute((const void*)stage_fused_kernel<G_, CD_, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedSoloLds)); \
    HIPCHK(hipFuncSetAttribute((const void*)stage_fused_kernel<G_, CD_, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedSoloLds));
    RAMA_FUSED_ATTR(16, 2) RAMA_FUSED_ATTR(16, 4) RAMA_FUSED_ATTR(32, 2) RAMA_FUSED_ATTR(32, 4) RAMA_FUSED_ATTR(64, 2) RAMA_FUSED_ATTR(64, 4)
#undef RAMA_FUSED_ATTR
    HIPCHK(hipFuncSetAttribute((const void*)attention_chain_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttnChainMaxLds));
    HIPCHK(hipFuncSetAttribute((const void*)attention_chain_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttnChainMaxLds));
    HIPCHK(hipFuncSetAttribute((const void*)attention_chain_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttnChainMaxLds));
#define RAMA_GC_ATTR(TPW_) \
    HIPCHK(hipFuncSetAttribute((const void*)gemm_chain_kernel<TPW_, CEPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_chain_lds_bytes(TPW_))); \
    HIPCHK(hipFuncSetAttribute((const void*)gemm_chain_kernel<TPW_, CEPI_RESID>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_chain_lds_bytes(TPW_))); \
    HIPCHK(hipFuncSetAttribute((const void*)gemm_chain_kernel<TPW_, CEPI_QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_chain_lds_bytes(TPW_))); \
    HIPCHK(hipFuncSetAttribute((const void*)gemm_chain_kernel<TPW_, CEPI_SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_chain_lds_bytes(TPW_)));
    RAMA_GC_ATTR(1) RAMA_GC_ATTR(2) RAMA_GC_ATTR(4) RAMA_GC_ATTR(8)
#undef RAMA_GC_ATTR
    *out = c;
    return 0;
}

static void drop_graph(rama_ctx* c) {
    if (c->bc.exec) { hipGraphExecDestroy(c->bc.exec); c->bc.exec = nullptr; }
    if (c->bc.graph) { hipGraphDestroy(c->bc.graph); c->bc.graph = nullptr; }
    c->bc.graph_bucket = -1;
    for (auto& g : c->gc) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
        g = GraphCache();
    }
    for (auto& e : c->sg) {
        if (e.g.exec) hipGraphExecDestroy(e.g.exec);
        if (e.g.graph) hipGraphDestroy(e.g.graph);
    }
    c->sg.clear();
}

extern "C" void rama_internal_drop_graphs(rama_ctx* c) { if (c) drop_graph(c); }      // model.hip: before a derived weight copy is freed

int rama_ctx_destroy(rama_ctx* c) {
    if (c && RAMA_PENDING(c)) (void)flush_pending(c);
    if (!c) return 0;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    drop_graph(c);
    for (auto e : c->kp.ev) hipEventDestroy(e);
    hipFree(c->ctl); hipFree(c->out); hipFree(c->forced); hipFree(c->argmax_result); hipFree(c->pbar); hipFree(c->attn_counter); hipFree(c->attn_part); hipFree(c->attn_scores); hipFree(c->fused_hand); hipFree(c->fused_epoch); hipFree(c->lead_slots);
    for (int i = 0; i < 2; i++) { hipFree(c->topp_keys[i]); hipFree(c->topp_vals[i]); }
    hipFree(c->topp_prefix); hipFree(c->topp_m); hipFree(c->topp_err); hipFree(c->pf_blob); hipFree(c->pc_blob); if (c->ring) hipHostFree(c->ring);
    hipFree(c->topp_bp); hipFree(c->topp_bi); hipFree(c->topp_bcount); hipFree(c->topp_racc);
    hipFree(c->topp_rk); hipFree(c->topp_bm); hipFree(c->topp_approx); hipFree(c->topp_dist); hipFree(c->topp_stats);
    hipFree(c->bc.toks); hipFree(c->bc.seqs); hipFree(c->bc.out); if (c->bc.ring) hipHostFree(c->bc.ring);
    hipHostFree(c->pinned_int); hipHostFree(c->pinned_tok);
    hipEventDestroy(c->t0); hipEventDestroy(c->t1);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

// The launches that hand data over INSIDE a kernel (attention+Wo, the one-launch stage) bound every wait and report a wait that gave up
// through the error word at pbar[1]; their results are then invalid.  Every synchronising exit reads the word -- once the stream is idle
// -- fails the call and clears it, so that neither garbage logits leave with rc 0 nor a stale word makes every later launch give up.
static int topp_dist_check(rama_ctx* c);
static int handoff_check(rama_ctx* c) {
    { const int rt = topp_dist_check(c); if (rt) return rt; }
    if (!c->handoff_dirty || !c->pbar) return 0;
    unsigned long long perr = 0;
    HIPCHK(hipMemcpyAsync(&perr, c->pbar + 1, sizeof perr, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->handoff_dirty = false;
    if (perr != 0) {
        hipMemsetAsync(c->pbar, 0, 4 * sizeof(unsigned long long), c->stream);   // counter, error word, base: start over
        hipStreamSynchronize(c->stream);
        return fail(RAMA_EINVAL, perr >= 0x3000ull ? "one-launch stage: a hand-off timed out" : "attention+Wo launch: hand-off timed out", __FILE__, __LINE__);
    }
    return 0;
}

int rama_sync(rama_ctx* c) {
    RAMA_ENTER(c);
    REQUIRE(c, RAMA_EINVAL, "rama_sync: ctx is NULL");
    HIPCHK(hipStreamSynchronize(c->stream));
    return handoff_check(c);
}

// 0: everything enqueued on the context's stream has run; 1: work is still running; anything else: the stream has failed (the error
// is also recorded for rama_last_error).  Never blocks -- what a host loop that polls the token rings uses to know when to stop.
int rama_stream_query(rama_ctx* c) {
    RAMA_ENTER(c);
    REQUIRE(c, RAMA_EINVAL, "rama_stream_query: ctx is NULL");
    const hipError_t e = hipStreamQuery(c->stream);
    if (e == hipSuccess) return 0;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 1; }
    return fail((int)e, "hipStreamQuery", __FILE__, __LINE__);
}

int rama_device_info(rama_ctx* c, char name[64], int* cus, size_t* hbm) {
    RAMA_ENTER(c);
    REQUIRE(c, RAMA_EINVAL, "ctx is NULL");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, c->device));
    if (name) {   // marketing name needs amdgpu.ids, absent on some boxes: fall back to the ISA name
        strncpy(name, prop.name[0] ? prop.name : prop.gcnArchName, 63);
        name[63] = 0;
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm) *hbm = prop.totalGlobalMem;
    return 0;
}

// ---------------------------------------------------------------- memory

int rama_alloc_f32(rama_ctx* c, size_t n, float** out) {
    RAMA_ENTER(c);
    REQUIRE(c && out, RAMA_EINVAL, "rama_alloc_f32: NULL argument");
    if (set_device(c)) return 1;
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float)));
    HIPCHK(hipMemsetAsync(p, 0, std::max<size_t>(n, 1) * sizeof(float), c->stream));
    *out = (float*)p;
    rama_internal_note_alloc((const float*)p, std::max<size_t>(n, 1));
    return 0;
}

int rama_copy_h2d_f32(rama_ctx* c, float* dst, const float* host, size_t n) {
    RAMA_ENTER(c);
    REQUIRE(c && (n == 0 || (dst && host)), RAMA_EINVAL, "rama_copy_h2d_f32: NULL argument");
    if (n == 0) return 0;
    { const int rf = rama_internal_forget_range(c, dst, n, 0); if (rf) return rf; }      // copies derived from what is overwritten here
    HIPCHK(hipMemcpyAsync(dst, host, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));   // htod_sync_copy semantics (hbm.rs:14-16)
    return 0;
}

int rama_upload_f32(rama_ctx* c, const float* host, size_t n, float** out) {
    RAMA_ENTER(c);
    int rc = rama_alloc_f32(c, n, out);
    if (rc) return rc;
    return rama_copy_h2d_f32(c, *out, host, n);
}

int rama_download_f32(rama_ctx* c, const float* src, size_t n, float* host) {
    RAMA_ENTER(c);
    REQUIRE(c && (n == 0 || (src && host)), RAMA_EINVAL, "rama_download_f32: NULL argument");
    if (n == 0) return 0;
    HIPCHK(hipMemcpyAsync(host, src, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return handoff_check(c);
}

int rama_free(rama_ctx* c, void* p) {
    RAMA_ENTER(c);
    REQUIRE(c, RAMA_EINVAL, "ctx is NULL");
    if (!p) return 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    { const int rf = rama_internal_forget_range(c, (const float*)p, 1, 1); if (rf) return rf; }      // (any range inside the allocation: entries are matched by overlap with it below)
    HIPCHK(hipFree(p));
    return 0;
}

// ---------------------------------------------------------------- launch helpers

static inline int ew_grid(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 2048); }
static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Matvec workgroup geometry: R_ rows (R2_ (w1,w3) row pairs) per workgroup of NW_ waves, CH_
// 1-KiB chunks per wave per step.  Geometry 3 (4 rows x 8 waves x 2 chunks) ships: in the
// full llama2-7B decode it gives 233 tok/s vs 223 / 221 / 231 for 0 / 1 / 2 (tools/tune.py).
// The bare-matvec sweep (tools/gemv_bench.hip) prefers 2 rows per workgroup, but the kernels
// with rmsnorm folded in re-read x and the norm gain per workgroup, which 4 rows amortise.
// The others stay selectable for A/B runs through rama_set_tuning(ctx, "geom", g).
#define DISPATCH_GEOM(c, KERNEL_CALL)                                                                          \
    switch ((c)->tune_geom) {                                                                                  \
        case 1: { constexpr int R_ = 2, R2_ = 1, CH_ = 4, NW_ = 4; (void)R_; (void)R2_; KERNEL_CALL; } break;  \
        case 2: { constexpr int R_ = 4, R2_ = 2, CH_ = 4, NW_ = 4; (void)R_; (void)R2_; KERNEL_CALL; } break;  \
        case 0: { constexpr int R_ = 2, R2_ = 1, CH_ = 2, NW_ = 8; (void)R_; (void)R2_; KERNEL_CALL; } break;  \
        case 4: { constexpr int R_ = 8, R2_ = 4, CH_ = 2, NW_ = 8; (void)R_; (void)R2_; KERNEL_CALL; } break;  \
        default: { constexpr int R_ = 4, R2_ = 2, CH_ = 2, NW_ = 8; (void)R_; (void)R2_; KERNEL_CALL; } break; \
    }

// Per-kernel timing: while a kernel class is being profiled, its next launch carries a start and
// a stop event ON THE DISPATCH ITSELF (hipExtLaunchKernelGGL), so the interval is the kernel's own
// begin..end as rocprofv3 sees it -- separate event records around the launch add ~3 us.
struct KTimer {
    rama_ctx* c; bool on;
    KTimer(rama_ctx* c_, int kid) : c(c_), on(false) {
        KProf& k = c->kp;
        if (k.kernel_id == kid && k.used < k.max_records) {
            on = true;
            c->cur_start = k.ev[2 * k.used]; c->cur_stop = k.ev[2 * k.used + 1];
        }
    }
    ~KTimer() {
        if (on) { c->kp.used++; c->cur_start = c->cur_stop = nullptr; }
    }
};
// launch on the context's stream; the first launch inside an armed KTimer scope takes the events
#define RAMA_LAUNCH(c, kernel, grid, block, shm, ...)                                                          \
    do {                                                                                                        \
        if ((c)->cur_start) {                                                                                   \
            hipExtLaunchKernelGGL(kernel, grid, block, shm, (c)->stream, (c)->cur_start, (c)->cur_stop, 0, __VA_ARGS__); \
            (c)->cur_start = nullptr;                                                                           \
        } else {                                                                                                \
            hipLaunchKernelGGL(kernel, grid, block, shm, (c)->stream, __VA_ARGS__);                             \
        }                                                                                                       \
    } while (0)

// the pending run of apply_position calls (rama_ctx::rope) as one launch: `count` consecutive heads are one vector of count x head_size floats
static int flush_rope(rama_ctx* c) {
    if (!c->rope.count) return 0;
    const int hs = c->rope.hs, dim = c->rope.count * hs, n = dim / 2;
    c->rope.count = 0;
    if (c->tune_ref_order) hipLaunchKernelGGL(rope_ref_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->rope.q, c->rope.k, (const float*)nullptr, c->rope.pr, c->rope.pi, dim, hs,
                                              (float*)nullptr, (float*)nullptr);
    else hipLaunchKernelGGL(apply_position_heads_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->rope.q, c->rope.k, c->rope.pr, c->rope.pi, hs, dim);
    LAUNCHCHK();
    return 0;
}
// the pending elementwise op (rama_ctx::ew) by itself
static int flush_ew(rama_ctx* c) {
    if (!c->ew.kind) return 0;
    const int kind = c->ew.kind;
    c->ew.kind = 0;
    const dim3 grid((unsigned)std::min<size_t>((c->ew.n + 255) / 256, 2048));
    if (kind == 1) {
        if (c->tune_ref_order) hipLaunchKernelGGL(sinu_ref_kernel, grid, dim3(256), 0, c->stream, c->ew.t, c->ew.n);
        else hipLaunchKernelGGL(sinu_kernel, grid, dim3(256), 0, c->stream, c->ew.t, c->ew.n);
    } else hipLaunchKernelGGL(copy_kernel, grid, dim3(256), 0, c->stream, c->ew.t, c->ew.s, c->ew.n);
    LAUNCHCHK();
    return 0;
}
static int check_matvec_shape(size_t width, size_t rows) {
    REQUIRE(width % 4 == 0, RAMA_EINVAL, "matmul: width % 4 != 0 (the reference CPU body panics here, cpu.rs:142-143)");
    REQUIRE(width > 0 && rows > 0, RAMA_EINVAL, "matmul: empty shape");
    REQUIRE((double)width * (double)rows * 4.0 < 2147483648.0, RAMA_EINVAL, "matmul: matrix >= 2 GiB, split it by layer");
    return 0;
}

static bool use_solo(const rama_ctx* c, int K) { return c->tune_solo < 0 ? K <= 2048 : c->tune_solo != 0; }

// plain W.x (EPI_STORE) or x += W.y (EPI_RESID)
template <bool NORM, int EPI>
static int launch_rows(rama_ctx* c, float* o, const float* W, const float* x, const float* nw, int K, int rows) {
    int rc = check_matvec_shape(K, rows);
    if (rc) return rc;
    GemvParams p{};
    p.w[0] = W; p.x = x; p.nw = nw; p.o[0] = o; p.K = K; p.rows = rows; p.nmat = 1;
    if (use_solo(c, K)) {      // small-K: one wave per 4 rows (kernels.hpp gemv_rows_solo)
        const dim3 grid(((rows + 3) / 4 + kSoloWaves - 1) / kSoloWaves), block(kSoloWaves * 64);
        if (K <= 512) RAMA_LAUNCH(c, (gemv_rows_solo<4, 2, NORM, EPI>), grid, block, 0, p);
        else RAMA_LAUNCH(c, (gemv_rows_solo<4, 4, NORM, EPI>), grid, block, 0, p);
        LAUNCHCHK();
        return 0;
    }
    if (!NORM && EPI == EPI_RESID && c->tune_geom == 3 && c->tune_resid_r2) {
        // the residual matvecs (Wo, W2) read no rmsnorm gain, so a 2-row workgroup's re-read of
        // x is cheap and the finer grain balances better; rows wider than
        // 8192 floats (W2) can also spread over 16 waves (resid_r2 = 2 / 3)
        if (K > 8192 && c->tune_resid_r2 == 2) RAMA_LAUNCH(c, (gemv_rows<2, 1, 16, NORM, EPI>), dim3((rows + 1) / 2), dim3(16 * 64), 0, p);
        else if (K > 8192 && c->tune_resid_r2 == 3) RAMA_LAUNCH(c, (gemv_rows<2, 4, 16, NORM, EPI>), dim3((rows + 1) / 2), dim3(16 * 64), 0, p);
        else RAMA_LAUNCH(c, (gemv_rows<2, 2, 8, NORM, EPI>), dim3((rows + 1) / 2), dim3(8 * 64), 0, p);
        LAUNCHCHK();
        return 0;
    }
    DISPATCH_GEOM(c, RAMA_LAUNCH(c, (gemv_rows<R_, CH_, NW_, NORM, EPI>), dim3((rows + R_ - 1) / R_), dim3(NW_ * 64), 0, p));
    LAUNCHCHK();
    return 0;
}

// ---------------------------------------------------------------- reference-order launches (ref_order.hpp)

static int launch_matvec_ref(rama_ctx* c, int nmat, float* const o[3], const float* const W[3], const float* x, int K, int rows) {
    REQUIRE(K % 4 == 0 && K > 0 && rows > 0, RAMA_EINVAL, "matmul: width % 4 != 0 (the reference CPU body panics here, cpu.rs:142-143)");
    RefMatParams p{};
    for (int m = 0; m < nmat; m++) { p.w[m] = W[m]; p.o[m] = o[m]; }
    p.x = x; p.K = K; p.rows = rows; p.lane_reduce = c->tune_lane_reduce;
    hipLaunchKernelGGL(matvec_ref_kernel, dim3((rows + 63) / 64, nmat), dim3(64), 0, c->stream, p);
    LAUNCHCHK();
    return 0;
}
static int launch_matvec_ref1(rama_ctx* c, float* o, const float* W, const float* x, int K, int rows) {
    float* const oo[3] = {o, nullptr, nullptr};
    const float* const ww[3] = {W, nullptr, nullptr};
    return launch_matvec_ref(c, 1, oo, ww, x, K, rows);
}
static int launch_rmsnorm_ref(rama_ctx* c, float* o, const float* x, const float* w, int n) {
    REQUIRE((size_t)n * sizeof(float) <= 64 * 1024, RAMA_EUNSUP, "rmsnorm (reference order): vector longer than 16384");
    hipLaunchKernelGGL(rmsnorm_ref_kernel, dim3(1), dim3(1024), (size_t)n * sizeof(float), c->stream, o, x, w, n);
    LAUNCHCHK();
    return 0;
}
static int launch_attention_ref(rama_ctx* c, float* xb, float* att, const float* q, const float* kc_layer, const float* vc_layer,
                                const Ctl* ctl, int pos, int dim, int head_size, int seq_len, int n_heads) {
    REQUIRE((size_t)seq_len * sizeof(float) <= 64 * 1024, RAMA_EUNSUP, "attention (reference order): seq_len longer than 16384");
    RefAttnParams p{};
    p.q = q; p.kc = kc_layer; p.vc = vc_layer; p.att = att; p.xb = xb; p.ctl = ctl; p.pos_val = pos;
    p.dim = dim; p.head_size = head_size; p.seq_len = seq_len;
    hipLaunchKernelGGL(attention_ref_kernel, dim3(n_heads), dim3(1024), (size_t)seq_len * sizeof(float), c->stream, p);
    LAUNCHCHK();
    return 0;
}

// ---------------------------------------------------------------- parity mode at streaming speed (chain.hpp)

// W waves per 16 rows, D blocks per wave in flight: by how many row groups share a CU (few groups -> more waves
// and deeper rings per group, so that >= ~128 KiB per CU are on the way); tune_chain_d = 100 W + D overrides
// norm: how the rmsnorm in front of the product is folded in when p.nw is given -- CNORM_EXACT (parity mode, narrow models: the exact
// sequential sum by lane ripples, K <= 1024), CNORM_TREE (tolerance mode: a tree-shaped sum, K <= 8192)
static bool chain_norm_fits(int K, int norm) { return K % 16 == 0 && K <= (norm == CNORM_TREE ? 8192 : 1024); }
template <int EPI>
static int launch_chain(rama_ctx* c, ChainParams& p, int norm = CNORM_NONE) {
    REQUIRE(p.K % 16 == 0 && p.K > 0 && p.rows > 0, RAMA_EINVAL, "chain-order matvec: width must be a multiple of 16");
    REQUIRE((p.nw != nullptr) == (norm != CNORM_NONE), RAMA_EINVAL, "chain-order matvec: a folded norm needs its gain vector");
    const int groups = p.nmat * ((p.rows + 15) / 16);
    p.lane_reduce = c->tune_lane_reduce;
    int W, D;
    if (c->tune_chain_d > 0) { W = c->tune_chain_d / 100; D = c->tune_chain_d % 100; }
    else if (EPI == CEPI_RESID && norm == CNORM_NONE && c->tune_chain_resid_d > 0) { W = c->tune_chain_resid_d / 100; D = c->tune_chain_resid_d % 100; }
    else if (EPI == CEPI_RESID && norm == CNORM_NONE && c->tune_chain_resid_d < 0 && groups <= std::max(c->cu_count, 1) && p.K / 16 > 64) {
        // [r5] the residual products with at most one row group per compute unit (llama2-7B's Wo and W2): ONE wave with a ring of 32 blocks -- no relay
        // through LDS, no barrier per chunk -- Wo 15.2 -> 13.6 us, W2 33.7 -> 32.7 (profiles/r05_experiments.md)
        W = 1; D = 32;
    }
    else {
        const int cus = std::max(c->cu_count, 1);
        if (groups >= 5 * cus) { W = 1; D = 16; }   // W1 | W3, classifier: several groups per SIMD, each with its own ring
        else { W = 2; D = 16; }                     // one to three groups per CU (tools/chain_sweep.py: 216 best or equal everywhere)
        if (p.K / 16 <= 2 * D) { W = 1; D = 16; }   // a row of a few blocks: nothing to relay
    }
    if (norm == CNORM_LEAD && c->tune_chain_lead_w > 0 && c->tune_chain_d <= 0) { W = c->tune_chain_lead_w; D = 16; }      // ("chain_lead_w": the leader launches' waves per row group)
    if ((size_t)(p.K + chain_pad_floats(W, D, 4)) * sizeof(float) > 64 * 1024) { W = 1; D = 16; }
    const size_t lds = (size_t)(p.K + chain_pad_floats(W, D, 4)) * sizeof(float);      // x + the zeros behind it
    REQUIRE(lds <= 64 * 1024, RAMA_EUNSUP, "chain-order matvec: row longer than ~15800 floats");
    // [r5] one wave per row group and more groups than compute units: the groups that do not divide by the CUs as halves (chain.hpp half_from; "chain_split")
    int nblocks = groups;
    p.half_from = 0;
    if (c->tune_chain_split && W == 1 && D == 16 && c->tune_chain_d <= 0 && norm != CNORM_EXACT && norm != CNORM_TREE) {
        const int cus = std::max(c->cu_count, 1), rem = groups % cus;
        // (more than half the CUs with a group more -- the classifier's 2 000 groups: the halves would give some CUs two again)
        if (groups > cus && rem > 0 && 2 * rem <= cus) { p.half_from = groups - rem; nblocks = groups + rem; }
    }
    const bool wo_like = EPI == CEPI_RESID && norm == CNORM_NONE && p.K == p.rows;      // the square residual product: Wo
    const dim3 grid(nblocks);
    if (norm == CNORM_LEAD) {      // the exact sum by a leader workgroup of this launch (grid + 1); geometry as without a norm
        REQUIRE(D == 16 && (W == 1 || W == 2) && p.K % 8 == 0 && p.K <= 4096 * W && (EPI == CEPI_QKV || EPI == CEPI_SWIGLU || EPI == CEPI_STORE) && p.lead && p.epoch && p.err,
                RAMA_EUNSUP, "chain-order matvec: no leader-norm instantiation for this shape");
        constexpr int E3 = (EPI == CEPI_QKV || EPI == CEPI_SWIGLU || EPI == CEPI_STORE) ? EPI : CEPI_QKV;
        const int per = (p.K + 64 * W - 1) / (64 * W);
        const size_t ldsl = std::max(lds, W == 1 ? sizeof(FastSumShared<1>) : sizeof(FastSumShared<2>));
        const dim3 gridl(nblocks + 1);
#define RAMA_CHAIN_L(W_, LR_) RAMA_LAUNCH(c, (gemv_chain_kernel<W_, 16, 4, E3, CNORM_LEAD, LR_>), gridl, dim3(W_ * 64), ldsl, p)
#define RAMA_CHAIN_LW(W_) do { if (per <= 8) RAMA_CHAIN_L(W_, 8); else if (per <= 16) RAMA_CHAIN_L(W_, 16); else if (per <= 32) RAMA_CHAIN_L(W_, 32); else RAMA_CHAIN_L(W_, 64); } while (0)
        if (W == 1) RAMA_CHAIN_LW(1); else RAMA_CHAIN_LW(2);
#undef RAMA_CHAIN_LW
#undef RAMA_CHAIN_L
        LAUNCHCHK();
        c->handoff_dirty = true;
        return 0;
    }
    if (norm != CNORM_NONE) {      // the rmsnorm folded in: all of x sits in the workgroup's registers (K <= 64 x threads)
        if (c->tune_chain_d <= 0 && W == 1 && p.K > 4096) W = 2;
        REQUIRE(chain_norm_fits(p.K, norm) && p.K <= 4096 * W && D == 16 && (W == 1 || W == 2) && EPI != CEPI_RESID, RAMA_EUNSUP, "chain-order matvec: no norm-folding instantiation for this shape");
        size_t ldsn = (size_t)(p.K + chain_pad_floats(W, D, 4)) * sizeof(float);
        if (norm == CNORM_EXACT) ldsn += ((size_t)p.K + ((size_t)p.K >> 5) + 4) * sizeof(float);      // + the squares, scan_slot layout
        constexpr int E = EPI != CEPI_RESID ? EPI : CEPI_QKV;
#define RAMA_CHAIN_N(W_, N_) RAMA_LAUNCH(c, (gemv_chain_kernel<W_, 16, 4, E, N_>), grid, dim3(W_ * 64), ldsn, p)
        if (norm == CNORM_TREE) { if (W == 1) RAMA_CHAIN_N(1, CNORM_TREE); else RAMA_CHAIN_N(2, CNORM_TREE); }
        else {
            REQUIRE(EPI == CEPI_QKV || EPI == CEPI_SWIGLU, RAMA_EUNSUP, "chain-order matvec: the lane-ripple norm folds into Wq|Wk|Wv and W1|W3 only");
            constexpr int E2 = (EPI == CEPI_QKV || EPI == CEPI_SWIGLU) ? EPI : CEPI_QKV;
            if (W == 1) RAMA_LAUNCH(c, (gemv_chain_kernel<1, 16, 4, E2, CNORM_EXACT>), grid, dim3(64), ldsn, p);
            else RAMA_LAUNCH(c, (gemv_chain_kernel<2, 16, 4, E2, CNORM_EXACT>), grid, dim3(128), ldsn, p);
        }
#undef RAMA_CHAIN_N
        LAUNCHCHK();
        return 0;
    }
#define RAMA_CHAIN(W_, D_) RAMA_LAUNCH(c, (gemv_chain_kernel<W_, D_, 4, EPI>), grid, dim3(W_ * 64), lds, p)
    if (W == 1 && D == 16) RAMA_CHAIN(1, 16);
    else if (W == 1 && D == 32 && wo_like) RAMA_LAUNCH(c, (gemv_chain_kernel<1, 32, 4, EPI, CNORM_NONE, 64, 1>), grid, dim3(64), lds, p);      // (Wo under a name of its own)
    else if (W == 1 && D == 32) RAMA_CHAIN(1, 32);
    else if (W == 2 && D == 16) RAMA_CHAIN(2, 16);
    else if (W == 2 && D == 32) RAMA_CHAIN(2, 32);

    else if (W == 4 && D == 16) RAMA_CHAIN(4, 16);
    else if (W == 4 && D == 32) RAMA_CHAIN(4, 32);
    else return fail(RAMA_EINVAL, "chain-order matvec: no such geometry", __FILE__, __LINE__);
#undef RAMA_CHAIN
    LAUNCHCHK();
    return 0;
}
// the pending run of parity-mode matmuls (rama_ctx::mm) as one chain-order launch
static bool rmsnorm_chain_ok(size_t n) { return n <= (size_t)kNormMax && (n + (n >> 5) + 2) * sizeof(float) <= 64 * 1024; }
static int launch_rmsnorm_chain(rama_ctx* c, float* o, const float* x, const float* w, int n, float* copy_to, int batch = 1, int stride = 0) {
    const size_t lds = ((size_t)n + ((size_t)n >> 5) + 2) * sizeof(float);
    RAMA_LAUNCH(c, rmsnorm_chain_kernel, dim3(batch), dim3(kNormThreads), lds, o, x, w, n, copy_to, stride);
    LAUNCHCHK();
    return 0;
}
// the recorded norm (rama_ctx::nrm) as a launch of its own: no run of matmuls took it
static int flush_norm(rama_ctx* c) {
    if (!c->nrm.on) return 0;
    c->nrm.on = false;
    return launch_rmsnorm_chain(c, c->nrm.o, c->nrm.x, c->nrm.w, c->nrm.n, nullptr);
}
static int flush_mm(rama_ctx* c) {
    if (!(c->mm.count && c->mm.norm)) { const int rn = flush_norm(c); if (rn) return rn; }
    if (!c->mm.count) return 0;
    ChainParams p{};
    for (int i = 0; i < c->mm.count; i++) { p.w[i] = c->mm.w[i]; p.o[i] = c->mm.o[i]; }
    p.x = c->mm.x; p.K = c->mm.K; p.rows = c->mm.rows; p.nmat = c->mm.count;
    c->mm.count = 0;
    if (c->mm.norm) {      // the run reads the recorded norm's output: the norm rides in the run's launch, whose leader also stores that output
        c->mm.norm = false; c->nrm.on = false;
        p.x = c->nrm.x; p.nw = c->nrm.w; p.xout = c->nrm.o;
        p.lead = c->lead_slots + 32 * (kLeadSlots + c->op_lead_next); p.epoch = c->fused_epoch; p.err = c->pbar + 1;
        const int rc = launch_chain<CEPI_STORE>(c, p, CNORM_LEAD);
        if (rc) return rc;
        if (++c->op_lead_next == kLeadSlots) {      // every word of the range carries this epoch: the next one
            c->op_lead_next = 0;
            hipLaunchKernelGGL(fused_epoch_kernel, dim3(1), dim3(1), 0, c->stream, c->fused_epoch);
            LAUNCHCHK();
        }
        return 0;
    }
    return launch_chain<CEPI_STORE>(c, p);
}
static int attn_chain_waves(int head_size, bool long_ctx) {
    const int want = long_ctx ? 8 : 4;
    return attn_chain_fits(head_size, want) ? want : (attn_chain_fits(head_size, 8) ? 8 : 16);
}
static bool attn_chain_ok(int head_size, int seq_len) {      // the largest variant a launch may pick must fit
    if (head_size % 4 || !attn_chain_fits(head_size, 16)) return false;
    return attn_chain_lds_floats(head_size, seq_len, attn_chain_waves(head_size, true)) * sizeof(float) + 16 <= kAttnChainMaxLds;
}
// the score scratch of parity mode's spread attention; inside a stream capture nothing is allocated (the caller then keeps
// the three-launch form, whose softmax rewrites each head's row from ONE workgroup)
static int ensure_attn_scores(rama_ctx* c, int n_heads, int seq_len) {
    const size_t need = (size_t)n_heads * (size_t)seq_len;
    if (need <= c->attn_scores_floats) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return 0; }
    if (c->attn_scores) { HIPCHK(hipStreamSynchronize(c->stream)); drop_graph(c); HIPCHK(hipFree(c->attn_scores)); c->attn_scores = nullptr; c->attn_scores_floats = 0; }      // (graphs captured for a smaller model hold the old address)
    HIPCHK(hipMalloc(&c->attn_scores, need * sizeof(float)));
    c->attn_scores_floats = need;
    return 0;
}

// long_ctx: 8 waves per head (twice the timesteps per score round, twice the loaders of the value tiles) -- from position 256 on
static int launch_attention_chain(rama_ctx* c, float* xb, float* att, const float* q, const float* kc_layer, const float* vc_layer,
                                  const Ctl* ctl, int pos, int dim, int head_size, int seq_len, int n_heads, bool long_ctx = false, bool spread_wanted = false) {
    // spread: the three-launch form for long contexts; it needs whole staged pieces and 32-column slices, a score
    // buffer that fits the softmax kernel's LDS, and the att buffer
    const bool spread = spread_wanted && att && head_size % kAttPiece == 0 && head_size % kValCols == 0 && head_size <= 256 &&
                        ((size_t)seq_len + ((size_t)seq_len >> 5) + 4) * sizeof(float) <= 60 * 1024;
    REQUIRE(aligned16(q) && aligned16(kc_layer) && aligned16(vc_layer) && dim % 4 == 0, RAMA_EINVAL, "attention: buffers must be 16-byte aligned");
    RefAttnParams p{};
    p.q = q; p.kc = kc_layer; p.vc = vc_layer; p.att = att; p.xb = xb; p.ctl = ctl; p.pos_val = pos;
    p.dim = dim; p.head_size = head_size; p.seq_len = seq_len;
    if (spread) {
        // two launches over the whole chip (chain.hpp; three with "attn_fv" = 0): needs att (the scores / probabilities travel through it)
        // (an armed kernel-class timer brackets the whole attention: its start event rides on the first launch, its stop event on the last)
        const int ngroups = (seq_len + 63) / 64;
        // the one-launch softmax + values form reads the scores in every (head, slice) workgroup while the slice-0 workgroup of the head writes
        // the probabilities: the scores therefore travel through a scratch of their own, never through the buffer that receives the probabilities
        const size_t fv_lds = attn_fused_values_lds_floats(seq_len) * sizeof(float);
        bool fv = c->tune_attn_fv && fv_lds <= 32 * 1024;
        if (fv) {
            const int rs = ensure_attn_scores(c, n_heads, seq_len); if (rs) return rs;
            fv = c->attn_scores_floats >= (size_t)n_heads * (size_t)seq_len;
        }
        p.sc = fv ? c->attn_scores : att;
        hipEvent_t ev_start = c->cur_start, ev_stop = c->cur_start ? c->cur_stop : nullptr;
        c->cur_start = nullptr;
        if (ev_start) hipExtLaunchKernelGGL(attn_scores_chain_kernel, dim3(n_heads, ngroups), dim3(64), 0, c->stream, ev_start, nullptr, 0, p);
        else hipLaunchKernelGGL(attn_scores_chain_kernel, dim3(n_heads, ngroups), dim3(64), 0, c->stream, p);
        LAUNCHCHK();
        if (fv) {             // the softmax repeated by every slice workgroup, one launch (chain.hpp [r4])
            if (ev_stop) hipExtLaunchKernelGGL(attn_softmax_values_chain_kernel, dim3(n_heads, head_size / kValCols), dim3(kFvSoftWaves * 64), fv_lds, c->stream, nullptr, ev_stop, 0, p);
            else hipLaunchKernelGGL(attn_softmax_values_chain_kernel, dim3(n_heads, head_size / kValCols), dim3(kFvSoftWaves * 64), fv_lds, c->stream, p);
            LAUNCHCHK();
            return 0;
        }
        hipLaunchKernelGGL(attn_softmax_chain_kernel, dim3(n_heads), dim3(kSoftWaves * 64), ((size_t)seq_len + ((size_t)seq_len >> 5) + 4) * sizeof(float), c->stream, p);
        LAUNCHCHK();
        if (ev_stop) hipExtLaunchKernelGGL(attn_values_chain_kernel, dim3(n_heads, head_size / kValCols), dim3(kValWaves * 64), 0, c->stream, nullptr, ev_stop, 0, p);
        else hipLaunchKernelGGL(attn_values_chain_kernel, dim3(n_heads, head_size / kValCols), dim3(kValWaves * 64), 0, c->stream, p);
        LAUNCHCHK();
        return 0;
    }
    const int nw = attn_chain_waves(head_size, long_ctx);
    const size_t lds = attn_chain_lds_floats(head_size, seq_len, nw) * sizeof(float) + 16;
    REQUIRE(lds <= kAttnChainMaxLds, RAMA_EUNSUP, "attention (parity mode): context too long for the score buffer");
#define RAMA_ATTN_CHAIN(NW_) RAMA_LAUNCH(c, (attention_chain_kernel<NW_>), dim3(n_heads), dim3(NW_ * 64), lds, p)
    if (nw == 4) RAMA_ATTN_CHAIN(4);
    else if (nw == 8) RAMA_ATTN_CHAIN(8);
    else RAMA_ATTN_CHAIN(16);
#undef RAMA_ATTN_CHAIN
    LAUNCHCHK();
    return 0;
}

// ---------------------------------------------------------------- Device<T> ops, 1:1

static inline bool ranges_overlap(const float* p0, size_t n0, const float* p1, size_t n1) { return p0 < p1 + n1 && p1 < p0 + n0; }
int rama_array_add(rama_ctx* c, float* t, const float* s, size_t n) {
    REQUIRE(c && (n == 0 || (t && s)), RAMA_EINVAL, "array_add: NULL argument");
    RAMA_WRITES(c, t, n);
    // [r5] the recorded matmul whose output this call adds to `t` (infer.rs:35-37, :46-47): ONE launch with the residual epilogue (product stored, t += product)
    if (c->mm.count == 1 && !c->mm.norm && !c->nrm.on && !c->rope.count && !c->ew.kind && c->tune_ref_order && c->tune_resid_fold && s == c->mm.o[0] && n == (size_t)c->mm.rows &&
        aligned16(t) && !ranges_overlap(t, n, c->mm.x, (size_t)c->mm.K) && !ranges_overlap(t, n, s, n)) {
        ChainParams p{};
        p.w[0] = c->mm.w[0]; p.o[0] = c->mm.o[0]; p.resid = t; p.x = c->mm.x; p.K = c->mm.K; p.rows = c->mm.rows; p.nmat = 1;
        c->mm.count = 0;
        return launch_chain<CEPI_RESID>(c, p);
    }
    RAMA_ENTER(c);
    if (!n) return 0;
    hipLaunchKernelGGL(array_add_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, t, s, n);
    LAUNCHCHK(); return 0;
}
int rama_array_mult(rama_ctx* c, float* t, const float* s, size_t n) {
    REQUIRE(c && (n == 0 || (t && s)), RAMA_EINVAL, "array_mult: NULL argument");
    RAMA_WRITES(c, t, n);
    if (c->ew.kind == 1 && !c->rope.count && !c->mm.count && c->ew.t == t && c->ew.n == n && n && !ranges_overlap(s, n, t, n)) {      // the recorded sinu and this product: one launch
        c->ew.kind = 0;
        if (c->tune_ref_order) hipLaunchKernelGGL(sinu_mult_ref_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, t, s, n);
        else hipLaunchKernelGGL(sinu_mult_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, t, s, n);
        LAUNCHCHK(); return 0;
    }
    RAMA_ENTER(c);
    if (!n) return 0;
    hipLaunchKernelGGL(array_mult_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, t, s, n);
    LAUNCHCHK(); return 0;
}
int rama_sinu(rama_ctx* c, float* o, size_t n) {
    RAMA_ENTER(c);
    REQUIRE(c && (n == 0 || o), RAMA_EINVAL, "sinu: NULL argument");
    RAMA_WRITES(c, o, n);
    if (!n) return 0;
    if (c->tune_ew_batch && c->own_stream) { c->ew.kind = 1; c->ew.t = o; c->ew.s = nullptr; c->ew.n = n; return 0; }      // recorded: the product that usually follows takes it along
    if (c->tune_ref_order) hipLaunchKernelGGL(sinu_ref_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, o, n);
    else hipLaunchKernelGGL(sinu_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, o, n);
    LAUNCHCHK(); return 0;
}
int rama_copy_from_slice(rama_ctx* c, float* t, const float* s, size_t n) {
    REQUIRE(c && (n == 0 || (t && s)), RAMA_EINVAL, "copy_from_slice: NULL argument");
    RAMA_WRITES(c, t, n);
    // [r5] infer.rs:20-33 as the fused entry runs it: a pending run of three matmuls (a recorded norm in front, perhaps), every head of its first two outputs
    // rotated, and now its second and third outputs copied into cache rows -- ONE launch with the Wq|Wk|Wv epilogue.  The first copy is recorded, the second
    // issues the launch; anything that does not fit issues what is pending in program order.
    if (c->mm.count == 3 && c->rope.count && c->tune_qkv_fold && c->tune_ref_order && n == (size_t)c->mm.rows && c->rope.count * c->rope.hs == c->mm.rows) {
        const auto& m = c->mm;
        const size_t hh = (size_t)c->rope.hs / 2;
        auto apart = [&](const float* p0, size_t n0) {      // from everything the launch reads or writes besides
            bool ok = !ranges_overlap(p0, n0, m.o[0], n) && !ranges_overlap(p0, n0, m.o[1], n) && !ranges_overlap(p0, n0, m.o[2], n) && !ranges_overlap(p0, n0, m.x, (size_t)m.K) &&
                      !ranges_overlap(p0, n0, c->rope.pr, hh) && !ranges_overlap(p0, n0, c->rope.pi, hh);
            if (ok && m.norm) ok = !ranges_overlap(p0, n0, c->nrm.x, (size_t)c->nrm.n) && !ranges_overlap(p0, n0, c->nrm.w, (size_t)c->nrm.n);
            return ok;
        };
        if (!c->ew.kind && s == m.o[1] && aligned16(t) && apart(t, n) && c->tune_ew_batch) { c->ew.kind = 2; c->ew.t = t; c->ew.s = s; c->ew.n = n; return 0; }
        if (c->ew.kind == 2 && c->ew.s == m.o[1] && c->ew.n == n && s == m.o[2] && aligned16(t) && apart(t, n) && !ranges_overlap(t, n, c->ew.t, n) &&
            !ranges_overlap(c->rope.pr, hh, m.o[0], n) && !ranges_overlap(c->rope.pr, hh, m.o[1], n) && !ranges_overlap(c->rope.pr, hh, m.o[2], n) &&
            !ranges_overlap(c->rope.pi, hh, m.o[0], n) && !ranges_overlap(c->rope.pi, hh, m.o[1], n) && !ranges_overlap(c->rope.pi, hh, m.o[2], n)) {
            ChainParams p{};
            for (int i = 0; i < 3; i++) { p.w[i] = m.w[i]; p.o[i] = m.o[i]; }
            p.x = m.x; p.K = m.K; p.rows = m.rows; p.nmat = 3;
            p.fr = c->rope.pr; p.fi = c->rope.pi; p.head_size = c->rope.hs; p.kc = c->ew.t; p.vc = t; p.pos_val = 0;      // (the table rows and cache rows of this position themselves)
            const bool norm = m.norm;
            c->mm.count = 0; c->mm.norm = false; c->rope.count = 0; c->ew.kind = 0;
            if (!norm) return launch_chain<CEPI_QKV>(c, p);
            c->nrm.on = false;
            p.x = c->nrm.x; p.nw = c->nrm.w; p.xout = c->nrm.o;
            p.lead = c->lead_slots + 32 * (kLeadSlots + c->op_lead_next); p.epoch = c->fused_epoch; p.err = c->pbar + 1;
            const int rc = launch_chain<CEPI_QKV>(c, p, CNORM_LEAD);
            if (rc) return rc;
            if (++c->op_lead_next == kLeadSlots) { c->op_lead_next = 0; hipLaunchKernelGGL(fused_epoch_kernel, dim3(1), dim3(1), 0, c->stream, c->fused_epoch); LAUNCHCHK(); }
            return 0;
        }
        { const int rf = flush_pending(c); if (rf) return rf; }
    }
    if (c->ew.kind == 2 && !c->rope.count && !c->mm.count && n && !ranges_overlap(s, n, c->ew.t, c->ew.n) && !ranges_overlap(t, n, c->ew.t, c->ew.n) &&
        !ranges_overlap(t, n, c->ew.s, c->ew.n) && !ranges_overlap(t, n, s, n)) {      // the recorded copy and this one: one launch
        c->ew.kind = 0;
        hipLaunchKernelGGL(copy2_kernel, dim3(ew_grid(n + c->ew.n)), dim3(256), 0, c->stream, c->ew.t, c->ew.s, c->ew.n, t, s, n);
        LAUNCHCHK(); return 0;
    }
    RAMA_ENTER(c);
    if (!n) return 0;
    if (c->tune_ew_batch && c->own_stream && !ranges_overlap(t, n, s, n)) { c->ew.kind = 2; c->ew.t = t; c->ew.s = s; c->ew.n = n; return 0; }
    hipLaunchKernelGGL(copy_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, t, s, n);
    LAUNCHCHK(); return 0;
}
int rama_rmsnorm(rama_ctx* c, float* o, const float* x, const float* w, size_t n) {
    RAMA_ENTER(c);
    REQUIRE(c && o && x && w && n > 0, RAMA_EINVAL, "rmsnorm: bad argument");
    RAMA_WRITES(c, o, n);
    if (c->tune_ref_order && c->tune_chain && rmsnorm_chain_ok(n) && c->tune_norm_fold && c->tune_chain_lead && c->tune_matmul_batch && c->own_stream && c->tune_chain_d <= 0 &&
        c->tune_chain_lead_w <= 0 && c->kp.kernel_id < 0 && n % 16 == 0 && n <= 4096 && n >= 64 && aligned16(x) && aligned16(w) && aligned16(o) && (o + n <= x || x + n <= o) && c->lead_slots && c->pbar) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c->stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {      // recorded: the run of matmuls on `o` carries it (flush_mm)
            c->nrm.o = o; c->nrm.x = x; c->nrm.w = w; c->nrm.n = (int)n; c->nrm.on = true;
            return 0;
        }
        (void)hipGetLastError();
    }
    if (c->tune_ref_order) return c->tune_chain && rmsnorm_chain_ok(n) ? launch_rmsnorm_chain(c, o, x, w, (int)n, nullptr) : launch_rmsnorm_ref(c, o, x, w, (int)n);
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(1), dim3(1024), 0, c->stream, o, x, w, (int)n);
    LAUNCHCHK(); return 0;
}
int rama_apply_position(rama_ctx* c, float* q, float* k, const float* pr, const float* pi, size_t head_size) {
    REQUIRE(c && q && k && pr && pi && head_size >= 2, RAMA_EINVAL, "apply_position: bad argument");
    RAMA_WRITES(c, q, head_size); RAMA_WRITES(c, k, head_size);
    // (whatever was recorded before this call comes first -- except a run of three matmuls whose first two outputs this call starts to rotate, or goes on rotating)
    const bool on_run = c->mm.count == 3 && c->tune_qkv_fold && !c->ew.kind && c->tune_rope_batch && c->own_stream && head_size % 2 == 0 && c->mm.rows % (int)head_size == 0 &&
                        (c->rope.count ? true : (q == c->mm.o[0] && k == c->mm.o[1]));
    if (!on_run && (c->mm.count | c->ew.kind | (int)c->nrm.on)) { const int rm = flush_pending(c); if (rm) return rm; }      // (a run of rotations by itself goes on)
    if (c->tune_rope_batch && c->own_stream && head_size % 2 == 0 && head_size <= 4096) {
        auto& r = c->rope;
        const int hs = (int)head_size;
        if (r.count && r.count < 4096 && hs == r.hs && pr == r.pr && pi == r.pi && q == r.q + (size_t)r.count * hs && k == r.k + (size_t)r.count * hs &&
            (!c->mm.count || (r.count + 1) * hs <= c->mm.rows)) { r.count++; return 0; }
        const int rf = flush_pending(c); if (rf) return rf;
        r.q = q; r.k = k; r.pr = pr; r.pi = pi; r.hs = hs; r.count = 1;
        return 0;
    }
    { const int rf = flush_pending(c); if (rf) return rf; }
    int n = (int)(head_size / 2);
    if (c->tune_ref_order) {
        hipLaunchKernelGGL(rope_ref_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, q, k, (const float*)nullptr, pr, pi, (int)head_size, (int)head_size,
                           (float*)nullptr, (float*)nullptr);
        LAUNCHCHK(); return 0;
    }
    hipLaunchKernelGGL(apply_position_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, q, k, pr, pi, (int)head_size);
    LAUNCHCHK(); return 0;
}
int rama_matmul(rama_ctx* c, float* o, const float* a, const float* b, size_t width, size_t o_rows, size_t o_cols) {
    if (c && (c->rope.count | c->ew.kind)) { const int rf = flush_pending(c); if (rf) return rf; }      // (a pending run of matmuls may be extended by this call, a recorded norm taken along: below)
    REQUIRE(c && o && a && b, RAMA_EINVAL, "matmul: NULL argument");
    REQUIRE(o_cols >= 1, RAMA_EINVAL, "matmul: o_cols == 0");
    RAMA_WRITES(c, o, o_rows * o_cols);
    int rc = check_matvec_shape(width, o_rows);
    if (rc) { (void)flush_mm(c); return rc; }
    const bool extendable = c->mm.count && c->mm.count < 3 && c->tune_ref_order && o_cols == 1 && c->mm.x == b && c->mm.K == (int)width && c->mm.rows == (int)o_rows;
    if (!extendable && c->mm.count) { rc = flush_mm(c); if (rc) return rc; }      // (no run pending: a recorded norm stays, this call may open the run that takes it)
    if (c->tune_ref_order && o_cols == 1) {
        // a layer-aligned view of a resident model's matrix streams the model's chain-order copy
        if (c->tune_chain && width % 16 == 0 && width <= 16000 && aligned16(b)) {
            rc = rama_internal_model_ensure_ptr(c, a, 1); if (rc) return rc;
            // ... and a matrix that belongs to no model -- uploaded by itself (hbm.rs:14-16), or a view no model copy covers (w1, w3: a model keeps
            // them interleaved) -- gets a chain-order copy of its own on first use ("chain_views"; freed with the tensor: rama_free)
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(c->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
            if (const float* ch = c->tune_chain_views ? rama_internal_chain_view(c, a, (int)o_rows, (int)width, capturing ? 1 : 0) : rama_internal_chain_lookup(a, (int)o_rows, (int)width)) {
                auto& mmb = c->mm;
                // [r5] recorded, not launched: the next call extends the run (same activations, same shape, an output that overlaps nothing of the run) or
                // issues it.  A run is one launch over all its row groups.
                auto overlaps = [](const float* p0, size_t n0, const float* p1, size_t n1) { return p0 < p1 + n1 && p1 < p0 + n0; };
                bool clash = overlaps(o, o_rows, b, width);
                for (int i = 0; i < mmb.count && !clash; i++) clash = overlaps(o, o_rows, mmb.o[i], o_rows);
                // the recorded norm rides with a run on its output; the run then reads the norm's input, which no output of the run may touch
                const bool takes_norm = c->nrm.on && (mmb.count ? mmb.norm : (b == c->nrm.o && (int)width == c->nrm.n));
                if (takes_norm && overlaps(o, o_rows, c->nrm.x, (size_t)c->nrm.n)) clash = true;
                if (c->tune_matmul_batch && c->own_stream && !capturing && !clash) {
                    if (!mmb.count) {
                        mmb.x = b; mmb.K = (int)width; mmb.rows = (int)o_rows; mmb.norm = takes_norm;
                        if (!takes_norm) { rc = flush_norm(c); if (rc) return rc; }
                    }
                    mmb.w[mmb.count] = ch; mmb.o[mmb.count] = o; mmb.count++;
                    if (mmb.count == 3 && !(c->tune_qkv_fold && c->tune_rope_batch && c->tune_ew_batch)) return flush_mm(c);      // (else: the rotations and cache copies may follow)
                    return 0;
                }
                rc = flush_mm(c); if (rc) return rc;
                ChainParams p{};
                p.w[0] = ch; p.o[0] = o; p.x = b; p.K = (int)width; p.rows = (int)o_rows; p.nmat = 1;
                return launch_chain<CEPI_STORE>(c, p);
            }
        }
        rc = flush_mm(c); if (rc) return rc;
        return launch_matvec_ref1(c, o, a, b, (int)width, (int)o_rows);
    }
    // (none of the paths below records: whatever is pending -- a parity-mode Device::rmsnorm whose output this call may read -- is issued first)
    rc = flush_mm(c); if (rc) return rc;
    if (o_cols != 1) {   // forward() never takes this path (o_cols is always 1, infer.rs:20-51)
        size_t n = o_rows * o_cols;
        if (c->tune_ref_order) {      // the reference's rounding order for this shape too (cpu.rs:137-151)
            REQUIRE(width % 4 == 0, RAMA_EINVAL, "matmul: width % 4 != 0 (the reference CPU body panics here, cpu.rs:142-143)");
            hipLaunchKernelGGL(matmul_cols_ref_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, o, a, b, (int)width, (int)o_rows, (int)o_cols, c->tune_lane_reduce);
            LAUNCHCHK(); return 0;
        }
        hipLaunchKernelGGL(matmul_generic, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, o, a, b, (int)width, (int)o_rows, (int)o_cols);
        LAUNCHCHK(); return 0;
    }
    if (!aligned16(a) || !aligned16(b)) {   // a view that does not start on a 16-byte boundary
        hipLaunchKernelGGL(matvec_unaligned, dim3((unsigned)((o_rows + 3) / 4)), dim3(kWG), 0, c->stream, o, a, b, (int)width, (int)o_rows);
        LAUNCHCHK(); return 0;
    }
    return launch_rows<false, EPI_STORE>(c, o, a, b, nullptr, (int)width, (int)o_rows);
}
int rama_softmax(rama_ctx* c, float* x, size_t n) {
    RAMA_ENTER(c);
    REQUIRE(c && x && n > 0, RAMA_EINVAL, "softmax: bad argument");
    RAMA_WRITES(c, x, n);
    if (c->tune_ref_order && c->tune_chain && rmsnorm_chain_ok(n)) {
        hipLaunchKernelGGL(softmax_chain_kernel, dim3(1), dim3(kNormThreads), (n + (n >> 5) + 2) * sizeof(float), c->stream, x, (int)n);
        LAUNCHCHK(); return 0;
    }
    if (c->tune_ref_order) { hipLaunchKernelGGL(softmax_ref_kernel, dim3(1), dim3(1024), 0, c->stream, x, (int)n); LAUNCHCHK(); return 0; }
    hipLaunchKernelGGL(softmax_kernel, dim3(1), dim3(1024), 0, c->stream, x, (int)n);
    LAUNCHCHK(); return 0;
}

static int attn_nsplit(const rama_ctx* c, int n_heads) {
    if (c->tune_attn_nsplit > 0) return c->tune_attn_nsplit;
    return std::max(1, std::min(16, c->cu_count / std::max(n_heads, 1)));
}

// scratch for the split-T partials; called outside any stream capture (no allocation inside one)
static int ensure_attn_part(rama_ctx* c, const rama_config* cfg) {
    if (set_device(c)) return 1;
    const int hs = cfg->dim / cfg->n_heads;
    const size_t need = (size_t)cfg->n_heads * attn_nsplit(c, cfg->n_heads) * (hs + 4);
    if (need > c->attn_part_floats) {
        if (c->attn_part) { HIPCHK(hipStreamSynchronize(c->stream)); drop_graph(c); HIPCHK(hipFree(c->attn_part)); c->attn_part = nullptr; }      // (graphs captured for a smaller model hold the old address)
        HIPCHK(hipMalloc(&c->attn_part, need * sizeof(float)));
        c->attn_part_floats = need;
    }
    if (c->tune_ref_order) { const int rs = ensure_attn_scores(c, cfg->n_heads, cfg->seq_len); if (rs) return rs; }
    return 0;
}

// Split-T attention is worth its extra (combine) launch once a head's cache no longer fits a
// couple of single-workgroup rounds; below the threshold one workgroup per head is faster.
constexpr int kSplitTPos = 256;
// Measured (tools/split_sweep.py, round 2: 8-wave slices, nt cache loads): at llama2-7B (32 heads x 128)
// splitting wins from ~position 256 on (227 -> 231 tok/s at 300, 217 -> 229 at 600, 183 -> 219 at 1900) and
// loses below it (237 vs 231 at 200); at the stories110M shape (12 heads x 64, 1024
// positions) the single-workgroup kernel wins everywhere (2 837 vs 2 082 tok/s at 900), because
// its whole per-head cache is a few rounds of one workgroup and the combine launch costs more.
static int split_threshold(const rama_ctx* c, const rama_config* cfg) {
    if (c->tune_split_pos >= 0) return c->tune_split_pos;
    const int hs = cfg->dim / cfg->n_heads;
    const int G = hs <= 64 ? 16 : (hs <= 128 ? 32 : 64);
    // a context the one-workgroup kernel's LDS score buffer cannot hold (seq_len > ~15 000) must split
    if ((size_t)(attn_scratch_floats(G) + cfg->seq_len) * sizeof(float) > 64 * 1024) return kSplitTPos;
    const long kv_bytes_per_pos = 2L * hs * 4;          // one head's K + V row
    return kv_bytes_per_pos * cfg->seq_len <= (1L << 20) ? (1 << 30) : kSplitTPos;   // whole head cache <= 1 MiB: never split
}

// positions up to which the 4-wave attention workgroup is used (it covers 64 timesteps per round;
// measured faster than the 16-wave one up to ~250 timesteps at head sizes 48 / 64)
// Below the split-T threshold an 8-wave workgroup per head covers 128 timesteps per round instead of
// the 16-wave kernel's 256: fewer waves to synchronise, measured +0.7 % tokens/s at llama2-7B for
// positions 8..135 (tools/small_attn_sweep.py).  Not where attention is merged with Wo (dim <= 1024).
static bool merge_wanted(const rama_ctx* c, int dim) { return c->tune_merge < 0 ? dim <= 1024 : c->tune_merge != 0; }
static bool small_attn_at(const rama_ctx* c, int pos, bool split, int dim) {
    if (split || c->tune_small_attn == 0) return false;
    if (c->tune_small_attn == 1) return true;
    return !merge_wanted(c, dim) && pos < c->tune_small_pos;
}

// Which launches a step at `pos` consists of (one hipGraph per variant): fast mode 0 = one 16-wave workgroup per head,
// 1 = split-T (long contexts), 2 = fewer waves per head (short contexts); parity mode 0 = 4 waves per head, 1 = 8 (pos >= 256),
// 2 = launches spread over the chip (pos >= tune_spread_pos); bar mode 0 = parity mode's 4 waves per head below bar_pos_eff, from there on the fast
// mode's variants as 1 = split-T, 2 = fewer waves per head, 3 = one 16-wave workgroup per head
static int bar_pos_eff(const rama_ctx* c) { return std::max(0, std::min(std::min(c->tune_bar_pos, c->tune_spread_pos), kLongAttnPos)); }
static int attn_variant(const rama_ctx* c, const rama_config* cfg, int pos) {
    if (c->tune_ref_order && !c->tune_tol && !(c->tune_bar && pos >= bar_pos_eff(c))) return pos >= c->tune_spread_pos ? 2 : (pos >= kLongAttnPos ? 1 : 0);
    const bool split = pos >= split_threshold(c, cfg);
    const int v = split ? 1 : (small_attn_at(c, pos, split, cfg->dim) ? 2 : 0);
    return (c->tune_ref_order && c->tune_bar && v == 0) ? 3 : v;
}
static int apply_attn_variant(rama_ctx* c, const rama_config* cfg, int pos) {
    c->split_attn = pos >= split_threshold(c, cfg);
    c->small_attn = small_attn_at(c, pos, c->split_attn, cfg->dim);
    c->long_attn = pos >= kLongAttnPos;
    c->spread_attn = pos >= c->tune_spread_pos;
    c->bar_fast = c->tune_ref_order && !c->tune_tol && c->tune_bar && pos >= bar_pos_eff(c);
    return c->variant = attn_variant(c, cfg, pos);
}

static int launch_attention(rama_ctx* c, float* xb, float* att, const float* q, const float* kc_layer,
                            const float* vc_layer, const Ctl* ctl, int pos, int dim, int head_size,
                            int seq_len, int n_heads, bool split = false) {
    REQUIRE(head_size % 4 == 0 && head_size >= 4 && head_size <= 256, RAMA_EUNSUP, "attention: head_size must be a multiple of 4 in [4, 256]");
    REQUIRE(aligned16(q) && aligned16(kc_layer) && aligned16(vc_layer) && aligned16(xb) && dim % 4 == 0, RAMA_EINVAL, "attention: buffers must be 16-byte aligned");
    AttnParams p{};
    p.q = q; p.kc = kc_layer; p.vc = vc_layer; p.att = att; p.xb = xb; p.ctl = ctl; p.pos_val = pos;
    p.dim = dim; p.head_size = head_size; p.seq_len = seq_len;
    const int G = head_size <= 64 ? 16 : (head_size <= 128 ? 32 : 64);
    if (split) {
        const int nsplit = attn_nsplit(c, n_heads);
        REQUIRE((size_t)n_heads * nsplit * (head_size + 4) <= c->attn_part_floats, RAMA_EINVAL, "attention: split-T scratch not prepared");
        p.part = c->attn_part; p.nsplit = nsplit; p.att = nullptr;
        const int chunk_max = (seq_len + nsplit - 1) / nsplit;
        dim3 grid(n_heads, nsplit);
        // template dispatch: G lanes per row x W waves per workgroup x cache-load policy
#define RAMA_SPLIT_LAUNCH(G_, W_, NT_, U_) RAMA_LAUNCH(c, (attention_kernel<G_, true, W_, NT_, U_>), grid, dim3(W_ * 64), (size_t)(attn_scratch_floats(G_, W_) + chunk_max) * sizeof(float), p)
#define RAMA_SPLIT_G(W_, NT_, U_) do { if (G == 16) RAMA_SPLIT_LAUNCH(16, W_, NT_, U_); else if (G == 32) RAMA_SPLIT_LAUNCH(32, W_, NT_, U_); else RAMA_SPLIT_LAUNCH(64, W_, NT_, U_); } while (0)
        const int W = c->tune_attn_waves;
        const bool deep = c->tune_attn_u == 16 && W <= 8;          // 16 rows of K and of V per lane in flight: 128 VGPRs, fine at <= 2 waves per SIMD
        if (c->tune_attn_nt) {
            if (W == 4) { if (deep) RAMA_SPLIT_G(4, true, 16); else RAMA_SPLIT_G(4, true, 8); }
            else if (W == 8) { if (deep) RAMA_SPLIT_G(8, true, 16); else RAMA_SPLIT_G(8, true, 8); }
            else RAMA_SPLIT_G(16, true, 8);
        } else {
            if (W == 4) { if (deep) RAMA_SPLIT_G(4, false, 16); else RAMA_SPLIT_G(4, false, 8); }
            else if (W == 8) { if (deep) RAMA_SPLIT_G(8, false, 16); else RAMA_SPLIT_G(8, false, 8); }
            else RAMA_SPLIT_G(16, false, 8);
        }
#undef RAMA_SPLIT_G
#undef RAMA_SPLIT_LAUNCH
        LAUNCHCHK();
        {
            const dim3 cg(n_heads), cb(((head_size + 63) / 64) * 64);
            const float* part = c->attn_part;
            if (c->tune_combine_v == 0) hipLaunchKernelGGL(attention_combine_loop_kernel, cg, cb, 0, c->stream, part, xb, head_size, nsplit);
            else if (nsplit <= 8) hipLaunchKernelGGL(attention_combine_kernel<8>, cg, cb, 0, c->stream, part, xb, head_size, nsplit);
            else if (nsplit <= 16) hipLaunchKernelGGL(attention_combine_kernel<16>, cg, cb, 0, c->stream, part, xb, head_size, nsplit);
            else hipLaunchKernelGGL(attention_combine_kernel<32>, cg, cb, 0, c->stream, part, xb, head_size, nsplit);
        }
        LAUNCHCHK();
        return 0;
    }
    if (c->small_attn) {     // short contexts: fewer waves per head, one round covers the whole context
        const int W = c->tune_small_waves;
        // score buffer: seq_len timesteps, or what 64 KiB hold when the context is longer (this variant only runs below
        // tune_small_pos, and such models split from position 256 on: split_threshold)
        const int cap4 = std::min(seq_len, (int)(64 * 1024 / sizeof(float)) - attn_scratch_floats(G, W));
        REQUIRE((ctl ? c->host_pos : pos) < cap4, RAMA_EUNSUP, "attention: position beyond the single-workgroup kernel's score buffer");
        size_t shm4 = (size_t)(attn_scratch_floats(G, W) + cap4) * sizeof(float);
#define RAMA_SMALL_G(W_) do { if (G == 16) RAMA_LAUNCH(c, (attention_kernel<16, false, W_>), dim3(n_heads), dim3(W_ * 64), shm4, p); \
                              else if (G == 32) RAMA_LAUNCH(c, (attention_kernel<32, false, W_>), dim3(n_heads), dim3(W_ * 64), shm4, p); \
                              else RAMA_LAUNCH(c, (attention_kernel<64, false, W_>), dim3(n_heads), dim3(W_ * 64), shm4, p); } while (0)
        if (W == 4) RAMA_SMALL_G(4); else RAMA_SMALL_G(8);
#undef RAMA_SMALL_G
        LAUNCHCHK();
        return 0;
    }
    // score buffer: seq_len timesteps, or what 64 KiB hold when the context is longer (such models run
    // split-T from position 256 on, split_threshold; a known position beyond the buffer is an error)
    const int cap = std::min(seq_len, (int)(64 * 1024 / sizeof(float)) - attn_scratch_floats(G));
    REQUIRE((ctl ? c->host_pos : pos) < cap, RAMA_EUNSUP, "attention: position beyond the single-workgroup kernel's score buffer");
    size_t shm = (size_t)(attn_scratch_floats(G) + cap) * sizeof(float);
    if (G == 16) RAMA_LAUNCH(c, (attention_kernel<16, false>), dim3(n_heads), dim3(kAttnThreads), shm, p);
    else if (G == 32) RAMA_LAUNCH(c, (attention_kernel<32, false>), dim3(n_heads), dim3(kAttnThreads), shm, p);
    else RAMA_LAUNCH(c, (attention_kernel<64, false>), dim3(n_heads), dim3(kAttnThreads), shm, p);
    LAUNCHCHK();
    return 0;
}

int rama_multi_head_attention(rama_ctx* c, float* xb, float* att, const float* q, const float* key_cache,
                              const float* value_cache, int layer, int dim, int pos, int head_size,
                              int seq_len, int n_heads) {
    RAMA_ENTER(c);
    REQUIRE(c && xb && att && q && key_cache && value_cache, RAMA_EINVAL, "multi_head_attention: NULL argument");
    REQUIRE(pos >= 0 && pos < seq_len && layer >= 0 && n_heads > 0 && n_heads * head_size == dim, RAMA_EINVAL, "multi_head_attention: bad shape");
    RAMA_WRITES(c, xb, dim); RAMA_WRITES(c, att, (size_t)n_heads * seq_len);
    const size_t lo = (size_t)layer * seq_len * dim;   // cpu.rs:28
    if (c->tune_ref_order && c->tune_bar && !c->tune_tol && pos >= bar_pos_eff(c)) {      // bar mode behind its switch: the fast path's single-workgroup kernel (the 1:1 path prepares no split-T scratch)
        c->small_attn = small_attn_at(c, pos, false, dim);
        return launch_attention(c, xb, att, q, key_cache + lo, value_cache + lo, nullptr, pos, dim, head_size, seq_len, n_heads);
    }
    if (c->tune_ref_order && c->tune_chain && attn_chain_ok(head_size, seq_len) && aligned16(q) && aligned16(key_cache + lo) && aligned16(value_cache + lo) && dim % 4 == 0)
        return launch_attention_chain(c, xb, att, q, key_cache + lo, value_cache + lo, nullptr, pos, dim, head_size, seq_len, n_heads, pos >= kLongAttnPos, pos >= c->tune_spread_pos);
    if (c->tune_ref_order) return launch_attention_ref(c, xb, att, q, key_cache + lo, value_cache + lo, nullptr, pos, dim, head_size, seq_len, n_heads);
    c->small_attn = small_attn_at(c, pos, false, dim);
    return launch_attention(c, xb, att, q, key_cache + lo, value_cache + lo, nullptr, pos, dim, head_size, seq_len, n_heads);
}

int rama_sample_argmax(rama_ctx* c, const float* logits, size_t n, int32_t* next_host) {
    RAMA_ENTER(c);
    REQUIRE(c && logits && next_host && n > 0, RAMA_EINVAL, "sample_argmax: bad argument");
    ArgmaxParams ap{};
    ap.logits = logits; ap.n = (int)n; ap.result = c->argmax_result;
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, c->stream, ap);
    LAUNCHCHK();
    HIPCHK(hipMemcpyAsync(c->pinned_int, c->argmax_result, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *next_host = c->pinned_int[0];
    return 0;
}

int rama_sample_topp(rama_ctx* c, const float* logits, size_t n, float temperature, float topp, float u, int32_t* next_host) {
    RAMA_ENTER(c);
    REQUIRE(c && logits && next_host && n > 1, RAMA_EINVAL, "sample_topp: bad argument");
    if (temperature == 0.0f) return rama_sample_argmax(c, logits, n, next_host);
    // Device::sample (cpu.rs:168-178) on the device; only the 4-byte result crosses PCIe (the
    // reference's GPU path downloads all n logits and samples on the host, gpu.rs:149-173)
    int rc = rama_sample_topp_dev(c, logits, n, temperature, topp, u, c->argmax_result);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->pinned_int, c->argmax_result, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    hipMemsetAsync(c->topp_err, 0, sizeof(unsigned), c->stream);
    REQUIRE(c->pinned_int[0] >= 0, RAMA_EINVAL, "sample_topp: no candidate above the cutoff (the reference underflows here)");
    *next_host = c->pinned_int[0];
    return 0;
}

int rama_ref_expf(rama_ctx* c, float* o, const float* x, size_t n) {
    RAMA_ENTER(c);
    REQUIRE(c && (n == 0 || (o && x)), RAMA_EINVAL, "ref_expf: NULL argument");
    if (!n) return 0;
    hipLaunchKernelGGL(expf_glibc_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, o, x, n);
    LAUNCHCHK(); return 0;
}

// ---------------------------------------------------------------- synthetic fill

int rama_fill_synth(rama_ctx* c, float* dst, size_t n, uint64_t seed, uint64_t tag, uint64_t offset, float scale, float bias) {
    RAMA_ENTER(c);
    REQUIRE(c && (n == 0 || dst), RAMA_EINVAL, "fill_synth: NULL argument");
    RAMA_WRITES(c, dst, n);
    if (!n) return 0;
    const uint64_t base = offset + tag * 0x9E3779B97F4A7C15ULL + seed * 0xD1B54A32D192ED03ULL;
    int grid = (int)std::min<size_t>((n + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(fill_synth_kernel, dim3(grid), dim3(256), 0, c->stream, dst, n, base, scale, bias);
    LAUNCHCHK();
    return 0;
}

// ---------------------------------------------------------------- fused decode path

static int check_cfg(const rama_config* cfg) {
    REQUIRE(cfg, RAMA_EINVAL, "config is NULL");
    REQUIRE(cfg->dim > 0 && cfg->hidden_dim > 0 && cfg->n_layers > 0 && cfg->n_heads > 0 && cfg->vocab_size > 0 && cfg->seq_len > 0, RAMA_EINVAL, "config: non-positive field");
    REQUIRE(cfg->n_kv_heads == cfg->n_heads, RAMA_EUNSUP, "config: n_kv_heads != n_heads (the reference indexes the cache with stride dim, infer.rs:31-33)");
    REQUIRE(cfg->dim % cfg->n_heads == 0, RAMA_EINVAL, "config: dim % n_heads != 0");
    REQUIRE(cfg->dim % 4 == 0 && cfg->hidden_dim % 4 == 0, RAMA_EINVAL, "config: dim and hidden_dim must be multiples of 4 (cpu.rs:142-143)");
    const int hs = cfg->dim / cfg->n_heads;
    REQUIRE(hs % 4 == 0 && hs <= 256, RAMA_EUNSUP, "config: head_size must be a multiple of 4, <= 256");
    return 0;
}

// attention + Wo as one launch (attn_wo.hpp) -- only if the WHOLE grid is resident at
// once according to the occupancy API, which is what makes its in-kernel wait deadlock-free.
// Returns 1 if launched, 0 if the caller must use the two separate launches, < 0 / > 0 on error.
static int try_launch_attn_wo(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s,
                              size_t li, float* kc, float* vc, bool* launched) {
    *launched = false;
    const int dim = cfg->dim, hs = dim / cfg->n_heads;
    const int G = hs <= 64 ? 16 : (hs <= 128 ? 32 : 64);
    const int gi = G == 16 ? 0 : (G == 32 ? 1 : 2);
    const int grid = (dim + 3) / 4;
    if (cfg->n_heads > grid || !aligned16(s->q) || !aligned16(s->xb) || !aligned16(kc) || !aligned16(vc)) return 0;
    const size_t lds = (size_t)(kPWaves * 4 + p_attn_lds_floats(G, cfg->seq_len)) * sizeof(float);
    if (lds > 64 * 1024) return 0;
    const void* fn = G == 16 ? (const void*)attn_wo_kernel<16> : (G == 32 ? (const void*)attn_wo_kernel<32> : (const void*)attn_wo_kernel<64>);
    if (c->merge_blocks_per_cu[gi] < 0 || c->merge_lds[gi] != lds) {
        int nb = 0;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, kPThreads, lds));
        c->merge_blocks_per_cu[gi] = nb; c->merge_lds[gi] = lds;
    }
    // The occupancy API over-reports by one block per CU only for kernels with > 80 SGPRs
    // (MI355X_MICROARCH.md, Residency); this kernel uses 58 (build remark), so its answer is
    // taken as is.  Should the grid not fit, the two separate launches are used instead.
    if ((long)c->merge_blocks_per_cu[gi] * c->cu_count < grid) return 0;
    AttnWoParams a{};
    a.dim = dim; a.n_heads = cfg->n_heads; a.seq_len = cfg->seq_len;
    a.q = s->q; a.kc = kc; a.vc = vc; a.xb = s->xb; a.x = s->x; a.wo = w->wo + li * (size_t)dim * dim;
    a.ctl = c->ctl; a.counter = c->attn_counter; a.err = c->pbar + 1;
    if (G == 16) hipLaunchKernelGGL(attn_wo_kernel<16>, dim3(grid), dim3(kPThreads), lds, c->stream, a);
    else if (G == 32) hipLaunchKernelGGL(attn_wo_kernel<32>, dim3(grid), dim3(kPThreads), lds, c->stream, a);
    else hipLaunchKernelGGL(attn_wo_kernel<64>, dim3(grid), dim3(kPThreads), lds, c->stream, a);
    LAUNCHCHK();
    *launched = true;
    c->handoff_dirty = true;
    return 0;
}

// a whole stage as two launches (layer_fused.hpp): embedding + counter reset, then every layer and the classifier chained by
// arrival counters.  *launched = false: the shape is not one it takes, the caller enqueues the separate launches.
static bool fused_wanted(const rama_ctx* c, int dim) { return c->tune_fused < 0 ? dim <= 1024 : c->tune_fused != 0; }
static int try_launch_fused(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s, const rama_stage* st, bool* launched) {
    *launched = false;
    const int dim = cfg->dim, hidden = cfg->hidden_dim, H = cfg->n_heads, hs = dim / H, V = cfg->vocab_size;
    const int nl = st->layer_end - st->layer_begin;
    if (nl <= 0 && (!st->do_cls || st->do_embed)) return 0;
    if (dim > kFusedMaxDim || hidden > kFusedMaxHidden || nl > kFusedMaxLayers || dim % 4 || hidden % 4 || hs % 4 || hs > 256 || cfg->seq_len % 4) return 0;
    const int G = hs <= 64 ? 16 : (hs <= 128 ? 32 : 64);
    size_t lds = (size_t)fused_lds_floats(G, cfg->seq_len, dim, hidden) * sizeof(float);
    if (lds > 64 * 1024) return 0;
    if (c->tune_fused_solo < 0 ? dim > 512 : c->tune_fused_solo != 0) lds = kFusedSoloLds;
    if ((double)hidden * dim * 4.0 >= 2147483648.0 || (double)V * dim * 4.0 >= 2147483648.0 || (double)cfg->seq_len * dim * 4.0 >= 2147483648.0) return 0;
    if (!aligned16(s->x) || !aligned16(s->q) || !aligned16(s->xb) || !aligned16(s->hb) || !aligned16(s->key_cache) || !aligned16(s->value_cache) ||
        !aligned16(w->wq) || !aligned16(w->wk) || !aligned16(w->wv) || !aligned16(w->wo) || !aligned16(w->w1) || !aligned16(w->w2) || !aligned16(w->w3) ||
        !aligned16(w->wcls) || !aligned16(w->rms_att_weight) || !aligned16(w->rms_ffn_weight) || !aligned16(w->rms_final_weight)) return 0;
    auto wgs = [](int units) { return (units + kPWaves - 1) / kPWaves; };
    FusedParams a{};
    a.dim = dim; a.hidden = hidden; a.n_heads = H; a.seq_len = cfg->seq_len; a.vocab = V; a.n_layers = nl; a.do_cls = st->do_cls ? 1 : 0;
    a.wq = w->wq; a.wk = w->wk; a.wv = w->wv; a.wo = w->wo; a.w1 = w->w1; a.w3 = w->w3; a.w2 = w->w2;
    a.g_att = w->rms_att_weight; a.g_ffn = w->rms_ffn_weight; a.g_final = w->rms_final_weight; a.wcls = w->wcls;
    a.emb = st->do_embed ? w->token_embedding_table : nullptr;
    a.x = s->x; a.q = s->q; a.k = s->k; a.v = s->v; a.xb = s->xb; a.hb = s->hb; a.logits = s->logits;
    a.kc = s->key_cache; a.vc = s->value_cache; a.fr = w->freq_cis_real; a.fi = w->freq_cis_imag;
    a.ctl = c->ctl; a.hand = c->fused_hand; a.epoch = c->fused_epoch; a.err = c->pbar + 1;
    const bool wide = hidden > 1024;             // W2 units: 2 rows x 8 chunks ahead instead of 4 x 4
    a.nA = wgs(3 * ((dim + 3) / 4)); a.nC = wgs((dim + 3) / 4); a.nD = wgs((hidden + 1) / 2); a.nE = wide ? wgs((dim + 1) / 2) : a.nC;
    const bool cd2 = dim <= 512;
    const long grid = (long)nl * (a.nA + H + a.nC + a.nD + a.nE) + (st->do_cls ? (cd2 ? wgs((V + 7) / 8) : wgs((V + 3) / 4)) : 0);
#define RAMA_FUSED_(G_, CD_) do { if (wide) hipLaunchKernelGGL((stage_fused_kernel<G_, CD_, 2, 8>), dim3((unsigned)grid), dim3(kPThreads), lds, c->stream, a); \
                                  else hipLaunchKernelGGL((stage_fused_kernel<G_, CD_, 4, 4>), dim3((unsigned)grid), dim3(kPThreads), lds, c->stream, a); } while (0)
#define RAMA_FUSED(G_, CD_) RAMA_FUSED_(G_, CD_)
    if (G == 16) { if (cd2) RAMA_FUSED(16, 2); else RAMA_FUSED(16, 4); }
    else if (G == 32) { if (cd2) RAMA_FUSED(32, 2); else RAMA_FUSED(32, 4); }
    else { if (cd2) RAMA_FUSED(64, 2); else RAMA_FUSED(64, 4); }
#undef RAMA_FUSED_
#undef RAMA_FUSED
    LAUNCHCHK();
    *launched = true;
    c->handoff_dirty = true;
    if (c->fused_chained) { c->fused_epoch_owed = true; return 0; }      // the sampler that follows advances the epoch
    hipLaunchKernelGGL(fused_epoch_kernel, dim3(1), dim3(1), 0, c->stream, c->fused_epoch);
    LAUNCHCHK();
    return 0;
}

// the fast path's norm-folding launches of a layer (also what tolerance mode's "tol_mask" swaps in for A/B runs)
// infer.rs:19-33: rmsnorm, Wq|Wk|Wv, RoPE, cache append
static int launch_fast_qkv(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s, size_t li, float* kc, float* vc) {
    const int dim = cfg->dim, hs = dim / cfg->n_heads;
    const size_t dd = (size_t)dim * dim;
    KTimer kt(c, RAMA_K_QKV);
    GemvParams p{};
    p.w[0] = w->wq + li * dd; p.w[1] = w->wk + li * dd; p.w[2] = w->wv + li * dd;
    p.x = s->x; p.nw = w->rms_att_weight + li * dim;
    p.o[0] = s->q; p.o[1] = s->k; p.o[2] = s->v;
    p.K = dim; p.rows = dim; p.nmat = 3;
    p.ctl = c->ctl; p.fr = w->freq_cis_real; p.fi = w->freq_cis_imag; p.head_size = hs;
    p.kc = kc; p.vc = vc;
    p.zero_me = c->attn_counter;
    if (use_solo(c, dim)) {
        const dim3 grid((3 * ((dim + 3) / 4) + kSoloWaves - 1) / kSoloWaves), block(kSoloWaves * 64);
        if (dim <= 512) RAMA_LAUNCH(c, (gemv_rows_solo<4, 2, true, EPI_QKV>), grid, block, 0, p);
        else RAMA_LAUNCH(c, (gemv_rows_solo<4, 4, true, EPI_QKV>), grid, block, 0, p);
    } else {
        DISPATCH_GEOM(c, RAMA_LAUNCH(c, (gemv_rows<R_, CH_, NW_, true, EPI_QKV>), dim3(3 * (dim / R_)), dim3(NW_ * 64), 0, p));
    }
    LAUNCHCHK();
    return 0;
}
// infer.rs:39-45: rmsnorm, W1|W3, SiLU * gate (w13i: the model's row-interleaved copy, or NULL)
static int launch_fast_w13(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s, size_t li, const float* w13i) {
    const int dim = cfg->dim, hidden = cfg->hidden_dim;
    const size_t hd = (size_t)hidden * dim;
    KTimer kt(c, RAMA_K_W13);
    if (w13i && c->tune_w13i) {
        // the model's row-interleaved copy: one [2 hidden, dim] matrix, rows (2i, 2i + 1) = (W1 row i, W3 row i)
        GemvParams p{};
        p.w[0] = w13i + li * 2 * hd; p.x = s->x; p.nw = w->rms_ffn_weight + li * dim; p.o[0] = s->hb;
        p.K = dim; p.rows = 2 * hidden; p.nmat = 1;
        if (use_solo(c, dim)) {
            const dim3 grid(((2 * hidden + 3) / 4 + kSoloWaves - 1) / kSoloWaves), block(kSoloWaves * 64);
            if (dim <= 512) RAMA_LAUNCH(c, (gemv_rows_solo<4, 2, true, EPI_SWIGLU_PAIR>), grid, block, 0, p);
            else RAMA_LAUNCH(c, (gemv_rows_solo<4, 4, true, EPI_SWIGLU_PAIR>), grid, block, 0, p);
        } else {
            RAMA_LAUNCH(c, (gemv_rows<4, 2, 8, true, EPI_SWIGLU_PAIR>), dim3((2 * hidden + 3) / 4), dim3(8 * 64), 0, p);
        }
    } else {
        SwigluParams p{};
        p.w1 = w->w1 + li * hd; p.w3 = w->w3 + li * hd; p.x = s->x; p.nw = w->rms_ffn_weight + li * dim;
        p.hb = s->hb; p.K = dim; p.rows = hidden;
        if (use_solo(c, dim)) {
            const dim3 grid(((hidden + 1) / 2 + kSoloWaves - 1) / kSoloWaves), block(kSoloWaves * 64);
            if (dim <= 512) RAMA_LAUNCH(c, (gemv_swiglu_solo<2, 2>), grid, block, 0, p);
            else RAMA_LAUNCH(c, (gemv_swiglu_solo<2, 4>), grid, block, 0, p);
        } else {
            DISPATCH_GEOM(c, RAMA_LAUNCH(c, (gemv_swiglu<R2_, CH_, NW_>), dim3((hidden + R2_ - 1) / R2_), dim3(NW_ * 64), 0, p));
        }
    }
    LAUNCHCHK();
    return 0;
}

// infer.rs:8-53 op by op in the reference's rounding order (ref_order.hpp); leaves EVERY RunState
// buffer as the CPU path does (x, xb, xb2, hb, hb2, q, k, v, att, logits, caches)
static int enqueue_stage_ref(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s, const rama_stage* st) {
    const int dim = cfg->dim, hidden = cfg->hidden_dim, hs = dim / cfg->n_heads;
    const size_t dd = (size_t)dim * dim, hd = (size_t)hidden * dim;
    int rc;
    REQUIRE(s->xb2 && s->hb2 && s->k && s->v, RAMA_EINVAL, "forward (reference order): xb2 / hb2 / k / v buffers are required");
    if (st->do_embed) {
        hipLaunchKernelGGL(embed_kernel, dim3((dim + 255) / 256), dim3(256), 0, c->stream, s->x, w->token_embedding_table, (const Ctl*)c->ctl, 0, dim);
        LAUNCHCHK();
    }
    for (int layer = st->layer_begin; layer < st->layer_end; layer++) {
        const size_t li = (size_t)(layer - st->layer_begin);
        float* kc = s->key_cache + li * cfg->seq_len * dim;
        float* vc = s->value_cache + li * cfg->seq_len * dim;
        rc = launch_rmsnorm_ref(c, s->xb, s->x, w->rms_att_weight + li * dim, dim); if (rc) return rc;          // infer.rs:19
        {   // :20-23
            float* const oo[3] = {s->q, s->k, s->v};
            const float* const ww[3] = {w->wq + li * dd, w->wk + li * dd, w->wv + li * dd};
            rc = launch_matvec_ref(c, 3, oo, ww, s->xb, dim, dim); if (rc) return rc;
        }
        hipLaunchKernelGGL(rope_ref_cursor_kernel, dim3((dim / 2 + 255) / 256), dim3(256), 0, c->stream, s->q, s->k, (const float*)s->v,
                           w->freq_cis_real, w->freq_cis_imag, dim, hs, kc, vc, (const Ctl*)c->ctl);                                // :25-33
        LAUNCHCHK();
        rc = launch_attention_ref(c, s->xb, s->att, s->q, kc, vc, c->ctl, 0, dim, hs, cfg->seq_len, cfg->n_heads); if (rc) return rc;   // :34
        rc = launch_matvec_ref1(c, s->xb2, w->wo + li * dd, s->xb, dim, dim); if (rc) return rc;                  // :35
        hipLaunchKernelGGL(array_add_kernel, dim3(ew_grid(dim)), dim3(256), 0, c->stream, s->x, (const float*)s->xb2, (size_t)dim);   // :37
        LAUNCHCHK();
        rc = launch_rmsnorm_ref(c, s->xb, s->x, w->rms_ffn_weight + li * dim, dim); if (rc) return rc;            // :39
        {   // :41-42
            float* const oo[3] = {s->hb, s->hb2, nullptr};
            const float* const ww[3] = {w->w1 + li * hd, w->w3 + li * hd, nullptr};
            rc = launch_matvec_ref(c, 2, oo, ww, s->xb, dim, hidden); if (rc) return rc;
        }
        hipLaunchKernelGGL(sinu_ref_kernel, dim3(ew_grid(hidden)), dim3(256), 0, c->stream, s->hb, (size_t)hidden);                  // :44
        LAUNCHCHK();
        hipLaunchKernelGGL(array_mult_kernel, dim3(ew_grid(hidden)), dim3(256), 0, c->stream, s->hb, (const float*)s->hb2, (size_t)hidden);   // :45
        LAUNCHCHK();
        rc = launch_matvec_ref1(c, s->xb, w->w2 + li * hd, s->hb, hidden, dim); if (rc) return rc;                // :46
        hipLaunchKernelGGL(array_add_kernel, dim3(ew_grid(dim)), dim3(256), 0, c->stream, s->x, (const float*)s->xb, (size_t)dim);    // :47
        LAUNCHCHK();
    }
    if (st->do_cls) {
        hipLaunchKernelGGL(copy_kernel, dim3(ew_grid(dim)), dim3(256), 0, c->stream, s->xb, (const float*)s->x, (size_t)dim);         // :49
        LAUNCHCHK();
        rc = launch_rmsnorm_ref(c, s->x, s->xb, w->rms_final_weight, dim); if (rc) return rc;                     // :50
        rc = launch_matvec_ref1(c, s->logits, w->wcls, s->x, dim, cfg->vocab_size); if (rc) return rc;            // :51
    }
    return 0;
}

// the same (every RunState buffer as the CPU path leaves it, bit for bit) on the model's chain-order weight
// copies: 7 launches per layer.  Returns false when a copy is missing (weights uploaded tensor by tensor,
// widths that are not whole 16-float blocks, no memory for the copy): the caller takes enqueue_stage_ref.
static int enqueue_stage_chain(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s, const rama_stage* st, bool* done) {
    *done = false;
    const int dim = cfg->dim, hidden = cfg->hidden_dim, hs = dim / cfg->n_heads, V = cfg->vocab_size;
    if (!c->tune_chain || dim % 16 || hidden % 16 || dim > 16000 || hidden > 16000 || !attn_chain_ok(hs, cfg->seq_len)) return 0;
    const bool layers = st->layer_end > st->layer_begin;
    const float *cq = nullptr, *ck = nullptr, *cv = nullptr, *co = nullptr, *c13 = nullptr, *c2 = nullptr, *ccls = nullptr;
    if (layers) {
        cq = rama_internal_chain_lookup(w->wq, dim, dim); ck = rama_internal_chain_lookup(w->wk, dim, dim);
        cv = rama_internal_chain_lookup(w->wv, dim, dim); co = rama_internal_chain_lookup(w->wo, dim, dim);
        c13 = rama_internal_chain_lookup(w->w1, 2 * hidden, dim); c2 = rama_internal_chain_lookup(w->w2, dim, hidden);
        if (!cq || !ck || !cv || !co || !c13 || !c2) return 0;
    }
    if (st->do_cls) { ccls = rama_internal_chain_lookup(w->wcls, V, dim); if (!ccls) return 0; }
    REQUIRE(s->xb2 && s->hb2 && s->k && s->v, RAMA_EINVAL, "forward (reference order): xb2 / hb2 / k / v buffers are required");
    *done = true;
    const size_t dd = (size_t)dim * dim, hd = (size_t)hidden * dim;                 // chain-order copies keep the row-major sizes (rows are multiples of 16)
    // tolerance mode ("ref_order" = 2): the matvecs as in parity mode, the norms folded into them as tree-shaped sums (dim <= 8192: all of x
    // in a workgroup's registers; wider models keep the exact-sum norm launches), the fast path's attention
    // "tol_mask" (A/B runs that say which op carries how much of the distance to the CPU path): 1 / 2 / 4 / 8 / 16 = the fast path's
    // Wq|Wk|Wv / Wo / W1|W3 / W2 / classifier launch instead of the chain-order one, 32 = parity mode's attention, 64 = its exact-sum norm launches
    const bool tol = c->tune_tol != 0;
    const int mask = tol ? c->tune_tol_mask : 0;
    const bool tol_fold = tol && dim <= 8192 && c->tune_chain_d <= 0 && !(mask & 64);
    // parity mode, narrow models: the exact norms ride in the matvecs that consume them ("chain_norm"; measured: stories15M +6.6 %; at dim
    // 768 the ripples cost more than the launch, -4 %)
    const int par_norm = (!tol && c->tune_chain_d <= 0 && c->tune_chain_norm && dim <= 512) ? CNORM_EXACT : CNORM_NONE;
    // parity mode, wider models ([r5]): the exact sum by a leader workgroup inside the consuming launch ("chain_lead"; x of <= 4096 floats per wave of the
    // matvec's workgroups, layer ranges of <= 256 layers)
    const bool lead_ok = !tol && par_norm == CNORM_NONE && c->tune_chain_lead && c->tune_chain_d <= 0 && dim % 8 == 0 && dim <= 4096 && dim / 16 > 32 &&
                         st->layer_end - st->layer_begin <= 256;      // (per-class timing keeps it: the `norm` class is then the final norm's launch alone, as in the timed step)
    const int lnorm = tol ? (tol_fold ? CNORM_TREE : CNORM_NONE) : (lead_ok ? CNORM_LEAD : par_norm);      // how the layer norms are folded
    bool led = false;
    const float* w13i = (tol && (mask & 4) && st->layer_end > st->layer_begin && (double)hidden * dim * 8.0 < 2147483648.0) ? rama_internal_w13_lookup(w->w1, w->w3) : nullptr;
    int rc;
    if (st->do_embed) {
        hipLaunchKernelGGL(embed_kernel, dim3((dim + 255) / 256), dim3(256), 0, c->stream, s->x, w->token_embedding_table, (const Ctl*)c->ctl, 0, dim);
        LAUNCHCHK();
    }
    for (int layer = st->layer_begin; layer < st->layer_end; layer++) {
        const size_t li = (size_t)(layer - st->layer_begin);
        float* kc = s->key_cache + li * cfg->seq_len * dim;
        float* vc = s->value_cache + li * cfg->seq_len * dim;
        // narrow models: the two norms of a layer ride in the matvecs that consume them (2 of 7 launches; "chain_norm")
        const bool fold = lnorm != CNORM_NONE;
        if (!fold && !(mask & 1)) { KTimer kt(c, RAMA_K_NORM); rc = launch_rmsnorm_chain(c, s->xb, s->x, w->rms_att_weight + li * dim, dim, nullptr); if (rc) return rc; }      // infer.rs:19
        if (mask & 1) { rc = launch_fast_qkv(c, cfg, w, s, li, kc, vc); if (rc) return rc; }
        else {   // :20-33: Wq | Wk | Wv, RoPE, cache append
            KTimer kt(c, RAMA_K_QKV);
            ChainParams p{};
            p.w[0] = cq + li * dd; p.w[1] = ck + li * dd; p.w[2] = cv + li * dd;
            p.o[0] = s->q; p.o[1] = s->k; p.o[2] = s->v; p.x = fold ? s->x : s->xb; p.nw = fold ? w->rms_att_weight + li * dim : nullptr;
            p.K = dim; p.rows = dim; p.nmat = 3;
            p.ctl = c->ctl; p.fr = w->freq_cis_real; p.fi = w->freq_cis_imag; p.head_size = hs; p.kc = kc; p.vc = vc;
            if (lnorm == CNORM_LEAD) { p.lead = c->lead_slots + 32 * (2 * li); p.epoch = c->fused_epoch; p.err = c->pbar + 1; led = true; }
            rc = launch_chain<CEPI_QKV>(c, p, fold ? lnorm : CNORM_NONE); if (rc) return rc;
        }
        {   // :34
            KTimer kt(c, RAMA_K_ATTN);
            if ((tol && !(mask & 32)) || c->bar_fast) rc = launch_attention(c, s->xb, s->att, s->q, kc, vc, c->ctl, 0, dim, hs, cfg->seq_len, cfg->n_heads, c->split_attn);
            else rc = launch_attention_chain(c, s->xb, s->att, s->q, kc, vc, c->ctl, 0, dim, hs, cfg->seq_len, cfg->n_heads, c->long_attn, c->spread_attn);
            if (rc) return rc;
        }
        if (mask & 2) { KTimer kt(c, RAMA_K_WO); rc = launch_rows<false, EPI_RESID>(c, s->x, w->wo + li * dd, s->xb, nullptr, dim, dim); if (rc) return rc; }
        else {   // :35-37: xb2 = Wo . xb; x += xb2
            KTimer kt(c, RAMA_K_WO);
            ChainParams p{};
            p.w[0] = co + li * dd; p.o[0] = s->xb2; p.resid = s->x; p.x = s->xb; p.K = dim; p.rows = dim; p.nmat = 1;
            rc = launch_chain<CEPI_RESID>(c, p); if (rc) return rc;
        }
        if (!fold && !(mask & 4)) { KTimer kt(c, RAMA_K_NORM); rc = launch_rmsnorm_chain(c, s->xb, s->x, w->rms_ffn_weight + li * dim, dim, nullptr); if (rc) return rc; }       // :39
        if (mask & 4) { rc = launch_fast_w13(c, cfg, w, s, li, w13i); if (rc) return rc; }
        else {   // :41-45: hb = silu(W1 . xb) * (hb2 = W3 . xb)
            KTimer kt(c, RAMA_K_W13);
            ChainParams p{};
            p.w[0] = c13 + li * 2 * hd; p.o[0] = s->hb; p.o[1] = s->hb2; p.x = fold ? s->x : s->xb; p.nw = fold ? w->rms_ffn_weight + li * dim : nullptr;
            p.K = dim; p.rows = 2 * hidden; p.nmat = 1;
            if (lnorm == CNORM_LEAD) { p.lead = c->lead_slots + 32 * (2 * li + 1); p.epoch = c->fused_epoch; p.err = c->pbar + 1; led = true; }
            rc = launch_chain<CEPI_SWIGLU>(c, p, fold ? lnorm : CNORM_NONE); if (rc) return rc;
        }
        if (mask & 8) { KTimer kt(c, RAMA_K_W2); rc = launch_rows<false, EPI_RESID>(c, s->x, w->w2 + li * hd, s->hb, nullptr, hidden, dim); if (rc) return rc; }
        else {   // :46-47: xb = W2 . hb; x += xb
            KTimer kt(c, RAMA_K_W2);
            ChainParams p{};
            p.w[0] = c2 + li * hd; p.o[0] = s->xb; p.resid = s->x; p.x = s->hb; p.K = hidden; p.rows = dim; p.nmat = 1;
            rc = launch_chain<CEPI_RESID>(c, p); if (rc) return rc;
        }
    }
    if (st->do_cls) {   // :49-51: xb = x; x = rmsnorm(xb); logits = Wcls . x
        // (tolerance mode with the norm folded into the classifier leaves x and xb as the fast path does: the residual stream,
        // not its normalised copy)
        if (mask & 16) { KTimer kt(c, RAMA_K_CLS); return launch_rows<true, EPI_STORE>(c, s->logits, w->wcls, s->x, w->rms_final_weight, dim, V); }
        if (!tol_fold) { KTimer kt(c, RAMA_K_NORM); rc = launch_rmsnorm_chain(c, s->x, s->x, w->rms_final_weight, dim, s->xb); if (rc) return rc; }
        KTimer kt(c, RAMA_K_CLS);
        ChainParams p{};
        p.w[0] = ccls; p.o[0] = s->logits; p.x = s->x; p.K = dim; p.rows = V; p.nmat = 1;
        if (tol_fold) p.nw = w->rms_final_weight;
        rc = launch_chain<CEPI_STORE>(c, p, tol_fold ? CNORM_TREE : CNORM_NONE); if (rc) return rc;
    }
    if (led) {      // the leaders' words carry the epoch: it advances after every stage that used them (layer_fused.hpp's convention)
        if (c->fused_chained) c->fused_epoch_owed = true;      // the sampler that follows advances it
        else { hipLaunchKernelGGL(fused_epoch_kernel, dim3(1), dim3(1), 0, c->stream, c->fused_epoch); LAUNCHCHK(); }
    }
    return 0;
}

// one (token, pos) step over a layer range; ctl on the device holds token/pos
static int enqueue_stage(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s,
                         const rama_stage* st) {
    if (c->tune_ref_order) {
        bool done = false;
        const int rc = enqueue_stage_chain(c, cfg, w, s, st, &done);
        return done ? rc : (rc ? rc : enqueue_stage_ref(c, cfg, w, s, st));
    }
    const int dim = cfg->dim, hidden = cfg->hidden_dim, hs = dim / cfg->n_heads;
    const size_t dd = (size_t)dim * dim, hd = (size_t)hidden * dim;
    if (fused_wanted(c, dim) && !c->split_attn && !c->small_attn && c->kp.kernel_id < 0) {
        bool launched = false;
        const int rc = try_launch_fused(c, cfg, w, s, st, &launched);
        if (rc || launched) return rc;
    }
    const float* w13i = (st->layer_end > st->layer_begin && (double)hidden * dim * 8.0 < 2147483648.0) ? rama_internal_w13_lookup(w->w1, w->w3) : nullptr;
    if (st->do_embed) {
        hipLaunchKernelGGL(embed_kernel, dim3((dim + 255) / 256), dim3(256), 0, c->stream, s->x, w->token_embedding_table, (const Ctl*)c->ctl, 0, dim);
        LAUNCHCHK();
    }
    for (int layer = st->layer_begin; layer < st->layer_end; layer++) {
        const size_t li = (size_t)(layer - st->layer_begin);
        float* kc = s->key_cache + li * cfg->seq_len * dim;
        float* vc = s->value_cache + li * cfg->seq_len * dim;
        { const int rcq = launch_fast_qkv(c, cfg, w, s, li, kc, vc); if (rcq) return rcq; }      // infer.rs:19-33
        bool merged = false;
        const bool want_merge = merge_wanted(c, dim);
        if (want_merge && !c->split_attn && !c->small_attn && c->kp.kernel_id < 0) {   // infer.rs:34-37 as one launch (per-kernel timing keeps them apart)
            int rc = try_launch_attn_wo(c, cfg, w, s, li, kc, vc, &merged);
            if (rc) return rc;
        }
        if (!merged) {
            {   // infer.rs:34
                KTimer kt(c, RAMA_K_ATTN);
                int rc = launch_attention(c, s->xb, s->att, s->q, kc, vc, c->ctl, 0, dim, hs, cfg->seq_len, cfg->n_heads, c->split_attn);
                if (rc) return rc;
            }
            {   // infer.rs:35-37: x += Wo . xb
                KTimer kt(c, RAMA_K_WO);
                int rc = launch_rows<false, EPI_RESID>(c, s->x, w->wo + li * dd, s->xb, nullptr, dim, dim);
                if (rc) return rc;
            }
        }
        { const int rcw = launch_fast_w13(c, cfg, w, s, li, w13i); if (rcw) return rcw; }        // infer.rs:39-45
        {   // infer.rs:46-47: x += W2 . hb
            KTimer kt(c, RAMA_K_W2);
            int rc = launch_rows<false, EPI_RESID>(c, s->x, w->w2 + li * hd, s->hb, nullptr, hidden, dim);
            if (rc) return rc;
        }
    }
    if (st->do_cls) {   // infer.rs:49-51
        KTimer kt(c, RAMA_K_CLS);
        int rc = launch_rows<true, EPI_STORE>(c, s->logits, w->wcls, s->x, w->rms_final_weight, dim, cfg->vocab_size);
        if (rc) return rc;
    }
    return 0;
}

static bool same_capture(const GraphCache& g, const rama_config* cfg, const rama_weights* w, const rama_run_state* s);

// enqueue_stage, replayed from a hipGraph when graph mode is on: token and position live in the device
// cursor (written by the caller just before), so the launches of a (state, stage, attention variant)
// are the same every time.  Used by rama_forward / rama_forward_stage* -- the per-token entry points
// of a trait-level host and of the pipeline stages (csrc/pipe.hip).
constexpr size_t kMaxStageGraphs = 48;
static int run_stage(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s, const rama_stage* st) {
    if (!c->graph_mode || c->kp.kernel_id >= 0) return enqueue_stage(c, cfg, w, s, st);
    const int variant = c->variant;
    rama_ctx::StageGraph* hit = nullptr;
    for (auto& e : c->sg)
        if (e.variant == variant && !memcmp(&e.st, st, sizeof *st) && same_capture(e.g, cfg, w, s)) { hit = &e; break; }
    if (!hit) {
        if (c->sg.size() >= kMaxStageGraphs) {       // evict the entry used longest ago
            size_t old = 0;
            for (size_t i = 1; i < c->sg.size(); i++) if (c->sg[i].used < c->sg[old].used) old = i;
            HIPCHK(hipStreamSynchronize(c->stream));
            if (c->sg[old].g.exec) hipGraphExecDestroy(c->sg[old].g.exec);
            if (c->sg[old].g.graph) hipGraphDestroy(c->sg[old].g.graph);
            c->sg.erase(c->sg.begin() + (long)old);
        }
        rama_ctx::StageGraph e;
        const bool dirty_before = c->handoff_dirty;
        c->handoff_dirty = false;
        HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue_stage(c, cfg, w, s, st);
        const hipError_t err = hipStreamEndCapture(c->stream, &e.g.graph);
        e.g.handoff = c->handoff_dirty;
        c->handoff_dirty = dirty_before;
        if (rc) { if (e.g.graph) hipGraphDestroy(e.g.graph); return rc; }
        HIPCHK(err);
        HIPCHK(hipGraphInstantiate(&e.g.exec, e.g.graph, nullptr, nullptr, 0));
        e.g.cfg = *cfg; e.g.w = *w; e.g.s = *s; e.g.valid = true; e.g.copies_gen = rama_internal_copies_generation(); e.st = *st; e.variant = variant;
        c->sg.push_back(e);
        hit = &c->sg.back();
    }
    hit->used = ++c->sg_clock;
    HIPCHK(hipGraphLaunch(hit->g.exec, c->stream));
    if (hit->g.handoff) c->handoff_dirty = true;
    return 0;
}

static int check_stage(const rama_config* cfg, const rama_weights* w, const rama_run_state* s, const rama_stage* st) {
    REQUIRE(w && s && st, RAMA_EINVAL, "forward: NULL argument");
    REQUIRE(st->layer_begin >= 0 && st->layer_begin <= st->layer_end && st->layer_end <= cfg->n_layers, RAMA_EINVAL, "forward: bad layer range");
    REQUIRE(cfg->dim % 4 == 0, RAMA_EINVAL, "forward: dim % 4 != 0");
    if (st->layer_end > st->layer_begin)
        REQUIRE(w->wq && w->wk && w->wv && w->wo && w->w1 && w->w2 && w->w3 && w->rms_att_weight && w->rms_ffn_weight && w->freq_cis_real && w->freq_cis_imag, RAMA_EINVAL, "forward: missing layer weights");
    if (st->do_embed) REQUIRE(w->token_embedding_table, RAMA_EINVAL, "forward: missing embedding table");
    if (st->do_cls) REQUIRE(w->wcls && w->rms_final_weight && s->logits, RAMA_EINVAL, "forward: missing classifier weights");
    REQUIRE(s->x && s->xb && s->hb && s->q && s->k && s->v && s->att && s->key_cache && s->value_cache, RAMA_EINVAL, "forward: missing state buffer");
    return 0;
}

// parity mode's chain-order weight copy: made on first use for a resident model -- and for weights that were uploaded tensor by tensor (hbm.rs:55-90),
// which are adopted first (model.hip rama_internal_adopt): the reference's own upload path then runs the same kernels as rama_model_load
static int ensure_chain_copy(rama_ctx* c, const rama_config* cfg, const rama_weights* w, const rama_stage* st) {
    const rama_stage full{0, cfg->n_layers, 1, 1};
    const int rc = rama_internal_adopt(c, cfg, st ? st : &full, w);
    return rc ? rc : rama_internal_model_ensure(c, w, 1);
}

int rama_forward_stage(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s,
                       int token, int pos, const rama_stage* st) {
    RAMA_ENTER(c);
    REQUIRE(c, RAMA_EINVAL, "ctx is NULL");
    if (set_device(c)) return 1;
    int rc = check_cfg(cfg); if (rc) return rc;
    rc = check_stage(cfg, w, s, st); if (rc) return rc;
    REQUIRE(pos >= 0 && pos < cfg->seq_len, RAMA_EINVAL, "forward: pos outside [0, seq_len)");
    REQUIRE(token >= 0 && token < cfg->vocab_size, RAMA_EINVAL, "forward: token outside the vocabulary");
    rc = ensure_attn_part(c, cfg); if (rc) return rc;
    if (c->tune_ref_order && c->tune_chain) { rc = ensure_chain_copy(c, cfg, w, st); if (rc) return rc; }
    hipLaunchKernelGGL(set_ctl_kernel, dim3(1), dim3(1), 0, c->stream, c->ctl, token, pos, 0, 0);
    LAUNCHCHK();
    c->embedded_x = nullptr;
    c->host_pos = -1;
    apply_attn_variant(c, cfg, pos);
    return run_stage(c, cfg, w, s, st);
}

// pipeline-stage variants: the token id stays in device memory end to end
int rama_forward_stage_devtok(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s,
                              const int32_t* token_dev, int pos, const rama_stage* st) {
    RAMA_ENTER(c);
    REQUIRE(c, RAMA_EINVAL, "ctx is NULL");
    if (set_device(c)) return 1;
    int rc = check_cfg(cfg); if (rc) return rc;
    rc = check_stage(cfg, w, s, st); if (rc) return rc;
    REQUIRE(pos >= 0 && pos < cfg->seq_len, RAMA_EINVAL, "forward: pos outside [0, seq_len)");
    REQUIRE(token_dev || !st->do_embed, RAMA_EINVAL, "forward: an embedding stage needs a token");
    rc = ensure_attn_part(c, cfg); if (rc) return rc;
    if (c->tune_ref_order && c->tune_chain) { rc = ensure_chain_copy(c, cfg, w, st); if (rc) return rc; }
    hipLaunchKernelGGL(set_ctl_dev_kernel, dim3(1), dim3(1), 0, c->stream, c->ctl, (const int*)token_dev, pos, cfg->vocab_size);
    LAUNCHCHK();
    c->embedded_x = nullptr;
    c->host_pos = -1;
    apply_attn_variant(c, cfg, pos);
    return run_stage(c, cfg, w, s, st);
}

// ---- device top-p sampler (topp_sort.hpp: block sorts + ranks + the exact running sum; every kernel hand-written)

// topp_pick_dist_kernel's hand-off words in one allocation: items | hdr | cross | epoch | bad
constexpr size_t kToppDistItems = sizeof(PickItem) * kPickChunk * kPickMaxChunks;
constexpr size_t kToppDistBytes = kToppDistItems + 8 * kPickMaxChunks + 16 + 16;
static ToppDistParams topp_dist_params(rama_ctx* c) {
    char* b = (char*)c->topp_dist;
    ToppDistParams d{};
    d.approx = c->topp_approx;
    d.items = (PickItem*)b;
    d.hdr = (unsigned long long*)(b + kToppDistItems);
    d.cross = d.hdr + kPickMaxChunks;
    d.epoch = (const unsigned*)(d.cross + 2);
    d.bad = (unsigned*)(d.cross + 2) + 1;
    return d;
}
// The error word of the distributed top-p pick (topp_pick.hpp ToppDistParams::bad; bit 0: a bounded wait gave up -- the launch then drained without
// ending the step: no token, the cursor not advanced; bit 1: a predicted binade did not hold -- the token came from wrong sums) is read at every
// synchronising exit like the hand-off word above: the call fails and the word is cleared, so that one hiccup neither leaves with rc 0 nor
// makes every later launch give up its waits after 1 024 spins (bit 0 is what the waiting workgroups watch).  The stream is idle when this runs.
static int topp_dist_check(rama_ctx* c) {
    if (!c->topp_dist || !c->topp_dist_dirty) return 0;      // (gated like handoff_dirty: no device round trip per rama_sync of a host-driven loop)
    c->topp_dist_dirty = false;
    unsigned bad = 0;
    unsigned* word = topp_dist_params(c).bad;
    HIPCHK(hipMemcpy(&bad, word, sizeof bad, hipMemcpyDeviceToHost));
    if (!bad) return 0;
    HIPCHK(hipMemset(word, 0, sizeof bad));
    return fail(RAMA_EINVAL, (bad & 1u) ? "top-p sampler: a hand-off of the distributed pick timed out (no token was produced)"
                                        : "top-p sampler: a predicted binade of the running sum did not hold (the token is not trustworthy)", __FILE__, __LINE__);
}
// test entry (not in the C ABI header): set the word, as a launch whose wait gave up would
extern "C" int rama_internal_topp_dist_poke(rama_ctx* c, unsigned value) {
    RAMA_ENTER(c);
    if (!c || !c->topp_dist) return 1;
    hipStreamSynchronize(c->stream);
    c->topp_dist_dirty = true;
    return hipMemcpy(topp_dist_params(c).bad, &value, sizeof value, hipMemcpyHostToDevice) != hipSuccess;
}
// diagnostics (not in the C ABI header): bit 0 a hand-off wait of topp_pick_dist_kernel timed out, bit 1 a predicted binade did not hold
extern "C" int rama_internal_topp_dist_bad(rama_ctx* c, unsigned* bad) {
    RAMA_ENTER(c);
    if (!c || !bad) return 1;
    *bad = 0;
    if (!c->topp_dist) return 0;
    hipStreamSynchronize(c->stream);
    return hipMemcpy(bad, topp_dist_params(c).bad, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess;
}

// scratch for n logits; called outside any stream capture
static int ensure_topp_scratch(rama_ctx* c, int n) {
    if (n <= c->topp_cap) return 0;
    if (set_device(c)) return 1;
    HIPCHK(hipStreamSynchronize(c->stream));
    const size_t nblk = ((size_t)n + kToppBlock - 1) / kToppBlock;
    hipFree(c->topp_keys[1]); hipFree(c->topp_vals[1]);
    HIPCHK(hipMalloc(&c->topp_keys[1], sizeof(float) * n));
    HIPCHK(hipMalloc(&c->topp_vals[1], sizeof(int) * n));
    hipFree(c->topp_prefix); HIPCHK(hipMalloc(&c->topp_prefix, sizeof(float) * n));
    if (!c->topp_m) { HIPCHK(hipMalloc(&c->topp_m, sizeof(int))); HIPCHK(hipMalloc(&c->topp_err, sizeof(unsigned))); HIPCHK(hipMemset(c->topp_err, 0, sizeof(unsigned))); }
    hipFree(c->topp_bp); hipFree(c->topp_bi); hipFree(c->topp_bcount); hipFree(c->topp_racc);
    HIPCHK(hipMalloc(&c->topp_racc, sizeof(int) * kToppBlock * std::max<size_t>(nblk, kToppMaxBlocks)));
    HIPCHK(hipMalloc(&c->topp_bp, sizeof(float) * kToppBlock * std::max<size_t>(nblk, kToppMaxBlocks)));
    HIPCHK(hipMalloc(&c->topp_bi, sizeof(int) * kToppBlock * std::max<size_t>(nblk, kToppMaxBlocks)));
    HIPCHK(hipMalloc(&c->topp_bcount, sizeof(int) * std::max<size_t>(((size_t)n + 511) / 512 + 1, std::max<size_t>(nblk, kToppMaxBlocks))));
    hipFree(c->topp_stats); HIPCHK(hipMalloc(&c->topp_stats, sizeof(ToppStats) * (((size_t)n + 1023) / 1024 + 1)));
    hipFree(c->topp_rk); hipFree(c->topp_bm); hipFree(c->topp_approx);
    HIPCHK(hipMalloc(&c->topp_rk, sizeof(unsigned long long) * kToppBlock * std::max<size_t>(nblk, kToppMaxBlocks)));
    HIPCHK(hipMalloc(&c->topp_bm, sizeof(unsigned long long) * kToppBlock * std::max<size_t>(nblk, kToppMaxBlocks)));
    HIPCHK(hipMalloc(&c->topp_approx, sizeof(float) * n));
    if (!c->topp_dist) { HIPCHK(hipMalloc(&c->topp_dist, kToppDistBytes)); HIPCHK(hipMemset(c->topp_dist, 0, kToppDistBytes)); }
    c->topp_cap = n;
    return 0;
}

// the sampling tail of a step: argmax (temperature 0) or top-p, then whatever `fin` asks for
// (result word, cursor advance, next embedding gather)
static int enqueue_sample_launches(rama_ctx* c, ArgmaxParams fin, float temperature, float topp, float u);
static int enqueue_sample(rama_ctx* c, ArgmaxParams fin, float temperature, float topp, float u) {
    // kernel class RAMA_K_SAMPLE: up to five launches, so the bracket is a pair of event records around them (eager mode only)
    KProf& k = c->kp;
    const bool timed = k.kernel_id == RAMA_K_SAMPLE && k.used < k.max_records;
    if (timed) HIPCHK(hipEventRecord(k.ev[2 * k.used], c->stream));
    const int rc = enqueue_sample_launches(c, fin, temperature, topp, u);
    if (timed) { HIPCHK(hipEventRecord(k.ev[2 * k.used + 1], c->stream)); k.used++; }
    return rc;
}
// [r4] the small-block ordering: partial statistics, BS-entry block sorts, (block, OB blocks) pair ranking, scatter
template <int BS, int OB>
static int enqueue_small_blocks(rama_ctx* c, ToppSortParams sp, int nstat) {
    sp.nblk = (sp.n + BS - 1) / BS;
    hipLaunchKernelGGL(topp_stats_kernel, dim3(nstat), dim3(1024), 0, c->stream, sp, c->topp_stats);
    LAUNCHCHK();
    hipLaunchKernelGGL(topp_blocksort_bs_kernel<BS>, dim3(sp.nblk), dim3(BS), 0, c->stream, sp, (const ToppStats*)c->topp_stats, nstat);
    LAUNCHCHK();
    hipLaunchKernelGGL((topp_rank_pairs_bs_kernel<BS, OB>), dim3(sp.nblk, (sp.nblk + OB - 1) / OB), dim3(BS / 2), 0, c->stream, sp);
    LAUNCHCHK();
    hipLaunchKernelGGL(topp_rank_scatter_bs_kernel<BS>, dim3((sp.nblk * BS + 1023) / 1024), dim3(1024), 0, c->stream, sp);
    LAUNCHCHK();
    return 0;
}

static int enqueue_sample_launches(rama_ctx* c, ArgmaxParams fin, float temperature, float topp, float u) {
    if (temperature == 0.0f) {
        hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, c->stream, fin);
        LAUNCHCHK();
        return 0;
    }
    REQUIRE(fin.n > 1 && fin.n <= c->topp_cap, RAMA_EINVAL, "top-p sampler: scratch not prepared");
    ToppParams tp{};
    tp.logits = fin.logits; tp.n = fin.n; tp.temperature = temperature; tp.topp = topp; tp.u = u;
    tp.keys = c->topp_keys[1]; tp.vals = c->topp_vals[1]; tp.prefix = c->topp_prefix; tp.m = c->topp_m; tp.err = c->topp_err;
    // block sorts in LDS, then every kept entry's rank by binary searches in the other blocks (topp_sort.hpp): through LDS
    // when the whole list fits one workgroup's LDS (n <= 32768), through global memory otherwise (or with "topp_sort" = 0)
    const bool lds_path = c->tune_topp_sort && fin.n <= kToppBlock * kToppMaxBlocks;
    ToppSortParams sp{};
    sp.logits = fin.logits; sp.n = fin.n; sp.temperature = temperature; sp.topp = topp;
    sp.bp = c->topp_bp; sp.bi = c->topp_bi; sp.bcount = c->topp_bcount; sp.keys = c->topp_keys[1]; sp.vals = c->topp_vals[1];
    sp.m = c->topp_m; sp.err = c->topp_err; sp.nblk = (fin.n + kToppBlock - 1) / kToppBlock;
    const bool pairs = lds_path && c->tune_topp_pairs && sp.nblk > 1;
    if (pairs) sp.racc = c->topp_racc;
    bool dist = false;
    if (pairs && c->tune_topp_block != kToppBlock) {
        sp.rk = c->topp_rk; sp.bm = c->topp_bm; sp.approx = c->topp_approx;
        dist = c->tune_topp_dist != 0;
        if (dist) sp.epoch = (unsigned*)topp_dist_params(c).epoch;
        // small blocks (topp_sort.hpp [r4]): the statistics once per 1024 logits, 1024-entry sorts on 8 waves (or 512 on 4), the ranking by
        // (block, block) pairs over the whole chip, a scatter launch
        const int nstat = (fin.n + 1023) / 1024;
        if (c->tune_topp_block == 1024) { if (int rc = enqueue_small_blocks<1024, 2>(c, sp, nstat)) return rc; }
        else if (int rc = enqueue_small_blocks<512, 4>(c, sp, nstat)) return rc;
    } else {
    if (fin.n <= kToppBlock * kToppMaxBlocks) hipLaunchKernelGGL(topp_blocksort_kernel<false>, dim3(sp.nblk), dim3(1024), 0, c->stream, sp);
    else hipLaunchKernelGGL(topp_blocksort_kernel<true>, dim3(sp.nblk), dim3(1024), 0, c->stream, sp);
    LAUNCHCHK();
    if (pairs) {
        hipLaunchKernelGGL(topp_rank_pairs_kernel, dim3(sp.nblk, sp.nblk), dim3(1024), 0, c->stream, sp);
        LAUNCHCHK();
        hipLaunchKernelGGL(topp_rank_scatter_kernel, dim3(sp.nblk * (kToppBlock / 1024)), dim3(1024), 0, c->stream, sp);
    } else if (lds_path) hipLaunchKernelGGL(topp_rank_kernel<kToppMaxBlocks>, dim3(sp.nblk * (kToppBlock / kRankThreads)), dim3(kRankThreads), 0, c->stream, sp);
    else hipLaunchKernelGGL(topp_rank_global_kernel, dim3(sp.nblk * (kToppBlock / 256)), dim3(256), 0, c->stream, sp);
    LAUNCHCHK();
    }
    if (lds_path && !c->tune_topp_keep_sums) tp.prefix = nullptr;
    // the running sums: the exact parallel scan with the list in LDS, or (longer lists) the staged lane ripple
    if (dist) { c->topp_dist_dirty = true; hipLaunchKernelGGL(topp_pick_dist_kernel, dim3(kPickMaxChunks), dim3(1024), 0, c->stream, tp, topp_dist_params(c), fin); }
    else if (lds_path) hipLaunchKernelGGL(topp_pick_scan_kernel, dim3(1), dim3(1024), 0, c->stream, tp, fin);
    else hipLaunchKernelGGL(topp_pick_kernel, dim3(1), dim3(1024), 0, c->stream, tp, fin);
    LAUNCHCHK();
    return 0;
}

int rama_sample_topp_dev(rama_ctx* c, const float* logits, size_t n, float temperature, float topp, float u, int32_t* result_dev) {
    RAMA_ENTER(c);
    REQUIRE(c && logits && result_dev && n > 1 && n < (1u << 30), RAMA_EINVAL, "sample_topp_dev: bad argument");
    int rc = temperature != 0.0f ? ensure_topp_scratch(c, (int)n) : 0;
    if (rc) return rc;
    ArgmaxParams ap{};
    ap.logits = logits; ap.n = (int)n; ap.result = (int*)result_dev;
    return enqueue_sample(c, ap, temperature, topp, u);
}

int rama_decode_sampler(rama_ctx* c, float temperature, float topp, float u) {
    RAMA_ENTER(c);
    REQUIRE(c, RAMA_EINVAL, "ctx is NULL");
    REQUIRE(temperature >= 0.0f && topp >= 0.0f && topp <= 1.0f && u >= 0.0f && u < 1.0f, RAMA_EINVAL, "decode_sampler: temperature >= 0, topp in [0,1], u in [0,1)");
    if (temperature != c->samp_T || topp != c->samp_topp || u != c->samp_u) {
        HIPCHK(hipStreamSynchronize(c->stream));
        drop_graph(c);
        c->samp_T = temperature; c->samp_topp = topp; c->samp_u = u;
    }
    return 0;
}

int rama_argmax_dev(rama_ctx* c, const float* logits, size_t n, int32_t* result_dev) {
    RAMA_ENTER(c);
    REQUIRE(c && logits && result_dev && n > 0, RAMA_EINVAL, "argmax_dev: bad argument");
    ArgmaxParams ap{};
    ap.logits = logits; ap.n = (int)n; ap.result = (int*)result_dev;
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, c->stream, ap);
    LAUNCHCHK();
    return 0;
}

int rama_forward(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s, int token, int pos) {
    RAMA_ENTER(c);
    REQUIRE(cfg, RAMA_EINVAL, "config is NULL");
    rama_stage st{0, cfg->n_layers, 1, 1};
    return rama_forward_stage(c, cfg, w, s, token, pos, &st);
}

// ---- token-batch passes on the matrix cores (prefill_mfma.hpp): up to kMfMaxTok tokens -- the forced
// prompt positions of one sequence (rama_prefill) or one token of each of several independent
// sequences (rama_decode_batch) -- go through a layer together, every weight row streamed once.

// One GEMM launch.  A unit = (row group, K-slice); the launch is cut into one even round over the CUs
// (the kernels hold one 8-wave workgroup per CU), consecutive units of a workgroup prefetching across
// their boundary.  ksplit > 1 only for EPI_STORE, whose K-slices land in slabs the next rmsnorm folds.
template <int RT, int EPI>
static int launch_mf(rama_ctx* c, MfParams& p, int pt) {
    constexpr bool across = EPI == EPI_QKV || EPI == EPI_SWIGLU;
    const int groups = (p.rows + (across ? 16 : 16 * RT) - 1) / (across ? 16 : 16 * RT);
    const int total = groups * p.ksplit;
    p.nunit = std::max(1, (total + std::max(1, c->cu_count) - 1) / std::max(1, c->cu_count));
    const dim3 grid((total + p.nunit - 1) / p.nunit), block(kMfThreads);
    if (p.tiled) {      // p.w[] point at the model's tile-order copies: contiguous 1-KiB weight reads, no lane permute
        if (pt == 1) hipLaunchKernelGGL((gemm_mfma_rows<1, RT, EPI, 2, 3>), grid, block, 0, c->stream, p);
        else if (pt == 2) hipLaunchKernelGGL((gemm_mfma_rows<2, RT, EPI, 2, 3>), grid, block, 0, c->stream, p);
        // MIX (loads scheduled into the MFMA stream) where tools/pf_mfma_bench.hip measures it faster: same arithmetic, same bits
        else if (pt <= 4) hipLaunchKernelGGL((gemm_mfma_rows<4, RT, EPI, 2, 3, 0, (EPI == EPI_QKV || EPI == EPI_STORE) ? 2 : 0>), grid, block, 0, c->stream, p);
        // 65 .. 128 tokens: one K-block per step; as many token tiles as the pass has (a pass of 70 positions pays for 80, not 128)
        else if (pt == 5) hipLaunchKernelGGL((gemm_mfma_rows<5, RT, EPI, 1, 3, 0, EPI == EPI_QKV ? 2 : 0>), grid, block, 0, c->stream, p);
        else if (pt == 6) hipLaunchKernelGGL((gemm_mfma_rows<6, RT, EPI, 1, 3, 0, EPI == EPI_QKV ? 2 : 0>), grid, block, 0, c->stream, p);
        else if (pt == 7) hipLaunchKernelGGL((gemm_mfma_rows<7, RT, EPI, 1, 3, 0, EPI == EPI_QKV ? 2 : 0>), grid, block, 0, c->stream, p);
        else hipLaunchKernelGGL((gemm_mfma_rows<8, RT, EPI, 1, 3, 0, EPI == EPI_QKV ? 2 : 0>), grid, block, 0, c->stream, p);
    } else {
        REQUIRE(pt <= 4, RAMA_EUNSUP, "token batch: more than 64 tokens per pass need the tile-order weight copy");
        if (pt == 1) hipLaunchKernelGGL((gemm_mfma_rows<1, RT, EPI>), grid, block, 0, c->stream, p);
        else if (pt == 2) hipLaunchKernelGGL((gemm_mfma_rows<2, RT, EPI>), grid, block, 0, c->stream, p);
        else hipLaunchKernelGGL((gemm_mfma_rows<4, RT, EPI>), grid, block, 0, c->stream, p);
    }
    LAUNCHCHK();
    return 0;
}

// p.w[0..n) = layer li of the given row-major tensors, or of their tile-order copies when the model has them all
static void mf_weights(const rama_ctx* c, MfParams& p, int n, const float* const* base, size_t per_layer, size_t li) {
    const float* t[3] = {nullptr, nullptr, nullptr};
    bool all = c->tune_tiled != 0;
    for (int i = 0; i < n && all; i++) { t[i] = rama_internal_tiled_lookup(base[i]); all = t[i] != nullptr; }
    for (int i = 0; i < n; i++) p.w[i] = (all ? t[i] : base[i]) + li * per_layer;
    p.tiled = all ? 1 : 0;
}

// K-slices of the Wo / W2 products: enough (row group, slice) units to give every CU one, while a
// slice still feeds the workgroup's 8 waves two steps each (16 blocks of 16 floats per wave-step pair)
static int mf_ksplit(const rama_ctx* c, int rows, int K) {
    const int groups = (rows + 31) / 32;
    int ks = 1;
    while (ks < 4 && groups * ks < c->cu_count && K / (ks * 2) >= 512) ks *= 2;
    return ks;
}

// scratch of a pass (tile layout, kMfMaxTok tokens): X residual stream, XN its rmsnorm, Q, XB, HB
// [hidden], SL = up to 4 K-slice slabs [dim]; then token ids, the sequence table, and (decode_batch)
// row-major logits [kMfMaxTok, vocab]
struct BatchScratch { float *X, *XN, *Q, *XB, *HB, *SL, *LG, *SSP; size_t slab; int* toks; SeqSlot* seqs; };

static int ensure_batch_scratch(rama_ctx* c, const rama_config* cfg, bool with_logits, BatchScratch* b) {
    const size_t T = kMfMaxTok, dim = cfg->dim, hidden = cfg->hidden_dim;
    const size_t ints = T + (sizeof(SeqSlot) / sizeof(int)) * T;      // token ids [T], then the sequence table [T]
    const size_t ssp = (T / 16) * kRmsParts * 16;        // partial sums of squares per (tile, part, token)
    const size_t need = T * (8 * dim + hidden) + ssp + ints + 64 + (with_logits ? T * (size_t)cfg->vocab_size : 0);
    if (need > c->pf_floats) {
        if (c->pf_blob) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->pf_blob)); c->pf_blob = nullptr; }
        // the chained-batch graph has the old scratch pointers baked in: it must not be replayed on freed memory
        if (c->bc.exec) { hipGraphExecDestroy(c->bc.exec); c->bc.exec = nullptr; }
        if (c->bc.graph) { hipGraphDestroy(c->bc.graph); c->bc.graph = nullptr; }
        c->bc.graph_bucket = -1;
        if (set_device(c)) return 1;
        HIPCHK(hipMalloc(&c->pf_blob, need * sizeof(float)));
        // stale tokens of a partly filled 16-token tile flow through the GEMMs as extra columns (never
        // stored): keep them finite
        HIPCHK(hipMemsetAsync(c->pf_blob, 0, need * sizeof(float), c->stream));
        c->pf_floats = need;
    }
    b->slab = T * dim;
    b->X = c->pf_blob; b->XN = b->X + T * dim; b->Q = b->XN + T * dim; b->XB = b->Q + T * dim;
    b->SL = b->XB + T * dim; b->HB = b->SL + 4 * T * dim;
    b->SSP = b->HB + T * hidden;
    b->toks = reinterpret_cast<int*>(b->SSP + ssp);
    b->seqs = reinterpret_cast<SeqSlot*>(b->toks + T);
    b->LG = reinterpret_cast<float*>(b->toks + ints + 64);      // ints is a multiple of 4: 16-byte aligned
    return 0;
}

// b.XN = rmsnorm(b.X += the nslab pending K-slices in b.SL) * gain, per token (prefill_mfma.hpp)
// With tune_norm_in_gemm (default) this is ONE launch: b.XN = b.X * gain, and the GEMM that consumes
// b.XN scales its outputs per token from the partial sums in b.SSP (MfParams::ssp = norm_ssp(c, b)).
static int launch_rmsnorm_tile(rama_ctx* c, const BatchScratch& b, const float* gain, int dim, int ntile, int nslab) {
    if (c->tune_norm_in_gemm) {
        hipLaunchKernelGGL(rms_fold_kernel, dim3(ntile, kRmsParts), dim3(kRmsFoldWaves * 64), 0, c->stream, b.X, b.SSP, dim, (const float*)b.SL, nslab, b.slab, b.XN, gain);
        LAUNCHCHK();
        return 0;
    }
    hipLaunchKernelGGL(rms_fold_kernel, dim3(ntile, kRmsParts), dim3(kRmsFoldWaves * 64), 0, c->stream, b.X, b.SSP, dim, (const float*)b.SL, nslab, b.slab, (float*)nullptr, (const float*)nullptr);
    LAUNCHCHK();
    hipLaunchKernelGGL(rms_scale_kernel, dim3(ntile, kRmsParts), dim3(256), 0, c->stream, b.XN, (const float*)b.X, gain, (const float*)b.SSP, dim);
    LAUNCHCHK();
    return 0;
}
static const float* norm_ssp(const rama_ctx* c, const BatchScratch& b) { return c->tune_norm_in_gemm ? b.SSP : nullptr; }

// tokens (prefill) / sequences (decode_batch) one weight pass can take: 128 when every matrix it streams has its
// tile-order copy (8 token tiles per wave need the contiguous 1-KiB weight reads), else 64
static int mf_pass_cap(const rama_ctx* c, const rama_weights* w, bool with_cls) {
    const float* ms[8] = {w->wq, w->wk, w->wv, w->wo, w->w1, w->w2, w->w3, w->wcls};
    bool all = c->tune_tiled != 0;
    for (int i = 0; i < (with_cls ? 8 : 7) && all; i++) all = rama_internal_tiled_lookup(ms[i]) != nullptr;
    return all ? kMfMaxTok : kMfMaxTokRows;
}

static bool mf_shape_ok(const rama_config* cfg) { return cfg->dim % 16 == 0 && cfg->hidden_dim % 16 == 0; }

// nt <= kMfMaxTok tokens (already embedded in b.X, tile layout) through every layer.
// seqs == false: consecutive positions p0.. of ONE sequence (its cache bases key_cache / value_cache);
// seqs == true: token t belongs to independent sequence t (device table b.seqs).
// tmax = the longest context any token of the pass attends to (timesteps): sizes the attention scratch.
// On return the residual stream is b.X + the *nslab_out K-slices in b.SL (folded by the caller).
static int run_layers_batched(rama_ctx* c, const rama_config* cfg, const rama_weights* w, const BatchScratch& b,
                              int nt, int p0, float* key_cache, float* value_cache, bool seqs, int tmax, int* nslab_out) {
    const int dim = cfg->dim, hidden = cfg->hidden_dim, hs = dim / cfg->n_heads;
    const size_t dd = (size_t)dim * dim, hd = (size_t)hidden * dim;
    const int ntile = (nt + 15) / 16, pt = ntile <= 1 ? 1 : (ntile == 2 ? 2 : (ntile <= 4 ? 4 : ntile));
    const int ks_wo = mf_ksplit(c, dim, dim), ks_w2 = mf_ksplit(c, dim, hidden);
    int pending = 0;      // K-slices of the previous product waiting in b.SL
    int rc;
    for (int layer = 0; layer < cfg->n_layers; layer++) {
        const size_t li = (size_t)layer;
        const size_t layer_off = li * cfg->seq_len * dim;
        MfParams p{};
        p.n_tok = nt; p.ksplit = 1; p.slab_floats = b.slab;
        // infer.rs:19 (+ the residual add of the previous layer's W2 product, :47)
        rc = launch_rmsnorm_tile(c, b, w->rms_att_weight + li * dim, dim, ntile, pending); if (rc) return rc;
        // infer.rs:20-33: Wq | Wk | Wv, RoPE, cache append
        { const float* bs[3] = {w->wq, w->wk, w->wv}; mf_weights(c, p, 3, bs, dd, li); }
        p.x = b.XN; p.o = b.Q; p.K = dim; p.rows = dim; p.ssp = norm_ssp(c, b);
        p.pos0 = p0; p.fr = w->freq_cis_real; p.fi = w->freq_cis_imag; p.head_size = hs;
        p.kc = key_cache ? key_cache + layer_off : nullptr; p.vc = value_cache ? value_cache + layer_off : nullptr;
        p.seqs = seqs ? b.seqs : nullptr; p.layer_off = layer_off;
        rc = launch_mf<3, EPI_QKV>(c, p, pt); if (rc) return rc;
        {   // infer.rs:34: query z attends to positions 0..pos(z)
            AttnParams a{};
            a.q = b.Q; a.kc = p.kc; a.vc = p.vc; a.att = nullptr; a.xb = b.XB; a.ctl = nullptr; a.pos_val = p0;
            a.dim = dim; a.head_size = hs; a.seq_len = cfg->seq_len; a.tiled = 1;
            a.seqs = seqs ? b.seqs : nullptr; a.layer_off = layer_off;
            const int G = hs <= 64 ? 16 : (hs <= 128 ? 32 : 64);
            dim3 grid(cfg->n_heads, 1, nt);
            const bool tiles = !seqs && c->tune_prefill_attn && (hs == 16 || hs == 32 || hs == 48 || hs == 64 || hs == 128);
            if (tiles) {            // one sequence: 16 queries per workgroup share every cache row (prefill_attn.hpp)
                const dim3 tg(cfg->n_heads, ntile);
#define RAMA_TILE_ATTN(W_) do { \
                    if (hs == 16) hipLaunchKernelGGL((attention_tile_mfma_kernel<1, W_>), tg, dim3(W_ * 64), 0, c->stream, a, nt); \
                    else if (hs == 32) hipLaunchKernelGGL((attention_tile_mfma_kernel<2, W_>), tg, dim3(W_ * 64), 0, c->stream, a, nt); \
                    else if (hs == 48) hipLaunchKernelGGL((attention_tile_mfma_kernel<3, W_>), tg, dim3(W_ * 64), 0, c->stream, a, nt); \
                    else if (hs == 64) hipLaunchKernelGGL((attention_tile_mfma_kernel<4, W_>), tg, dim3(W_ * 64), 0, c->stream, a, nt); \
                    else hipLaunchKernelGGL((attention_tile_mfma_kernel<8, W_>), tg, dim3(W_ * 64), 0, c->stream, a, nt); } while (0)
                if (tmax >= 512) RAMA_TILE_ATTN(8);     // long contexts: 8 waves share the key tiles
                else RAMA_TILE_ATTN(4);
#undef RAMA_TILE_ATTN
            } else
            // scores scratch: tmax = the longest context of the pass (timesteps)
            if (tmax <= 256) {      // short contexts: 4-wave workgroups, 8 of them per CU
                size_t shm = (size_t)(attn_scratch_floats(G, 4) + tmax) * sizeof(float);
                if (G == 16) hipLaunchKernelGGL((attention_kernel<16, false, 4>), grid, dim3(256), shm, c->stream, a);
                else if (G == 32) hipLaunchKernelGGL((attention_kernel<32, false, 4>), grid, dim3(256), shm, c->stream, a);
                else hipLaunchKernelGGL((attention_kernel<64, false, 4>), grid, dim3(256), shm, c->stream, a);
            } else {
                size_t shm = (size_t)(attn_scratch_floats(G) + tmax) * sizeof(float);
                REQUIRE(shm <= 64 * 1024, RAMA_EUNSUP, "token batch: context too long for the single-workgroup attention kernel");
                if (G == 16) hipLaunchKernelGGL((attention_kernel<16, false>), grid, dim3(kAttnThreads), shm, c->stream, a);
                else if (G == 32) hipLaunchKernelGGL((attention_kernel<32, false>), grid, dim3(kAttnThreads), shm, c->stream, a);
                else hipLaunchKernelGGL((attention_kernel<64, false>), grid, dim3(kAttnThreads), shm, c->stream, a);
            }
            LAUNCHCHK();
        }
        // infer.rs:35: Wo . xb as K-slices; the residual add (:37) rides in the next rmsnorm
        { const float* bs[1] = {w->wo}; mf_weights(c, p, 1, bs, dd, li); }
        p.x = b.XB; p.o = b.SL; p.K = dim; p.rows = dim; p.ksplit = ks_wo; p.ssp = nullptr;
        rc = launch_mf<2, EPI_STORE>(c, p, pt); if (rc) return rc;
        // infer.rs:37,39
        rc = launch_rmsnorm_tile(c, b, w->rms_ffn_weight + li * dim, dim, ntile, ks_wo); if (rc) return rc;
        // infer.rs:41-45: W1 | W3, SiLU * gate
        { const float* bs[2] = {w->w1, w->w3}; mf_weights(c, p, 2, bs, hd, li); }
        p.x = b.XN; p.o = b.HB; p.K = dim; p.rows = hidden; p.ksplit = 1; p.ssp = norm_ssp(c, b);
        rc = launch_mf<2, EPI_SWIGLU>(c, p, pt); if (rc) return rc;
        // infer.rs:46: W2 . hb as K-slices (:47 rides in the next rmsnorm / the caller's fold)
        { const float* bs[1] = {w->w2}; mf_weights(c, p, 1, bs, hd, li); }
        p.x = b.HB; p.o = b.SL; p.K = hidden; p.rows = dim; p.ksplit = ks_w2; p.ssp = nullptr;
        rc = launch_mf<2, EPI_STORE>(c, p, pt); if (rc) return rc;
        pending = ks_w2;
    }
    *nslab_out = pending;
    return 0;
}

// ---- parity mode: the forced prompt positions in the reference's rounding order, 32 per weight pass (chain.hpp
// gemm_chain_kernel).  Every position's cache rows are what its own forward() writes, bit for bit: the same chains per
// output, the same exact sums in the norms and the softmax, the same attention per (head, position).  The LAST position
// then goes through forward() itself, which leaves the run state (x, xb, q, .., logits) exactly as the reference's loop
// does (mod.rs:187-194).  *done = false: shape or copies not available, nothing was enqueued.
template <int EPI>
static int launch_gemm_chain(rama_ctx* c, GemmChainParams& p) {
    p.lane_reduce = c->tune_lane_reduce;
    const dim3 grid(p.nmat * ((p.rows + 15) / 16)), block(kGcThreads);
    if (p.n_tok <= 4) hipLaunchKernelGGL((gemm_chain_kernel<1, EPI>), grid, block, gemm_chain_lds_bytes(1), c->stream, p);
    else if (p.n_tok <= 8) hipLaunchKernelGGL((gemm_chain_kernel<2, EPI>), grid, block, gemm_chain_lds_bytes(2), c->stream, p);
    else if (p.n_tok <= 16) hipLaunchKernelGGL((gemm_chain_kernel<4, EPI>), grid, block, gemm_chain_lds_bytes(4), c->stream, p);
    else hipLaunchKernelGGL((gemm_chain_kernel<8, EPI>), grid, block, gemm_chain_lds_bytes(8), c->stream, p);
    LAUNCHCHK();
    return 0;
}
constexpr int kGcMaxTok = 8 * kGcWaves;

// the layers of one pass of nt <= 16 tokens whose residual rows sit in b.X: consecutive positions p0.. of one sequence
// (key_cache / value_cache its cache bases), or -- seqs != NULL -- token t of independent sequence t (device table)
struct ChainBatch { float *X, *XN, *Q, *XB, *HB, *ATT; const float *cq, *ck, *cv, *co, *c13, *c2; int nw; size_t att_lds; };
static int chain_batch_layers(rama_ctx* c, const rama_config* cfg, const rama_weights* w, const ChainBatch& b, int nt, int p0,
                              float* key_cache, float* value_cache, const SeqSlot* seqs) {
    const int dim = cfg->dim, hidden = cfg->hidden_dim, hs = dim / cfg->n_heads, H = cfg->n_heads, seq = cfg->seq_len;
    const size_t dd = (size_t)dim * dim, hd = (size_t)hidden * dim;
    int rc;
    for (int layer = 0; layer < cfg->n_layers; layer++) {
        const size_t li = (size_t)layer, layer_off = li * seq * dim;
        float* kc = key_cache ? key_cache + layer_off : nullptr;
        float* vc = value_cache ? value_cache + layer_off : nullptr;
        rc = launch_rmsnorm_chain(c, b.XN, b.X, w->rms_att_weight + li * dim, dim, nullptr, nt, dim); if (rc) return rc;      // infer.rs:19
        {   // :20-33
            GemmChainParams p{};
            p.w[0] = b.cq + li * dd; p.w[1] = b.ck + li * dd; p.w[2] = b.cv + li * dd; p.nmat = 3; p.K = dim; p.rows = dim;
            p.x = b.XN; p.xstride = dim; p.o[0] = b.Q; p.ostride = dim; p.n_tok = nt;
            p.pos0 = p0; p.fr = w->freq_cis_real; p.fi = w->freq_cis_imag; p.head_size = hs; p.kc = kc; p.vc = vc;
            p.seqs = seqs; p.layer_off = layer_off;
            rc = launch_gemm_chain<CEPI_QKV>(c, p); if (rc) return rc;
        }
        {   // :34, one workgroup per (head, token)
            RefAttnParams a{};
            a.q = b.Q; a.kc = kc; a.vc = vc; a.att = b.ATT; a.xb = b.XB; a.ctl = nullptr; a.pos_val = p0;
            a.dim = dim; a.head_size = hs; a.seq_len = seq; a.tok_stride = dim; a.att_stride = H * seq;
            a.seqs = seqs; a.layer_off = layer_off;
            const dim3 grid(H, nt);
            if (b.nw == 4) hipLaunchKernelGGL((attention_chain_kernel<4>), grid, dim3(4 * 64), b.att_lds, c->stream, a);
            else if (b.nw == 8) hipLaunchKernelGGL((attention_chain_kernel<8>), grid, dim3(8 * 64), b.att_lds, c->stream, a);
            else hipLaunchKernelGGL((attention_chain_kernel<16>), grid, dim3(16 * 64), b.att_lds, c->stream, a);
            LAUNCHCHK();
        }
        {   // :35-37
            GemmChainParams p{};
            p.w[0] = b.co + li * dd; p.nmat = 1; p.K = dim; p.rows = dim; p.x = b.XB; p.xstride = dim; p.n_tok = nt;
            p.resid = b.X; p.rstride = dim;
            rc = launch_gemm_chain<CEPI_RESID>(c, p); if (rc) return rc;
        }
        rc = launch_rmsnorm_chain(c, b.XN, b.X, w->rms_ffn_weight + li * dim, dim, nullptr, nt, dim); if (rc) return rc;      // :39
        {   // :41-45
            GemmChainParams p{};
            p.w[0] = b.c13 + li * 2 * hd; p.nmat = 1; p.K = dim; p.rows = 2 * hidden; p.x = b.XN; p.xstride = dim; p.n_tok = nt;
            p.o[0] = b.HB; p.ostride = hidden;
            rc = launch_gemm_chain<CEPI_SWIGLU>(c, p); if (rc) return rc;
        }
        {   // :46-47
            GemmChainParams p{};
            p.w[0] = b.c2 + li * hd; p.nmat = 1; p.K = hidden; p.rows = dim; p.x = b.HB; p.xstride = hidden; p.n_tok = nt;
            p.resid = b.X; p.rstride = dim;
            rc = launch_gemm_chain<CEPI_RESID>(c, p); if (rc) return rc;
        }
    }
    return 0;
}

// shape check, chain-order copies and scratch of the parity-mode token batches.  *ok = false: not available for this
// shape / these weights (the caller falls back to one forward() per token).  Scratch, row-major per token: X residual
// stream, XN its norm, Q, XB attention output, HB; att rows; token ids; the sequence table; (with_logits) LG [16][vocab]
struct ChainScratch { ChainBatch b; int* toks; SeqSlot* seqs; float* LG; const float* ccls; };
static int chain_batch_setup(rama_ctx* c, const rama_config* cfg, const rama_weights* w, bool with_logits, ChainScratch* out, bool* ok) {
    *ok = false;
    const int dim = cfg->dim, hidden = cfg->hidden_dim, hs = dim / cfg->n_heads, H = cfg->n_heads, seq = cfg->seq_len, V = cfg->vocab_size;
    if (!c->tune_chain || !c->tune_prefill_chain || dim % 16 || hidden % 16 || dim > 16000 || hidden > 16000 || !attn_chain_ok(hs, seq)) return 0;
    if (!rmsnorm_chain_ok((size_t)dim)) return 0;
    int rc = ensure_chain_copy(c, cfg, w, nullptr); if (rc) return rc;
    const float* cq = rama_internal_chain_lookup(w->wq, dim, dim), *ck = rama_internal_chain_lookup(w->wk, dim, dim);
    const float* cv = rama_internal_chain_lookup(w->wv, dim, dim), *co = rama_internal_chain_lookup(w->wo, dim, dim);
    const float* c13 = rama_internal_chain_lookup(w->w1, 2 * hidden, dim), *c2 = rama_internal_chain_lookup(w->w2, dim, hidden);
    const float* ccls = with_logits ? rama_internal_chain_lookup(w->wcls, V, dim) : nullptr;
    if (!cq || !ck || !cv || !co || !c13 || !c2 || (with_logits && !ccls)) return 0;
    const size_t T = kGcMaxTok;
    const size_t slot_floats = (sizeof(SeqSlot) * T + sizeof(float) - 1) / sizeof(float);
    const size_t need = T * (4 * (size_t)dim + hidden) + T * (size_t)H * seq + 64 + slot_floats + 4 + (with_logits ? T * (size_t)V : 0);
    if (need > c->pc_floats) {
        if (c->pc_blob) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->pc_blob)); c->pc_blob = nullptr; }
        HIPCHK(hipMalloc(&c->pc_blob, need * sizeof(float)));
        c->pc_floats = need;
    }
    float* X = c->pc_blob, *XN = X + T * dim, *Q = XN + T * dim, *XB = Q + T * dim, *HB = XB + T * dim, *ATT = HB + T * hidden;
    out->toks = reinterpret_cast<int*>(ATT + T * (size_t)H * seq);
    out->seqs = reinterpret_cast<SeqSlot*>(out->toks + 64);                       // 8-byte aligned: every term above is a multiple of 16 floats
    out->LG = reinterpret_cast<float*>(out->toks + 64) + slot_floats + 4 - ((slot_floats) & 3);      // 16-byte aligned
    out->ccls = ccls;
    const int nw = attn_chain_waves(hs, false);
    const size_t att_lds = attn_chain_lds_floats(hs, seq, nw) * sizeof(float) + 16;
    REQUIRE(att_lds <= kAttnChainMaxLds, RAMA_EUNSUP, "token batch (parity mode): context too long for the score buffer");
    out->b = ChainBatch{X, XN, Q, XB, HB, ATT, cq, ck, cv, co, c13, c2, nw, att_lds};
    *ok = true;
    return 0;
}

static int prefill_chain(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s,
                         const int32_t* tokens_host, int n_tokens, int pos0, bool* done) {
    *done = false;
    if (n_tokens < 2) return 0;
    const int dim = cfg->dim;
    ChainScratch sc{};
    int rc = chain_batch_setup(c, cfg, w, false, &sc, done); if (rc || !*done) return rc;
    float* X = sc.b.X; int* toks = sc.toks;
    const ChainBatch& cb = sc.b;
    c->embedded_x = nullptr; c->host_pos = -1;
    const int n_batch = n_tokens - 1;                             // the last position runs as forward()
    for (int c0 = 0; c0 < n_batch; c0 += kGcMaxTok) {
        const int nt = std::min(kGcMaxTok, n_batch - c0), p0 = pos0 + c0;
        HIPCHK(hipStreamSynchronize(c->stream));                  // the pinned staging buffer is free again
        memcpy(c->pinned_tok, tokens_host + c0, sizeof(int) * nt);
        HIPCHK(hipMemcpyAsync(toks, c->pinned_tok, sizeof(int) * nt, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(embed_rows_kernel, dim3((dim + 255) / 256, nt), dim3(256), 0, c->stream, X, w->token_embedding_table, (const int*)toks, nt, dim);
        LAUNCHCHK();
        rc = chain_batch_layers(c, cfg, w, cb, nt, p0, s->key_cache, s->value_cache, nullptr); if (rc) return rc;
    }
    return rama_forward(c, cfg, w, s, tokens_host[n_tokens - 1], pos0 + n_tokens - 1);
}

// rama_decode_batch in parity mode: the sequences share every weight pass, 32 at a time, through the same kernels; each
// sequence's appended cache rows and its logits are bit for bit those of its own forward().  (x / xb / q .. of the states
// are not maintained -- the call's contract, see the header.)
static int decode_batch_chain(rama_ctx* c, const rama_config* cfg, const rama_weights* w, const rama_run_state* states,
                              const int32_t* tokens_host, const int32_t* pos_host, int n_seq, bool* done) {
    *done = false;
    const int dim = cfg->dim, V = cfg->vocab_size;
    if (n_seq < 2) return 0;
    ChainScratch sc{};
    int rc = chain_batch_setup(c, cfg, w, true, &sc, done); if (rc || !*done) return rc;
    c->embedded_x = nullptr; c->host_pos = -1;
    SeqSlot* slots = reinterpret_cast<SeqSlot*>(c->pinned_tok + kMfMaxTok);
    for (int c0 = 0; c0 < n_seq; c0 += kGcMaxTok) {
        const int nt = std::min(kGcMaxTok, n_seq - c0);
        HIPCHK(hipStreamSynchronize(c->stream));                  // the pinned staging buffers are free again
        for (int i = 0; i < nt; i++) {
            c->pinned_tok[i] = tokens_host[c0 + i];
            slots[i].kc = states[c0 + i].key_cache; slots[i].vc = states[c0 + i].value_cache; slots[i].pos = pos_host[c0 + i]; slots[i].pad = 0;
        }
        HIPCHK(hipMemcpyAsync(sc.toks, c->pinned_tok, sizeof(int) * nt, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(sc.seqs, slots, sizeof(SeqSlot) * nt, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(embed_rows_kernel, dim3((dim + 255) / 256, nt), dim3(256), 0, c->stream, sc.b.X, w->token_embedding_table, (const int*)sc.toks, nt, dim);
        LAUNCHCHK();
        rc = chain_batch_layers(c, cfg, w, sc.b, nt, 0, nullptr, nullptr, sc.seqs); if (rc) return rc;
        // infer.rs:49-51 per sequence: x = rmsnorm(x), logits = Wcls . x
        rc = launch_rmsnorm_chain(c, sc.b.XN, sc.b.X, w->rms_final_weight, dim, nullptr, nt, dim); if (rc) return rc;
        GemmChainParams p{};
        p.w[0] = sc.ccls; p.nmat = 1; p.K = dim; p.rows = V; p.x = sc.b.XN; p.xstride = dim; p.o[0] = sc.LG; p.ostride = V; p.n_tok = nt;
        rc = launch_gemm_chain<CEPI_STORE>(c, p); if (rc) return rc;
        for (int i = 0; i < nt; i++)
            HIPCHK(hipMemcpyAsync(states[c0 + i].logits, sc.LG + (size_t)i * V, sizeof(float) * V, hipMemcpyDeviceToDevice, c->stream));
    }
    return 0;
}

int rama_prefill(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s,
                 const int32_t* tokens_host, int n_tokens, int pos0) {
    RAMA_ENTER(c);
    REQUIRE(c && tokens_host, RAMA_EINVAL, "prefill: NULL argument");
    int rc = check_cfg(cfg); if (rc) return rc;
    rama_stage st{0, cfg->n_layers, 1, 1};
    rc = check_stage(cfg, w, s, &st); if (rc) return rc;
    REQUIRE(n_tokens >= 1 && pos0 >= 0 && pos0 + n_tokens <= cfg->seq_len, RAMA_EINVAL, "prefill: positions outside [0, seq_len)");
    for (int i = 0; i < n_tokens; i++) REQUIRE(tokens_host[i] >= 0 && tokens_host[i] < cfg->vocab_size, RAMA_EINVAL, "prefill: token outside the vocabulary");
    if (set_device(c)) return 1;
    const int dim = cfg->dim;
    const size_t att_floats = (size_t)attn_scratch_floats((dim / cfg->n_heads) <= 64 ? 16 : ((dim / cfg->n_heads) <= 128 ? 32 : 64)) + pos0 + n_tokens;
    if (c->tune_ref_order) {      // parity mode: the chain-order token-batch kernels when the shape and the copies allow
        bool done = false;
        rc = prefill_chain(c, cfg, w, s, tokens_host, n_tokens, pos0, &done);
        if (rc || done) return rc;
    }
    if (c->tune_ref_order || !mf_shape_ok(cfg) || att_floats * sizeof(float) > 64 * 1024) {
        // reference-order mode; widths that are not whole 16-float blocks, or contexts the one-workgroup attention cannot
        // hold: the reference's own schedule, one forward() per forced token (mod.rs:187-194)
        for (int i = 0; i < n_tokens; i++) { rc = rama_forward(c, cfg, w, s, tokens_host[i], pos0 + i); if (rc) return rc; }
        return 0;
    }
    BatchScratch b{};
    if (c->tune_tiled) { rc = rama_internal_model_ensure(c, w, 2); if (rc) return rc; }
    rc = ensure_batch_scratch(c, cfg, false, &b); if (rc) return rc;
    c->embedded_x = nullptr; c->host_pos = -1;
    int last_nt = 0, nslab = 0;
    // positions per weight pass: 128 when every layer matrix has its tile-order copy, else 64
    const int per_pass = std::min(c->tune_prefill_tok, mf_pass_cap(c, w, false));
    for (int c0 = 0; c0 < n_tokens; c0 += per_pass) {
        const int nt = std::min(per_pass, n_tokens - c0), p0 = pos0 + c0;
        last_nt = nt;
        // the ids go through the context's pinned staging buffer: the caller's array may be gone
        // before the copy runs
        HIPCHK(hipStreamSynchronize(c->stream));
        memcpy(c->pinned_tok, tokens_host + c0, sizeof(int) * nt);
        HIPCHK(hipMemcpyAsync(b.toks, c->pinned_tok, sizeof(int) * nt, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(embed_tile_kernel, dim3((dim / 4 + 255) / 256, nt), dim3(256), 0, c->stream, b.X, w->token_embedding_table, (const int*)b.toks, nt, dim);
        LAUNCHCHK();
        rc = run_layers_batched(c, cfg, w, b, nt, p0, s->key_cache, s->value_cache, false, p0 + nt, &nslab);
        if (rc) return rc;
    }
    // the last position's residual stream (+ the last W2 product), then infer.rs:49-51 for it only
    // (generate() ignores the logits of the forced positions before it)
    hipLaunchKernelGGL(untile_fold_kernel, dim3((dim / 4 + 255) / 256, 1), dim3(256), 0, c->stream, s->x, (const float*)b.X, last_nt - 1, dim,
                       (const float*)b.SL, nslab, b.slab);
    LAUNCHCHK();
    return launch_rows<true, EPI_STORE>(c, s->logits, w->wcls, s->x, w->rms_final_weight, dim, cfg->vocab_size);
}

// One pass for the n_seq sequences of the device tables b.toks / b.seqs: embedding rows, every layer, final norm and
// the classifier as one more GEMM into b.LG [n_seq, vocab] (row-major).  tmax bounds the longest context (score buffers).
static int enqueue_batch_pass(rama_ctx* c, const rama_config* cfg, const rama_weights* w, const BatchScratch& b, int n_seq, int tmax) {
    const int dim = cfg->dim, V = cfg->vocab_size;
    hipLaunchKernelGGL(embed_tile_kernel, dim3((dim / 4 + 255) / 256, n_seq), dim3(256), 0, c->stream, b.X, w->token_embedding_table, (const int*)b.toks, n_seq, dim);
    LAUNCHCHK();
    int nslab = 0;
    int rc = run_layers_batched(c, cfg, w, b, n_seq, 0, nullptr, nullptr, true, tmax, &nslab);
    if (rc) return rc;
    // infer.rs:49-51 for every sequence: fold the last product, final rmsnorm, classifier
    const int ntile = (n_seq + 15) / 16, pt = ntile <= 1 ? 1 : (ntile == 2 ? 2 : (ntile <= 4 ? 4 : ntile));
    rc = launch_rmsnorm_tile(c, b, w->rms_final_weight, dim, ntile, nslab); if (rc) return rc;
    MfParams p{};
    p.n_tok = n_seq; p.ksplit = 1; { const float* bs[1] = {w->wcls}; mf_weights(c, p, 1, bs, 0, 0); }
    p.x = b.XN; p.o = b.LG; p.o_stride = V; p.K = dim; p.rows = V; p.ssp = norm_ssp(c, b);
    return launch_mf<2, EPI_STORE_ROWS>(c, p, pt);
}

// ---- one decode step for up to kMfMaxTok INDEPENDENT sequences (the server's concurrent requests,
// SURVEY 8e): every weight row is streamed once for all of them.  No reference counterpart; the
// contract is "what forward(token_i, pos_i) leaves in state_i, for every i": cache rows + logits.
int rama_decode_batch(rama_ctx* c, const rama_config* cfg, const rama_weights* w, const rama_run_state* states,
                      const int32_t* tokens_host, const int32_t* pos_host, int n_seq) {
    RAMA_ENTER(c);
    REQUIRE(c && states && tokens_host && pos_host, RAMA_EINVAL, "decode_batch: NULL argument");
    REQUIRE(n_seq >= 1 && n_seq <= kMfMaxTok, RAMA_EINVAL, "decode_batch: 1..128 sequences per call");
    int rc = check_cfg(cfg); if (rc) return rc;
    rama_stage st{0, cfg->n_layers, 1, 1};
    if (set_device(c)) return 1;
    if (c->tune_tiled && !c->tune_ref_order && n_seq > kMfMaxTokRows) { rc = rama_internal_model_ensure(c, w, 2); if (rc) return rc; }
    REQUIRE(n_seq <= kMfMaxTokRows || c->tune_ref_order || mf_pass_cap(c, w, true) >= n_seq, RAMA_EINVAL,
            "decode_batch: more than 64 sequences per call need a resident model's tile-order weight copies");
    for (int i = 0; i < n_seq; i++) {
        rc = check_stage(cfg, w, &states[i], &st); if (rc) return rc;
        REQUIRE(tokens_host[i] >= 0 && tokens_host[i] < cfg->vocab_size, RAMA_EINVAL, "decode_batch: token outside the vocabulary");
        REQUIRE(pos_host[i] >= 0 && pos_host[i] < cfg->seq_len, RAMA_EINVAL, "decode_batch: position outside [0, seq_len)");
        for (int j = 0; j < i; j++) REQUIRE(states[j].key_cache != states[i].key_cache && states[j].logits != states[i].logits, RAMA_EINVAL, "decode_batch: two sequences share a run state");
    }
    const int dim = cfg->dim, V = cfg->vocab_size;
    int tmax = 1;
    for (int i = 0; i < n_seq; i++) tmax = std::max(tmax, pos_host[i] + 1);
    const size_t att_floats = (size_t)attn_scratch_floats((dim / cfg->n_heads) <= 64 ? 16 : ((dim / cfg->n_heads) <= 128 ? 32 : 64)) + tmax;
    if (c->tune_ref_order) {      // parity mode: the chain-order token-batch kernels when the shape and the copies allow
        bool done = false;
        rc = decode_batch_chain(c, cfg, w, states, tokens_host, pos_host, n_seq, &done);
        if (rc || done) return rc;
    }
    if (c->tune_ref_order || !mf_shape_ok(cfg) || V % 4 != 0 || att_floats * sizeof(float) > 64 * 1024) {   // see rama_prefill: one forward() per sequence
        for (int i = 0; i < n_seq; i++) {
            rama_run_state si = states[i];
            rc = rama_forward(c, cfg, w, &si, tokens_host[i], pos_host[i]); if (rc) return rc;
        }
        return 0;
    }
    BatchScratch b{};
    if (c->tune_tiled) { rc = rama_internal_model_ensure(c, w, 2); if (rc) return rc; }
    rc = ensure_batch_scratch(c, cfg, true, &b); if (rc) return rc;
    c->embedded_x = nullptr; c->host_pos = -1;
    // ids and the sequence table go through pinned staging (the source arrays are the caller's / locals)
    HIPCHK(hipStreamSynchronize(c->stream));
    SeqSlot* slots = reinterpret_cast<SeqSlot*>(c->pinned_tok + kMfMaxTok);
    for (int i = 0; i < n_seq; i++) {
        c->pinned_tok[i] = tokens_host[i];
        slots[i].kc = states[i].key_cache; slots[i].vc = states[i].value_cache; slots[i].pos = pos_host[i]; slots[i].pad = 0;
    }
    HIPCHK(hipMemcpyAsync(b.toks, c->pinned_tok, sizeof(int) * n_seq, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(b.seqs, slots, sizeof(SeqSlot) * n_seq, hipMemcpyHostToDevice, c->stream));
    rc = enqueue_batch_pass(c, cfg, w, b, n_seq, tmax);
    if (rc) return rc;
    for (int i = 0; i < n_seq; i++)
        HIPCHK(hipMemcpyAsync(states[i].logits, b.LG + (size_t)i * V, sizeof(float) * V, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

// ---- the same pass CHAINED ON THE DEVICE: every sequence's (token, position) lives in device memory, a step ends with
// one argmax per sequence that writes the next token and advances the position, and a step is one hipGraph replay --
// no per-sequence download, no host round trip per step (rama_decode_batch costs one call and n_seq 4-byte downloads
// per step when the tokens are fed back through the host).
struct BatchArgmaxParams { const float* logits; int n; int* toks; SeqSlot* seqs; int* out; int out_cap; int* ring; };      // ring: host-visible copy of out, token + 1 (0: not yet)
__global__ __launch_bounds__(1024) void argmax_batch_kernel(BatchArgmaxParams p) {
    // Device::sample at temperature 0 per sequence (cpu.rs:163-167: the LAST maximal index), then mod.rs:196-203:
    // token = next, pos += 1
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lg = p.logits + (size_t)b * p.n;
    float bv = -INFINITY; int bi = -1;
    const int n4 = p.n >> 2;                                      // rows of the logits slab are 16-byte aligned (n % 4 == 0)
    const f4* l4 = reinterpret_cast<const f4*>(lg);
    for (int i0 = tid; i0 < n4; i0 += 8 * 1024) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * 1024; v[u] = i < n4 ? l4[i] : f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY}; }
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
        idx = idx < 0 ? 0 : idx;
        p.toks[b] = idx;
        p.seqs[b].pos += 1;
        const int k = p.seqs[b].pad;                              // tokens this sequence has produced so far
        if (k < p.out_cap) {
            p.out[(size_t)b * p.out_cap + k] = idx;
            if (p.ring) __hip_atomic_store(p.ring + (size_t)b * p.out_cap + k, idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        p.seqs[b].pad = k + 1;
    }
}

int rama_decode_batch_begin(rama_ctx* c, const rama_config* cfg, const rama_weights* w, const rama_run_state* states,
                            const int32_t* tokens_host, const int32_t* pos_host, int n_seq, int max_steps) {
    RAMA_ENTER(c);
    REQUIRE(c && states && tokens_host && pos_host, RAMA_EINVAL, "decode_batch_begin: NULL argument");
    REQUIRE(n_seq >= 1 && n_seq <= kMfMaxTok, RAMA_EINVAL, "decode_batch_begin: 1..128 sequences");
    REQUIRE(max_steps >= 1 && max_steps <= (1 << 20), RAMA_EINVAL, "decode_batch_begin: bad max_steps");
    int rc = check_cfg(cfg); if (rc) return rc;
    REQUIRE(mf_shape_ok(cfg) && cfg->vocab_size % 4 == 0 && !c->tune_ref_order, RAMA_EUNSUP,
            "decode_batch_begin: needs dim, hidden_dim multiples of 16, vocab_size a multiple of 4, fast mode");
    rama_stage st{0, cfg->n_layers, 1, 1};
    if (set_device(c)) return 1;
    if (c->tune_tiled && n_seq > kMfMaxTokRows) { rc = rama_internal_model_ensure(c, w, 2); if (rc) return rc; }
    REQUIRE(n_seq <= kMfMaxTokRows || mf_pass_cap(c, w, true) >= n_seq, RAMA_EINVAL,
            "decode_batch_begin: more than 64 sequences need a resident model's tile-order weight copies");
    int pmax = 0;
    for (int i = 0; i < n_seq; i++) {
        rc = check_stage(cfg, w, &states[i], &st); if (rc) return rc;
        REQUIRE(tokens_host[i] >= 0 && tokens_host[i] < cfg->vocab_size, RAMA_EINVAL, "decode_batch_begin: token outside the vocabulary");
        REQUIRE(pos_host[i] >= 0 && pos_host[i] + max_steps <= cfg->seq_len, RAMA_EINVAL, "decode_batch_begin: position + max_steps beyond seq_len");
        for (int j = 0; j < i; j++) REQUIRE(states[j].key_cache != states[i].key_cache, RAMA_EINVAL, "decode_batch_begin: two sequences share a run state");
        pmax = std::max(pmax, pos_host[i]);
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    auto& bc = c->bc;
    drop_graph(c);
    if (!bc.toks) { HIPCHK(hipMalloc(&bc.toks, sizeof(int) * kMfMaxTok)); HIPCHK(hipMalloc(&bc.seqs, sizeof(SeqSlot) * kMfMaxTok)); }
    if (bc.out_cap < max_steps) {
        hipFree(bc.out); bc.out = nullptr;
        if (bc.ring) { hipHostFree(bc.ring); bc.ring = nullptr; }
        HIPCHK(hipMalloc(&bc.out, sizeof(int) * (size_t)kMfMaxTok * max_steps));
        HIPCHK(hipHostMalloc(&bc.ring, sizeof(int) * (size_t)kMfMaxTok * max_steps, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&bc.ring_dev), bc.ring, 0));
        bc.out_cap = max_steps;
    }
    memset(bc.ring, 0, sizeof(int) * (size_t)kMfMaxTok * bc.out_cap);      // (the stream was drained above: nothing is on its way)
    SeqSlot* slots = reinterpret_cast<SeqSlot*>(c->pinned_tok + kMfMaxTok);
    for (int i = 0; i < n_seq; i++) {
        c->pinned_tok[i] = tokens_host[i];
        slots[i].kc = states[i].key_cache; slots[i].vc = states[i].value_cache; slots[i].pos = pos_host[i]; slots[i].pad = 0;
    }
    HIPCHK(hipMemcpyAsync(bc.toks, c->pinned_tok, sizeof(int) * n_seq, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(bc.seqs, slots, sizeof(SeqSlot) * n_seq, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    bc.n_seq = n_seq; bc.pos_max = pmax; bc.steps_done = 0; bc.cfg = *cfg; bc.w = *w;
    if (c->tune_tiled) { rc = rama_internal_model_ensure(c, w, 2); if (rc) return rc; }
    BatchScratch b{};
    return ensure_batch_scratch(c, cfg, true, &b);
}

int rama_decode_batch_steps(rama_ctx* c, int n_steps) {
    RAMA_ENTER(c);
    REQUIRE(c && c->bc.n_seq > 0, RAMA_EINVAL, "decode_batch_steps: call rama_decode_batch_begin first");
    REQUIRE(n_steps >= 0 && c->bc.steps_done + n_steps <= c->bc.out_cap, RAMA_EINVAL, "decode_batch_steps: more steps than rama_decode_batch_begin allowed for");
    if (set_device(c)) return 1;
    auto& bc = c->bc;
    const rama_config* cfg = &bc.cfg;
    BatchScratch b{};
    int rc = ensure_batch_scratch(c, cfg, true, &b); if (rc) return rc;
    b.toks = bc.toks; b.seqs = bc.seqs;                           // the chain's own cursors, not the scratch of a single pass
    c->embedded_x = nullptr; c->host_pos = -1;
    for (int i = 0; i < n_steps; i++) {
        // the score buffers are sized by the longest context: one graph per bucket of 256 timesteps
        const int tmax = bc.pos_max + 1, bucket = (tmax + 255) / 256;
        BatchArgmaxParams ap{b.LG, cfg->vocab_size, bc.toks, bc.seqs, bc.out, bc.out_cap, bc.ring_dev};
        if (!c->graph_mode) {
            rc = enqueue_batch_pass(c, cfg, &bc.w, b, bc.n_seq, bucket * 256); if (rc) return rc;
            hipLaunchKernelGGL(argmax_batch_kernel, dim3(bc.n_seq), dim3(1024), 0, c->stream, ap);
            LAUNCHCHK();
        } else {
            if (bc.graph_bucket != bucket) {
                if (bc.exec) { hipGraphExecDestroy(bc.exec); bc.exec = nullptr; }
                if (bc.graph) { hipGraphDestroy(bc.graph); bc.graph = nullptr; }
                HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                rc = enqueue_batch_pass(c, cfg, &bc.w, b, bc.n_seq, bucket * 256);
                if (!rc) { hipLaunchKernelGGL(argmax_batch_kernel, dim3(bc.n_seq), dim3(1024), 0, c->stream, ap); }
                const hipError_t e = hipStreamEndCapture(c->stream, &bc.graph);
                if (rc) return rc;
                HIPCHK(e);
                HIPCHK(hipGraphInstantiate(&bc.exec, bc.graph, nullptr, nullptr, 0));
                bc.graph_bucket = bucket;
            }
            HIPCHK(hipGraphLaunch(bc.exec, c->stream));
        }
        bc.pos_max++; bc.steps_done++;
    }
    return 0;
}

int rama_decode_batch_tokens(rama_ctx* c, int32_t* out_host, int max_per_seq, int* n_per_seq) {
    RAMA_ENTER(c);
    REQUIRE(c && out_host && n_per_seq && c->bc.n_seq > 0, RAMA_EINVAL, "decode_batch_tokens: bad argument");
    auto& bc = c->bc;
    const int n = std::min(bc.steps_done, max_per_seq);
    HIPCHK(hipStreamSynchronize(c->stream));
    { const int rh = handoff_check(c); if (rh) return rh; }
    for (int s_ = 0; s_ < bc.n_seq && n > 0; s_++)
        HIPCHK(hipMemcpy(out_host + (size_t)s_ * max_per_seq, bc.out + (size_t)s_ * bc.out_cap, sizeof(int) * n, hipMemcpyDeviceToHost));
    *n_per_seq = n;
    return 0;
}

// tokens `from`.. sequence `seq` of the chained batch has produced so far, without touching the stream (the host-visible
// ring of rama_decode_stream_poll, one row per sequence): a server hands each request its tokens as they appear
int rama_decode_batch_stream_poll(rama_ctx* c, int seq, int from, int32_t* out_host, int max_tokens, int* n_ready) {
    RAMA_ENTER(c);
    REQUIRE(c && n_ready && c->bc.n_seq > 0 && c->bc.ring && seq >= 0 && seq < c->bc.n_seq && from >= 0 && max_tokens >= 0 && (max_tokens == 0 || out_host),
            RAMA_EINVAL, "decode_batch_stream_poll: bad argument");
    const int* row = c->bc.ring + (size_t)seq * c->bc.out_cap;
    int n = 0;
    while (n < max_tokens && from + n < c->bc.out_cap) {
        const int v = __atomic_load_n(row + from + n, __ATOMIC_ACQUIRE);
        if (v == 0) break;
        out_host[n++] = v - 1;
    }
    *n_ready = n;
    return 0;
}

// ---- device-chained greedy decode (generate() at T == 0, mod.rs:169-206)

// a new generation: nothing of the old one may still be on its way to the ring, then the used part is cleared
static int ring_reset(rama_ctx* c) {
    if (c->ring_hi > 0) {
        HIPCHK(hipStreamSynchronize(c->stream));
        memset(c->ring, 0, sizeof(int) * (size_t)std::min(c->ring_hi, c->out_cap));
        c->ring_hi = 0;
    }
    return 0;
}

int rama_decode_begin(rama_ctx* c, int token, int pos, const int32_t* forced_host, int n_forced) {
    RAMA_ENTER(c);
    REQUIRE(c, RAMA_EINVAL, "ctx is NULL");
    { int rr = ring_reset(c); if (rr) return rr; }
    REQUIRE(n_forced >= 0 && n_forced <= c->forced_cap, RAMA_EINVAL, "decode_begin: too many forced tokens");
    REQUIRE(n_forced == 0 || forced_host, RAMA_EINVAL, "decode_begin: forced list is NULL");
    if (n_forced) {
        HIPCHK(hipMemcpyAsync(c->forced, forced_host, sizeof(int) * n_forced, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    hipLaunchKernelGGL(set_ctl_kernel, dim3(1), dim3(1), 0, c->stream, c->ctl, token, pos, n_forced, 0);
    LAUNCHCHK();
    c->embedded_x = nullptr;   // the first decode step must gather x = emb[token] itself
    c->host_pos = pos;         // the chained loop advances pos by one per step: the host can mirror it
    return 0;
}

// One chained step: layers + classifier + (argmax, cursor advance, next token's embedding
// gather).  x already holds emb[token] on entry (rama_decode_steps primes it once).
static int enqueue_decode_step(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s) {
    rama_stage st{0, cfg->n_layers, 0, 1};
    c->fused_chained = true; c->fused_epoch_owed = false;
    int rc = enqueue_stage(c, cfg, w, s, &st);
    c->fused_chained = false;
    if (rc) return rc;
    ArgmaxParams ap{};
    ap.logits = s->logits; ap.n = cfg->vocab_size;
    ap.ctl = c->ctl; ap.forced = c->forced; ap.out = c->out; ap.out_cap = c->out_cap; ap.ring = c->ring_dev;
    ap.emb = w->token_embedding_table; ap.x = s->x; ap.dim = cfg->dim;
    if (c->fused_epoch_owed) ap.epoch = c->fused_epoch;
    rc = enqueue_sample(c, ap, c->samp_T, c->samp_topp, c->samp_u);
    if (rc && c->fused_epoch_owed) {      // the stage launch is enqueued and counted on the sampler to advance the epoch: do it here, or the next
        hipLaunchKernelGGL(fused_epoch_kernel, dim3(1), dim3(1), 0, c->stream, c->fused_epoch);      // launch would take this one's tagged vectors for its own
        (void)hipGetLastError();
    }
    c->fused_epoch_owed = false;
    return rc;
}

static bool same_capture(const GraphCache& g, const rama_config* cfg, const rama_weights* w, const rama_run_state* s) {
    return g.valid && g.copies_gen == rama_internal_copies_generation() && !memcmp(&g.cfg, cfg, sizeof *cfg) && !memcmp(&g.w, w, sizeof *w) && !memcmp(&g.s, s, sizeof *s);
}

int rama_decode_steps(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s, int n_steps) {
    RAMA_ENTER(c);
    REQUIRE(c, RAMA_EINVAL, "ctx is NULL");
    if (set_device(c)) return 1;
    int rc = check_cfg(cfg); if (rc) return rc;
    rama_stage st{0, cfg->n_layers, 1, 1};
    rc = check_stage(cfg, w, s, &st); if (rc) return rc;
    REQUIRE(n_steps >= 0, RAMA_EINVAL, "decode_steps: n_steps < 0");
    if (n_steps == 0) return 0;
    if (c->embedded_x != s->x) {   // infer.rs:13 for the first step; later steps get it from the argmax kernel
        hipLaunchKernelGGL(embed_kernel, dim3((cfg->dim + 255) / 256), dim3(256), 0, c->stream, s->x, w->token_embedding_table, (const Ctl*)c->ctl, 0, cfg->dim);
        LAUNCHCHK();
        c->embedded_x = s->x;
    }
    REQUIRE(c->host_pos >= 0, RAMA_EINVAL, "decode_steps: call rama_decode_begin first");
    rc = ensure_attn_part(c, cfg); if (rc) return rc;
    if (c->tune_ref_order && c->tune_chain) { rc = ensure_chain_copy(c, cfg, w, nullptr); if (rc) return rc; }
    if (c->samp_T != 0.0f) { rc = ensure_topp_scratch(c, cfg->vocab_size); if (rc) return rc; c->topp_dist_dirty = true; }      // (a replayed graph enqueues the pick without passing enqueue_sample)
    REQUIRE(c->host_pos + n_steps <= cfg->seq_len, RAMA_EINVAL, "decode_steps: would run past seq_len");
    const bool graphs = c->graph_mode && c->kp.kernel_id < 0;
    auto variant_at = [&](int pos) { return attn_variant(c, cfg, pos); };
    for (int i = 0; i < n_steps;) {
        // the attention variant depends on the position, which the host mirrors step by step
        const int v = apply_attn_variant(c, cfg, c->host_pos);
        int take = 1;
        if (graphs) {
            const int M = c->tune_graph_steps > 0 ? c->tune_graph_steps : (cfg->dim <= 1024 ? 4 : 1);
            if (M > 1 && n_steps - i >= M && variant_at(c->host_pos + M - 1) == v) take = M;   // the variant changes at most once, monotonically
            GraphCache& g = c->gc[v + (take > 1 ? 4 : 0)];
            if (!same_capture(g, cfg, w, s) || g.steps != take) {
                if (g.exec) hipGraphExecDestroy(g.exec);
                if (g.graph) hipGraphDestroy(g.graph);
                g = GraphCache();
                const bool dirty_before = c->handoff_dirty;
                c->handoff_dirty = false;
                HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                for (int k = 0; k < take && !rc; k++) rc = enqueue_decode_step(c, cfg, w, s);
                hipError_t e = hipStreamEndCapture(c->stream, &g.graph);
                g.handoff = c->handoff_dirty;
                c->handoff_dirty = dirty_before;
                if (rc) return rc;
                HIPCHK(e);
                HIPCHK(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
                g.cfg = *cfg; g.w = *w; g.s = *s; g.valid = true; g.copies_gen = rama_internal_copies_generation(); g.steps = take;
            }
            HIPCHK(hipGraphLaunch(g.exec, c->stream));
            if (g.handoff) c->handoff_dirty = true;
        } else {
            rc = enqueue_decode_step(c, cfg, w, s);
            if (rc) return rc;
        }
        c->host_pos += take;
        i += take;
    }
    c->ring_hi = std::min(c->out_cap, c->ring_hi + n_steps);
    return 0;
}

// Tokens the chained loop has produced so far, WITHOUT touching the stream: entries `from`.. of the host-visible ring the
// sampling launch writes next to its device list (generate_stream's channel, mod.rs:209-248: each token as it is produced).
// Returns at once; *n_ready may be 0.  Errors of the loop itself are reported by rama_decode_tokens at the end.
int rama_decode_stream_poll(rama_ctx* c, int from, int32_t* out_host, int max_tokens, int* n_ready) {
    RAMA_ENTER(c);
    REQUIRE(c && n_ready && from >= 0 && max_tokens >= 0 && (max_tokens == 0 || out_host), RAMA_EINVAL, "decode_stream_poll: bad argument");
    int n = 0;
    while (n < max_tokens && from + n < c->out_cap) {
        const int v = __atomic_load_n(c->ring + from + n, __ATOMIC_ACQUIRE);
        if (v == 0) break;
        out_host[n++] = v - 1;
    }
    *n_ready = n;
    return 0;
}

int rama_decode_tokens(rama_ctx* c, int32_t* out_host, int max_tokens, int* n_out) {
    RAMA_ENTER(c);
    REQUIRE(c && n_out, RAMA_EINVAL, "decode_tokens: NULL argument");
    Ctl h;
    HIPCHK(hipMemcpyAsync(&h, c->ctl, sizeof h, hipMemcpyDeviceToHost, c->stream));
    unsigned long long perr = 0;
    HIPCHK(hipMemcpyAsync(&perr, c->pbar + 1, sizeof perr, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->handoff_dirty = false;
    if (perr != 0) {   // a bounded wait inside a launch gave up (attn_wo.hpp's counter, layer_fused.hpp's tagged vectors: 0x3000 / 0x3001): the results are invalid
        hipMemsetAsync(c->pbar, 0, 4 * sizeof(unsigned long long), c->stream);   // counter, error word, base: start over
        return fail(RAMA_EINVAL, perr >= kFusedErr ? "one-launch stage: a hand-off timed out" : "attention+Wo launch: hand-off timed out", __FILE__, __LINE__);
    }
    { const int rt = topp_dist_check(c); if (rt) return rt; }
    if (c->topp_err) {
        unsigned terr = 0;
        HIPCHK(hipMemcpy(&terr, c->topp_err, sizeof terr, hipMemcpyDeviceToHost));
        if (terr) {
            hipMemset(c->topp_err, 0, sizeof terr);
            return fail(RAMA_EINVAL, "top-p sampler: no probability above the cutoff (the reference underflows here, infer.rs:56-84)", __FILE__, __LINE__);
        }
    }
    int n = std::min(std::min(h.n_out, c->out_cap), max_tokens);
    if (n > 0 && out_host) {
        HIPCHK(hipMemcpyAsync(out_host, c->out, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    *n_out = n;
    return 0;
}

int rama_generate_greedy(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s,
                         const int32_t* prompt_host, int n_prompt, int steps, int32_t* out_host) {
    RAMA_ENTER(c);
    return rama_generate(c, cfg, w, s, prompt_host, n_prompt, steps, 0.0f, 0.9f, 0.0f, out_host);
}

// enqueue the whole generate() loop (mod.rs:169-206); the caller collects the tokens (all at the end, or as they appear)
// (*deferred != NULL: the decode steps are NOT enqueued, their count is returned there -- the streaming caller feeds them
// in pieces, see rama_generate_stream)
static int generate_enqueue(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s,
                            const int32_t* prompt_host, int n_prompt, int steps, float temperature, float topp, float u,
                            int* deferred = nullptr) {
    REQUIRE(c && cfg, RAMA_EINVAL, "generate_greedy: NULL argument");
    { int rcs = rama_decode_sampler(c, temperature, topp, u); if (rcs) return rcs; }
    REQUIRE(steps >= 0 && steps <= cfg->seq_len, RAMA_EINVAL, "generate_greedy: steps > seq_len (the reference does not bound-check, SURVEY section 5)");
    REQUIRE(steps <= c->out_cap, RAMA_EINVAL, "generate_greedy: too many steps");
    REQUIRE(n_prompt >= 0 && (n_prompt == 0 || prompt_host), RAMA_EINVAL, "generate_greedy: bad prompt");
    n_prompt = std::min(n_prompt, c->forced_cap);
    int rc;
    if (c->tune_prefill && n_prompt >= 3 && steps > n_prompt) {
        // Positions 0..n_prompt carry known tokens (BOS, then the prompt; mod.rs:182-191) and their
        // `next` is forced, so they go through the layers together (rama_prefill); the logits of
        // position n_prompt give the first sampled token and the chained loop takes over.
        std::vector<int32_t> toks(n_prompt + 1);
        toks[0] = 1;
        for (int i = 0; i < n_prompt; i++) toks[i + 1] = prompt_host[i];
        for (int i = 0; i < n_prompt; i++) REQUIRE(prompt_host[i] >= 0 && prompt_host[i] < cfg->vocab_size, RAMA_EINVAL, "generate_greedy: prompt token outside the vocabulary");
        rc = ring_reset(c); if (rc) return rc;
        rc = rama_prefill(c, cfg, w, s, toks.data(), n_prompt + 1, 0);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(c->out, prompt_host, sizeof(int) * n_prompt, hipMemcpyHostToDevice, c->stream));   // forced `next`s
        for (int i = 0; i < n_prompt; i++) __atomic_store_n(c->ring + i, prompt_host[i] + 1, __ATOMIC_RELEASE);   // ... known at once
        c->ring_hi = n_prompt + 1;
        hipLaunchKernelGGL(set_ctl_kernel, dim3(1), dim3(1), 0, c->stream, c->ctl, toks[n_prompt], n_prompt, 0, n_prompt);
        LAUNCHCHK();
        ArgmaxParams ap{};
        ap.logits = s->logits; ap.n = cfg->vocab_size;
        ap.ctl = c->ctl; ap.forced = c->forced; ap.out = c->out; ap.out_cap = c->out_cap; ap.ring = c->ring_dev;
        ap.emb = w->token_embedding_table; ap.x = s->x; ap.dim = cfg->dim;
        if (c->samp_T != 0.0f) { rc = ensure_topp_scratch(c, cfg->vocab_size); if (rc) return rc; }
        rc = enqueue_sample(c, ap, c->samp_T, c->samp_topp, c->samp_u);               // out[n_prompt], cursor -> n_prompt + 1, next x
        if (rc) return rc;
        c->embedded_x = s->x;
        c->host_pos = n_prompt + 1;
        if (deferred) { *deferred = steps - n_prompt - 1; return 0; }
        rc = rama_decode_steps(c, cfg, w, s, steps - n_prompt - 1);
        if (rc) return rc;
    } else {
        rc = rama_decode_begin(c, /*BOS*/ 1, 0, prompt_host, n_prompt);
        if (rc) return rc;
        if (deferred) { *deferred = steps; return 0; }
        rc = rama_decode_steps(c, cfg, w, s, steps);
        if (rc) return rc;
    }
    return 0;
}

int rama_generate(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s,
                  const int32_t* prompt_host, int n_prompt, int steps, float temperature, float topp, float u,
                  int32_t* out_host) {
    RAMA_ENTER(c);
    REQUIRE(out_host, RAMA_EINVAL, "generate_greedy: NULL argument");
    int rc = generate_enqueue(c, cfg, w, s, prompt_host, n_prompt, steps, temperature, topp, u);
    if (rc) return rc;
    int n = 0;
    return rama_decode_tokens(c, out_host, steps, &n);
}

// generate_stream (mod.rs:209-248): the same loop, every token handed to `on_token(user, index, token)` on the calling
// thread as soon as the device has produced it -- the loop itself runs on, chained on the device; the host only watches
// the ring.  out_host (may be NULL) receives the whole list at the end, as rama_generate does.
int rama_generate_stream(rama_ctx* c, const rama_config* cfg, const rama_weights* w, rama_run_state* s,
                         const int32_t* prompt_host, int n_prompt, int steps, float temperature, float topp, float u,
                         void (*on_token)(void* user, int index, int32_t token), void* user, int32_t* out_host) {
    RAMA_ENTER(c);
    REQUIRE(on_token, RAMA_EINVAL, "generate_stream: NULL callback");
    int todo = 0;
    int rc = generate_enqueue(c, cfg, w, s, prompt_host, n_prompt, steps, temperature, topp, u, &todo);
    if (rc) return rc;
    // The steps are fed to the stream a few at a time and never more than kAhead beyond the last token seen: a whole
    // generation enqueued at once fills the hardware queue (200 steps of llama2-7B are 32 000 packets), the enqueueing
    // thread then sits in hipGraphLaunch and the first token is looked at half a generation late.
    constexpr int kChunk = 8, kAhead = 24;
    int seen = 0, fed = steps - todo;                             // tokens handed over; tokens whose production is enqueued
    auto hand_over = [&]() -> int {
        int32_t buf[64];
        int n = 0;
        int r = rama_decode_stream_poll(c, seen, buf, std::min(64, steps - seen), &n); if (r) return r;
        for (int i = 0; i < n; i++) on_token(user, seen + i, buf[i]);
        seen += n;
        return 0;
    };
    while (seen < steps) {
        if (fed < steps && fed - seen < kAhead) {
            const int n = std::min(kChunk, steps - fed);
            rc = rama_decode_steps(c, cfg, w, s, n); if (rc) return rc;
            fed += n;
        }
        const int before = seen;
        rc = hand_over(); if (rc) return rc;
        if (seen == before && fed == steps) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); continue; }
            if (q != hipSuccess) return fail((int)q, "generate_stream: the stream failed while tokens were outstanding", __FILE__, __LINE__);
            rc = hand_over(); if (rc) return rc;                  // everything has run: what is there now is all there will be
            if (seen == before) break;
        }
    }
    std::vector<int32_t> tmp;
    if (!out_host) { tmp.resize(steps > 0 ? steps : 1); out_host = tmp.data(); }
    int n = 0;
    rc = rama_decode_tokens(c, out_host, steps, &n);                 // also reports what went wrong inside the loop, if anything
    if (rc) return rc;
    REQUIRE(seen == steps, RAMA_EINVAL, "generate_stream: the loop ended before every token had appeared");
    return 0;
}

int rama_set_tuning(rama_ctx* c, const char* key, int value) {
    RAMA_ENTER(c);
    REQUIRE(c && key, RAMA_EINVAL, "set_tuning: NULL argument");
    if (!strcmp(key, "split_pos")) {
        REQUIRE(value >= -1, RAMA_EINVAL, "set_tuning: split_pos must be >= -1");
        c->tune_split_pos = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "resid_r2")) {
        REQUIRE(value >= 0 && value <= 3, RAMA_EINVAL, "set_tuning: resid_r2 must be 0..3");
        c->tune_resid_r2 = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "prefill")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: prefill must be 0 or 1");
        c->tune_prefill = value;
        return 0;
    }
    if (!strcmp(key, "fused") || !strcmp(key, "fused_solo")) {
        REQUIRE(value >= -1 && value <= 1, RAMA_EINVAL, "set_tuning: fused / fused_solo must be -1, 0 or 1");
        if (key[5]) c->tune_fused_solo = value; else c->tune_fused = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "merge")) {
        REQUIRE(value >= -1 && value <= 1, RAMA_EINVAL, "set_tuning: merge must be -1, 0 or 1");
        c->tune_merge = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "attn_nsplit") || !strcmp(key, "attn_nt")) {
        const bool ns = !strcmp(key, "attn_nsplit");
        REQUIRE(value >= 0 && value <= (ns ? 32 : 1), RAMA_EINVAL, "set_tuning: attn_nsplit must be 0..32, attn_nt 0 or 1");
        HIPCHK(hipStreamSynchronize(c->stream));
        drop_graph(c);
        if (ns) { c->tune_attn_nsplit = value; if (c->attn_part) { hipFree(c->attn_part); c->attn_part = nullptr; c->attn_part_floats = 0; } }
        else c->tune_attn_nt = value;
        return 0;
    }
    if (!strcmp(key, "small_attn_waves") || !strcmp(key, "small_attn_pos")) {
        const bool wv = !strcmp(key, "small_attn_waves");
        REQUIRE(wv ? (value == 4 || value == 8) : value >= 0, RAMA_EINVAL, "set_tuning: small_attn_waves must be 4 or 8, small_attn_pos >= 0");
        HIPCHK(hipStreamSynchronize(c->stream));
        drop_graph(c);
        if (wv) c->tune_small_waves = value; else c->tune_small_pos = value;
        return 0;
    }
    if (!strcmp(key, "combine_v")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: combine_v must be 0 or 1");
        HIPCHK(hipStreamSynchronize(c->stream));
        drop_graph(c);
        c->tune_combine_v = value;
        return 0;
    }
    if (!strcmp(key, "norm_in_gemm")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: norm_in_gemm must be 0 or 1");
        c->tune_norm_in_gemm = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "tiled")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: tiled must be 0 or 1");
        c->tune_tiled = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "chain_lead")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: chain_lead must be 0 or 1");
        c->tune_chain_lead = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "qkv_fold")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: qkv_fold must be 0 or 1");
        c->tune_qkv_fold = value;
        return 0;
    }
    if (!strcmp(key, "resid_fold")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: resid_fold must be 0 or 1");
        c->tune_resid_fold = value;
        return 0;
    }
    if (!strcmp(key, "norm_fold")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: norm_fold must be 0 or 1");
        c->tune_norm_fold = value;
        return 0;
    }
    if (!strcmp(key, "chain_split")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: chain_split must be 0 or 1");
        c->tune_chain_split = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "chain_lead_w")) {
        REQUIRE(value >= 0 && value <= 2, RAMA_EINVAL, "set_tuning: chain_lead_w must be 0, 1 or 2");
        c->tune_chain_lead_w = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "chain_resid_d")) {
        REQUIRE(value == 0 || value == -1 || ((value / 100 == 1 || value / 100 == 2 || value / 100 == 4) && (value % 100 == 16 || value % 100 == 32)), RAMA_EINVAL, "set_tuning: chain_resid_d must be -1, 0 or 100 W + D, W in {1, 2, 4}, D in {16, 32}");
        c->tune_chain_resid_d = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "ew_batch")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: ew_batch must be 0 or 1");
        c->tune_ew_batch = value;
        return 0;
    }
    if (!strcmp(key, "matmul_batch")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: matmul_batch must be 0 or 1");
        c->tune_matmul_batch = value;
        return 0;
    }
    if (!strcmp(key, "rope_batch")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: rope_batch must be 0 or 1");
        c->tune_rope_batch = value;
        return 0;
    }
    if (!strcmp(key, "chain_views")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: chain_views must be 0 or 1");
        c->tune_chain_views = value;
        return 0;
    }
    if (!strcmp(key, "chain_norm")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: chain_norm must be 0 or 1");
        c->tune_chain_norm = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "prefill_chain")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: prefill_chain must be 0 or 1");
        c->tune_prefill_chain = value;
        return 0;
    }
    if (!strcmp(key, "prefill_tok")) {
        REQUIRE(value == 64 || value == 128, RAMA_EINVAL, "set_tuning: prefill_tok must be 64 or 128");
        c->tune_prefill_tok = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "prefill_attn")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: prefill_attn must be 0 or 1");
        c->tune_prefill_attn = value;
        return 0;
    }
    if (!strcmp(key, "graph_steps")) {
        REQUIRE(value == -1 || (value >= 1 && value <= 32), RAMA_EINVAL, "set_tuning: graph_steps must be -1 or 1..32");
        HIPCHK(hipStreamSynchronize(c->stream));
        drop_graph(c);
        c->tune_graph_steps = value;
        return 0;
    }
    if (!strcmp(key, "attn_u")) {
        REQUIRE(value == 8 || value == 16, RAMA_EINVAL, "set_tuning: attn_u must be 8 or 16");
        HIPCHK(hipStreamSynchronize(c->stream));
        drop_graph(c);
        c->tune_attn_u = value;
        return 0;
    }
    if (!strcmp(key, "attn_waves")) {
        REQUIRE(value == 16 || value == 8 || value == 4, RAMA_EINVAL, "set_tuning: attn_waves must be 16, 8 or 4");
        HIPCHK(hipStreamSynchronize(c->stream));
        drop_graph(c);
        c->tune_attn_waves = value;
        return 0;
    }
    if (!strcmp(key, "small_attn")) {
        REQUIRE(value >= -1 && value <= 1, RAMA_EINVAL, "set_tuning: small_attn must be -1, 0 or 1");
        c->tune_small_attn = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "topp_block")) {
        REQUIRE(value == 512 || value == 1024 || value == 2048, RAMA_EINVAL, "set_tuning: topp_block must be 512, 1024 or 2048");
        c->tune_topp_block = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "spread_pos")) {
        REQUIRE(value >= 64 && value <= (1 << 20), RAMA_EINVAL, "set_tuning: spread_pos must be 64 .. 2^20");
        c->tune_spread_pos = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "attn_fv")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: attn_fv must be 0 or 1");
        c->tune_attn_fv = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "topp_sort") || !strcmp(key, "topp_keep_sums") || !strcmp(key, "topp_pairs") || !strcmp(key, "topp_dist")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: topp_sort / topp_keep_sums / topp_pairs / topp_dist must be 0 or 1");
        if (!strcmp(key, "topp_sort")) c->tune_topp_sort = value; else if (!strcmp(key, "topp_pairs")) c->tune_topp_pairs = value;
        else if (!strcmp(key, "topp_dist")) c->tune_topp_dist = value; else c->tune_topp_keep_sums = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "w13i")) {
        REQUIRE(value == 0 || value == 1, RAMA_EINVAL, "set_tuning: w13i must be 0 or 1");
        c->tune_w13i = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "solo")) {
        REQUIRE(value >= -1 && value <= 1, RAMA_EINVAL, "set_tuning: solo must be -1, 0 or 1");
        c->tune_solo = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "ref_order")) {
        REQUIRE(value >= 0 && value <= 3, RAMA_EINVAL, "set_tuning: ref_order must be 0, 1, 2 or 3");
        c->tune_ref_order = value != 0; c->tune_tol = value == 2; c->tune_bar = value == 3;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "bar_pos")) {
        REQUIRE(value >= 0, RAMA_EINVAL, "set_tuning: bar_pos must be >= 0");
        c->tune_bar_pos = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "lane_reduce")) {
        REQUIRE(value >= 0 && value <= 2, RAMA_EINVAL, "set_tuning: lane_reduce must be 0 (pairwise), 1 (strided) or 2 (sequential)");
        c->tune_lane_reduce = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "tol_mask")) {
        REQUIRE(value >= 0 && value < 128, RAMA_EINVAL, "set_tuning: tol_mask must be 0..127");
        c->tune_tol_mask = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "chain") || !strcmp(key, "chain_d")) {
        const bool cd = !strcmp(key, "chain_d");
        REQUIRE(cd ? (value == 0 || (value >= 116 && value <= 432)) : (value == 0 || value == 1), RAMA_EINVAL, "set_tuning: chain must be 0 or 1, chain_d 0 or 100 W + D");
        if (cd) c->tune_chain_d = value; else c->tune_chain = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    if (!strcmp(key, "geom")) {
        REQUIRE(value >= 0 && value <= 4, RAMA_EINVAL, "set_tuning: geom must be 0..4");
        c->tune_geom = value;
        hipStreamSynchronize(c->stream);
        drop_graph(c);
        return 0;
    }
    return fail(RAMA_EINVAL, "set_tuning: unknown key", __FILE__, __LINE__);
}

int rama_set_graph_mode(rama_ctx* c, int enabled) {
    RAMA_ENTER(c);
    REQUIRE(c, RAMA_EINVAL, "ctx is NULL");
    c->graph_mode = enabled != 0;
    if (!enabled) { hipStreamSynchronize(c->stream); drop_graph(c); }
    return 0;
}

// ---------------------------------------------------------------- measurement

int rama_timer_start(rama_ctx* c) {
    RAMA_ENTER(c);
    REQUIRE(c, RAMA_EINVAL, "ctx is NULL");
    HIPCHK(hipEventRecord(c->t0, c->stream));
    return 0;
}
int rama_timer_stop(rama_ctx* c, float* ms) {
    RAMA_ENTER(c);
    REQUIRE(c && ms, RAMA_EINVAL, "timer_stop: NULL argument");
    HIPCHK(hipEventRecord(c->t1, c->stream));
    HIPCHK(hipEventSynchronize(c->t1));
    HIPCHK(hipEventElapsedTime(ms, c->t0, c->t1));
    return 0;
}

int rama_kprof_enable(rama_ctx* c, int kernel_id, int max_records) {
    RAMA_ENTER(c);
    REQUIRE(c && kernel_id >= 0 && kernel_id < RAMA_K_COUNT && max_records > 0, RAMA_EINVAL, "kprof_enable: bad argument");
    KProf& k = c->kp;
    while ((int)k.ev.size() < 2 * max_records) {
        hipEvent_t e; HIPCHK(hipEventCreate(&e)); k.ev.push_back(e);
    }
    k.kernel_id = kernel_id; k.max_records = max_records; k.used = 0;
    return 0;
}
int rama_kprof_read(rama_ctx* c, int* n_launches, double* total_ms) {
    RAMA_ENTER(c);
    REQUIRE(c && n_launches && total_ms, RAMA_EINVAL, "kprof_read: NULL argument");
    KProf& k = c->kp;
    HIPCHK(hipStreamSynchronize(c->stream));
    double tot = 0.0;
    for (int i = 0; i < k.used; i++) {
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, k.ev[2 * i], k.ev[2 * i + 1]));
        tot += ms;
    }
    *n_launches = k.used; *total_ms = tot;
    k.kernel_id = -1; k.used = 0;
    return 0;
}

// ---------------------------------------------------------------- state

int rama_state_create(rama_ctx* c, const rama_config* cfg, int n_local_layers, rama_run_state* out) {
    RAMA_ENTER(c);
    REQUIRE(c && out, RAMA_EINVAL, "state_create: NULL argument");
    int rc = check_cfg(cfg); if (rc) return rc;
    REQUIRE(n_local_layers >= 0 && n_local_layers <= cfg->n_layers, RAMA_EINVAL, "state_create: bad layer count");
    // one blob, every buffer 256-byte aligned; sizes as ram.rs:7-23 (kv_dim == dim here)
    auto al = [](size_t n) { return (n + 63) & ~(size_t)63; };
    const size_t d = cfg->dim, h = cfg->hidden_dim;
    const size_t kv = (size_t)std::max(n_local_layers, 1) * cfg->seq_len * d;
    size_t sizes[12] = {d, d, d, h, h, d, d, d, (size_t)cfg->n_heads * cfg->seq_len, (size_t)cfg->vocab_size, kv, kv};
    size_t total = 0;
    for (size_t z : sizes) total += al(z);
    float* blob = nullptr;
    rc = rama_alloc_f32(c, total, &blob);
    if (rc) return rc;
    float** fields[12] = {&out->x, &out->xb, &out->xb2, &out->hb, &out->hb2, &out->q, &out->k, &out->v, &out->att, &out->logits, &out->key_cache, &out->value_cache};
    size_t off = 0;
    for (int i = 0; i < 12; i++) { *fields[i] = blob + off; off += al(sizes[i]); }
    return 0;
}

int rama_state_free(rama_ctx* c, rama_run_state* s) {
    RAMA_ENTER(c);
    REQUIRE(c && s, RAMA_EINVAL, "state_free: NULL argument");
    int rc = rama_free(c, s->x);   // x is the blob base
    memset(s, 0, sizeof *s);
    return rc;
}
