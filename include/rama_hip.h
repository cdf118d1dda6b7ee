/*
 * rama_hip.h -- C ABI of the MI355X (gfx950) backend for oliverhu/rama's decode path.
 *
 * This is the drop-in boundary: exactly what a Rust `impl Device<HipSlice> for Hip`
 * (reference trait: engine/src/device/device.rs:3-24) would bind over `extern "C"`,
 * in place of the reference's cudarc/NVRTC/cuBLAS backend (engine/src/device/gpu.rs,
 * engine/src/device/math.cu, engine/src/transformer/hbm.rs).  Plain pointers and
 * sizes only; no C++ / torch types.  See INTEGRATION.md for the Rust-side stub.
 *
 * Conventions
 *  - Every `float*` / `const float*` argument is a DEVICE pointer already offset by
 *    the caller's View.range.start (reference: `cudaview()`, gpu.rs:51-69), except
 *    where the name says `host`.
 *  - All work is enqueued on the context's HIP stream and is asynchronous unless the
 *    function copies to host memory (those synchronise, like cudarc's *_sync_* calls).
 *  - Return value: 0 = ok; > 0 = a hipError_t; < 0 = RAMA_E*.  The reference trait has
 *    no Result and unwraps every driver call (panic); a host wrapper that wants that
 *    behaviour aborts on non-zero.  rama_last_error() gives a message for the calling
 *    thread's last failure.
 *  - Thread-safety: one context = one stream; calls on one context must be serialised
 *    by the caller.  Distinct contexts / run states may be used from distinct threads
 *    over shared read-only weights (reference: one RunState per request, lib.rs:134-147).
 */
#ifndef RAMA_HIP_H
#define RAMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAMA_OK        0
#define RAMA_EINVAL   (-1)  /* bad argument (null pointer, width % 4 != 0, pos >= seq_len ...) */
#define RAMA_EUNSUP   (-2)  /* shape the reference path does not support (n_kv_heads != n_heads) */
#define RAMA_EIO      (-3)  /* checkpoint file problem */
#define RAMA_ENOMEM   (-4)

typedef struct rama_ctx rama_ctx;
typedef struct rama_model rama_model;

/* ---------------------------------------------------------------- lifecycle
 * replaces GPU::new (gpu.rs:213-234: device 0, NVRTC compile, cuBLAS handle).
 * `hip_stream` may be NULL (the context creates its own non-blocking stream) or an
 * existing hipStream_t to adopt (e.g. a framework's current stream). */
int  rama_ctx_create(int device, void *hip_stream, rama_ctx **out);
int  rama_ctx_destroy(rama_ctx *ctx);
int  rama_sync(rama_ctx *ctx);                       /* hipStreamSynchronize; also reports an in-kernel hand-off that timed out (one-launch stage, attention+Wo) */
/* Non-blocking: 0 = everything enqueued on the context's stream has run, 1 = still running, anything else = the
 * stream has failed (what a host loop polling rama_decode_stream_poll / rama_decode_batch_stream_poll ends on). */
int  rama_stream_query(rama_ctx *ctx);
const char *rama_last_error(void);
/* name (<= 63 chars), compute units, total HBM bytes of the context's device */
int  rama_device_info(rama_ctx *ctx, char name[64], int *compute_units, size_t *hbm_bytes);

/* ---------------------------------------------------------------- memory
 * replaces hbm.rs:14-16 `allocate` (= htod_sync_copy) and dtoh_sync_copy_into
 * (gpu.rs:196-209 to_cpu, hbm.rs:38-51 into_state). */
int  rama_alloc_f32(rama_ctx *ctx, size_t n, float **out);            /* zero-filled, cf. ram.rs:10-21 */
int  rama_upload_f32(rama_ctx *ctx, const float *host, size_t n, float **out);  /* alloc + H2D */
int  rama_copy_h2d_f32(rama_ctx *ctx, float *dst, const float *host, size_t n);
int  rama_download_f32(rama_ctx *ctx, const float *src, size_t n, float *host);
int  rama_free(rama_ctx *ctx, void *device_ptr);

/* ---------------------------------------------------------------- Device<T> ops, 1:1
 * engine/src/device/device.rs:3-24; arithmetic follows the CPU backend
 * (engine/src/device/cpu.rs), never math.cu (which is buggy: SURVEY.md section 2.1). */
int  rama_array_add(rama_ctx *ctx, float *target, const float *source, size_t n);   /* device.rs:4  cpu.rs:16-21 */
int  rama_array_mult(rama_ctx *ctx, float *target, const float *source, size_t n);  /* device.rs:5  cpu.rs:59-64 */
int  rama_sinu(rama_ctx *ctx, float *o, size_t n);                                   /* device.rs:6  cpu.rs:54-57 */
int  rama_copy_from_slice(rama_ctx *ctx, float *target, const float *source, size_t n); /* device.rs:9 cpu.rs:66-72 */
int  rama_rmsnorm(rama_ctx *ctx, float *o, const float *x, const float *weight, size_t n); /* device.rs:10 cpu.rs:99-117 */
/* device.rs:12 cpu.rs:74-97: one head of q and k rotated in place by (pos_real[i], pos_img[i]) */
int  rama_apply_position(rama_ctx *ctx, float *q, float *k, const float *pos_real,
                         const float *pos_img, size_t head_size);
/* device.rs:13 cpu.rs:127-153: o[r*o_cols + c] = sum_k a[r*width + k] * b[k*o_cols + c];
 * a = weight matrix [o_rows, width] row-major, b = activations.  width % 4 != 0 ->
 * RAMA_EINVAL (the reference CPU body panics there). */
int  rama_matmul(rama_ctx *ctx, float *o, const float *a, const float *b,
                 size_t width, size_t o_rows, size_t o_cols);
int  rama_softmax(rama_ctx *ctx, float *x, size_t n);                                /* device.rs:14 cpu.rs:119-125 */
/* device.rs:7-8 cpu.rs:23-52; flat argument list as math.cu:94 calculate_attention +
 * math.cu:72 update_xb take it.  key_cache/value_cache are the full [L, seq_len, dim] buffers. */
int  rama_multi_head_attention(rama_ctx *ctx, float *xb, float *att, const float *q,
                               const float *key_cache, const float *value_cache,
                               int layer, int dim, int pos, int head_size, int seq_len, int n_heads);
/* device.rs:16 Device::sample, T == 0 leg (cpu.rs:163-167): argmax on the device, ties ->
 * last maximal index; only the 4-byte result crosses PCIe (reference GPU path clones and
 * downloads all logits per token, gpu.rs:153). */
int  rama_sample_argmax(rama_ctx *ctx, const float *logits, size_t n, int32_t *next_host);
/* T != 0 leg (cpu.rs:168-177, infer.rs:55-85): temperature scale (only if T < 1), softmax,
 * top-p.  `u` = the uniform draw; the reference re-seeds ChaCha20 every call (cpu.rs:161-162,
 * gpu.rs:151-152) so its draw is one constant.  Runs on the device (rama_sample_topp_dev below);
 * only the 4-byte result crosses PCIe.  RAMA_EINVAL when no probability exceeds the cutoff. */
int  rama_sample_topp(rama_ctx *ctx, const float *logits, size_t n, float temperature,
                      float topp, float u, int32_t *next_host);

/* ---------------------------------------------------------------- model + state
 * engine/src/transformer/mod.rs:128-138 */
typedef struct {
    int32_t dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, seq_len;
    int32_t shared_weight;
} rama_config;

/* engine/src/transformer/state.rs:53-74; device pointers.  For a pipeline stage the
 * per-layer tensors hold only layers [layer_begin, layer_end) (index = layer - layer_begin);
 * tensors a stage does not own are NULL. */
typedef struct {
    const float *token_embedding_table;
    const float *rms_att_weight, *rms_ffn_weight;
    const float *wq, *wk, *wv, *wo, *w1, *w2, *w3;
    const float *rms_final_weight;
    const float *freq_cis_real, *freq_cis_imag;
    const float *wcls;            /* == token_embedding_table when shared (state.rs:111-117) */
} rama_weights;

/* engine/src/transformer/state.rs:3-17, sized as ram.rs:7-23 (caches: [n_local_layers, seq_len, dim]) */
typedef struct {
    float *x, *xb, *xb2, *hb, *hb2, *q, *k, *v, *att, *logits, *key_cache, *value_cache;
} rama_run_state;

typedef struct {
    int32_t layer_begin, layer_end;   /* this stage's layers */
    int32_t do_embed;                 /* 1: gather x from token_embedding_table (infer.rs:13) */
    int32_t do_cls;                   /* 1: final rmsnorm + classifier (infer.rs:49-51) */
} rama_stage;

/* llama2.c v0 checkpoint (header mod.rs:141-166, tensor order ram.rs:28-51): mmap the file,
 * one hipMalloc for the tensor blob, one staged H2D copy (replaces 6.7 G four-byte reads,
 * utils/read.rs:25-33, + 14 htod_sync_copy, hbm.rs:55-90). */
int  rama_model_load(rama_ctx *ctx, const char *path, rama_model **out);
/* the same for one pipeline stage: only the stage's layers (and the embedding / classifier tensors it
 * needs) are copied to this device */
int  rama_model_load_stage(rama_ctx *ctx, const char *path, const rama_stage *stage, rama_model **out);
/* Synthetic weights of a given shape, generated in HBM by an integer-hash fill kernel
 * (bit-identical to oracle_fill_synth); only layers of `stage` are materialised.
 * rope_real/rope_imag: host tables [seq_len, head_size/2] or NULL (computed here). */
int  rama_model_synth(rama_ctx *ctx, const rama_config *cfg, uint64_t seed, const rama_stage *stage,
                      const float *rope_real_host, const float *rope_imag_host, rama_model **out);
/* Write a whole model as a llama2.c v0 file -- the format engine/export/export.py:75-127
 * (legacy_export) produces and transformer/ram.rs:28-51 reads, so upstream Rama loads it too
 * (SURVEY section 8 row f2).  RAMA_EINVAL for a model that holds only a pipeline stage. */
int  rama_model_save(rama_ctx *ctx, const rama_model *model, const char *path);
/* A model also keeps W1 and W3 row-interleaved per layer (row i of W1, then row i of W3: +11.5 GB at
 * llama2-7B): the fused decode path finds that copy by the addresses of w1 / w3 in rama_weights and
 * streams ONE contiguous block per workgroup; weights uploaded tensor by tensor (rama_upload_f32) take
 * the two-tensor kernel.  rama_set_tuning "w13i" = 0 turns the lookup off.
 * And every matrix once more in MFMA tile order (+27 GB at llama2-7B; skipped when fewer than 16 GiB of
 * HBM would remain, or with RAMA_NO_TILED=1 in the environment): rama_prefill / rama_decode_batch read
 * that copy -- one contiguous 1-KiB weight read per wave -- found the same way ("tiled" = 0: off).
 * rama_model_bytes counts the checkpoint tensors only.
 * The tile-order and the chain-order copies are made on FIRST USE (rama_prefill / rama_decode_batch*; "ref_order" != 0):
 * that call allocates and synchronises the device, so it must not sit inside the caller's own stream capture --
 * a caller that captures sets RAMA_EAGER_COPIES=1 in the environment (the copies are then made by rama_model_load /
 * rama_model_synth) or makes one warm-up call first.  rama_model_release_copies gives them back. */
int  rama_model_config(const rama_model *m, rama_config *cfg);
int  rama_model_weights(const rama_model *m, rama_weights *w);
size_t rama_model_bytes(const rama_model *m);
/* Give the derived copies back (they are made again on the next call that wants them): mask 1 = the chain-order
 * copy parity mode streams, 2 = the tile-order copy of the token-batch passes, 3 = both -- llama2-7B: 92 GB -> 38 GB.
 * Synchronises the context's stream and drops its captured graphs (they hold the copies' addresses). */
int  rama_model_release_copies(rama_ctx *ctx, rama_model *m, int mask);
int  rama_model_free(rama_ctx *ctx, rama_model *m);

int  rama_state_create(rama_ctx *ctx, const rama_config *cfg, int n_local_layers, rama_run_state *out);
int  rama_state_free(rama_ctx *ctx, rama_run_state *s);

/* bit-exact device twin of oracle_fill_synth: dst[i] = bias + (float)(ih4(i+offset) - 131070) * scale */
int  rama_fill_synth(rama_ctx *ctx, float *dst, size_t n, uint64_t seed, uint64_t tag,
                     uint64_t offset, float scale, float bias);

/* ---------------------------------------------------------------- fused decode path
 * engine/src/transformer/infer.rs:8-53 forward(cfg, wv, rsv, token, pos, device): the same
 * values in logits / key_cache / value_cache / x (per-layer residual), computed with fused
 * launches (rmsnorm folded into the following matvec, RoPE + cache append into the QKV
 * epilogue, SiLU*gate into W1/W3, residual adds into Wo / W2).  Scratch buffers the fusion
 * makes dead (xb2, hb2, and x/xb after the final norm) are not written; the 1:1 ops above
 * reproduce the full choreography. */
int  rama_forward(rama_ctx *ctx, const rama_config *cfg, const rama_weights *w,
                  rama_run_state *s, int token, int pos);
int  rama_forward_stage(rama_ctx *ctx, const rama_config *cfg, const rama_weights *w,
                        rama_run_state *s, int token, int pos, const rama_stage *stage);

/* Batched-prompt prefill (no reference counterpart): the same state as
 * n_tokens calls rama_forward(tokens[i], pos0 + i) -- KV-cache rows pos0..pos0+n-1 of every layer,
 * residual x and logits of the LAST position -- with the weights streamed once per 128 positions (64 when the
 * weights are not a resident rama_model's: the 128-position kernels read the model's tile-order copies), as
 * dense fp32 GEMMs on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, csrc/prefill_mfma.hpp).
 * dim or hidden_dim not a multiple of 16: falls back to one rama_forward per token.
 * Parity mode ("ref_order" = 1) on a resident model: the positions go through token-batch kernels in the reference's
 * rounding order, 32 per weight pass, the last one through rama_forward -- cache rows, logits and run state are
 * bit for bit those of one rama_forward per position ("prefill_chain" = 0 gives exactly that loop). */
int  rama_prefill(rama_ctx *ctx, const rama_config *cfg, const rama_weights *w, rama_run_state *s,
                  const int32_t *tokens_host, int n_tokens, int pos0);

/* One decode step for up to 128 INDEPENDENT sequences (64 without a resident rama_model's tile-order copies; the
 * server's concurrent requests), every
 * weight row streamed once for all of them (no reference counterpart; same MFMA kernels as the
 * prefill).  states[i] is sequence i's run state; afterwards it holds what
 * rama_forward(token_i, pos_i) would have left in it: the appended cache rows and the logits
 * (x / xb / q ... scratch is not maintained).  Sequences may sit at different positions; two entries
 * must not share a state.  Parity mode ("ref_order" = 1) on a resident model: 32 sequences per weight pass through the
 * chain-order token-batch kernels, every sequence's logits and cache rows bit for bit those of its own rama_forward. */
int  rama_decode_batch(rama_ctx *ctx, const rama_config *cfg, const rama_weights *w,
                       const rama_run_state *states, const int32_t *tokens_host,
                       const int32_t *positions_host, int n_seq);
/* The same pass chained on the device: rama_decode_batch_begin uploads every sequence's (token, position) once;
 * each step of rama_decode_batch_steps ends with one argmax per sequence (Device::sample at temperature 0, ties to the
 * last index) that writes the sequence's next token and advances its position in device memory, so a step needs no
 * host round trip and -- in graph mode -- is ONE hipGraph replay.  rama_decode_batch_tokens downloads what the
 * sequences produced, out_host[s * max_per_seq + k] = token k of sequence s.  position_i + max_steps <= seq_len.
 * (Fast mode only; the logits stay in the pass's scratch, states[i].logits is not written.) */
int  rama_decode_batch_begin(rama_ctx *ctx, const rama_config *cfg, const rama_weights *w,
                             const rama_run_state *states, const int32_t *tokens_host,
                             const int32_t *positions_host, int n_seq, int max_steps);
int  rama_decode_batch_steps(rama_ctx *ctx, int n_steps);
int  rama_decode_batch_tokens(rama_ctx *ctx, int32_t *out_host, int max_per_seq, int *n_per_seq);
/* tokens `from`.. that sequence `seq` of the chained batch has produced SO FAR, without touching the stream (a host-visible
 * ring per sequence, as rama_decode_stream_poll): a server hands each request its tokens as they appear. */
int  rama_decode_batch_stream_poll(rama_ctx *ctx, int seq, int from, int32_t *out_tokens_host, int max_tokens, int *n_ready);

/* Layer-pipeline stage variants (no reference counterpart: the reference is single-device).
 * The token id is read from / written to DEVICE memory, so a stage boundary is one RCCL
 * send/recv of x[dim] (and of one int32 from the last stage back to the first) with no host
 * round trip.  token_dev may be NULL for a stage that does not embed. */
int  rama_forward_stage_devtok(rama_ctx *ctx, const rama_config *cfg, const rama_weights *w,
                               rama_run_state *s, const int32_t *token_dev, int pos,
                               const rama_stage *stage);
int  rama_argmax_dev(rama_ctx *ctx, const float *logits, size_t n, int32_t *result_dev);

/* ---------------------------------------------------------------- layer pipeline over RCCL (csrc/pipe.hip)
 * One process per GPU; rank r owns a contiguous layer range -- as even as possible, the first L mod N ranks
 * one layer more -- (rama_model_load_stage / rama_model_synth with a stage), n_seq >= N sequences in flight.  Rank 0 obtains a unique id and ships its
 * RAMA_PIPE_ID_BYTES bytes to the other ranks by any side channel (a file, an environment variable,
 * a torch.distributed / MPI broadcast); every rank then creates its end.  RCCL is loaded with dlopen on
 * first use.  All exchanges are enqueued on the context's stream. */
#define RAMA_PIPE_ID_BYTES 128
typedef struct rama_pipe rama_pipe;
int  rama_pipe_unique_id(void *id_out /* RAMA_PIPE_ID_BYTES */);
int  rama_pipe_create(rama_ctx *ctx, const void *id_bytes, int rank, int world, rama_pipe **out);
int  rama_pipe_destroy(rama_pipe *pipe);
/* the communicator's own count of ranks and this end's rank in it (ncclCommCount / ncclCommUserRank) */
int  rama_pipe_comm_info(const rama_pipe *pipe, int *n_ranks, int *rank);
/* One grouped exchange (ncclGroupStart .. ncclGroupEnd): each of the four legs is skipped when its
 * buffer is NULL.  x legs carry float[n], token legs one int32; peers are ranks of the pipe. */
int  rama_pipe_exchange(rama_pipe *pipe, const float *send_x, size_t n_send_x, int send_x_peer,
                        float *recv_x, size_t n_recv_x, int recv_x_peer,
                        const int32_t *send_tok, int send_tok_peer, int32_t *recv_tok, int recv_tok_peer);
/* The schedule: item j = (slot j % S, position j / S), S = max(n_seq, world) slots per round (slots
 * beyond n_seq idle); rank r computes item tick - r at each tick (BOS at position 0, the forced
 * prompt tokens next, mod.rs:182-191, then the token the last rank sampled: argmax at temperature 0,
 * else the device top-p sampler), then exchanges: x[dim] to rank r + 1, the sampled id from the last
 * rank to rank 0.  wrap > 0: a sequence that reaches position `wrap` starts a new generation at 0 in
 * the same slot. */
typedef struct {
    int32_t n_seq, n_pos, wrap;
    const int32_t *prompt;        /* host: forced prompt tokens, the same for every sequence */
    int32_t n_prompt;
    float temperature, topp, u;   /* Device::sample on the last rank (u = the reference's constant draw) */
    int32_t *out_tokens_dev;      /* rank 0, optional: device [n_seq, n_pos], the id sampled AFTER each position */
} rama_pipe_plan;
/* the schedule alone (no GPU): 1 + (*seq, *pos) if `rank` of `world` computes an item at `tick`, 0 if idle */
int  rama_pipe_item(const rama_pipe_plan *plan, int world, int rank, int tick, int *seq, int *pos);
int  rama_pipe_total_ticks(const rama_pipe *pipe, const rama_pipe_plan *plan);   /* S * n_pos + world - 1 */
int  rama_pipe_plan_ticks(const rama_pipe_plan *plan, int world);                 /* the same without a communicator */
/* [r6] EVERYTHING one tick of one rank consists of, as pure arithmetic (no GPU, no communicator): rama_pipe_run_ticks is a loop over this function
 * and executes exactly what it says -- so a CPU test that lets gloo ranks compute and exchange by it (tests/test_pipeline_gloo.py) validates the
 * native loop's own bookkeeping.  kinds: RAMA_PIPE_NONE | RAMA_PIPE_X (float[dim]) | RAMA_PIPE_TOKEN (one int32).
 * token_kind: 0 = BOS (mod.rs:182), 1 = forced prompt token `token` (mod.rs:190-191), 2 = the sequence's device token word (rank 0: what the last
 * rank sampled; other ranks take no token).  recv_pos: the position the received token was sampled AFTER (rank 0's history row). */
enum { RAMA_PIPE_NONE = 0, RAMA_PIPE_X = 1, RAMA_PIPE_TOKEN = 2 };
typedef struct {
    int32_t on, seq, pos, pos_wrapped, token_kind, token;
    int32_t samples;                                   /* this rank runs Device::sample behind the item (the last rank) */
    int32_t send_kind, send_seq, send_peer;
    int32_t recv_kind, recv_seq, recv_peer, recv_pos;
} rama_pipe_tick;
int  rama_pipe_tick_plan(const rama_pipe_plan *plan, int world, int rank, int tick, rama_pipe_tick *out);
/* Runs ticks [tick_from, tick_to) of this rank: states[s] is sequence s's run state for this stage
 * (its x is the hand-off buffer), tok_dev[s] its device token word.  Asynchronous on the context's
 * stream; after the last tick of the last rank tok_dev[s] holds sequence s's newest token. */
int  rama_pipe_run_ticks(rama_pipe *pipe, const rama_config *cfg, const rama_weights *w, rama_run_state *states,
                         int32_t *const *tok_dev, const rama_stage *stage, const rama_pipe_plan *plan,
                         int tick_from, int tick_to);
const char *rama_pipe_last_error(void);

/* generate() loop of mod.rs:169-206 at temperature 0, chained on the device: token = 1 (BOS)
 * at pos 0; while pos < steps: forward; next = pos < n_prompt ? prompt[pos] : argmax(logits);
 * out[pos] = next.  No per-token host round trip; out_tokens_host receives `steps` ids. */
int  rama_generate_greedy(rama_ctx *ctx, const rama_config *cfg, const rama_weights *w,
                          rama_run_state *s, const int32_t *prompt_tokens_host, int n_prompt,
                          int steps, int32_t *out_tokens_host);

/* Device::sample for temperature != 0 without the logits download (reference: cpu.rs:155-179,
 * sample_top_q infer.rs:55-85; the reference GPU path copies 128 KB to the host per token,
 * gpu.rs:149-173): logits (/ T if T < 1) -> softmax -> keep p > (1 - topp) / (n - 1) -> stable
 * descending sort -> cut where the running sum exceeds topp -> draw with r = u * sum.  `u` is the
 * uniform draw; the reference re-seeds its generator on every call, so it is the same constant
 * for every token.  temperature == 0 is the argmax.  Result in device memory;
 * -1 when no probability exceeds the cutoff (the reference's index arithmetic underflows there). */
int  rama_sample_topp_dev(rama_ctx *ctx, const float *logits, size_t n, float temperature, float topp,
                          float u, int32_t *result_dev);
/* Sampler of the chained decode loop (rama_decode_steps): temperature 0 (default) = argmax. */
int  rama_decode_sampler(rama_ctx *ctx, float temperature, float topp, float u);
/* generate() (mod.rs:169-206) for any temperature, chained on the device: no per-token host round
 * trip; rama_generate_greedy is this with temperature 0. */
int  rama_generate(rama_ctx *ctx, const rama_config *cfg, const rama_weights *w, rama_run_state *s,
                   const int32_t *prompt_tokens_host, int n_prompt, int steps, float temperature,
                   float topp, float u, int32_t *out_tokens_host);
/* The same loop in pieces, for timing: begin sets (token, pos) and the forced-token list;
 * each decode_steps call enqueues n more (forward + argmax + advance) steps. */
int  rama_decode_begin(rama_ctx *ctx, int token, int pos, const int32_t *forced_tokens_host, int n_forced);
int  rama_decode_steps(rama_ctx *ctx, const rama_config *cfg, const rama_weights *w,
                       rama_run_state *s, int n_steps);
/* tokens produced since rama_decode_begin (synchronises) */
int  rama_decode_tokens(rama_ctx *ctx, int32_t *out_tokens_host, int max_tokens, int *n_out);
/* Tokens `from`.. that the chained loop has produced SO FAR, without touching the stream (mod.rs:209-248 generate_stream
 * sends each token as it is produced): the sampling launch writes every token also to a host-visible ring (pinned,
 * device-mapped memory, one system-scope store per token), which this call reads.  Returns at once; *n_ready may be 0.
 * What goes wrong inside the loop is reported by rama_decode_tokens at the end. */
int  rama_decode_stream_poll(rama_ctx *ctx, int from, int32_t *out_tokens_host, int max_tokens, int *n_ready);
/* generate_stream (mod.rs:209-248): rama_generate with every token handed to on_token(user, index, token) on the calling
 * thread as soon as the device has produced it; the loop itself stays chained on the device (no host round trip per
 * token).  out_tokens_host (may be NULL) receives the whole list at the end. */
int  rama_generate_stream(rama_ctx *ctx, const rama_config *cfg, const rama_weights *w, rama_run_state *s,
                          const int32_t *prompt_tokens_host, int n_prompt, int steps, float temperature,
                          float topp, float u, void (*on_token)(void *user, int index, int32_t token), void *user,
                          int32_t *out_tokens_host);
/* 1: replay launches from hipGraphs (default 0 = eager launches): rama_decode_steps / rama_generate capture a
 * decode step (or a few, "graph_steps") once per attention variant; rama_forward and rama_forward_stage* keep one
 * graph per (run state, stage, attention variant) -- token and position travel through the device cursor, so a
 * trait-level host or a pipeline stage issues one cursor write and one graph launch per token.  Setting 0
 * synchronises and drops every captured graph; free a run state only after that (or after rama_ctx_destroy). */
int  rama_set_graph_mode(rama_ctx *ctx, int enabled);
/* Modes and performance knobs.  The MODE keys change which of the reference's admissible roundings is reproduced; every other key leaves results
 * unchanged (parity mode: the same bits; fast mode: up to fp32 summation order).  Changing a key synchronises and drops captured graphs.
 * Measurements and history of every key: DESIGN.md appendix A.
 *
 * MODE
 *   "ref_order" = 0|1|2|3
 *        0 (library default) FAST: fused multiply-adds, tree-shaped sums -- closer to the exact logits than the CPU path, ~1.5e-4 from it at llama2-7B depth.
 *        1 PARITY (the default of every host mirror: C++ CLI, Rust shim, bench.py's `value`): every op in the reference CPU path's own rounding order
 *          (cpu.rs: 4-lane strided matvec sums with separate multiply and add, front-to-back rmsnorm / softmax / attention sums, glibc's expf restated) on
 *          chain-order weight copies (csrc/chain.hpp; made on first use, +27 GB at llama2-7B).  Bit-identical to oracle/rama_oracle.c -- i.e. to cpu.rs with
 *          the two orders it leaves to its crates fixed: "lane_reduce" below, and rayon's softmax sum taken as ONE front-to-back sum.  Weights uploaded
 *          tensor by tensor (hbm.rs:55-90) run the same kernels: the fused entries ADOPT them, rama_matmul makes a chain-order copy of a model-less matrix
 *          on first use; rama_free, rama_copy_h2d_f32 and every entry that WRITES a tensor (an op's output, rama_fill_synth) drop what was derived from it.
 *          A resident model's own tensors (rama_model_weights) are immutable by contract.
 *        2 TOLERANCE experiment: chain-order matvecs, tree-summed norms folded in, fast attention (1.4e-4 from the CPU path: the instrument of "tol_mask").
 *        3 BAR: parity up to position "bar_pos" - 1, the fast path's attention from there on (matvecs and norms stay exact).  Not bit-identical behind the
 *          switch; measured <= 1e-4 from the oracle over the whole 2 048-position context at llama2-7B depth, +7 % tokens/s at position 1 900.
 *   "lane_reduce" = 0|1|2 : parity / bar mode: the order of cpu.rs:148 `v.reduce_add()` -- 0 pairwise (l0+l1)+(l2+l3) (default), 1 strided (l0+l2)+(l1+l3),
 *        2 sequential ((l0+l1)+l2)+l3.  wide::f32x4 leaves it to the build's target features; the Rust shim asks the crate at start-up and sets this.
 *   "bar_pos" = N : first position of bar mode's fast attention (default 128; clamped to "spread_pos" and 256)
 *   "tol_mask" = 0..127 : with "ref_order" = 2, ops swapped for A/B runs (1 / 2 / 4 / 8 / 16 = the FAST Wq|Wk|Wv / Wo / W1|W3 / W2 / classifier launch,
 *        32 = parity mode's attention, 64 = its exact norms; tools/tol_sweep.py)
 *
 * FAST MODE  (default in brackets)
 *   "geom" 0..3 [3] matvec workgroup geometry | "resid_r2" 0..3 [2] geometry of Wo / W2 | "solo" -1|0|1 [-1: rows <= 2048 floats] one wave per row group
 *   "w13i" 0|1 [1] W1|W3 from the row-interleaved copy | "fused" / "fused_solo" -1|0|1 [-1: dim <= 1024 / dim > 512] a whole stage as ONE launch (layer_fused.hpp)
 *   "merge" -1|0|1 [-1: dim <= 1024] attention + Wo as one launch | "split_pos" -1|N [-1: 256 when a head's K+V cache exceeds 1 MiB] split-T attention from N
 *   "attn_nsplit" 0..32 [0], "attn_waves" 16|8|4 [8], "attn_nt" 0|1 [1], "attn_u" 8|16 [8], "combine_v" 0|1 [1] : split-T geometry
 *   "small_attn" -1|0|1 [-1], "small_attn_waves" 4|8 [8], "small_attn_pos" N [256] : fewer-wave attention at short contexts
 *   "graph_steps" -1|1..32 [-1: 4 for dim <= 1024, else 1] decode steps per captured graph | "prefill" 0|1 [1] rama_generate_greedy's prompt through rama_prefill
 *   "prefill_tok" 64|128 [128], "prefill_attn" 0|1 [1], "tiled" 0|1 [1], "norm_in_gemm" 0|1 [1] : the token-batch passes (prefill_mfma.hpp, prefill_attn.hpp)
 *   "topp_sort", "topp_pairs", "topp_dist" 0|1 [1], "topp_block" 512|1024|2048 [1024], "topp_keep_sums" 0|1 [0] : the device top-p sampler's phases
 *
 * PARITY MODE  (same bits whatever the setting)
 *   "chain" 0|1 [1] chain-order copies (0: one thread per row) | "chain_d" 0|100 W + D [0], "chain_resid_d" -1|0|100 W + D [-1], "chain_lead_w" 0|1|2 [0] : geometry
 *   "chain_norm" 0|1 [1] exact norms folded into the matvecs (dim <= 512) | "chain_lead" 0|1 [1] the norms' exact sums by a leader workgroup of the consuming launch
 *   "chain_split" 0|1 [1] remainder row groups as half groups | "chain_views" 0|1 [1] chain-order copies of model-less matrices
 *   "spread_pos" 64..2^20 [128] exact attention spread over the chip from this position | "attn_fv" 0|1 [1] its softmax + value chains as one launch
 *   "prefill_chain" 0|1 [1] prompt positions through the chain-order token-batch kernels
 *   "rope_batch", "matmul_batch", "ew_batch", "norm_fold", "resid_fold", "qkv_fold" 0|1 [1] : the 1:1 Device ops are RECORDED and issued merged (49 -> 6 launches
 *        per layer; every hazard falls back to program order) */
int  rama_set_tuning(rama_ctx *ctx, const char *key, int value);

/* glibc 2.35 expf (the exp the reference's f32::exp calls on Linux) as the reference-order kernels
 * evaluate it, elementwise: the bit-exactness test's handle on it */
int  rama_ref_expf(rama_ctx *ctx, float *o, const float *x, size_t n);

/* ---------------------------------------------------------------- measurement
 * HIP events on the context's stream (the stream the kernels are launched on). */
int  rama_timer_start(rama_ctx *ctx);
int  rama_timer_stop(rama_ctx *ctx, float *elapsed_ms);      /* synchronises */
/* Per-kernel timing: while enabled, every launch of kernel class `kernel_id` on the fused
 * path is bracketed by an event pair (eager mode only).  Classes: */
#define RAMA_K_QKV   0   /* rmsnorm + Wq|Wk|Wv matvec + RoPE + cache append */
#define RAMA_K_ATTN  1   /* multi-head attention */
#define RAMA_K_WO    2   /* Wo matvec + residual */
#define RAMA_K_W13   3   /* rmsnorm + W1|W3 matvec + SiLU*gate */
#define RAMA_K_W2    4   /* W2 matvec + residual */
#define RAMA_K_CLS   5   /* final rmsnorm + classifier matvec */
#define RAMA_K_NORM  6   /* an rmsnorm launch of its own (parity mode: the exact sequential sum of squares) */
#define RAMA_K_SAMPLE 7  /* Device::sample on the device: argmax, or the top-p sampler's launches (event records around them) */
#define RAMA_K_COUNT 8
int  rama_kprof_enable(rama_ctx *ctx, int kernel_id, int max_records);
int  rama_kprof_read(rama_ctx *ctx, int *n_launches, double *total_ms);  /* synchronises, disables */

#ifdef __cplusplus
}
#endif
#endif /* RAMA_HIP_H */
