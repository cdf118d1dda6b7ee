//! Raw bindings of include/rama_hip.h: every entry point (first the ones the Device trait and the optional
//! fast path need, then the rest).
//! `*mut f32` / `*const f32` are DEVICE pointers; every function returns 0 or an error code and
//! leaves a message in `rama_last_error()` (thread-local).
#![allow(non_camel_case_types, dead_code)]
use core::ffi::{c_char, c_int, c_void};

#[repr(C)] pub struct rama_ctx { _p: [u8; 0] }
#[repr(C)] pub struct rama_model { _p: [u8; 0] }

/// transformer/mod.rs:128-138 with C ints
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rama_config {
    pub dim: i32, pub hidden_dim: i32, pub n_layers: i32, pub n_heads: i32,
    pub n_kv_heads: i32, pub vocab_size: i32, pub seq_len: i32, pub shared_weight: i32,
}
/// transformer/state.rs:53-74 as device pointers
#[repr(C)] #[derive(Clone, Copy)]
pub struct rama_weights {
    pub token_embedding_table: *const f32, pub rms_att_weight: *const f32, pub rms_ffn_weight: *const f32,
    pub wq: *const f32, pub wk: *const f32, pub wv: *const f32, pub wo: *const f32,
    pub w1: *const f32, pub w2: *const f32, pub w3: *const f32,
    pub rms_final_weight: *const f32, pub freq_cis_real: *const f32, pub freq_cis_imag: *const f32,
    pub wcls: *const f32,
}
/// transformer/state.rs:3-17 as device pointers
#[repr(C)] #[derive(Clone, Copy)]
pub struct rama_run_state {
    pub x: *mut f32, pub xb: *mut f32, pub xb2: *mut f32, pub hb: *mut f32, pub hb2: *mut f32,
    pub q: *mut f32, pub k: *mut f32, pub v: *mut f32, pub att: *mut f32, pub logits: *mut f32,
    pub key_cache: *mut f32, pub value_cache: *mut f32,
}

extern "C" {
    pub fn rama_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut rama_ctx) -> c_int;
    pub fn rama_ctx_destroy(ctx: *mut rama_ctx) -> c_int;
    pub fn rama_sync(ctx: *mut rama_ctx) -> c_int;
    pub fn rama_stream_query(ctx: *mut rama_ctx) -> c_int;
    pub fn rama_last_error() -> *const c_char;

    pub fn rama_alloc_f32(ctx: *mut rama_ctx, n: usize, out: *mut *mut f32) -> c_int;
    pub fn rama_upload_f32(ctx: *mut rama_ctx, host: *const f32, n: usize, out: *mut *mut f32) -> c_int;
    pub fn rama_download_f32(ctx: *mut rama_ctx, src: *const f32, n: usize, host: *mut f32) -> c_int;
    pub fn rama_free(ctx: *mut rama_ctx, p: *mut c_void) -> c_int;

    // one per Device<T> method (device/device.rs:4-21)
    pub fn rama_array_add(ctx: *mut rama_ctx, target: *mut f32, source: *const f32, n: usize) -> c_int;
    pub fn rama_array_mult(ctx: *mut rama_ctx, target: *mut f32, source: *const f32, n: usize) -> c_int;
    pub fn rama_sinu(ctx: *mut rama_ctx, o: *mut f32, n: usize) -> c_int;
    pub fn rama_copy_from_slice(ctx: *mut rama_ctx, target: *mut f32, source: *const f32, n: usize) -> c_int;
    pub fn rama_rmsnorm(ctx: *mut rama_ctx, o: *mut f32, x: *const f32, weight: *const f32, n: usize) -> c_int;
    pub fn rama_apply_position(ctx: *mut rama_ctx, q: *mut f32, k: *mut f32, pos_real: *const f32,
                               pos_img: *const f32, head_size: usize) -> c_int;
    pub fn rama_matmul(ctx: *mut rama_ctx, o: *mut f32, a: *const f32, b: *const f32,
                       width: usize, o_rows: usize, o_cols: usize) -> c_int;
    pub fn rama_softmax(ctx: *mut rama_ctx, x: *mut f32, n: usize) -> c_int;
    pub fn rama_multi_head_attention(ctx: *mut rama_ctx, xb: *mut f32, att: *mut f32, q: *const f32,
                                     key_cache: *const f32, value_cache: *const f32, layer: c_int, dim: c_int,
                                     pos: c_int, head_size: c_int, seq_len: c_int, n_heads: c_int) -> c_int;
    pub fn rama_sample_argmax(ctx: *mut rama_ctx, logits: *const f32, n: usize, next: *mut i32) -> c_int;
    pub fn rama_sample_topp(ctx: *mut rama_ctx, logits: *const f32, n: usize, temperature: f32,
                            topp: f32, u: f32, next: *mut i32) -> c_int;

    // optional fast path: forward() / generate() as single calls, same observable state
    pub fn rama_forward(ctx: *mut rama_ctx, cfg: *const rama_config, w: *const rama_weights,
                        s: *mut rama_run_state, token: c_int, pos: c_int) -> c_int;
    pub fn rama_prefill(ctx: *mut rama_ctx, cfg: *const rama_config, w: *const rama_weights,
                        s: *mut rama_run_state, tokens_host: *const i32, n_tokens: c_int, pos0: c_int) -> c_int;
    /// one decode step for up to 64 independent sequences (states: array of n_seq run states)
    pub fn rama_decode_batch(ctx: *mut rama_ctx, cfg: *const rama_config, w: *const rama_weights,
                             states: *const rama_run_state, tokens_host: *const i32, positions_host: *const i32,
                             n_seq: c_int) -> c_int;
    pub fn rama_decode_batch_begin(ctx: *mut rama_ctx, cfg: *const rama_config, w: *const rama_weights, states: *const rama_run_state,
                                   tokens_host: *const i32, positions_host: *const i32, n_seq: c_int, max_steps: c_int) -> c_int;
    pub fn rama_decode_batch_steps(ctx: *mut rama_ctx, n_steps: c_int) -> c_int;
    pub fn rama_decode_batch_tokens(ctx: *mut rama_ctx, out_host: *mut i32, max_per_seq: c_int, n_per_seq: *mut c_int) -> c_int;
    pub fn rama_generate(ctx: *mut rama_ctx, cfg: *const rama_config, w: *const rama_weights,
                         s: *mut rama_run_state, prompt_tokens_host: *const i32, n_prompt: c_int, steps: c_int,
                         temperature: f32, topp: f32, u: f32, out_tokens_host: *mut i32) -> c_int;
    pub fn rama_set_graph_mode(ctx: *mut rama_ctx, enabled: c_int) -> c_int;

    // resident model straight from a v0 checkpoint (mmap + one H2D copy)
    pub fn rama_model_load(ctx: *mut rama_ctx, path: *const c_char, out: *mut *mut rama_model) -> c_int;
    pub fn rama_model_config(m: *const rama_model, cfg: *mut rama_config) -> c_int;
    pub fn rama_model_weights(m: *const rama_model, w: *mut rama_weights) -> c_int;
    pub fn rama_model_free(ctx: *mut rama_ctx, m: *mut rama_model) -> c_int;
    pub fn rama_model_release_copies(ctx: *mut rama_ctx, m: *mut rama_model, mask: c_int) -> c_int;
    /// one pipeline stage's tensors only (layers [layer_begin, layer_end), embedding / classifier as flagged)
    pub fn rama_model_load_stage(ctx: *mut rama_ctx, path: *const c_char, stage: *const rama_stage,
                                 out: *mut *mut rama_model) -> c_int;

    // performance / parity knobs: "ref_order" = 1 switches every op to the CPU backend's rounding order
    // (bit-identical logits, for diffing against `cargo run --release`)
    pub fn rama_set_tuning(ctx: *mut rama_ctx, key: *const c_char, value: c_int) -> c_int;

    // layer pipeline over RCCL, one process per GPU (csrc/pipe.hip)
    pub fn rama_pipe_unique_id(id_out: *mut u8) -> c_int;                       // RAMA_PIPE_ID_BYTES = 128
    pub fn rama_pipe_create(ctx: *mut rama_ctx, id: *const u8, rank: c_int, world: c_int, out: *mut *mut rama_pipe) -> c_int;
    pub fn rama_pipe_destroy(pipe: *mut rama_pipe) -> c_int;
    pub fn rama_pipe_comm_info(pipe: *const rama_pipe, n_ranks: *mut c_int, rank: *mut c_int) -> c_int;
    pub fn rama_pipe_total_ticks(pipe: *const rama_pipe, plan: *const rama_pipe_plan) -> c_int;
    pub fn rama_pipe_run_ticks(pipe: *mut rama_pipe, cfg: *const rama_config, w: *const rama_weights,
                               states: *mut rama_run_state, tok_dev: *const *mut i32, stage: *const rama_stage,
                               plan: *const rama_pipe_plan, tick_from: c_int, tick_to: c_int) -> c_int;
    pub fn rama_pipe_exchange(pipe: *mut rama_pipe, send_x: *const f32, n_send_x: usize, send_x_peer: c_int,
                              recv_x: *mut f32, n_recv_x: usize, recv_x_peer: c_int,
                              send_tok: *const i32, send_tok_peer: c_int, recv_tok: *mut i32, recv_tok_peer: c_int) -> c_int;
    pub fn rama_pipe_item(plan: *const rama_pipe_plan, world: c_int, rank: c_int, tick: c_int, seq: *mut c_int, pos: *mut c_int) -> c_int;
    pub fn rama_pipe_plan_ticks(plan: *const rama_pipe_plan, world: c_int) -> c_int;
    pub fn rama_pipe_tick_plan(plan: *const rama_pipe_plan, world: c_int, rank: c_int, tick: c_int, out: *mut rama_pipe_tick) -> c_int;
    pub fn rama_pipe_last_error() -> *const c_char;

    // the rest of include/rama_hip.h, for hosts that want more than the trait: stage-wise forwards
    // (pipeline), the chained decode loop piece by piece, device-resident sampling, model helpers,
    // measurement.  tests/test_abi_and_host.py keeps this block and the header in step.
    pub fn rama_device_info(ctx: *mut rama_ctx, name: *mut c_char /* [64] */, compute_units: *mut c_int, hbm_bytes: *mut usize) -> c_int;
    pub fn rama_copy_h2d_f32(ctx: *mut rama_ctx, dst: *mut f32, host: *const f32, n: usize) -> c_int;
    pub fn rama_fill_synth(ctx: *mut rama_ctx, dst: *mut f32, n: usize, seed: u64, tag: u64, offset: u64, scale: f32, bias: f32) -> c_int;
    pub fn rama_forward_stage(ctx: *mut rama_ctx, cfg: *const rama_config, w: *const rama_weights, s: *mut rama_run_state,
                              token: c_int, pos: c_int, stage: *const rama_stage) -> c_int;
    pub fn rama_forward_stage_devtok(ctx: *mut rama_ctx, cfg: *const rama_config, w: *const rama_weights, s: *mut rama_run_state,
                                     token_dev: *const i32, pos: c_int, stage: *const rama_stage) -> c_int;
    pub fn rama_argmax_dev(ctx: *mut rama_ctx, logits: *const f32, n: usize, result_dev: *mut i32) -> c_int;
    pub fn rama_sample_topp_dev(ctx: *mut rama_ctx, logits: *const f32, n: usize, temperature: f32, topp: f32, u: f32,
                                result_dev: *mut i32) -> c_int;
    pub fn rama_generate_greedy(ctx: *mut rama_ctx, cfg: *const rama_config, w: *const rama_weights, s: *mut rama_run_state,
                                prompt_tokens_host: *const i32, n_prompt: c_int, steps: c_int, out_tokens_host: *mut i32) -> c_int;
    pub fn rama_decode_begin(ctx: *mut rama_ctx, token: c_int, pos: c_int, forced_tokens_host: *const i32, n_forced: c_int) -> c_int;
    pub fn rama_decode_sampler(ctx: *mut rama_ctx, temperature: f32, topp: f32, u: f32) -> c_int;
    pub fn rama_decode_steps(ctx: *mut rama_ctx, cfg: *const rama_config, w: *const rama_weights, s: *mut rama_run_state, n_steps: c_int) -> c_int;
    pub fn rama_decode_tokens(ctx: *mut rama_ctx, out_tokens_host: *mut i32, max_tokens: c_int, n_out: *mut c_int) -> c_int;
    pub fn rama_decode_batch_stream_poll(ctx: *mut rama_ctx, seq: c_int, from: c_int, out_tokens_host: *mut i32, max_tokens: c_int, n_ready: *mut c_int) -> c_int;
    pub fn rama_decode_stream_poll(ctx: *mut rama_ctx, from: c_int, out_tokens_host: *mut i32, max_tokens: c_int, n_ready: *mut c_int) -> c_int;
    pub fn rama_generate_stream(ctx: *mut rama_ctx, cfg: *const rama_config, w: *const rama_weights, s: *mut rama_run_state,
                                prompt_tokens_host: *const i32, n_prompt: c_int, steps: c_int, temperature: f32, topp: f32, u: f32,
                                on_token: Option<unsafe extern "C" fn(user: *mut c_void, index: c_int, token: i32)>, user: *mut c_void,
                                out_tokens_host: *mut i32) -> c_int;
    pub fn rama_state_create(ctx: *mut rama_ctx, cfg: *const rama_config, n_local_layers: c_int, out: *mut rama_run_state) -> c_int;
    pub fn rama_state_free(ctx: *mut rama_ctx, s: *mut rama_run_state) -> c_int;
    pub fn rama_model_synth(ctx: *mut rama_ctx, cfg: *const rama_config, seed: u64, stage: *const rama_stage,
                            rope_real_host: *const f32, rope_imag_host: *const f32, out: *mut *mut rama_model) -> c_int;
    pub fn rama_model_save(ctx: *mut rama_ctx, model: *const rama_model, path: *const c_char) -> c_int;
    pub fn rama_model_bytes(m: *const rama_model) -> usize;
    pub fn rama_ref_expf(ctx: *mut rama_ctx, o: *mut f32, x: *const f32, n: usize) -> c_int;
    pub fn rama_timer_start(ctx: *mut rama_ctx) -> c_int;
    pub fn rama_timer_stop(ctx: *mut rama_ctx, elapsed_ms: *mut f32) -> c_int;
    pub fn rama_kprof_enable(ctx: *mut rama_ctx, kernel_id: c_int, max_records: c_int) -> c_int;
    pub fn rama_kprof_read(ctx: *mut rama_ctx, n_launches: *mut c_int, total_ms: *mut f64) -> c_int;
}

#[repr(C)] pub struct rama_pipe { _p: [u8; 0] }
#[repr(C)] #[derive(Clone, Copy)]
pub struct rama_stage { pub layer_begin: i32, pub layer_end: i32, pub do_embed: i32, pub do_cls: i32 }
#[repr(C)]
pub struct rama_pipe_plan {
    pub n_seq: i32, pub n_pos: i32, pub wrap: i32,
    pub prompt: *const i32, pub n_prompt: i32,
    pub temperature: f32, pub topp: f32, pub u: f32,
    pub out_tokens_dev: *mut i32,
}

/// one tick of one rank of the layer pipeline, decided (include/rama_hip.h rama_pipe_tick): what rama_pipe_run_ticks executes
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rama_pipe_tick {
    pub on: i32, pub seq: i32, pub pos: i32, pub pos_wrapped: i32, pub token_kind: i32, pub token: i32,
    pub samples: i32,
    pub send_kind: i32, pub send_seq: i32, pub send_peer: i32,
    pub recv_kind: i32, pub recv_seq: i32, pub recv_peer: i32, pub recv_pos: i32,
}
