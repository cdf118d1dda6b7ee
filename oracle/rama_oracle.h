/*
 * rama_oracle.h -- CPU restatement of oliverhu/rama's fp32 Llama-2 decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rama_amd/ (the product) may include,
 * link, import or execute anything in oracle/.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU
 * baseline -- never as the thing shipped.
 *
 * Each function cites the reference file:line (relative to the reference repo
 * root) whose arithmetic it restates.  The reference is Rust (rayon + wide::f32x4);
 * it cannot be compiled in this image (no cargo/rustc), so this is a restatement
 * ("port"), pinned by
 *   - the one known-answer vector the reference's own test holds
 *     (engine/src/device/gpu.rs:249-288, test_blas), and
 *   - golden logits / intermediates produced in the build container by importing
 *     the reference's own PyTorch model definition (engine/export/model.py,
 *     export.py) -- tests/golden/, generator tools/make_goldens.py.
 *
 * Floating-point contract: every op is plain IEEE fp32, products and sums rounded
 * separately (the reference never uses FMA: wide::f32x4 `+=`/`*`, Rust iterators),
 * so this file must be compiled with -ffp-contract=off.
 */
#ifndef RAMA_ORACLE_H
#define RAMA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* engine/src/transformer/mod.rs:128-138 (struct Config) */
typedef struct {
    int32_t dim;
    int32_t hidden_dim;
    int32_t n_layers;
    int32_t n_heads;
    int32_t n_kv_heads;
    int32_t vocab_size;    /* always positive here; sign handled by the header parser */
    int32_t seq_len;
    int32_t shared_weight; /* 1 when the file's vocab_size was > 0 (mod.rs:150-157) */
} oracle_config;

/* engine/src/transformer/state.rs:53-74 (TransformerWeights), host pointers.
 * wcls aliases token_embedding_table when shared (state.rs:111-117). */
typedef struct {
    const float *token_embedding_table; /* [vocab, dim] */
    const float *rms_att_weight;        /* [L, dim] */
    const float *rms_ffn_weight;        /* [L, dim] */
    const float *wq, *wk, *wv, *wo;     /* [L, dim, dim] */
    const float *w1;                    /* [L, hidden, dim] */
    const float *w2;                    /* [L, dim, hidden] */
    const float *w3;                    /* [L, hidden, dim] */
    const float *rms_final_weight;      /* [dim] */
    const float *freq_cis_real;         /* [seq_len, head_size/2] */
    const float *freq_cis_imag;         /* [seq_len, head_size/2] */
    const float *wcls;                  /* [vocab, dim] */
} oracle_weights;

/* engine/src/transformer/state.rs:3-17 (RunState), sized as ram.rs:7-23 */
typedef struct {
    float *x, *xb, *xb2;   /* [dim] */
    float *hb, *hb2;       /* [hidden] */
    float *q, *k, *v;      /* [dim] */
    float *att;            /* [n_heads, seq_len] */
    float *logits;         /* [vocab] */
    float *key_cache;      /* [L, seq_len, dim]  (stride dim: infer.rs:31-33) */
    float *value_cache;    /* [L, seq_len, dim] */
} oracle_state;

/* Number of OpenMP threads the row-/head-parallel loops use (rayon's global pool
 * in the reference).  0 = leave the OpenMP default.  Results do not depend on it. */
void oracle_set_threads(int n);
int  oracle_get_threads(void);

/* ---- the two orders the reference does not fix itself -------------------------------------------------------------
 * (1) cpu.rs:148 `v.reduce_add()`: the final sum of the four lane sums belongs to wide 0.7.x (engine/Cargo.toml:18) and
 *     to the target features of the build: pairwise (l0+l1)+(l2+l3) | strided (l0+l2)+(l1+l3) | sequential
 *     ((l0+l1)+l2)+l3.  Default: pairwise.
 * (2) cpu.rs:190 `x.par_iter().sum::<f32>()`: rayon 1.8 (engine/Cargo.toml:14) halves the producer while its splitter
 *     has splits left -- as this file's author reads rayon's plumbing, `splits` starts at the pool's thread count and is
 *     halved per level, i.e. floor(log2(threads)) + 1 levels without steals (2 leaves with ONE thread, 32 with 16),
 *     more when a half is stolen -- sums every leaf front to back and adds the halves left + right.  The reference is
 *     therefore not bit-reproducible against itself across pool sizes (nor, with steals, run to run).
 *     oracle_set_softmax_split(levels): 2^levels leaves; default 0 = ONE front-to-back sum, the canonical order every
 *     round of this repo compared against (a pool whose producer is never split).
 * tools/ref_self_spread.py measures how far these admissible executions sit from each other at llama2-7B depth
 * (profiles/r06_reference_self_spread.json): that distance is what "within 1e-4 of the reference" can mean at all. */
enum { ORACLE_LANES_PAIRWISE = 0, ORACLE_LANES_STRIDED = 1, ORACLE_LANES_SEQUENTIAL = 2 };
void oracle_set_lane_reduce(int order);
int  oracle_get_lane_reduce(void);
void oracle_set_softmax_split(int levels);
int  oracle_get_softmax_split(void);

/* ---- Device<Vec<f32>> for CPU, engine/src/device/cpu.rs ---- */
void oracle_array_add(float *target, const float *source, size_t n);          /* cpu.rs:16-21 */
void oracle_array_mult(float *target, const float *source, size_t n);         /* cpu.rs:59-64 */
void oracle_expf_array(float *o, const float *x, size_t n);                        /* libm expf = Rust f32::exp on Linux */
void oracle_sinu(float *o, size_t n);                                          /* cpu.rs:54-57 */
void oracle_copy_from_slice(float *target, const float *source, size_t n);    /* cpu.rs:66-72 */
void oracle_rmsnorm(float *o, const float *x, const float *weight, size_t n); /* cpu.rs:99-117 */
void oracle_apply_position(float *q, float *k, const float *pos_real,
                           const float *pos_img, size_t head_size);           /* cpu.rs:74-97 */
/* cpu.rs:127-153.  Returns 0, or -1 where the reference would panic
 * (width % 4 != 0: the f32x4::from(&a[..4]) slice at cpu.rs:143). */
int  oracle_matmul(float *o, const float *a, const float *b,
                   size_t width, size_t o_rows, size_t o_cols);
void oracle_softmax(float *x, size_t n);                                       /* cpu.rs:119-125,187-192 */
void oracle_multi_head_attention(const oracle_config *cfg, oracle_state *s,
                                 int layer, int pos);                          /* cpu.rs:23-52 */
/* T==0 leg of Device::sample, cpu.rs:163-167: on ties the LAST maximal index wins. */
int  oracle_argmax(const float *logits, size_t n);
/* T!=0 leg, cpu.rs:168-177 + infer.rs:55-85.  `u` stands in for the ChaCha20 draw
 * `rng.gen::<f32>()`, which is the same constant every call because the generator is
 * re-seeded per call (cpu.rs:161-162).  Mutates logits like the reference. */
int  oracle_sample(float *logits, size_t n, float temperature, float topp, float u);
/* infer.rs:55-85 on given probabilities; `u` in place of rng.gen::<f32>() */
int  oracle_sample_top_q(const float *p, size_t num, float topp, float u);

/* ---- engine/src/transformer/infer.rs:8-53 ---- */
void oracle_forward(const oracle_config *cfg, const oracle_weights *w,
                    oracle_state *s, int token, int pos);
/* The same op sequence restricted to layers [layer_begin, layer_end); embedding
 * gather iff do_embed, final norm + classifier iff do_cls.  Used to check a
 * layer-pipeline stage; oracle_forward == (0, n_layers, 1, 1). */
void oracle_forward_range(const oracle_config *cfg, const oracle_weights *w,
                          oracle_state *s, int token, int pos,
                          int layer_begin, int layer_end, int do_embed, int do_cls);

/* fp64-accumulated arbiter of the same network (not the reference's arithmetic:
 * used only to report how far BOTH fp32 paths sit from the exact answer). */
void oracle_forward_f64(const oracle_config *cfg, const oracle_weights *w,
                        oracle_state *s, int token, int pos);

/* ---- synthetic weights (no reference counterpart; real checkpoints are not
 * available offline).  Integer-only hash + Irwin-Hall(4) so that CPU, numpy and
 * the HIP fill kernel produce bit-identical floats:
 *   z   = mix64(offset + i + tag*0x9E3779B97F4A7C15 + seed*0xD1B54A32D192ED03)
 *         (mix64 = the splitmix64 finaliser)
 *   s   = sum of the four 16-bit fields of z            (0 .. 262140)
 *   out = bias + (float)((int)s - 131070) * scale        (multiply, then add)
 * std(out) = scale * 37837.2272...; `offset` is the flat index of dst[0] in the
 * full tensor, so a layer range can be generated on its own.                  */
void oracle_fill_synth(float *dst, size_t n, uint64_t seed, uint64_t tag,
                       uint64_t offset, float scale, float bias);

#ifdef __cplusplus
}
#endif
#endif
