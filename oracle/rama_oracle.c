/*
 * rama_oracle.c -- CPU restatement of oliverhu/rama's decode path.  See rama_oracle.h.
 * TEST INFRASTRUCTURE ONLY (checker + timed CPU baseline); never shipped, never on
 * the product path.  Compile with -ffp-contract=off (the reference never fuses).
 *
 * Parallel structure follows the reference: matmul is parallel over output rows
 * (rayon par_iter_mut, cpu.rs:137), attention over heads (cpu.rs:32), sinu over
 * elements (cpu.rs:56).  Where the reference uses rayon's order-nondeterministic
 * parallel sum (softmax_num, cpu.rs:190) the DEFAULT is one front-to-back fp32 sum
 * (the order of a producer that is never split); oracle_set_softmax_split(levels)
 * gives rayon's halving tree with 2^levels leaves instead (rama_oracle.h).
 */
#include "rama_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 0;
/* The two places where the reference leaves the order of a sum to a crate (see rama_oracle.h): switchable, so that
 * the distance between the reference's own admissible executions can be measured (tools/ref_self_spread.py). */
static int g_lane_reduce = ORACLE_LANES_PAIRWISE;
static int g_softmax_split = 0;

void oracle_set_lane_reduce(int order) { g_lane_reduce = order; }
int  oracle_get_lane_reduce(void) { return g_lane_reduce; }
void oracle_set_softmax_split(int levels) { g_softmax_split = levels < 0 ? 0 : levels; }
int  oracle_get_softmax_split(void) { return g_softmax_split; }

/* wide::f32x4::reduce_add of the four lane sums, cpu.rs:148 */
static inline float reduce_add4(float v0, float v1, float v2, float v3, int order) {
    if (order == ORACLE_LANES_STRIDED) return (v0 + v2) + (v1 + v3);
    if (order == ORACLE_LANES_SEQUENTIAL) return ((v0 + v1) + v2) + v3;
    return (v0 + v1) + (v2 + v3);
}

/* rayon's `par_iter().sum::<f32>()`, cpu.rs:190, with `levels` rounds of halving (the producer is split at len / 2
 * while both halves keep >= 1 element), every leaf summed front to back, the halves' sums added left + right */
static float sum_split(const float *x, size_t n, int levels) {
    if (levels <= 0 || n < 2) {
        float sum = 0.0f;
        for (size_t i = 0; i < n; i++) sum += x[i];
        return sum;
    }
    const size_t mid = n / 2;
    const float left = sum_split(x, mid, levels - 1);
    const float right = sum_split(x + mid, n - mid, levels - 1);
    return left + right;
}

void oracle_set_threads(int n) {
    g_threads = n;
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
}

int oracle_get_threads(void) {
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

/* cpu.rs:16-21  target[i] += source[i] */
void oracle_array_add(float *target, const float *source, size_t n) {
    for (size_t i = 0; i < n; i++) target[i] += source[i];
}

/* cpu.rs:59-64  target[i] *= source[i] */
void oracle_array_mult(float *target, const float *source, size_t n) {
    for (size_t i = 0; i < n; i++) target[i] *= source[i];
}

/* cpu.rs:54-57  a = a * (1.0 / (1.0 + exp(-a))) */
void oracle_sinu(float *o, size_t n) {
#pragma omp parallel for schedule(static) if (n >= 4096)
    for (size_t i = 0; i < n; i++) {
        float a = o[i];
        o[i] = a * (1.0f / (1.0f + expf(-a)));
    }
}

/* cpu.rs:66-72 */
void oracle_copy_from_slice(float *target, const float *source, size_t n) {
    memmove(target, source, n * sizeof(float));
}

/* the libm exp the restatement (and, through Rust's f32::exp, the reference) evaluates,
 * elementwise: lets the tests pin the HIP reference-order kernels' expf to it bit for bit */
void oracle_expf_array(float *o, const float *x, size_t n) {
    for (size_t i = 0; i < n; i++) o[i] = expf(x[i]);
}

/* cpu.rs:99-117: v = 1/sqrt(sum(x*x)/len + 1e-5); o[i] = weight[i] * (v * x[i]) */
void oracle_rmsnorm(float *o, const float *x, const float *weight, size_t n) {
    float ss = 0.0f;
    for (size_t i = 0; i < n; i++) ss += x[i] * x[i];
    float v = 1.0f / sqrtf(ss / (float)n + 1e-5f);
    for (size_t i = 0; i < n; i++) o[i] = weight[i] * (v * x[i]);
}

/* cpu.rs:74-97: adjacent pairs (2i, 2i+1) rotated by (pos_real[i], pos_img[i]) */
void oracle_apply_position(float *q, float *k, const float *pos_real,
                           const float *pos_img, size_t head_size) {
    for (size_t i = 0; i < head_size / 2; i++) {
        float fcr = pos_real[i], fci = pos_img[i];
        float q0 = q[2 * i], q1 = q[2 * i + 1];
        q[2 * i]     = q0 * fcr - q1 * fci;
        q[2 * i + 1] = q0 * fci + q1 * fcr;
        float k0 = k[2 * i], k1 = k[2 * i + 1];
        k[2 * i]     = k0 * fcr - k1 * fci;
        k[2 * i + 1] = k0 * fci + k1 * fcr;
    }
}

/* cpu.rs:127-153.  Per output element idx: r = idx / o_cols, c = idx % o_cols;
 * four lane sums over k = j (mod 4) (wide::f32x4 `v += a_wide * b_wide`, separate
 * multiply and add), then reduce_add.  The order of that final 4-term sum belongs to the
 * `wide` crate and to the build's target features, not to the reference: pairwise
 * (l0+l1)+(l2+l3) (an SSE3 hadd pair), strided (l0+l2)+(l1+l3) (movehl + shuffle, the
 * plain SSE2 idiom) or front to back ((l0+l1)+l2)+l3 (array sum) -- 1 ulp apart at
 * most.  oracle_set_lane_reduce selects it; the default is pairwise (what rounds 1-5
 * of this repo compared against). */
int oracle_matmul(float *o, const float *a, const float *b,
                  size_t width, size_t o_rows, size_t o_cols) {
    if (width % 4 != 0 || o_cols == 0) return -1;
    size_t n_out = o_rows * o_cols;
    const int order = g_lane_reduce;
    /* tiny products stay on the calling thread (fork/join costs more than the work) */
    const int par = (n_out * width) >= ((size_t)1 << 17);
    if (o_cols == 1) {
#pragma omp parallel for schedule(static) if (par)
        for (size_t r = 0; r < n_out; r++) {
            const float *ar = a + r * width;
            float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
            for (size_t k = 0; k < width; k += 4) {
                v0 += ar[k]     * b[k];
                v1 += ar[k + 1] * b[k + 1];
                v2 += ar[k + 2] * b[k + 2];
                v3 += ar[k + 3] * b[k + 3];
            }
            o[r] = reduce_add4(v0, v1, v2, v3, order);
        }
        return 0;
    }
#pragma omp parallel for schedule(static) if (par)
    for (size_t idx = 0; idx < n_out; idx++) {
        size_t r = idx / o_cols, c = idx % o_cols;
        const float *ar = a + r * width;
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
        for (size_t k = 0; k < width; k += 4) {
            v0 += ar[k]     * b[k * o_cols + c];
            v1 += ar[k + 1] * b[(k + 1) * o_cols + c];
            v2 += ar[k + 2] * b[(k + 2) * o_cols + c];
            v3 += ar[k + 3] * b[(k + 3) * o_cols + c];
        }
        o[idx] = reduce_add4(v0, v1, v2, v3, order);
    }
    return 0;
}

/* CPU::softmax_num cpu.rs:187-192 (Device::softmax cpu.rs:119-125 is the same body):
 * max, a = exp(a - max), sum, a /= sum. */
void oracle_softmax(float *x, size_t n) {
    if (n == 0) return;
    float mx = x[0];
    for (size_t i = 1; i < n; i++) mx = x[i] > mx ? x[i] : mx;
    for (size_t i = 0; i < n; i++) x[i] = expf(x[i] - mx);
    const float sum = sum_split(x, n, g_softmax_split);      /* 0 levels: one front-to-back sum */
    for (size_t i = 0; i < n; i++) x[i] /= sum;
}

/* cpu.rs:23-52.  Per head h (parallel): att[t] = (sum_i q[i]*k[t][i]) / sqrt(hs) for
 * t in 0..=pos (sequential fp32 sum, Iterator::sum); softmax over 0..=pos; xb = 0;
 * xb[i] += att[t] * v[t][i] with t ascending.  Cache offset lo + t*dim + h*hs with
 * lo = layer*seq_len*dim (cpu.rs:28,35,45). */
void oracle_multi_head_attention(const oracle_config *cfg, oracle_state *s,
                                 int layer, int pos) {
    const size_t dim = (size_t)cfg->dim;
    const size_t hs = dim / (size_t)cfg->n_heads;
    const size_t lo = (size_t)layer * (size_t)cfg->seq_len * dim;
    const float scale_div = sqrtf((float)hs);
    const int par = ((size_t)(pos + 1) * dim) >= ((size_t)1 << 15);
#pragma omp parallel for schedule(static) if (par)
    for (int h = 0; h < cfg->n_heads; h++) {
        float *att = s->att + (size_t)h * (size_t)cfg->seq_len;
        const float *q = s->q + (size_t)h * hs;
        float *xb = s->xb + (size_t)h * hs;
        for (int t = 0; t <= pos; t++) {
            const float *k = s->key_cache + lo + (size_t)t * dim + (size_t)h * hs;
            float acc = 0.0f;
            for (size_t i = 0; i < hs; i++) acc += q[i] * k[i];
            att[t] = acc / scale_div;
        }
        oracle_softmax(att, (size_t)pos + 1);
        for (size_t i = 0; i < hs; i++) xb[i] = 0.0f;
        for (int t = 0; t <= pos; t++) {
            const float *v = s->value_cache + lo + (size_t)t * dim + (size_t)h * hs;
            float a = att[t];
            for (size_t i = 0; i < hs; i++) xb[i] += a * v[i];
        }
    }
}

/* cpu.rs:165-167: reduce(|(i1,v1),(i2,v2)| if v1 > v2 {(i1,v1)} else {(i2,v2)}):
 * the accumulator survives only on strictly-greater, so the last maximum wins. */
int oracle_argmax(const float *logits, size_t n) {
    size_t bi = 0;
    float bv = logits[0];
    for (size_t i = 1; i < n; i++) {
        if (!(bv > logits[i])) { bi = i; bv = logits[i]; }
    }
    return (int)bi;
}

typedef struct { float p; size_t idx; } prob_index_t;

static int cmp_prob_desc(const void *A, const void *B) {
    const prob_index_t *a = (const prob_index_t *)A, *b = (const prob_index_t *)B;
    if (a->p > b->p) return -1;
    if (a->p < b->p) return 1;
    /* slice::sort_by is stable: equal probabilities keep ascending index order */
    return (a->idx > b->idx) - (a->idx < b->idx);
}

/* infer.rs:55-85 sample_top_q, with `u` in place of rng.gen::<f32>() */
static int sample_top_q(const float *p, size_t num, float topp, float u) {
    float cutoff = (1.0f - topp) / (float)(num - 1);
    prob_index_t *pi = (prob_index_t *)malloc(num * sizeof(prob_index_t));
    size_t m = 0;
    for (size_t i = 0; i < num; i++)
        if (p[i] > cutoff) { pi[m].p = p[i]; pi[m].idx = i; m++; }
    if (m == 0) { free(pi); return -1; } /* reference: `len() - 1` underflow panic */
    qsort(pi, m, sizeof(prob_index_t), cmp_prob_desc);
    float cum = 0.0f;
    size_t last = m - 1;
    for (size_t i = 0; i < m; i++) {
        cum += pi[i].p;
        if (cum > topp) { last = i; break; }
    }
    float r = u * cum;
    float cdf = 0.0f;
    int out = (int)pi[last].idx;
    for (size_t i = 0; i < last; i++) {
        cdf += pi[i].p;
        if (r < cdf) { out = (int)pi[i].idx; break; }
    }
    free(pi);
    return out;
}

/* the same on given probabilities (for the hand-worked vectors of tests/test_sampler_const.py) */
int oracle_sample_top_q(const float *p, size_t num, float topp, float u) { return sample_top_q(p, num, topp, u); }

/* Device::sample cpu.rs:155-179 */
int oracle_sample(float *logits, size_t n, float temperature, float topp, float u) {
    if (temperature == 0.0f) return oracle_argmax(logits, n);
    if (temperature < 1.0f)  /* cpu.rs:170-172: T > 1 has no effect */
        for (size_t i = 0; i < n; i++) logits[i] /= temperature;
    oracle_softmax(logits, n);
    return sample_top_q(logits, n, topp, u);
}

/* infer.rs:8-53 restricted to a layer range */
void oracle_forward_range(const oracle_config *cfg, const oracle_weights *w,
                          oracle_state *s, int token, int pos,
                          int layer_begin, int layer_end, int do_embed, int do_cls) {
    const size_t dim = (size_t)cfg->dim;
    const size_t hidden = (size_t)cfg->hidden_dim;
    const size_t hs = dim / (size_t)cfg->n_heads;

    if (do_embed) /* infer.rs:13 */
        oracle_copy_from_slice(s->x, w->token_embedding_table + (size_t)token * dim, dim);

    const float *pos_real = w->freq_cis_real + (size_t)pos * (hs / 2); /* infer.rs:15 */
    const float *pos_img  = w->freq_cis_imag + (size_t)pos * (hs / 2); /* infer.rs:16 */

    for (int layer = layer_begin; layer < layer_end; layer++) {        /* infer.rs:18 */
        const size_t l = (size_t)layer;
        oracle_rmsnorm(s->xb, s->x, w->rms_att_weight + l * dim, dim); /* :19 */
        /* :20-21 issue the Wq product twice; it is idempotent, once suffices */
        oracle_matmul(s->q, w->wq + l * dim * dim, s->xb, dim, dim, 1);
        oracle_matmul(s->k, w->wk + l * dim * dim, s->xb, dim, dim, 1); /* :22 */
        oracle_matmul(s->v, w->wv + l * dim * dim, s->xb, dim, dim, 1); /* :23 */

        for (int h = 0; h < cfg->n_heads; h++)                          /* :25-29 */
            oracle_apply_position(s->q + (size_t)h * hs, s->k + (size_t)h * hs,
                                  pos_real, pos_img, hs);

        const size_t lo = l * (size_t)cfg->seq_len * dim;               /* :31 */
        oracle_copy_from_slice(s->key_cache + lo + (size_t)pos * dim, s->k, dim);   /* :32 */
        oracle_copy_from_slice(s->value_cache + lo + (size_t)pos * dim, s->v, dim); /* :33 */
        oracle_multi_head_attention(cfg, s, layer, pos);                /* :34 */
        oracle_matmul(s->xb2, w->wo + l * dim * dim, s->xb, dim, dim, 1); /* :35 */
        oracle_array_add(s->x, s->xb2, dim);                            /* :37 */

        oracle_rmsnorm(s->xb, s->x, w->rms_ffn_weight + l * dim, dim);  /* :39 */
        oracle_matmul(s->hb,  w->w1 + l * hidden * dim, s->xb, dim, hidden, 1); /* :41 */
        oracle_matmul(s->hb2, w->w3 + l * hidden * dim, s->xb, dim, hidden, 1); /* :42 */
        oracle_sinu(s->hb, hidden);                                     /* :44 */
        oracle_array_mult(s->hb, s->hb2, hidden);                       /* :45 */
        oracle_matmul(s->xb, w->w2 + l * dim * hidden, s->hb, hidden, dim, 1); /* :46 */
        oracle_array_add(s->x, s->xb, dim);                             /* :47 */
    }
    if (do_cls) {
        oracle_copy_from_slice(s->xb, s->x, dim);                       /* :49 */
        oracle_rmsnorm(s->x, s->xb, w->rms_final_weight, dim);          /* :50 */
        oracle_matmul(s->logits, w->wcls, s->x, dim, (size_t)cfg->vocab_size, 1); /* :51 */
    }
}

void oracle_forward(const oracle_config *cfg, const oracle_weights *w,
                    oracle_state *s, int token, int pos) {
    oracle_forward_range(cfg, w, s, token, pos, 0, cfg->n_layers, 1, 1);
}

/* ---------------- fp64 arbiter (same network, double accumulation) ------------- */

static void mv64(float *o, const float *a, const float *b, size_t width, size_t rows) {
#pragma omp parallel for schedule(static) if (rows * width >= ((size_t)1 << 17))
    for (size_t r = 0; r < rows; r++) {
        double acc = 0.0;
        const float *ar = a + r * width;
        for (size_t k = 0; k < width; k++) acc += (double)ar[k] * (double)b[k];
        o[r] = (float)acc;
    }
}

static void rms64(float *o, const float *x, const float *w, size_t n) {
    double ss = 0.0;
    for (size_t i = 0; i < n; i++) ss += (double)x[i] * (double)x[i];
    double v = 1.0 / sqrt(ss / (double)n + 1e-5);
    for (size_t i = 0; i < n; i++) o[i] = (float)((double)w[i] * (v * (double)x[i]));
}

void oracle_forward_f64(const oracle_config *cfg, const oracle_weights *w,
                        oracle_state *s, int token, int pos) {
    const size_t dim = (size_t)cfg->dim, hidden = (size_t)cfg->hidden_dim;
    const size_t hs = dim / (size_t)cfg->n_heads;
    memcpy(s->x, w->token_embedding_table + (size_t)token * dim, dim * sizeof(float));
    const float *pr = w->freq_cis_real + (size_t)pos * (hs / 2);
    const float *pi = w->freq_cis_imag + (size_t)pos * (hs / 2);
    for (int layer = 0; layer < cfg->n_layers; layer++) {
        const size_t l = (size_t)layer;
        rms64(s->xb, s->x, w->rms_att_weight + l * dim, dim);
        mv64(s->q, w->wq + l * dim * dim, s->xb, dim, dim);
        mv64(s->k, w->wk + l * dim * dim, s->xb, dim, dim);
        mv64(s->v, w->wv + l * dim * dim, s->xb, dim, dim);
        for (size_t j = 0; j < dim / 2; j++) {
            size_t i = j % (hs / 2);
            double c = pr[i], sn = pi[i];
            double q0 = s->q[2 * j], q1 = s->q[2 * j + 1];
            s->q[2 * j] = (float)(q0 * c - q1 * sn); s->q[2 * j + 1] = (float)(q0 * sn + q1 * c);
            double k0 = s->k[2 * j], k1 = s->k[2 * j + 1];
            s->k[2 * j] = (float)(k0 * c - k1 * sn); s->k[2 * j + 1] = (float)(k0 * sn + k1 * c);
        }
        const size_t lo = l * (size_t)cfg->seq_len * dim;
        memcpy(s->key_cache + lo + (size_t)pos * dim, s->k, dim * sizeof(float));
        memcpy(s->value_cache + lo + (size_t)pos * dim, s->v, dim * sizeof(float));
#pragma omp parallel for schedule(static) if (((size_t)(pos + 1) * dim) >= ((size_t)1 << 15))
        for (int h = 0; h < cfg->n_heads; h++) {
            float *att = s->att + (size_t)h * (size_t)cfg->seq_len;
            const float *q = s->q + (size_t)h * hs;
            double mx = -INFINITY;
            for (int t = 0; t <= pos; t++) {
                const float *k = s->key_cache + lo + (size_t)t * dim + (size_t)h * hs;
                double acc = 0.0;
                for (size_t i = 0; i < hs; i++) acc += (double)q[i] * (double)k[i];
                acc /= sqrt((double)hs);
                att[t] = (float)acc;
                if (acc > mx) mx = acc;
            }
            double sum = 0.0;
            for (int t = 0; t <= pos; t++) sum += exp((double)att[t] - mx);
            for (size_t i = 0; i < hs; i++) {
                double acc = 0.0;
                for (int t = 0; t <= pos; t++) {
                    const float *v = s->value_cache + lo + (size_t)t * dim + (size_t)h * hs;
                    acc += exp((double)att[t] - mx) / sum * (double)v[i];
                }
                s->xb[(size_t)h * hs + i] = (float)acc;
            }
        }
        mv64(s->xb2, w->wo + l * dim * dim, s->xb, dim, dim);
        for (size_t i = 0; i < dim; i++) s->x[i] += s->xb2[i];
        rms64(s->xb, s->x, w->rms_ffn_weight + l * dim, dim);
        mv64(s->hb, w->w1 + l * hidden * dim, s->xb, dim, hidden);
        mv64(s->hb2, w->w3 + l * hidden * dim, s->xb, dim, hidden);
        for (size_t i = 0; i < hidden; i++) {
            double a = s->hb[i];
            s->hb[i] = (float)(a * (1.0 / (1.0 + exp(-a))) * (double)s->hb2[i]);
        }
        mv64(s->xb, w->w2 + l * dim * hidden, s->hb, hidden, dim);
        for (size_t i = 0; i < dim; i++) s->x[i] += s->xb[i];
    }
    memcpy(s->xb, s->x, dim * sizeof(float));
    rms64(s->x, s->xb, w->rms_final_weight, dim);
    mv64(s->logits, w->wcls, s->x, dim, (size_t)cfg->vocab_size);
}

/* ---------------- synthetic weights ------------------------------------------- */

static inline uint64_t splitmix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

void oracle_fill_synth(float *dst, size_t n, uint64_t seed, uint64_t tag,
                       uint64_t offset, float scale, float bias) {
    const uint64_t base = offset + tag * 0x9E3779B97F4A7C15ULL + seed * 0xD1B54A32D192ED03ULL;
#pragma omp parallel for schedule(static) if (n >= ((size_t)1 << 16))
    for (size_t i = 0; i < n; i++) {
        uint64_t z = splitmix64((uint64_t)i + base);
        int32_t sum = (int32_t)(z & 0xFFFF) + (int32_t)((z >> 16) & 0xFFFF) +
                      (int32_t)((z >> 32) & 0xFFFF) + (int32_t)(z >> 48);
        dst[i] = bias + (float)(sum - 131070) * scale; /* -ffp-contract=off: mul, then add */
    }
}
