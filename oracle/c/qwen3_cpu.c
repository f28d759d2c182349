/*
 * CPU port of the reference's Qwen3 CPU decode path (TEST INFRASTRUCTURE ONLY -- see
 * oracle/__init__.py).  Used (a) as bench.py's `cpu_baseline` ("port": the reference itself
 * cannot be built here -- no Rust toolchain, candle 0.11 not vendored) and (b) as a second,
 * independently written checker against the numpy oracle.
 *
 * Follows the reference's CPU branch (crane-core/src/models/qwen3/modeling.rs):
 *   merged QKV projection            :318-323
 *   per-head RMSNorm(q,k) then RoPE   :341-359  (rotate-half, modules/rotary.rs:29-46,372-409)
 *   KV cache append                   :366      (modules/kv_cache.rs:38-101)
 *   B=1 flash attention, online softmax, f32 accumulators, GQA by integer division :380-456
 *   o_proj, SwiGLU MLP (merged gate||up) :419, :608-631
 *   final norm + lm_head on the LAST position only :1024-1035
 * cfg.kv_bf16 selects the rounding of K/V rows as they enter the cache: 0 none (the reference's F32 CPU cache), 1 bf16 (the
 * model dtype of its GPU path), 2 IEEE binary16 (the device's default CM_KV_F16 pages).
 * Weights are bf16-stored / f32-computed (candle CPU has no bf16 matmul, modeling.rs:1630-1631:
 * the reference's CPU dtype is F32; we keep the bf16 storage so a 8B model fits and state so).
 * Deterministic synthetic weights: same generator as crane_amd/synth.py (bit-identical).
 *
 * Build: make -C oracle/c   (gcc -O3 -march=native -fopenmp -ffp-contract=off)
 */
#include "qc_common.h"

typedef struct {
    int V, H, I, L, Hq, Hkv, D, max_seq;
    float eps;
    double theta;
    int tie, qk_norm, kv_bf16;
} qc_cfg;

typedef struct {
    uint16_t *qkv, *o, *gate_up, *down, *ln1, *ln2, *qn, *kn;
    float *k, *v;              /* [Hkv, max_seq, D] */
} qc_layer;

typedef struct qc_model {
    qc_cfg c;
    uint16_t *embed, *lm_head, *norm;
    qc_layer* layers;
    float *cos, *sin;
    float *x, *xn, *qkv, *attn, *gu, *h;
    int len;
} qc_model;

float qc_f16_round(float f) { return f16_round(f); }     /* known-answer hook: tests/test_c_oracle.py pins it on numpy's float16 */

/* checkpoint name prefix of the decoder's tensors: "model." (Qwen3ForCausalLM), "model.language_model." when the decoder is the
 * text model of a vision-language checkpoint (prefix probing, qwen3_5/model.rs:65-74); applies to the next qc_create */
static char g_prefix[64] = "model.";
void qc_set_name_prefix(const char* p) { snprintf(g_prefix, sizeof g_prefix, "%s", p && *p ? p : "model."); }

qc_model* qc_create(const qc_cfg* cfg, uint64_t seed) {
    qc_model* m = (qc_model*)calloc(1, sizeof *m);
    m->c = *cfg;
    const qc_cfg* c = &m->c;
    const int H = c->H, D = c->D, I = c->I, qd = c->Hq * D, kd = c->Hkv * D;
    char name[256];
    m->embed = xmalloc((size_t)c->V * H * 2);
    snprintf(name, sizeof name, "%sembed_tokens.weight", g_prefix);
    synth_rows(m->embed, H, name, seed, 1.0, 0.f, 0, c->V, H);
    m->norm = xmalloc((size_t)H * 2);
    snprintf(name, sizeof name, "%snorm.weight", g_prefix);
    synth_rows(m->norm, H, name, seed, 0.1, 1.f, 0, 1, H);
    if (c->tie) m->lm_head = m->embed;      /* tied: same tensor (modeling.rs:786-794) */
    else { m->lm_head = xmalloc((size_t)c->V * H * 2); synth_rows(m->lm_head, H, "lm_head.weight", seed, 1.0 / sqrt((double)H), 0.f, 0, c->V, H); }
    m->layers = (qc_layer*)calloc((size_t)c->L, sizeof(qc_layer));
    for (int li = 0; li < c->L; ++li) {
        qc_layer* w = &m->layers[li];
        const double sH = 1.0 / sqrt((double)H);
#define NM(suffix) (snprintf(name, sizeof name, "%slayers.%d.%s", g_prefix, li, suffix), name)
        w->qkv = xmalloc((size_t)(qd + 2 * kd) * H * 2);          /* cat(q,k,v) rows (modeling.rs:187-204) */
        synth_rows(w->qkv, H, NM("self_attn.q_proj.weight"), seed, sH, 0.f, 0, qd, H);
        synth_rows(w->qkv + (size_t)qd * H, H, NM("self_attn.k_proj.weight"), seed, sH, 0.f, 0, kd, H);
        synth_rows(w->qkv + (size_t)(qd + kd) * H, H, NM("self_attn.v_proj.weight"), seed, sH, 0.f, 0, kd, H);
        w->o = xmalloc((size_t)H * qd * 2);
        synth_rows(w->o, qd, NM("self_attn.o_proj.weight"), seed, 1.0 / sqrt((double)qd), 0.f, 0, H, qd);
        if (c->qk_norm) {
            w->qn = xmalloc((size_t)D * 2); w->kn = xmalloc((size_t)D * 2);
            synth_rows(w->qn, D, NM("self_attn.q_norm.weight"), seed, 0.1, 1.f, 0, 1, D);
            synth_rows(w->kn, D, NM("self_attn.k_norm.weight"), seed, 0.1, 1.f, 0, 1, D);
        }
        w->gate_up = xmalloc((size_t)2 * I * H * 2);              /* gate || up (modeling.rs:582-588) */
        synth_rows(w->gate_up, H, NM("mlp.gate_proj.weight"), seed, sH, 0.f, 0, I, H);
        synth_rows(w->gate_up + (size_t)I * H, H, NM("mlp.up_proj.weight"), seed, sH, 0.f, 0, I, H);
        w->down = xmalloc((size_t)H * I * 2);
        synth_rows(w->down, I, NM("mlp.down_proj.weight"), seed, 1.0 / sqrt((double)I), 0.f, 0, H, I);
        w->ln1 = xmalloc((size_t)H * 2); w->ln2 = xmalloc((size_t)H * 2);
        synth_rows(w->ln1, H, NM("input_layernorm.weight"), seed, 0.1, 1.f, 0, 1, H);
        synth_rows(w->ln2, H, NM("post_attention_layernorm.weight"), seed, 0.1, 1.f, 0, 1, H);
        w->k = (float*)xmalloc((size_t)kd * c->max_seq * 4);
        w->v = (float*)xmalloc((size_t)kd * c->max_seq * 4);
#undef NM
    }
    /* RotaryEmbedding::new (rotary.rs:29-46): inv_freq f64 -> f32, freqs = pos*inv in f32 */
    const int half = D / 2;
    m->cos = (float*)xmalloc((size_t)c->max_seq * half * 4);
    m->sin = (float*)xmalloc((size_t)c->max_seq * half * 4);
    for (int p = 0; p < c->max_seq; ++p)
        for (int i = 0; i < half; ++i) {
            const float inv = (float)(1.0 / pow(c->theta, (double)(2 * i) / (double)D));
            const float f = (float)p * inv;
            m->cos[(size_t)p * half + i] = cosf(f);
            m->sin[(size_t)p * half + i] = sinf(f);
        }
    m->x = (float*)xmalloc((size_t)H * 4); m->xn = (float*)xmalloc((size_t)H * 4);
    m->qkv = (float*)xmalloc((size_t)(qd + 2 * kd) * 4); m->attn = (float*)xmalloc((size_t)qd * 4);
    m->gu = (float*)xmalloc((size_t)2 * I * 4); m->h = (float*)xmalloc((size_t)I * 4);
    return m;
}

void qc_destroy(qc_model* m) {
    if (!m) return;
    for (int li = 0; li < m->c.L; ++li) {
        qc_layer* w = &m->layers[li];
        free(w->qkv); free(w->o); free(w->gate_up); free(w->down); free(w->ln1); free(w->ln2); free(w->qn); free(w->kn);
        free(w->k); free(w->v);
    }
    if (m->lm_head != m->embed) free(m->lm_head);
    free(m->embed); free(m->norm); free(m->layers); free(m->cos); free(m->sin);
    free(m->x); free(m->xn); free(m->qkv); free(m->attn); free(m->gu); free(m->h);
    free(m);
}

static void rms_norm(const float* x, const uint16_t* w, float* out, int n, float eps) {
    float ss = 0.f;
    for (int i = 0; i < n; ++i) ss += x[i] * x[i];
    const float r = 1.0f / sqrtf(ss / (float)n + eps);
    for (int i = 0; i < n; ++i) out[i] = x[i] * r * bf2f(w[i]);
}

static void decode_one(qc_model* m, uint32_t tok, int pos, float* logits /* or NULL */) {
    const qc_cfg* c = &m->c;
    const int H = c->H, D = c->D, I = c->I, Hq = c->Hq, Hkv = c->Hkv, qd = Hq * D, kd = Hkv * D, half = D / 2;
    const int n_rep = Hq / Hkv;
    const float scale = (float)(1.0 / sqrt((double)D));
    for (int i = 0; i < H; ++i) m->x[i] = bf2f(m->embed[(size_t)tok * H + i]);
    for (int li = 0; li < c->L; ++li) {
        qc_layer* w = &m->layers[li];
        rms_norm(m->x, w->ln1, m->xn, H, c->eps);
        gemv(w->qkv, m->xn, m->qkv, qd + 2 * kd, H);
        float* q = m->qkv; float* k = m->qkv + qd; float* v = m->qkv + qd + kd;
        const float* cs = m->cos + (size_t)pos * half; const float* sn = m->sin + (size_t)pos * half;
        for (int h = 0; h < Hq + Hkv; ++h) {                      /* q heads then k heads */
            float* p = (h < Hq) ? q + (size_t)h * D : k + (size_t)(h - Hq) * D;
            if (c->qk_norm) {
                const uint16_t* nw = (h < Hq) ? w->qn : w->kn;
                float ss = 0.f;
                for (int i = 0; i < D; ++i) ss += p[i] * p[i];
                const float r = 1.0f / sqrtf(ss / (float)D + c->eps);
                for (int i = 0; i < D; ++i) p[i] = p[i] * r * bf2f(nw[i]);
            }
            for (int i = 0; i < half; ++i) {
                const float x1 = p[i], x2 = p[i + half];
                p[i] = x1 * cs[i] - x2 * sn[i];
                p[i + half] = x1 * sn[i] + x2 * cs[i];
            }
        }
        for (int g = 0; g < Hkv; ++g)
            for (int i = 0; i < D; ++i) {
                float kk = k[(size_t)g * D + i], vv = v[(size_t)g * D + i];
                if (c->kv_bf16 == 1) { kk = bf2f(f2bf(kk)); vv = bf2f(f2bf(vv)); }
                else if (c->kv_bf16 == 2) { kk = f16_round(kk); vv = f16_round(vv); }
                w->k[((size_t)g * c->max_seq + pos) * D + i] = kk;
                w->v[((size_t)g * c->max_seq + pos) * D + i] = vv;
            }
        /* single-pass online-softmax attention per head (candle cpu flash_attn semantics) */
#pragma omp parallel for schedule(static)
        for (int h = 0; h < Hq; ++h) {
            const float* qh = q + (size_t)h * D;
            const float* kh = w->k + (size_t)(h / n_rep) * c->max_seq * D;
            const float* vh = w->v + (size_t)(h / n_rep) * c->max_seq * D;
            float mx = -INFINITY, l = 0.f;
            float acc[256];
            for (int i = 0; i < D; ++i) acc[i] = 0.f;
            for (int j = 0; j <= pos; ++j) {
                float s = 0.f;
                for (int i = 0; i < D; ++i) s += qh[i] * kh[(size_t)j * D + i];
                s *= scale;
                const float mn = s > mx ? s : mx;
                const float a = expf(mx - mn), p = expf(s - mn);
                l = l * a + p;
                for (int i = 0; i < D; ++i) acc[i] = acc[i] * a + p * vh[(size_t)j * D + i];
                mx = mn;
            }
            for (int i = 0; i < D; ++i) m->attn[(size_t)h * D + i] = acc[i] / l;
        }
        gemv(w->o, m->attn, m->xn, H, qd);
        for (int i = 0; i < H; ++i) m->x[i] += m->xn[i];
        rms_norm(m->x, w->ln2, m->xn, H, c->eps);
        gemv(w->gate_up, m->xn, m->gu, 2 * I, H);
        for (int i = 0; i < I; ++i) { const float g = m->gu[i]; m->h[i] = (g / (1.0f + expf(-g))) * m->gu[I + i]; }
        gemv(w->down, m->h, m->xn, H, I);
        for (int i = 0; i < H; ++i) m->x[i] += m->xn[i];
    }
    if (logits) {
        rms_norm(m->x, m->norm, m->xn, H, c->eps);
        gemv(m->lm_head, m->xn, logits, c->V, H);
    }
}

/* forward_step: tokens processed causally one position at a time (same math as a batched
 * causal prefill); logits of the last position only. */
int qc_forward(qc_model* m, const uint32_t* ids, int n, int start_pos, float* logits) {
    if (!m || n <= 0 || start_pos < 0 || start_pos + n > m->c.max_seq) return -1;
    for (int i = 0; i < n; ++i) {
        if (ids[i] >= (uint32_t)m->c.V) return -2;
        decode_one(m, ids[i], start_pos + i, (i == n - 1) ? logits : NULL);
    }
    m->len = start_pos + n;
    return 0;
}

/* Y[t*ldy + n] = W[n,:] . X[t*ldx + :]  for T token rows: the weight matrix is streamed ONCE for all rows (a 1024-token prompt
 * costs 1024 decode steps' flops but one pass over the 16 GB of weights instead of 1024).  Every (n, t) dot product is the
 * arithmetic of gemv() -- the same 16 lane accumulators in ascending k, the same tail, the same fixed tree -- so a row of the
 * result is bit-identical to the token-serial forward (tests/test_c_oracle.py::test_batched_prompt_pass_is_bit_identical). */
#define QC_RB 4
#define QC_TB 6
static void gemm_rows(const uint16_t* W, const float* X, size_t ldx, float* Y, size_t ldy, int N, int K, int T) {
#pragma omp parallel for schedule(dynamic, 4)
    for (int n0 = 0; n0 < N; n0 += QC_RB) {
        const int nr = N - n0 < QC_RB ? N - n0 : QC_RB;
        for (int t0 = 0; t0 < T; t0 += QC_TB) {
            const int nt = T - t0 < QC_TB ? T - t0 : QC_TB;
            v16f acc[QC_RB][QC_TB];
            for (int r = 0; r < QC_RB; ++r) for (int t = 0; t < QC_TB; ++t) acc[r][t] = (v16f){0};
            int k = 0;
            if (nr == QC_RB && nt == QC_TB) {
                for (; k + 16 <= K; k += 16) {
                    v16f f[QC_RB];
                    for (int r = 0; r < QC_RB; ++r) {
                        v16h h; memcpy(&h, W + (size_t)(n0 + r) * K + k, sizeof h);
                        const v16u u = __builtin_convertvector(h, v16u) << 16;
                        memcpy(&f[r], &u, sizeof f[r]);
                    }
                    for (int t = 0; t < QC_TB; ++t) {
                        v16f xv; memcpy(&xv, X + (size_t)(t0 + t) * ldx + k, sizeof xv);
                        for (int r = 0; r < QC_RB; ++r) acc[r][t] += f[r] * xv;
                    }
                }
            } else {
                for (; k + 16 <= K; k += 16)
                    for (int r = 0; r < nr; ++r) {
                        v16h h; v16f f; memcpy(&h, W + (size_t)(n0 + r) * K + k, sizeof h);
                        const v16u u = __builtin_convertvector(h, v16u) << 16;
                        memcpy(&f, &u, sizeof f);
                        for (int t = 0; t < nt; ++t) {
                            v16f xv; memcpy(&xv, X + (size_t)(t0 + t) * ldx + k, sizeof xv);
                            acc[r][t] += f * xv;
                        }
                    }
            }
            for (int r = 0; r < nr; ++r)
                for (int t = 0; t < nt; ++t) {
                    float a[16];
                    memcpy(a, &acc[r][t], sizeof a);
                    const uint16_t* w = W + (size_t)(n0 + r) * K;
                    const float* x = X + (size_t)(t0 + t) * ldx;
                    for (int kk = k; kk < K; ++kk) a[kk & 15] += bf2f(w[kk]) * x[kk];
                    float s8[8];
                    for (int j = 0; j < 8; ++j) s8[j] = a[j] + a[j + 8];
                    Y[(size_t)(t0 + t) * ldy + n0 + r] = ((s8[0] + s8[4]) + (s8[1] + s8[5])) + ((s8[2] + s8[6]) + (s8[3] + s8[7]));
                }
        }
    }
}

/* qc_forward for a whole prompt, layer-major (the shape of the reference's batched causal prefill, modeling.rs:984-1036 with
 * seq_len > 1): per layer every position is normed, projected, rotated and appended, attends causally over the cache, and goes
 * through the MLP; logits of the LAST position only.  Position by position it is decode_one()'s arithmetic, so the result is
 * bit-identical to qc_forward(); what changes is that each weight matrix is read once.  Used for long model-written caches
 * (bench.py parity.model_written_cache_ctx1024: 1024 tokens of Qwen3-8B). */
int qc_forward_batched(qc_model* m, const uint32_t* ids, int n, int start_pos, float* logits) {
    if (!m || n <= 0 || start_pos < 0 || start_pos + n > m->c.max_seq) return -1;
    const qc_cfg* c = &m->c;
    const int H = c->H, D = c->D, I = c->I, Hq = c->Hq, Hkv = c->Hkv, qd = Hq * D, kd = Hkv * D, half = D / 2;
    const int n_rep = Hq / Hkv, QR = qd + 2 * kd;
    const float scale = (float)(1.0 / sqrt((double)D));
    for (int i = 0; i < n; ++i) if (ids[i] >= (uint32_t)c->V) return -2;
    float* X = (float*)xmalloc((size_t)n * H * 4);
    float* XN = (float*)xmalloc((size_t)n * H * 4);
    float* QKV = (float*)xmalloc((size_t)n * QR * 4);
    float* AT = (float*)xmalloc((size_t)n * qd * 4);
    float* GU = (float*)xmalloc((size_t)n * 2 * I * 4);
    float* HB = (float*)xmalloc((size_t)n * I * 4);
    for (int t = 0; t < n; ++t)
        for (int i = 0; i < H; ++i) X[(size_t)t * H + i] = bf2f(m->embed[(size_t)ids[t] * H + i]);
    for (int li = 0; li < c->L; ++li) {
        qc_layer* w = &m->layers[li];
#pragma omp parallel for schedule(static)
        for (int t = 0; t < n; ++t) rms_norm(X + (size_t)t * H, w->ln1, XN + (size_t)t * H, H, c->eps);
        gemm_rows(w->qkv, XN, (size_t)H, QKV, (size_t)QR, QR, H, n);
#pragma omp parallel for schedule(static)
        for (int t = 0; t < n; ++t) {
            const int pos = start_pos + t;
            float* q = QKV + (size_t)t * QR; float* k = q + qd; float* v = q + qd + kd;
            const float* cs = m->cos + (size_t)pos * half; const float* sn = m->sin + (size_t)pos * half;
            for (int h = 0; h < Hq + Hkv; ++h) {
                float* p = (h < Hq) ? q + (size_t)h * D : k + (size_t)(h - Hq) * D;
                if (c->qk_norm) {
                    const uint16_t* nw = (h < Hq) ? w->qn : w->kn;
                    float ss = 0.f;
                    for (int i = 0; i < D; ++i) ss += p[i] * p[i];
                    const float r = 1.0f / sqrtf(ss / (float)D + c->eps);
                    for (int i = 0; i < D; ++i) p[i] = p[i] * r * bf2f(nw[i]);
                }
                for (int i = 0; i < half; ++i) {
                    const float x1 = p[i], x2 = p[i + half];
                    p[i] = x1 * cs[i] - x2 * sn[i];
                    p[i + half] = x1 * sn[i] + x2 * cs[i];
                }
            }
            for (int g = 0; g < Hkv; ++g)
                for (int i = 0; i < D; ++i) {
                    float kk = k[(size_t)g * D + i], vv = v[(size_t)g * D + i];
                    if (c->kv_bf16 == 1) { kk = bf2f(f2bf(kk)); vv = bf2f(f2bf(vv)); }
                    else if (c->kv_bf16 == 2) { kk = f16_round(kk); vv = f16_round(vv); }
                    w->k[((size_t)g * c->max_seq + pos) * D + i] = kk;
                    w->v[((size_t)g * c->max_seq + pos) * D + i] = vv;
                }
        }
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
        for (int t = 0; t < n; ++t)
            for (int h = 0; h < Hq; ++h) {
                const int pos = start_pos + t;
                const float* qh = QKV + (size_t)t * QR + (size_t)h * D;
                const float* kh = w->k + (size_t)(h / n_rep) * c->max_seq * D;
                const float* vh = w->v + (size_t)(h / n_rep) * c->max_seq * D;
                float mx = -INFINITY, l = 0.f;
                float acc[256];
                for (int i = 0; i < D; ++i) acc[i] = 0.f;
                for (int j = 0; j <= pos; ++j) {
                    float s = 0.f;
                    for (int i = 0; i < D; ++i) s += qh[i] * kh[(size_t)j * D + i];
                    s *= scale;
                    const float mn = s > mx ? s : mx;
                    const float a = expf(mx - mn), p = expf(s - mn);
                    l = l * a + p;
                    for (int i = 0; i < D; ++i) acc[i] = acc[i] * a + p * vh[(size_t)j * D + i];
                    mx = mn;
                }
                for (int i = 0; i < D; ++i) AT[(size_t)t * qd + (size_t)h * D + i] = acc[i] / l;
            }
        gemm_rows(w->o, AT, (size_t)qd, XN, (size_t)H, H, qd, n);
#pragma omp parallel for schedule(static)
        for (int t = 0; t < n; ++t) {
            float* x = X + (size_t)t * H; const float* y = XN + (size_t)t * H;
            for (int i = 0; i < H; ++i) x[i] += y[i];
        }
#pragma omp parallel for schedule(static)
        for (int t = 0; t < n; ++t) rms_norm(X + (size_t)t * H, w->ln2, XN + (size_t)t * H, H, c->eps);
        gemm_rows(w->gate_up, XN, (size_t)H, GU, (size_t)2 * I, 2 * I, H, n);
#pragma omp parallel for schedule(static)
        for (int t = 0; t < n; ++t)
            for (int i = 0; i < I; ++i) { const float g = GU[(size_t)t * 2 * I + i]; HB[(size_t)t * I + i] = (g / (1.0f + expf(-g))) * GU[(size_t)t * 2 * I + I + i]; }
        gemm_rows(w->down, HB, (size_t)I, XN, (size_t)H, H, I, n);
#pragma omp parallel for schedule(static)
        for (int t = 0; t < n; ++t) {
            float* x = X + (size_t)t * H; const float* y = XN + (size_t)t * H;
            for (int i = 0; i < H; ++i) x[i] += y[i];
        }
    }
    if (logits) {
        rms_norm(X + (size_t)(n - 1) * H, m->norm, m->xn, H, c->eps);
        gemv(m->lm_head, m->xn, logits, c->V, H);
    }
    memcpy(m->x, X + (size_t)(n - 1) * H, (size_t)H * 4);
    free(X); free(XN); free(QKV); free(AT); free(GU); free(HB);
    m->len = start_pos + n;
    return 0;
}

/* fill the KV cache of positions [0, ctx) with deterministic values (bench set-up only) */
void qc_fill_kv(qc_model* m, int ctx, uint64_t seed) {
    const qc_cfg* c = &m->c;
    for (int li = 0; li < c->L; ++li)
        for (int kv = 0; kv < 2; ++kv) {
            float* dst = kv ? m->layers[li].v : m->layers[li].k;
            const uint32_t ts = fmix32((uint32_t)seed * 2654435761u + (uint32_t)li * 2 + 1 + kv);
#pragma omp parallel for schedule(static)
            for (int g = 0; g < c->Hkv; ++g)
                for (int p = 0; p < ctx; ++p)
                    for (int i = 0; i < c->D; ++i) {
                        const uint32_t h = fmix32((uint32_t)(((size_t)g * ctx + p) * c->D + i) * 0x9E3779B1u + ts);
                        const int k = (int)((h & 0xFF) + ((h >> 8) & 0xFF) + ((h >> 16) & 0xFF) + (h >> 24)) - 510;
                        dst[((size_t)g * c->max_seq + p) * c->D + i] = bf2f(f2bf((float)k * (1.0f / 147.80054f)));
                    }
        }
    m->len = ctx;
}

/* the same fill in the element order of the device's paged pool ([page][kv head][token in page][D], `page` tokens per
 * page, whole pages): bit-identical to cm_debug_fill_kv (crane_amd/csrc/kernels_misc.hip kv_fill_kernel), so a decode
 * step at the benchmark's context can be compared logit by logit */
void qc_fill_kv_paged(qc_model* m, int ctx, uint64_t seed, int page) {
    const qc_cfg* c = &m->c;
    for (int li = 0; li < c->L; ++li)
        for (int kv = 0; kv < 2; ++kv) {
            float* dst = kv ? m->layers[li].v : m->layers[li].k;
            const uint32_t ts = fmix32((uint32_t)seed * 2654435761u + (uint32_t)li * 2 + 1 + kv);
#pragma omp parallel for schedule(static)
            for (int g = 0; g < c->Hkv; ++g)
                for (int p = 0; p < ctx; ++p)
                    for (int i = 0; i < c->D; ++i) {
                        const size_t idx = (((size_t)(p / page) * c->Hkv + g) * page + (size_t)(p % page)) * c->D + i;
                        const uint32_t h = fmix32((uint32_t)idx * 0x9E3779B1u + ts);
                        const int k = (int)((h & 0xFF) + ((h >> 8) & 0xFF) + ((h >> 16) & 0xFF) + (h >> 24)) - 510;
                        dst[((size_t)g * c->max_seq + p) * c->D + i] = bf2f(f2bf((float)k * (1.0f / 147.80054f)));
                    }
        }
    m->len = ctx;
}

/* OpenMP team size of the forward (default: OMP_NUM_THREADS / all visible CPUs).  More threads than physical cores of
 * one socket made the GPU hosts SLOWER (barrier cost, remote NUMA reads), so the caller sizes it. */
void qc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int qc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
