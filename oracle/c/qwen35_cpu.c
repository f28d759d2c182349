/*
 * CPU port of the reference's Qwen 3.5 / 3.6 / 3.8 hybrid decode path -- Gated Delta Net layers + gated softmax attention
 * (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).  Used (a) as bench.py's `cpu_baseline` / in-run parity checker
 * for BASELINE configs[2] (Qwen3.5-0.8B) -- kind "port": the reference needs a Rust toolchain + candle 0.11, both absent --
 * and (b) as a second, independently written checker next to the numpy oracle (oracle/qwen3_5_oracle.py).
 *
 * Token-serial (every position is a decode step: the reference's S = 1 branches), f32 compute on bf16-stored weights.
 * Follows (paths under crane-core/src):
 *   models/qwen3_5/modeling.rs:45-79      Qwen35RmsNorm        x / sqrt(mean(x^2) + eps) * (1 + w)
 *   models/qwen3_5/modeling.rs:106-124    MRotaryEmbedding::new: base^e in F32 over rot_dim, freqs = pos * inv in f32
 *   models/qwen3_5/modeling.rs:263-279    apply_mrope: rotate-half inside the first rot_dim dims, the rest passes through
 *   models/qwen3_5/modeling.rs:413-564    FullAttention: q_proj -> per-head [q | gate] (:428-455), QK-norm (1 + w) (:464-468),
 *                                         GQA-grouped decode (:481-524), y * sigmoid(gate) (:516-522), o_proj
 *   models/qwen3_5/modeling.rs:576-629    Mlp: silu(gate) * up -> down (three linears)
 *   models/qwen3_5/model.rs:395-510       embed -> layers (full_attention_interval, config.rs:231-241) -> head (last position)
 *   ops/gdn/layer.rs:122-238              GatedDeltaNet::forward, split_qkv, value head -> key head v / (NV / NK) (Interleaved)
 *   ops/gdn/conv.rs:23-133                depthwise causal conv1d (k = 4) + SiLU, rolling window of k - 1 inputs
 *   ops/gdn/backend.rs:26-71,169-211      l2_norm (eps 1e-6), beta = sigmoid(b), g = -exp(A_log) * softplus(a + dt_bias)
 *   ops/gdn/backend.rs:90-156             delta rule: S *= exp(g); kv = S^T k; delta = (v - kv) beta; S += k delta^T; y = S^T (q / sqrt(K))
 *   ops/gdn/norm.rs:39-45                 RmsNormGated: rms_norm(y, w) * silu(z), PLAIN weight
 * cfg.kv_round: rounding of K/V rows entering the cache (0 none = the reference's F32 CPU cache, 1 bf16, 2 binary16).
 * Pinned on the HF Qwen3_5ForCausalLM fixtures tests/golden/qwen3_5_*.npz (tests/test_c_oracle.py).
 */
#include "qc_common.h"

typedef struct {
    int V, H, I, L, Hq, Hkv, D, max_seq;
    float eps;
    double theta;
    int tie, rot_dim, interval, NK, NV, Kd, Vd, conv_k, kv_round;
} q5_cfg;

typedef struct {
    int full;
    uint16_t *q, *k, *v, *o, *qn, *kn;                                         /* full attention */
    uint16_t *in_qkv, *in_z, *in_b, *in_a, *conv_w, *A_log, *dt_bias, *gnorm, *out_proj;   /* Gated Delta Net */
    uint16_t *gate, *up, *down, *ln1, *ln2;
    float *kc, *vc;            /* [Hkv, max_seq, D] */
    float *conv_state;         /* [conv_dim][conv_k - 1], oldest first */
    float *state;              /* [NV][Kd][Vd] */
} q5_layer;

typedef struct q5_model {
    q5_cfg c;
    uint16_t *embed, *lm_head, *norm;
    q5_layer* layers;
    float *cos, *sin;          /* [max_seq, rot_dim / 2] */
    float *x, *xn, *big, *attn, *g, *u, *h, *mixed, *z, *ba;
    int len;
} q5_model;

static uint16_t* synth_new(const char* name, uint64_t seed, double std, float off, int rows, int cols) {
    uint16_t* p = (uint16_t*)xmalloc((size_t)rows * cols * 2);
    synth_rows(p, (size_t)cols, name, seed, std, off, 0, rows, cols);
    return p;
}

static int q5_full(const q5_cfg* c, int li) { return ((li + 1) % c->interval) == 0; }

q5_model* q5_create(const q5_cfg* cfg, uint64_t seed) {
    q5_model* m = (q5_model*)calloc(1, sizeof *m);
    m->c = *cfg;
    const q5_cfg* c = &m->c;
    const int H = c->H, D = c->D, I = c->I, KD = c->NK * c->Kd, VD = c->NV * c->Vd, CD = 2 * KD + VD;
    const double sH = 1.0 / sqrt((double)H);
    char name[256];
    m->embed = synth_new("model.embed_tokens.weight", seed, 1.0, 0.f, c->V, H);
    m->norm = synth_new("model.norm.weight", seed, 0.1, 0.f, 1, H);
    m->lm_head = c->tie ? m->embed : synth_new("lm_head.weight", seed, sH, 0.f, c->V, H);
    m->layers = (q5_layer*)calloc((size_t)c->L, sizeof(q5_layer));
    for (int li = 0; li < c->L; ++li) {
        q5_layer* w = &m->layers[li];
#define NM(suffix) (snprintf(name, sizeof name, "model.layers.%d.%s", li, suffix), name)
        w->full = q5_full(c, li);
        if (w->full) {
            w->q = synth_new(NM("self_attn.q_proj.weight"), seed, sH, 0.f, c->Hq * D * 2, H);
            w->k = synth_new(NM("self_attn.k_proj.weight"), seed, sH, 0.f, c->Hkv * D, H);
            w->v = synth_new(NM("self_attn.v_proj.weight"), seed, sH, 0.f, c->Hkv * D, H);
            w->o = synth_new(NM("self_attn.o_proj.weight"), seed, 1.0 / sqrt((double)(c->Hq * D)), 0.f, H, c->Hq * D);
            w->qn = synth_new(NM("self_attn.q_norm.weight"), seed, 0.1, 0.f, 1, D);
            w->kn = synth_new(NM("self_attn.k_norm.weight"), seed, 0.1, 0.f, 1, D);
            w->kc = (float*)xmalloc((size_t)c->Hkv * c->max_seq * D * 4);
            w->vc = (float*)xmalloc((size_t)c->Hkv * c->max_seq * D * 4);
        } else {
            w->in_qkv = synth_new(NM("linear_attn.in_proj_qkv.weight"), seed, sH, 0.f, CD, H);
            w->in_z = synth_new(NM("linear_attn.in_proj_z.weight"), seed, sH, 0.f, VD, H);
            w->in_b = synth_new(NM("linear_attn.in_proj_b.weight"), seed, sH, 0.f, c->NV, H);
            w->in_a = synth_new(NM("linear_attn.in_proj_a.weight"), seed, sH, 0.f, c->NV, H);
            w->conv_w = synth_new(NM("linear_attn.conv1d.weight"), seed, 0.5, 0.f, CD, c->conv_k);
            w->A_log = synth_new(NM("linear_attn.A_log"), seed, 0.1, -2.f, 1, c->NV);
            w->dt_bias = synth_new(NM("linear_attn.dt_bias"), seed, 0.1, 0.f, 1, c->NV);
            w->gnorm = synth_new(NM("linear_attn.norm.weight"), seed, 0.1, 1.f, 1, c->Vd);
            w->out_proj = synth_new(NM("linear_attn.out_proj.weight"), seed, 1.0 / sqrt((double)VD), 0.f, H, VD);
            w->conv_state = (float*)calloc((size_t)CD * (c->conv_k - 1), 4);
            w->state = (float*)calloc((size_t)c->NV * c->Kd * c->Vd, 4);
        }
        w->gate = synth_new(NM("mlp.gate_proj.weight"), seed, sH, 0.f, I, H);
        w->up = synth_new(NM("mlp.up_proj.weight"), seed, sH, 0.f, I, H);
        w->down = synth_new(NM("mlp.down_proj.weight"), seed, 1.0 / sqrt((double)I), 0.f, H, I);
        w->ln1 = synth_new(NM("input_layernorm.weight"), seed, 0.1, 0.f, 1, H);
        w->ln2 = synth_new(NM("post_attention_layernorm.weight"), seed, 0.1, 0.f, 1, H);
#undef NM
    }
    /* MRotaryEmbedding::new (modeling.rs:106-124): inv_freq = 1 / base^(2i / rot_dim) with base and exponent in F32 */
    const int half = c->rot_dim / 2;
    m->cos = (float*)xmalloc((size_t)c->max_seq * half * 4);
    m->sin = (float*)xmalloc((size_t)c->max_seq * half * 4);
    for (int p = 0; p < c->max_seq; ++p)
        for (int i = 0; i < half; ++i) {
            const float inv = 1.0f / powf((float)c->theta, (float)i * 2.0f / (float)c->rot_dim);
            const float f = (float)p * inv;
            m->cos[(size_t)p * half + i] = cosf(f);
            m->sin[(size_t)p * half + i] = sinf(f);
        }
    int big = c->Hq * D * 2;
    if (CD > big) big = CD;
    m->x = (float*)xmalloc((size_t)H * 4); m->xn = (float*)xmalloc((size_t)H * 4);
    m->big = (float*)xmalloc((size_t)(big + 2 * c->Hkv * D) * 4);
    m->attn = (float*)xmalloc((size_t)((c->Hq * D > VD ? c->Hq * D : VD)) * 4);
    m->g = (float*)xmalloc((size_t)I * 4); m->u = (float*)xmalloc((size_t)I * 4); m->h = (float*)xmalloc((size_t)I * 4);
    m->mixed = (float*)xmalloc((size_t)CD * 4); m->z = (float*)xmalloc((size_t)VD * 4); m->ba = (float*)xmalloc((size_t)2 * c->NV * 4);
    return m;
}

void q5_destroy(q5_model* m) {
    if (!m) return;
    for (int li = 0; li < m->c.L; ++li) {
        q5_layer* w = &m->layers[li];
        free(w->q); free(w->k); free(w->v); free(w->o); free(w->qn); free(w->kn);
        free(w->in_qkv); free(w->in_z); free(w->in_b); free(w->in_a); free(w->conv_w); free(w->A_log); free(w->dt_bias);
        free(w->gnorm); free(w->out_proj); free(w->gate); free(w->up); free(w->down); free(w->ln1); free(w->ln2);
        free(w->kc); free(w->vc); free(w->conv_state); free(w->state);
    }
    if (m->lm_head != m->embed) free(m->lm_head);
    free(m->embed); free(m->norm); free(m->layers); free(m->cos); free(m->sin);
    free(m->x); free(m->xn); free(m->big); free(m->attn); free(m->g); free(m->u); free(m->h); free(m->mixed); free(m->z); free(m->ba);
    free(m);
}

/* clear_kv_cache (qwen3_5/model.rs:822-826): KV length, conv windows and recurrent states back to zero */
void q5_clear(q5_model* m) {
    const q5_cfg* c = &m->c;
    const int CD = 2 * c->NK * c->Kd + c->NV * c->Vd;
    for (int li = 0; li < c->L; ++li) {
        q5_layer* w = &m->layers[li];
        if (w->full) continue;
        memset(w->conv_state, 0, (size_t)CD * (c->conv_k - 1) * 4);
        memset(w->state, 0, (size_t)c->NV * c->Kd * c->Vd * 4);
    }
    m->len = 0;
}

/* Qwen35RmsNorm (modeling.rs:45-79): the stored weight is w, the model applies (1 + w) */
static void rms_norm_1p(const float* x, const uint16_t* w, float* out, int n, float eps) {
    float ss = 0.f;
    for (int i = 0; i < n; ++i) ss += x[i] * x[i];
    const float r = 1.0f / sqrtf(ss / (float)n + eps);
    for (int i = 0; i < n; ++i) out[i] = x[i] * r * (1.0f + bf2f(w[i]));
}

static inline float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
static inline float siluf_(float v) { return v / (1.0f + expf(-v)); }

static void full_attention(q5_model* m, q5_layer* w, int pos) {
    const q5_cfg* c = &m->c;
    const int H = c->H, D = c->D, Hq = c->Hq, Hkv = c->Hkv, n_rep = Hq / Hkv, rot = c->rot_dim, half = rot / 2;
    float* qg = m->big;                         /* [Hq][2 D]: per head q | gate (modeling.rs:428-455) */
    float* k = m->big + (size_t)Hq * D * 2;
    float* v = k + (size_t)Hkv * D;
    gemv(w->q, m->xn, qg, Hq * D * 2, H);
    gemv(w->k, m->xn, k, Hkv * D, H);
    gemv(w->v, m->xn, v, Hkv * D, H);
    const float* cs = m->cos + (size_t)pos * half;
    const float* sn = m->sin + (size_t)pos * half;
    for (int h = 0; h < Hq + Hkv; ++h) {
        float* p = h < Hq ? qg + (size_t)h * 2 * D : k + (size_t)(h - Hq) * D;
        const uint16_t* nw = h < Hq ? w->qn : w->kn;
        float ss = 0.f;
        for (int i = 0; i < D; ++i) ss += p[i] * p[i];
        const float r = 1.0f / sqrtf(ss / (float)D + c->eps);
        for (int i = 0; i < D; ++i) p[i] = p[i] * r * (1.0f + bf2f(nw[i]));
        for (int i = 0; i < half; ++i) {        /* rotate-half inside the rotary slice only (modeling.rs:263-279) */
            const float x1 = p[i], x2 = p[i + half];
            p[i] = x1 * cs[i] - x2 * sn[i];
            p[i + half] = x1 * sn[i] + x2 * cs[i];
        }
    }
    for (int g = 0; g < Hkv; ++g)
        for (int i = 0; i < D; ++i) {
            float kk = k[(size_t)g * D + i], vv = v[(size_t)g * D + i];
            if (c->kv_round == 1) { kk = bf2f(f2bf(kk)); vv = bf2f(f2bf(vv)); }
            else if (c->kv_round == 2) { kk = f16_round(kk); vv = f16_round(vv); }
            w->kc[((size_t)g * c->max_seq + pos) * D + i] = kk;
            w->vc[((size_t)g * c->max_seq + pos) * D + i] = vv;
        }
    const float scale = (float)(1.0 / sqrt((double)D));
#pragma omp parallel for schedule(static)
    for (int h = 0; h < Hq; ++h) {
        const float* qh = qg + (size_t)h * 2 * D;
        const float* gate = qh + D;
        const float* kh = w->kc + (size_t)(h / n_rep) * c->max_seq * D;
        const float* vh = w->vc + (size_t)(h / n_rep) * c->max_seq * D;
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
        for (int i = 0; i < D; ++i) m->attn[(size_t)h * D + i] = (acc[i] / l) * sigmoidf_(gate[i]);   /* (:516-522) */
    }
    gemv(w->o, m->attn, m->xn, H, Hq * D);
}

static void gated_delta_net(q5_model* m, q5_layer* w) {
    const q5_cfg* c = &m->c;
    const int H = c->H, NK = c->NK, NV = c->NV, K = c->Kd, V = c->Vd, KD = NK * K, VD = NV * V, CD = 2 * KD + VD;
    const int ker = c->conv_k, vpg = NV / NK;
    gemv(w->in_qkv, m->xn, m->mixed, CD, H);
    gemv(w->in_z, m->xn, m->z, VD, H);
    gemv(w->in_b, m->xn, m->ba, NV, H);
    gemv(w->in_a, m->xn, m->ba + NV, NV, H);
    /* depthwise causal conv (k taps over the window ending at this token, conv.rs:47-57) + SiLU; roll the window */
    for (int ch = 0; ch < CD; ++ch) {
        float* st = w->conv_state + (size_t)ch * (ker - 1);
        float o = 0.f;
        for (int j = 0; j < ker - 1; ++j) o += st[j] * bf2f(w->conv_w[(size_t)ch * ker + j]);
        o += m->mixed[ch] * bf2f(w->conv_w[(size_t)ch * ker + ker - 1]);
        for (int j = 0; j + 1 < ker - 1; ++j) st[j] = st[j + 1];
        st[ker - 2] = m->mixed[ch];
        m->mixed[ch] = siluf_(o);
    }
    const float* q = m->mixed;
    const float* k = m->mixed + KD;
    const float* v = m->mixed + 2 * KD;
    const float qscale = (float)(1.0 / sqrt((double)K));
#pragma omp parallel for schedule(static)
    for (int hv = 0; hv < NV; ++hv) {
        const int kh = hv / vpg;                               /* Interleaved (HF) order (layer.rs:194-238) */
        float qn[256], kn[256], kv[256], delta[256];
        float sq = 0.f, sk = 0.f;
        for (int i = 0; i < K; ++i) { sq += q[kh * K + i] * q[kh * K + i]; sk += k[kh * K + i] * k[kh * K + i]; }
        const float rq = 1.0f / sqrtf(sq + 1e-6f), rk = 1.0f / sqrtf(sk + 1e-6f);      /* l2_norm (backend.rs:26-56) */
        for (int i = 0; i < K; ++i) { qn[i] = q[kh * K + i] * rq * qscale; kn[i] = k[kh * K + i] * rk; }
        const float beta = sigmoidf_(m->ba[hv]);
        const float g = -expf(bf2f(w->A_log[hv])) * logf(1.0f + expf(m->ba[NV + hv] + bf2f(w->dt_bias[hv])));   /* backend.rs:197-211 */
        const float decay = expf(g);
        float* S = w->state + (size_t)hv * K * V;
        for (int j = 0; j < V; ++j) kv[j] = 0.f;
        for (int i = 0; i < K; ++i) {
            float* row = S + (size_t)i * V;
            for (int j = 0; j < V; ++j) { row[j] *= decay; kv[j] += row[j] * kn[i]; }
        }
        for (int j = 0; j < V; ++j) delta[j] = (v[hv * V + j] - kv[j]) * beta;
        float y[256];
        for (int j = 0; j < V; ++j) y[j] = 0.f;
        for (int i = 0; i < K; ++i) {
            float* row = S + (size_t)i * V;
            for (int j = 0; j < V; ++j) { row[j] += kn[i] * delta[j]; y[j] += row[j] * qn[i]; }
        }
        /* RmsNormGated (norm.rs:39-45): rms_norm(y, w) * silu(z), plain weight */
        float ss = 0.f;
        for (int j = 0; j < V; ++j) ss += y[j] * y[j];
        const float r = 1.0f / sqrtf(ss / (float)V + c->eps);
        for (int j = 0; j < V; ++j) m->attn[(size_t)hv * V + j] = y[j] * r * bf2f(w->gnorm[j]) * siluf_(m->z[(size_t)hv * V + j]);
    }
    gemv(w->out_proj, m->attn, m->xn, H, VD);
}

static void decode_one5(q5_model* m, uint32_t tok, int pos, float* logits) {
    const q5_cfg* c = &m->c;
    const int H = c->H, I = c->I;
    for (int i = 0; i < H; ++i) m->x[i] = bf2f(m->embed[(size_t)tok * H + i]);
    for (int li = 0; li < c->L; ++li) {
        q5_layer* w = &m->layers[li];
        rms_norm_1p(m->x, w->ln1, m->xn, H, c->eps);
        if (w->full) full_attention(m, w, pos);
        else gated_delta_net(m, w);
        for (int i = 0; i < H; ++i) m->x[i] += m->xn[i];
        rms_norm_1p(m->x, w->ln2, m->xn, H, c->eps);
        gemv(w->gate, m->xn, m->g, I, H);
        gemv(w->up, m->xn, m->u, I, H);
        for (int i = 0; i < I; ++i) m->h[i] = siluf_(m->g[i]) * m->u[i];
        gemv(w->down, m->h, m->xn, H, I);
        for (int i = 0; i < H; ++i) m->x[i] += m->xn[i];
    }
    if (logits) {
        rms_norm_1p(m->x, m->norm, m->xn, H, c->eps);
        gemv(m->lm_head, m->xn, logits, c->V, H);
    }
}

/* forward_step: positions processed one at a time (same math as the chunked causal prefill, prefill.rs:56-136); logits of
 * the last position only.  start_pos == 0 clears the recurrent state like Model::generate does (model.rs:853-860); a call
 * must continue exactly at the cached length (a recurrent state cannot be rewound). */
int q5_forward(q5_model* m, const uint32_t* ids, int n, int start_pos, float* logits) {
    if (!m || n <= 0 || start_pos < 0 || start_pos + n > m->c.max_seq) return -1;
    if (start_pos == 0) q5_clear(m);
    else if (start_pos != m->len) return -3;
    for (int i = 0; i < n; ++i) {
        if (ids[i] >= (uint32_t)m->c.V) return -2;
        decode_one5(m, ids[i], start_pos + i, (i == n - 1) ? logits : NULL);
    }
    m->len = start_pos + n;
    return 0;
}

/* bench set-up: the K/V rows of positions [0, ctx) of every full-attention layer get the values cm_debug_fill_kv writes on the
 * device (paged element order, tseed from the ABSOLUTE layer index); conv windows and recurrent states are zero, as on the
 * device (cm_debug_fill_kv resets them) */
void q5_fill_kv_paged(q5_model* m, int ctx, uint64_t seed, int page) {
    const q5_cfg* c = &m->c;
    q5_clear(m);
    for (int li = 0; li < c->L; ++li) {
        if (!m->layers[li].full) continue;
        for (int kv = 0; kv < 2; ++kv) {
            float* dst = kv ? m->layers[li].vc : m->layers[li].kc;
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
    }
    m->len = ctx;
}
