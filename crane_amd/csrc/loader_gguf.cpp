// GGUF checkpoint loader (dense Qwen3) + in-situ Q8_0 quantisation.
//   names / metadata : qwen3/modeling.rs:242-282 (attn_q/k/v/output, attn_q_norm/k_norm), :593-606 (ffn_gate/up/down),
//                      :678-696 (attn_norm / ffn_norm), :821-960 (general.architecture, {arch}.block_count,
//                      embedding_length, feed_forward_length, attention.head_count(_kv), attention.key_length (128),
//                      context_length (32768), attention.layer_norm_rms_epsilon (1e-6), rope.freq_base (1e6);
//                      use_qk_norm = has blk.0.attn_q_norm.weight; tied = no output.weight; vocab = token_embd rows)
//   quantised tensors stay quantised: linears (QMatMul, ops/linear.rs:18-51), the embedding table
//   (quantized_embedding, hunyuan_dense/modeling.rs:52-64) and its tied use as lm_head; norms are dequantised to f32.
// The GGUF byte stream is permuted into the kernel layouts of kernels_quant.hip on the host (no re-quantisation).
#include <cmath>
#include <cstring>

#include "gguf.h"
#include "model.h"

namespace cm {

namespace {

using cmgguf::File;
using cmgguf::TensorInfo;

float f16_to_f32(uint16_t h) {
    const uint32_t sgn = (h >> 15) & 1, ex = (h >> 10) & 0x1F, mant = h & 0x3FF;
    float f;
    if (ex == 0) f = std::ldexp((float)mant, -24);
    else if (ex == 31) f = mant ? NAN : INFINITY;
    else f = std::ldexp((float)(mant | 0x400), (int)ex - 25);
    return sgn ? -f : f;
}

std::string arch_of(const File& g) {
    const cmgguf::Value* a = g.meta("general.architecture");
    return (a && a->type == 8) ? a->s : "qwen3";
}

// host staging of one repacked matrix
struct Packed {
    int fmt = QFMT_NONE;
    std::vector<uint8_t> p0, p1, p2, p3;
};

// rows [row0, row0 + nrows) of a [*, Kfull] tensor, columns [k0, k0 + K) of them (K < 0: all; tensor parallelism cuts the
// row-parallel matrices along K -- on whole ggml blocks, so a rank's blocks are the blocks of the unsharded matrix)
void pack_rows(const TensorInfo& t, int row0, int nrows, int Kfull, Packed& out, int k0 = 0, int K = -1) {
    if (K < 0) K = Kfull;
    const size_t nb32 = (size_t)K / 32, nb256 = (size_t)K / 256;
    const size_t fb32 = (size_t)Kfull / 32, fb256 = (size_t)Kfull / 256, b32_0 = (size_t)k0 / 32, b256_0 = (size_t)k0 / 256;
    const bool b32 = t.type == cmgguf::Q8_0 || t.type == cmgguf::Q4_0 || t.type == cmgguf::Q5_0;
    const int blk = b32 ? 32 : 256;
    if (k0 % blk || K % blk) throw CmError(CM_ERR_UNSUPPORTED, "tensor-parallel cut of " + t.name + " does not fall on a quantisation block");
    if (t.type == cmgguf::Q8_0) {
        if (Kfull % 32) throw CmError(CM_ERR_UNSUPPORTED, "Q8_0 needs K % 32 == 0 (" + t.name + ")");
        out.fmt = QFMT_Q8_0;
        const size_t o0 = out.p0.size(), o1 = out.p1.size();
        out.p0.resize(o0 + (size_t)nrows * K);
        out.p1.resize(o1 + (size_t)nrows * nb32 * 2);
        for (int r = 0; r < nrows; ++r) {
            const uint8_t* src = t.data + ((size_t)(row0 + r) * fb32 + b32_0) * 34;
            for (size_t b = 0; b < nb32; ++b) {
                memcpy(&out.p1[o1 + ((size_t)r * nb32 + b) * 2], src + b * 34, 2);
                memcpy(&out.p0[o0 + (size_t)r * K + b * 32], src + b * 34 + 2, 32);
            }
        }
    } else if (t.type == cmgguf::Q4_0 || t.type == cmgguf::Q5_0) {
        // Q4_0: y = d (q - 8), Q5_0: y = d (q - 16) with q - offset in [-8, 7] / [-16, 15]: an int8 code under the block's own f16
        // scale, i.e. EXACTLY a Q8_0 block.  The codes are widened at load into the Q8_0 stream layout (no re-quantisation, the
        // same integer-dot arithmetic ggml_vec_dot_q4_0_q8_0 / _q5_0_q8_0 compute: sum (q - off) q8 times d_w d_x); the price is
        // the footprint -- 1.06 bytes per weight in HBM instead of 0.56 / 0.69.  Format coverage, not the formats' bandwidth.
        if (Kfull % 32) throw CmError(CM_ERR_UNSUPPORTED, "Q4_0 / Q5_0 need K % 32 == 0 (" + t.name + ")");
        const bool q5 = t.type == cmgguf::Q5_0;
        const size_t bs = q5 ? 22 : 18;
        out.fmt = QFMT_Q8_0;
        const size_t o0 = out.p0.size(), o1 = out.p1.size();
        out.p0.resize(o0 + (size_t)nrows * K);
        out.p1.resize(o1 + (size_t)nrows * nb32 * 2);
        for (int r = 0; r < nrows; ++r) {
            const uint8_t* src = t.data + ((size_t)(row0 + r) * fb32 + b32_0) * bs;
            for (size_t b = 0; b < nb32; ++b) {
                const uint8_t* blkp = src + b * bs;
                memcpy(&out.p1[o1 + ((size_t)r * nb32 + b) * 2], blkp, 2);
                int8_t* dst = (int8_t*)&out.p0[o0 + (size_t)r * K + b * 32];
                if (!q5) {
                    for (int j = 0; j < 16; ++j) { dst[j] = (int8_t)((blkp[2 + j] & 0xF) - 8); dst[j + 16] = (int8_t)((blkp[2 + j] >> 4) - 8); }
                } else {
                    uint32_t qh; memcpy(&qh, blkp + 2, 4);
                    for (int j = 0; j < 16; ++j) {
                        const int h0 = (int)((qh >> j) & 1u) << 4, h1 = (int)((qh >> (j + 16)) & 1u) << 4;
                        dst[j] = (int8_t)(((blkp[6 + j] & 0xF) | h0) - 16);
                        dst[j + 16] = (int8_t)(((blkp[6 + j] >> 4) | h1) - 16);
                    }
                }
            }
        }
    } else if (t.type == cmgguf::Q4_K) {
        if (Kfull % 256) throw CmError(CM_ERR_UNSUPPORTED, "Q4_K needs K % 256 == 0 (" + t.name + ")");
        out.fmt = QFMT_Q4_K;
        const size_t o0 = out.p0.size(), o1 = out.p1.size();
        out.p0.resize(o0 + (size_t)nrows * K / 2);
        out.p1.resize(o1 + (size_t)nrows * nb256 * 16);
        for (int r = 0; r < nrows; ++r) {
            const uint8_t* src = t.data + ((size_t)(row0 + r) * fb256 + b256_0) * 144;
            for (size_t b = 0; b < nb256; ++b) {
                memcpy(&out.p1[o1 + ((size_t)r * nb256 + b) * 16], src + b * 144, 16);
                memcpy(&out.p0[o0 + ((size_t)r * nb256 + b) * 128], src + b * 144 + 16, 128);
            }
        }
    } else if (t.type == cmgguf::Q6_K) {
        if (Kfull % 256) throw CmError(CM_ERR_UNSUPPORTED, "Q6_K needs K % 256 == 0 (" + t.name + ")");
        out.fmt = QFMT_Q6_K;
        const size_t o0 = out.p0.size(), o1 = out.p1.size(), o2 = out.p2.size(), o3 = out.p3.size();
        out.p0.resize(o0 + (size_t)nrows * K / 2);
        out.p1.resize(o1 + (size_t)nrows * K / 4);
        out.p2.resize(o2 + (size_t)nrows * K / 16);
        out.p3.resize(o3 + (size_t)nrows * nb256 * 2);
        for (int r = 0; r < nrows; ++r) {
            const uint8_t* src = t.data + ((size_t)(row0 + r) * fb256 + b256_0) * 210;
            for (size_t b = 0; b < nb256; ++b) {
                const size_t bi = (size_t)r * nb256 + b;
                memcpy(&out.p0[o0 + bi * 128], src + b * 210, 128);
                memcpy(&out.p1[o1 + bi * 64], src + b * 210 + 128, 64);
                memcpy(&out.p2[o2 + bi * 16], src + b * 210 + 192, 16);
                memcpy(&out.p3[o3 + bi * 2], src + b * 210 + 208, 2);
            }
        }
    } else if (t.type == cmgguf::Q3_K) {
        // Q3_K: y = d (sc_j - 32) (q - 4 | q) with 16 sub-blocks of 16, 6-bit scales and 3-bit codes in [-4, 3] -- EXACTLY a Q6_K block
        // with int8 scales sc_j - 32 in [-32, 31] and 6-bit codes q + 32: widened at load into the Q6_K stream layout (no
        // re-quantisation; ggml_vec_dot_q3_K_q8_K and _q6_K_q8_K are the same integer arithmetic: sum_j scale_j sum_16 q q8, times d d8).
        // The price is the footprint (0.82 bytes per weight instead of 0.43): format coverage, as for Q4_0 / Q5_0.
        if (Kfull % 256) throw CmError(CM_ERR_UNSUPPORTED, "Q3_K needs K % 256 == 0 (" + t.name + ")");
        out.fmt = QFMT_Q6_K;
        const size_t o0 = out.p0.size(), o1 = out.p1.size(), o2 = out.p2.size(), o3 = out.p3.size();
        out.p0.resize(o0 + (size_t)nrows * K / 2);
        out.p1.resize(o1 + (size_t)nrows * K / 4);
        out.p2.resize(o2 + (size_t)nrows * K / 16);
        out.p3.resize(o3 + (size_t)nrows * nb256 * 2);
        for (int r = 0; r < nrows; ++r) {
            const uint8_t* src = t.data + ((size_t)(row0 + r) * fb256 + b256_0) * 110;
            for (size_t b = 0; b < nb256; ++b) {
                const uint8_t* blkp = src + b * 110;
                const uint8_t *hm = blkp, *qs = blkp + 32, *sb = blkp + 96;
                const size_t bi = (size_t)r * nb256 + b;
                // the 16 six-bit scales (dequantize_row_q3_K's aux[] shuffle)
                uint32_t aux[4]; memcpy(aux, sb, 12);
                const uint32_t k1 = 0x03030303u, k2 = 0x0f0f0f0fu, tmp = aux[2];
                aux[2] = ((aux[0] >> 4) & k2) | (((tmp >> 4) & k1) << 4);
                aux[3] = ((aux[1] >> 4) & k2) | (((tmp >> 6) & k1) << 4);
                aux[0] = (aux[0] & k2) | (((tmp >> 0) & k1) << 4);
                aux[1] = (aux[1] & k2) | (((tmp >> 2) & k1) << 4);
                const uint8_t* s6 = (const uint8_t*)aux;
                for (int j = 0; j < 16; ++j) out.p2[o2 + bi * 16 + j] = (uint8_t)(int8_t)((int)s6[j] - 32);
                memcpy(&out.p3[o3 + bi * 2], blkp + 108, 2);
                // codes in weight order, biased by 32 (Q6_K stores q + 32 in 0 .. 63)
                uint8_t c[256];
                for (int n = 0; n < 2; ++n)
                    for (int j = 0; j < 4; ++j)
                        for (int l = 0; l < 32; ++l) {
                            const int lo = (qs[32 * n + l] >> (2 * j)) & 3, hb = (hm[l] >> (4 * n + j)) & 1;
                            c[128 * n + 32 * j + l] = (uint8_t)(lo - (hb ? 0 : 4) + 32);
                        }
                uint8_t* ql = &out.p0[o0 + bi * 128];
                uint8_t* qh = &out.p1[o1 + bi * 64];
                for (int h = 0; h < 2; ++h)
                    for (int l = 0; l < 32; ++l) {
                        const uint8_t a1 = c[128 * h + l], a2 = c[128 * h + 32 + l], a3 = c[128 * h + 64 + l], a4 = c[128 * h + 96 + l];
                        ql[64 * h + l] = (uint8_t)((a1 & 0xF) | ((a3 & 0xF) << 4));
                        ql[64 * h + 32 + l] = (uint8_t)((a2 & 0xF) | ((a4 & 0xF) << 4));
                        qh[32 * h + l] = (uint8_t)((a1 >> 4) | ((a2 >> 4) << 2) | ((a3 >> 4) << 4) | ((a4 >> 4) << 6));
                    }
            }
        }
    } else {
        throw CmError(CM_ERR_UNSUPPORTED, "GGUF tensor " + t.name + ": ggml type " + std::to_string(t.type) +
                                              " is not supported for matrices (Q4_0, Q5_0, Q8_0, Q3_K, Q4_K, Q6_K)");
    }
}

const uint8_t* upload(Model& m, const std::vector<uint8_t>& v) {
    if (v.empty()) return nullptr;
    uint8_t* d = (uint8_t*)m.dalloc<uint16_t>((v.size() + 1) / 2, true);
    CM_HIP(hipMemcpy(d, v.data(), v.size(), hipMemcpyHostToDevice));
    return d;
}

QWeight to_device(Model& m, const Packed& p, int N, int K) {
    QWeight w;
    w.fmt = p.fmt; w.N = N; w.K = K;
    w.p0 = upload(m, p.p0); w.p1 = upload(m, p.p1); w.p2 = upload(m, p.p2); w.p3 = upload(m, p.p3);
    m.quant_weight_bytes += w.bytes();
    return w;
}

void check_shape(const TensorInfo& t, uint64_t rows, uint64_t cols) {
    if (t.shape.size() != 2 || t.shape[0] != rows || t.shape[1] != cols)
        throw CmError(CM_ERR_IO, "GGUF tensor " + t.name + " has unexpected shape");
}

QWeight load_matrix(Model& m, const File& g, const std::string& name, int N, int K) {
    const TensorInfo& t = g.tensor(name);
    check_shape(t, (uint64_t)N, (uint64_t)K);
    Packed p;
    pack_rows(t, 0, N, K, p);
    return to_device(m, p, N, K);
}

// rows [row0, row0 + nrows) x columns [k0, k0 + Kl) of the [N, K] tensor `name`
QWeight load_matrix_part(Model& m, const File& g, const std::string& name, int N, int K, int row0, int nrows, int k0, int Kl) {
    const TensorInfo& t = g.tensor(name);
    check_shape(t, (uint64_t)N, (uint64_t)K);
    Packed p;
    pack_rows(t, row0, nrows, K, p, k0, Kl);
    return to_device(m, p, nrows, Kl);
}

float* load_vec_f32(Model& m, const File& g, const std::string& name, int n) {
    const TensorInfo& t = g.tensor(name);
    if (t.numel() != (uint64_t)n) throw CmError(CM_ERR_IO, "GGUF tensor " + name + " has unexpected length");
    std::vector<float> h((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (t.type == cmgguf::F32) memcpy(&h[(size_t)i], t.data + (size_t)i * 4, 4);
        else if (t.type == cmgguf::F16) { uint16_t v; memcpy(&v, t.data + (size_t)i * 2, 2); h[(size_t)i] = f16_to_f32(v); }
        else if (t.type == cmgguf::BF16) { uint16_t v; memcpy(&v, t.data + (size_t)i * 2, 2); const uint32_t u = (uint32_t)v << 16; memcpy(&h[(size_t)i], &u, 4); }
        else throw CmError(CM_ERR_UNSUPPORTED, "GGUF tensor " + name + ": vectors must be F32 / F16 / BF16");
    }
    float* d = m.dalloc<float>((size_t)n, true);
    CM_HIP(hipMemcpy(d, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    return d;
}

}  // namespace

std::string gguf_config_json(const std::string& path) {
    File g(path);
    const std::string a = arch_of(g);
    const bool hybrid = a == "qwen35" || a == "qwen3_5" || a == "qwen35moe";
    if (a != "qwen3" && !hybrid) throw CmError(CM_ERR_UNSUPPORTED, "GGUF architecture '" + a + "' not implemented (qwen3, qwen35)");
    auto need_u = [&](const std::string& k) -> uint64_t {
        const cmgguf::Value* v = g.meta(a + "." + k);
        if (!v) throw CmError(CM_ERR_IO, "cannot find " + a + "." + k + " in GGUF metadata");
        return v->as_u64();
    };
    auto opt_u = [&](const std::string& k, uint64_t def) { const cmgguf::Value* v = g.meta(a + "." + k); return v ? v->as_u64() : def; };
    auto opt_f = [&](const std::string& k, double def) { const cmgguf::Value* v = g.meta(a + "." + k); return v ? v->as_f64() : def; };
    const TensorInfo& emb = g.tensor("token_embd.weight");
    if (emb.shape.size() != 2) throw CmError(CM_ERR_IO, "token_embd.weight must be a matrix");
    char buf[2048];
    if (!hybrid) {
        snprintf(buf, sizeof buf,
                 "{\"model_type\":\"qwen3\",\"hidden_size\":%llu,\"num_hidden_layers\":%llu,\"num_attention_heads\":%llu,"
                 "\"num_key_value_heads\":%llu,\"head_dim\":%llu,\"intermediate_size\":%llu,\"vocab_size\":%llu,"
                 "\"max_position_embeddings\":%llu,\"rms_norm_eps\":%.9g,\"rope_theta\":%.9g,\"tie_word_embeddings\":%s,\"use_qk_norm\":%s}",
                 (unsigned long long)need_u("embedding_length"), (unsigned long long)need_u("block_count"),
                 (unsigned long long)need_u("attention.head_count"), (unsigned long long)need_u("attention.head_count_kv"),
                 (unsigned long long)opt_u("attention.key_length", 128), (unsigned long long)need_u("feed_forward_length"),
                 (unsigned long long)emb.shape[0], (unsigned long long)opt_u("context_length", 32768),
                 opt_f("attention.layer_norm_rms_epsilon", 1e-6), opt_f("rope.freq_base", 1e6),
                 g.has("output.weight") ? "false" : "true", g.has("blk.0.attn_q_norm.weight") ? "true" : "false");
        return buf;
    }
    // Qwen 3.5 family (qwen3_5/model.rs:155-287)
    const uint64_t head_dim = need_u("attention.key_length"), L = need_u("block_count"), Hq = need_u("attention.head_count");
    const uint64_t nv = need_u("ssm.time_step_rank"), inner = need_u("ssm.inner_size");
    const uint64_t rot = opt_u("rope.dimension_count", head_dim / 4);
    uint64_t interval = opt_u("full_attention_interval", 4);
    // layer kinds come from tensor presence (model.rs:223-232); they must follow the interval rule this runtime uses
    int first_full = -1;
    for (uint64_t i = 0; i < L; ++i)
        if (!g.has("blk." + std::to_string(i) + ".ssm_a")) { first_full = (int)i; break; }
    if (first_full >= 0 && !g.meta(a + ".full_attention_interval")) interval = (uint64_t)first_full + 1;
    for (uint64_t i = 0; i < L; ++i) {
        const bool full = !g.has("blk." + std::to_string(i) + ".ssm_a");
        if (full != (((i + 1) % interval) == 0)) throw CmError(CM_ERR_UNSUPPORTED, "GGUF layer kinds do not follow full_attention_interval");
    }
    bool gate = true;
    if (first_full >= 0) {
        const TensorInfo& q = g.tensor("blk." + std::to_string(first_full) + ".attn_q.weight");
        gate = q.shape.size() == 2 && q.shape[0] == 2 * Hq * head_dim;            // model.rs:235-248
    }
    std::string sec = "[";
    if (const cmgguf::Value* v = g.meta(a + ".rope.dimension_sections"))
        for (size_t i = 0; i < v->arr.size() && i < 3; ++i) sec += (i ? "," : "") + std::to_string((long long)std::max<int64_t>(0, (int64_t)v->arr[i].as_f64()));
    sec += "]";
    snprintf(buf, sizeof buf,
             "{\"model_type\":\"qwen3_5_text\",\"hidden_size\":%llu,\"num_hidden_layers\":%llu,\"num_attention_heads\":%llu,"
             "\"num_key_value_heads\":%llu,\"head_dim\":%llu,\"intermediate_size\":%llu,\"vocab_size\":%llu,"
             "\"max_position_embeddings\":%llu,\"rms_norm_eps\":%.9g,\"tie_word_embeddings\":%s,\"full_attention_interval\":%llu,"
             "\"linear_conv_kernel_dim\":%llu,\"linear_key_head_dim\":%llu,\"linear_value_head_dim\":%llu,\"linear_num_key_heads\":%llu,"
             "\"linear_num_value_heads\":%llu,\"attn_output_gate\":%s,"
             "\"rope_parameters\":{\"rope_theta\":%.9g,\"partial_rotary_factor\":%.9g,\"mrope_section\":%s}}",
             (unsigned long long)need_u("embedding_length"), (unsigned long long)L, (unsigned long long)Hq,
             (unsigned long long)need_u("attention.head_count_kv"), (unsigned long long)head_dim,
             (unsigned long long)need_u("feed_forward_length"), (unsigned long long)emb.shape[0],
             (unsigned long long)opt_u("context_length", 262144), opt_f("attention.layer_norm_rms_epsilon", 1e-6),
             g.has("output.weight") ? "false" : "true", (unsigned long long)interval,
             (unsigned long long)need_u("ssm.conv_kernel"), (unsigned long long)need_u("ssm.state_size"),
             (unsigned long long)(inner / std::max<uint64_t>(1, nv)), (unsigned long long)need_u("ssm.group_count"),
             (unsigned long long)nv, gate ? "true" : "false", opt_f("rope.freq_base", 1e7),
             (double)rot / (double)head_dim, sec.c_str());
    return buf;
}

// Dense Qwen3 under tensor parallelism: the Megatron cut of loader.cpp on the quantised tensors -- q heads, the kv heads of
// this rank (replicated when there are fewer kv heads than ranks) and the gate / up rows by ROW; o_proj and down_proj by
// COLUMN, i.e. along K, on whole ggml blocks (32 for Q8_0, 256 for the K-quants: a rank's blocks are the file's blocks).
static void load_dense_shard(Model& m, const File& g) {
    const Config& c = m.cfg;
    const int H = c.H, D = c.D, I = c.I, qd = m.Hq_l * D, kd = m.Hkv_l * D, Il = m.I_l, r = m.rank;
    for (int li = 0; li < c.L; ++li) {
        LayerW& w = m.layers[(size_t)li];
        w.full = true;
        const std::string p = "blk." + std::to_string(li) + ".";
        const TensorInfo &tq = g.tensor(p + "attn_q.weight"), &tk = g.tensor(p + "attn_k.weight"), &tv = g.tensor(p + "attn_v.weight");
        check_shape(tq, (uint64_t)c.Hq * D, (uint64_t)H); check_shape(tk, (uint64_t)c.Hkv * D, (uint64_t)H); check_shape(tv, (uint64_t)c.Hkv * D, (uint64_t)H);
        if (tq.type == tk.type && tk.type == tv.type) {
            Packed pk;
            pack_rows(tq, r * qd, qd, H, pk); pack_rows(tk, m.kvh0 * D, kd, H, pk); pack_rows(tv, m.kvh0 * D, kd, H, pk);
            w.q_qkv[0] = to_device(m, pk, qd + 2 * kd, H);
            w.n_qkv = 1; w.qkv_row0[0] = 0;
        } else {
            w.q_qkv[0] = load_matrix_part(m, g, p + "attn_q.weight", c.Hq * D, H, r * qd, qd, 0, H);
            w.q_qkv[1] = load_matrix_part(m, g, p + "attn_k.weight", c.Hkv * D, H, m.kvh0 * D, kd, 0, H);
            w.q_qkv[2] = load_matrix_part(m, g, p + "attn_v.weight", c.Hkv * D, H, m.kvh0 * D, kd, 0, H);
            w.n_qkv = 3; w.qkv_row0[0] = 0; w.qkv_row0[1] = qd; w.qkv_row0[2] = qd + kd;
        }
        w.q_o = load_matrix_part(m, g, p + "attn_output.weight", H, c.Hq * D, 0, H, r * qd, qd);
        if (c.qk_norm) {
            w.qn = load_vec_f32(m, g, p + "attn_q_norm.weight", D);
            w.kn = load_vec_f32(m, g, p + "attn_k_norm.weight", D);
        }
        const TensorInfo &tg = g.tensor(p + "ffn_gate.weight"), &tu = g.tensor(p + "ffn_up.weight");
        check_shape(tg, (uint64_t)I, (uint64_t)H); check_shape(tu, (uint64_t)I, (uint64_t)H);
        if (tg.type == tu.type) {
            Packed pk;                                              // row 2j = gate_j, row 2j+1 = up_j of this rank's columns
            for (int j = 0; j < Il; ++j) { pack_rows(tg, r * Il + j, 1, H, pk); pack_rows(tu, r * Il + j, 1, H, pk); }
            w.q_gate_up = to_device(m, pk, 2 * Il, H);
        } else {
            w.split_gate_up = true;
            w.q_gate = load_matrix_part(m, g, p + "ffn_gate.weight", I, H, r * Il, Il, 0, H);
            w.q_up = load_matrix_part(m, g, p + "ffn_up.weight", I, H, r * Il, Il, 0, H);
            if (!m.gu_tmp) m.gu_tmp = m.dalloc<float>((size_t)2 * Il);
        }
        w.q_down = load_matrix_part(m, g, p + "ffn_down.weight", H, I, 0, H, r * Il, Il);
        w.ln1 = load_vec_f32(m, g, p + "attn_norm.weight", H);
        w.ln2 = load_vec_f32(m, g, p + "ffn_norm.weight", H);
    }
}

// one f32 / f16 / bf16 vector of the file as floats on the host
static std::vector<float> host_vec(const File& g, const std::string& name, size_t n) {
    const TensorInfo& t = g.tensor(name);
    if (t.numel() != (uint64_t)n) throw CmError(CM_ERR_IO, "GGUF tensor " + name + " has unexpected length");
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) {
        if (t.type == cmgguf::F32) memcpy(&h[i], t.data + i * 4, 4);
        else if (t.type == cmgguf::F16) { uint16_t v; memcpy(&v, t.data + i * 2, 2); h[i] = f16_to_f32(v); }
        else if (t.type == cmgguf::BF16) { uint16_t v; memcpy(&v, t.data + i * 2, 2); const uint32_t u = (uint32_t)v << 16; memcpy(&h[i], &u, 4); }
        else throw CmError(CM_ERR_UNSUPPORTED, "GGUF tensor " + name + ": vectors must be F32 / F16 / BF16");
    }
    return h;
}
static float* upload_f32(Model& m, const std::vector<float>& h) {
    float* d = m.dalloc<float>(h.size(), true);
    CM_HIP(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    return d;
}

// The hybrid family (llama.cpp `qwen35`) under tensor parallelism (round 4).  Rank r owns key heads [r NK_l, (r + 1) NK_l) of the
// Gated Delta Net layers and the value heads paired with them.  The file orders the value heads CHUNKED (key head of value head v
// = v mod NK, ops/gdn/config.rs:13-22), so a rank's value heads are NOT contiguous: local value head j = c * NK_l + kl is the
// file's c * NK + r * NK_l + kl -- the rank's own heads again in chunked order, which is what the kernels' `chunked` flag
// expects.  Rows (q / k / v conv channels, z, the bf16 b / a rows, A_log, dt_bias) are gathered by head; ssm_out is cut by COLUMN
// in runs of NK_l value heads (whole ggml blocks: any NK_l for the 32-weight formats, an even NK_l for the K-quants -- two heads =
// one 256-block; Qwen3.8-27B at TP = 8 has NK_l = 2).  Full-attention layers: this rank's q heads (query and gate rows of the
// per-head [q | gate] layout), its kv head(s), attn_output by column; the MLP as for the dense family.
static void load_hybrid_shard(Model& m, const File& g) {
    const Config& c = m.cfg;
    const int H = c.H, D = c.D, I = c.I, Il = m.I_l, r = m.rank, qd = m.Hq_l * D, kd = m.Hkv_l * D;
    const int NKl = c.NK, NVl = c.NV, NKg = c.NK_g, NVg = c.NV_g, Kd = c.Kd, Vd = c.Vd, vpg = NVl / NKl;
    const int KDg = NKg * Kd, VDg = NVg * Vd, CDg = 2 * KDg + VDg, kdl = NKl * Kd, vdl = NVl * Vd, cdl = 2 * kdl + vdl;
    auto gv = [&](int j) { return (j / NKl) * NKg + r * NKl + (j % NKl); };          // file index of local value head j
    m.gdn_chunked = true;
    int gdn_idx = 0;
    for (int li = 0; li < c.L; ++li) {
        LayerW& w = m.layers[(size_t)li];
        w.full = c.layer_full(li);
        const std::string p = "blk." + std::to_string(li) + ".";
        if (!w.full) {
            w.gdn_idx = gdn_idx++;
            const TensorInfo &tqkv = g.tensor(p + "attn_qkv.weight"), &tz = g.tensor(p + "attn_gate.weight");
            check_shape(tqkv, (uint64_t)CDg, (uint64_t)H); check_shape(tz, (uint64_t)VDg, (uint64_t)H);
            auto pack_qkv = [&](Packed& pk) {
                pack_rows(tqkv, r * kdl, kdl, H, pk);
                pack_rows(tqkv, KDg + r * kdl, kdl, H, pk);
                for (int j = 0; j < NVl; ++j) pack_rows(tqkv, 2 * KDg + gv(j) * Vd, Vd, H, pk);
            };
            auto pack_z = [&](Packed& pk) { for (int j = 0; j < NVl; ++j) pack_rows(tz, gv(j) * Vd, Vd, H, pk); };
            if (tqkv.type == tz.type) {
                Packed pk;
                pack_qkv(pk); pack_z(pk);
                w.q_in_proj = to_device(m, pk, cdl + vdl, H);
            } else {
                Packed pa, pb;
                pack_qkv(pa); pack_z(pb);
                w.q_in_proj = to_device(m, pa, cdl, H);
                w.q_in_proj_z = to_device(m, pb, vdl, H);
            }
            std::vector<uint16_t> ba((size_t)2 * NVl * H);               // b rows then a rows of this rank's value heads, bf16
            for (int which = 0; which < 2; ++which) {
                const std::vector<float> full = host_vec(g, p + (which == 0 ? "ssm_beta.weight" : "ssm_alpha.weight"), (size_t)NVg * H);
                for (int j = 0; j < NVl; ++j)
                    for (int k = 0; k < H; ++k) {
                        uint32_t u; memcpy(&u, &full[(size_t)gv(j) * H + k], 4);
                        ba[((size_t)which * NVl + j) * H + k] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
                    }
            }
            w.in_proj_ba = m.dalloc<uint16_t>(ba.size(), true);
            CM_HIP(hipMemcpy(w.in_proj_ba, ba.data(), ba.size() * 2, hipMemcpyHostToDevice));
            m.quant_weight_bytes += ba.size() * 2;
            {   // ssm_out [H, VDg]: per row the vpg runs of NK_l heads
                const TensorInfo& to = g.tensor(p + "ssm_out.weight");
                check_shape(to, (uint64_t)H, (uint64_t)VDg);
                Packed po;
                for (int row = 0; row < H; ++row)
                    for (int cc = 0; cc < vpg; ++cc) pack_rows(to, row, 1, VDg, po, (cc * NKg + r * NKl) * Vd, NKl * Vd);
                w.q_out_proj = to_device(m, po, H, vdl);
            }
            {   // conv taps [conv_dim, k]: q channels, k channels, v channels of the rank's heads
                const std::vector<float> full = host_vec(g, p + "ssm_conv1d.weight", (size_t)CDg * c.conv_k);
                std::vector<float> loc((size_t)cdl * c.conv_k);
                auto put = [&](int dst_ch, int src_ch, int n) { memcpy(&loc[(size_t)dst_ch * c.conv_k], &full[(size_t)src_ch * c.conv_k], (size_t)n * c.conv_k * 4); };
                put(0, r * kdl, kdl);
                put(kdl, KDg + r * kdl, kdl);
                for (int j = 0; j < NVl; ++j) put(2 * kdl + j * Vd, 2 * KDg + gv(j) * Vd, Vd);
                w.conv_w = upload_f32(m, loc);
            }
            {
                const TensorInfo& ta = g.tensor(p + "ssm_a");
                if (ta.numel() != (uint64_t)NVg || ta.type != cmgguf::F32) throw CmError(CM_ERR_IO, "ssm_a must be F32 [num_v_heads]");
                const std::vector<float> sa = host_vec(g, p + "ssm_a", (size_t)NVg), dt = host_vec(g, p + "ssm_dt.bias", (size_t)NVg);
                std::vector<float> al((size_t)NVl), db((size_t)NVl);
                for (int j = 0; j < NVl; ++j) { al[(size_t)j] = logf(-sa[(size_t)gv(j)]); db[(size_t)j] = dt[(size_t)gv(j)]; }
                w.A_log = upload_f32(m, al);
                w.dt_bias = upload_f32(m, db);
            }
            w.gnorm = load_vec_f32(m, g, p + "ssm_norm.weight", Vd);
        } else {
            const TensorInfo &tq = g.tensor(p + "attn_q.weight"), &tk = g.tensor(p + "attn_k.weight"), &tv = g.tensor(p + "attn_v.weight");
            check_shape(tq, (uint64_t)2 * c.Hq * D, (uint64_t)H); check_shape(tk, (uint64_t)c.Hkv * D, (uint64_t)H); check_shape(tv, (uint64_t)c.Hkv * D, (uint64_t)H);
            auto pack_q = [&](Packed& pk) {          // [all q | all gate] of this rank's heads out of the per-head [q | gate] rows
                for (int h = 0; h < m.Hq_l; ++h) pack_rows(tq, (r * m.Hq_l + h) * 2 * D, D, H, pk);
                for (int h = 0; h < m.Hq_l; ++h) pack_rows(tq, (r * m.Hq_l + h) * 2 * D + D, D, H, pk);
            };
            if (tq.type == tk.type && tk.type == tv.type) {
                Packed pk;
                pack_q(pk); pack_rows(tk, m.kvh0 * D, kd, H, pk); pack_rows(tv, m.kvh0 * D, kd, H, pk);
                w.q_qkv[0] = to_device(m, pk, 2 * qd + 2 * kd, H);
                w.n_qkv = 1; w.qkv_row0[0] = 0;
            } else {
                Packed pq;
                pack_q(pq);
                w.q_qkv[0] = to_device(m, pq, 2 * qd, H);
                w.q_qkv[1] = load_matrix_part(m, g, p + "attn_k.weight", c.Hkv * D, H, m.kvh0 * D, kd, 0, H);
                w.q_qkv[2] = load_matrix_part(m, g, p + "attn_v.weight", c.Hkv * D, H, m.kvh0 * D, kd, 0, H);
                w.n_qkv = 3; w.qkv_row0[0] = 0; w.qkv_row0[1] = 2 * qd; w.qkv_row0[2] = 2 * qd + kd;
            }
            w.q_o = load_matrix_part(m, g, p + "attn_output.weight", H, c.Hq * D, 0, H, r * qd, qd);
            if (c.qk_norm) {
                w.qn = load_vec_f32(m, g, p + "attn_q_norm.weight", D);
                w.kn = load_vec_f32(m, g, p + "attn_k_norm.weight", D);
            }
        }
        const TensorInfo &tg = g.tensor(p + "ffn_gate.weight"), &tu = g.tensor(p + "ffn_up.weight");
        check_shape(tg, (uint64_t)I, (uint64_t)H); check_shape(tu, (uint64_t)I, (uint64_t)H);
        if (tg.type == tu.type) {
            Packed pk;
            for (int j = 0; j < Il; ++j) { pack_rows(tg, r * Il + j, 1, H, pk); pack_rows(tu, r * Il + j, 1, H, pk); }
            w.q_gate_up = to_device(m, pk, 2 * Il, H);
        } else {
            w.split_gate_up = true;
            w.q_gate = load_matrix_part(m, g, p + "ffn_gate.weight", I, H, r * Il, Il, 0, H);
            w.q_up = load_matrix_part(m, g, p + "ffn_up.weight", I, H, r * Il, Il, 0, H);
            if (!m.gu_tmp) m.gu_tmp = m.dalloc<float>((size_t)2 * Il);
        }
        w.q_down = load_matrix_part(m, g, p + "ffn_down.weight", H, I, 0, H, r * Il, Il);
        w.ln1 = load_vec_f32(m, g, p + "attn_norm.weight", H);
        w.ln2 = load_vec_f32(m, g, p + "post_attention_norm.weight", H);
    }
}

void load_from_gguf(Model& m, const std::string& path) {
    try {
        File g(path);
        const Config& c = m.cfg;
        const int H = c.H, D = c.D, I = c.I;
        m.quantized = true;
        m.q_embed = load_matrix(m, g, "token_embd.weight", c.V, H);          // replicated: every rank embeds the token itself
        m.quant_weight_bytes -= m.q_embed.bytes();                 // only one row is read per token
        m.norm = load_vec_f32(m, g, "output_norm.weight", H);
        // lm_head: this rank's vocabulary rows [v0, v0 + v_eff) (tensor parallelism; everything at tp = 1)
        const int v_eff = std::max(0, std::min(m.V_l, c.V - m.v0));
        if (!c.tie) { if (v_eff > 0) m.q_lm_head = load_matrix_part(m, g, "output.weight", c.V, H, m.v0, v_eff, 0, H); }
        else { m.q_lm_head = m.q_embed.rows(m.v0, v_eff); m.quant_weight_bytes += m.q_lm_head.bytes(); }
        m.layers.resize((size_t)c.L);
        if (m.tp != 1) {
            if (c.hybrid) load_hybrid_shard(m, g); else load_dense_shard(m, g);
            CM_HIP(hipStreamSynchronize(m.stream));
            return;
        }
        const int qd = c.Hq * D, kd = c.Hkv * D;
        m.gdn_chunked = c.hybrid;                  // llama.cpp orders the GDN value heads Chunked (ops/gdn/config.rs:13-22)
        int gdn_idx = 0;
        for (int li = 0; li < c.L; ++li) {
            LayerW& w = m.layers[(size_t)li];
            w.full = c.layer_full(li);
            const std::string p = "blk." + std::to_string(li) + ".";
            if (!w.full) {
                // Gated Delta Net block (qwen3_5/modeling.rs:725-765): attn_qkv = in_proj_qkv, attn_gate = z, ssm_beta / ssm_alpha =
                // b / a (dequantised, never quantised), ssm_conv1d [conv_dim, k], ssm_a = -exp(A_log), ssm_dt.bias, ssm_norm, ssm_out
                w.gdn_idx = gdn_idx++;
                const int cd = c.conv_dim(), vd = c.value_dim(), nv = c.NV;
                const TensorInfo &tqkv = g.tensor(p + "attn_qkv.weight"), &tz = g.tensor(p + "attn_gate.weight");
                check_shape(tqkv, (uint64_t)cd, (uint64_t)H); check_shape(tz, (uint64_t)vd, (uint64_t)H);
                if (tqkv.type == tz.type) {
                    Packed pk;
                    pack_rows(tqkv, 0, cd, H, pk); pack_rows(tz, 0, vd, H, pk);
                    w.q_in_proj = to_device(m, pk, cd + vd, H);
                } else {
                    w.q_in_proj = load_matrix(m, g, p + "attn_qkv.weight", cd, H);
                    w.q_in_proj_z = load_matrix(m, g, p + "attn_gate.weight", vd, H);
                }
                // b rows then a rows, kept in the model dtype (bf16)
                std::vector<uint16_t> ba((size_t)2 * nv * H);
                for (int which = 0; which < 2; ++which) {
                    const TensorInfo& t = g.tensor(p + (which == 0 ? "ssm_beta.weight" : "ssm_alpha.weight"));
                    check_shape(t, (uint64_t)nv, (uint64_t)H);
                    for (size_t i = 0; i < (size_t)nv * H; ++i) {
                        float f;
                        if (t.type == cmgguf::F32) memcpy(&f, t.data + i * 4, 4);
                        else if (t.type == cmgguf::F16) { uint16_t v; memcpy(&v, t.data + i * 2, 2); f = f16_to_f32(v); }
                        else if (t.type == cmgguf::BF16) { uint16_t v; memcpy(&v, t.data + i * 2, 2); const uint32_t u = (uint32_t)v << 16; memcpy(&f, &u, 4); }
                        else throw CmError(CM_ERR_UNSUPPORTED, "ssm_alpha / ssm_beta must be F32 / F16 / BF16");
                        uint32_t u; memcpy(&u, &f, 4);
                        ba[(size_t)which * nv * H + i] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
                    }
                }
                w.in_proj_ba = m.dalloc<uint16_t>(ba.size(), true);
                CM_HIP(hipMemcpy(w.in_proj_ba, ba.data(), ba.size() * 2, hipMemcpyHostToDevice));
                m.quant_weight_bytes += ba.size() * 2;
                w.q_out_proj = load_matrix(m, g, p + "ssm_out.weight", H, vd);
                w.conv_w = load_vec_f32(m, g, p + "ssm_conv1d.weight", cd * c.conv_k);
                {   // A_log = log(-ssm_a)
                    const TensorInfo& t = g.tensor(p + "ssm_a");
                    if (t.numel() != (uint64_t)nv || t.type != cmgguf::F32) throw CmError(CM_ERR_IO, "ssm_a must be F32 [num_v_heads]");
                    std::vector<float> al((size_t)nv);
                    for (int i = 0; i < nv; ++i) { float v; memcpy(&v, t.data + (size_t)i * 4, 4); al[(size_t)i] = logf(-v); }
                    w.A_log = m.dalloc<float>((size_t)nv, true);
                    CM_HIP(hipMemcpy(w.A_log, al.data(), (size_t)nv * 4, hipMemcpyHostToDevice));
                }
                w.dt_bias = load_vec_f32(m, g, p + "ssm_dt.bias", nv);
                w.gnorm = load_vec_f32(m, g, p + "ssm_norm.weight", c.Vd);
            } else {
            const TensorInfo &tq = g.tensor(p + "attn_q.weight"), &tk = g.tensor(p + "attn_k.weight"), &tv = g.tensor(p + "attn_v.weight");
            const int qrows = c.hybrid ? 2 * qd : qd;
            check_shape(tq, (uint64_t)qrows, (uint64_t)H); check_shape(tk, (uint64_t)kd, (uint64_t)H); check_shape(tv, (uint64_t)kd, (uint64_t)H);
            // hybrid: attn_q keeps HF's per-head [query | gate] rows (modeling.rs:376-378); HBM layout is [all q | all gate | k | v]
            auto pack_q = [&](Packed& pk) {
                if (!c.hybrid) { pack_rows(tq, 0, qd, H, pk); return; }
                for (int h = 0; h < c.Hq; ++h) pack_rows(tq, h * 2 * D, D, H, pk);
                for (int h = 0; h < c.Hq; ++h) pack_rows(tq, h * 2 * D + D, D, H, pk);
            };
            if (tq.type == tk.type && tk.type == tv.type) {
                Packed pk;
                pack_q(pk); pack_rows(tk, 0, kd, H, pk); pack_rows(tv, 0, kd, H, pk);
                w.q_qkv[0] = to_device(m, pk, qrows + 2 * kd, H);
                w.n_qkv = 1; w.qkv_row0[0] = 0;
            } else {
                Packed pq;
                pack_q(pq);
                w.q_qkv[0] = to_device(m, pq, qrows, H);
                w.q_qkv[1] = load_matrix(m, g, p + "attn_k.weight", kd, H);
                w.q_qkv[2] = load_matrix(m, g, p + "attn_v.weight", kd, H);
                w.n_qkv = 3; w.qkv_row0[0] = 0; w.qkv_row0[1] = qrows; w.qkv_row0[2] = qrows + kd;
            }
            w.q_o = load_matrix(m, g, p + "attn_output.weight", H, qd);
            if (c.qk_norm) {
                w.qn = load_vec_f32(m, g, p + "attn_q_norm.weight", D);         // hybrid: stored with the +1 already folded
                w.kn = load_vec_f32(m, g, p + "attn_k_norm.weight", D);
            }
            }
            const TensorInfo &tg = g.tensor(p + "ffn_gate.weight"), &tu = g.tensor(p + "ffn_up.weight");
            check_shape(tg, (uint64_t)I, (uint64_t)H); check_shape(tu, (uint64_t)I, (uint64_t)H);
            if (tg.type == tu.type) {
                Packed pk;                                              // row 2j = gate_j, row 2j+1 = up_j
                for (int j = 0; j < I; ++j) { pack_rows(tg, j, 1, H, pk); pack_rows(tu, j, 1, H, pk); }
                w.q_gate_up = to_device(m, pk, 2 * I, H);
            } else {
                w.split_gate_up = true;
                w.q_gate = load_matrix(m, g, p + "ffn_gate.weight", I, H);
                w.q_up = load_matrix(m, g, p + "ffn_up.weight", I, H);
                if (!m.gu_tmp) m.gu_tmp = m.dalloc<float>((size_t)2 * I);
            }
            w.q_down = load_matrix(m, g, p + "ffn_down.weight", H, I);
            w.ln1 = load_vec_f32(m, g, p + "attn_norm.weight", H);             // hybrid: +1 pre-folded (modeling.rs:693-695)
            w.ln2 = load_vec_f32(m, g, p + (c.hybrid ? "post_attention_norm.weight" : "ffn_norm.weight"), H);
        }
        CM_HIP(hipStreamSynchronize(m.stream));
    } catch (const CmError&) {
        throw;
    } catch (const std::exception& e) {
        throw CmError(CM_ERR_IO, e.what());
    }
}

// ------------------------------------------------------------------------------------
// ISQ: quantise the bf16 linears already in HBM to Q8_0 and drop the bf16 copies.  The reference quantises every
// linear of the decoder plus (GGUF path only) keeps the embedding quantised; here: qkv, o, gate_up, down, lm_head
// (when untied); the embedding table stays bf16 (one row per token) and so does a tied lm_head.
// ------------------------------------------------------------------------------------
void Model::isq_q8_0(int mode) {
    // Tensor parallelism: every rank quantises ITS shard (dense family).  Q8_0 blocks run along K, and the row-parallel
    // shards (o_proj, down_proj) cut K on multiples of head_dim / of the intermediate slice, so a rank's blocks are the
    // blocks the unsharded matrix would have -- the codes are the same as quantise-then-shard.
    // (hybrid family, round 4: the same holds -- out_proj cuts K on whole 128-wide value heads, in_proj rows keep the full K = H;
    //  checked end to end by tests/test_gpu_tp_group.py::test_isq_under_tensor_parallelism)
    if (vcfg.present) throw CmError(CM_ERR_UNSUPPORTED, "ISQ of the vision-language checkpoints is not implemented");
    auto quant = [&](uint16_t* src, int N, int K) -> QWeight {
        if (K % 32) throw CmError(CM_ERR_UNSUPPORTED, "ISQ Q8_0 needs the input dimension to be a multiple of 32");
        QWeight w;
        w.fmt = QFMT_Q8_0; w.N = N; w.K = K;
        uint8_t* codes = (uint8_t*)dalloc<uint16_t>(((size_t)N * K + 1) / 2, true);
        uint8_t* d = (uint8_t*)dalloc<uint16_t>((size_t)N * (K / 32), true);
        launch_isq_q8_0(src, (size_t)K, N, K, codes, d, stream, mode);
        w.p0 = codes; w.p1 = d;
        quant_weight_bytes += w.bytes();
        return w;
    };
    const int H = cfg.H, D = cfg.D;
    for (LayerW& w : layers) {
        if (w.full) {
            const int rows = (cfg.hybrid ? 2 * Hq_l + 2 * Hkv_l : Hq_l + 2 * Hkv_l) * D;     // hybrid: [q | gate | k | v]
            w.q_qkv[0] = quant(w.qkv, rows, H); w.n_qkv = 1; w.qkv_row0[0] = 0;
            w.q_o = quant(w.o, H, Hq_l * D);
        } else {
            // Gated Delta Net: in_proj_qkv / in_proj_z / out_proj are quantised, the a / b gate projections never are
            // (ops/gdn/projection.rs:78-83)
            const int qz = cfg.conv_dim() + cfg.value_dim(), nba = 2 * cfg.NV;
            w.q_in_proj = quant(w.in_proj, qz, H);
            w.in_proj_ba = dalloc<uint16_t>((size_t)nba * H, true);
            CM_HIP(hipMemcpyAsync(w.in_proj_ba, w.in_proj + (size_t)qz * H, (size_t)nba * H * sizeof(uint16_t), hipMemcpyDeviceToDevice, stream));
            quant_weight_bytes += (uint64_t)nba * H * 2;
            w.q_out_proj = quant(w.out_proj, H, cfg.value_dim());
        }
        w.q_gate_up = quant(w.gate_up, 2 * I_l, H);
        w.q_down = quant(w.down, H, I_l);
    }
    // only a head the model owns is quantised (and freed below): a tied head -- or the hybrid family's fallback to the
    // embedding when the checkpoint has no lm_head.weight -- aliases the bf16 table that embed_row keeps reading.
    // Rows = the v_eff rows the loader allocated for this rank's vocabulary shard [v0, v0 + v_eff).
    const int v_eff_q = std::max(0, std::min(V_l, cfg.V - v0));
    if (lm_head_owned && v_eff_q > 0) q_lm_head = quant(lm_head, v_eff_q, H);
    CM_HIP(hipStreamSynchronize(stream));
    for (LayerW& w : layers) {
        dfree(w.qkv); dfree(w.o); dfree(w.gate_up); dfree(w.down); dfree(w.in_proj); dfree(w.out_proj);
        w.qkv = w.o = w.gate_up = w.down = w.in_proj = w.out_proj = nullptr;
    }
    if (lm_head_owned && q_lm_head.fmt != QFMT_NONE) { dfree(lm_head); lm_head = nullptr; lm_head_owned = false; }
    quantized = true;
}

}  // namespace cm
