// Weight loading: HF safetensors checkpoint (mmap, sharded) or deterministic synthetic
// weights generated on the device.  Both produce the same HBM layout:
//   qkv      [(Hq_l + 2 Hkv_l) * D, H]   q rows | k rows | v rows   (merge: modeling.rs:187-204)
//   o        [H, Hq_l * D]                column slice of o_proj for this TP rank
//   gate_up  [2 * I_l, H]                 row 2j = gate_j, row 2j+1 = up_j (so SiLU*mul fuses
//                                         into the GEMV epilogue; reference concatenates
//                                         gate || up instead, modeling.rs:582-588)
//   down     [H, I_l]                     column slice
//   embed    [V, H] replicated; lm_head = embed when tied (modeling.rs:786-794) else rows [v0, v0+V_l)
// Tensor names: SURVEY.md Appendix A (the vb.pp(..) calls of qwen3/modeling.rs:166-221,566-580,
// 658-669,771-794).
#include <cmath>
#include <cstring>

#include "model.h"
#include "safetensors.h"
#include "tp.h"

namespace cm {

namespace {

uint32_t fnv1a32(const std::string& s) {
    uint32_t h = 0x811C9DC5u;
    for (unsigned char c : s) { h ^= c; h *= 0x01000193u; }
    return h;
}

struct Source {
    virtual ~Source() {}
    // copy rows [row0,row0+nrows) x cols [col0,col0+ncols) of tensor `name` (logical shape
    // [rows, full_cols]) to dst with row stride dst_stride (elements)
    virtual void fetch(const std::string& name, int rows, int full_cols, int row0, int nrows, int col0, int ncols,
                       uint16_t* dst, size_t dst_stride) = 0;
    virtual bool has(const std::string& name) = 0;
    virtual bool synthetic() const { return false; }
};

struct SynthSource : Source {
    Model& m;
    uint64_t seed;
    SynthSource(Model& mm, uint64_t s) : m(mm), seed(s) {}
    bool has(const std::string&) override { return true; }
    bool synthetic() const override { return true; }
    void spec(const std::string& name, double& std, float& off) {
        const Config& c = m.cfg;
        off = 0.f;
        auto ends = [&](const char* suf) {
            const size_t n = strlen(suf);
            return name.size() >= n && name.compare(name.size() - n, n, suf) == 0;
        };
        const bool hy = c.hybrid;
        if (name.compare(0, 13, "model.visual.") == 0) {         // crane_amd/synth.py qwen3_5_vl_specs
            const VisionCfg& v = m.vcfg;
            const int MH = v.hidden * v.merge * v.merge;
            if (ends("norm.weight") || ends("norm1.weight") || ends("norm2.weight")) { std = 0.1; off = 1.0f; }
            else if (ends(".bias")) std = 0.1;
            else if (ends("pos_embed.weight")) std = 0.5;
            else if (ends("patch_embed.proj.weight")) std = 1.0 / std::sqrt((double)v.patch_dim());
            else if (ends("mlp.linear_fc2.weight")) std = 1.0 / std::sqrt((double)v.inter);
            else if (name.find("merger") != std::string::npos && (ends("linear_fc1.weight") || ends("linear_fc2.weight"))) std = 1.0 / std::sqrt((double)MH);
            else std = 1.0 / std::sqrt((double)v.hidden);
            return;
        }
        if (ends("embed_tokens.weight")) std = 1.0;
        else if (ends("linear_attn.norm.weight")) { std = 0.1; off = 1.0f; }
        else if (ends("norm.weight") || ends("layernorm.weight")) { std = 0.1; off = hy ? 0.0f : 1.0f; }
        else if (ends("conv1d.weight")) std = 0.5;
        else if (ends("A_log")) { std = 0.1; off = -2.0f; }
        else if (ends("dt_bias")) std = 0.1;
        else if (ends("linear_attn.out_proj.weight")) std = 1.0 / std::sqrt((double)(c.NV_g * c.Vd));
        else if (ends("o_proj.weight")) std = 1.0 / std::sqrt((double)(c.Hq * c.D));
        else if (ends("down_proj.weight")) std = 1.0 / std::sqrt((double)c.I);
        else std = 1.0 / std::sqrt((double)c.H);      // q/k/v/gate/up/lm_head: fan_in = H
    }
    void fetch(const std::string& name, int, int full_cols, int row0, int nrows, int col0, int ncols, uint16_t* dst,
               size_t dst_stride) override {
        double sd; float off;
        spec(name, sd, off);
        const uint32_t tseed = fmix32(fnv1a32(name) ^ (uint32_t)((uint32_t)seed * 0x85EBCA6Bu + 0x1234567u));
        const float mul = (float)(sd / std::sqrt(21845.0));
        launch_synth_fill(dst, dst_stride, nrows, ncols, row0, col0, full_cols, tseed, mul, off, m.stream);
    }
};

struct FileSource : Source {
    Model& m;
    cmst::Checkpoint ck;
    std::vector<uint16_t> tmp;
    FileSource(Model& mm, const std::string& dir) : m(mm), ck(dir) {}
    bool has(const std::string& name) override { return ck.has(name); }
    void fetch(const std::string& name, int rows, int full_cols, int row0, int nrows, int col0, int ncols,
               uint16_t* dst, size_t dst_stride) override {
        const cmst::TensorView& t = ck.get(name);
        if (t.numel() != (int64_t)rows * full_cols)
            throw CmError(CM_ERR_IO, "tensor " + name + " has unexpected shape");
        const uint8_t* src = t.data;
        if (t.dtype != "BF16") {
            // F32 / F16 checkpoints: round to bf16 (the model dtype) on the host
            if (t.dtype != "F32" && t.dtype != "F16") throw CmError(CM_ERR_UNSUPPORTED, "tensor dtype " + t.dtype + " not supported (" + name + ")");
            if ((size_t)t.nbytes != (size_t)rows * full_cols * (t.dtype == "F32" ? 4u : 2u)) throw CmError(CM_ERR_IO, "tensor " + name + " truncated");
            tmp.resize((size_t)nrows * ncols);
            for (int r = 0; r < nrows; ++r)
                for (int c = 0; c < ncols; ++c) {
                    const size_t i = (size_t)(row0 + r) * full_cols + col0 + c;
                    float f;
                    if (t.dtype == "F32") memcpy(&f, t.data + i * 4, 4);
                    else if (t.dtype == "F16") {
                        uint16_t h; memcpy(&h, t.data + i * 2, 2);
                        const uint32_t sgn = (h >> 15) & 1, ex = (h >> 10) & 0x1F, mant = h & 0x3FF;
                        if (ex == 0) f = std::ldexp((float)mant, -24);
                        else if (ex == 31) f = mant ? NAN : INFINITY;
                        else f = std::ldexp((float)(mant | 0x400), (int)ex - 25);
                        if (sgn) f = -f;
                    } else throw CmError(CM_ERR_UNSUPPORTED, "tensor dtype " + t.dtype + " not supported (" + name + ")");
                    uint32_t u; memcpy(&u, &f, 4);
                    tmp[(size_t)r * ncols + c] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
                }
            CM_HIP(hipMemcpy2D(dst, dst_stride * 2, tmp.data(), (size_t)ncols * 2, (size_t)ncols * 2, (size_t)nrows,
                               hipMemcpyHostToDevice));
            return;
        }
        if ((size_t)t.nbytes != (size_t)rows * full_cols * 2) throw CmError(CM_ERR_IO, "tensor " + name + " truncated");
        CM_HIP(hipMemcpy2D(dst, dst_stride * 2, src + ((size_t)row0 * full_cols + col0) * 2, (size_t)full_cols * 2,
                           (size_t)ncols * 2, (size_t)nrows, hipMemcpyHostToDevice));
    }
};

// cm_tp_shard_plan: records what build() would copy for this rank instead of copying it
struct PlanSource : Source {
    Model& m;
    std::string js;
    explicit PlanSource(Model& mm) : m(mm) {}
    bool has(const std::string& name) override {
        return name.compare(0, 21, "model.language_model.") != 0 && name.compare(0, 15, "language_model.") != 0;
    }
    void fetch(const std::string& name, int rows, int full_cols, int row0, int nrows, int col0, int ncols, uint16_t* dst,
               size_t dst_stride) override {
        const size_t addr = (size_t)dst;
        size_t ai = 0, off = 0;
        for (size_t i = 0; i < m.plan_allocs.size(); ++i)
            if (addr >= m.plan_allocs[i].first && addr < m.plan_allocs[i].first + m.plan_allocs[i].second) { ai = i; off = (addr - m.plan_allocs[i].first) / 2; }
        if (!js.empty()) js += ", ";
        js += "{\"tensor\": \"" + name + "\", \"rows\": " + std::to_string(rows) + ", \"cols\": " + std::to_string(full_cols) +
              ", \"row0\": " + std::to_string(row0) + ", \"nrows\": " + std::to_string(nrows) + ", \"col0\": " + std::to_string(col0) +
              ", \"ncols\": " + std::to_string(ncols) + ", \"dst\": " + std::to_string(ai) + ", \"dst_off\": " + std::to_string(off) +
              ", \"dst_stride\": " + std::to_string(dst_stride) + "}";
    }
};

// small tensors the kernels want in f32 (norm weights with the Qwen3.5 "+1" folded in, conv taps, A_log, dt_bias):
// fetched as bf16 like everything else, then widened on the device
float* fetch_f32(Model& m, Source& src, const std::string& name, int n, float add, int total = -1, int start = 0) {
    if (total < 0) total = n;
    uint16_t* tmp = m.dalloc<uint16_t>((size_t)n);
    src.fetch(name, 1, total, 0, 1, start, n, tmp, (size_t)n);
    float* out = m.dalloc<float>((size_t)n, true);
    if (!m.plan_only) launch_bf16_to_f32(tmp, out, (size_t)n, add, m.stream);
    return out;
}

void build(Model& m, Source& src) {
    const Config& c = m.cfg;
    const int H = c.H, D = c.D, I = c.I;
    const float off = c.norm_off;
    // Qwen3.5 checkpoints keep the LM under `model.language_model.` (VLM) or `model.` (text-only);
    // prefix probing as qwen3_5/model.rs:65-74
    std::string pre = (src.synthetic() && m.vcfg.present) ? "model.language_model." : "model.";
    if (c.hybrid && !src.synthetic()) {
        for (const char* cand : {"model.language_model.", "language_model.", "model.", ""}) {
            if (src.has(std::string(cand) + "embed_tokens.weight")) { pre = cand; break; }
        }
    }
    m.embed = m.dalloc<uint16_t>((size_t)c.V * H, true);
    src.fetch(pre + "embed_tokens.weight", c.V, H, 0, c.V, 0, H, m.embed, (size_t)H);
    m.norm = fetch_f32(m, src, pre + "norm.weight", H, off);
    const int v_eff = std::max(0, std::min(m.V_l, c.V - m.v0));
    std::string head_name = "lm_head.weight";
    bool have_head = !c.tie && src.has(head_name);
    if (!c.tie && !have_head && c.hybrid && src.has(pre + "lm_head.weight")) { head_name = pre + "lm_head.weight"; have_head = true; }
    if (!c.tie && !have_head) {
        if (c.hybrid) {}   // falls back to tied, like qwen3_5/model.rs:106-123
        else throw CmError(CM_ERR_IO, "tie_word_embeddings=false but lm_head.weight is missing");
    }
    if (have_head) {
        m.lm_head = m.dalloc<uint16_t>((size_t)std::max(1, v_eff) * H, true);
        m.lm_head_owned = true;
        if (v_eff > 0) src.fetch(head_name, c.V, H, m.v0, v_eff, 0, H, m.lm_head, (size_t)H);
    } else {
        m.lm_head = m.embed + (size_t)m.v0 * H;     // tied: same tensor, no copy
    }
    m.layers.resize((size_t)c.L);
    const int qd = m.Hq_l * D, kd = m.Hkv_l * D;
    int gdn_idx = 0;
    for (int li = 0; li < c.L; ++li) {
        LayerW& w = m.layers[(size_t)li];
        const std::string p = pre + "layers." + std::to_string(li) + ".";
        w.full = c.layer_full(li);
        if (w.full && !c.hybrid) {
            w.qkv = m.dalloc<uint16_t>((size_t)(qd + 2 * kd) * H, true);
            src.fetch(p + "self_attn.q_proj.weight", c.Hq * D, H, m.rank * qd, qd, 0, H, w.qkv, (size_t)H);
            src.fetch(p + "self_attn.k_proj.weight", c.Hkv * D, H, m.kvh0 * D, kd, 0, H, w.qkv + (size_t)qd * H, (size_t)H);
            src.fetch(p + "self_attn.v_proj.weight", c.Hkv * D, H, m.kvh0 * D, kd, 0, H, w.qkv + (size_t)(qd + kd) * H, (size_t)H);
        } else if (w.full) {
            // q_proj rows are per head [q (D) | gate (D)] (qwen3_5/modeling.rs:428-455); HBM layout here is
            // [all q | all gate | k | v] so q/gate are contiguous vectors for the attention kernel
            w.qkv = m.dalloc<uint16_t>((size_t)(2 * qd + 2 * kd) * H, true);
            for (int h = 0; h < m.Hq_l; ++h) {
                const int hg = m.rank * m.Hq_l + h;                 // checkpoint head index of local head h
                src.fetch(p + "self_attn.q_proj.weight", c.Hq * 2 * D, H, hg * 2 * D, D, 0, H, w.qkv + (size_t)h * D * H, (size_t)H);
                src.fetch(p + "self_attn.q_proj.weight", c.Hq * 2 * D, H, hg * 2 * D + D, D, 0, H, w.qkv + (size_t)(qd + h * D) * H, (size_t)H);
            }
            src.fetch(p + "self_attn.k_proj.weight", c.Hkv * D, H, m.kvh0 * D, kd, 0, H, w.qkv + (size_t)(2 * qd) * H, (size_t)H);
            src.fetch(p + "self_attn.v_proj.weight", c.Hkv * D, H, m.kvh0 * D, kd, 0, H, w.qkv + (size_t)(2 * qd + kd) * H, (size_t)H);
        }
        if (w.full) {
            w.o = m.dalloc<uint16_t>((size_t)H * qd, true);
            src.fetch(p + "self_attn.o_proj.weight", H, c.Hq * D, 0, H, m.rank * qd, qd, w.o, (size_t)qd);
            if (c.qk_norm && src.has(p + "self_attn.q_norm.weight")) {
                w.qn = fetch_f32(m, src, p + "self_attn.q_norm.weight", D, off);
                w.kn = fetch_f32(m, src, p + "self_attn.k_norm.weight", D, off);
            }
        } else {
            w.gdn_idx = gdn_idx++;
            // local (this rank) and checkpoint-wide dims; rank r owns key heads [r NK, (r+1) NK) and their value heads
            const int cd = c.conv_dim(), vd = c.value_dim(), nv = c.NV, kdl = c.key_dim();
            const int KDg = c.NK_g * c.Kd, VDg = c.NV_g * c.Vd, CDg = 2 * KDg + VDg, r = m.rank;
            const int rows = cd + vd + 2 * nv, rows_pad = (rows + 127) / 128 * 128;
            w.in_proj = m.dalloc<uint16_t>((size_t)rows_pad * H, true);
            if (!m.plan_only) CM_HIP(hipMemsetAsync(w.in_proj, 0, (size_t)rows_pad * H * sizeof(uint16_t), m.stream));
            const std::string qn = p + "linear_attn.in_proj_qkv.weight";
            src.fetch(qn, CDg, H, r * kdl, kdl, 0, H, w.in_proj, (size_t)H);                                   // q
            src.fetch(qn, CDg, H, KDg + r * kdl, kdl, 0, H, w.in_proj + (size_t)kdl * H, (size_t)H);           // k
            src.fetch(qn, CDg, H, 2 * KDg + r * vd, vd, 0, H, w.in_proj + (size_t)(2 * kdl) * H, (size_t)H);   // v
            src.fetch(p + "linear_attn.in_proj_z.weight", VDg, H, r * vd, vd, 0, H, w.in_proj + (size_t)cd * H, (size_t)H);
            src.fetch(p + "linear_attn.in_proj_b.weight", c.NV_g, H, r * nv, nv, 0, H, w.in_proj + (size_t)(cd + vd) * H, (size_t)H);
            src.fetch(p + "linear_attn.in_proj_a.weight", c.NV_g, H, r * nv, nv, 0, H, w.in_proj + (size_t)(cd + vd + nv) * H, (size_t)H);
            w.out_proj = m.dalloc<uint16_t>((size_t)H * vd, true);
            src.fetch(p + "linear_attn.out_proj.weight", H, VDg, 0, H, r * vd, vd, w.out_proj, (size_t)vd);
            // conv taps [conv_dim, 1, k] -> the three channel ranges of this rank, contiguous in [q | k | v] order
            const int ck = c.conv_k;
            const std::string cn = p + "linear_attn.conv1d.weight";
            if (m.tp == 1) {
                w.conv_w = fetch_f32(m, src, cn, cd * ck, 0.f);
            } else {
                uint16_t* tmp = m.dalloc<uint16_t>((size_t)cd * ck);
                src.fetch(cn, 1, CDg * ck, 0, 1, (r * kdl) * ck, kdl * ck, tmp, (size_t)kdl * ck);
                src.fetch(cn, 1, CDg * ck, 0, 1, (KDg + r * kdl) * ck, kdl * ck, tmp + (size_t)kdl * ck, (size_t)kdl * ck);
                src.fetch(cn, 1, CDg * ck, 0, 1, (2 * KDg + r * vd) * ck, vd * ck, tmp + (size_t)2 * kdl * ck, (size_t)vd * ck);
                w.conv_w = m.dalloc<float>((size_t)cd * ck, true);
                if (!m.plan_only) launch_bf16_to_f32(tmp, w.conv_w, (size_t)cd * ck, 0.f, m.stream);
            }
            w.A_log = fetch_f32(m, src, p + "linear_attn.A_log", nv, 0.f, c.NV_g, r * nv);
            w.dt_bias = fetch_f32(m, src, p + "linear_attn.dt_bias", nv, 0.f, c.NV_g, r * nv);
            w.gnorm = fetch_f32(m, src, p + "linear_attn.norm.weight", c.Vd, 0.f);              // plain weight (norm.rs:39-45)
        }
        w.gate_up = m.dalloc<uint16_t>((size_t)2 * m.I_l * H, true);
        src.fetch(p + "mlp.gate_proj.weight", I, H, m.rank * m.I_l, m.I_l, 0, H, w.gate_up, (size_t)2 * H);
        src.fetch(p + "mlp.up_proj.weight", I, H, m.rank * m.I_l, m.I_l, 0, H, w.gate_up + H, (size_t)2 * H);
        w.down = m.dalloc<uint16_t>((size_t)H * m.I_l, true);
        src.fetch(p + "mlp.down_proj.weight", H, I, 0, H, m.rank * m.I_l, m.I_l, w.down, (size_t)m.I_l);
        w.ln1 = fetch_f32(m, src, p + "input_layernorm.weight", H, off);
        w.ln2 = fetch_f32(m, src, p + "post_attention_layernorm.weight", H, off);
    }
    if (m.vcfg.present) {
        // vision tower under `model.visual.` (qwen3_5/vlm.rs:125)
        const VisionCfg& v = m.vcfg;
        const std::string vp = "model.visual.";
        const int VH = v.hidden, VI = v.inter, PD = v.patch_dim(), MH = VH * v.merge * v.merge;
        auto mat = [&](const std::string& n, int rows, int cols) {
            uint16_t* p = m.dalloc<uint16_t>((size_t)rows * cols, true);
            src.fetch(n, rows, cols, 0, rows, 0, cols, p, (size_t)cols);
            return p;
        };
        m.vw.patch_w = mat(vp + "patch_embed.proj.weight", VH, PD);       // [VH, C, T, P, P] flattened
        m.vw.patch_b = fetch_f32(m, src, vp + "patch_embed.proj.bias", VH, 0.f);
        m.vw.pos_table = mat(vp + "pos_embed.weight", v.num_pos, VH);
        m.vw.blocks.resize((size_t)v.depth);
        for (int i = 0; i < v.depth; ++i) {
            VisionBlockW& b = m.vw.blocks[(size_t)i];
            const std::string bp = vp + "blocks." + std::to_string(i) + ".";
            b.n1w = fetch_f32(m, src, bp + "norm1.weight", VH, 0.f); b.n1b = fetch_f32(m, src, bp + "norm1.bias", VH, 0.f);
            b.n2w = fetch_f32(m, src, bp + "norm2.weight", VH, 0.f); b.n2b = fetch_f32(m, src, bp + "norm2.bias", VH, 0.f);
            b.qkv_w = mat(bp + "attn.qkv.weight", 3 * VH, VH); b.qkv_b = fetch_f32(m, src, bp + "attn.qkv.bias", 3 * VH, 0.f);
            b.proj_w = mat(bp + "attn.proj.weight", VH, VH); b.proj_b = fetch_f32(m, src, bp + "attn.proj.bias", VH, 0.f);
            b.fc1_w = mat(bp + "mlp.linear_fc1.weight", VI, VH); b.fc1_b = fetch_f32(m, src, bp + "mlp.linear_fc1.bias", VI, 0.f);
            b.fc2_w = mat(bp + "mlp.linear_fc2.weight", VH, VI); b.fc2_b = fetch_f32(m, src, bp + "mlp.linear_fc2.bias", VH, 0.f);
        }
        m.vw.mn_w = fetch_f32(m, src, vp + "merger.norm.weight", VH, 0.f); m.vw.mn_b = fetch_f32(m, src, vp + "merger.norm.bias", VH, 0.f);
        m.vw.mfc1_w = mat(vp + "merger.linear_fc1.weight", MH, MH); m.vw.mfc1_b = fetch_f32(m, src, vp + "merger.linear_fc1.bias", MH, 0.f);
        m.vw.mfc2_w = mat(vp + "merger.linear_fc2.weight", v.out_hidden, MH); m.vw.mfc2_b = fetch_f32(m, src, vp + "merger.linear_fc2.bias", v.out_hidden, 0.f);
        // DeepStack mergers (qwen3_vl/vision.rs:335-341): PatchMerger with use_postshuffle_norm = true
        m.vw.deep.resize(v.deepstack.size());
        for (size_t k = 0; k < v.deepstack.size(); ++k) {
            VisionW::Merger& d = m.vw.deep[k];
            const std::string dp = vp + "deepstack_merger_list." + std::to_string(k) + ".";
            d.n_w = fetch_f32(m, src, dp + "norm.weight", MH, 0.f); d.n_b = fetch_f32(m, src, dp + "norm.bias", MH, 0.f);
            d.fc1_w = mat(dp + "linear_fc1.weight", MH, MH); d.fc1_b = fetch_f32(m, src, dp + "linear_fc1.bias", MH, 0.f);
            d.fc2_w = mat(dp + "linear_fc2.weight", v.out_hidden, MH); d.fc2_b = fetch_f32(m, src, dp + "linear_fc2.bias", v.out_hidden, 0.f);
        }
    }
    if (!m.plan_only) CM_HIP(hipStreamSynchronize(m.stream));
}

}  // namespace

// Host only: the copies load_from_dir would make for rank `tp_rank` of `tp_size` -- every (tensor, row range, column range) ->
// (allocation, element offset, row stride) -- and the rank's shard geometry.  The SAME build() the device loader runs, over a
// recording source and a model that never touches a device: what tests/test_tp_sharding.py compares with crane_amd/tp.py.
std::string tp_shard_plan_json(const std::string& config_json, int tp_size, int tp_rank) {
    Model m;
    m.plan_only = true;
    cm_opts o{};
    o.tp_size = tp_size; o.tp_rank = tp_rank;
    m.init_common(config_json, &o);
    PlanSource src(m);
    build(m, src);
    std::string js = "{\"tp\": " + std::to_string(m.tp) + ", \"rank\": " + std::to_string(m.rank) + ", \"Hq_l\": " + std::to_string(m.Hq_l) +
                     ", \"Hkv_l\": " + std::to_string(m.Hkv_l) + ", \"kvh0\": " + std::to_string(m.kvh0) + ", \"I_l\": " + std::to_string(m.I_l) +
                     ", \"V_l\": " + std::to_string(m.V_l) + ", \"v0\": " + std::to_string(m.v0) + ", \"NK_l\": " + std::to_string(m.cfg.NK) +
                     ", \"NV_l\": " + std::to_string(m.cfg.NV) + ", \"weight_bytes\": " + std::to_string(m.weight_bytes) + ", \"copies\": [" + src.js + "]}";
    return js;
}

void load_from_dir(Model& m, const std::string& dir) {
    try {
        FileSource src(m, dir);
        build(m, src);
    } catch (const CmError&) {
        throw;
    } catch (const std::exception& e) {
        throw CmError(CM_ERR_IO, e.what());
    }
}

void load_synthetic(Model& m, uint64_t seed) {
    SynthSource src(m, seed);
    build(m, src);
}

}  // namespace cm
