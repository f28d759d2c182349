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
};

struct SynthSource : Source {
    Model& m;
    uint64_t seed;
    SynthSource(Model& mm, uint64_t s) : m(mm), seed(s) {}
    bool has(const std::string&) override { return true; }
    void spec(const std::string& name, double& std, float& off) {
        const Config& c = m.cfg;
        off = 0.f;
        auto ends = [&](const char* suf) {
            const size_t n = strlen(suf);
            return name.size() >= n && name.compare(name.size() - n, n, suf) == 0;
        };
        if (ends("embed_tokens.weight")) std = 1.0;
        else if (ends("norm.weight") || ends("layernorm.weight")) { std = 0.1; off = 1.0f; }
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

void build(Model& m, Source& src) {
    const Config& c = m.cfg;
    const int H = c.H, D = c.D, I = c.I;
    m.embed = m.dalloc<uint16_t>((size_t)c.V * H, true);
    src.fetch("model.embed_tokens.weight", c.V, H, 0, c.V, 0, H, m.embed, (size_t)H);
    m.norm = m.dalloc<uint16_t>((size_t)H, true);
    src.fetch("model.norm.weight", 1, H, 0, 1, 0, H, m.norm, (size_t)H);
    const int v_eff = std::max(0, std::min(m.V_l, c.V - m.v0));
    const bool have_head = !c.tie && src.has("lm_head.weight");
    if (!c.tie && !have_head) throw CmError(CM_ERR_IO, "tie_word_embeddings=false but lm_head.weight is missing");
    if (have_head) {
        m.lm_head = m.dalloc<uint16_t>((size_t)std::max(1, v_eff) * H, true);
        if (v_eff > 0) src.fetch("lm_head.weight", c.V, H, m.v0, v_eff, 0, H, m.lm_head, (size_t)H);
    } else {
        m.lm_head = m.embed + (size_t)m.v0 * H;     // tied: same tensor, no copy
    }
    m.layers.resize((size_t)c.L);
    const int qd = m.Hq_l * D, kd = m.Hkv_l * D;
    for (int li = 0; li < c.L; ++li) {
        LayerW& w = m.layers[(size_t)li];
        const std::string p = "model.layers." + std::to_string(li) + ".";
        w.qkv = m.dalloc<uint16_t>((size_t)(qd + 2 * kd) * H, true);
        src.fetch(p + "self_attn.q_proj.weight", c.Hq * D, H, m.rank * qd, qd, 0, H, w.qkv, (size_t)H);
        src.fetch(p + "self_attn.k_proj.weight", c.Hkv * D, H, m.kvh0 * D, kd, 0, H, w.qkv + (size_t)qd * H, (size_t)H);
        src.fetch(p + "self_attn.v_proj.weight", c.Hkv * D, H, m.kvh0 * D, kd, 0, H, w.qkv + (size_t)(qd + kd) * H, (size_t)H);
        w.o = m.dalloc<uint16_t>((size_t)H * qd, true);
        src.fetch(p + "self_attn.o_proj.weight", H, c.Hq * D, 0, H, m.rank * qd, qd, w.o, (size_t)qd);
        if (c.qk_norm && src.has(p + "self_attn.q_norm.weight")) {
            w.qn = m.dalloc<uint16_t>((size_t)D, true);
            w.kn = m.dalloc<uint16_t>((size_t)D, true);
            src.fetch(p + "self_attn.q_norm.weight", 1, D, 0, 1, 0, D, w.qn, (size_t)D);
            src.fetch(p + "self_attn.k_norm.weight", 1, D, 0, 1, 0, D, w.kn, (size_t)D);
        }
        w.gate_up = m.dalloc<uint16_t>((size_t)2 * m.I_l * H, true);
        src.fetch(p + "mlp.gate_proj.weight", I, H, m.rank * m.I_l, m.I_l, 0, H, w.gate_up, (size_t)2 * H);
        src.fetch(p + "mlp.up_proj.weight", I, H, m.rank * m.I_l, m.I_l, 0, H, w.gate_up + H, (size_t)2 * H);
        w.down = m.dalloc<uint16_t>((size_t)H * m.I_l, true);
        src.fetch(p + "mlp.down_proj.weight", H, I, 0, H, m.rank * m.I_l, m.I_l, w.down, (size_t)m.I_l);
        w.ln1 = m.dalloc<uint16_t>((size_t)H, true);
        w.ln2 = m.dalloc<uint16_t>((size_t)H, true);
        src.fetch(p + "input_layernorm.weight", 1, H, 0, 1, 0, H, w.ln1, (size_t)H);
        src.fetch(p + "post_attention_layernorm.weight", 1, H, 0, 1, 0, H, w.ln2, (size_t)H);
    }
    CM_HIP(hipStreamSynchronize(m.stream));
}

}  // namespace

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
