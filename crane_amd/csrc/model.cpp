// Host runtime: see model.h.  One Model = one replica (or one TP rank) on one GPU, with
// its own HIP stream; nothing global, so N handles = N replicas
// (ModelBackend is `Send`, never shared: crane-serve/src/lib.rs:1129-1132).
#include "model.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "json_min.h"
#include "tp.h"

namespace cm {

template <typename T>
T* Model::dalloc(size_t n, bool count_weight) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    if (plan_only) {
        const size_t base = 0x10000000ull + plan_bump;
        plan_bump += (bytes + 255) / 256 * 256;
        plan_allocs.emplace_back(base, bytes);
        if (count_weight) weight_bytes += bytes;
        return (T*)base;
    }
    CM_HIP(hipMalloc(&p, bytes));
    // CM_DEBUG_POISON=<byte>[,<first>,<last>]: fill every new allocation (or allocations number first..last of this handle) with
    // the byte -- recycled device memory is not zero, a kernel that reads a buffer before anything wrote it must show up in tests
    // on a FRESH process too (0xFF = NaN patterns)
    static const char* poison = getenv("CM_DEBUG_POISON");
    if (poison) {
        int b = 0xFF, lo = 0, hi = 1 << 30;
        sscanf(poison, "%i,%d,%d", &b, &lo, &hi);
        const int idx = (int)allocs.size();
        if (idx >= lo && idx <= hi) {       // on the handle's own (non-blocking) stream: ordered before whatever fills the buffer
            CM_HIP(hipMemsetAsync(p, b, bytes, stream));
            CM_HIP(hipStreamSynchronize(stream));
        }
    }
    allocs.push_back(p);
    alloc_sizes.push_back(count_weight ? bytes : 0);
    if (count_weight) weight_bytes += bytes;
    return (T*)p;
}
template uint16_t* Model::dalloc<uint16_t>(size_t, bool);
template float* Model::dalloc<float>(size_t, bool);
template int* Model::dalloc<int>(size_t, bool);
template uint32_t* Model::dalloc<uint32_t>(size_t, bool);

void Model::dfree(void* p) {
    if (!p) return;
    for (size_t i = 0; i < allocs.size(); ++i)
        if (allocs[i] == p) {
            weight_bytes -= alloc_sizes[i];
            (void)hipFree(p);
            allocs.erase(allocs.begin() + (long)i);
            alloc_sizes.erase(alloc_sizes.begin() + (long)i);
            return;
        }
}

Model::~Model() {
    if (plan_only) return;
    if (stream) (void)hipStreamSynchronize(stream);
    for (int v = 0; v < 5; ++v) {
        if (graph_exec[v]) (void)hipGraphExecDestroy(graph_exec[v]);
        if (graph[v]) (void)hipGraphDestroy(graph[v]);
    }
    rccl.reset();
    for (void* p : allocs) (void)hipFree(p);
    if (h_bt) (void)hipHostFree(h_bt);
    if (h_st) (void)hipHostFree(h_st);
    if (h_ring) (void)hipHostFree(h_ring);
    if (h_logits) (void)hipHostFree(h_logits);
    if (h_ids) (void)hipHostFree(h_ids);
    if (h_stb) (void)hipHostFree(h_stb);
    if (h_btb) (void)hipHostFree(h_btb);
    if (h_logitsb) (void)hipHostFree(h_logitsb);
    if (h_pen) (void)hipHostFree(h_pen);
    if (h_tk) (void)hipHostFree(h_tk);
    if (h_stab) (void)hipHostFree(h_stab);
    if (v_ev0) (void)hipEventDestroy(v_ev0);
    if (v_ev1) (void)hipEventDestroy(v_ev1);
    if (tk_cand) (void)hipFree(tk_cand);
    if (tk_in) (void)hipFree(tk_in);
    if (stream) (void)hipStreamDestroy(stream);
}

// ------------------------------------------------------------------------------------
// construction
// ------------------------------------------------------------------------------------
void Model::init_common(const std::string& config_json, const cm_opts* o) {
    if (o) {
        opts = *o;
        if (opts.abi_version != 0 && (opts.abi_version < 2 || opts.abi_version > CM_ABI_VERSION))    // 2: the same layout with zeroed reserved words
            throw CmError(CM_ERR_INVALID, "cm_opts.abi_version mismatch");
    }
    cmjson::ValuePtr j;
    try {
        j = cmjson::parse(config_json);
    } catch (const std::exception& e) {
        throw CmError(CM_ERR_IO, std::string("config.json: ") + e.what());
    }
    if (j->kind != cmjson::Value::Obj) throw CmError(CM_ERR_IO, "config.json: not an object");
    const cmjson::Value* root = j.get();
    // VLM checkpoints nest the LM config under text_config
    if (root->has("text_config") && !root->has("hidden_size")) root = root->get("text_config");
    cfg.model_type = root->string("model_type", j->string("model_type", "qwen3"));
    cfg.V = (int)root->integer("vocab_size", 0);
    cfg.H = (int)root->integer("hidden_size", 0);
    cfg.I = (int)root->integer("intermediate_size", 0);
    cfg.L = (int)root->integer("num_hidden_layers", 0);
    cfg.Hq = (int)root->integer("num_attention_heads", 0);
    cfg.Hkv = (int)root->integer("num_key_value_heads", cfg.Hq);
    cfg.max_pos = (int)root->integer("max_position_embeddings", 0);
    cfg.eps = (float)root->number("rms_norm_eps", 1e-6);
    cfg.theta = root->number("rope_theta", 1e6);                 // default_rope_theta modeling.rs:107
    cfg.attention_bias = root->boolean("attention_bias", false);
    cfg.qk_norm = root->boolean("use_qk_norm", true);            // default true modeling.rs:111
    cfg.tie = root->boolean("tie_word_embeddings", true);        // default true modeling.rs:113
    cfg.D = root->has("head_dim") ? (int)root->integer("head_dim", 0) : (cfg.Hq ? cfg.H / cfg.Hq : 0);
    cfg.eos = root->has("eos_token_id") && root->get("eos_token_id")->kind == cmjson::Value::Num
                  ? root->integer("eos_token_id", -1) : -1;
    if (cfg.V <= 0 || cfg.H <= 0 || cfg.I <= 0 || cfg.L <= 0 || cfg.Hq <= 0 || cfg.Hkv <= 0 || cfg.max_pos <= 0)
        throw CmError(CM_ERR_IO, "config.json: missing required field");
    cfg.rot_dim = cfg.D;
    if (const cmjson::Value* vc = j->get("vision_config")) {     // VisionConfig (qwen3_5/config.rs:120-160, qwen3_vl/config.rs:18-44)
        if (vc->kind == cmjson::Value::Obj && root != j.get()) {
            vcfg.present = true;
            vcfg.depth = (int)vc->integer("depth", 0);
            vcfg.hidden = (int)vc->integer("hidden_size", 0);
            vcfg.heads = (int)vc->integer("num_heads", 0);
            vcfg.inter = (int)vc->integer("intermediate_size", 0);
            vcfg.patch = (int)vc->integer("patch_size", 16);
            vcfg.tpatch = (int)vc->integer("temporal_patch_size", 2);
            vcfg.merge = (int)vc->integer("spatial_merge_size", 2);
            vcfg.in_ch = (int)vc->integer(vc->has("in_channels") ? "in_channels" : "in_chans", 3);
            vcfg.out_hidden = (int)vc->integer("out_hidden_size", 0);
            vcfg.num_pos = (int)vc->integer("num_position_embeddings", 0);
            vcfg.act = vc->string("hidden_act", "gelu_pytorch_tanh") == "gelu_pytorch_tanh" ? 1 : 2;
            const char* mg = getenv("CM_VISION_MERGER_GELU");
            vcfg.merger_act = (mg && std::string(mg) == "erf") ? 2 : 1;
            vcfg.image_token = j->integer("image_token_id", -1);
            if (const cmjson::Value* ds = vc->get("deepstack_visual_indexes"))
                for (auto& e : ds->arr) vcfg.deepstack.push_back((int)e->num);
            for (size_t k = 0; k < vcfg.deepstack.size(); ++k)
                if (vcfg.deepstack[k] < 0 || vcfg.deepstack[k] >= vcfg.depth || (k && vcfg.deepstack[k] <= vcfg.deepstack[k - 1]))
                    throw CmError(CM_ERR_IO, "bad deepstack_visual_indexes");
            if (vcfg.depth <= 0 || vcfg.heads <= 0 || vcfg.hidden % vcfg.heads) throw CmError(CM_ERR_IO, "bad vision_config");
            if (vcfg.hidden / vcfg.heads != 64) throw CmError(CM_ERR_UNSUPPORTED, "vision head_dim must be 64 (72 not implemented)");
            const int side = (int)std::lround(std::sqrt((double)vcfg.num_pos));
            if (side * side != vcfg.num_pos) throw CmError(CM_ERR_IO, "num_position_embeddings is not a perfect square");
        }
    }
    if (cfg.model_type == "qwen3_vl" || cfg.model_type == "qwen3_vl_text") {
        // Qwen3-VL: the dense Qwen3 decoder (qwen3_vl/text.rs:34-260) + 3-axis interleaved MRoPE over the whole head (HF
        // Qwen3VLTextRotaryEmbedding; the reference's dead code rotates at a 1-D offset, identical for text-only prompts)
        cfg.model_type = "qwen3";
        cfg.tie = root->has("tie_word_embeddings") ? root->boolean("tie_word_embeddings", true) : j->boolean("tie_word_embeddings", true);
        cfg.theta = root->number("rope_theta", 5e6);
        cfg.mrope_sec[0] = 24; cfg.mrope_sec[1] = 20; cfg.mrope_sec[2] = 20;
        for (const char* key : {"rope_parameters", "rope_scaling"})
            if (const cmjson::Value* rp = root->get(key)) {
                if (rp->kind != cmjson::Value::Obj) continue;
                cfg.theta = rp->number("rope_theta", cfg.theta);
                if (const cmjson::Value* ms = rp->get("mrope_section"))
                    for (size_t i = 0; i < ms->arr.size() && i < 3; ++i) cfg.mrope_sec[i] = (int)ms->arr[i]->num;
            }
        if (cfg.mrope_sec[0] + cfg.mrope_sec[1] + cfg.mrope_sec[2] != cfg.D / 2) throw CmError(CM_ERR_INVALID, "mrope_section must sum to head_dim / 2");
    }
    if (cfg.model_type == "qwen3_5" || cfg.model_type == "qwen3_5_text") {
        // TextConfig (qwen3_5/config.rs:47-110); tie_word_embeddings defaults to FALSE here (:80-81)
        cfg.hybrid = true;
        cfg.norm_off = 1.0f;
        cfg.qk_norm = true;
        cfg.tie = root->has("tie_word_embeddings") ? root->boolean("tie_word_embeddings", false)
                                                   : j->boolean("tie_word_embeddings", false);
        cfg.interval = (int)root->integer("full_attention_interval", 4);
        if (cfg.interval <= 0) throw CmError(CM_ERR_INVALID, "full_attention_interval must be positive");
        cfg.conv_k = (int)root->integer("linear_conv_kernel_dim", 4);
        cfg.Kd = (int)root->integer("linear_key_head_dim", 0);
        cfg.Vd = (int)root->integer("linear_value_head_dim", 0);
        cfg.NK = (int)root->integer("linear_num_key_heads", 0);
        cfg.NV = (int)root->integer("linear_num_value_heads", 0);
        cfg.attn_gate = root->boolean("attn_output_gate", true);
        double prf = 0.25;
        cfg.theta = 1e7;
        if (const cmjson::Value* rp = root->get("rope_parameters")) {
            cfg.theta = rp->number("rope_theta", 1e7);
            prf = rp->number("partial_rotary_factor", 0.25);
        }
        cfg.rot_dim = (int)((double)cfg.D * prf);                  // config.rs:226-229
        if (const cmjson::Value* rp = root->get("rope_parameters"))
            if (const cmjson::Value* ms = rp->get("mrope_section"))
                for (size_t i = 0; i < ms->arr.size() && i < 3; ++i) cfg.mrope_sec[i] = (int)ms->arr[i]->num;
        if (const cmjson::Value* lt = root->get("layer_types")) {   // HF spelling; must agree with the interval rule
            for (size_t i = 0; i < lt->arr.size() && (int)i < cfg.L; ++i) {
                const bool full = lt->arr[i]->str == "full_attention";
                if (full != (((int)i + 1) % cfg.interval == 0))
                    throw CmError(CM_ERR_UNSUPPORTED, "layer_types does not follow full_attention_interval");
            }
        }
        if (cfg.Kd != 128 || cfg.Vd != 128 || cfg.conv_k != 4)
            throw CmError(CM_ERR_UNSUPPORTED, "GDN kernel is built for head dims 128/128 and conv kernel 4");
        if (cfg.NK <= 0 || cfg.NV % cfg.NK) throw CmError(CM_ERR_INVALID, "linear_num_value_heads % linear_num_key_heads != 0");
        if (!cfg.attn_gate) throw CmError(CM_ERR_UNSUPPORTED, "attn_output_gate=false not implemented");
        if (cfg.rot_dim <= 0 || cfg.rot_dim % 2 || cfg.rot_dim > cfg.D) throw CmError(CM_ERR_INVALID, "bad partial_rotary_factor");
        cfg.NK_g = cfg.NK; cfg.NV_g = cfg.NV;
        if (opts.tp_size > 1) {
            // TP for the hybrid family (new design): each rank owns NK/tp key heads and the NV/tp value
            // heads paired with them (HF "Interleaved" order keeps a key head's value heads contiguous)
            if (cfg.NK % opts.tp_size || cfg.NV % opts.tp_size)
                throw CmError(CM_ERR_INVALID, "tp_size must divide linear_num_key_heads and linear_num_value_heads");
            cfg.NK /= opts.tp_size; cfg.NV /= opts.tp_size;
        }
    } else if (cfg.model_type != "qwen3") {
        throw CmError(CM_ERR_UNSUPPORTED, "model_type '" + cfg.model_type + "' not implemented (qwen3, qwen3_5)");
    }
    if (cfg.attention_bias) throw CmError(CM_ERR_UNSUPPORTED, "attention_bias not implemented");
    if (cfg.D != 128 && cfg.D != 256) throw CmError(CM_ERR_UNSUPPORTED, "head_dim must be 128 or 256");
    if (cfg.H % 8 || cfg.I % 8) throw CmError(CM_ERR_UNSUPPORTED, "hidden/intermediate must be multiples of 8");
    if (cfg.Hq % cfg.Hkv) throw CmError(CM_ERR_INVALID, "num_attention_heads % num_key_value_heads != 0");

    dev = opts.device;
    if (!plan_only) {
        CM_HIP(hipSetDevice(dev));
        hipDeviceProp_t prop;
        CM_HIP(hipGetDeviceProperties(&prop, dev));
        num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        CM_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    }

    tp = opts.tp_size > 0 ? opts.tp_size : 1;
    rank = opts.tp_rank;
    if (rank < 0 || rank >= tp) throw CmError(CM_ERR_INVALID, "tp_rank out of range");
    if (cfg.Hq % tp || cfg.I % tp || (cfg.I / tp) % 8)
        throw CmError(CM_ERR_INVALID, "tp_size must divide heads and intermediate size");
    if (!(cfg.Hkv % tp == 0 || tp % cfg.Hkv == 0))
        throw CmError(CM_ERR_INVALID, "tp_size incompatible with num_key_value_heads");
    Hq_l = cfg.Hq / tp;
    if (cfg.Hkv >= tp) { Hkv_l = cfg.Hkv / tp; kvh0 = rank * Hkv_l; }
    else { Hkv_l = 1; kvh0 = rank * cfg.Hkv / tp; }          // KV heads replicated tp/Hkv times
    nrep = Hq_l / Hkv_l;
    if (!(nrep == 1 || nrep == 2 || nrep == 3 || nrep == 4 || nrep == 6 || nrep == 8))
        throw CmError(CM_ERR_UNSUPPORTED, "GQA group size not instantiated");
    I_l = cfg.I / tp;
    V_l = (cfg.V + tp - 1) / tp;
    v0 = rank * V_l;

    page = opts.kv_block_size ? (int)opts.kv_block_size : 64;
    max_seq = opts.max_seq_len ? (int)opts.max_seq_len : std::min(cfg.max_pos, 32768);
    if (max_seq > cfg.max_pos) max_seq = cfg.max_pos;
    max_pages_per_seq = (max_seq + page - 1) / page;
    const int max_seqs = opts.max_seqs ? (int)opts.max_seqs : 8;
    n_pages = opts.kv_pool_tokens ? (int64_t)((opts.kv_pool_tokens + page - 1) / page)
                                  : (int64_t)max_seqs * max_pages_per_seq;
    if (n_pages < max_pages_per_seq) n_pages = max_pages_per_seq;
    nsplit = std::max(1, std::min(64, num_cu / std::max(1, Hkv_l)));
    // every environment switch of the model is read here, once, at cm_create (tuning / A-B switches; none changes results
    // beyond summation order except CM_QUANT_ACT and CM_QUANT_PREFILL, which select a documented arithmetic)
    if (const char* e = getenv("CM_TP_GRAPH")) tp_graph = atoi(e) != 0;
    if (const char* e = getenv("CM_QUANT_PREFILL")) quant_prefill = atoi(e) != 0;
    if (const char* e = getenv("CM_QUANT_PREFILL_INT8")) q8_prefill_want = atoi(e) != 0;
    no_prefill = getenv("CM_NO_PREFILL") != nullptr;
    if (const char* e = getenv("CM_GEMVM")) use_mfma_gemv = atoi(e) != 0;
    if (const char* e = getenv("CM_BATCH_GEMM_MIN")) batch_gemm_min = std::max(0, std::min((int)GEMV_MAXB, atoi(e)));
    if (const char* e = getenv("CM_LM_HEAD_GEMM_MIN")) lm_head_gemm_min = std::max(0, atoi(e));
    if (const char* e = getenv("CM_Q_GEMM_MIN")) q_gemm_min = std::max(0, atoi(e));
    if (const char* e = getenv("CM_ATTN_OUTQ")) attn_outq = atoi(e) != 0;
    if (const char* e = getenv("CM_BATCH_MAX")) batch_max = std::max(8, std::min((int)GEMV_MAXB, atoi(e) / 8 * 8));
    if (const char* e = getenv("CM_QUANT_ACT")) quant_act_int = std::string(e) != "f32";
    if (const char* e = getenv("CM_ATTN_HEADS_MAX")) attn_heads_max = atoll(e);
    if (const char* e = getenv("CM_ATTN_NS")) attn_ns = std::max(1, std::min(nsplit, atoi(e)));
    if (const char* e = getenv("CM_ATTN_MFMA_MIN")) attn_mfma_min = atoll(e);
    if (const char* e = getenv("CM_GDN_DEFER_NORM")) gdn_defer_norm = atoi(e) != 0;
    if (const char* e = getenv("CM_ENGINE_HYBRID")) hybrid_engine = atoi(e) != 0;
    if (const char* e = getenv("CM_PREFILL_SEG_BATCH")) seg_batch = atoi(e) != 0;
    if (const char* e = getenv("CM_ATTN_BATCH_NS_MIN")) attn_batch_ns_min = std::max(1, atoi(e));
    if (const char* e = getenv("CM_ATTN_MFMA_MIN_BATCH")) attn_mfma_min_batch = atoll(e);
    if (const char* e = getenv("CM_ATTN_MFMA_WIDE_MIN")) attn_mfma_wide_min = atoll(e);
    nsplit_mfma = std::max(nsplit, std::min(64, 2 * num_cu / std::max(1, Hkv_l)));
    if (const char* e = getenv("CM_ATTN_MFMA_NSPLIT")) nsplit_mfma = std::max(1, std::min(64, atoi(e)));
    use_graph = opts.use_graph >= 0;
    // public cm_kv_dtype -> internal page element type (KV_*, the template parameter of the attention kernels)
    switch (opts.kv_dtype) {
        case CM_KV_F16: kv_mode = KV_F16; break;
        case CM_KV_F32: kv_mode = KV_F32; break;
        case CM_KV_INT8: kv_mode = KV_INT8; break;
        case CM_KV_INT4: kv_mode = KV_INT4; break;
        case CM_KV_BF16: kv_mode = KV_BF16; break;
        default: throw CmError(CM_ERR_INVALID, "bad kv_dtype");
    }
    kv_f32 = kv_mode == KV_F32;
    kv_esize = kv_f32 ? 4 : (kvq() ? 1 : 2);
    seqs.resize((size_t)max_seqs + 1);
    seqs[0].used = true;
}

void Model::alloc_runtime() {
    const int H = cfg.H, D = cfg.D;
    x = dalloc<float>(H);
    y = dalloc<float>(H);
    gdn_layers = 0;
    for (int i = 0; i < cfg.L; ++i) if (!cfg.layer_full(i)) ++gdn_layers;
    in_proj_rows = cfg.hybrid ? cfg.conv_dim() + cfg.value_dim() + 2 * cfg.NV : 0;
    in_proj_pad = (in_proj_rows + 127) / 128 * 128;
    const size_t attn_rows = (size_t)(cfg.hybrid ? 2 * Hq_l + 2 * Hkv_l : Hq_l + 2 * Hkv_l) * D;
    qkv = dalloc<float>(std::max(attn_rows, (size_t)in_proj_pad));
    attn = dalloc<float>(std::max((size_t)Hq_l * D, (size_t)(cfg.hybrid ? cfg.value_dim() : 0)));
    if (gdn_layers > 0) {
        conv_slot_elems = (size_t)gdn_layers * 2 * cfg.conv_dim() * (cfg.conv_k - 1);
        state_slot_elems = (size_t)gdn_layers * cfg.NV * cfg.Kd * cfg.Vd;
        conv_pool = dalloc<float>(conv_slot_elems * seqs.size());
        state_pool = dalloc<float>(state_slot_elems * seqs.size());
        gdn_scratch = dalloc<float>((size_t)MAXB * cfg.NV * (cfg.Vd + 4));
        gdn_ticket = dalloc<int>((size_t)MAXB * cfg.NV);
        CM_HIP(hipMemsetAsync(gdn_ticket, 0, (size_t)MAXB * cfg.NV * sizeof(int), stream));
        CM_HIP(hipMemsetAsync(conv_pool, 0, conv_slot_elems * seqs.size() * sizeof(float), stream));
        CM_HIP(hipMemsetAsync(state_pool, 0, state_slot_elems * seqs.size() * sizeof(float), stream));
    }
    hbuf = dalloc<float>(I_l);
    logits = dalloc<float>((size_t)V_l * tp);
    part_o = dalloc<float>((size_t)Hq_l * std::max(nsplit, nsplit_mfma) * D);
    part_ml = dalloc<float>((size_t)Hq_l * std::max(nsplit, nsplit_mfma) * 2);
    const int v_eff = std::max(0, std::min(V_l, cfg.V - v0));
    lm_grid = (quantized && q_lm_head.fmt != QFMT_NONE) ? gemvq_grid(v_eff, num_cu, q_lm_head.fmt) : gemv_grid(v_eff, H, num_cu, false);
    pmax = dalloc<float>((size_t)lm_grid * tp);
    pidx = dalloc<int>((size_t)lm_grid * tp);
    st = (StepState*)dalloc<int>(sizeof(StepState) / sizeof(int));
    ring = (uint32_t*)dalloc<int>(RING);
    CM_HIP(hipMemsetAsync(st, 0, sizeof(StepState), stream));
    CM_HIP(hipHostMalloc((void**)&h_st, sizeof(StepState)));
    CM_HIP(hipHostMalloc((void**)&h_ring, RING * sizeof(uint32_t)));
    CM_HIP(hipHostMalloc((void**)&h_logits, (size_t)V_l * tp * sizeof(float)));
    CM_HIP(hipHostMalloc((void**)&h_bt, (size_t)max_pages_per_seq * sizeof(int32_t)));
    d_bt = dalloc<int>(max_pages_per_seq);
    CM_HIP(hipMemsetAsync(d_bt, 0, (size_t)max_pages_per_seq * sizeof(int32_t), stream));

    // KV pool
    page_elems = (size_t)Hkv_l * page * D;
    kv_index.assign((size_t)cfg.L, -1);
    n_kv_layers = 0;
    for (int i = 0; i < cfg.L; ++i) if (cfg.layer_full(i)) kv_index[(size_t)i] = n_kv_layers++;
    kv_row_bytes = kv_mode == KV_F32 ? (size_t)D * 4 : (kv_mode == KV_INT8 ? (size_t)D : (kv_mode == KV_INT4 ? (size_t)D / 2 : (size_t)D * 2));
    page_bytes = (size_t)Hkv_l * page * (kv_row_bytes + (kvq() ? 4 : 0));
    kv_pool = (uint8_t*)dalloc<uint16_t>(((size_t)n_kv_layers * 2 * n_pages * page_bytes + 1) / 2);
    free_pages.resize((size_t)n_pages);
    for (int64_t i = 0; i < n_pages; ++i) free_pages[(size_t)i] = (int32_t)(n_pages - 1 - i);
    page_ref.assign((size_t)n_pages, 0);

    // RoPE tables exactly as RotaryEmbedding::new (modules/rotary.rs:29-46):
    // inv_freq in f64 then cast to f32; freqs = pos(f32) * inv(f32); cos/sin in f32.
    // Qwen3.5: MRotaryEmbedding::new (qwen3_5/modeling.rs:106-124) computes base^e in F32 over rot_dim.
    const int half = cfg.rot_dim / 2;
    std::vector<float> inv(half), hc((size_t)max_seq * half), hs((size_t)max_seq * half);
    for (int i = 0; i < half; ++i) {
        if (cfg.hybrid) inv[i] = 1.0f / powf((float)cfg.theta, (float)i * 2.0f / (float)cfg.rot_dim);
        else inv[i] = (float)(1.0 / std::pow(cfg.theta, (double)(2 * i) / (double)D));
    }
    for (int p = 0; p < max_seq; ++p)
        for (int i = 0; i < half; ++i) {
            const float f = (float)p * inv[i];
            hc[(size_t)p * half + i] = cosf(f);
            hs[(size_t)p * half + i] = sinf(f);
        }
    cos = dalloc<float>(hc.size());
    sin = dalloc<float>(hs.size());
    CM_HIP(hipMemcpy(cos, hc.data(), hc.size() * sizeof(float), hipMemcpyHostToDevice));
    CM_HIP(hipMemcpy(sin, hs.data(), hs.size() * sizeof(float), hipMemcpyHostToDevice));

    // CM_DEBUG_FORCE_RCCL: route the tp=1 reductions through a 1-rank RCCL communicator so the collective code path
    // can be exercised on a single GPU.
    const bool force_cc = tp == 1 && (opts.debug_flags & CM_DEBUG_FORCE_RCCL) != 0;
    if (tp > 1 || force_cc) {
        rccl.reset(new Rccl());
        UniqueId self_id;
        const void* id = opts.tp_unique_id;
        if (force_cc && !id) { Rccl::unique_id(&self_id); id = &self_id; }
        // in-process group (cm_opts.tp_mode = CM_TP_IN_PROCESS): the peer-store transport, or RCCL with the group's own id --
        // ncclCommInitRank is called by all rank threads concurrently, as RCCL requires of ranks that share a process
        // (group + RCCL: a rendezvous first -- a rank that failed while loading its shard or allocating its runtime has raised
        // PeerShared::fail by now, and every other rank leaves here with an error instead of blocking inside ncclCommInitRank,
        // which waits for all n ranks and cannot be released)
        if (peer_shared && !peer_shared->use_peer) peer_shared->arrive_and_wait();
        if (peer_shared && peer_shared->use_peer) rccl->init_peer(peer_shared, rank, num_cu, stream);
        else rccl->init(tp, rank, peer_shared ? (const void*)&peer_shared->uid : id, stream, (opts.debug_flags & CM_DEBUG_TP_LOCAL) != 0);
    }
    build_engine();
    CM_HIP(hipStreamSynchronize(stream));
}

// ------------------------------------------------------------------------------------
// persistent decode kernel (kernels_engine.hip): phase table, attention operands, granule buffers
// ------------------------------------------------------------------------------------
bool Model::engine_eligible(std::string* why) const {
    auto no = [&](const char* m) { if (why) *why = m; return false; };
    if (quantized) return no("quantised weights");
    // tensor parallelism: only ONE rank alone with the exchange stubbed (CM_DEBUG_TP_LOCAL: the all-reduce is the identity, so a
    // rank's step is a TP = 1 step at the shard's widths) -- what `bench.py --tp-local` times and tests/test_gpu_tp_shards.py checks.
    // With real peers the exchange would have to live on the kernel's hand-off edges: the launch path runs there.
    if ((tp != 1 || rccl) && !(rccl && rccl->fake)) return no("tensor parallelism");
    // ... and there it is opt-in (cm_opts.engine = 1): measured SLOWER than the launch path on every shard size -- Qwen3-8B rank 0 at
    // TP = 2 / 4 / 8: 2.19 / 1.64 / 1.67 ms against 2.07 / 1.48 / 1.22 ms.  A shard's phases are 1-10 us of streaming, so none of the
    // ~6 dependent cross-CU hand-offs of a layer (~4 us each at these sizes) hides behind the weight stream
    if (rccl && rccl->fake && opts.engine <= 0) return no("tensor-parallel shard: the persistent kernel is opt-in (cm_opts.engine = 1), slower than the launches");
    // hybrid family: the per-layer chain only (out_proj / o_proj -> gate||up -> down_proj -> next layer's in_proj / QKV around the
    // separate Gated-Delta-Net / attention launches); both kinds of token mixer must hand over a vector of the same length
    if (cfg.hybrid && Hq_l * cfg.D != cfg.value_dim()) return no("hybrid: attention output and GDN value widths differ");
    // measured slower than the launch path on Qwen3.8-27B (103 vs 110 tok/s: its GEMVs are large enough to stream at 6.1 TB/s
    // as separate launches, the chain's hand-off traffic costs more than the three boundaries it removes): opt-in only
    if (cfg.hybrid && !hybrid_engine && opts.engine <= 0) return no("hybrid per-layer chain is opt-in (cm_opts.engine = 1 or CM_ENGINE_HYBRID=1)");
    const int Ko = Hq_l * cfg.D;
    // dependency chunk (= K elements of one weight batch): 2048 where every width is a multiple of it, else 1024 (Qwen3-0.6B:
    // hidden 1024, intermediate 3072) in the default kernel configuration
    const int ch = (Ko % 1024 || cfg.H % 1024 || I_l % 1024) ? 512 : (Ko % 2048 || cfg.H % 2048 || I_l % 2048) ? 1024 : 2048;
    if (Ko % ch || cfg.H % ch || I_l % ch) return no("projection widths must be multiples of 512");
    if (cfg.H % 1024) return no("hidden size must be a multiple of 1024");
    if (!engine_has_chunk(ch)) return no("512- / 1024-element chunks only in the default kernel configuration");
    const EngCfg ec = engine_config();
    if (cfg.H / 1024 > 2 * ec.nsw || Ko / 1024 > 2 * ec.nsw) return no("input vector of the first phase too long for the stream waves");
    if (I_l / ch > 20 || Ko / ch > 20 || cfg.H / ch > 20) return no("a projection input too long for the chunk counters");
    if (cfg.H / 1024 > 16) return no("hidden size too large for the sum-of-squares slots");
    const int TW = num_cu * ec.nsw;
    if ((cfg.H / 2 + TW - 1) / TW > 4) return no("hidden size too large for the residual slots");
    return true;
}

// the whole token in one launch needs the attention inside the kernel: bf16 pages, head_dim 128, GQA group of 4, one
// workgroup per (kv head, token split) with at most 32 splits
bool Model::engine_full_eligible() const {
    if (cfg.D != 128 || !engine_has_nrep(nrep) || (kv_mode != KV_BF16 && kv_mode != KV_F16) || !cfg.qk_norm) return false;
    // (at most 32 token splits per kv head: a rank with ONE kv head -- Qwen3-8B at TP = 8 -- runs its attention on 32 of the 256 workgroups)
    const int ns = std::min(32, num_cu / std::max(1, Hkv_l)), opb = ns > 0 ? nrep * cfg.D / ns : 0;
    return ns >= 1 && ns <= 32 && (nrep * cfg.D) % ns == 0 && opb >= 2 && opb <= 16 && opb % 2 == 0 && cfg.D % opb == 0;
}

void Model::build_engine() {
    engine_on = false;
    engine_full = false;
    if (opts.engine < 0) return;
    std::string why;
    const bool ok = engine_eligible(&why);
    // default (0): on whenever the shapes allow it (CM_ENGINE=0 turns it off for A/B runs); 1: required
    if (opts.engine == 0) { const char* e = getenv("CM_ENGINE"); if (e && atoi(e) <= 0) return; }
    if (!ok) {
        if (opts.engine > 0) throw CmError(CM_ERR_UNSUPPORTED, "cm_opts.engine = 1: " + why);
        return;
    }
    const EngCfg ec = engine_config();
    const int H = cfg.H, D = cfg.D, Ko = Hq_l * D, TW = num_cu * ec.nsw;
    eng_chunk = (Ko % 1024 || H % 1024 || I_l % 1024) ? 512 : (Ko % 2048 || H % 2048 || I_l % 2048) ? 1024 : 2048;
    // (LDS input buffers and granule buffers in whole 1024-element staging passes: a 512-multiple vector's last pass is half padding)
    auto up1k = [](int v) { return (v + 1023) / 1024 * 1024; };
    const int x0 = std::max(up1k(Ko), H), x1 = H, xh = up1k(I_l);
    eng_xf_total = x0 + x1 + xh;
    auto gpw = [&](int N) { return (N / 2 + TW - 1) / TW; };
    eng_gpw_res = gpw(H);
    engine_full = !cfg.hybrid && engine_full_eligible();
    if (const char* e = getenv("CM_ENGINE_FULL")) engine_full = engine_full && atoi(e) != 0;
    if (const char* e = getenv("CM_ENGINE_FULL_MAX")) eng_full_max_ctx = atoll(e);
    // 1024-element chunks = Qwen3-0.6B-sized phases (a few MB of weights each): nothing for a poll to disturb, so inputs are
    // probed at once and without a pause (0.6B: 1365 -> 1395 tok/s; the same setting costs Qwen3-8B 13 %)
    if (eng_chunk == 1024 && !cfg.hybrid) eng_tune = 0;
    if (const char* e = getenv("CM_ENG_TUNE")) eng_tune = (int)strtol(e, nullptr, 0);
    if (const char* e = getenv("CM_ENG_DBG")) eng_dbg = atoi(e);          // kernel timing experiments, results invalid
    EngArgs probe{};
    probe.xf_total = eng_xf_total; probe.gpw_res = eng_gpw_res;
    probe.attn = engine_full ? (const EngAttnL*)1 : nullptr;
    if (engine_lds_bytes(probe, ec.nsw, ec.ncw) > 160 * 1024 - 256) {
        probe.attn = nullptr;
        engine_full = false;
        if (engine_lds_bytes(probe, ec.nsw, ec.ncw) > 160 * 1024 - 256) {
            if (opts.engine > 0) throw CmError(CM_ERR_UNSUPPORTED, "cm_opts.engine = 1: input vectors do not fit LDS");
            return;
        }
    }
    const int qkv_rows = (cfg.hybrid ? 2 * Hq_l + 2 * Hkv_l : Hq_l + 2 * Hkv_l) * D;      // (hybrid: q carries its output gate)
    int gblk_env[4] = {0, 0, 0, 0};
    if (const char* e = getenv("CM_ENG_GBLK")) sscanf(e, "%d,%d,%d,%d", &gblk_env[0], &gblk_env[1], &gblk_env[2], &gblk_env[3]);
    std::vector<EngPhase> prog((size_t)cfg.L * 4 + 1);
    std::vector<EngAttnL> at((size_t)cfg.L);
    for (int li = 0; li < cfg.L; ++li) {
        const LayerW& w = layers[(size_t)li];
        EngPhase* p = &prog[(size_t)li * 4];
        for (int k = 0; k < 4; ++k) { p[k] = EngPhase{}; p[k].layer = li; p[k].useq = li; p[k].in_tag = li + 1; p[k].out_tag = li + 1; }
        // RMSNorm + merged QKV: input = residual after the previous layer's down_proj
        p[0].W = w.full ? w.qkv : w.in_proj; p[0].nw = w.ln1; p[0].N = w.full ? qkv_rows : in_proj_rows; p[0].K = H; p[0].kind = ENG_STORE;
        p[0].xoff = 0; p[0].xbuf = 0; p[0].in_edge = ENG_E_X0; p[0].in_tag = li; p[0].out_edge = ENG_E_QKV;
        // o_proj + residual: input = attention output (the comm waves run the attention first)
        p[1].W = w.full ? w.o : w.out_proj; p[1].N = H; p[1].K = Ko; p[1].kind = ENG_RESADD;
        p[1].xoff = 0; p[1].xbuf = 3; p[1].in_edge = ENG_E_ATTN; p[1].out_edge = ENG_E_X1; p[1].pre_attn = 1;
        // RMSNorm + gate||up + SiLU*mul
        p[2].W = w.gate_up; p[2].nw = w.ln2; p[2].N = 2 * I_l; p[2].K = H; p[2].kind = ENG_SILUMUL;
        p[2].xoff = x0; p[2].xbuf = 1; p[2].in_edge = ENG_E_X1; p[2].out_edge = ENG_E_H;
        // down_proj + residual
        p[3].W = w.down; p[3].N = H; p[3].K = I_l; p[3].kind = ENG_RESADD;
        p[3].xoff = x0 + x1; p[3].xbuf = 2; p[3].in_edge = ENG_E_H; p[3].out_edge = ENG_E_X0;
        for (int k = 0; k < 4; ++k) {
            p[k].gpw = gpw(p[k].N); p[k].nb = p[k].K / eng_chunk;
            // row groups kept open (walked chunk-major): a consumer with few chunks needs several open groups so that the
            // LAST chunk of its input is needed late; a producer should close groups early so that the first chunks of
            // its output exist early.  Long rows (down_proj) go group by group, short rows three groups at a time.
            p[k].gblk = p[k].nb > 2 ? 1 : std::min(p[k].gpw, 3);
            // QKV with the attention inside the launch: the q rounds form the first block (chunk-major among themselves), the k / v
            // rows follow -- q is then complete a third of the phase earlier and the attention starts on it (8B: 0.7165 -> 0.72)
            if (k == 0 && engine_full && p[0].nb <= 2) p[0].gblk = std::max(1, std::min(std::min(p[0].gpw, 3), (Hq_l * D / 2 + TW - 1) / TW));
            if (gblk_env[k] > 0) p[k].gblk = std::min(std::min(p[k].gpw, 6), gblk_env[k]);      // CM_ENG_GBLK (tuning)
        }
        {
            // pre_attn of the o_proj phase = batches per stream wave of the QKV phase after which every q row is complete (the
            // in-kernel attention starts on q and the old tokens there, the k / v rows -- the last row groups -- follow): walk the
            // phase in the kernel's order (blocks of gblk row groups, chunk-major inside a block)
            const int q_rounds = std::min(p[0].gpw, (Hq_l * D / 2 + TW - 1) / TW);
            int qb = 0, batch = 0;
            for (int gb = 0; gb * p[0].gblk < p[0].gpw; ++gb) {
                const int cnt = std::min(p[0].gblk, p[0].gpw - gb * p[0].gblk);
                for (int kb = 0; kb < p[0].nb; ++kb)
                    for (int gg = 0; gg < cnt; ++gg) {
                        ++batch;
                        if (kb == p[0].nb - 1 && gb * p[0].gblk + gg < q_rounds) qb = batch;
                    }
            }
            p[1].pre_attn = std::max(1, qb);
        }
        at[(size_t)li] = EngAttnL{kpool(li), vpool(li), w.qn, w.kn};
    }
    {
        // the whole token's LAST phase (round 5): final RMSNorm + lm_head as one more row-streaming projection -- its input is the
        // residual the last down_proj publishes, its rows go to `logits` as plain stores, every stream wave keeps the arg-max of
        // the rows it produced; the embedding row is read by the launch itself (EngArgs::embed).  A token is then TWO launches
        // (this + argmax_final) instead of four: the 1.24 GB table streams behind layer 36 without a launch boundary.
        EngPhase& h = prog[(size_t)cfg.L * 4];
        h = EngPhase{};
        h.layer = cfg.L; h.useq = cfg.L; h.in_tag = cfg.L; h.out_tag = cfg.L + 1;
        h.W = lm_head; h.nw = norm; h.N = std::max(0, std::min(V_l, cfg.V - v0)); h.K = H; h.kind = ENG_STORE;
        h.xoff = 0; h.xbuf = 0; h.in_edge = ENG_E_X0; h.out_edge = -1;
        h.gpw = gpw(h.N); h.nb = h.K / eng_chunk; h.gblk = h.nb > 2 ? 1 : std::min(h.gpw, 3);
        // (2048-element chunks only: at the Qwen3-0.6B widths -- 1024-wide rows, 4 KB batches -- the separate head GEMV is faster:
        // 1385 vs 1342 tok/s; Qwen3-8B 367 -> 370, Qwen3-VL-2B 995 -> 1005 with the head inside)
        eng_head = engine_full && eng_chunk == 2048 && !quantized && (!rccl || rccl->fake) && embed != nullptr && lm_head != nullptr && h.N % 2 == 0 && h.N > 0;
        if (const char* e = getenv("CM_ENG_HEAD")) eng_head = eng_head && atoi(e) != 0;
        if (eng_head) { eng_pmax = dalloc<float>((size_t)TW); eng_pidx = dalloc<int>((size_t)TW); }
    }
    eng_prog = (EngPhase*)dalloc<int>(prog.size() * sizeof(EngPhase) / sizeof(int));
    CM_HIP(hipMemcpy(eng_prog, prog.data(), prog.size() * sizeof(EngPhase), hipMemcpyHostToDevice));
    eng_attn = (EngAttnL*)dalloc<int>(at.size() * sizeof(EngAttnL) / sizeof(int));
    CM_HIP(hipMemcpy(eng_attn, at.data(), at.size() * sizeof(EngAttnL), hipMemcpyHostToDevice));
    const int ns = std::max(1, num_cu / std::max(1, Hkv_l));
    const size_t gsz[ENG_NEDGE] = {(size_t)H, (size_t)std::max(qkv_rows, cfg.hybrid ? in_proj_rows : 0), (size_t)Hkv_l * ns * nrep * (D + 2), (size_t)up1k(Hq_l * D), (size_t)H, (size_t)up1k(I_l)};
    for (int e = 0; e < ENG_NEDGE; ++e) {
        eng_gsz[e] = gsz[e];
        eng_gran[e] = (unsigned long long*)dalloc<int>(gsz[e] * 2);
        CM_HIP(hipMemset(eng_gran[e], 0, gsz[e] * 8));       // tag 0 is never a valid epoch
    }
    // epoch base 1: every tag of the first launch is >= 2, the zero-filled granules never match
    const uint32_t one = 1;
    CM_HIP(hipMemcpy(&st->rsv[1], &one, 4, hipMemcpyHostToDevice));
    if (!engine_prepare(engine_lds_bytes(probe, ec.nsw, ec.ncw), engine_full ? nrep : 4, eng_chunk)) {
        if (opts.engine > 0) throw CmError(CM_ERR_DEVICE, "cm_opts.engine = 1: kernel attribute");
        return;
    }
    engine_on = true;
    engine_capable = true;
    engine_full_capable = engine_full;
}

void Model::drop_graphs() {
    if (stream) (void)hipStreamSynchronize(stream);
    for (int v = 0; v < 5; ++v) {
        if (graph_exec[v]) { (void)hipGraphExecDestroy(graph_exec[v]); graph_exec[v] = nullptr; }
        if (graph[v]) { (void)hipGraphDestroy(graph[v]); graph[v] = nullptr; }
        graph_ok[v] = false;
    }
}

EngArgs Model::engine_args_common() const {
    EngArgs e{};
    e.prog = eng_prog;
    for (int k = 0; k < ENG_NEDGE; ++k) e.gran[k] = eng_gran[k];
    e.xres = x; e.ctl = (uint32_t*)&st->rsv[1];
    e.st = st; e.block_table = d_bt; e.cos = cos; e.sin = sin;
    e.epoch_step = cfg.L + 3;
    e.H = cfg.H; e.gpw_res = eng_gpw_res; e.xf_total = eng_xf_total;
    e.Hkv = Hkv_l; e.page = page; e.max_pages = max_pages_per_seq;
    e.q_off = 0; e.k_off = Hq_l * cfg.D; e.v_off = e.k_off + Hkv_l * cfg.D;
    e.eps = cfg.eps; e.scale = (float)(1.0 / std::sqrt((double)cfg.D));
    e.kv_f16 = kv_mode == KV_F16 ? 1 : 0;
    e.nrep = nrep; e.chunk = eng_chunk;
    e.dbg = eng_dbg; e.tune = eng_tune;
    return e;
}

// per-layer launch: o_proj -> gate||up -> down_proj -> QKV of the next layer (attention stays a separate launch)
EngArgs Model::engine_args(int li) const {
    EngArgs e = engine_args_common();
    e.p0 = 4 * li + 1; e.p1 = std::min(4 * li + 5, 4 * cfg.L);
    e.plain_last = li + 1 < cfg.L ? 1 : 0;
    e.vin = attn; e.vout = qkv;
    e.ub0 = li + 1; e.ub1 = li; e.ub2 = li; e.ub3 = li;
    return e;
}

// whole token: every projection and the attention of every layer in one launch
EngArgs Model::engine_args_full() const {
    EngArgs e = engine_args_common();
    e.p0 = 0; e.p1 = 4 * cfg.L;
    e.attn = eng_attn;
    e.vin = x;
    if (eng_head) {            // + embedding row in, final norm + lm_head + arg-max partials out
        e.p1 = 4 * cfg.L + 1; e.plain_last = 1; e.vout = logits;
        e.vout = logits + (size_t)rank * V_l; e.idx_base = v0;
        e.embed = embed; e.embed_V = cfg.V; e.pmax = eng_pmax; e.pidx = eng_pidx;
    }
    return e;
}

// cm_debug_read("engine_trace"): a few warm launches, then ONE launch of the instrumented instantiation;
// out[((block * waves + wave) * ENG_TRACE_PH + phase) * ENG_TRACE_EV + event] = microseconds since the earliest stamp (0 = not recorded;
// event 3 of a comm wave = number of granule sweeps that found stale tags)
void Model::engine_trace(float* out, size_t n) {
    if (!engine_on) throw CmError(CM_ERR_INVALID, "the persistent decode kernel is not active on this model");
    const EngCfg ec = engine_config();
    const size_t waves = (size_t)(ec.nsw + ec.ncw), total = (size_t)num_cu * waves * ENG_TRACE_PH * ENG_TRACE_EV;
    if (n != total) throw CmError(CM_ERR_RANGE, "engine_trace: expected " + std::to_string(total) + " values");
    unsigned long long* d = nullptr;
    CM_HIP(hipMalloc((void**)&d, total * 8));
    CM_HIP(hipMemsetAsync(d, 0, total * 8, stream));
    const int layers_w = std::max(1, cfg.L - 1);
    for (int i = 0; i < 6; ++i) launch_engine(engine_full ? engine_args_full() : engine_args(i % layers_w), num_cu, stream);
    EngArgs e = engine_full ? engine_args_full() : engine_args(6 % layers_w);
    e.trace = d;
    launch_engine(e, num_cu, stream, true);
    std::vector<unsigned long long> h(total);
    CM_HIP(hipStreamSynchronize(stream));
    CM_HIP(hipMemcpy(h.data(), d, total * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    auto is_count = [&](size_t i) { return ((i % ENG_TRACE_EV) == 3 || (i % ENG_TRACE_EV) == 7) && ((i / (ENG_TRACE_EV * ENG_TRACE_PH)) % waves) >= (size_t)ec.nsw; };
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < total; ++i)
        if (!is_count(i) && h[i] != 0 && h[i] < t0) t0 = h[i];
    for (size_t i = 0; i < total; ++i) {
        if (is_count(i)) out[i] = (float)h[i];
        else out[i] = h[i] == 0 ? 0.f : (float)((double)(h[i] - t0) * 0.01) + 0.01f;    // 100 MHz -> us
    }
    engine_check();
}

// The persistent kernel spin-waits across all of its workgroups, so every one of them must be resident at the same time: one
// workgroup per CU on an otherwise idle device (see crane_mi355.h).  A second tenant -- another handle's kernels, another
// process, a profiler's serialisation -- can break that; the bounded spins then raise the error word and the launch ends with
// garbage in the residual stream and in the K/V rows of the step.  That must not kill the handle: the word is cleared, the
// persistent kernel switched off for this handle (the per-projection launch path takes over), the captured graphs dropped,
// and the caller replays the step(s) it had enqueued -- a replay rewrites the same K/V rows (dense family only: a hybrid
// model's recurrent state was advanced by the failed step, so its callers raise instead -- hybrid_engine_abort).
bool Model::engine_failed() {
    if (!engine_on) return false;
    CM_HIP(hipMemcpyAsync(h_st, st, sizeof(StepState), hipMemcpyDeviceToHost, stream));
    CM_HIP(hipStreamSynchronize(stream));
    if (h_st->rsv[2] == 0) return false;
    eng_fail_code = (uint32_t)h_st->rsv[2];
    const int32_t zero = 0;
    CM_HIP(hipMemcpy(&st->rsv[2], &zero, 4, hipMemcpyHostToDevice));
    engine_on = false;
    drop_graphs();
    return true;
}

// Hybrid (Gated-Delta-Net) handles on the opt-in persistent chain: a timed-out launch has advanced the recurrent state, so
// the step cannot be replayed; switch the chain off, and tell the caller the sequence state is gone.
void Model::hybrid_engine_abort() {
    (void)engine_failed();
    char b[256];
    snprintf(b, sizeof b, "persistent decode kernel timed out (code 0x%x) on a hybrid model: the Gated-Delta-Net state of the "
             "sequence was advanced by the failed step and cannot be replayed; truncate the sequence to 0 and re-prefill "
             "(the persistent path is now off for this handle)", (unsigned)eng_fail_code);
    throw CmError(CM_ERR_DEVICE, b);
}

void Model::engine_check() {
    if (!engine_failed()) return;
    char b[192];
    snprintf(b, sizeof b, "persistent decode kernel timed out (code 0x%x): its workgroups were not co-resident; it is now switched "
             "off for this handle (per-projection launches), repeat the call", (unsigned)eng_fail_code);
    throw CmError(CM_ERR_DEVICE, b);
}

// ------------------------------------------------------------------------------------
// paged KV allocator + sequences
// ------------------------------------------------------------------------------------
Seq& Model::seq(int s) {
    if (s < 0 || s >= (int)seqs.size() || !seqs[(size_t)s].used) throw CmError(CM_ERR_INVALID, "invalid sequence handle");
    return seqs[(size_t)s];
}

int Model::seq_alloc() {
    for (size_t i = 1; i < seqs.size(); ++i)
        if (!seqs[i].used) { seqs[i] = Seq(); seqs[i].used = true; return (int)i; }
    throw CmError(CM_ERR_OOM, "no free sequence slot (raise cm_opts.max_seqs)");
}

void Model::seq_truncate(int s, size_t new_len) {
    Seq& q = seq(s);
    if ((int64_t)new_len > q.len) throw CmError(CM_ERR_RANGE, "truncate beyond cached length");
    if (gdn_layers > 0 && (int64_t)new_len != q.len) {
        // a recurrent (GDN) state cannot be rewound: only "clear" is possible, like the reference,
        // which re-prefills from scratch after preemption (engine/mod.rs:430-504)
        if (new_len != 0) throw CmError(CM_ERR_UNSUPPORTED, "Gated-Delta-Net state cannot be truncated to a non-zero length");
        reset_gdn_state(s);
    }
    // a sequence cut back INTO an image prompt would rotate its next tokens at cache position + rope_delta < 0
    if (new_len != 0 && (int64_t)new_len + (int64_t)q.rope_delta < 0)
        throw CmError(CM_ERR_RANGE, "truncate into an image prompt: rotary position would be negative (truncate to 0 instead)");
    const size_t keep = (new_len + page - 1) / page;
    while (q.pages.size() > keep) {
        const int32_t p = q.pages.back();
        q.pages.pop_back();
        if (--page_ref[(size_t)p] == 0) free_pages.push_back(p);
    }
    q.len = (int64_t)new_len;
    if (new_len == 0) q.rope_delta = 0;
    if (active_seq == s) active_pages_uploaded = std::min(active_pages_uploaded, q.pages.size());
}

void Model::reset_gdn_state(int slot) {
    if (gdn_layers == 0) return;
    CM_HIP(hipMemsetAsync(conv_pool + conv_slot_elems * (size_t)slot, 0, conv_slot_elems * sizeof(float), stream));
    CM_HIP(hipMemsetAsync(state_pool + state_slot_elems * (size_t)slot, 0, state_slot_elems * sizeof(float), stream));
}

void Model::seq_free(int s) {
    if (s == 0) { seq_truncate(0, 0); return; }
    seq_truncate(s, 0);
    seqs[(size_t)s].used = false;
    if (active_seq == s) active_seq = -1;
}

int Model::seq_fork(int src) {
    Seq& a = seq(src);
    const int d = seq_alloc();
    Seq& b = seqs[(size_t)d];
    b.len = a.len;
    b.rope_delta = a.rope_delta;
    b.pages = a.pages;
    for (int32_t p : b.pages) page_ref[(size_t)p]++;
    if (gdn_layers > 0) {      // recurrent + conv state travel with the sequence
        CM_HIP(hipMemcpyAsync(conv_pool + conv_slot_elems * (size_t)d, conv_pool + conv_slot_elems * (size_t)src,
                              conv_slot_elems * sizeof(float), hipMemcpyDeviceToDevice, stream));
        CM_HIP(hipMemcpyAsync(state_pool + state_slot_elems * (size_t)d, state_pool + state_slot_elems * (size_t)src,
                              state_slot_elems * sizeof(float), hipMemcpyDeviceToDevice, stream));
    }
    // copy-on-write of the last, partially filled page: both forks will append into it
    if (!b.pages.empty() && (a.len % page) != 0) {
        if (free_pages.empty()) { seq_free(d); throw CmError(CM_ERR_OOM, "KV pool exhausted"); }
        cow_page(d, b.pages.size() - 1);
    }
    return d;
}

// give `q` a private copy of its page `idx` (all layers, K and V) when the page is shared with a fork
void Model::cow_page(int s, size_t idx) {
    Seq& q = seqs[(size_t)s];
    const int32_t oldp = q.pages[idx];
    if (page_ref[(size_t)oldp] <= 1) return;
    if (free_pages.empty()) throw CmError(CM_ERR_OOM, "KV pool exhausted");
    const int32_t newp = free_pages.back();
    free_pages.pop_back();
    page_ref[(size_t)newp] = 1;
    page_ref[(size_t)oldp]--;
    q.pages[idx] = newp;
    const size_t pitch = (size_t)n_pages * page_bytes;
    CM_HIP(hipMemcpy2DAsync(kv_pool + (size_t)newp * page_bytes, pitch, kv_pool + (size_t)oldp * page_bytes, pitch,
                            page_bytes, (size_t)n_kv_layers * 2, hipMemcpyDeviceToDevice, stream));
    if (active_seq == s) active_pages_uploaded = std::min(active_pages_uploaded, idx);    // re-upload the table entry
}

void Model::ensure_pages(int s, int64_t upto_len) {
    Seq& q = seq(s);
    if (upto_len > max_seq) throw CmError(CM_ERR_RANGE, "sequence longer than max_seq_len");
    const size_t need = (size_t)((upto_len + page - 1) / page);
    // A page shared with a fork must not be appended into.  seq_fork copies the partially filled last page, but a fork (or
    // its parent) can later be cut back INTO a fully shared page (cm_seq_truncate, or forward() with start_pos < len): the
    // page that holds the first position about to be written is made private here.
    if (q.len < upto_len && (q.len % page) != 0) {
        const size_t idx = (size_t)(q.len / page);
        if (idx < q.pages.size()) cow_page(s, idx);
    }
    while (q.pages.size() < need) {
        if (free_pages.empty()) throw CmError(CM_ERR_OOM, "KV pool exhausted");
        const int32_t p = free_pages.back();
        free_pages.pop_back();
        page_ref[(size_t)p] = 1;
        q.pages.push_back(p);
    }
}

void Model::activate(int s) {
    Seq& q = seq(s);
    if (active_seq != s) {
        CM_HIP(hipStreamSynchronize(stream));    // pinned mirror is about to be rewritten
        active_seq = s;
        active_pages_uploaded = 0;
    }
    if (active_pages_uploaded < q.pages.size()) {
        const size_t a = active_pages_uploaded, n = q.pages.size() - a;
        memcpy(h_bt + a, q.pages.data() + a, n * sizeof(int32_t));
        CM_HIP(hipMemcpyAsync(d_bt + a, h_bt + a, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        active_pages_uploaded = q.pages.size();
    }
}

uint64_t Model::kv_bytes() const {
    uint64_t pages_used = 0;
    for (auto r : page_ref) if (r > 0) ++pages_used;
    return pages_used * page_bytes * 2ull * (uint64_t)n_kv_layers;
}

uint64_t Model::decode_bytes_per_token(size_t ctx) const {
    // 2 B x every weight element once + KV read at ctx (+ GDN state read+write), this rank
    const uint64_t H = cfg.H, D = cfg.D;
    const uint64_t mlp = 3ull * I_l * H + 2 * H;
    uint64_t w_elems = 0, extra = 0;
    for (int i = 0; i < cfg.L; ++i) {
        if (cfg.layer_full(i)) {
            const uint64_t qrows = (uint64_t)(cfg.hybrid ? 2 * Hq_l : Hq_l) * D;
            w_elems += (qrows + 2ull * Hkv_l * D) * H + H * (uint64_t)Hq_l * D + (cfg.qk_norm ? 2 * D : 0) + mlp;
            extra += 2ull * Hkv_l * ctx * (kv_row_bytes + (kvq() ? 4 : 0));
        } else {
            w_elems += (uint64_t)in_proj_rows * H + H * (uint64_t)cfg.value_dim() + mlp;
            extra += 2ull * cfg.NV * cfg.Kd * cfg.Vd * 4 + 2ull * cfg.conv_dim() * (cfg.conv_k - 1) * 4 +
                     (uint64_t)cfg.conv_dim() * cfg.conv_k * 4;
        }
    }
    const uint64_t v_eff = (uint64_t)std::max(0, std::min(V_l, cfg.V - v0));
    w_elems += v_eff * H + H /*final norm*/ + H /*embedding row*/;
    if (quantized) {
        // every quantised matrix once (codes + scales) + f32 norm vectors + one embedding row + KV
        uint64_t b = quant_weight_bytes + (uint64_t)cfg.L * (2 * H + (cfg.qk_norm ? 2 * D : 0)) * 4 + H * 4;
        if (q_lm_head.fmt == QFMT_NONE) b += v_eff * H * 2;           // tied bf16 table used as lm_head (ISQ)
        b += q_embed.fmt != QFMT_NONE ? q_embed.bytes() / (uint64_t)cfg.V : H * 2;
        return b + extra;
    }
    return w_elems * 2 + extra;
}

// ------------------------------------------------------------------------------------
// decode step
// ------------------------------------------------------------------------------------
void Model::enqueue_decode_step(bool advance) {
    const int H = cfg.H, D = cfg.D;
    hipStream_t s = stream;
    if (engine_on && attn_variant == 4 && eng_head) {
        // the whole token in ONE launch: embedding row, every projection and the attention of every layer, final norm + lm_head with
        // per-wave arg-max partials (kernels_engine.hip); argmax_final picks the winner and advances the step state
        if (!launch_engine(engine_args_full(), num_cu, s)) throw CmError(CM_ERR_DEVICE, "persistent decode kernel launch");
        logits_gathered = false;
        launch_argmax_final(eng_pmax, eng_pidx, num_cu * engine_config().nsw, st, ring, RING - 1, advance ? 1 : 0, 1, s);
        return;
    }
    if (quantized && q_embed.fmt != QFMT_NONE) launch_embed_row_q(q_embed, st, x, H, cfg.V, s);
    else launch_embed_row(embed, st, x, H, cfg.V, 1, s);
    const int qkv_rows = (cfg.hybrid ? 2 * Hq_l + 2 * Hkv_l : Hq_l + 2 * Hkv_l) * D;
    if (engine_on && attn_variant == 4) {
        // the whole token in ONE launch: every projection and the attention of every layer (kernels_engine.hip)
        if (!launch_engine(engine_args_full(), num_cu, s)) throw CmError(CM_ERR_DEVICE, "persistent decode kernel launch");
        enqueue_lm_head(advance);
        return;
    }
    if (engine_on) {
        // per-layer persistent launches: QKV of layer 0 as a plain launch, then per layer the attention kernels + ONE launch
        // for o_proj -> gate||up -> down_proj -> QKV of the next layer
        // (hybrid family: the first projection is in_proj of a Gated-Delta-Net layer, the token mixer between two chain launches
        // is the GDN step -- with its own gated norm: the chain's out_proj stages a plain vector -- or the gated attention)
        GemvArgs g{};
        g.W = layers[0].full ? layers[0].qkv : layers[0].in_proj; g.x = x; g.nw = layers[0].ln1; g.y = qkv;
        g.N = layers[0].full ? qkv_rows : in_proj_rows; g.K = H; g.ldw = H; g.eps = cfg.eps;
        launch_gemv(PRO_RMSNORM, EPI_STORE, g, gemv_grid(g.N, g.K, num_cu), s);
        for (int li = 0; li < cfg.L; ++li) {
            const LayerW& w = layers[(size_t)li];
            if (!w.full) {
                GdnArgs ga{};
                ga.proj = qkv; ga.conv_w = w.conv_w; ga.conv_pool = conv_pool; ga.state_pool = state_pool;
                ga.A_log = w.A_log; ga.dt_bias = w.dt_bias; ga.gnorm_w = w.gnorm; ga.out = attn; ga.st = st;
                ga.proj_stride = in_proj_pad; ga.out_stride = cfg.value_dim(); ga.S = 1; ga.NV = cfg.NV; ga.vpg = cfg.NV / cfg.NK; ga.chunked = gdn_chunked ? 1 : 0;
                ga.key_dim = cfg.key_dim(); ga.layer_idx = w.gdn_idx; ga.gdn_layers = gdn_layers; ga.eps = cfg.eps; ga.n_seq = 1;
                ga.gdn_scratch = gdn_scratch; ga.gdn_ticket = gdn_ticket;
                launch_gdn(ga, s);
                if (!launch_engine(engine_args(li), num_cu, s)) throw CmError(CM_ERR_DEVICE, "persistent decode kernel launch");
                continue;
            }
            AttnDecArgs a{};
            a.qkv = qkv; a.qnw = w.qn; a.knw = w.kn; a.cos = cos; a.sin = sin; a.st = st; a.block_table = d_bt;
            a.kpool = kpool(li); a.vpool = vpool(li); a.part_o = part_o; a.part_ml = part_ml;
            a.q_off = 0; a.k_off = (cfg.hybrid ? 2 * Hq_l : Hq_l) * D; a.v_off = a.k_off + Hkv_l * D;
            a.gate = cfg.hybrid ? qkv + (size_t)Hq_l * D : nullptr; a.rot_dim = cfg.rot_dim;
            a.Hkv = Hkv_l; a.page = page; a.max_pages = max_pages_per_seq; a.page_bytes = page_bytes; a.eps = cfg.eps;
            a.scale = (float)(1.0 / std::sqrt((double)D));
            if (attn_variant >= 2) {
                if (!launch_attn_decode_mfma(a, D, nrep, attn_variant == 3 ? nsplit_mfma : nsplit, kv_mode, attn, 0, 1, s)) throw CmError(CM_ERR_UNSUPPORTED, "GQA group size / head_dim");
            } else if (!launch_attn_decode(a, D, nrep, attn_splits_force ? attn_splits_force : nsplit, kv_mode, attn, 0, 1, s)) throw CmError(CM_ERR_UNSUPPORTED, "GQA group size / head_dim");
            if (!launch_engine(engine_args(li), num_cu, s)) throw CmError(CM_ERR_DEVICE, "persistent decode kernel launch");
        }
        enqueue_lm_head(advance);
        return;
    }
    for (int li = 0; li < cfg.L; ++li) {
        const LayerW& w = layers[(size_t)li];
        if (quantized) { enqueue_quant_layer(li); continue; }
        GemvArgs g{};
        if (!w.full) {
            // ---- Gated Delta Net layer (ops/gdn/layer.rs:122-182): in_proj GEMV, fused GDN kernel, out_proj GEMV ----
            g.W = w.in_proj; g.x = x; g.nw = w.ln1; g.y = qkv; g.N = in_proj_rows; g.K = H; g.ldw = H; g.eps = cfg.eps;
            launch_gemv(PRO_RMSNORM, EPI_STORE, g, gemv_grid(g.N, g.K, num_cu), s);
            GdnArgs ga{};
            ga.proj = qkv; ga.conv_w = w.conv_w; ga.conv_pool = conv_pool; ga.state_pool = state_pool;
            ga.A_log = w.A_log; ga.dt_bias = w.dt_bias; ga.gnorm_w = w.gnorm; ga.out = attn; ga.st = st;
            ga.proj_stride = in_proj_pad; ga.out_stride = cfg.value_dim(); ga.S = 1; ga.NV = cfg.NV; ga.vpg = cfg.NV / cfg.NK; ga.chunked = gdn_chunked ? 1 : 0;
            ga.key_dim = cfg.key_dim(); ga.layer_idx = w.gdn_idx; ga.gdn_layers = gdn_layers; ga.eps = cfg.eps; ga.n_seq = 1;
            ga.gdn_scratch = gdn_scratch; ga.gdn_ticket = gdn_ticket;
            // the gated RMSNorm of the value heads moves into out_proj's prologue: the GDN step then ends at its raw y, without
            // the ticket + read-back hand-off between the four workgroups of a head (gdn_defer_norm; needs the 4-workgroup step)
            // (every GEMV workgroup repeats the norm of all heads while staging: a win for 16 heads -- Qwen3.5-0.8B 1349 -> 1394
            // tok/s --, a loss for the 48 heads of Qwen3.8-27B, 110.2 -> 108.8: there the step keeps its own norm)
            const bool defer = gdn_defer_norm && gdn_scratch != nullptr && cfg.value_dim() % 128 == 0 && cfg.value_dim() <= gdn_defer_max;
            ga.defer_norm = defer ? 1 : 0;
            launch_gdn(ga, s);
            g = GemvArgs{};
            g.W = w.out_proj; g.x = attn; g.N = H; g.K = cfg.value_dim(); g.ldw = g.K; g.eps = cfg.eps;
            g.gdn_z = qkv + (2 * cfg.key_dim() + cfg.value_dim()); g.gdn_w = w.gnorm;
            const int opro_g = defer ? PRO_GDNNORM : PRO_PLAIN;
            if (!rccl) { g.y = x; g.res = x; launch_gemv(opro_g, EPI_RESADD, g, gemv_grid(g.N, g.K, num_cu), s); }
            else {
                g.y = y; g.res = x;
                launch_gemv(opro_g, (rank == 0 || rccl->fake) ? EPI_RESADD : EPI_STORE, g, gemv_grid(g.N, g.K, num_cu), s);
                rccl->all_reduce_sum_f32(y, x, (size_t)H, s);
            }
        } else {
        // (1) RMSNorm + merged QKV projection
        g.W = w.qkv; g.x = x; g.nw = w.ln1; g.y = qkv; g.N = qkv_rows; g.K = H; g.ldw = H; g.eps = cfg.eps;
        launch_gemv(PRO_RMSNORM, EPI_STORE, g, gemv_grid(g.N, g.K, num_cu), s);
        // (2) QK-norm + RoPE + KV append + paged split-KV attention (+ sigmoid output gate for Qwen3.5)
        AttnDecArgs a{};
        a.qkv = qkv; a.qnw = w.qn; a.knw = w.kn; a.cos = cos; a.sin = sin; a.st = st; a.block_table = d_bt;
        a.kpool = kpool(li); a.vpool = vpool(li); a.part_o = part_o; a.part_ml = part_ml;
        a.q_off = 0;
        a.k_off = (cfg.hybrid ? 2 * Hq_l : Hq_l) * D; a.v_off = a.k_off + Hkv_l * D;
        a.gate = cfg.hybrid ? qkv + (size_t)Hq_l * D : nullptr;
        a.rot_dim = cfg.rot_dim;
        a.Hkv = Hkv_l; a.page = page; a.max_pages = max_pages_per_seq; a.page_bytes = page_bytes; a.eps = cfg.eps; a.scale = (float)(1.0 / std::sqrt((double)D));
        const bool heads = attn_variant == 1 && (kv_mode == KV_BF16 || kv_mode == KV_F32);      // short context: per-head blocks, merge fused into o_proj's prologue
        if (heads) {
            if (!launch_attn_decode_heads(a, D, nrep, attn_ns, kv_f32, 1, s)) throw CmError(CM_ERR_UNSUPPORTED, "head_dim");
        } else if (attn_variant >= 2) {      // bf16 KV: matrix-core flash-decode (32 token splits, 64 for long contexts)
            if (!launch_attn_decode_mfma(a, D, nrep, attn_variant == 3 ? nsplit_mfma : nsplit, kv_mode, attn, 0, 1, s)) throw CmError(CM_ERR_UNSUPPORTED, "GQA group size / head_dim");
        } else if (!launch_attn_decode(a, D, nrep, attn_splits_force ? attn_splits_force : nsplit, kv_mode, attn, 0, 1, s)) throw CmError(CM_ERR_UNSUPPORTED, "GQA group size / head_dim");
        // (3) o_proj + residual
        g = GemvArgs{};
        g.W = w.o; g.x = attn; g.N = H; g.K = Hq_l * D; g.ldw = g.K;
        const int opro = heads ? PRO_ATTNCOMB : PRO_PLAIN;
        if (heads) { g.x = part_o; g.part_ml = part_ml; g.gate = a.gate; g.ns = attn_ns; g.dshift = D == 128 ? 7 : 8; }
        if (!rccl) { g.y = x; g.res = x; launch_gemv(opro, EPI_RESADD, g, gemv_grid(g.N, g.K, num_cu), s); }
        else {
            g.y = y; g.res = x;
            launch_gemv(opro, (rank == 0 || rccl->fake) ? EPI_RESADD : EPI_STORE, g, gemv_grid(g.N, g.K, num_cu), s);
            rccl->all_reduce_sum_f32(y, x, (size_t)H, s);
        }
        }   // full-attention layer
        // (4) RMSNorm + gate||up + SiLU*mul
        g = GemvArgs{};
        g.W = w.gate_up; g.x = x; g.nw = w.ln2; g.y = hbuf; g.N = 2 * I_l; g.K = H; g.ldw = H; g.eps = cfg.eps;
        launch_gemv(PRO_RMSNORM, EPI_SILUMUL, g, gemv_grid(g.N, g.K, num_cu), s);
        // (5) down_proj + residual
        g = GemvArgs{};
        g.W = w.down; g.x = hbuf; g.N = H; g.K = I_l; g.ldw = I_l;
        if (!rccl) { g.y = x; g.res = x; launch_gemv(PRO_PLAIN, EPI_RESADD, g, gemv_grid(g.N, g.K, num_cu), s); }
        else {
            g.y = y; g.res = x;
            launch_gemv(PRO_PLAIN, (rank == 0 || rccl->fake) ? EPI_RESADD : EPI_STORE, g, gemv_grid(g.N, g.K, num_cu), s);
            rccl->all_reduce_sum_f32(y, x, (size_t)H, s);
        }
    }
    enqueue_lm_head(advance);
}

// one decoder layer over quantised weights (LinearLayer::Quantized, ops/linear.rs:18-51): dense Qwen3, or the hybrid
// family's gated full-attention / Gated-Delta-Net layers
void Model::enqueue_quant_layer(int li) {
    const LayerW& w = layers[(size_t)li];
    const int D = cfg.D, H = cfg.H;
    hipStream_t s = stream;
    auto qg = [&](int pro, int epi, const QWeight& qw, const float* xin, const float* nw, float* yout, const float* res) {
        GemvQArgs q{};
        q.w = qw; q.x = xin; q.nw = nw; q.y = yout; q.res = res; q.eps = cfg.eps; q.act_int = quant_act_int ? 1 : 0;
        if (!launch_gemvq(pro, epi, q, gemvq_grid(qw.N, num_cu, qw.fmt), s)) throw CmError(CM_ERR_UNSUPPORTED, "quantised weight format");
    };
    if (!w.full) {
        const int qz = cfg.conv_dim() + cfg.value_dim();
        qg(PRO_RMSNORM, EPI_STORE, w.q_in_proj, x, w.ln1, qkv, nullptr);
        if (w.q_in_proj_z.fmt != QFMT_NONE) qg(PRO_RMSNORM, EPI_STORE, w.q_in_proj_z, x, w.ln1, qkv + w.q_in_proj.N, nullptr);
        // the a / b gate rows stay bf16: the Gated-Delta-Net step computes its head's two dot products itself (GdnArgs::ba_w; round 6: one
        // launch less per layer; cm_debug_set("gdn_ba_fused", 0): the GEMV launch of rounds 2-5)
        const bool ba_fused = gdn_ba_fused && gdn_scratch != nullptr && cfg.Kd == 128 && cfg.Vd == 128 && H % 4 == 0;
        if (!ba_fused) {
            GemvArgs g{};
            g.W = w.in_proj_ba; g.x = x; g.nw = w.ln1; g.y = qkv + qz; g.N = 2 * cfg.NV; g.K = H; g.ldw = H; g.eps = cfg.eps;
            launch_gemv(PRO_RMSNORM, EPI_STORE, g, gemv_grid(g.N, g.K, num_cu), s);
        }
        GdnArgs ga{};
        if (ba_fused) { ga.ba_w = w.in_proj_ba; ga.ba_x = x; ga.ba_nw = w.ln1; ga.ba_H = H; }
        ga.proj = qkv; ga.conv_w = w.conv_w; ga.conv_pool = conv_pool; ga.state_pool = state_pool;
        ga.A_log = w.A_log; ga.dt_bias = w.dt_bias; ga.gnorm_w = w.gnorm; ga.out = attn; ga.st = st;
        ga.proj_stride = in_proj_pad; ga.out_stride = cfg.value_dim(); ga.S = 1; ga.NV = cfg.NV; ga.vpg = cfg.NV / cfg.NK; ga.chunked = gdn_chunked ? 1 : 0;
        ga.key_dim = cfg.key_dim(); ga.layer_idx = w.gdn_idx; ga.gdn_layers = gdn_layers; ga.eps = cfg.eps; ga.n_seq = 1;
            ga.gdn_scratch = gdn_scratch; ga.gdn_ticket = gdn_ticket;
        // (round 6) Q8_0-layout out_proj in the integer-dot mode: the head's gated RMSNorm runs in the GEMV's prologue, in front of its row
        // quantiser (PRO_GDNNORM, as on the bf16 path) instead of behind a ticket hand-off between the four workgroups of a head
        const bool defer = gdn_defer_norm && gdn_scratch != nullptr && quant_act_int && w.q_out_proj.fmt == QFMT_Q8_0 &&
                           cfg.value_dim() % 128 == 0 && cfg.Vd == 128;
        ga.defer_norm = defer ? 1 : 0;
        launch_gdn(ga, s);
        auto qgo = [&](int epi, float* yout, const float* res) {
            GemvQArgs q{};
            q.w = w.q_out_proj; q.x = attn; q.y = yout; q.res = res; q.eps = cfg.eps; q.act_int = quant_act_int ? 1 : 0;
            q.gdn_z = qkv + (2 * cfg.key_dim() + cfg.value_dim()); q.gdn_w = w.gnorm;
            if (!launch_gemvq(defer ? PRO_GDNNORM : PRO_PLAIN, epi, q, gemvq_grid(q.w.N, num_cu, q.w.fmt), s)) throw CmError(CM_ERR_UNSUPPORTED, "quantised weight format");
        };
        if (!rccl) qgo(EPI_RESADD, x, x);
        else {       // row-parallel out_proj over this rank's value heads: partial sums, rank 0 carries the residual
            qgo((rank == 0 || rccl->fake) ? EPI_RESADD : EPI_STORE, y, x);
            rccl->all_reduce_sum_f32(y, x, (size_t)H, s);
        }
    } else {
        for (int i = 0; i < w.n_qkv; ++i) qg(PRO_RMSNORM, EPI_STORE, w.q_qkv[i], x, w.ln1, qkv + w.qkv_row0[i], nullptr);
        AttnDecArgs a{};
        a.qkv = qkv; a.qnw = w.qn; a.knw = w.kn; a.cos = cos; a.sin = sin; a.st = st; a.block_table = d_bt;
        a.kpool = kpool(li); a.vpool = vpool(li); a.part_o = part_o; a.part_ml = part_ml;
        a.q_off = 0; a.k_off = (cfg.hybrid ? 2 * Hq_l : Hq_l) * D; a.v_off = a.k_off + Hkv_l * D;
        a.gate = cfg.hybrid ? qkv + (size_t)Hq_l * D : nullptr; a.rot_dim = cfg.rot_dim;
        a.Hkv = Hkv_l; a.page = page; a.max_pages = max_pages_per_seq; a.page_bytes = page_bytes; a.eps = cfg.eps;
        a.scale = (float)(1.0 / std::sqrt((double)D));
        if (attn_variant >= 2) {
            if (!launch_attn_decode_mfma(a, D, nrep, attn_variant == 3 ? nsplit_mfma : nsplit, kv_mode, attn, 0, 1, s)) throw CmError(CM_ERR_UNSUPPORTED, "GQA group size / head_dim");
        } else if (!launch_attn_decode(a, D, nrep, attn_splits_force ? attn_splits_force : nsplit, kv_mode, attn, 0, 1, s)) throw CmError(CM_ERR_UNSUPPORTED, "GQA group size / head_dim");
        if (!rccl) qg(PRO_PLAIN, EPI_RESADD, w.q_o, attn, nullptr, x, x);
        else {       // row-parallel: partial sums over this rank's heads, rank 0 carries the residual
            qg(PRO_PLAIN, (rank == 0 || rccl->fake) ? EPI_RESADD : EPI_STORE, w.q_o, attn, nullptr, y, x);
            rccl->all_reduce_sum_f32(y, x, (size_t)H, s);
        }
    }
    if (!w.split_gate_up) {
        qg(PRO_RMSNORM, EPI_SILUMUL, w.q_gate_up, x, w.ln2, hbuf, nullptr);
    } else {
        qg(PRO_RMSNORM, EPI_STORE, w.q_gate, x, w.ln2, gu_tmp, nullptr);
        qg(PRO_RMSNORM, EPI_STORE, w.q_up, x, w.ln2, gu_tmp + I_l, nullptr);
        launch_silu_mul(gu_tmp, gu_tmp + I_l, hbuf, I_l, s);
    }
    if (!rccl) qg(PRO_PLAIN, EPI_RESADD, w.q_down, hbuf, nullptr, x, x);
    else {
        qg(PRO_PLAIN, (rank == 0 || rccl->fake) ? EPI_RESADD : EPI_STORE, w.q_down, hbuf, nullptr, y, x);
        rccl->all_reduce_sum_f32(y, x, (size_t)H, s);
    }
}

void Model::enqueue_lm_head(bool advance) {
    // final norm + lm_head (last position only, modeling.rs:1024-1035) + arg-max
    const int H = cfg.H;
    hipStream_t s = stream;
    logits_gathered = false;
    const int v_eff = std::max(0, std::min(V_l, cfg.V - v0));
    if (quantized && q_lm_head.fmt != QFMT_NONE) {
        GemvQArgs q{};
        q.w = q_lm_head.rows(0, v_eff); q.x = x; q.nw = norm; q.y = logits + (size_t)rank * V_l;
        q.pmax = pmax + (size_t)rank * lm_grid; q.pidx = pidx + (size_t)rank * lm_grid; q.idx_base = v0; q.eps = cfg.eps;
        q.act_int = quant_act_int ? 1 : 0;
        if (!launch_gemvq(PRO_RMSNORM, EPI_ARGMAX, q, lm_grid, s)) throw CmError(CM_ERR_UNSUPPORTED, "quantised lm_head format");
        if (rccl) {
            rccl->all_gather(pmax + (size_t)rank * lm_grid, pmax, (size_t)lm_grid * sizeof(float), s);
            rccl->all_gather(pidx + (size_t)rank * lm_grid, pidx, (size_t)lm_grid * sizeof(int), s);
        }
        if (rccl && rccl->fake) launch_argmax_final(pmax + (size_t)rank * lm_grid, pidx + (size_t)rank * lm_grid, lm_grid, st, ring, RING - 1, advance ? 1 : 0, 1, s);
        else launch_argmax_final(pmax, pidx, lm_grid * tp, st, ring, RING - 1, advance ? 1 : 0, 1, s);
        return;
    }
    GemvArgs g{};
    g.W = lm_head; g.x = x; g.nw = norm; g.y = logits + (size_t)rank * V_l; g.N = v_eff; g.K = H; g.ldw = H;
    g.eps = cfg.eps; g.pmax = pmax + (size_t)rank * lm_grid; g.pidx = pidx + (size_t)rank * lm_grid; g.idx_base = v0;
    launch_gemv(PRO_RMSNORM, EPI_ARGMAX, g, lm_grid, s);
    if (rccl) {
        rccl->all_gather(pmax + (size_t)rank * lm_grid, pmax, (size_t)lm_grid * sizeof(float), s);
        rccl->all_gather(pidx + (size_t)rank * lm_grid, pidx, (size_t)lm_grid * sizeof(int), s);
    }
    if (rccl && rccl->fake) launch_argmax_final(pmax + (size_t)rank * lm_grid, pidx + (size_t)rank * lm_grid, lm_grid, st, ring, RING - 1, advance ? 1 : 0, 1, s);
    else launch_argmax_final(pmax, pidx, lm_grid * tp, st, ring, RING - 1, advance ? 1 : 0, 1, s);
}

// ------------------------------------------------------------------------------------
// prefill (S > 1): MFMA GEMMs + causal flash attention over the paged cache
// ------------------------------------------------------------------------------------
// split-K partial tiles of short prompts / narrow GEMMs (launch_gemm): its own allocation so that the vision tower, which
// only needs this workspace, does not allocate the text path's prompt buffers (and the f32 K/V shadows of int8 / int4 pages)
void Model::ensure_gemm_workspace() {
    if (!pWS) {
        pWS = dalloc<float>(gemm_ws_floats);
    }
}

bool Model::q8_prefill_eligible() const {
    if (!quantized || !quant_act_int) return false;
    const int qz = cfg.hybrid ? cfg.conv_dim() + cfg.value_dim() : 0;
    for (const LayerW& w : layers) {
        if (w.split_gate_up) return false;
        if (!gemm_q8_ok(w.q_gate_up, QGEMM_MAXM) || !gemm_q8_ok(w.q_down, QGEMM_MAXM) || (w.q_gate_up.N / 2) % 32 != 0) return false;
        if (w.full) {
            if (w.n_qkv < 1) return false;
            for (int i = 0; i < w.n_qkv; ++i) if (!gemm_q8_ok(w.q_qkv[i], QGEMM_MAXM)) return false;
            if (!gemm_q8_ok(w.q_o, QGEMM_MAXM)) return false;
        } else {
            // Gated-Delta-Net layer: [qkv | z] on the int8 cores; the bf16 a / b gate rows as one 128-column bf16 GEMM tile behind them
            if (!gemm_q8_ok(w.q_in_proj, QGEMM_MAXM) || !gemm_q8_ok(w.q_out_proj, QGEMM_MAXM)) return false;
            if (w.q_in_proj_z.fmt != QFMT_NONE && !gemm_q8_ok(w.q_in_proj_z, QGEMM_MAXM)) return false;
            if (w.in_proj_ba == nullptr || qz % 128 != 0 || in_proj_pad - qz < 128 || 2 * cfg.NV > 128) return false;
            if (w.q_in_proj.N + (w.q_in_proj_z.fmt != QFMT_NONE ? w.q_in_proj_z.N : 0) != qz) return false;
        }
    }
    return true;
}

int Model::prefill_chunk_rows() const {
    int c = opts.prefill_chunk ? (int)opts.prefill_chunk : 2048;       // PREFILL_CHUNK_SIZE engine/mod.rs:65
    // one sequence never has more than max_seq rows in a pass, but the prompts of several sequences share one (prefill_multi):
    // the row capacity is bounded by what all slots together can hold, not by one sequence
    const long all_rows = (long)max_seq * (long)std::max<size_t>(1, seqs.size());
    if ((long)c > all_rows) c = (int)all_rows;
    return c < 1 ? 1 : c;
}

void Model::ensure_prefill_buffers() {
    if (pX) return;
    const int H = cfg.H, D = cfg.D;
    chunk = prefill_chunk_rows();
    chunk_pad = (chunk + 127) / 128 * 128;
    // prompt activations as bf16 hi + lo (two products per GEMM: what the 1e-3 bound of a bf16 checkpoint needs) unless the caller asked
    // for plain bf16 -- or left the choice open (0) on QUANTISED weights: their scratch copy for the GEMMs is already rounded to bf16
    // (2^-9 per weight, tests: 1e-2 of the oracle on the dequantised weights), the activations' low halves would refine past that
    prefill_split2 = default_prefill_split2();
    const int qkv_rows = (cfg.hybrid ? 2 * Hq_l + 2 * Hkv_l : Hq_l + 2 * Hkv_l) * D;
    prefill_ok = (qkv_rows % 128 == 0) && (H % 128 == 0) && ((2 * I_l) % 128 == 0) && (H % 32 == 0) && (I_l % 32 == 0) &&
                 (!cfg.hybrid || cfg.value_dim() % 32 == 0);
    if (!prefill_ok) return;
    pX = dalloc<float>((size_t)chunk * H);
    ensure_gemm_workspace();
    if (rccl) pY = dalloc<float>((size_t)chunk * H);
    pQKV = dalloc<float>((size_t)chunk * std::max(qkv_rows, in_proj_pad));
    if (cfg.hybrid) {
        pGY = dalloc<float>((size_t)chunk * cfg.value_dim());
        gdn_pre_q = dalloc<float>((size_t)chunk * cfg.key_dim());      // three-pass Gated-Delta-Net prefill (kernels_gdn.hip)
        gdn_pre_k = dalloc<float>((size_t)chunk * cfg.key_dim());
        gdn_pre_v = dalloc<float>((size_t)chunk * cfg.value_dim());
        gdn_pre_bd = dalloc<float>((size_t)chunk * cfg.NV * 2);
        gdn_pre_g = dalloc<float>((size_t)chunk * cfg.NV);
        if (cfg.Kd == 128 && cfg.Vd == 128)           // chunk-parallel scan scratch: W | U0 | P | G per (64-token chunk, value head)
            gdn_ck = dalloc<float>((size_t)((chunk + GDN_CK - 1) / GDN_CK) * cfg.NV * GDN_CK_FLOATS);
    }
    auto z = [&](size_t n) {
        uint16_t* p = dalloc<uint16_t>(n);
        CM_HIP(hipMemsetAsync(p, 0, n * sizeof(uint16_t), stream));
        return p;
    };
    pXN_hi = z((size_t)chunk_pad * H); pXN_lo = z((size_t)chunk_pad * H);
    pQ_hi = z((size_t)chunk_pad * Hq_l * D); pQ_lo = z((size_t)chunk_pad * Hq_l * D);
    const size_t at_cols = std::max((size_t)Hq_l * D, (size_t)(cfg.hybrid ? cfg.value_dim() : 0));
    pAT_hi = z((size_t)chunk_pad * at_cols); pAT_lo = z((size_t)chunk_pad * at_cols);
    pHH_hi = z((size_t)chunk_pad * I_l); pHH_lo = z((size_t)chunk_pad * I_l);
    if (kvq()) {
        const size_t shadow = (size_t)max_pages_per_seq * Hkv_l * page * D;
        kshadow = dalloc<float>(shadow);
        vshadow = dalloc<float>(shadow);
        std::vector<int32_t> ident((size_t)max_pages_per_seq);
        for (int i = 0; i < max_pages_per_seq; ++i) ident[(size_t)i] = i;
        d_ident_bt = dalloc<int>((size_t)max_pages_per_seq);
        CM_HIP(hipMemcpy(d_ident_bt, ident.data(), ident.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    // Q8_0-layout weights (GGUF Q8_0 / Q4_0 / Q5_0, every ISQ mode), integer-dot activation mode: the prompt's projections run on the
    // int8 matrix cores in panels of <= 128 rows (prefill_layers) -- the arithmetic of the decode step and of candle's CPU QMatMul
    // (ops/linear.rs:18-51: activation row -> Q8_0 blocks, ggml_vec_dot_q8_0_q8_0 per output), no dequantised copy of any matrix
    q8_prefill = q8_prefill_want && q8_prefill_eligible();
    if (q8_prefill) {
        ensure_batch_buffers();                                    // qx_codes / qx_scales pairs (sized for a whole pass there)
        pATf = dalloc<float>((size_t)chunk * Hq_l * D);
        pHf = dalloc<float>((size_t)chunk * I_l);                  // silu(gate) * up of the pass, f32 (the down projection's quantiser input)
        for (LayerW& w : layers) {
            if (w.full) continue;
            w.ba_pad = dalloc<uint16_t>((size_t)128 * H);          // the a / b rows as one zero-padded 128-row GEMM tile
            CM_HIP(hipMemsetAsync(w.ba_pad, 0, (size_t)128 * H * sizeof(uint16_t), stream));
            CM_HIP(hipMemcpyAsync(w.ba_pad, w.in_proj_ba, (size_t)2 * cfg.NV * H * sizeof(uint16_t), hipMemcpyDeviceToDevice, stream));
        }
    }
    if (quantized && !q8_prefill) {
        // one dequantised matrix at a time: bf16 hi plane + lo plane (parity mode), and the f32 gate|up sums of the two-pass GEMM
        wq_scratch_elems = std::max(std::max((size_t)2 * I_l * H, (size_t)qkv_rows * H), (size_t)in_proj_pad * H);
        wq_scratch = dalloc<uint16_t>(2 * wq_scratch_elems);
        pGU = dalloc<float>((size_t)chunk * 2 * I_l);
        // CM_QUANT_PREFILL_CACHE=1 (opt-in; round 5 switched it on by itself whenever the copies fit a quarter of the free HBM): keep every
        // dequantised matrix (hi + lo: 4 bytes per weight -- more than the bf16 checkpoint) instead of dequantising it again per pass
        // (4.5 % of a Q8_0 serving run, before Q8_0-layout models moved to the int8 prompt pass).  An allocation that fails mid-pass
        // falls back to the scratch (deq_w)
        wq_cache = false;
        if (const char* e = getenv("CM_QUANT_PREFILL_CACHE")) wq_cache = atoi(e) != 0;
    }
    d_ids = (uint32_t*)dalloc<int>(chunk);
    CM_HIP(hipHostMalloc((void**)&h_ids, (size_t)chunk * sizeof(uint32_t)));
}

// the decoder layers over the S rows of pX.  The row-wise work (norms, every GEMM) runs once over all rows; what mixes tokens --
// QK-norm / RoPE / KV append, the causal attention, the Gated-Delta-Net scan -- runs per SEGMENT: rows [row0, row0 + S) of one
// sequence at positions start_pos..., on that sequence's pages and state slot.  One segment = the classic single-prompt chunk;
// several = the prompts of several requests sharing one pass over the weights (prefill_multi).
void Model::prefill_layers(int S, const PrefillSeg* segs, int nseg, size_t off) {
    const int H = cfg.H, D = cfg.D;
    hipStream_t s = stream;
    const int qkv_rows = (cfg.hybrid ? 2 * Hq_l + 2 * Hkv_l : Hq_l + 2 * Hkv_l) * D;
    const bool sp2 = prefill_split2;
    // the bf16 operand of a quantised projection: `fill(dst)` enqueues its dequantisation -- into the layer's own copy the first time
    // (cache on), else into the one scratch matrix every time (stream order keeps that safe)
    // Parity mode (hi + lo activations, the default): the operand is ALSO two bf16 terms -- a dequantised ggml weight (code x f16 scale,
    // K-quants: x 6-bit sub-scale - min) does not fit bf16's 8 significand bits, and the single rounded plane alone put the prompt
    // pass 3.2e-3 .. 3.7e-3 from the f32 oracle on the dequantised weights (round 5); with the lo plane (launch_gemm: a second pass
    // A_hi . W_lo^T) operand and activations both carry 16 bits.  cm_opts.prefill_split = 1: one plane each, the fast approximate mode.
    struct WOp { const uint16_t* hi; const uint16_t* lo; };
    auto deq_w = [&](LayerW& lw, int slot, size_t elems, auto&& fill) -> WOp {
        if (wq_cache && !lw.dq[slot]) {
            try { lw.dq[slot] = dalloc<uint16_t>(2 * elems); fill(lw.dq[slot], lw.dq[slot] + elems); }
            catch (const CmError& e) {                   // HBM ran out mid-pass: this and every later matrix through the scratch
                if (e.code != CM_ERR_OOM) throw;
                (void)hipGetLastError();
                wq_cache = false;
            }
        }
        if (wq_cache && lw.dq[slot]) return WOp{lw.dq[slot], lw.dq[slot] + elems};
        if (lw.dq[slot]) return WOp{lw.dq[slot], lw.dq[slot] + elems};      // (cached before the cache was switched off)
        fill(wq_scratch, sp2 ? wq_scratch + wq_scratch_elems : nullptr);
        return WOp{wq_scratch, wq_scratch + wq_scratch_elems};
    };
    auto set_w = [&](GemmArgs& ga, const WOp& op) { ga.W = op.hi; ga.W_lo = sp2 ? op.lo : nullptr; };
    // the RMSNorm in front of a projection is written by the split-K reduction launch of the row-parallel projection before it when
    // that GEMM splits K (GemmArgs::norm_w; launch_gemm runs the row kernel itself when it does not): o_proj / out_proj -> ln2,
    // down_proj -> the next layer's ln1 (not under TP: the norm follows the all-reduce; not across a DeepStack injection)
    auto next_norm = [&](GemmArgs& ga, const float* nw) {
        ga.norm_w = nw; ga.norm_hi = pXN_hi; ga.norm_lo = sp2 ? pXN_lo : nullptr; ga.norm_eps = cfg.eps;
    };
    // ---- int8 prompt pass (q8_prefill): rows in panels of <= QGEMM_MAXM; quantiser + int8-MFMA GEMM per projection and panel ----
    const bool q8p = q8_prefill;
    const int xs8 = S > (int)QGEMM_MAXM ? q8_xs : (int)QGEMM_MAXM;       // scale-row stride of this pass's code buffers
    // the rows currently held as codes: (source, norm weight, K, kind: 0 = Q8_0 blocks, 1 = Q8_K groups for Q4_K weights)
    const float* q8_src = nullptr; const float* q8_nw = nullptr; int q8_K = 0, q8_kind = -1;
    auto q8_in = [&](const QWeight& qw, const float* xin, int ldx, const float* nw, int m) {
        const int kind = (qw.fmt == QFMT_Q4_K || qw.fmt == QFMT_Q6_K) ? 1 : 0;
        if (q8_src == xin && q8_nw == nw && q8_K == qw.K && q8_kind == kind) return;
        if (kind) launch_quant_rows_q8k(xin, ldx, nw, cfg.eps, qx_codes, qx_scales, m, qw.K, s, xs8);
        else launch_quant_rows_q8(xin, ldx, nw, cfg.eps, qx_codes, qx_scales, m, qw.K, s, xs8);
        q8_src = xin; q8_nw = nw; q8_K = qw.K; q8_kind = kind;
        if (q_capture && !kind) q_capture_rows(m, qw.K, xs8);
    };
    // y (+)= W . rows(xin)^T over the pass's m rows (quantised above unless the codes already hold them); next_nw / next_plain: the rows
    // written are the next projection's input -- quantised as Q8_0 blocks by the reduction launch (or the unsplit gate|up GEMM itself)
    auto q8_mm = [&](const QWeight& qw, const float* xin, int ldx, const float* nw, int epi, float* y, int ldy, int m, const float* next_nw, bool next_plain) {
        q8_in(qw, xin, ldx, nw, m);
        QGemmArgs qg{};
        qg.w = qw; qg.xq = qx_codes; qg.xd = qx_scales; qg.M = m; qg.xs = xs8;
        QNext nx{next_nw, cfg.eps, qx_codes, qx_scales, qx_codes2, qx_scales2};
        const int kout = epi == EPI_SILUMUL ? qw.N / 2 : qw.N;
        const bool want_next = (next_nw != nullptr || next_plain) && ldy == kout;
        int fused = 0;
        if (!launch_gemm_q8(qg, epi, y, ldy, pWS, gemm_ws_floats, num_cu, s, want_next ? &nx : nullptr, &fused))
            throw CmError(CM_ERR_UNSUPPORTED, "int8 prompt GEMM shape");
        if (fused == 2) { std::swap(qx_codes, qx_codes2); std::swap(qx_scales, qx_scales2); }
        if (fused) { q8_src = y; q8_nw = next_nw; q8_K = kout; q8_kind = 0; if (q_capture) q_capture_rows(m, kout, xs8); }
        else if (epi == EPI_RESADD && q8_src == y) q8_src = nullptr;       // (the rows the codes were made from have changed)
    };
    // the rest of a layer after the token mixer, panel by panel: o_proj / out_proj over the mixer's f32 rows `ain` [S][AC], gate|up,
    // down_proj -- a panel's rows of silu(gate) * up never leave the [128][I] scratch
    auto q8_tail = [&](const LayerW& w, const QWeight& wo, const float* ain, int AC, int li) {
        if (!rccl) q8_mm(wo, ain, AC, nullptr, EPI_RESADD, pX, H, S, w.ln2, false);
        else {
            q8_mm(wo, ain, AC, nullptr, EPI_STORE, pY, H, S, nullptr, false);
            rccl->all_reduce_sum_f32(pY, pY, (size_t)S * H, s);
            launch_add_rows(pX, pY, (size_t)S * H, s);
            q8_src = nullptr;
        }
        q8_mm(w.q_gate_up, pX, H, w.ln2, EPI_SILUMUL, pHf, I_l, S, nullptr, true);
        if (!rccl) { q8_mm(w.q_down, pHf, I_l, nullptr, EPI_RESADD, pX, H, S, nullptr, false); q8_src = nullptr; }
        else {
            q8_mm(w.q_down, pHf, I_l, nullptr, EPI_STORE, pY, H, S, nullptr, false);
            rccl->all_reduce_sum_f32(pY, pY, (size_t)S * H, s);
            launch_add_rows(pX, pY, (size_t)S * H, s);
            q8_src = nullptr;
        }
        if (li < deep_layers && splice_map_dev != nullptr) {
            launch_add_rows_map(pX, vDeep + (size_t)li * deep_stride, splice_map_dev + off, S, H, s);
            q8_src = nullptr;
        }
    };
    bool xn_ready = false;
    for (int li = 0; li < cfg.L; ++li) {
        LayerW& w = layers[(size_t)li];
        if (!xn_ready && !(q8p && w.full)) launch_rmsnorm_rows(pX, w.ln1, pXN_hi, sp2 ? pXN_lo : nullptr, S, H, cfg.eps, s);
        xn_ready = false;
        GemmArgs g{};
        g.ws = pWS; g.ws_floats = gemm_ws_floats; g.wide256 = gemm256;
        if (!w.full) {
            // ---- Gated Delta Net layer: in_proj GEMM, sequential delta-rule scan, out_proj GEMM ----
            g.A_hi = pXN_hi; g.A_lo = sp2 ? pXN_lo : nullptr; g.W = w.in_proj; g.C = pQKV; g.ldc = in_proj_pad;
            if (q8p) {
                // [qkv | z] rows on the int8 matrix cores, panel by panel; the bf16 a / b rows as one 128-column bf16 GEMM tile (hi + lo
                // activations from the rmsnorm_rows launch above) into the columns behind them
                const int qz = cfg.conv_dim() + cfg.value_dim();
                q8_mm(w.q_in_proj, pX, H, w.ln1, EPI_STORE, pQKV, in_proj_pad, S, nullptr, false);
                if (w.q_in_proj_z.fmt != QFMT_NONE) q8_mm(w.q_in_proj_z, pX, H, w.ln1, EPI_STORE, pQKV + w.q_in_proj.N, in_proj_pad, S, nullptr, false);
                g.W = w.ba_pad; g.C = pQKV + qz; g.M = S; g.N = 128; g.K = H;
                if (!launch_gemm(g, GEPI_STORE, s)) throw CmError(CM_ERR_UNSUPPORTED, "gemm shape");
            } else {
            if (quantized) {     // [qkv | z] dequantised, then the bf16 b / a rows, then the zero padding of the merged matrix
                const size_t qz = (size_t)cfg.conv_dim() + cfg.value_dim(), nba = (size_t)2 * cfg.NV;
                set_w(g, deq_w(w, 0, (size_t)in_proj_pad * H, [&](uint16_t* dst, uint16_t* lo) {
                    launch_dequant_bf16(w.q_in_proj, dst, 1, 0, s, lo);
                    if (w.q_in_proj_z.fmt != QFMT_NONE)
                        launch_dequant_bf16(w.q_in_proj_z, dst + (size_t)w.q_in_proj.N * H, 1, 0, s, lo ? lo + (size_t)w.q_in_proj.N * H : nullptr);
                    CM_HIP(hipMemcpyAsync(dst + qz * H, w.in_proj_ba, nba * H * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
                    CM_HIP(hipMemsetAsync(dst + (qz + nba) * H, 0, ((size_t)in_proj_pad - qz - nba) * H * sizeof(uint16_t), s));
                    if (lo) CM_HIP(hipMemsetAsync(lo + qz * H, 0, ((size_t)in_proj_pad - qz) * H * sizeof(uint16_t), s));      // (the a / b rows are bf16: no lo term)
                }));
            }
            g.M = S; g.N = in_proj_pad; g.K = H;
            if (!launch_gemm(g, GEPI_STORE, s)) throw CmError(CM_ERR_UNSUPPORTED, "gemm shape");
            }
            GdnArgs ga{};
            ga.conv_w = w.conv_w; ga.conv_pool = conv_pool; ga.state_pool = state_pool;
            ga.A_log = w.A_log; ga.dt_bias = w.dt_bias; ga.gnorm_w = w.gnorm; ga.st = nullptr;
            ga.proj_stride = in_proj_pad; ga.out_stride = cfg.value_dim(); ga.NV = cfg.NV; ga.vpg = cfg.NV / cfg.NK; ga.chunked = gdn_chunked ? 1 : 0;
            ga.key_dim = cfg.key_dim(); ga.layer_idx = w.gdn_idx; ga.gdn_layers = gdn_layers; ga.eps = cfg.eps;
            ga.n_seq = 1;
            ga.pre_q = gdn_pre_q; ga.pre_k = gdn_pre_k; ga.pre_v = gdn_pre_v; ga.pre_bd = gdn_pre_bd; ga.pre_g = gdn_pre_g; ga.ck = gdn_ck_on ? gdn_ck : nullptr;
            // the conv windows are double-buffered by position parity: every launch must advance an ODD
            // number of positions, so an even chunk is scanned as (S-1) + 1.  One scan per sequence of the pass (its own state slot).
            for (int gi = 0; gi < nseg; ++gi) {
                const PrefillSeg& sg = segs[gi];
                ga.slot = sg.seq;
                int done = 0;
                while (done < sg.S) {
                    const int part = ((sg.S - done) % 2 == 1) ? (sg.S - done) : (sg.S - done - 1);
                    ga.proj = pQKV + (size_t)(sg.row0 + done) * in_proj_pad; ga.out = pGY + (size_t)(sg.row0 + done) * cfg.value_dim();
                    ga.start_pos = sg.start_pos + done; ga.S = part;
                    launch_gdn(ga, s);
                    done += part;
                }
            }
            if (q8p) { q8_tail(w, w.q_out_proj, pGY, cfg.value_dim(), li); continue; }
            launch_split_rows(pGY, pAT_hi, sp2 ? pAT_lo : nullptr, (size_t)S * cfg.value_dim(), s);
            g = GemmArgs{}; g.ws = pWS; g.ws_floats = gemm_ws_floats; g.wide256 = gemm256;
            g.A_hi = pAT_hi; g.A_lo = sp2 ? pAT_lo : nullptr; g.W = w.out_proj; g.M = S; g.N = H; g.K = cfg.value_dim(); g.ldc = H;
            if (quantized) set_w(g, deq_w(w, 1, (size_t)w.q_out_proj.N * w.q_out_proj.K, [&](uint16_t* dst, uint16_t* lo) { launch_dequant_bf16(w.q_out_proj, dst, 1, 0, s, lo); }));
            if (!rccl) { g.C = pX; next_norm(g, w.ln2); launch_gemm(g, GEPI_RESADD, s); }
            else {
                g.C = pY; launch_gemm(g, GEPI_STORE, s);
                rccl->all_reduce_sum_f32(pY, pY, (size_t)S * H, s);
                launch_add_rows(pX, pY, (size_t)S * H, s);
            }
        } else {
        if (q8p) {
            for (int i = 0; i < w.n_qkv; ++i) q8_mm(w.q_qkv[i], pX, H, w.ln1, EPI_STORE, pQKV + w.qkv_row0[i], qkv_rows, S, nullptr, false);
        } else {
        g.A_hi = pXN_hi; g.A_lo = (sp2 && !(prefill_lo_mask & 1)) ? pXN_lo : nullptr; g.W = w.qkv; g.C = pQKV; g.ldc = qkv_rows;
        if (quantized) {      // one dequantised matrix at a time in the bf16 scratch (stream order keeps it safe)
            set_w(g, deq_w(w, 0, (size_t)qkv_rows * H, [&](uint16_t* dst, uint16_t* lo) {
                for (int i = 0; i < w.n_qkv; ++i)
                    launch_dequant_bf16(w.q_qkv[i], dst + (size_t)w.qkv_row0[i] * H, 1, 0, s, lo ? lo + (size_t)w.qkv_row0[i] * H : nullptr);
            }));
        }
        g.M = S; g.N = qkv_rows; g.K = H;
        if (!launch_gemm(g, GEPI_STORE, s)) throw CmError(CM_ERR_UNSUPPORTED, "gemm shape");
        }
        // QK-norm + RoPE + KV append and the causal attention run once per sequence of the pass (its own pages and positions)
        const bool kvq = this->kvq();
        if (kvq && nseg != 1) throw CmError(CM_ERR_UNSUPPORTED, "multi-sequence prompt pass over quantised KV pages");
        if (nseg > 1 && seg_batch && seg_tables_ok && !kvq) {
            // every sequence of the pass in ONE RoPE / KV-append launch and ONE causal-attention launch (segment tables on the
            // device; 16 prompts of 128 tokens were 16 launches of 64 workgroups each, per kernel and layer)
            QkRopeArgs q{};
            q.qkv = pQKV; q.qnw = w.qn; q.knw = w.kn; q.cos = cos; q.sin = sin; q.block_table = d_btb;
            q.kpool = kpool(li); q.vpool = vpool(li); q.q_hi = pQ_hi; q.q_lo = pQ_lo;
            q.Hq = Hq_l; q.Hkv = Hkv_l; q.page = page; q.start_pos = 0; q.eps = cfg.eps;
            q.row_stride = qkv_rows; q.q_off = 0; q.k_off = (cfg.hybrid ? 2 * Hq_l : Hq_l) * D; q.v_off = q.k_off + Hkv_l * D;
            q.rot_dim = cfg.rot_dim; q.pos3 = nullptr; q.pos3_stride = pos3_stride;
            q.sec_h = cfg.mrope_sec[1]; q.sec_w = cfg.mrope_sec[2];
            q.scale = (float)(1.0 / std::sqrt((double)D));
            q.segs = d_segtab; q.rowseg = d_rowseg;
            launch_qknorm_rope_kv(q, D, S, kv_mode, s);
            AttnPreArgs at{};
            at.q_hi = pQ_hi; at.q_lo = pQ_lo; at.block_table = d_btb;
            at.kpool = kpool(li); at.vpool = vpool(li);
            at.out_hi = pAT_hi; at.out_lo = pAT_lo;
            at.S = S; at.Hq = Hq_l; at.Hkv = Hkv_l; at.nrep = nrep;
            at.page = page; at.start_pos = 0; at.causal = 1;
            at.gate = cfg.hybrid ? pQKV + (size_t)Hq_l * D : nullptr; at.gate_stride = qkv_rows;
            at.segs = d_segtab; at.tiles = d_tiles; at.ntiles = seg_ntiles;
            if (q8p) at.out_f32 = pATf;
            launch_attn_prefill(at, D, kv_f32 ? KV_F32 : kv_mode, s);
        } else
        for (int gi = 0; gi < nseg; ++gi) {
        const PrefillSeg& sg = segs[gi];
        const int sp = sg.start_pos;
        QkRopeArgs q{};
        q.qkv = pQKV + (size_t)sg.row0 * qkv_rows; q.qnw = w.qn; q.knw = w.kn; q.cos = cos; q.sin = sin; q.block_table = sg.bt;
        q.kpool = kpool(li); q.vpool = vpool(li); q.q_hi = pQ_hi + (size_t)sg.row0 * Hq_l * D; q.q_lo = pQ_lo + (size_t)sg.row0 * Hq_l * D;
        q.Hq = Hq_l; q.Hkv = Hkv_l; q.page = page; q.start_pos = sp; q.eps = cfg.eps;
        q.row_stride = qkv_rows; q.q_off = 0; q.k_off = (cfg.hybrid ? 2 * Hq_l : Hq_l) * D; q.v_off = q.k_off + Hkv_l * D;
        q.rot_dim = cfg.rot_dim; q.pos3 = (pos3_dev && nseg == 1) ? pos3_dev + off : nullptr; q.pos3_stride = pos3_stride;
        q.rope_delta = sg.rope_delta;      // text tokens after an image prompt rotate at pos + delta, like the decode step
        q.sec_h = cfg.mrope_sec[1]; q.sec_w = cfg.mrope_sec[2];
        q.scale = (float)(1.0 / std::sqrt((double)D));
        if (kvq) {
            // KvCache::Quant (qwen3_5/kv_cache.rs:303-325): append() returns dequantize(full cache); the attention of this
            // chunk therefore reads an f32 shadow of the layer: old tokens dequantised from the pages, new tokens written
            // next to their codes by the append kernel
            q.page_bytes = page_bytes; q.kshadow = kshadow; q.vshadow = vshadow;
            launch_kvq_dequant_prefix(kpool(li), vpool(li), sg.bt, kshadow, vshadow, sp, Hkv_l, page, D, kv_mode, page_bytes, s);
        }
        launch_qknorm_rope_kv(q, D, sg.S, kv_mode, s);
        AttnPreArgs at{};
        at.q_hi = q.q_hi; at.q_lo = q.q_lo; at.block_table = kvq ? d_ident_bt : sg.bt;
        at.kpool = kvq ? (void*)kshadow : kpool(li); at.vpool = kvq ? (void*)vshadow : vpool(li);
        at.out_hi = pAT_hi + (size_t)sg.row0 * Hq_l * D; at.out_lo = pAT_lo + (size_t)sg.row0 * Hq_l * D;
        at.S = sg.S; at.Hq = Hq_l; at.Hkv = Hkv_l; at.nrep = nrep;
        at.page = page; at.start_pos = sp; at.causal = 1;
        at.gate = cfg.hybrid ? pQKV + (size_t)sg.row0 * qkv_rows + (size_t)Hq_l * D : nullptr; at.gate_stride = qkv_rows;
        if (q8p) at.out_f32 = pATf + (size_t)sg.row0 * Hq_l * D;
        launch_attn_prefill(at, D, (kv_f32 || kvq) ? KV_F32 : kv_mode, s);
        }
        if (q8p) { q8_tail(w, w.q_o, pATf, Hq_l * D, li); continue; }
        g = GemmArgs{}; g.ws = pWS; g.ws_floats = gemm_ws_floats; g.wide256 = gemm256;
        g.A_hi = pAT_hi; g.A_lo = (sp2 && !(prefill_lo_mask & 2)) ? pAT_lo : nullptr; g.W = w.o; g.M = S; g.N = H; g.K = Hq_l * D; g.ldc = H;
        if (quantized) set_w(g, deq_w(w, 1, (size_t)w.q_o.N * w.q_o.K, [&](uint16_t* dst, uint16_t* lo) { launch_dequant_bf16(w.q_o, dst, 1, 0, s, lo); }));
        if (!rccl) { g.C = pX; next_norm(g, w.ln2); launch_gemm(g, GEPI_RESADD, s); }
        else {
            g.C = pY; launch_gemm(g, GEPI_STORE, s);
            rccl->all_reduce_sum_f32(pY, pY, (size_t)S * H, s);
            launch_add_rows(pX, pY, (size_t)S * H, s);
        }
        }   // full-attention layer
        if (rccl) launch_rmsnorm_rows(pX, w.ln2, pXN_hi, sp2 ? pXN_lo : nullptr, S, H, cfg.eps, s);
        g = GemmArgs{}; g.ws = pWS; g.ws_floats = gemm_ws_floats; g.wide256 = gemm256;
        g.A_hi = pXN_hi; g.A_lo = (sp2 && !(prefill_lo_mask & 4)) ? pXN_lo : nullptr; g.W = w.gate_up; g.M = S; g.N = 2 * I_l; g.K = H;
        if (quantized) {
            set_w(g, deq_w(w, 2, (size_t)2 * I_l * H, [&](uint16_t* dst, uint16_t* lo) {
                if (!w.split_gate_up) launch_dequant_bf16(w.q_gate_up, dst, 1, 0, s, lo);
                else { launch_dequant_bf16(w.q_gate, dst, 2, 0, s, lo); launch_dequant_bf16(w.q_up, dst, 2, 1, s, lo); }
            }));
            g.gu_tmp = pGU;
        }
        g.H_hi = pHH_hi; g.H_lo = sp2 ? pHH_lo : nullptr;
        launch_gemm(g, GEPI_SILUMUL, s);
        g = GemmArgs{}; g.ws = pWS; g.ws_floats = gemm_ws_floats; g.wide256 = gemm256;
        g.A_hi = pHH_hi; g.A_lo = (sp2 && !(prefill_lo_mask & 8)) ? pHH_lo : nullptr; g.W = w.down; g.M = S; g.N = H; g.K = I_l; g.ldc = H;
        if (quantized) set_w(g, deq_w(w, 3, (size_t)w.q_down.N * w.q_down.K, [&](uint16_t* dst, uint16_t* lo) { launch_dequant_bf16(w.q_down, dst, 1, 0, s, lo); }));
        if (!rccl) {
            g.C = pX;
            xn_ready = li + 1 < cfg.L && !(li < deep_layers && splice_map_dev != nullptr);
            if (xn_ready) next_norm(g, layers[(size_t)li + 1].ln1);
            launch_gemm(g, GEPI_RESADD, s);
        }
        else {
            g.C = pY; launch_gemm(g, GEPI_STORE, s);
            rccl->all_reduce_sum_f32(pY, pY, (size_t)S * H, s);
            launch_add_rows(pX, pY, (size_t)S * H, s);
        }
        // DeepStack (qwen3_vl/text.rs:262-333): the li-th feature map of the vision tower is added onto the hidden states
        // of the visual positions after decoder layer li
        if (li < deep_layers && splice_map_dev != nullptr)
            launch_add_rows_map(pX, vDeep + (size_t)li * deep_stride, splice_map_dev + off, S, H, s);
    }
}

// Whole prompts of several sequences in ONE pass over the weights (the continuous-batching engine's prefill step: a prompt of
// 128 tokens alone fills one m-tile and costs what 1024 rows cost).  Every sequence starts at position 0 (a re-prefill after
// preemption included); together at most prefill_chunk tokens and MAXB sequences.  Per sequence the result is what forward()
// gives: its K/V pages (and GDN state), the arg-max of its last position in greedy_out[i], its logits in logitsb[i * V].
void Model::prefill_multi(const int32_t* sq, const uint32_t* const* ids, const size_t* lens, size_t n_items, uint32_t* greedy_out) {
    if (n_items == 0 || n_items > (size_t)MAXB) throw CmError(CM_ERR_INVALID, "prefill_multi: 1..128 sequences");
    if (kvq()) throw CmError(CM_ERR_UNSUPPORTED, "prefill_multi over quantised KV pages");
    if (rccl && cfg.V % tp != 0) throw CmError(CM_ERR_UNSUPPORTED, "prefill_multi under tensor parallelism needs vocab_size divisible by tp_size");
    ensure_prefill_buffers();
    if (!prefill_ok || no_prefill || (quantized && !quant_prefill)) throw CmError(CM_ERR_UNSUPPORTED, "prefill_multi: the MFMA prompt pass is not available for this model / mode");
    ensure_batch_buffers();
    const int H = cfg.H;
    hipStream_t s = stream;
    size_t total = 0;
    for (size_t i = 0; i < n_items; ++i) {
        if (!ids[i] || lens[i] == 0) throw CmError(CM_ERR_INVALID, "empty input");
        for (size_t c = 0; c < i; ++c) if (sq[c] == sq[i]) throw CmError(CM_ERR_INVALID, "sequence appears twice in one pass");
        if (lens[i] > (size_t)max_seq) throw CmError(CM_ERR_RANGE, "sequence longer than max_seq_len");
        for (size_t k = 0; k < lens[i]; ++k) if (ids[i][k] >= (uint32_t)cfg.V) throw CmError(CM_ERR_RANGE, "token id >= vocab_size");
        (void)seq(sq[i]);
        total += lens[i];
    }
    if (total > (size_t)chunk) throw CmError(CM_ERR_RANGE, "prefill_multi: more tokens than one prefill chunk");
    CM_HIP(hipStreamSynchronize(s));                             // pinned staging reuse
    std::vector<PrefillSeg> segs(n_items);
    int row = 0;
    for (size_t i = 0; i < n_items; ++i) {
        seq_truncate(sq[i], 0);                                  // self.clear_kv_cache() of a fresh prompt (also resets the GDN state)
        ensure_pages(sq[i], (int64_t)lens[i]);
        Seq& q = seq(sq[i]);
        memcpy(h_ids + row, ids[i], lens[i] * sizeof(uint32_t));
        memcpy(h_btb + i * (size_t)max_pages_per_seq, q.pages.data(), q.pages.size() * sizeof(int32_t));
        segs[i] = PrefillSeg{row, (int)lens[i], 0, sq[i], 0, d_btb + i * (size_t)max_pages_per_seq};
        StepState& hs = h_stb[i];
        memset(&hs, 0, sizeof hs);
        hs.token = ids[i][lens[i] - 1]; hs.pos = (int32_t)lens[i] - 1; hs.slot = sq[i];
        row += (int)lens[i];
    }
    if (active_seq >= 0) { active_seq = -1; active_pages_uploaded = 0; }     // d_bt is not touched, but GDN slots / states moved on
    // segment tables of the pass for the one-launch RoPE / attention kernels: row -> segment, and the (segment, query tile) list
    // sorted by ascending key-tile count (the kernel walks it from the back: longest tiles first)
    seg_tables_ok = false;
    if (seg_batch && n_items > 1) {
        if (!d_segtab) {
            d_segtab = (PrefillSegDev*)dalloc<int>((size_t)MAXB * sizeof(PrefillSegDev) / sizeof(int));
            d_rowseg = dalloc<int32_t>((size_t)chunk);
            d_tiles = (int2*)dalloc<int>(2 * ((size_t)chunk / 64 + MAXB + 1));
        }
        std::vector<PrefillSegDev> hs(n_items);
        std::vector<int32_t> rs(total);
        std::vector<int2> tl;
        for (size_t i = 0; i < n_items; ++i) {
            hs[i] = PrefillSegDev{segs[i].row0, segs[i].S, 0, 0, (int)(i * (size_t)max_pages_per_seq), 0, 0, 0};
            for (int r = 0; r < segs[i].S; ++r) rs[(size_t)segs[i].row0 + r] = (int32_t)i;
            for (int qt = 0; qt < (segs[i].S + 63) / 64; ++qt) tl.push_back(int2{(int)i, qt});
        }
        std::stable_sort(tl.begin(), tl.end(), [](const int2& x, const int2& y) { return x.y < y.y; });     // start_pos = 0: key tiles = qt + 1
        seg_ntiles = (int)tl.size();
        CM_HIP(hipMemcpy(d_segtab, hs.data(), hs.size() * sizeof(PrefillSegDev), hipMemcpyHostToDevice));
        CM_HIP(hipMemcpy(d_rowseg, rs.data(), rs.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        CM_HIP(hipMemcpy(d_tiles, tl.data(), tl.size() * sizeof(int2), hipMemcpyHostToDevice));
        seg_tables_ok = true;
    }
    CM_HIP(hipMemcpyAsync(d_ids, h_ids, total * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    CM_HIP(hipMemcpyAsync(d_btb, h_btb, n_items * (size_t)max_pages_per_seq * sizeof(int32_t), hipMemcpyHostToDevice, s));
    CM_HIP(hipMemcpyAsync(stb, h_stb, n_items * sizeof(StepState), hipMemcpyHostToDevice, s));
    if (quantized && q_embed.fmt != QFMT_NONE) launch_embed_rows_q(q_embed, d_ids, pX, (int)total, H, cfg.V, s);
    else launch_embed_rows(embed, d_ids, pX, (int)total, H, cfg.V, s);
    prefill_layers((int)total, segs.data(), (int)n_items, 0);
    for (size_t i = 0; i < n_items; ++i)                         // last position of every prompt -> the rows of the batched head
        CM_HIP(hipMemcpyAsync(xb + i * (size_t)H, pX + (size_t)(segs[i].row0 + segs[i].S - 1) * H, (size_t)H * sizeof(float), hipMemcpyDeviceToDevice, s));
    lm_head_rows((int)n_items, true);
    CM_HIP(hipMemcpyAsync(h_stb, stb, n_items * sizeof(StepState), hipMemcpyDeviceToHost, s));
    CM_HIP(hipStreamSynchronize(s));
    for (size_t i = 0; i < n_items; ++i) {
        seq(sq[i]).len = (int64_t)lens[i];
        if (greedy_out) greedy_out[i] = h_stb[i].next;
    }
    logits_gathered = false;
}

void Model::prefill(const uint32_t* ids, size_t n, size_t start_pos) {
    const int H = cfg.H;
    hipStream_t s = stream;
    for (size_t off = 0; off < n; off += (size_t)chunk) {
        const int S = (int)std::min<size_t>((size_t)chunk, n - off);
        const int sp = (int)(start_pos + off);
        CM_HIP(hipStreamSynchronize(s));                       // h_ids reuse
        if (embeds_host != nullptr) {                          // forward_embeds: the caller's hidden rows ARE the layer-0 input
            CM_HIP(hipMemcpyAsync(pX, embeds_host + off * (size_t)H, (size_t)S * H * sizeof(float), hipMemcpyHostToDevice, s));
            CM_HIP(hipStreamSynchronize(s));                   // (pageable source: do not let the caller's buffer race)
        } else {
        memcpy(h_ids, ids + off, (size_t)S * sizeof(uint32_t));
        CM_HIP(hipMemcpyAsync(d_ids, h_ids, (size_t)S * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        if (quantized && q_embed.fmt != QFMT_NONE) launch_embed_rows_q(q_embed, d_ids, pX, S, H, cfg.V, s);
        else launch_embed_rows(embed, d_ids, pX, S, H, cfg.V, s);
        }
        if (splice_map_dev) launch_splice_rows(pX, vFeat, splice_map_dev + off, S, H, s);   // image rows over <|image_pad|>
        { const PrefillSeg one{0, S, sp, active_seq, seqs[(size_t)active_seq].rope_delta, d_bt}; prefill_layers(S, &one, 1, off); }
        if (off + (size_t)S >= n) {     // last chunk: logits of the LAST position only (modeling.rs:1032-1035)
            CM_HIP(hipMemcpyAsync(x, pX + (size_t)(S - 1) * H, (size_t)H * sizeof(float), hipMemcpyDeviceToDevice, s));
            launch_set_state(st, ids ? ids[n - 1] : 0u, (int32_t)(start_pos + n - 1), active_seq, seqs[(size_t)active_seq].rope_delta, s);
            ++ring_count;
            enqueue_lm_head(true);
        }
    }
}

void Model::run_decode_step(bool advance, int64_t ctx_len) {
    (void)advance;   // the device state always advances; hosts that drive positions overwrite it
    ++ring_count;    // host mirror of st->pad (ring write index)
    // attention variant by context length (host-known): one captured graph per variant
    attn_variant = (attn_heads_max > 0 && ctx_len <= attn_heads_max) ? 1 : 0;
    if (attn_mfma_min > 0 && ctx_len >= attn_mfma_min && kv_mode != KV_F32 && (cfg.D == 128 || cfg.D == 256) && (page & (page - 1)) == 0)
        attn_variant = ctx_len >= attn_mfma_wide_min ? 3 : 2;
    if (engine_on && engine_full && ctx_len <= eng_full_max_ctx) attn_variant = 4;    // whole-token persistent launch
    const int v = attn_variant;
    logits_gathered = false;
    // Tensor parallelism: the RCCL all-reduces / all-gathers are captured INTO the decode-step graph (one launch per
    // token instead of ~290 kernel + 74 collective launches, which is what bounds an eager TP step).  The first step
    // runs eagerly so that RCCL's lazy connection set-up happens outside a capture.  CM_TP_GRAPH=0 keeps TP eager.
    if (rccl && !rccl_warm) { rccl_warm = true; enqueue_decode_step(true); return; }
    if (use_graph && (!rccl || tp_graph)) {
        if (!graph_ok[v] && graph[v] == nullptr) {
            hipError_t e = hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed);
            if (e == hipSuccess) {
                bool threw = false;
                try { enqueue_decode_step(true); } catch (...) { threw = true; }
                e = hipStreamEndCapture(stream, &graph[v]);
                if (!threw && e == hipSuccess && graph[v]) e = hipGraphInstantiate(&graph_exec[v], graph[v], nullptr, nullptr, 0);
                graph_ok[v] = !threw && e == hipSuccess && graph_exec[v] != nullptr;
            }
            if (!graph_ok[v]) { use_graph = false; (void)hipGetLastError(); }
        }
        if (graph_ok[v]) { CM_HIP(hipGraphLaunch(graph_exec[v], stream)); return; }
    }
    enqueue_decode_step(true);
}

void Model::fetch_logits(float* out) {
    gather_logits();
    CM_HIP(hipMemcpyAsync(h_logits, logits, (size_t)V_l * tp * sizeof(float), hipMemcpyDeviceToHost, stream));
    CM_HIP(hipStreamSynchronize(stream));
    memcpy(out, h_logits, (size_t)cfg.V * sizeof(float));
}

// final norm + lm_head + arg-max over the nb rows of xb (one sequence per row): logits into logitsb[b * V], the winners into
// stb[b].next.  Shared by the batched decode step and the multi-prompt prefill.
void Model::lm_head_rows(int nb, bool want_rows) {
    const int H = cfg.H;
    hipStream_t s = stream;
    // lm_head over the vocabulary shard [v0, v0 + V_l) of this rank (TP = 1: the whole table): logits land in their
    // columns of the [nb][V] rows, the per-block maxima in this rank's [MAXB][lm_gridb] slab of pmaxb / pidxb
    const int v_eff = std::max(0, std::min(V_l, cfg.V - v0));
    const size_t slab = (size_t)MAXB * lm_gridb;
    int lmg = 0;
    if (quantized && q_lm_head.fmt != QFMT_NONE && !rccl && q_gemm_min > 0 && nb >= q_gemm_min && qx_codes != nullptr &&
        gemm_q8_ok(q_lm_head.rows(0, v_eff), nb)) {
        // large groups over a Q8_0-layout head: one int8-MFMA pass over the table (kernels_quant_gemm.hip; the rows are written in
        // place, the table's 1187 column tiles fill the chip unsplit) + the row arg-max of the bf16 GEMM branch below
        if (q_lm_head.fmt == QFMT_Q4_K || q_lm_head.fmt == QFMT_Q6_K) launch_quant_rows_q8k(xb, H, norm, cfg.eps, qx_codes, qx_scales, nb, H, s);
        else { launch_quant_rows_q8(xb, H, norm, cfg.eps, qx_codes, qx_scales, nb, H, s); if (q_capture) q_capture_rows(nb, H); }
        QGemmArgs qg{};
        qg.w = q_lm_head.rows(0, v_eff); qg.xq = qx_codes; qg.xd = qx_scales; qg.M = nb;
        if (!launch_gemm_q8(qg, EPI_STORE, logitsb + (size_t)rank * V_l, cfg.V, nullptr, 0, num_cu, s)) throw CmError(CM_ERR_UNSUPPORTED, "quantised lm_head shape");
        lmg = std::min(lm_gridb, 64);
        launch_argmax_rows(logitsb + (size_t)rank * V_l, cfg.V, v_eff, v0, pmaxb + (size_t)rank * slab, pidxb + (size_t)rank * slab, lmg, nb, s);
        if (nb > GEMV_MAXB) launch_argmax_final(pmaxb + (size_t)rank * slab, pidxb + (size_t)rank * slab, lmg, stb, ring, RING - 1, 0, nb, s);
    } else if (quantized && q_lm_head.fmt != QFMT_NONE) {
        const int cap = gemvqb_max_seqs(q_lm_head.fmt, H);
        if (cap == 0) throw CmError(CM_ERR_UNSUPPORTED, "batched decode: hidden size too large for the quantised batched GEMV");
        const int stepq = cap == 8 ? (int)GEMV_MAXB : cap;
        lmg = gemvqb_grid(q_lm_head.fmt, v_eff, H, std::min(stepq, nb), num_cu);
        for (int m0 = 0; m0 < nb; m0 += stepq) {
            GemvQBArgs q{};
            q.w = q_lm_head.rows(0, v_eff); q.x = xb + (size_t)m0 * H; q.nw = norm;
            q.y = logitsb + (size_t)m0 * cfg.V + (size_t)rank * V_l; q.res = q.y;
            q.pmax = pmaxb + (size_t)rank * slab + (size_t)m0 * lmg; q.pidx = pidxb + (size_t)rank * slab + (size_t)m0 * lmg;
            q.idx_base = v0; q.n_seq = std::min(stepq, nb - m0); q.ldx = H; q.ldy = cfg.V; q.eps = cfg.eps;
            if (!launch_gemvqb(PRO_RMSNORM, EPI_ARGMAX, q, lmg, s)) throw CmError(CM_ERR_UNSUPPORTED, "quantised lm_head format");
            if (nb > GEMV_MAXB && !(rccl && !rccl->fake)) launch_argmax_final(q.pmax, q.pidx, lmg, stb + m0, ring, RING - 1, 0, q.n_seq, s);
        }
    } else if (lm_head_gemm_min > 0 && nb >= lm_head_gemm_min && pXN_hi != nullptr && prefill_ok && nb <= chunk && v_eff % 128 == 0 && H % 32 == 0) {
        // Large groups: the head as ONE MFMA GEMM over the rows of the group (final RMSNorm rows as bf16 hi + lo, like the
        // projections of the step) + a row arg-max over the logits it wrote.  The matrix-core GEMV is issue-bound from ~17 rows
        // on: 1.25 ms per 64-row pass over the 1.24 GB table at Qwen3-8B -- two passes for 128 sequences were 2.5 ms of a 14 ms
        // round; the GEMM streams the table once.
        launch_rmsnorm_rows(xb, norm, pXN_hi, pXN_lo, nb, H, cfg.eps, s);
        GemmArgs g{};
        g.ws = pWS; g.ws_floats = gemm_ws_floats; g.wide256 = 0;
        g.A_hi = pXN_hi; g.A_lo = pXN_lo; g.W = lm_head; g.C = logitsb + (size_t)rank * V_l; g.ldc = cfg.V; g.M = nb; g.N = v_eff; g.K = H;
        if (!launch_gemm(g, GEPI_STORE, s)) throw CmError(CM_ERR_UNSUPPORTED, "gemm shape");
        lmg = std::min(lm_gridb, 64);
        launch_argmax_rows(logitsb + (size_t)rank * V_l, cfg.V, v_eff, v0, pmaxb + (size_t)rank * slab, pidxb + (size_t)rank * slab, lmg, nb, s);
        if (nb > GEMV_MAXB && !(rccl && !rccl->fake))
            launch_argmax_final(pmaxb + (size_t)rank * slab, pidxb + (size_t)rank * slab, lmg, stb, ring, RING - 1, 0, nb, s);
    } else {
        // (more than GEMV_MAXB rows -- GEMM path, one rank -- take the matrix-core GEMV in two passes, each with its own
        // grid and arg-max reduction)
        for (int m0 = 0; m0 < nb; m0 += GEMV_MAXB) {
            const int nc = std::min((int)GEMV_MAXB, nb - m0);
            GemvBArgs g{};
            g.W = lm_head; g.x = xb + (size_t)m0 * H; g.nw = norm; g.y = logitsb + (size_t)m0 * cfg.V + (size_t)rank * V_l;
            g.N = v_eff; g.K = H; g.ldw = H; g.ldx = H; g.ldy = cfg.V; g.n_seq = nc;
            g.eps = cfg.eps; g.pmax = pmaxb + (size_t)rank * slab + (size_t)m0 * lm_gridb; g.pidx = pidxb + (size_t)rank * slab + (size_t)m0 * lm_gridb;
            g.idx_base = v0;
            const bool lm_mfma = use_mfma_gemv && gemvm_ok(EPI_ARGMAX, nc, H);
            lmg = lm_mfma ? gemvm_grid(v_eff, H, num_cu, nc) : gemvb_grid(v_eff, H, num_cu);
            if (lm_mfma) launch_gemvm(PRO_RMSNORM, EPI_ARGMAX, g, lmg, s);
            else launch_gemvb(PRO_RMSNORM, EPI_ARGMAX, g, lmg, s);
            if (nb > GEMV_MAXB) launch_argmax_final(g.pmax, g.pidx, lmg, stb + m0, ring, RING - 1, 0, nc, s);
        }
    }
    if (rccl && !rccl->fake) {
        rccl->all_gather(pmaxb + (size_t)rank * slab, pmaxb, slab * sizeof(float), s);
        rccl->all_gather(pidxb + (size_t)rank * slab, pidxb, slab * sizeof(int), s);
        launch_argmax_final(pmaxb, pidxb, lmg, stb, ring, RING - 1, 0, nb, s, tp, slab);
        if (want_rows)         // full rows only when somebody reads them (host copy / device sampler)
            for (int b = 0; b < nb; ++b)
                rccl->all_gather(logitsb + (size_t)b * cfg.V + (size_t)rank * V_l, logitsb + (size_t)b * cfg.V, (size_t)V_l * sizeof(float), s);
    } else if (nb <= GEMV_MAXB) {
        launch_argmax_final(pmaxb + (size_t)rank * slab, pidxb + (size_t)rank * slab, lmg, stb, ring, RING - 1, 0, nb, s);
    }
}

// ------------------------------------------------------------------------------------
// batched decode step: step_batch_decode (backend.rs:107-121, qwen3/modeling.rs:1202-1234) without padding,
// masks or KV copies -- up to MAXB sequences share ONE pass over the weights
// ------------------------------------------------------------------------------------
void Model::ensure_batch_buffers() {
    if (stb) return;
    const int H = cfg.H, D = cfg.D;
    const size_t attn_rows = (size_t)(cfg.hybrid ? 2 * Hq_l + 2 * Hkv_l : Hq_l + 2 * Hkv_l) * D;
    ldq = (int)std::max(attn_rows, (size_t)in_proj_pad);
    const size_t at_cols = std::max((size_t)Hq_l * D, (size_t)(cfg.hybrid ? cfg.value_dim() : 0));
    stb = (StepState*)dalloc<int>(MAXB * sizeof(StepState) / sizeof(int));
    d_btb = dalloc<int>((size_t)MAXB * max_pages_per_seq);
    xb = dalloc<float>((size_t)MAXB * H);
    qkvb = dalloc<float>((size_t)MAXB * ldq);
    attnb = dalloc<float>((size_t)MAXB * at_cols);
    hbb = dalloc<float>((size_t)MAXB * I_l);
    logitsb = dalloc<float>((size_t)MAXB * cfg.V);
    part_ob = dalloc<float>((size_t)MAXB * Hq_l * std::max(nsplit, nsplit_mfma) * D);
    part_mlb = dalloc<float>((size_t)MAXB * Hq_l * std::max(nsplit, nsplit_mfma) * 2);
    if (rccl) yb = dalloc<float>((size_t)MAXB * H);
    int g = std::max(std::max(gemvb_grid(cfg.V, H, num_cu), gemvm_grid(cfg.V, H, num_cu)), gemvm_grid(cfg.V, H, num_cu, GEMV_MAXB));
    if (quantized && q_lm_head.fmt != QFMT_NONE) g = std::max(std::max(g, gemvqb_grid(q_lm_head.fmt, cfg.V, H, GEMV_MAXB, num_cu)), gemvqb_grid(q_lm_head.fmt, cfg.V, H, 8, num_cu));
    if (gu_tmp) gu_tmpb = dalloc<float>((size_t)MAXB * 2 * I_l);
    if (quantized) {                                    // activation rows of a decode group as Q8_0 blocks (kernels_quant_gemm.hip)
        const size_t kmax = std::max(std::max((size_t)H, (size_t)I_l), at_cols);
        // (the int8 prompt pass quantises ALL rows of a pass at once: the same buffers, sized for a prefill chunk, scale rows q8_xs apart)
        size_t rows = QGEMM_MAXM;
        q8_xs = QGEMM_MAXM;
        if (q8_prefill_want && q8_prefill_eligible()) { q8_xs = (prefill_chunk_rows() + 255) / 256 * 256; rows = std::max(rows, (size_t)q8_xs); }
        // (Q8_K groups for Q4_K weights: 160 bytes and 5 scales per 128 elements -- 5/4 of the Q8_0 blocks' codes and scales)
        qx_codes = (signed char*)dalloc<int>(rows * (kmax + kmax / 4) / 4 + 64);
        qx_scales = dalloc<float>((kmax / 32 + kmax / 128 + 8) * rows);
        qx_codes2 = (signed char*)dalloc<int>(rows * (kmax + kmax / 4) / 4 + 64);      // second pair: a GEMM that quantises its own output rows
        qx_scales2 = dalloc<float>((kmax / 32 + kmax / 128 + 8) * rows);               // cannot overwrite the codes it is reading
    }
    pmaxb = dalloc<float>((size_t)MAXB * g * tp);       // TP: one [MAXB][g] slab per rank (all-gathered in place)
    pidxb = dalloc<int>((size_t)MAXB * g * tp);
    lm_gridb = g;
    CM_HIP(hipHostMalloc((void**)&h_stb, MAXB * sizeof(StepState)));
    CM_HIP(hipHostMalloc((void**)&h_btb, (size_t)MAXB * max_pages_per_seq * sizeof(int32_t)));
    CM_HIP(hipHostMalloc((void**)&h_logitsb, (size_t)MAXB * cfg.V * sizeof(float)));
    CM_HIP(hipMemsetAsync(d_btb, 0, (size_t)MAXB * max_pages_per_seq * sizeof(int32_t), stream));
}

void Model::decode_batch(const int32_t* sq, const uint32_t* toks, size_t n, float* logits_out, uint32_t* greedy_out,
                         const std::function<void(size_t, int)>* after_group) {
    if (rccl && cfg.V % tp != 0) throw CmError(CM_ERR_UNSUPPORTED, "batched decode under tensor parallelism needs vocab_size divisible by tp_size");
    if (quantized && !quant_act_int)
        throw CmError(CM_ERR_UNSUPPORTED, "batched decode over quantised weights needs the integer-dot activation mode (CM_QUANT_ACT unset)");
    ensure_batch_buffers();
    const int H = cfg.H, D = cfg.D;
    hipStream_t s = stream;
    const size_t at_cols = std::max((size_t)Hq_l * D, (size_t)(cfg.hybrid ? cfg.value_dim() : 0));
    const int qkv_rows = (cfg.hybrid ? 2 * Hq_l + 2 * Hkv_l : Hq_l + 2 * Hkv_l) * D;
    // sequences per pass over the weights: up to batch_max (64) on the bf16 matrix-core GEMVs and the integer-dot quantised
    // GEMVs (groups of 8 share the stream through the L2); 8 where a VALU batched GEMV is part of the step (bf16 without the
    // matrix-core kernel, and the hybrid family's quantised layers, whose a / b gate rows stay bf16)
    size_t gsz = ((quantized && !cfg.hybrid) || (!quantized && use_mfma_gemv)) ? (size_t)batch_max : (size_t)8;
    // batch_gemm_min or more sequences (bf16 weights): the four projections of a layer run as the prompt pass's
    // MFMA GEMMs over the nb rows (M = nb, split-K; activations as bf16 hi + lo like the parity-mode prompt, whatever
    // cm_opts.prefill_split says) -- from ~17 sequences on the batched GEMVs are issue-bound, the GEMM still streams the
    // weights once.  Rows then differ from the single-sequence step by the GEMM's summation order (~1e-6), not bit for bit.
    bool gemm_b_ok = false;
    if (!quantized && batch_gemm_min > 0 && n >= (size_t)batch_gemm_min) {
        ensure_prefill_buffers();
        gemm_b_ok = prefill_ok;
        // one 128-row M tile costs the GEMM what 64 rows cost: groups of up to MAXB (batch_gemm_min <= GEMV_MAXB, so a group
        // beyond the GEMV kernels' 64 always takes the GEMM path)
        // (tensor parallelism: groups stay at GEMV_MAXB, the size the sharded lm_head's gather is laid out for)
        if (gemm_b_ok && use_mfma_gemv && !rccl) gsz = std::max(gsz, std::min<size_t>((size_t)MAXB, (size_t)chunk));
    }
    // quantised weights in the Q8_0 layout: q_gemm_min or more sequences run their projections as ONE int8-MFMA pass over the codes
    // for up to MAXB rows (kernels_quant_gemm.hip); tensors in other formats of the same model keep the batched GEMV in steps
    // (round 6: also the hybrid family -- in_proj / in_proj_z / out_proj and the gated attention's projections; its bf16 a / b gate rows
    // take the matrix-core GEMV in steps of 64 rows -- and under tensor parallelism: partial sums into yb, one all-reduce per projection;
    // TP groups stay at GEMV_MAXB, the size the sharded head's gather is laid out for)
    const bool qgemm_ok = quantized && q_gemm_min > 0 && n >= (size_t)q_gemm_min && qx_codes != nullptr;
    if (qgemm_ok) { ensure_gemm_workspace(); gsz = std::max(gsz, rccl ? (size_t)GEMV_MAXB : (size_t)MAXB); }
    for (size_t g0 = 0; g0 < n; g0 += gsz) {
        const int nb = (int)std::min<size_t>(gsz, n - g0);
        CM_HIP(hipStreamSynchronize(s));                             // pinned staging reuse
        int64_t longest = 0;
        for (int b = 0; b < nb; ++b) longest = std::max(longest, seq(sq[g0 + b]).len + 1);
        const bool heads_b = attn_heads_max > 0 && longest <= attn_heads_max && (kv_mode == KV_BF16 || kv_mode == KV_F32) && nb <= 8;
        const bool gemm_b = gemm_b_ok && nb >= batch_gemm_min && nb <= chunk;
        // y[nb, N] (+)= A[nb, K] . W^T through launch_gemm; A = the bf16 hi + lo rows produced by rows_in
        // next_norm (GEPI_RESADD into xb only): RMSNorm(xb) * next_norm -> pXN_hi / pXN_lo, the A operand of the NEXT projection,
        // written by the GEMM's split-K reduction launch (GemmArgs::norm_w) instead of a rmsnorm_rows launch of its own
        auto gm = [&](int epi, const uint16_t* A_hi, const uint16_t* A_lo, const uint16_t* W, float* C, int ldc, int N, int K, const float* next_norm = nullptr) {
            GemmArgs g{};
            g.ws = pWS; g.ws_floats = gemm_ws_floats; g.wide256 = gemm256;
            g.A_hi = A_hi; g.A_lo = A_lo; g.W = W; g.C = C; g.ldc = ldc; g.M = nb; g.N = N; g.K = K;
            g.H_hi = pHH_hi; g.H_lo = pHH_lo;
            if (next_norm) { g.norm_w = next_norm; g.norm_hi = pXN_hi; g.norm_lo = pXN_lo; g.norm_eps = cfg.eps; }
            if (!launch_gemm(g, epi, s)) throw CmError(CM_ERR_UNSUPPORTED, "gemm shape");
        };
        // row-parallel projection + residual into xb; TP: partial sums over this rank's K slice in yb (rank 0 carries the
        // residual), one all-reduce for all nb rows -- the arrangement of rp above
        auto gmr = [&](const uint16_t* A_hi, const uint16_t* A_lo, const uint16_t* W, int K, const float* next_norm = nullptr) {
            if (!rccl) { gm(GEPI_RESADD, A_hi, A_lo, W, xb, H, H, K, next_norm); return; }
            if (next_norm) throw CmError(CM_ERR_INVALID, "internal: next_norm under tensor parallelism");
            if (rank == 0 || rccl->fake) {
                CM_HIP(hipMemcpyAsync(yb, xb, (size_t)nb * H * sizeof(float), hipMemcpyDeviceToDevice, s));
                gm(GEPI_RESADD, A_hi, A_lo, W, yb, H, H, K);
            } else gm(GEPI_STORE, A_hi, A_lo, W, yb, H, H, K);
            rccl->all_reduce_sum_f32(yb, xb, (size_t)nb * H, s);
        };
        for (int b = 0; b < nb; ++b) {
            const int sidx = sq[g0 + b];
            Seq& q = seq(sidx);
            for (int c = 0; c < b; ++c) if (sq[g0 + c] == sidx) throw CmError(CM_ERR_INVALID, "sequence appears twice in one batch");
            if (toks[g0 + b] >= (uint32_t)cfg.V) throw CmError(CM_ERR_RANGE, "token id >= vocab_size");
            if (q.len + 1 > max_seq) throw CmError(CM_ERR_RANGE, "sequence longer than max_seq_len");
            ensure_pages(sidx, q.len + 1);
            StepState& hs = h_stb[b];
            memset(&hs, 0, sizeof hs);
            hs.token = toks[g0 + b]; hs.pos = (int32_t)q.len; hs.slot = sidx; hs.rsv[0] = q.rope_delta;
            memcpy(h_btb + (size_t)b * max_pages_per_seq, q.pages.data(), q.pages.size() * sizeof(int32_t));
        }
        CM_HIP(hipMemcpyAsync(stb, h_stb, (size_t)nb * sizeof(StepState), hipMemcpyHostToDevice, s));
        CM_HIP(hipMemcpyAsync(d_btb, h_btb, (size_t)nb * max_pages_per_seq * sizeof(int32_t), hipMemcpyHostToDevice, s));
        if (quantized && q_embed.fmt != QFMT_NONE) launch_embed_row_q(q_embed, stb, xb, H, cfg.V, s, nb);
        else launch_embed_row(embed, stb, xb, H, cfg.V, nb, s);
        auto gb = [&](int pro, int epi, const uint16_t* W, const float* xin, int ldx, const float* nw, float* y, int ldy, int N, int K) {
            GemvBArgs g{};
            g.W = W; g.x = xin; g.nw = nw; g.y = y; g.res = y; g.N = N; g.K = K; g.ldw = K; g.ldx = ldx; g.ldy = ldy; g.n_seq = nb;
            g.eps = cfg.eps;
            if (use_mfma_gemv && gemvm_ok(epi, nb, K)) {       // sequences as MFMA rows: the weight stream is the only cost
                if (gemvm_nkt(K) > 1 && epi == EPI_STORE) CM_HIP(hipMemsetAsync(y, 0, (size_t)nb * ldy * sizeof(float), s));
                launch_gemvm(pro, epi, g, gemvm_grid(N, K, num_cu, nb), s);
                return;
            }
            launch_gemvb(pro, epi, g, gemvb_grid(N, K, num_cu), s);
        };
        // row-parallel projection + residual into xb (o_proj / out_proj / down_proj).  TP: every rank holds partial sums
        // over its K slice in yb (rank 0 carries the residual), one all-reduce for all nb rows puts the sum back into xb
        auto rp = [&](const uint16_t* W, const float* xin, int ldx, int K) {
            if (!rccl) { gb(PRO_PLAIN, EPI_RESADD, W, xin, ldx, nullptr, xb, H, H, K); return; }
            const bool carry = rank == 0 || rccl->fake;
            GemvBArgs g{};
            g.W = W; g.x = xin; g.y = yb; g.res = xb; g.N = H; g.K = K; g.ldw = K; g.ldx = ldx; g.ldy = H; g.n_seq = nb; g.eps = cfg.eps;
            if (use_mfma_gemv && gemvm_ok(EPI_STORE, nb, K)) {
                int epi = carry ? EPI_RESADD : EPI_STORE;
                if (gemvm_nkt(K) > 1) {        // split K accumulates with atomics onto yb: seed it with the residual / zero
                    if (carry) CM_HIP(hipMemcpyAsync(yb, xb, (size_t)nb * H * sizeof(float), hipMemcpyDeviceToDevice, s));
                    else CM_HIP(hipMemsetAsync(yb, 0, (size_t)nb * H * sizeof(float), s));
                    epi = EPI_STORE;
                }
                launch_gemvm(PRO_PLAIN, epi, g, gemvm_grid(H, K, num_cu, nb), s);
            } else {
                launch_gemvb(PRO_PLAIN, carry ? EPI_RESADD : EPI_STORE, g, gemvb_grid(H, K, num_cu), s);
            }
            rccl->all_reduce_sum_f32(yb, xb, (size_t)nb * H, s);
        };
        // quantised weights: one pass over the codes for all nb sequences (gemvqb: activations quantised per sequence,
        // integer dots -- row for row the arithmetic of the single-sequence step)
        // the rows most recently quantised for the int8-MFMA path: projections that read the same input (q / k / v tensors, gate and
        // up, in_proj and in_proj_z) share one quantiser launch; forgotten at every layer and whenever a residual is added
        const float* qx_src = nullptr; const float* qx_nw = nullptr; int qx_K = 0, qx_kind = 0;      // (kind: 0 = Q8_0 blocks, 1 = Q8_K groups for Q4_K weights)
        // next_nw / next_plain: the rows this projection writes are the NEXT projection's input (RMSNorm weight next_nw, or no norm):
        // the int8-MFMA path quantises them on its reduction launch (QNext)
        auto qb = [&](int pro, int epi, const QWeight& qw, const float* xin, int ldx, const float* nw, float* y, int ldy,
                      const float* next_nw = nullptr, bool next_plain = false, QDefer* defer = nullptr) {
            if (defer) { defer->ks = 1; defer->slice = 0; defer->ws = nullptr; }
            if (qgemm_ok && nb >= q_gemm_min && (epi == EPI_STORE || epi == EPI_RESADD || epi == EPI_SILUMUL) && gemm_q8_ok(qw, nb)) {
                const float* nwe = pro == PRO_RMSNORM ? nw : nullptr;
                const int kind = (qw.fmt == QFMT_Q4_K || qw.fmt == QFMT_Q6_K) ? 1 : 0;
                if (qx_src != xin || qx_nw != nwe || qx_K != qw.K || qx_kind != kind) {
                    if (kind) launch_quant_rows_q8k(xin, ldx, nwe, cfg.eps, qx_codes, qx_scales, nb, qw.K, s);
                    else launch_quant_rows_q8(xin, ldx, nwe, cfg.eps, qx_codes, qx_scales, nb, qw.K, s);
                    qx_src = xin; qx_nw = nwe; qx_K = qw.K; qx_kind = kind;
                    if (q_capture && !kind) q_capture_rows(nb, qw.K);
                }
                QGemmArgs qg{};
                qg.w = qw; qg.xq = qx_codes; qg.xd = qx_scales; qg.M = nb;
                const int kout = epi == EPI_SILUMUL ? qw.N / 2 : qw.N;
                QNext nx{next_nw, cfg.eps, qx_codes, qx_scales, qx_codes2, qx_scales2};
                const bool want_next = (next_nw != nullptr || next_plain) && ldy == kout;
                int fused = 0;
                if (launch_gemm_q8(qg, epi, y, ldy, pWS, gemm_ws_floats, num_cu, s, want_next ? &nx : nullptr, &fused, defer)) {
                    if (fused == 2) { std::swap(qx_codes, qx_codes2); std::swap(qx_scales, qx_scales2); }      // (the other pair is the current one now)
                    if (fused) { qx_src = y; qx_nw = next_nw; qx_K = kout; qx_kind = 0; if (q_capture) q_capture_rows(nb, kout); }      // (the codes now hold the rows just written, as Q8_0 blocks)
                    else if (epi == EPI_RESADD) qx_src = nullptr;
                    return;
                }
            }
            if (epi == EPI_RESADD) qx_src = nullptr;
            const int cap = gemvqb_max_seqs(qw.fmt, qw.K);
            if (cap == 0) throw CmError(CM_ERR_UNSUPPORTED, "batched decode: K too large for the quantised batched GEMV");
            const int stepq = cap == 8 ? (int)GEMV_MAXB : cap;          // 8-sequence kernels take up to 64 (L2-sharing groups of 8)
            for (int m0 = 0; m0 < nb; m0 += stepq) {
                GemvQBArgs q{};
                q.w = qw; q.x = xin + (size_t)m0 * ldx; q.nw = nw; q.y = y + (size_t)m0 * ldy; q.res = q.y;
                q.n_seq = std::min(stepq, nb - m0); q.ldx = ldx; q.ldy = ldy; q.eps = cfg.eps;
                const int grid = gemvqb_grid(qw.fmt, qw.N, qw.K, q.n_seq, num_cu);
                if (!launch_gemvqb(pro, epi, q, grid, s)) throw CmError(CM_ERR_UNSUPPORTED, "quantised weight format");
            }
        };
        auto qrp = [&](const QWeight& qw, const float* xin, int ldx, const float* next_nw = nullptr) {        // quantised row-parallel projection + residual
            if (!rccl) { qb(PRO_PLAIN, EPI_RESADD, qw, xin, ldx, nullptr, xb, H, next_nw); return; }
            const bool carry = rank == 0 || rccl->fake;
            if (qgemm_ok && nb >= q_gemm_min && gemm_q8_ok(qw, nb)) {
                // int8 matrix cores: this rank's partial sums over its K slice into yb (rank 0 carries the residual), one all-reduce
                if (carry) CM_HIP(hipMemcpyAsync(yb, xb, (size_t)nb * H * sizeof(float), hipMemcpyDeviceToDevice, s));
                qb(PRO_PLAIN, carry ? EPI_RESADD : EPI_STORE, qw, xin, ldx, nullptr, yb, H);
                rccl->all_reduce_sum_f32(yb, xb, (size_t)nb * H, s);
                return;
            }
            const int cap = gemvqb_max_seqs(qw.fmt, qw.K);
            if (cap == 0) throw CmError(CM_ERR_UNSUPPORTED, "batched decode: K too large for the quantised batched GEMV");
            const int stepq = cap == 8 ? (int)GEMV_MAXB : cap;
            for (int m0 = 0; m0 < nb; m0 += stepq) {
                GemvQBArgs q{};
                q.w = qw; q.x = xin + (size_t)m0 * ldx; q.y = yb + (size_t)m0 * H; q.res = xb + (size_t)m0 * H;
                q.n_seq = std::min(stepq, nb - m0); q.ldx = ldx; q.ldy = H; q.eps = cfg.eps;
                const int grid = gemvqb_grid(qw.fmt, qw.N, qw.K, q.n_seq, num_cu);
                if (!launch_gemvqb(PRO_PLAIN, carry ? EPI_RESADD : EPI_STORE, q, grid, s)) throw CmError(CM_ERR_UNSUPPORTED, "quantised weight format");
            }
            rccl->all_reduce_sum_f32(yb, xb, (size_t)nb * H, s);
        };
        // large groups on the GEMM path: the RMSNorm in front of a projection is written by the split-K reduction of the projection
        // BEFORE it (o_proj / out_proj -> ln2, down_proj -> the next layer's ln1): 2 launches less per layer
        const bool fuse_norm = gemm_b && !quantized && !rccl;
        bool xn_ready = false;                                      // pXN already holds this layer's input norm
        for (int li = 0; li < cfg.L; ++li) {
            const LayerW& w = layers[(size_t)li];
            if (!(qx_src == xb && qx_nw == w.ln1)) qx_src = nullptr;       // (kept: the rows down_proj's reduction quantised for this layer)
            if (!w.full) {
                if (quantized) {
                    const int qz = cfg.conv_dim() + cfg.value_dim();
                    qb(PRO_RMSNORM, EPI_STORE, w.q_in_proj, xb, H, w.ln1, qkvb, ldq);
                    if (w.q_in_proj_z.fmt != QFMT_NONE) qb(PRO_RMSNORM, EPI_STORE, w.q_in_proj_z, xb, H, w.ln1, qkvb + w.q_in_proj.N, ldq);
                    // the a / b gate rows stay bf16: computed by the Gated-Delta-Net step itself (GdnArgs::ba_w, below); else the matrix-core
                    // GEMV in steps of <= 64 rows (a 128-row int8 group) / the VALU GEMV
                    const bool ba_fused_b = gdn_ba_fused && gdn_scratch != nullptr && cfg.Kd == 128 && cfg.Vd == 128 && H % 4 == 0;
                    for (int m0 = 0; m0 < nb && !ba_fused_b; m0 += GEMV_MAXB) {
                        GemvBArgs g{};
                        g.W = w.in_proj_ba; g.x = xb + (size_t)m0 * H; g.nw = w.ln1; g.y = qkvb + (size_t)m0 * ldq + qz; g.res = g.y;
                        g.N = 2 * cfg.NV; g.K = H; g.ldw = H; g.ldx = H; g.ldy = ldq; g.n_seq = std::min((int)GEMV_MAXB, nb - m0); g.eps = cfg.eps;
                        if (use_mfma_gemv && g.n_seq > 8 && gemvm_ok(EPI_STORE, g.n_seq, H)) {
                            if (gemvm_nkt(H) > 1) CM_HIP(hipMemset2DAsync(g.y, (size_t)ldq * sizeof(float), 0, (size_t)g.N * sizeof(float), (size_t)g.n_seq, s));
                            launch_gemvm(PRO_RMSNORM, EPI_STORE, g, gemvm_grid(g.N, H, num_cu, g.n_seq), s);
                        } else launch_gemvb(PRO_RMSNORM, EPI_STORE, g, gemvb_grid(g.N, g.K, num_cu), s);
                    }
                } else if (gemm_b) {
                    if (!xn_ready) launch_rmsnorm_rows(xb, w.ln1, pXN_hi, pXN_lo, nb, H, cfg.eps, s);
                    gm(GEPI_STORE, pXN_hi, pXN_lo, w.in_proj, qkvb, ldq, in_proj_pad, H);      // (rows padded to 128 with zero weights)
                } else
                gb(PRO_RMSNORM, EPI_STORE, w.in_proj, xb, H, w.ln1, qkvb, ldq, in_proj_rows, H);
                GdnArgs ga{};
                ga.proj = qkvb; ga.conv_w = w.conv_w; ga.conv_pool = conv_pool; ga.state_pool = state_pool;
                ga.A_log = w.A_log; ga.dt_bias = w.dt_bias; ga.gnorm_w = w.gnorm; ga.out = attnb; ga.st = stb;
                ga.proj_stride = in_proj_pad; ga.out_stride = cfg.value_dim(); ga.S = 1; ga.NV = cfg.NV; ga.vpg = cfg.NV / cfg.NK; ga.chunked = gdn_chunked ? 1 : 0;
                ga.key_dim = cfg.key_dim(); ga.layer_idx = w.gdn_idx; ga.gdn_layers = gdn_layers; ga.eps = cfg.eps;
                ga.n_seq = nb; ga.batch_proj_stride = ldq; ga.batch_out_stride = (int)at_cols;
                ga.gdn_scratch = gdn_scratch; ga.gdn_ticket = gdn_ticket;
                if (quantized && gdn_ba_fused && gdn_scratch != nullptr && cfg.Kd == 128 && cfg.Vd == 128 && H % 4 == 0) {
                    ga.ba_w = w.in_proj_ba; ga.ba_x = xb; ga.ba_nw = w.ln1; ga.ba_H = H;
                }
                launch_gdn(ga, s);
                if (quantized) qrp(w.q_out_proj, attnb, (int)at_cols);
                else if (gemm_b) {
                    launch_split_rows2d(attnb, (int)at_cols, pAT_hi, pAT_lo, nb, cfg.value_dim(), s);
                    gmr(pAT_hi, pAT_lo, w.out_proj, cfg.value_dim(), fuse_norm ? w.ln2 : nullptr);
                } else rp(w.out_proj, attnb, (int)at_cols, cfg.value_dim());
            } else {
                // (the attention path of this group, decided here because the qkv projection may leave its K-split slices to it)
                const bool full_b = Hkv_l * nb >= 2 * num_cu;
                const int64_t mf_min = full_b && attn_mfma_min > 0 ? std::min<int64_t>(attn_mfma_min, attn_mfma_min_batch) : attn_mfma_min;
                const bool mf = mf_min > 0 && longest >= mf_min && kv_mode != KV_F32 && (D == 128 || D == 256) && (page & (page - 1)) == 0;
                // nb sequences already multiply the block count: fewer token splits per sequence keep ~2 blocks per CU
                const int ns_b = mf ? std::max(attn_batch_ns_min, std::min(longest >= attn_mfma_wide_min ? nsplit_mfma : nsplit, 2 * num_cu / std::max(1, Hkv_l * nb)))
                                    : (attn_splits_force ? attn_splits_force : std::max(attn_batch_ns_min, std::min(nsplit, 2 * num_cu / std::max(1, Hkv_l * nb))));
                // the single-split matrix-core kernel of a quantised group: adds the K-split slices of the int8 qkv GEMM in its prologue (no
                // reduction launch) and writes the Q8_0 blocks of its rows for the int8 o_proj GEMM (no quantiser launch)
                const bool attn_q = attn_outq && mf && qgemm_ok && nb >= q_gemm_min && attn_decode_single_split(ns_b, D) && !cfg.hybrid &&
                                    !(heads_b && !quantized && !rccl);
                QDefer qdef{1, 0, nullptr};
                if (quantized) {
                    const bool defer_ok = attn_q && w.n_qkv == 1 && ldq == w.q_qkv[0].N && w.qkv_row0[0] == 0;
                    for (int i = 0; i < w.n_qkv; ++i) qb(PRO_RMSNORM, EPI_STORE, w.q_qkv[i], xb, H, w.ln1, qkvb + w.qkv_row0[i], ldq, nullptr, false, defer_ok ? &qdef : nullptr);
                }
                else if (gemm_b) {
                    if (!xn_ready) launch_rmsnorm_rows(xb, w.ln1, pXN_hi, pXN_lo, nb, H, cfg.eps, s);
                    gm(GEPI_STORE, pXN_hi, pXN_lo, w.qkv, qkvb, ldq, qkv_rows, H);
                } else gb(PRO_RMSNORM, EPI_STORE, w.qkv, xb, H, w.ln1, qkvb, ldq, qkv_rows, H);
                AttnDecArgs a{};
                a.qkv = qkvb; a.qnw = w.qn; a.knw = w.kn; a.cos = cos; a.sin = sin; a.st = stb; a.block_table = d_btb;
                a.kpool = kpool(li); a.vpool = vpool(li); a.part_o = part_ob; a.part_ml = part_mlb;
                a.q_off = 0; a.k_off = (cfg.hybrid ? 2 * Hq_l : Hq_l) * D; a.v_off = a.k_off + Hkv_l * D;
                a.gate = cfg.hybrid ? qkvb + (size_t)Hq_l * D : nullptr;
                a.qkv_stride = ldq; a.bt_stride = max_pages_per_seq; a.rot_dim = cfg.rot_dim;
                a.Hkv = Hkv_l; a.page = page; a.max_pages = max_pages_per_seq; a.page_bytes = page_bytes; a.eps = cfg.eps; a.scale = (float)(1.0 / std::sqrt((double)D));
                if (heads_b && !quantized && !rccl) {
                    if (!launch_attn_decode_heads(a, D, nrep, attn_ns, kv_f32, nb, s)) throw CmError(CM_ERR_UNSUPPORTED, "head_dim");
                    GemvBArgs g{};
                    g.W = w.o; g.x = part_ob; g.y = xb; g.res = xb; g.N = H; g.K = Hq_l * D; g.ldw = g.K; g.ldx = Hq_l * attn_ns * D; g.ldy = H;
                    g.n_seq = nb; g.eps = cfg.eps; g.part_ml = part_mlb; g.gate = a.gate; g.gate_stride = ldq; g.ns = attn_ns;
                    g.dshift = D == 128 ? 7 : 8;
                    launch_gemvb(PRO_ATTNCOMB, EPI_RESADD, g, gemvb_grid(g.N, g.K, num_cu), s);
                } else {
                // a group whose (kv head, sequence) pairs alone fill the chip twice takes ONE token split per sequence and the
                // matrix-core kernel from 64 tokens on (the VALU kernel's 16-lane rows pay per token, not per tile): engine at
                // max_running 128, contexts 128-256: 9567 -> 10264 tok/s
                // one split per sequence: the kernel normalises itself (no combine launch) and, in front of the o_proj GEMM of a large
                // group, writes the bf16 hi + lo planes the GEMM reads (no split_rows2d launch either)
                const bool planes = gemm_b && !quantized && attn_decode_single_split(ns_b, D);
                if (planes) { a.out1_hi = pAT_hi; a.out1_lo = pAT_lo; a.out1_cols = Hq_l * D; }
                // ... or, in front of the int8 o_proj GEMM of a quantised group, ALSO the Q8_0 blocks of the rows (no quantiser launch)
                const bool codes = attn_q && gemm_q8_ok(w.q_o, nb) && w.q_o.fmt == QFMT_Q8_0 && (Hq_l * D) % 64 == 0;
                if (codes) { a.out1_q = qx_codes; a.out1_qd = qx_scales; a.out1_cols = Hq_l * D; }
                if (qdef.ks > 1) { a.qkv = qdef.ws; a.qkv_stride = w.q_qkv[0].N; a.qkv_ns = qdef.ks; a.qkv_slice = qdef.slice; }      // (the slices of the qkv GEMM, rows N floats apart)
                if (mf) {
                    if (!launch_attn_decode_mfma(a, D, nrep, ns_b, kv_mode, attnb, (int)at_cols, nb, s)) throw CmError(CM_ERR_UNSUPPORTED, "GQA group size / head_dim");
                } else if (!launch_attn_decode(a, D, nrep, ns_b, kv_mode, attnb, (int)at_cols, nb, s)) throw CmError(CM_ERR_UNSUPPORTED, "GQA group size / head_dim");
                if (codes) { qx_src = attnb; qx_nw = nullptr; qx_K = Hq_l * D; qx_kind = 0; if (q_capture) q_capture_rows(nb, Hq_l * D); }      // (the codes hold the attention rows)
                if (quantized) qrp(w.q_o, attnb, (int)at_cols, w.ln2);
                else if (gemm_b) {
                    if (!planes) launch_split_rows2d(attnb, (int)at_cols, pAT_hi, pAT_lo, nb, Hq_l * D, s);
                    gmr(pAT_hi, pAT_lo, w.o, Hq_l * D, fuse_norm ? w.ln2 : nullptr);
                } else rp(w.o, attnb, (int)at_cols, Hq_l * D);
                }
            }
            if (quantized) {
                if (!w.split_gate_up) {
                    qb(PRO_RMSNORM, EPI_SILUMUL, w.q_gate_up, xb, H, w.ln2, hbb, I_l, nullptr, true);
                } else {
                    qb(PRO_RMSNORM, EPI_STORE, w.q_gate, xb, H, w.ln2, gu_tmpb, 2 * I_l);
                    qb(PRO_RMSNORM, EPI_STORE, w.q_up, xb, H, w.ln2, gu_tmpb + I_l, 2 * I_l);
                    launch_silu_mul(gu_tmpb, gu_tmpb + I_l, hbb, I_l, s, nb, 2 * I_l, I_l);
                }
                qrp(w.q_down, hbb, I_l, li + 1 < cfg.L ? layers[(size_t)li + 1].ln1 : nullptr);
                continue;
            }
            if (gemm_b) {
                if (!fuse_norm) launch_rmsnorm_rows(xb, w.ln2, pXN_hi, pXN_lo, nb, H, cfg.eps, s);
                gm(GEPI_SILUMUL, pXN_hi, pXN_lo, w.gate_up, nullptr, 0, 2 * I_l, H);         // -> pHH_hi / pHH_lo
                // (the next layer's input norm rides on down_proj's reduction launch)
                xn_ready = fuse_norm && li + 1 < cfg.L;
                gmr(pHH_hi, pHH_lo, w.down, I_l, xn_ready ? layers[(size_t)li + 1].ln1 : nullptr);
                continue;
            }
            gb(PRO_RMSNORM, EPI_SILUMUL, w.gate_up, xb, H, w.ln2, hbb, I_l, 2 * I_l, H);
            rp(w.down, hbb, I_l, I_l);
        }
        lm_head_rows(nb, logits_out != nullptr || after_group != nullptr);
        CM_HIP(hipMemcpyAsync(h_stb, stb, (size_t)nb * sizeof(StepState), hipMemcpyDeviceToHost, s));
        if (logits_out) CM_HIP(hipMemcpyAsync(h_logitsb, logitsb, (size_t)nb * cfg.V * sizeof(float), hipMemcpyDeviceToHost, s));
        CM_HIP(hipStreamSynchronize(s));
        for (int b = 0; b < nb; ++b) {
            seq(sq[g0 + b]).len += 1;
            if (greedy_out) greedy_out[g0 + b] = h_stb[b].next;
        }
        if (logits_out) memcpy(logits_out + g0 * (size_t)cfg.V, h_logitsb, (size_t)nb * cfg.V * sizeof(float));
        if (after_group) (*after_group)(g0, nb);
    }
}

// ------------------------------------------------------------------------------------
// forward_step (ModelBackend::forward_step, backend.rs:41; Model::forward_step model.rs:177-184)
// ------------------------------------------------------------------------------------
void Model::forward(int s, const uint32_t* ids, size_t n, size_t start_pos, float* logits_out, uint32_t* greedy_out) {
    if (n == 0 || ids == nullptr) throw CmError(CM_ERR_INVALID, "empty input");
    Seq& q = seq(s);
    if ((int64_t)start_pos > q.len) throw CmError(CM_ERR_RANGE, "start_pos beyond cached length");
    if (start_pos + n > (size_t)max_seq) throw CmError(CM_ERR_RANGE, "start_pos + n exceeds max_seq_len");
    for (size_t i = 0; i < n; ++i)
        if (ids[i] >= (uint32_t)cfg.V) throw CmError(CM_ERR_RANGE, "token id >= vocab_size");
    if ((int64_t)start_pos < q.len) seq_truncate(s, start_pos);   // re-prefill over an old suffix
    // appending into a page shared with a fork is not allowed: fork already copied the partial page
    ensure_pages(s, (int64_t)(start_pos + n));
    activate(s);
    bool use_prefill = false, st_fresh = false;
    // quantised weights: each matrix is dequantised to a bf16 scratch in front of its MFMA GEMM (CM_QUANT_PREFILL=0:
    // token-serial, i.e. the decode kernels' integer-dot arithmetic for the prompt too)
    if (n >= 2 && (!quantized || quant_prefill) && !no_prefill) {
        ensure_prefill_buffers();
        use_prefill = prefill_ok;
    }
    if (use_prefill) {
        prefill(ids, n, start_pos);
    } else {
        for (int attempt = 0; attempt < 2; ++attempt) {
            for (size_t i = 0; i < n; ++i) {     // token-serial path (also the parity cross-check of prefill)
                launch_set_state(st, ids[i], (int32_t)(start_pos + i), s, q.rope_delta, stream);
                run_decode_step(true, (int64_t)(start_pos + i + 1));
            }
            if (!engine_on) break;
            // ONE copy of the step state serves the time-out check and the greedy token (no second host sync per token)
            CM_HIP(hipMemcpyAsync(h_st, st, sizeof(StepState), hipMemcpyDeviceToHost, stream));
            CM_HIP(hipStreamSynchronize(stream));
            if (h_st->rsv[2] == 0) { st_fresh = true; break; }
            // a timed-out persistent launch switched itself off.  Dense family: the same steps again on the per-projection
            // launches (they rewrite the K/V rows of these positions -- appends are idempotent).  Hybrid family: the failed
            // attempt already advanced the conv window and the delta-rule state of this sequence, possibly on garbage; a
            // replay would advance them a second time, so there is no silent retry -- the call fails and the caller has to
            // cm_seq_truncate(seq, 0) / re-prefill (the persistent path is off for the handle from here on).
            if (cfg.hybrid) { q.len = (int64_t)start_pos; hybrid_engine_abort(); }
            (void)engine_failed();
        }
    }
    q.len = (int64_t)(start_pos + n);
    if (greedy_out) {
        if (!st_fresh) {
            CM_HIP(hipMemcpyAsync(h_st, st, sizeof(StepState), hipMemcpyDeviceToHost, stream));
            CM_HIP(hipStreamSynchronize(stream));
        }
        *greedy_out = h_st->next;
    }
    if (logits_out) fetch_logits(logits_out);
    if (!greedy_out && !logits_out) CM_HIP(hipStreamSynchronize(stream));
}

// ------------------------------------------------------------------------------------
// ModelForCausalLM::generate (based.rs:7-31; qwen3/model.rs:275-349)
// ------------------------------------------------------------------------------------
static bool is_eos(const cm_gen_config& g, uint32_t t) {
    for (int i = 0; i < 4; ++i) if (g.eos_token_id[i] >= 0 && (uint32_t)g.eos_token_id[i] == t) return true;
    return false;
}

void Model::generate(const uint32_t* prompt, size_t n_prompt, const cm_gen_config* gc, uint32_t* out, size_t* n_out,
                     cm_token_cb cb, void* user) {
    if (!gc || !out || !n_out) throw CmError(CM_ERR_INVALID, "null argument");
    if (n_prompt == 0) throw CmError(CM_ERR_INVALID, "empty prompt");
    const cm_gen_config g = *gc;
    if (n_prompt + g.max_new_tokens > (size_t)max_seq) throw CmError(CM_ERR_RANGE, "prompt + max_new_tokens exceeds max_seq_len");
    seq_truncate(0, 0);                                   // self.clear_kv_cache() (model.rs:281)
    size_t n = n_prompt;
    memcpy(out, prompt, n_prompt * sizeof(uint32_t));
    *n_out = n;
    if (g.max_new_tokens == 0) return;
    // LogitsProcessor::new(seed, temperature, top_p) (model.rs:289-297): None or ~0 temperature == ArgMax
    const bool sampling = g.temperature >= 1e-7f;
    const bool penalty = (std::fabs(g.repetition_penalty - 1.0f) >= 1.1920929e-7f && g.repetition_penalty > 0.f) ||
                         g.frequency_penalty != 0.f || g.presence_penalty != 0.f;
    const bool device_pick = sampling || penalty;
    cm_sample_params sp{};
    sp.temperature = sampling ? g.temperature : 0.f;
    sp.top_p = g.top_p; sp.top_k = g.top_k;
    sp.repetition_penalty = (std::fabs(g.repetition_penalty - 1.0f) >= 1.1920929e-7f) ? g.repetition_penalty : 1.0f;
    sp.frequency_penalty = g.frequency_penalty; sp.presence_penalty = g.presence_penalty;
    sp.repeat_last_n = g.repeat_last_n;
    sp.seed = ((uint64_t)g.seed_hi << 32) | g.seed_lo;
    if (sp.seed == 0) sp.seed = 299792458ull;
    // penalties (apply_repeat_penalty over the last repeat_last_n tokens, model.rs:306-315), top-k / top-p /
    // temperature and the draw all run on the logits resident in HBM (model_sample.cpp)
    sp.repeat_last_n = 0;
    auto device_sample = [&]() -> uint32_t {
        const size_t w = std::min(n, (size_t)g.repeat_last_n);       // tokens.len().saturating_sub(repeat_last_n) (model.rs:307)
        const uint32_t t = sample(sp, out + (n - w), w, true);
        ++sp.draw;
        return t;
    };

    // step 0: whole prompt at start_pos 0 (model.rs:299-304)
    uint32_t tok = 0;
    if (device_pick) { forward(0, prompt, n_prompt, 0, nullptr, nullptr); tok = device_sample(); }
    else forward(0, prompt, n_prompt, 0, nullptr, &tok);
    size_t produced = 0;
    bool stop = false;
    auto emit = [&](uint32_t t) {
        out[n++] = t; ++produced;
        if (is_eos(g, t)) { stop = true; return; }          // EOS is pushed, then loop breaks (model.rs:318-327)
        if (cb && cb(user, t) != 0) stop = true;
    };
    emit(tok);
    if (device_pick) {
        while (!stop && produced < g.max_new_tokens) {
            forward(0, &out[n - 1], 1, n - 1, nullptr, nullptr);
            emit(device_sample());
        }
    } else {
        // device-chained greedy decode: the arg-max kernel feeds the next step's token/pos in HBM
        const size_t chunk = std::min<size_t>(g.sync_every ? g.sync_every : 1, (size_t)RING);   // the device token ring holds RING steps
        Seq& q = seq(0);
        while (!stop && produced < g.max_new_tokens) {
            const size_t want = std::min(chunk, (size_t)g.max_new_tokens - produced);
            ensure_pages(0, (int64_t)(q.len + (int64_t)want));
            activate(0);
            launch_set_state(st, out[n - 1], (int32_t)q.len, 0, q.rope_delta, stream);
            const uint32_t ring0 = ring_count;
            for (size_t i = 0; i < want; ++i) run_decode_step(true, q.len + (int64_t)i + 1);
            CM_HIP(hipMemcpyAsync(h_ring, ring, RING * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            CM_HIP(hipStreamSynchronize(stream));
            if (engine_on && cfg.hybrid) {                  // recurrent state cannot be replayed (see forward()): fail loudly
                CM_HIP(hipMemcpy(h_st, st, sizeof(StepState), hipMemcpyDeviceToHost));
                if (h_st->rsv[2] != 0) hybrid_engine_abort();
            } else if (engine_on && engine_failed()) continue;   // nothing of this chunk was emitted: the same steps again, launch path
            size_t used = 0;
            for (size_t i = 0; i < want && !stop; ++i) { emit(h_ring[(ring0 + i) & (RING - 1)]); ++used; }
            q.len += (int64_t)used;     // tokens decoded past an EOS/stop stay beyond len and are overwritten later
        }
    }
    *n_out = n;
}

void Model::bench_decode(uint32_t first, size_t k, uint32_t* toks, float* ms) {
    if (k == 0 || k > (size_t)RING) throw CmError(CM_ERR_INVALID, "k must be in 1..4096");
    Seq& q = seq(0);
    ensure_pages(0, q.len + (int64_t)k);
    activate(0);
    launch_set_state(st, first, (int32_t)q.len, 0, q.rope_delta, stream);
    const uint32_t ring0 = ring_count;
    hipEvent_t e0, e1;
    CM_HIP(hipEventCreate(&e0));
    CM_HIP(hipEventCreate(&e1));
    CM_HIP(hipEventRecord(e0, stream));
    for (size_t i = 0; i < k; ++i) run_decode_step(true, q.len + (int64_t)i + 1);
    CM_HIP(hipEventRecord(e1, stream));
    CM_HIP(hipEventSynchronize(e1));
    float t = 0.f;
    CM_HIP(hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (ms) *ms = t;
    q.len += (int64_t)k;
    engine_check();
    if (toks) {
        CM_HIP(hipMemcpy(h_ring, ring, RING * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < k; ++i) toks[i] = h_ring[(ring0 + i) & (RING - 1)];
    }
}

void Model::bench_kernel(const std::string& which, size_t iters, float* ms, uint64_t* bytes) {
    if (iters == 0) throw CmError(CM_ERR_INVALID, "iters == 0");
    const int H = cfg.H, D = cfg.D;
    const int v_eff = std::max(0, std::min(V_l, cfg.V - v0));
    uint64_t b = 0;
    // "pgemm_<qkv|o|gate_up|down>@<M>": one prompt-pass GEMM of a layer over M rows of random activations, through launch_gemm
    // exactly as prefill() calls it (cycling through the layers' weights); *bytes_out = useful flops 2 M N K of one launch
    int pg_M = 0;
    std::string pg;
    if (which.rfind("pgemm_", 0) == 0) {
        const size_t at = which.find('@');
        if (at == std::string::npos) throw CmError(CM_ERR_INVALID, "pgemm_<proj>@<rows>");
        pg = which.substr(6, at - 6);
        pg_M = atoi(which.c_str() + at + 1);
        ensure_prefill_buffers();
        if (!prefill_ok || quantized || pg_M < 1 || pg_M > chunk) throw CmError(CM_ERR_INVALID, "pgemm: rows must be in 1..prefill_chunk (bf16 weights)");
        const uint32_t ts = fmix32(0x1234567u);
        launch_synth_fill(pXN_hi, (size_t)H, pg_M, H, 0, 0, H, ts, 0.01f, 0.f, stream);
        launch_synth_fill(pXN_lo, (size_t)H, pg_M, H, 0, 0, H, ts + 1, 0.00004f, 0.f, stream);
        launch_synth_fill(pAT_hi, (size_t)Hq_l * D, pg_M, Hq_l * D, 0, 0, Hq_l * D, ts + 2, 0.01f, 0.f, stream);
        launch_synth_fill(pAT_lo, (size_t)Hq_l * D, pg_M, Hq_l * D, 0, 0, Hq_l * D, ts + 3, 0.00004f, 0.f, stream);
        launch_synth_fill(pHH_hi, (size_t)I_l, pg_M, I_l, 0, 0, I_l, ts + 4, 0.01f, 0.f, stream);
        launch_synth_fill(pHH_lo, (size_t)I_l, pg_M, I_l, 0, 0, I_l, ts + 5, 0.00004f, 0.f, stream);
    }
    auto one = [&](size_t i) {
        size_t li = i % (size_t)cfg.L;
        if (pg_M > 0) {
            while (!layers[li].full) li = (li + 1) % (size_t)cfg.L;
            const LayerW& w = layers[li];
            const bool sp2 = prefill_split2;
            GemmArgs g{};
            g.ws = pWS; g.ws_floats = gemm_ws_floats; g.wide256 = gemm256; g.M = pg_M;
            int epi = GEPI_STORE;
            if (pg == "qkv") { g.A_hi = pXN_hi; g.A_lo = sp2 ? pXN_lo : nullptr; g.W = w.qkv; g.C = pQKV; g.N = (Hq_l + 2 * Hkv_l) * D; g.K = H; g.ldc = g.N; }
            else if (pg == "o") { g.A_hi = pAT_hi; g.A_lo = sp2 ? pAT_lo : nullptr; g.W = w.o; g.C = pX; g.N = H; g.K = Hq_l * D; g.ldc = H; epi = GEPI_RESADD; }
            else if (pg == "gate_up") { g.A_hi = pXN_hi; g.A_lo = sp2 ? pXN_lo : nullptr; g.W = w.gate_up; g.N = 2 * I_l; g.K = H; g.H_hi = pHH_hi; g.H_lo = sp2 ? pHH_lo : nullptr; epi = GEPI_SILUMUL; }
            else if (pg == "down") { g.A_hi = pHH_hi; g.A_lo = sp2 ? pHH_lo : nullptr; g.W = w.down; g.C = pX; g.N = H; g.K = I_l; g.ldc = H; epi = GEPI_RESADD; }
            else throw CmError(CM_ERR_INVALID, "pgemm: qkv | o | gate_up | down");
            if (!launch_gemm(g, epi, stream)) throw CmError(CM_ERR_UNSUPPORTED, "gemm shape");
            b = 2ull * (uint64_t)g.M * (uint64_t)g.N * (uint64_t)g.K;
            return;
        }
        if ((which == "qkv" || which == "o") && !layers[li].full) li = (size_t)(cfg.interval - 1);
        const LayerW& w = layers[li];
        if (quantized) {
            GemvQArgs q{};
            q.eps = cfg.eps; q.act_int = quant_act_int ? 1 : 0;
            int pro = PRO_PLAIN, epi = EPI_RESADD;
            if (which == "qkv") { q.w = w.q_qkv[0]; q.x = x; q.nw = w.ln1; q.y = qkv; pro = PRO_RMSNORM; epi = EPI_STORE; }
            else if (which == "o") { q.w = w.q_o; q.x = attn; q.y = y; q.res = x; }
            else if (which == "gate_up" && !w.split_gate_up) { q.w = w.q_gate_up; q.x = x; q.nw = w.ln2; q.y = hbuf; pro = PRO_RMSNORM; epi = EPI_SILUMUL; }
            else if (which == "down") { q.w = w.q_down; q.x = hbuf; q.y = y; q.res = x; }
            else if (which == "lm_head" && q_lm_head.fmt != QFMT_NONE) { q.w = q_lm_head; q.x = x; q.nw = norm; q.y = logits; q.pmax = pmax; q.pidx = pidx; pro = PRO_RMSNORM; epi = EPI_ARGMAX; }
            else throw CmError(CM_ERR_INVALID, "unknown / unavailable kernel name for quantised weights");
            const int grid = which == "lm_head" ? lm_grid : gemvq_grid(q.w.N, num_cu, q.w.fmt);
            if (!launch_gemvq(pro, epi, q, grid, stream)) throw CmError(CM_ERR_UNSUPPORTED, "quantised weight format");
            b = q.w.bytes() + (uint64_t)q.w.K * 4 + (q.nw ? (uint64_t)q.w.K * 4 : 0);
            return;
        }
        if (which == "chain") {      // the persistent chain launch of a layer that has a successor (4 phases)
            if (!engine_on) throw CmError(CM_ERR_INVALID, "the persistent chain kernel is not active on this model");
            const uint64_t Ko = (uint64_t)Hq_l * D, qrows = (uint64_t)(Hq_l + 2 * Hkv_l) * D;
            const uint64_t wl = ((uint64_t)H * Ko + 2ull * I_l * H + (uint64_t)H * I_l + qrows * H) * 2;     // one layer's weights, bf16
            if (engine_full) {       // the whole-token launch: every layer's weights + the KV read at the current context
                if (!launch_engine(engine_args_full(), num_cu, stream)) throw CmError(CM_ERR_DEVICE, "persistent decode kernel launch");
                b = (wl + 2ull * H * 4 + 2ull * D * 4) * (uint64_t)cfg.L + 2ull * cfg.L * Hkv_l * (uint64_t)seq(0).len * kv_row_bytes + 2ull * H * 4;
                if (eng_head) b += (uint64_t)std::max(0, std::min(V_l, cfg.V - v0)) * H * 2 + (uint64_t)H * 4 + (uint64_t)H * 2;      // + lm_head, final norm, embedding row
                return;
            }
            const int lc = cfg.L > 1 ? (int)(i % (size_t)(cfg.L - 1)) : 0;
            if (!launch_engine(engine_args(lc), num_cu, stream)) throw CmError(CM_ERR_DEVICE, "persistent decode kernel launch");
            if (cfg.hybrid) {        // out_proj / o_proj + gate||up + down_proj of layer lc + the in-projection of layer lc + 1 (in_proj or QKV)
                const uint64_t nxt = cfg.L > 1 ? (uint64_t)(layers[(size_t)lc + 1].full ? (2 * Hq_l + 2 * Hkv_l) * D : in_proj_rows) : 0;
                b = ((uint64_t)H * Ko + 2ull * I_l * H + (uint64_t)H * I_l + nxt * H) * 2 + Ko * 4 + (uint64_t)H * 4 + 2ull * H * 4;
                return;
            }
            b = (cfg.L > 1 ? wl : wl - qrows * H * 2) + Ko * 4 + (uint64_t)H * 4 + 2ull * H * 4;            // + attn in, x in/out, 2 norm vectors
            return;
        }
        GemvArgs g{};
        g.eps = cfg.eps;
        if (which == "qkv") {
            g.W = w.qkv; g.x = x; g.nw = w.ln1; g.y = qkv; g.N = (Hq_l + 2 * Hkv_l) * D; g.K = H; g.ldw = H;
            launch_gemv(PRO_RMSNORM, EPI_STORE, g, gemv_grid(g.N, g.K, num_cu), stream);
        } else if (which == "o") {
            g.W = w.o; g.x = attn; g.y = y; g.res = x; g.N = H; g.K = Hq_l * D; g.ldw = g.K;
            launch_gemv(PRO_PLAIN, EPI_RESADD, g, gemv_grid(g.N, g.K, num_cu), stream);
        } else if (which == "gate_up") {
            g.W = w.gate_up; g.x = x; g.nw = w.ln2; g.y = hbuf; g.N = 2 * I_l; g.K = H; g.ldw = H;
            launch_gemv(PRO_RMSNORM, EPI_SILUMUL, g, gemv_grid(g.N, g.K, num_cu), stream);
        } else if (which == "down") {
            g.W = w.down; g.x = hbuf; g.y = y; g.res = x; g.N = H; g.K = I_l; g.ldw = I_l;
            launch_gemv(PRO_PLAIN, EPI_RESADD, g, gemv_grid(g.N, g.K, num_cu), stream);
        } else if (which == "lm_head") {
            g.W = lm_head; g.x = x; g.nw = norm; g.y = logits + (size_t)rank * V_l; g.N = v_eff; g.K = H; g.ldw = H;
            g.pmax = pmax + (size_t)rank * lm_grid; g.pidx = pidx + (size_t)rank * lm_grid; g.idx_base = v0;
            launch_gemv(PRO_RMSNORM, EPI_ARGMAX, g, lm_grid, stream);
        } else throw CmError(CM_ERR_INVALID, "unknown kernel name");
        b = (uint64_t)g.N * g.K * 2 + (uint64_t)g.K * 4 + (g.nw ? (uint64_t)g.K * 2 : 0);
    };
    for (size_t i = 0; i < std::min<size_t>(iters, 8); ++i) one(i);      // warm-up
    hipEvent_t e0, e1;
    CM_HIP(hipEventCreate(&e0));
    CM_HIP(hipEventCreate(&e1));
    CM_HIP(hipEventRecord(e0, stream));
    for (size_t i = 0; i < iters; ++i) one(i);
    CM_HIP(hipEventRecord(e1, stream));
    CM_HIP(hipEventSynchronize(e1));
    float t = 0.f;
    CM_HIP(hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (ms) *ms = t / (float)iters;
    if (bytes) *bytes = b;
    engine_check();
}

// test hook (q_capture): the Q8_0 activation rows currently in qx_codes / qx_scales, appended to q_cap as floats
void Model::q_capture_rows(int nb, int K, int xs) {
    std::vector<signed char> c((size_t)nb * K);
    std::vector<float> d((size_t)(K / 32) * xs);
    CM_HIP(hipStreamSynchronize(stream));
    CM_HIP(hipMemcpy(c.data(), qx_codes, c.size(), hipMemcpyDeviceToHost));
    CM_HIP(hipMemcpy(d.data(), qx_scales, d.size() * sizeof(float), hipMemcpyDeviceToHost));
    size_t at = q_cap.size();
    q_cap.resize(at + 2 + c.size() + (size_t)nb * (K / 32));
    float* o = q_cap.data() + at;
    *o++ = (float)K; *o++ = (float)nb;
    for (size_t i = 0; i < c.size(); ++i) *o++ = (float)c[i];
    for (int m = 0; m < nb; ++m)
        for (int b = 0; b < K / 32; ++b) *o++ = d[(size_t)b * xs + m];
}

// test hook: rows x [m][k] through the int8-MFMA GEMM of a quantised projection (row quantiser of the weight's vec-dot type + launch_gemm_q8,
// plain input, store epilogue): y [m][n].  The kernel-level check of the decode-group / prompt GEMM against the oracle's vecdot rows.
void Model::debug_qgemm(int layer, const std::string& which, const float* xh, size_t m, size_t k, float* yh, size_t n) {
    if (!quantized) throw CmError(CM_ERR_INVALID, "model has no quantised weights");
    QWeight w;
    if (which == "lm_head") w = q_lm_head;
    else {
        if (layer < 0 || layer >= cfg.L) throw CmError(CM_ERR_RANGE, "layer out of range");
        const LayerW& lw = layers[(size_t)layer];
        if (which == "qkv0") w = lw.q_qkv[0];
        else if (which == "qkv1" && lw.n_qkv > 1) w = lw.q_qkv[1];
        else if (which == "qkv2" && lw.n_qkv > 2) w = lw.q_qkv[2];
        else if (which == "o") w = lw.q_o;
        else if (which == "gate_up") w = lw.q_gate_up;
        else if (which == "gate") w = lw.q_gate;
        else if (which == "up") w = lw.q_up;
        else if (which == "down") w = lw.q_down;
    }
    if (w.fmt == QFMT_NONE) throw CmError(CM_ERR_INVALID, "no such quantised projection: " + which);
    if ((size_t)w.K != k || (size_t)w.N != n || m < 1) throw CmError(CM_ERR_RANGE, "debug_qgemm: expected k=" + std::to_string(w.K) + " n=" + std::to_string(w.N));
    if (!gemm_q8_ok(w, (int)m)) throw CmError(CM_ERR_UNSUPPORTED, "debug_qgemm: this projection is not on the int8-MFMA GEMM (format / shape)");
    ensure_gemm_workspace();
    const int xs = m > (size_t)QGEMM_MAXM ? (int)((m + 255) / 256 * 256) : (int)QGEMM_MAXM;
    float *dx = nullptr, *dy = nullptr, *dsc = nullptr;
    signed char* dq = nullptr;
    CM_HIP(hipMalloc((void**)&dx, m * k * sizeof(float)));
    CM_HIP(hipMalloc((void**)&dy, m * n * sizeof(float)));
    CM_HIP(hipMalloc((void**)&dq, m * (k + k / 4) + 256));
    CM_HIP(hipMalloc((void**)&dsc, ((k / 128) * 5 + k / 32 + 8) * (size_t)xs * sizeof(float)));
    CM_HIP(hipMemcpyAsync(dx, xh, m * k * sizeof(float), hipMemcpyHostToDevice, stream));
    if (w.fmt == QFMT_Q4_K || w.fmt == QFMT_Q6_K) launch_quant_rows_q8k(dx, (int)k, nullptr, cfg.eps, dq, dsc, (int)m, (int)k, stream, xs);
    else launch_quant_rows_q8(dx, (int)k, nullptr, cfg.eps, dq, dsc, (int)m, (int)k, stream, xs);
    QGemmArgs qg{};
    qg.w = w; qg.xq = dq; qg.xd = dsc; qg.M = (int)m; qg.xs = xs;
    const bool ok = launch_gemm_q8(qg, EPI_STORE, dy, (int)n, pWS, gemm_ws_floats, num_cu, stream);
    if (ok) CM_HIP(hipMemcpyAsync(yh, dy, m * n * sizeof(float), hipMemcpyDeviceToHost, stream));
    CM_HIP(hipStreamSynchronize(stream));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dq); (void)hipFree(dsc);
    if (!ok) throw CmError(CM_ERR_UNSUPPORTED, "debug_qgemm: launch refused");
}

void Model::debug_qgemv(int layer, const std::string& which, const float* xh, size_t k, float* yh, size_t n) {
    if (!quantized) throw CmError(CM_ERR_INVALID, "model has no quantised weights");
    QWeight w;
    if (which == "lm_head") w = q_lm_head;
    else {
        if (layer < 0 || layer >= cfg.L) throw CmError(CM_ERR_RANGE, "layer out of range");
        const LayerW& lw = layers[(size_t)layer];
        if (which == "qkv0") w = lw.q_qkv[0];
        else if (which == "qkv1" && lw.n_qkv > 1) w = lw.q_qkv[1];
        else if (which == "qkv2" && lw.n_qkv > 2) w = lw.q_qkv[2];
        else if (which == "o") w = lw.q_o;
        else if (which == "gate_up") w = lw.q_gate_up;
        else if (which == "gate") w = lw.q_gate;
        else if (which == "up") w = lw.q_up;
        else if (which == "down") w = lw.q_down;
    }
    if (w.fmt == QFMT_NONE) throw CmError(CM_ERR_INVALID, "no such quantised projection: " + which);
    if ((size_t)w.K != k || (size_t)w.N != n) throw CmError(CM_ERR_RANGE, "debug_qgemv: expected k=" + std::to_string(w.K) + " n=" + std::to_string(w.N));
    float *dx = nullptr, *dy = nullptr;
    CM_HIP(hipMalloc((void**)&dx, k * sizeof(float)));
    CM_HIP(hipMalloc((void**)&dy, n * sizeof(float)));
    CM_HIP(hipMemcpyAsync(dx, xh, k * sizeof(float), hipMemcpyHostToDevice, stream));
    GemvQArgs q{};
    q.w = w; q.x = dx; q.y = dy; q.eps = cfg.eps; q.act_int = quant_act_int ? 1 : 0;
    const bool ok = launch_gemvq(PRO_PLAIN, EPI_STORE, q, gemvq_grid(w.N, num_cu, w.fmt), stream);
    if (ok) CM_HIP(hipMemcpyAsync(yh, dy, n * sizeof(float), hipMemcpyDeviceToHost, stream));
    CM_HIP(hipStreamSynchronize(stream));
    (void)hipFree(dx); (void)hipFree(dy);
    if (!ok) throw CmError(CM_ERR_UNSUPPORTED, "quantised weight format");
}

void Model::debug_fill_kv(size_t ctx, uint64_t seed) {
    seq_truncate(0, 0);
    ensure_pages(0, (int64_t)ctx);
    activate(0);
    Seq& q = seq(0);
    for (int li = 0; li < cfg.L; ++li) {
        if (!cfg.layer_full(li)) continue;
        if (kvq()) {
            const size_t code_bytes = (size_t)Hkv_l * page * kv_row_bytes;
            launch_kv_fill_quant(kpool(li), d_bt, (int)q.pages.size(), page_bytes, code_bytes, fmix32((uint32_t)seed * 2654435761u + (uint32_t)li * 2 + 1), stream);
            launch_kv_fill_quant(vpool(li), d_bt, (int)q.pages.size(), page_bytes, code_bytes, fmix32((uint32_t)seed * 2654435761u + (uint32_t)li * 2 + 2), stream);
            continue;
        }
        launch_kv_fill(kpool(li), kv_mode, d_bt, (int)q.pages.size(), page_elems, fmix32((uint32_t)seed * 2654435761u + (uint32_t)li * 2 + 1), (size_t)page * cfg.D, cfg.Hkv, kvh0, stream);
        launch_kv_fill(vpool(li), kv_mode, d_bt, (int)q.pages.size(), page_elems, fmix32((uint32_t)seed * 2654435761u + (uint32_t)li * 2 + 2), (size_t)page * cfg.D, cfg.Hkv, kvh0, stream);
    }
    CM_HIP(hipStreamSynchronize(stream));
    q.len = (int64_t)ctx;
}

}  // namespace cm
