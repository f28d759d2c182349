// extern "C" surface (include/crane_mi355.h): translates C++ exceptions into status codes
// + per-handle error strings (anyhow::Result on the Rust side).
#include <cstring>
#include <string>
#include <vector>

#include "model.h"
#include "safetensors.h"
#include "tp.h"
#include "tp_group.h"

using cm::CmError;
using cm::Model;

static_assert(sizeof(cm_opts) == 96, "cm_opts grew: the round-4 fields live in the former reserved words");

// m: the replica, the SPMD rank, or rank 0 of an in-process tensor-parallel group (grp: ranks 1 .. n-1 + their threads)
struct cm_model {
    Model m;
    std::unique_ptr<cm::TpGroup> grp;
    // rank 0's transport points into the group's shared state: it goes first, then the worker threads and the peer ranks
    ~cm_model() { (void)hipSetDevice(m.dev); m.rccl.reset(); grp.reset(); }
};

namespace cm {
long peer_selftest(int n, int device, int iters, int count);      // tp_group.cpp
Model& model_of(cm_model* h) { return h->m; }
TpGroup* group_of(cm_model* h) { return h->grp.get(); }
}

static thread_local std::string g_err;

template <typename F>
static int guard(cm_model* h, F&& f) {
    try {
        if (h) (void)hipSetDevice(h->m.dev);
        f();
        return CM_OK;
    } catch (const CmError& e) {
        if (h) h->m.err = e.what(); else g_err = e.what();
        return e.code;
    } catch (const std::exception& e) {
        if (h) h->m.err = e.what(); else g_err = e.what();
        return CM_ERR_INVALID;
    } catch (...) {
        if (h) h->m.err = "unknown error"; else g_err = "unknown error";
        return CM_ERR_INVALID;
    }
}

// f(Model&, primary) on every rank the handle owns: the one Model of a replica / SPMD rank, or all ranks of an in-process
// group at once (rank 0 = primary on the calling thread).  Results come from the primary; the other ranks are handed scratch
// outputs -- they must take the same path through the collectives.
template <typename F>
static int guard_all(cm_model* h, F&& f) {
    return guard(h, [&] {
        if (!h->grp) { f(h->m, true); return; }
        h->grp->run([&](int r) { f(h->grp->model(r), r == 0); });
    });
}
// scratch output of a non-primary rank
template <typename T>
static T* out_or(bool primary, T* real, std::vector<T>& tmp, size_t n) {
    if (primary || !real) return real;
    tmp.resize(n);
    return tmp.data();
}

// cm_opts.isq, else CRANE_ISQ (qwen3_5/model.rs:615-626 isq_from_env): q8_0, q4_0, q5_0 -- the three 32-weight formats whose
// reference quantisers are restated (oracle/gguf_oracle.py).  The K-quant quantisers of ggml (make_qkx2_quants search) are not, so the
// K-quant names are refused instead of being quantised with a different search
static void apply_isq(cm::Model& m) {
    uint32_t isq = m.opts.isq;
    if (isq == 0) {
        if (const char* e = getenv("CRANE_ISQ")) {
            std::string v(e);
            for (char& c : v) c = (char)tolower((unsigned char)c);
            if (v == "q8_0") isq = CM_ISQ_Q8_0;
            else if (v == "q4_0") isq = CM_ISQ_Q4_0;
            else if (v == "q5_0") isq = CM_ISQ_Q5_0;
            else if (!v.empty()) throw CmError(CM_ERR_UNSUPPORTED, "CRANE_ISQ='" + v + "': q8_0, q4_0 and q5_0 are implemented");
        }
    }
    if (isq == 0) return;
    if (isq == CM_ISQ_Q8_0) m.isq_q8_0(8);
    else if (isq == CM_ISQ_Q4_0) m.isq_q8_0(4);
    else if (isq == CM_ISQ_Q5_0) m.isq_q8_0(5);
    else throw CmError(CM_ERR_UNSUPPORTED, "cm_opts.isq: CM_ISQ_Q8_0, CM_ISQ_Q4_0 and CM_ISQ_Q5_0 are implemented");
}

// Model::new for every rank the options ask for.  CM_TP_IN_PROCESS: the ranks load their shards concurrently (one thread per
// device); the transport's rendezvous happens inside alloc_runtime.
template <typename LOAD>
static void create_ranks(cm_model* h, const std::string& cfg, const cm_opts* opts, LOAD&& load) {
    if (!opts || opts->tp_mode == CM_TP_SPMD) {
        h->m.init_common(cfg, opts);
        load(h->m);
        h->m.alloc_runtime();
        return;
    }
    if (opts->tp_mode != CM_TP_IN_PROCESS) throw CmError(CM_ERR_INVALID, "cm_opts.tp_mode");
    if (opts->abi_version != 0 && opts->abi_version < 3) throw CmError(CM_ERR_INVALID, "cm_opts.tp_mode needs CM_ABI_VERSION >= 3");
    if (opts->debug_flags & (CM_DEBUG_TP_LOCAL | CM_DEBUG_FORCE_RCCL)) throw CmError(CM_ERR_INVALID, "debug_flags do not apply to an in-process group");
    h->grp.reset(new cm::TpGroup(opts->tp_size, opts->tp_devices, opts->device, opts->tp_collective));
    cm::TpGroup& g = *h->grp;
    g.rank0 = &h->m;
    g.run([&](int r) {
        Model& m = g.model(r);
        cm_opts o = *opts;
        o.tp_mode = CM_TP_SPMD; o.tp_devices = nullptr; o.tp_unique_id = nullptr;
        o.tp_rank = r; o.device = g.shared.devs[(size_t)r];
        m.peer_shared = &g.shared;
        m.init_common(cfg, &o);
        load(m);
        m.alloc_runtime();
    });
}

extern "C" {

int cm_create(const char* model_dir, const cm_opts* opts, cm_model** out) {
    if (!model_dir || !out) { g_err = "null argument"; return CM_ERR_INVALID; }
    *out = nullptr;
    cm_model* h = nullptr;
    int rc = guard(nullptr, [&] {
        const std::string path(model_dir);
        const bool gguf = path.size() > 5 && path.compare(path.size() - 5, 5, ".gguf") == 0;   // ModelFormat::Auto (qwen3/model.rs:55-71)
        std::string cfg;
        try { cfg = gguf ? cm::gguf_config_json(path) : cmst::read_text(path + "/config.json"); }
        catch (const CmError&) { throw; }
        catch (const std::exception& e) { throw CmError(CM_ERR_IO, e.what()); }
        h = new cm_model();
        create_ranks(h, cfg, opts, [&](Model& m) {
            if (gguf) cm::load_from_gguf(m, path);
            else { cm::load_from_dir(m, path); apply_isq(m); }
        });
    });
    if (rc != CM_OK) { delete h; return rc; }
    *out = h;
    return CM_OK;
}

int cm_create_synthetic(const char* config_json, uint64_t seed, const cm_opts* opts, cm_model** out) {
    if (!config_json || !out) { g_err = "null argument"; return CM_ERR_INVALID; }
    *out = nullptr;
    cm_model* h = nullptr;
    int rc = guard(nullptr, [&] {
        h = new cm_model();
        create_ranks(h, config_json, opts, [&](Model& m) { cm::load_synthetic(m, seed); apply_isq(m); });
    });
    if (rc != CM_OK) { delete h; return rc; }
    *out = h;
    return CM_OK;
}

void cm_destroy(cm_model* h) {
    if (!h) return;
    (void)hipSetDevice(h->m.dev);
    delete h;
}

const char* cm_last_error(const cm_model* h) { return h ? h->m.err.c_str() : g_err.c_str(); }
const char* cm_last_global_error(void) { return g_err.c_str(); }

static void emit(const std::string& txt, char* out, size_t cap, size_t* needed) {
    *needed = txt.size() + 1;
    if (cap >= txt.size() + 1) memcpy(out, txt.c_str(), txt.size() + 1);
    else if (cap) throw CmError(CM_ERR_RANGE, "buffer too small");
}

int cm_checkpoint_inspect(const char* model_dir, char* json_out, size_t cap, size_t* needed) {
    if (!model_dir || !needed || (cap && !json_out)) { g_err = "null argument"; return CM_ERR_INVALID; }
    return guard(nullptr, [&] {
        // host only: the shard discovery + header parsing the loader runs before anything touches the device
        cmst::Checkpoint ck(model_dir);
        std::string js = "{";
        bool first = true;
        for (const std::string& n : ck.names()) {
            const cmst::TensorView& t = ck.get(n);
            uint64_t h = 1469598103934665603ull;                       // FNV-1a over the tensor bytes
            for (size_t i = 0; i < t.nbytes; ++i) { h ^= t.data[i]; h *= 1099511628211ull; }
            js += (first ? "\"" : ", \"") + n + "\": {\"dtype\": \"" + t.dtype + "\", \"shape\": [";
            for (size_t i = 0; i < t.shape.size(); ++i) js += (i ? ", " : "") + std::to_string(t.shape[i]);
            js += "], \"nbytes\": " + std::to_string(t.nbytes) + ", \"fnv1a\": \"" + std::to_string(h) + "\"}";
            first = false;
        }
        js += "}";
        emit(js, json_out, cap, needed);
    });
}

int cm_tp_shard_plan(const char* config_json, int32_t tp_size, int32_t tp_rank, char* json_out, size_t cap, size_t* needed) {
    if (!config_json || !needed || (cap && !json_out)) { g_err = "null argument"; return CM_ERR_INVALID; }
    return guard(nullptr, [&] { emit(cm::tp_shard_plan_json(config_json, tp_size, tp_rank), json_out, cap, needed); });
}

int cm_gguf_config(const char* path, char* json_out, size_t cap, size_t* needed) {
    if (!path || !needed || (cap && !json_out)) { g_err = "null argument"; return CM_ERR_INVALID; }
    return guard(nullptr, [&] {
        const std::string cfg = cm::gguf_config_json(path);          // host only: mmap + metadata / tensor directory
        *needed = cfg.size() + 1;
        if (cap >= cfg.size() + 1) memcpy(json_out, cfg.c_str(), cfg.size() + 1);
        else if (cap) throw CmError(CM_ERR_RANGE, "buffer too small for the config JSON");
    });
}

int cm_tp_unique_id(void* out128) {
    if (!out128) { g_err = "null argument"; return CM_ERR_INVALID; }
    return guard(nullptr, [&] { cm::Rccl::unique_id(out128); });
}

size_t cm_num_layers(const cm_model* h) { return h ? (size_t)h->m.cfg.L : 0; }
size_t cm_vocab_size(const cm_model* h) { return h ? (size_t)h->m.cfg.V : 0; }
size_t cm_hidden_size(const cm_model* h) { return h ? (size_t)h->m.cfg.H : 0; }
size_t cm_max_seq_len(const cm_model* h) { return h ? (size_t)h->m.max_seq : 0; }
uint64_t cm_kv_bytes(const cm_model* h) { return h ? h->m.kv_bytes() : 0; }
uint64_t cm_weight_bytes(const cm_model* h) { return h ? h->m.weight_bytes : 0; }
uint64_t cm_decode_bytes_per_token(const cm_model* h, size_t ctx) { return h ? h->m.decode_bytes_per_token(ctx) : 0; }
int cm_tp_ranks(const cm_model* h) {
    if (!h) return 0;
    if (!h->m.rccl) return 1;
    return h->m.rccl->fake ? 0 : h->m.rccl->nranks;       // (an in-process group: its n ranks, RCCL or peer-store alike)
}
int cm_engine_active(const cm_model* h) { return h && h->m.engine_on ? (h->m.engine_full ? 2 : 1) : 0; }

int cm_forward_step(cm_model* h, const uint32_t* ids, size_t n, size_t start_pos, float* logits_out) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool primary) {
        if (!logits_out) throw CmError(CM_ERR_INVALID, "logits_out is null");
        std::vector<float> tmp;
        m.forward(0, ids, n, start_pos, out_or(primary, logits_out, tmp, (size_t)m.cfg.V), nullptr);
    });
}

int cm_forward_step_greedy(cm_model* h, const uint32_t* ids, size_t n, size_t start_pos, uint32_t* token_out) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool primary) {
        if (!token_out) throw CmError(CM_ERR_INVALID, "token_out is null");
        uint32_t t = 0;
        m.forward(0, ids, n, start_pos, nullptr, primary ? token_out : &t);
    });
}

void cm_clear_kv(cm_model* h) {
    if (!h) return;
    (void)guard_all(h, [&](Model& m, bool) { m.seq_truncate(0, 0); });
}

int cm_warmup(cm_model* h) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool) {
        // generate(&[45, 546, 456], max 5) then clear (qwen3/model.rs:261-267)
        uint32_t prompt[3] = {45u % (uint32_t)m.cfg.V, 546u % (uint32_t)m.cfg.V, 456u % (uint32_t)m.cfg.V};
        uint32_t out[8];
        size_t n = 0;
        cm_gen_config g;
        memset(&g, 0, sizeof g);
        g.max_new_tokens = 5; g.temperature = -1.f; g.top_p = -1.f; g.repetition_penalty = 1.f;
        for (int i = 0; i < 4; ++i) g.eos_token_id[i] = -1;
        m.generate(prompt, 3, &g, out, &n, nullptr, nullptr);
        m.seq_truncate(0, 0);
    });
}

int cm_generate(cm_model* h, const uint32_t* prompt, size_t n_prompt, const cm_gen_config* cfg, uint32_t* tokens_out,
                size_t* n_out, cm_token_cb cb, void* user) {
    if (!h) return CM_ERR_INVALID;
    if (!h->grp) return guard(h, [&] { h->m.generate(prompt, n_prompt, cfg, tokens_out, n_out, cb, user); });
    // in-process group: every rank runs the same loop over identical (gathered) logits; the caller's callback runs on rank 0
    // only and its verdict (continue / stop) is handed to the other ranks token by token
    struct Ctx { cm::TpGroup* g; cm_token_cb cb; void* user; bool primary; size_t seen; };
    auto tramp = [](void* u, uint32_t t) -> int {
        Ctx* c = (Ctx*)u;
        cm::PeerShared& ps = c->g->shared;
        if (c->primary) {
            const int v = c->cb(c->user, t);
            std::lock_guard<std::mutex> lk(ps.mu);
            c->g->cb.verdict.push_back(v);
            ps.cv.notify_all();
            return v;
        }
        std::unique_lock<std::mutex> lk(ps.mu);
        ps.cv.wait(lk, [&] { return ps.failed || c->g->cb.verdict.size() > c->seen; });
        if (c->g->cb.verdict.size() <= c->seen) throw CmError(CM_ERR_DEVICE, "tensor-parallel group: another rank failed");
        return c->g->cb.verdict[c->seen++];
    };
    return guard_all(h, [&](Model& m, bool primary) {
        if (!cfg || !tokens_out || !n_out) throw CmError(CM_ERR_INVALID, "null argument");
        Ctx c{h->grp.get(), cb, user, primary, 0};
        std::vector<uint32_t> tmp;
        size_t n_tmp = 0;
        uint32_t* o = primary ? tokens_out : (tmp.resize(n_prompt + cfg->max_new_tokens + 1), tmp.data());
        m.generate(prompt, n_prompt, cfg, o, primary ? n_out : &n_tmp, cb ? +tramp : nullptr, cb ? &c : nullptr);
    });
}

int cm_seq_alloc(cm_model* h, int32_t* seq_out) {
    if (!h || !seq_out) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool primary) { const int q = m.seq_alloc(); if (primary) *seq_out = q; });   // (the ranks' allocators are deterministic copies)
}
int cm_seq_free(cm_model* h, int32_t seq) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool) { (void)m.seq(seq); m.seq_free(seq); });
}
int cm_seq_fork(cm_model* h, int32_t src, int32_t* seq_out) {
    if (!h || !seq_out) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool primary) { const int q = m.seq_fork(src); if (primary) *seq_out = q; });
}
int64_t cm_seq_len(const cm_model* h, int32_t seq) {
    if (!h || seq < 0 || seq >= (int32_t)h->m.seqs.size() || !h->m.seqs[(size_t)seq].used) return -1;
    return h->m.seqs[(size_t)seq].len;
}
int cm_seq_truncate(cm_model* h, int32_t seq, size_t new_len) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool) { m.seq_truncate(seq, new_len); });
}
int cm_seq_forward(cm_model* h, int32_t seq, const uint32_t* ids, size_t n, size_t start_pos, float* logits_out,
                   uint32_t* greedy_out) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool primary) {
        std::vector<float> tmp; uint32_t t = 0;
        m.forward(seq, ids, n, start_pos, out_or(primary, logits_out, tmp, (size_t)m.cfg.V), (primary || !greedy_out) ? greedy_out : &t);
    });
}

int cm_prefill_batch(cm_model* h, const int32_t* seqs, const uint32_t* const* ids, const size_t* lens, size_t n, float* logits_out,
                     uint32_t* greedy_out) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool primary) {
        if (!seqs || !ids || !lens) throw cm::CmError(CM_ERR_INVALID, "null argument");
        std::vector<uint32_t> tg;
        m.prefill_multi(seqs, ids, lens, n, out_or(primary, greedy_out, tg, n));
        if (logits_out && primary) {
            m.ensure_batch_buffers();
            CM_HIP(hipMemcpy(logits_out, m.logitsb, n * (size_t)m.cfg.V * sizeof(float), hipMemcpyDeviceToHost));
        }
    });
}

int cm_decode_batch(cm_model* h, const int32_t* seqs, const uint32_t* last_tokens, size_t n, float* logits_out,
                    uint32_t* greedy_out) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool primary) {
        if (!seqs || !last_tokens || n == 0) throw CmError(CM_ERR_INVALID, "empty batch");
        std::vector<float> tl; std::vector<uint32_t> tg;
        float* lo = out_or(primary, logits_out, tl, n * (size_t)m.cfg.V);
        uint32_t* go = out_or(primary, greedy_out, tg, n);
        // One pass over the weights for a group of sequences (Model::decode_batch: up to 128 / 64 / 8); a single sequence, and
        // the combinations the batched step does not cover (quantised weights with f32 activations, TP with a vocabulary
        // that does not divide), take the ordinary decode path one sequence at a time.
        const size_t V = (size_t)m.cfg.V;
        const bool batched = (!m.rccl || m.cfg.V % m.tp == 0) && (!m.quantized || m.quant_act_int);
        if (n >= 2 && batched) { m.decode_batch(seqs, last_tokens, n, lo, go); return; }
        for (size_t i = 0; i < n; ++i) {
            const int64_t len = m.seq(seqs[i]).len;
            m.forward(seqs[i], &last_tokens[i], 1, (size_t)len, lo ? lo + i * V : nullptr, go ? go + i : nullptr);
        }
    });
}

int64_t cm_image_token_id(const cm_model* h) { return (h && h->m.vcfg.present) ? (int64_t)h->m.vcfg.image_token : -1; }

int cm_vision_encode(cm_model* h, const float* pixel_values, size_t n_patches, const uint32_t* grid_thw, size_t n_images,
                     float* features_out, size_t* rows_out) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool primary) {      // (the tower is replicated: every rank encodes, rank 0 reports)
        const int rows = m.vision_encode(pixel_values, n_patches, grid_thw, n_images);
        if (!primary) return;
        if (rows_out) *rows_out = (size_t)rows;
        if (features_out) {
            CM_HIP(hipStreamSynchronize(m.stream));
            CM_HIP(hipMemcpy(features_out, m.vFeat, (size_t)rows * m.vcfg.out_hidden * sizeof(float), hipMemcpyDeviceToHost));
        }
    });
}

int cm_vlm_forward(cm_model* h, int32_t seq, const uint32_t* ids, size_t n, size_t start_pos, const float* pixel_values,
                   size_t n_patches, const uint32_t* grid_thw, size_t n_images, float* logits_out, uint32_t* greedy_out) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool primary) {
        std::vector<float> tmp; uint32_t t = 0;
        m.vlm_forward(seq, ids, n, start_pos, pixel_values, n_patches, grid_thw, n_images, out_or(primary, logits_out, tmp, (size_t)m.cfg.V),
                      (primary || !greedy_out) ? greedy_out : &t);
    });
}

int cm_embed_tokens(cm_model* h, const uint32_t* ids, size_t n, float* embeds_out) {
    if (!h) return CM_ERR_INVALID;
    return guard(h, [&] { h->m.embed_tokens(ids, n, embeds_out); });
}

int cm_forward_embeds(cm_model* h, int32_t seq, const float* embeds, size_t n, const int32_t* pos3, size_t start_pos,
                      float* logits_out, uint32_t* greedy_out) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool primary) {
        std::vector<float> tmp; uint32_t t = 0;
        m.forward_embeds(seq, embeds, n, pos3, start_pos, out_or(primary, logits_out, tmp, (size_t)m.cfg.V), (primary || !greedy_out) ? greedy_out : &t);
    });
}

int cm_sample(cm_model* h, const cm_sample_params* p, const uint32_t* context, size_t n_context, uint32_t* token_out) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool primary) {     // (gathers the vocabulary shards: every rank takes part, all draw the same token)
        if (!p || !token_out) throw cm::CmError(CM_ERR_INVALID, "null argument");
        const uint32_t t = m.sample(*p, context, n_context);
        if (primary) *token_out = t;
    });
}

int cm_topk(cm_model* h, const float* logits, size_t n, uint32_t k, uint32_t* idx_out, float* val_out) {
    if (!h) return CM_ERR_INVALID;
    return guard(h, [&] { h->m.topk(logits, n, k, idx_out, val_out); });
}

int cm_read_logits(cm_model* h, float* logits_out) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool primary) {
        if (!logits_out) throw cm::CmError(CM_ERR_INVALID, "null argument");
        std::vector<float> tmp;
        m.fetch_logits(out_or(primary, logits_out, tmp, (size_t)m.cfg.V));
    });
}

int cm_debug_qgemv(cm_model* h, int32_t layer, const char* which, const float* x, size_t k, float* y, size_t n) {
    if (!h || !which || !x || !y) return CM_ERR_INVALID;
    return guard(h, [&] {
        // rank-0-only hooks would run one rank of a group into its collectives alone (a 2 s time-out, then CM_ERR_DEVICE)
        if (h->grp) throw CmError(CM_ERR_UNSUPPORTED, "cm_debug_qgemv: not on an in-process tensor-parallel group");
        h->m.debug_qgemv(layer, which, x, k, y, n);
    });
}

int cm_debug_qgemm(cm_model* h, int32_t layer, const char* which, const float* x, size_t m, size_t k, float* y, size_t n) {
    if (!h || !which || !x || !y) return CM_ERR_INVALID;
    return guard(h, [&] {
        if (h->grp) throw CmError(CM_ERR_UNSUPPORTED, "cm_debug_qgemm: not on an in-process tensor-parallel group");
        h->m.debug_qgemm(layer, which, x, m, k, y, n);
    });
}

int cm_bench_decode(cm_model* h, uint32_t first_token, size_t k, uint32_t* tokens_out, float* ms_out) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool primary) {
        std::vector<uint32_t> tt; float ms = 0.f;
        m.bench_decode(first_token, k, out_or(primary, tokens_out, tt, k), (primary || !ms_out) ? ms_out : &ms);
    });
}

int cm_bench_kernel(cm_model* h, const char* which, size_t iters, float* ms_out, uint64_t* bytes_out) {
    if (!h || !which) return CM_ERR_INVALID;
    return guard(h, [&] {
        if (h->grp) throw CmError(CM_ERR_UNSUPPORTED, "cm_bench_kernel: not on an in-process tensor-parallel group");
        h->m.bench_kernel(which, iters, ms_out, bytes_out);
    });
}

int cm_debug_fill_kv(cm_model* h, size_t ctx, uint64_t seed) {
    if (!h) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& m, bool) { m.debug_fill_kv(ctx, seed); });
}

int cm_debug_read(cm_model* h, const char* what, float* out, size_t n) {
    if (!h || !what || !out) return CM_ERR_INVALID;
    // "<name>@<r>": the buffer of rank r of an in-process group (tests: every rank must hold the same residual stream)
    std::string w0 = what;
    Model* mp = &h->m;
    const size_t at = w0.find('@');
    if (at != std::string::npos) {
        const int r = atoi(w0.c_str() + at + 1);
        if (!h->grp || r < 0 || r >= h->grp->n) return CM_ERR_INVALID;
        mp = &h->grp->model(r);
        w0.resize(at);
    }
    return guard(h, [&] {
        (void)hipSetDevice(mp->dev);
        const float* src = nullptr;
        size_t avail = 0;
        const std::string w = w0;
        // (the persistent kernel never runs under tensor parallelism and the tower's debug buffers live on rank 0: no '@r' form)
        if (mp != &h->m && (w == "engine_trace" || w.rfind("eng_", 0) == 0 || w == "vision_ms" || w == "deepstack"))
            throw CmError(CM_ERR_INVALID, "cm_debug_read: '" + w + "' has no per-rank form");
        if (w == "engine_trace") { h->m.engine_trace(out, n); return; }
        if (w.rfind("eng_", 0) == 0) {     // value halves of a granule buffer of the persistent kernel (last writer wins)
            static const char* names[cm::ENG_NEDGE] = {"eng_x0", "eng_qkv", "eng_part", "eng_attn", "eng_x1", "eng_h"};
            int e = -1;
            for (int k = 0; k < cm::ENG_NEDGE; ++k) if (w == names[k]) e = k;
            if (e < 0 || !h->m.engine_on) throw CmError(CM_ERR_INVALID, "unknown granule buffer / engine off");
            if (n > h->m.eng_gsz[e]) throw CmError(CM_ERR_RANGE, "read beyond granule buffer");
            std::vector<unsigned long long> g(n);
            CM_HIP(hipStreamSynchronize(h->m.stream));
            CM_HIP(hipMemcpy(g.data(), h->m.eng_gran[e], n * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n; ++i) { const uint32_t b = (uint32_t)g[i]; memcpy(&out[i], &b, 4); }
            return;
        }
        if (w == "vision_ms") {            // HIP-event time of the tower's kernels in the LAST cm_vision_encode / cm_vlm_forward, milliseconds
            if (n < 1 || !h->m.v_timed) throw CmError(CM_ERR_INVALID, "vision_ms: no vision encode has run");
            CM_HIP(hipEventSynchronize(h->m.v_ev1));
            CM_HIP(hipEventElapsedTime(&out[0], h->m.v_ev0, h->m.v_ev1));
            return;
        }
        if (w == "deepstack") {            // [n_deepstack][rows][out_hidden] DeepStack feature maps of the LAST cm_vision_encode / cm_vlm_forward
            const size_t nd = h->m.vcfg.deepstack.size(), oh = (size_t)h->m.vcfg.out_hidden;
            if (nd == 0 || !h->m.vDeep || n % (nd * oh) != 0 || n / nd > h->m.deep_stride) throw CmError(CM_ERR_RANGE, "deepstack: n must be n_maps * rows * out_hidden");
            CM_HIP(hipStreamSynchronize(h->m.stream));
            for (size_t k = 0; k < nd; ++k)
                CM_HIP(hipMemcpy(out + k * (n / nd), h->m.vDeep + k * h->m.deep_stride, (n / nd) * sizeof(float), hipMemcpyDeviceToHost));
            return;
        }
        if (w == "q_capture_len") {        // floats recorded since cm_debug_set("q_capture", 1) (model.h)
            // one value: exact below 2^24 floats (refused above); two values: size % 2^24, size / 2^24 (a prompt pass at real widths)
            if (n < 1) throw CmError(CM_ERR_RANGE, "q_capture_len: one or two values");
            const size_t sz = mp->q_cap.size();
            if (n == 1 && sz >= ((size_t)1 << 24)) throw CmError(CM_ERR_RANGE, "q_capture_len: record longer than 2^24 floats -- read two values (low 24 bits, high part)");
            out[0] = (float)(n == 1 ? sz : (sz & (((size_t)1 << 24) - 1)));
            if (n >= 2) out[1] = (float)(sz >> 24);
            return;
        }
        if (w == "q_capture") {            // the records; reading them clears the list
            if (n != mp->q_cap.size()) throw CmError(CM_ERR_RANGE, "q_capture: n must be q_capture_len");
            memcpy(out, mp->q_cap.data(), n * sizeof(float));
            mp->q_cap.clear();
            return;
        }
        if (w == "hidden") { src = mp->x; avail = (size_t)mp->cfg.H; }
        else if (w == "logits") { src = mp->logits; avail = (size_t)mp->V_l * mp->tp; }
        else if (w == "attn") { src = mp->attn; avail = (size_t)mp->Hq_l * mp->cfg.D; }
        else if (w == "qkv") { src = mp->qkv; avail = (size_t)(mp->Hq_l + 2 * mp->Hkv_l) * mp->cfg.D; }
        else if (w == "hbuf") { src = mp->hbuf; avail = (size_t)mp->I_l; }
        else throw CmError(CM_ERR_INVALID, "unknown buffer name");
        if (n > avail) throw CmError(CM_ERR_RANGE, "read beyond buffer");
        CM_HIP(hipStreamSynchronize(mp->stream));
        CM_HIP(hipMemcpy(out, src, n * sizeof(float), hipMemcpyDeviceToHost));
    });
}

// test hook: the peer-store all-reduce / all-gather alone, n rank threads on `device`; returns the number of wrong elements (< 0: error)
long cm_debug_peer_selftest(int32_t n_ranks, int32_t device, int32_t iters, int32_t count) {
    long bad = -1;
    const int rc = guard(nullptr, [&] { bad = cm::peer_selftest(n_ranks, device, iters, count); });
    return rc == CM_OK ? bad : -(long)(rc < 0 ? -rc : rc) - 1000;
}

// test hook: the launch plan of the int8 decode-group GEMM for a shape (pure host logic)
int cm_debug_qgemm_plan(int32_t m, int32_t n, int32_t k, int32_t epi, uint64_t ws_floats, int32_t num_cu, int64_t out[8]) {
    if (!out || num_cu < 1) return CM_ERR_INVALID;
    const int fmt = (epi >> 8) & 0xFF;       // bits 8 .. 15: ggml type of the weights (0 = the Q8_0 layout, 12 = Q4_K, 14 = Q6_K)
    const cm::QGemmPlan p = cm::plan_gemm_q8(m, n, k, epi & 0xFF, ws_floats > 0, (size_t)ws_floats, num_cu, fmt ? fmt : (int)cm::QFMT_Q8_0);
    out[0] = p.ok; out[1] = p.direct; out[2] = 4 * p.mh; out[3] = 32 * p.mh * p.mt; out[4] = p.qg; out[5] = p.groups; out[6] = p.ks; out[7] = p.grid;
    return CM_OK;
}

int cm_debug_set(cm_model* h, const char* key, int64_t value) {
    if (!h || !key) return CM_ERR_INVALID;
    return guard_all(h, [&](Model& mm, bool) {
        const std::string k = key;
        if (k == "no_prefill") mm.no_prefill = value != 0;
        else if (k == "quant_prefill") mm.quant_prefill = value != 0;
        else if (k == "prefill_q8") {       // before the first prompt pass of the handle: 0 = prompts over Q8_0-layout weights on the dequantised-to-bf16 GEMMs
            if (mm.pX != nullptr) throw CmError(CM_ERR_INVALID, "prefill_q8: set before the first prompt pass");
            mm.q8_prefill_want = value != 0;
        }
        else if (k == "attn_outq") mm.attn_outq = value != 0;
        else if (k == "prefill_split") mm.prefill_split2 = value < 0 ? mm.default_prefill_split2() : value != 1;   // 1: plain bf16, 0 / 2: hi + lo, -1: back to cm_opts
        // which decode-attention kernel / persistent-kernel mode a step uses (tests and A/B runs; one hipGraph per variant, so
        // the thresholds switch live -- a switch that changes what a variant enqueues drops the captured graphs)
        else if (k == "attn_mfma_min") mm.attn_mfma_min = value;
        else if (k == "attn_mfma_wide_min") mm.attn_mfma_wide_min = value;
        else if (k == "attn_heads_max") mm.attn_heads_max = value;
        else if (k == "attn_ns") { mm.attn_ns = (int)std::max<long long>(1, std::min<long long>(value, mm.nsplit)); mm.drop_graphs(); }
        else if (k == "gemm256") mm.gemm256 = (int)value;
        else if (k == "gdn_ba_fused") { mm.gdn_ba_fused = value != 0; mm.drop_graphs(); }
        else if (k == "sample_rows") mm.sample_rows_on = value != 0;
        else if (k == "prefill_seg_batch") mm.seg_batch = value != 0;
        else if (k == "prefill_lo_mask") mm.prefill_lo_mask = (int)value;
        else if (k == "gdn_chunked") mm.gdn_ck_on = value != 0;
        else if (k == "gdn_defer_norm") { mm.gdn_defer_norm = value != 0; mm.drop_graphs(); }
        else if (k == "tp_graph") { mm.tp_graph = value != 0; mm.drop_graphs(); }      // CM_TP_GRAPH: RCCL collectives captured into the decode graph
        else if (k == "lm_head_gemm_min") mm.lm_head_gemm_min = (int)std::max<long long>(0, value);
        else if (k == "q_gemm_min") mm.q_gemm_min = (int)std::max<long long>(0, value);
        else if (k == "q_capture") { mm.q_capture = value != 0; mm.q_cap.clear(); }
        else if (k == "quant_act_int") mm.quant_act_int = value != 0;          // CM_QUANT_ACT: 1 = ggml vec_dot (integer) semantics, 0 = f32 activations
        else if (k == "vision_merger_gelu") mm.vcfg.merger_act = value == 2 ? 2 : 1;   // CM_VISION_MERGER_GELU: 1 tanh form (reference), 2 erf (HF)
        else if (k == "engine") { mm.drop_graphs(); mm.engine_on = value > 0 && mm.engine_capable; }
        else if (k == "engine_full") { mm.drop_graphs(); mm.engine_full = value != 0 && mm.engine_full_capable; }
        else if (k == "batch_gemm_min") mm.batch_gemm_min = (int)std::max<long long>(0, std::min<long long>(value, Model::GEMV_MAXB));   // a group beyond the GEMV kernels' 64 must take the GEMM path
        else if (k == "attn_splits") mm.attn_splits_force = (int)std::max<long long>(0, std::min<long long>(value, mm.nsplit));
        else throw CmError(CM_ERR_INVALID, "unknown debug switch");
    });
}

}  // extern "C"
