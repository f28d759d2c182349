// Continuous-batching engine on the paged KV pool -- the caller of the hot path.
//
// Mirrors the scheduling core of the reference:
//   Scheduler::schedule            crane-serve/src/engine/scheduler.rs:67-98   (prefill priority, FIFO)
//   InferenceEngine::accept_request engine/mod.rs:525-599  (reject long prompts, clamp max_tokens)
//   step_prefill                   engine/mod.rs:643-731   (whole prompt, sample first token, promote)
//   step_decode_batch              engine/mod.rs:822-1057  (one token for every running sequence)
//   evict_if_needed                engine/mod.rs:430-504   (victim = longest running sequence, back of the queue,
//                                                           effective_max_running capped until one finishes)
//   Sequence::should_stop / finish_reason  engine/sequence.rs:75-125
//   sampling::sample               engine/sampling.rs:169-373 (context = last repeat_last_n tokens)
// What is different by design: sequences own KV pages, so a decode round is cm_decode_batch over page tables --
// there is no swap_in/swap_out, pad_and_stack_kv_caches or extract_batch_kv; the budget is the page pool itself
// (free pages) instead of tracked bytes; a preempted sequence keeps its generated tokens and re-prefills
// prompt ++ generated (the reference truncates to the prompt and re-emits).
#include <algorithm>
#include <cstring>
#include <deque>
#include <unordered_map>

#include "model.h"
#include "tp_group.h"

namespace cm {

Model& model_of(cm_model* h);    // api.cpp

struct Request {
    uint64_t id = 0;
    int seq = -1;                       // model sequence handle while running
    std::vector<uint32_t> tokens;       // prompt ++ generated (Sequence::tokens)
    size_t prompt_len = 0;
    uint32_t max_tokens = 0;
    cm_sample_params sp{};
    bool greedy_plain = false;          // temperature <= 0 and no penalties: the arg-max of the step itself
    int64_t eos[4] = {-1, -1, -1, -1};
    bool cancelled = false;

    size_t num_generated() const { return tokens.size() - prompt_len; }
    bool last_is_eos() const {
        if (tokens.empty()) return false;
        for (int i = 0; i < 4; ++i) if (eos[i] >= 0 && (uint32_t)eos[i] == tokens.back()) return true;
        return false;
    }
    bool should_stop() const { return num_generated() >= max_tokens || last_is_eos(); }     // sequence.rs:75-86
};

struct Engine {
    cm_model* handle = nullptr;
    Model* m = nullptr;
    cm_engine_opts opts{};
    size_t max_running = 0;
    long effective_max_running = -1;    // Scheduler::effective_max_running (None == -1)
    uint64_t next_id = 1;
    std::deque<uint64_t> waiting, running;
    std::unordered_map<uint64_t, Request> reqs;
    std::deque<cm_engine_event> events;
    cm_engine_stats stats{};
    std::string err;

    Request& req(uint64_t id) { return reqs.at(id); }

    void emit_token(const Request& r, uint32_t t) {
        cm_engine_event e{}; e.req_id = r.id; e.kind = CM_EV_TOKEN; e.token = t; events.push_back(e);
    }
    void release(Request& r) {
        if (r.seq >= 0) { m->seq_free(r.seq); r.seq = -1; }
    }
    void drop(uint64_t id) {
        waiting.erase(std::remove(waiting.begin(), waiting.end(), id), waiting.end());
        running.erase(std::remove(running.begin(), running.end(), id), running.end());
        reqs.erase(id);
    }
    void finish(uint64_t id, uint32_t reason) {                     // finish_sequence (engine/mod.rs:1265-1316)
        Request& r = req(id);
        cm_engine_event e{};
        e.req_id = id; e.kind = CM_EV_FINISHED; e.finish_reason = reason;
        e.prompt_tokens = (uint32_t)r.prompt_len; e.completion_tokens = (uint32_t)r.num_generated();
        events.push_back(e);
        if (reason != CM_FINISH_CANCELLED) { stats.completed++; stats.completion_tokens += r.num_generated(); }
        release(r);
        drop(id);
        effective_max_running = -1;                                  // the cap is lifted when a sequence finishes
    }
    void fail(uint64_t id, int code, const std::string& msg) {      // send_error (engine/mod.rs:1256-1263)
        cm_engine_event e{}; e.req_id = id; e.kind = CM_EV_ERROR; e.error = code; events.push_back(e);
        err = msg;
        stats.failed++;
        auto it = reqs.find(id);
        if (it != reqs.end()) { release(it->second); drop(id); }
    }

    size_t pages_for(size_t len) const { return (len + (size_t)m->page - 1) / (size_t)m->page; }

    // evict_if_needed (engine/mod.rs:430-504): free pages by preempting the longest running sequence(s)
    bool make_room(size_t need_pages, uint64_t keep) {
        bool evicted = false;
        while (m->free_pages.size() < need_pages) {
            uint64_t victim = 0; size_t vlen = 0; bool found = false;
            for (uint64_t id : running) {
                if (id == keep) continue;
                const size_t l = req(id).tokens.size();
                if (!found || l > vlen) { victim = id; vlen = l; found = true; }
            }
            if (!found) break;
            Request& v = req(victim);
            release(v);
            running.erase(std::remove(running.begin(), running.end(), victim), running.end());
            waiting.push_back(victim);                               // back, not front: no thrashing
            stats.preemptions++;
            evicted = true;
        }
        if (evicted) effective_max_running = (long)running.size() + (keep && std::find(running.begin(), running.end(), keep) == running.end() ? 1 : 0);
        return m->free_pages.size() >= need_pages;
    }

    uint32_t pick(Request& r, float* dev_logits, uint32_t greedy_tok) {
        if (r.greedy_plain) return greedy_tok;                       // sampling.rs:191-210 fast path
        const size_t w = std::min(r.tokens.size(), (size_t)opts.repeat_last_n);     // sampling.rs:217
        cm_sample_params sp = r.sp;
        sp.repeat_last_n = 0;
        sp.draw = (uint32_t)r.num_generated();
        return m->sample(sp, r.tokens.data() + (r.tokens.size() - w), w, false, dev_logits);
    }

    void step_prefill(uint64_t id) {
        Request& r = req(id);
        try {
            if (r.seq < 0) r.seq = m->seq_alloc();
            if (!make_room(pages_for(r.tokens.size() + 1), id))
                throw CmError(CM_ERR_OOM, "prompt does not fit in the KV pool");
            uint32_t greedy = 0;
            m->forward(r.seq, r.tokens.data(), r.tokens.size(), 0, nullptr, &greedy);
            stats.prefill_steps++;
            const uint32_t t = pick(r, nullptr, greedy);
            r.tokens.push_back(t);
            emit_token(r, t);
            if (r.should_stop()) finish(id, r.last_is_eos() ? CM_FINISH_STOP : CM_FINISH_LENGTH);
            else running.push_back(id);                              // promote_to_running
        } catch (const CmError& e) {
            fail(id, e.code, e.what());
        }
    }

    // Several waiting prompts in ONE pass over the weights (Model::prefill_multi).  The reference prefills one whole prompt per
    // step while running < max_running (scheduler.rs:67-98) -- back to back when several are waiting; batching them changes no
    // token and no event order, only the cost: a 128-token prompt alone occupies one m-tile of every GEMM and costs what 1024
    // rows cost.  Returns false when the pass is not applicable (the caller falls back to step_prefill of the first prompt).
    // Cancellation is looked at once per engine step (step()), i.e. once per pass of up to MAXB prompts here against once per
    // prompt on the one-prompt path: a request cancelled while its pass runs still emits its first token, then finishes as
    // cancelled at the next step -- the API is single-threaded, so a cancel cannot arrive inside a step anyway.
    bool step_prefill_many(size_t cap) {
        if (!opts.batch_prefill || m->kvq() || m->no_prefill || (m->quantized && !m->quant_prefill) || (m->rccl && m->cfg.V % m->tp != 0)) return false;
        m->ensure_prefill_buffers();
        if (!m->prefill_ok) return false;
        std::vector<uint64_t> ids;
        size_t total = 0, pages = 0;
        for (uint64_t id : waiting) {
            if (running.size() + ids.size() >= cap || ids.size() >= (size_t)Model::MAXB) break;
            const Request& r = req(id);
            if (total + r.tokens.size() > (size_t)m->chunk) break;                    // FIFO: never skip ahead of a long prompt
            const size_t need = pages_for(r.tokens.size() + 1);
            if (pages + need > m->free_pages.size()) break;                           // no eviction for the batched pass
            total += r.tokens.size(); pages += need;
            ids.push_back(id);
        }
        if (ids.size() < 2) return false;
        for (size_t k = 0; k < ids.size(); ++k) waiting.pop_front();
        std::vector<int32_t> sq(ids.size());
        std::vector<const uint32_t*> ptr(ids.size());
        std::vector<size_t> len(ids.size());
        std::vector<uint32_t> greedy(ids.size());
        try {
            for (size_t k = 0; k < ids.size(); ++k) {
                Request& r = req(ids[k]);
                if (r.seq < 0) r.seq = m->seq_alloc();
                sq[k] = r.seq; ptr[k] = r.tokens.data(); len[k] = r.tokens.size();
            }
            m->prefill_multi(sq.data(), ptr.data(), len.data(), ids.size(), greedy.data());
        } catch (const CmError&) {
            // isolate the offender the way the one-prompt path does: every request of the failed pass goes through
            // step_prefill on its own, in FIFO order, so only the prompt that cannot be processed fails (its error event),
            // the others emit exactly what the batched pass would have emitted
            for (uint64_t id : ids) if (reqs.count(id)) step_prefill(id);
            return true;
        }
        // the sampled rows of the pass: one set of sampler launches and ONE host sync instead of one per prompt
        std::vector<uint32_t> picked(ids.size());
        try {
            std::vector<Model::SampleReq> rows;
            for (size_t k = 0; k < ids.size(); ++k) {
                Request& r = req(ids[k]);
                if (r.greedy_plain) { picked[k] = greedy[k]; continue; }
                const size_t w = std::min(r.tokens.size(), (size_t)opts.repeat_last_n);
                cm_sample_params sp = r.sp;
                sp.repeat_last_n = 0;
                sp.draw = (uint32_t)r.num_generated();
                rows.push_back({(int)k, sp, r.tokens.data() + (r.tokens.size() - w), w, false, m->logitsb + k * (size_t)m->cfg.V});
            }
            if (!rows.empty()) {
                uint32_t got[Model::SAMPLE_SLOTS];
                m->sample_enqueue_rows(rows.data(), (int)rows.size());
                m->sample_collect((int)ids.size(), got);
                for (const auto& q : rows) picked[(size_t)q.slot] = got[q.slot];
            }
        } catch (const CmError&) {
            // the row sampler failed (e.g. one request's parameters): re-run the requests one by one (the prompt pass is
            // repeated for them -- K/V appends are idempotent, a GDN sequence restarts from position 0 in forward())
            for (uint64_t id : ids) if (reqs.count(id)) step_prefill(id);
            return true;
        }
        for (size_t k = 0; k < ids.size(); ++k) {
            Request& r = req(ids[k]);
            stats.prefill_steps++;
            try {
                const uint32_t t = picked[k];
                r.tokens.push_back(t);
                emit_token(r, t);
                if (r.should_stop()) finish(ids[k], r.last_is_eos() ? CM_FINISH_STOP : CM_FINISH_LENGTH);
                else running.push_back(ids[k]);
            } catch (const CmError& e) {
                fail(ids[k], e.code, e.what());
            }
        }
        return true;
    }

    void step_decode(std::vector<uint64_t> batch) {
        // every sequence appends one token: make sure the pages exist, evicting if the pool is short
        size_t reserved = 0;
        for (size_t i = 0; i < batch.size();) {
            Request& r = req(batch[i]);
            if (r.seq < 0) { ++i; continue; }                        // preempted by an earlier iteration
            const size_t have = m->seq(r.seq).pages.size(), want = pages_for(r.tokens.size());
            const size_t need = want > have ? want - have : 0;
            if (need == 0 || make_room(reserved + need, batch[i])) { reserved += need; ++i; continue; }
            fail(batch[i], CM_ERR_OOM, "KV pool exhausted");
            batch.erase(batch.begin() + (long)i);
        }
        batch.erase(std::remove_if(batch.begin(), batch.end(), [&](uint64_t id) {
            return std::find(running.begin(), running.end(), id) == running.end(); }), batch.end());   // preempted above
        if (batch.empty()) return;
        stats.decode_rounds++;
        try {
            if (batch.size() == 1 || (m->rccl && m->cfg.V % m->tp != 0) || (m->quantized && !m->quant_act_int)) {
                // step_decode_sequential (engine/mod.rs:1064-1170): graph-replayed single-sequence step
                for (uint64_t id : batch) {
                    Request& r = req(id);
                    uint32_t greedy = 0;
                    m->forward(r.seq, &r.tokens.back(), 1, r.tokens.size() - 1, nullptr, &greedy);
                    const uint32_t t = pick(r, nullptr, greedy);
                    r.tokens.push_back(t);
                    emit_token(r, t);
                }
            } else {
                std::vector<int32_t> sq(batch.size());
                std::vector<uint32_t> toks(batch.size()), greedy(batch.size()), out(batch.size());
                for (size_t i = 0; i < batch.size(); ++i) { sq[i] = req(batch[i]).seq; toks[i] = req(batch[i]).tokens.back(); }
                const std::function<void(size_t, int)> after = [&](size_t g0, int nb) {
                    // every sampled row of the group is enqueued on its own sampler slot; ONE host sync for the group
                    uint32_t picked[Model::SAMPLE_SLOTS];
                    std::vector<Model::SampleReq> rows;
                    for (int b = 0; b < nb; ++b) {
                        Request& r = req(batch[g0 + (size_t)b]);
                        if (r.greedy_plain) continue;
                        const size_t w = std::min(r.tokens.size(), (size_t)opts.repeat_last_n);
                        cm_sample_params sp = r.sp;
                        sp.repeat_last_n = 0;
                        sp.draw = (uint32_t)r.num_generated();
                        rows.push_back({b, sp, r.tokens.data() + (r.tokens.size() - w), w, false, m->logitsb + (size_t)b * m->cfg.V});
                    }
                    if (!rows.empty()) { m->sample_enqueue_rows(rows.data(), (int)rows.size()); m->sample_collect(nb, picked); }
                    for (int b = 0; b < nb; ++b) {
                        Request& r = req(batch[g0 + (size_t)b]);
                        out[g0 + (size_t)b] = r.greedy_plain ? m->h_stb[b].next : picked[b];
                    }
                };
                m->decode_batch(sq.data(), toks.data(), batch.size(), nullptr, greedy.data(), &after);
                for (size_t i = 0; i < batch.size(); ++i) {
                    Request& r = req(batch[i]);
                    r.tokens.push_back(out[i]);
                    emit_token(r, out[i]);
                }
            }
        } catch (const CmError& e) {
            // batch-decode errors fail every live sequence of the batch (engine/mod.rs:945-954)
            for (uint64_t id : batch) if (reqs.count(id)) fail(id, e.code, e.what());
            return;
        }
        for (uint64_t id : batch) {
            Request& r = req(id);
            if (r.should_stop()) finish(id, r.last_is_eos() ? CM_FINISH_STOP : CM_FINISH_LENGTH);
        }
    }

    void step() {
        // check_cancelled (engine/mod.rs:601-620)
        std::vector<uint64_t> gone;
        for (auto& kv : reqs) if (kv.second.cancelled) gone.push_back(kv.first);
        for (uint64_t id : gone) finish(id, CM_FINISH_CANCELLED);
        // Scheduler::schedule (scheduler.rs:67-98)
        const size_t cap = effective_max_running >= 0 ? (size_t)effective_max_running : max_running;
        if (running.size() < cap && !waiting.empty()) {
            if (waiting.size() >= 2 && cap - running.size() >= 2 && step_prefill_many(cap)) return;
            const uint64_t id = waiting.front(); waiting.pop_front();
            step_prefill(id);
        } else if (!running.empty()) {
            step_decode(std::vector<uint64_t>(running.begin(), running.end()));
        } else if (!waiting.empty()) {
            const uint64_t id = waiting.front(); waiting.pop_front();
            step_prefill(id);
        }
    }
};

}  // namespace cm

// e: the engine over the handle's model.  On an in-process tensor-parallel handle (cm_opts.tp_mode = CM_TP_IN_PROCESS) every
// rank runs its OWN copy of the scheduler over its own shard (peers): the engines see the same submissions and cancels in the
// same order and sample identical tokens from identical gathered logits, so they take the same decisions step by step; a
// step runs on all ranks at once (TpGroup::run) and the caller is handed rank 0's events.
struct cm_engine {
    cm::Engine e;
    cm::TpGroup* grp = nullptr;
    std::vector<std::unique_ptr<cm::Engine>> peers;      // ranks 1 .. n-1
    cm::Engine& rank(int r) { return r == 0 ? e : *peers[(size_t)r - 1]; }
    int n() const { return grp ? grp->n : 1; }
};

using cm::CmError;

namespace cm { TpGroup* group_of(cm_model* h); }

static void engine_init(cm::Engine& e, cm_model* handle, cm::Model* m, const cm_engine_opts* opts) {
    e.handle = handle;
    e.m = m;
    if (opts) e.opts = *opts;
    if (e.opts.repeat_last_n == 0) e.opts.repeat_last_n = 64;
    e.opts.batch_prefill = opts ? (opts->batch_prefill >= 0) : 1;      // 0 default (on), 1 on, -1 off -> stored as bool
    const size_t slots = m->seqs.size() > 1 ? m->seqs.size() - 1 : 1;
    e.max_running = e.opts.max_running ? std::min<size_t>(e.opts.max_running, slots) : slots;
    if (e.opts.seed == 0) e.opts.seed = 299792458ull;
    e.stats.total_pages = (uint64_t)m->n_pages;
}

// f(engine of rank r) on every rank (concurrently for a group); the first error wins
template <typename F>
static int engine_all(cm_engine* h, F&& f) {
    int rc = CM_OK;
    try {
        if (!h->grp) { (void)hipSetDevice(h->e.m->dev); f(h->e); }
        else h->grp->run([&](int r) { f(h->rank(r)); });
    } catch (const CmError& x) { h->e.err = x.what(); rc = x.code; }
    catch (const std::exception& x) { h->e.err = x.what(); rc = CM_ERR_INVALID; }
    for (auto& p : h->peers) p->events.clear();           // only rank 0's events are reported
    return rc;
}

extern "C" {

int cm_engine_create(cm_model* m, const cm_engine_opts* opts, cm_engine** out) {
    if (!m || !out) return CM_ERR_INVALID;
    cm_engine* h = new cm_engine();
    engine_init(h->e, m, &cm::model_of(m), opts);
    h->grp = cm::group_of(m);
    if (h->grp)
        for (int r = 1; r < h->grp->n; ++r) {
            h->peers.emplace_back(new cm::Engine());
            engine_init(*h->peers.back(), m, &h->grp->model(r), opts);
        }
    *out = h;
    return CM_OK;
}

void cm_engine_destroy(cm_engine* h) {
    if (!h) return;
    (void)engine_all(h, [](cm::Engine& e) {
        for (auto& kv : e.reqs) if (kv.second.seq >= 0) { try { e.m->seq_free(kv.second.seq); } catch (...) {} }
    });
    delete h;
}

static int engine_submit_one(cm::Engine& e, const cm_request* r, uint64_t* id_out) {
    if (!r->tokens || r->n_tokens == 0) { e.err = "empty prompt"; e.stats.failed++; return CM_ERR_INVALID; }
    if (r->n_tokens > (size_t)e.m->max_seq - 1) {
        e.err = "Prompt length (" + std::to_string(r->n_tokens) + ") exceeds server max_seq_len (" + std::to_string(e.m->max_seq) + ")";
        e.stats.failed++;
        return CM_ERR_RANGE;
    }
    for (size_t i = 0; i < r->n_tokens; ++i)
        if (r->tokens[i] >= (uint32_t)e.m->cfg.V) { e.err = "token id >= vocab_size"; e.stats.failed++; return CM_ERR_RANGE; }
    cm::Request q;
    q.id = e.next_id++;
    q.tokens.assign(r->tokens, r->tokens + r->n_tokens);
    q.prompt_len = r->n_tokens;
    q.max_tokens = (uint32_t)std::min<size_t>(r->max_tokens, (size_t)e.m->max_seq - r->n_tokens);   // effective_max_tokens
    memcpy(q.eos, r->eos_token_id, sizeof q.eos);
    q.sp.temperature = r->temperature < 0.f ? 1.0f : r->temperature;     // None -> 1.0 (sampling.rs:229)
    q.sp.top_p = r->top_p; q.sp.top_k = r->top_k;
    q.sp.repetition_penalty = r->repetition_penalty == 0.f ? 1.0f : r->repetition_penalty;
    q.sp.frequency_penalty = r->frequency_penalty; q.sp.presence_penalty = r->presence_penalty;
    q.sp.seed = r->seed ? r->seed : (e.opts.seed ^ (q.id * 0x9E3779B97F4A7C15ull));
    q.greedy_plain = !(q.sp.temperature > 0.f) && q.sp.repetition_penalty == 1.0f && q.sp.frequency_penalty == 0.f &&
                     q.sp.presence_penalty == 0.f;
    *id_out = q.id;
    e.stats.prompt_tokens += r->n_tokens;
    e.waiting.push_back(q.id);
    e.reqs.emplace(q.id, std::move(q));
    return CM_OK;
}

int cm_engine_submit(cm_engine* h, const cm_request* r, uint64_t* id_out) {
    if (!h || !r || !id_out) return CM_ERR_INVALID;
    const int rc = engine_submit_one(h->e, r, id_out);       // host-only bookkeeping: the same on every rank, in the same order
    for (auto& p : h->peers) { uint64_t id = 0; (void)engine_submit_one(*p, r, &id); }
    return rc;
}

int cm_engine_cancel(cm_engine* h, uint64_t id) {
    if (!h) return CM_ERR_INVALID;
    auto it = h->e.reqs.find(id);
    if (it == h->e.reqs.end()) { h->e.err = "unknown request id"; return CM_ERR_INVALID; }
    it->second.cancelled = true;
    for (auto& p : h->peers) { auto jt = p->reqs.find(id); if (jt != p->reqs.end()) jt->second.cancelled = true; }
    return CM_OK;
}

int cm_engine_step(cm_engine* h, cm_engine_event* ev, size_t cap, size_t* n) {
    if (!h || !n || (cap && !ev)) return CM_ERR_INVALID;
    cm::Engine& e = h->e;
    int rc = CM_OK;
    if (e.events.size() < cap || cap == 0) rc = engine_all(h, [](cm::Engine& x) { x.step(); });
    size_t k = 0;
    while (k < cap && !e.events.empty()) { ev[k++] = e.events.front(); e.events.pop_front(); }
    *n = k;
    return rc;
}

int cm_engine_step_many(cm_engine* h, size_t max_steps, cm_engine_event* ev, size_t cap, size_t* n) {
    if (!h || !n || !ev || cap == 0) return CM_ERR_INVALID;
    cm::Engine& e = h->e;
    int rc = CM_OK;
    // a step emits at most 2 events per running sequence (token + finished); stop while that still fits in `cap`
    const size_t per_step = 2 * (size_t)std::max<long>(1, e.opts.max_running > 0 ? (long)e.opts.max_running : (long)e.m->seqs.size());
    // always make progress: with an empty event queue one step runs even when its worst case (2 events per running
    // sequence) would not fit `cap` -- the surplus stays queued for the next call -- otherwise a small buffer would
    // return n = 0 with work pending and the caller would spin forever
    // (the loop condition reads rank 0's queues only: every rank's engine is in the same state between steps)
    for (size_t i = 0; rc == CM_OK && i < max_steps && (e.events.size() + per_step <= cap || (i == 0 && e.events.empty())); ++i) {
        if (e.waiting.empty() && e.running.empty()) break;
        rc = engine_all(h, [](cm::Engine& x) { x.step(); });
    }
    size_t k = 0;
    while (k < cap && !e.events.empty()) { ev[k++] = e.events.front(); e.events.pop_front(); }
    *n = k;
    return rc;
}

int cm_engine_has_work(const cm_engine* h) {
    if (!h) return 0;
    return (!h->e.waiting.empty() || !h->e.running.empty() || !h->e.events.empty()) ? 1 : 0;
}

int cm_engine_get_stats(const cm_engine* h, cm_engine_stats* out) {
    if (!h || !out) return CM_ERR_INVALID;
    *out = h->e.stats;
    out->waiting = h->e.waiting.size();
    out->running = h->e.running.size();
    out->free_pages = h->e.m->free_pages.size();
    out->total_pages = (uint64_t)h->e.m->n_pages;
    return CM_OK;
}

const char* cm_engine_last_error(const cm_engine* h) { return h ? h->e.err.c_str() : "null engine"; }

}  // extern "C"
