// Host side of the device sampler: Sequence::sample (crane-serve/src/engine/sampling.rs:169-373) with every
// [V]-sized step on the GPU.  Only the <= repeat_last_n context ids go up and one u32 comes back per token.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "kernels.h"
#include "model.h"
#include "tp.h"

namespace cm {

static constexpr int SM_BLOCKS = 256;

void Model::ensure_sampler() {
    if (tk_idx) return;
    // one scratch set per sampler slot (slot 0 doubles as the single-row path)
    tk_hist = dalloc<uint32_t>((size_t)SAMPLE_SLOTS * (4096 + 4));
    CM_HIP(hipMemsetAsync(tk_hist, 0, (size_t)SAMPLE_SLOTS * (4096 + 4) * sizeof(uint32_t), stream));
    tk_idx = dalloc<uint32_t>((size_t)SAMPLE_SLOTS * 512);
    tk_val = dalloc<float>((size_t)SAMPLE_SLOTS * 512);
    d_pen = dalloc<uint32_t>((size_t)SAMPLE_SLOTS * 2 * PEN_CAP);
    d_tok = dalloc<uint32_t>(SAMPLE_SLOTS);
    tk_cand_row_cap = std::max<size_t>((size_t)topk_blocks(cfg.V) * 64, 4096);
    tk_cand_rows = (unsigned long long*)dalloc<uint32_t>((size_t)SAMPLE_SLOTS * tk_cand_row_cap * 2);
    sm_pmax = dalloc<float>(SM_BLOCKS);
    sm_pidx = dalloc<int>(SM_BLOCKS);
    CM_HIP(hipHostMalloc((void**)&h_pen, (size_t)SAMPLE_SLOTS * 2 * PEN_CAP * sizeof(uint32_t)));
    CM_HIP(hipHostMalloc((void**)&h_tk, (1 + 512 + 512) * sizeof(uint32_t)));
    d_stab = (SampleRowDev*)dalloc<uint32_t>((size_t)SAMPLE_SLOTS * sizeof(SampleRowDev) / sizeof(uint32_t));
    CM_HIP(hipHostMalloc((void**)&h_stab, (size_t)SAMPLE_SLOTS * sizeof(SampleRowDev)));
}

void Model::gather_logits() {
    if (rccl && !logits_gathered) {
        rccl->all_gather(logits + (size_t)rank * V_l, logits, (size_t)V_l * sizeof(float), stream);
        logits_gathered = true;
    }
}

void Model::topk(const float* host_logits, size_t n, uint32_t k, uint32_t* idx_out, float* val_out) {
    if (!idx_out) throw CmError(CM_ERR_INVALID, "null argument");
    ensure_sampler();
    const float* src = logits;
    if (host_logits) {
        if (n == 0 || n > (size_t)1 << 30) throw CmError(CM_ERR_RANGE, "n out of range");
        if (n > tk_in_cap) {
            if (tk_in) (void)hipFree(tk_in);
            tk_in = nullptr; tk_in_cap = 0;
            CM_HIP(hipMalloc((void**)&tk_in, n * sizeof(float)));
            tk_in_cap = n;
        }
        CM_HIP(hipMemcpyAsync(tk_in, host_logits, n * sizeof(float), hipMemcpyHostToDevice, stream));
        src = tk_in;
    } else {
        gather_logits();
        n = (size_t)cfg.V;
    }
    if (k == 0 || k > 512 || (size_t)k > n) throw CmError(CM_ERR_RANGE, "k must be in 1..min(512, n)");
    const size_t need = std::max<size_t>((size_t)topk_blocks((int)n) * (size_t)topk_pad((int)k), 4096);
    if (need > tk_cand_cap) {
        if (tk_cand) (void)hipFree(tk_cand);
        tk_cand = nullptr; tk_cand_cap = 0;
        CM_HIP(hipMalloc((void**)&tk_cand, need * sizeof(unsigned long long)));
        tk_cand_cap = need;
    }
    launch_topk(src, (int)n, (int)k, tk_cand, tk_hist, tk_hist + 4096, tk_idx, tk_val, stream);
    CM_HIP(hipMemcpyAsync(h_tk + 1, tk_idx, k * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    CM_HIP(hipMemcpyAsync(h_tk + 1 + 512, tk_val, k * sizeof(float), hipMemcpyDeviceToHost, stream));
    CM_HIP(hipStreamSynchronize(stream));
    memcpy(idx_out, h_tk + 1, k * sizeof(uint32_t));
    if (val_out) memcpy(val_out, h_tk + 1 + 512, k * sizeof(float));
}

uint32_t Model::sample(const cm_sample_params& p, const uint32_t* ctx, size_t n_ctx, bool true_div, float* dev_logits) {
    sample_enqueue(0, p, ctx, n_ctx, true_div, dev_logits);
    uint32_t t = 0;
    sample_collect(1, &t);
    return t;
}

void Model::sample_collect(int n_slots, uint32_t* tokens_out) {
    CM_HIP(hipMemcpyAsync(h_tk, d_tok, (size_t)n_slots * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    CM_HIP(hipStreamSynchronize(stream));
    memcpy(tokens_out, h_tk, (size_t)n_slots * sizeof(uint32_t));
}

// penalty window -> distinct ids + counts (sampling.rs:422-478), written as [ids | counts] at `dst`; returns the distinct count
static int penalty_list(const cm_sample_params& p, const uint32_t* ctx, size_t n_ctx, int V, uint32_t* dst, int cap) {
    const size_t start = (p.repeat_last_n && n_ctx > p.repeat_last_n) ? n_ctx - p.repeat_last_n : 0;
    std::unordered_map<uint32_t, uint32_t> pos;
    std::vector<uint32_t> ids, counts;
    for (size_t i = start; i < n_ctx; ++i) {
        const uint32_t t = ctx[i];
        if (t >= (uint32_t)V) continue;
        auto it = pos.find(t);
        if (it != pos.end()) { counts[it->second]++; continue; }
        if ((int)ids.size() == cap) throw CmError(CM_ERR_RANGE, "more than 8192 distinct tokens in the penalty window");
        pos.emplace(t, (uint32_t)ids.size());
        ids.push_back(t); counts.push_back(1);
    }
    const int nd = (int)ids.size();
    for (int i = 0; i < nd; ++i) { dst[i] = ids[(size_t)i]; dst[nd + i] = counts[(size_t)i]; }
    return nd;
}

// Every sampled row of a decode group (or of a batched prompt pass) in ONE set of launches: the rows' parameters, scratch
// pointers and packed penalty lists go up in two copies, eight kernels with blockIdx.y = row do what sample_enqueue does per
// row (same device code, same scratch per slot: the tokens are the same).  Rows that need the full-vocabulary Gumbel-max (no
// top-k, no top-p) share one partial buffer and stay on the per-row path.
void Model::sample_enqueue_rows(const SampleReq* rq, int n) {
    if (n <= 0) return;
    ensure_sampler();
    if (!sample_rows_on || n == 1) {
        for (int i = 0; i < n; ++i) sample_enqueue(rq[i].slot, rq[i].p, rq[i].ctx, rq[i].n_ctx, rq[i].true_div, rq[i].dev_logits);
        return;
    }
    const int V = cfg.V;
    size_t pack = 0;                       // u32 offset into h_pen / d_pen (packed: the per-slot regions are not used here)
    int nt = 0, max_pen = 0;
    std::vector<int> later;
    for (int i = 0; i < n; ++i) {
        const SampleReq& q = rq[i];
        const cm_sample_params& p = q.p;
        if (q.slot < 0 || q.slot >= SAMPLE_SLOTS || q.dev_logits == nullptr) throw CmError(CM_ERR_INVALID, "sampler row");
        const bool top_p_active = p.top_p > 0.f && p.top_p < 1.f;
        int k = 1, sample = 0;
        if (p.temperature > 0.f) {
            k = (int)p.top_k;
            if (k == 0 && top_p_active) k = 64;
            k = std::min(std::min(k, 64), V);
            if (!(k > 0 && (k < V || top_p_active))) { later.push_back(i); continue; }
            sample = 1;
        }
        SampleRowDev& r = h_stab[nt++];
        r.logits = q.dev_logits;
        r.hist = tk_hist + (size_t)q.slot * (4096 + 4); r.sel = r.hist + 4096;
        r.cand = tk_cand_rows + (size_t)q.slot * tk_cand_row_cap;
        r.val = tk_val + (size_t)q.slot * 512;
        r.tok = d_tok + q.slot;
        r.idx_out = sample ? tk_idx + (size_t)q.slot * 512 : r.tok;
        r.k = k; r.kp = topk_pad(k); r.sample = sample; r.true_div = q.true_div ? 1 : 0;
        r.temperature = p.temperature; r.top_p = top_p_active ? p.top_p : 0.f;
        r.seed_lo = (uint32_t)p.seed; r.seed_hi = (uint32_t)(p.seed >> 32); r.draw = p.draw;
        const bool rep = p.repetition_penalty != 1.0f && p.repetition_penalty > 0.f;
        const bool fp = p.frequency_penalty != 0.f || p.presence_penalty != 0.f;
        r.rp = rep ? p.repetition_penalty : 1.0f; r.rp_inv = (float)(1.0 / (double)r.rp);
        r.fp = p.frequency_penalty; r.pp = p.presence_penalty;
        r.pen_n = 0; r.pen_ids = nullptr; r.pen_counts = nullptr;
        if ((rep || fp) && q.ctx && q.n_ctx) {
            const int nd = penalty_list(p, q.ctx, q.n_ctx, V, h_pen + pack, PEN_CAP);
            if (nd) {
                r.pen_n = nd; r.pen_ids = d_pen + pack; r.pen_counts = d_pen + pack + nd;
                pack += (size_t)2 * nd;
                max_pen = std::max(max_pen, nd);
            }
        }
    }
    if (nt) {
        if (pack) CM_HIP(hipMemcpyAsync(d_pen, h_pen, pack * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        CM_HIP(hipMemcpyAsync(d_stab, h_stab, (size_t)nt * sizeof(SampleRowDev), hipMemcpyHostToDevice, stream));
        launch_sample_rows(d_stab, nt, V, max_pen, stream);
    }
    // the per-row path stages its penalty list in the row's own slot region of the pinned buffer: if the packed lists above
    // reach into that region, their copy has to have happened before the host overwrites it
    for (int i : later) {
        if (pack > (size_t)rq[i].slot * 2 * PEN_CAP) CM_HIP(hipStreamSynchronize(stream));
        sample_enqueue(rq[i].slot, rq[i].p, rq[i].ctx, rq[i].n_ctx, rq[i].true_div, rq[i].dev_logits);
    }
}

void Model::sample_enqueue(int slot, const cm_sample_params& p, const uint32_t* ctx, size_t n_ctx, bool true_div, float* dev_logits) {
    if (slot < 0 || slot >= SAMPLE_SLOTS) throw CmError(CM_ERR_INVALID, "sampler slot out of range");
    ensure_sampler();
    float* logits = dev_logits ? dev_logits : this->logits;      // a row of the batched step, or the last forward's logits
    if (!dev_logits) gather_logits();
    const int V = cfg.V;
    uint32_t* hp = h_pen + (size_t)slot * 2 * PEN_CAP;           // pinned staging of this slot (reused only after a sync)
    uint32_t* dp = d_pen + (size_t)slot * 2 * PEN_CAP;
    uint32_t* hist = tk_hist + (size_t)slot * (4096 + 4);
    uint32_t* idx = tk_idx + (size_t)slot * 512;
    float* val = tk_val + (size_t)slot * 512;
    unsigned long long* cand = tk_cand_rows + (size_t)slot * tk_cand_row_cap;
    uint32_t* tok = d_tok + slot;
    // ---- penalties over the window (sampling.rs:422-478): distinct ids + counts, applied on the device ----
    const bool rep = p.repetition_penalty != 1.0f && p.repetition_penalty > 0.f;      // strict sentinel (sampling.rs:431)
    const bool fp = p.frequency_penalty != 0.f || p.presence_penalty != 0.f;
    if ((rep || fp) && ctx && n_ctx) {
        const size_t start = (p.repeat_last_n && n_ctx > p.repeat_last_n) ? n_ctx - p.repeat_last_n : 0;
        std::unordered_map<uint32_t, uint32_t> pos;
        int nd = 0;
        for (size_t i = start; i < n_ctx; ++i) {
            const uint32_t t = ctx[i];
            if (t >= (uint32_t)V) continue;
            auto it = pos.find(t);
            if (it != pos.end()) { hp[PEN_CAP + it->second]++; continue; }
            if (nd == PEN_CAP) throw CmError(CM_ERR_RANGE, "more than 8192 distinct tokens in the penalty window");
            pos.emplace(t, (uint32_t)nd);
            hp[nd] = t; hp[PEN_CAP + nd] = 1; ++nd;
        }
        if (nd) {
            for (int i = 0; i < nd; ++i) hp[nd + i] = hp[PEN_CAP + i];                // [ids | counts] in one H2D copy
            CM_HIP(hipMemcpyAsync(dp, hp, (size_t)2 * nd * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            launch_penalties(logits, dp, dp + nd, nd, rep ? p.repetition_penalty : 1.0f, true_div, p.frequency_penalty,
                             p.presence_penalty, V, stream);
        }
    }
    if (!(p.temperature > 0.f)) {                                  // greedy: arg-max, first index on ties
        launch_topk(logits, V, 1, cand, hist, hist + 4096, tok, val, stream);
    } else {
        const bool top_p_active = p.top_p > 0.f && p.top_p < 1.f;
        int k = (int)p.top_k;
        if (k == 0 && top_p_active) k = 64;                        // CRANE_TOPP_FALLBACK_TOPK default (sampling.rs:263-267)
        k = std::min(std::min(k, 64), V);                          // sampling.rs:268
        // k == V (a vocabulary of <= 64 tokens) with an active top-p still goes through the sorted top-k path: the nucleus
        // cut needs the descending order (the reference falls back to the nucleus LogitsProcessor there)
        if (k > 0 && (k < V || top_p_active)) {
            launch_topk(logits, V, k, cand, hist, hist + 4096, idx, val, stream);
            launch_sample_topk(idx, val, k, p.temperature, top_p_active ? p.top_p : 0.f, p.seed, p.draw, tok, stream);
        } else {
            // full-vocabulary Gumbel-max shares one partial buffer: stream order serialises the slots
            launch_gumbel_full(logits, V, p.temperature, p.seed, p.draw, sm_pmax, sm_pidx, SM_BLOCKS, tok, stream);
        }
    }
}

}  // namespace cm
