// Qwen 3.5-VL: vision tower + VLM glue on the device.
//   vision_encode : Qwen3_5VisionModel::forward        (qwen3_5/vision.rs:558-584)
//   vlm_forward   : Qwen3_5VLModel::forward             (qwen3_5/vlm.rs:250-285): encode images, splice the image
//                   rows over the <|image_pad|> embeddings, 3-axis MRoPE positions (build_position_ids :190-241),
//                   text prefill; the sequence keeps the MRoPE counter for later decode steps (:294-301).
// Host logic here = the index arithmetic the reference also does on the host (pos-embed interpolation
// indices/weights, rotary coordinates, cu_seqlens, position ids); all tensor math runs in HIP kernels.
#include <cmath>
#include <cstring>

#include "model.h"

namespace cm {

void Model::ensure_vision_buffers(int n_patches) {
    if (!vcfg.present) throw CmError(CM_ERR_UNSUPPORTED, "model has no vision tower");
    if (n_patches <= v_cap) return;
    if (v_cap != 0) throw CmError(CM_ERR_RANGE, "image exceeds the vision scratch (4096 patches)");
    const int cap = std::max(n_patches, 4096), pad = (cap + 127) / 128 * 128;
    const int VH = vcfg.hidden, VI = vcfg.inter, M = vcfg.merge * vcfg.merge;
    const size_t wide = (size_t)std::max({vcfg.patch_dim(), VI, VH * M, 3 * VH});
    vX = dalloc<float>((size_t)cap * VH);
    vQKV = dalloc<float>((size_t)cap * 3 * VH);
    vPix = dalloc<float>((size_t)cap * vcfg.patch_dim());
    vFeat = dalloc<float>((size_t)(cap / M + 1) * vcfg.out_hidden);
    deep_stride = (size_t)(cap / M + 1) * vcfg.out_hidden;
    if (!vcfg.deepstack.empty()) vDeep = dalloc<float>(deep_stride * vcfg.deepstack.size());
    vCos = dalloc<float>((size_t)cap * 64);
    vSin = dalloc<float>((size_t)cap * 64);
    const size_t pages = (size_t)(cap + 63) / 64;
    vkv_lo_off = pages * vcfg.heads * 64 * 64;
    vK = dalloc<uint16_t>(2 * vkv_lo_off);
    vV = dalloc<uint16_t>(2 * vkv_lo_off);
    vW4 = dalloc<float>((size_t)4 * cap);
    vIdx = dalloc<int>((size_t)4 * cap);
    vBt = dalloc<int>(pages);
    if (const char* e = getenv("CM_VIT_KSPLIT")) vit_ksplit = std::max(1, std::min(8, atoi(e)));
    if (vit_ksplit > 1) {
        vPartO = dalloc<float>((size_t)vit_ksplit * cap * VH);
        vPartML = dalloc<float>((size_t)vit_ksplit * cap * vcfg.heads * 2);
    }
    auto z = [&](size_t n) { uint16_t* p = dalloc<uint16_t>(n); CM_HIP(hipMemsetAsync(p, 0, n * 2, stream)); return p; };
    vA_hi = z((size_t)pad * wide); vA_lo = z((size_t)pad * wide);
    vB_hi = z((size_t)pad * wide); vB_lo = z((size_t)pad * wide);
    vQ_hi = z((size_t)pad * VH); vQ_lo = z((size_t)pad * VH);
    std::vector<int32_t> ident(pages);
    for (size_t i = 0; i < pages; ++i) ident[i] = (int32_t)i;
    CM_HIP(hipMemcpyAsync(vBt, ident.data(), pages * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    CM_HIP(hipStreamSynchronize(stream));
    v_cap = cap;
}

int Model::vision_encode(const float* pix, size_t n_patches, const uint32_t* grid, size_t n_img) {
    if (!pix || !grid || n_img == 0 || n_patches == 0) throw CmError(CM_ERR_INVALID, "empty image input");
    const int VH = vcfg.hidden, heads = vcfg.heads, hd = VH / heads, m = vcfg.merge, M = m * m;
    size_t expect = 0;
    for (size_t g = 0; g < n_img; ++g) {
        if (grid[3 * g + 1] % m || grid[3 * g + 2] % m) throw CmError(CM_ERR_INVALID, "grid_thw not divisible by spatial_merge_size");
        expect += (size_t)grid[3 * g] * grid[3 * g + 1] * grid[3 * g + 2];
    }
    if (expect != n_patches) throw CmError(CM_ERR_INVALID, "pixel_values rows != sum(t*h*w) of grid_thw");
    ensure_vision_buffers((int)n_patches);
    ensure_gemm_workspace();                   // (the split-K workspace of the GEMMs; not the text path's prompt buffers)
    const int N = (int)n_patches;
    hipStream_t s = stream;

    // ---- host index arithmetic ----
    // (a) bilinear pos-embed blend in merge-block-major row order (vision.rs:382-489)
    std::vector<int32_t> idx((size_t)4 * N);
    std::vector<float> wts((size_t)4 * N), cs((size_t)N * hd), sn((size_t)N * hd);
    const int side = (int)std::lround(std::sqrt((double)vcfg.num_pos));
    const int quarter = hd / 4;
    std::vector<float> inv((size_t)quarter);
    for (int i = 0; i < quarter; ++i) inv[(size_t)i] = 1.0f / powf(10000.0f, (float)(2 * i) / (float)(hd / 2));   // vision.rs:285-297
    std::vector<std::pair<int, int>> frames;      // [start, end) token ranges, one per temporal frame (vision.rs:543-556)
    int row = 0;
    for (size_t g = 0; g < n_img; ++g) {
        const int t = (int)grid[3 * g], h = (int)grid[3 * g + 1], w = (int)grid[3 * g + 2];
        auto lin = [&](int n, int i) { return n == 1 ? 0.0f : (float)i * ((float)(side - 1) / (float)(n - 1)); };
        for (int tt = 0; tt < t; ++tt) {
            frames.emplace_back(row, row + h * w);
            for (int br = 0; br < h / m; ++br) for (int bc = 0; bc < w / m; ++bc)
                for (int ir = 0; ir < m; ++ir) for (int ic = 0; ic < m; ++ic) {
                    const int r = br * m + ir, c = bc * m + ic;
                    const float hv = lin(h, r), wv = lin(w, c);
                    const int hf = (int)floorf(hv), wf = (int)floorf(wv);
                    const int hc = std::min((int)ceilf(hv), side - 1), wc = std::min((int)ceilf(wv), side - 1);
                    const float dh = hv - (float)hf, dw = wv - (float)wf;
                    idx[0 * N + row] = hf * side + wf; idx[1 * N + row] = hf * side + wc;
                    idx[2 * N + row] = hc * side + wf; idx[3 * N + row] = hc * side + wc;
                    wts[0 * N + row] = (1.0f - dh) * (1.0f - dw); wts[1 * N + row] = (1.0f - dh) * dw;
                    wts[2 * N + row] = dh * (1.0f - dw); wts[3 * N + row] = dh * dw;
                    // (b) 2-D rotary: emb = cat(rot, rot), rot = [row freqs | col freqs] (vision.rs:491-541,565-569)
                    for (int d = 0; d < hd; ++d) {
                        const int e = d % (hd / 2);
                        const float f = (e < quarter) ? (float)r * inv[(size_t)e] : (float)c * inv[(size_t)(e - quarter)];
                        cs[(size_t)row * hd + d] = cosf(f); sn[(size_t)row * hd + d] = sinf(f);
                    }
                    ++row;
                }
        }
    }
    CM_HIP(hipStreamSynchronize(s));
    CM_HIP(hipMemcpyAsync(vIdx, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    CM_HIP(hipMemcpyAsync(vW4, wts.data(), wts.size() * sizeof(float), hipMemcpyHostToDevice, s));
    CM_HIP(hipMemcpyAsync(vCos, cs.data(), cs.size() * sizeof(float), hipMemcpyHostToDevice, s));
    CM_HIP(hipMemcpyAsync(vSin, sn.data(), sn.size() * sizeof(float), hipMemcpyHostToDevice, s));
    CM_HIP(hipMemcpyAsync(vPix, pix, (size_t)N * vcfg.patch_dim() * sizeof(float), hipMemcpyHostToDevice, s));
    CM_HIP(hipStreamSynchronize(s));          // the host vectors die at scope exit
    // device time of the tower itself (inputs resident in HBM from here on): cm_debug_read("vision_ms") after the call
    if (!v_ev0) { CM_HIP(hipEventCreate(&v_ev0)); CM_HIP(hipEventCreate(&v_ev1)); }
    CM_HIP(hipEventRecord(v_ev0, s));

    // nw / nb (GEPI_RESADD into vX): LayerNorm(vX) with these weights -> vA_hi / vA_lo, the A operand of the NEXT projection, written by
    // the GEMM's split-K reduction launch (GemmArgs::norm_w) -- the launch_layernorm_rows of the next sub-block
    auto gemm = [&](const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* W, const float* bias, int Mrows, int Ncols, int K,
                    int epi, float* C, uint16_t* h_hi, uint16_t* h_lo, int act, const float* nw = nullptr, const float* nb = nullptr) {
        GemmArgs g{};
        g.ws = pWS; g.ws_floats = pWS ? gemm_ws_floats : 0;      // split-K: at 784 patches a GEMM has 56-224 output tiles for 256 CUs
        g.ksplit_cap = 512;                                       // (round 6 sweep, profiles/r06_vit_knob_sweep.log: 3.92 -> 3.78 ms of tower time against the text path's 768)
        g.A_hi = a_hi; g.A_lo = a_lo; g.W = W; g.bias = bias; g.M = Mrows; g.N = Ncols; g.K = K; g.C = C; g.ldc = Ncols;
        g.H_hi = h_hi; g.H_lo = h_lo; g.act = act;
        if (nw) { g.norm_w = nw; g.norm_b = nb; g.norm_hi = vA_hi; g.norm_lo = vA_lo; g.norm_eps = 1e-6f; }
        if (!launch_gemm(g, epi, s)) throw CmError(CM_ERR_UNSUPPORTED, "vision GEMM shape (N % 128, K % 32)");
    };
    // patch embed (Conv3d == linear over the flattened patch) + bias, + pos embed
    launch_split_rows(vPix, vA_hi, vA_lo, (size_t)N * vcfg.patch_dim(), s);
    gemm(vA_hi, vA_lo, vw.patch_w, vw.patch_b, N, VH, vcfg.patch_dim(), GEPI_STORE, vX, nullptr, nullptr, 0);
    launch_pos_embed_add(vX, vw.pos_table, vIdx, vW4, N, VH, s);
    const float scale = (float)(1.0 / std::sqrt((double)hd));
    bool n1_ready = false;                                        // vA already holds the LayerNorm the next projection reads
    for (int li = 0; li < vcfg.depth; ++li) {
        const VisionBlockW& b = vw.blocks[(size_t)li];
        if (!n1_ready) launch_layernorm_rows(vX, b.n1w, b.n1b, vA_hi, vA_lo, N, VH, 1e-6f, s);
        gemm(vA_hi, vA_lo, b.qkv_w, b.qkv_b, N, 3 * VH, VH, GEPI_STORE, vQKV, nullptr, nullptr, 0);
        launch_vit_rope_kv(vQKV, vCos, vSin, vQ_hi, vQ_lo, vK, vV, vkv_lo_off, N, heads, scale, s);
        for (auto& fr : frames) {                                  // full attention inside each frame (vision.rs:145-172)
            AttnPreArgs at{};
            at.q_hi = vQ_hi + (size_t)fr.first * VH; at.q_lo = vQ_lo + (size_t)fr.first * VH;
            at.block_table = vBt; at.kpool = vK; at.vpool = vV; at.kv_lo_off = vkv_lo_off;
            at.out_hi = vB_hi + (size_t)fr.first * VH; at.out_lo = vB_lo + (size_t)fr.first * VH;
            at.S = fr.second - fr.first; at.Hq = heads; at.Hkv = heads; at.nrep = 1; at.page = 64;
            at.start_pos = fr.first; at.causal = 0; at.kv_lo = fr.first; at.kv_hi = fr.second;
            // enough (query tile, head, key run) blocks for ~3 waves per SIMD; a run is never shorter than two key tiles
            at.ksplit = std::max(1, std::min(vit_ksplit, (at.S + 127) / 128)); at.part_o = vPartO; at.part_ml = vPartML;
            launch_attn_prefill(at, 64, KV_BF16X2, s);
        }
        gemm(vB_hi, vB_lo, b.proj_w, b.proj_b, N, VH, VH, GEPI_RESADD, vX, nullptr, nullptr, 0, b.n2w, b.n2b);      // + norm2 -> vA
        gemm(vA_hi, vA_lo, b.fc1_w, b.fc1_b, N, vcfg.inter, VH, GEPI_ACT_SPLIT, nullptr, vB_hi, vB_lo, vcfg.act);
        // fc2 + the NEXT norm over vX rows (the next block's norm1, the merger's norm after the last block) -- unless a DeepStack
        // tap of this layer needs vA for its own (regrouped-row) LayerNorm first
        bool tap = false;
        for (size_t k = 0; k < vcfg.deepstack.size(); ++k) tap = tap || vcfg.deepstack[k] == li;
        const bool last = li + 1 == vcfg.depth;
        n1_ready = !tap;
        gemm(vB_hi, vB_lo, b.fc2_w, b.fc2_b, N, VH, vcfg.inter, GEPI_RESADD, vX, nullptr, nullptr, 0,
             tap ? nullptr : (last ? vw.mn_w : vw.blocks[(size_t)li + 1].n1w), tap ? nullptr : (last ? vw.mn_b : vw.blocks[(size_t)li + 1].n1b));
        // DeepStack tap (qwen3_vl/vision.rs:572-579): PatchMerger with the LayerNorm over the regrouped 4 x hidden row
        // (use_postshuffle_norm, :236-276; the regrouping is free in merge-block-major order)
        for (size_t k = 0; k < vcfg.deepstack.size(); ++k) {
            if (vcfg.deepstack[k] != li) continue;
            const VisionW::Merger& d = vw.deep[k];
            const int Gd = N / M, MH = VH * M;
            launch_layernorm_rows(vX, d.n_w, d.n_b, vA_hi, vA_lo, Gd, MH, 1e-6f, s);
            gemm(vA_hi, vA_lo, d.fc1_w, d.fc1_b, Gd, MH, MH, GEPI_ACT_SPLIT, nullptr, vB_hi, vB_lo, vcfg.merger_act);
            gemm(vB_hi, vB_lo, d.fc2_w, d.fc2_b, Gd, vcfg.out_hidden, MH, GEPI_STORE, vDeep + k * deep_stride, nullptr, nullptr, 0);
        }
    }
    // PatchMerger (vision.rs:254-278): LayerNorm over hidden, rows regrouped 4 -> 1 (free: block-major order)
    if (!n1_ready) launch_layernorm_rows(vX, vw.mn_w, vw.mn_b, vA_hi, vA_lo, N, VH, 1e-6f, s);
    const int G = N / M;
    gemm(vA_hi, vA_lo, vw.mfc1_w, vw.mfc1_b, G, VH * M, VH * M, GEPI_ACT_SPLIT, nullptr, vB_hi, vB_lo, vcfg.merger_act);
    gemm(vB_hi, vB_lo, vw.mfc2_w, vw.mfc2_b, G, vcfg.out_hidden, VH * M, GEPI_STORE, vFeat, nullptr, nullptr, 0);
    CM_HIP(hipEventRecord(v_ev1, s));
    v_timed = true;
    return G;
}

void Model::vlm_forward(int sidx, const uint32_t* ids, size_t n, size_t start_pos, const float* pix, size_t n_patches,
                        const uint32_t* grid, size_t n_img, float* logits_out, uint32_t* greedy_out) {
    if (n == 0 || !ids) throw CmError(CM_ERR_INVALID, "empty input");
    if (!vcfg.present) throw CmError(CM_ERR_UNSUPPORTED, "model has no vision tower");
    Seq& q = seq(sidx);
    if ((int64_t)start_pos != q.len) throw CmError(CM_ERR_RANGE, "vlm_forward must continue exactly at the cached length");
    if (start_pos + n > (size_t)max_seq) throw CmError(CM_ERR_RANGE, "start_pos + n exceeds max_seq_len");
    for (size_t i = 0; i < n; ++i) if (ids[i] >= (uint32_t)cfg.V) throw CmError(CM_ERR_RANGE, "token id >= vocab_size");
    ensure_prefill_buffers();
    if (!prefill_ok) throw CmError(CM_ERR_UNSUPPORTED, "prefill GEMM shapes unsupported for this model");
    if ((int)n > chunk) throw CmError(CM_ERR_UNSUPPORTED, "image prompts longer than prefill_chunk are not implemented");
    const int rows = vision_encode(pix, n_patches, grid, n_img);
    if (vcfg.out_hidden != cfg.H) throw CmError(CM_ERR_INVALID, "vision out_hidden_size != text hidden_size");
    // build_position_ids (vlm.rs:190-241) + splice map (vlm.rs:433-468)
    std::vector<int32_t> pos3(3 * n), map(n, -1);
    int32_t nxt = (int32_t)start_pos + q.rope_delta;
    size_t img = 0, i = 0;
    int feat_row = 0;
    const int m = vcfg.merge;
    while (i < n) {
        if ((long long)ids[i] != vcfg.image_token) {
            pos3[i] = pos3[n + i] = pos3[2 * n + i] = nxt++;
            ++i;
            continue;
        }
        if (img >= n_img) throw CmError(CM_ERR_INVALID, "more image spans than grid_thw entries");
        const int gt = (int)grid[3 * img], gh = (int)grid[3 * img + 1] / m, gw = (int)grid[3 * img + 2] / m;
        const size_t span = (size_t)gt * gh * gw, hw = (size_t)gh * gw;
        if (i + span > n) throw CmError(CM_ERR_INVALID, "not enough image placeholder tokens for the image grid");
        for (size_t k = 0; k < span; ++k) {
            if ((long long)ids[i + k] != vcfg.image_token) throw CmError(CM_ERR_INVALID, "image placeholder span is not contiguous");
            pos3[i + k] = nxt + (int32_t)(k / hw);
            pos3[n + i + k] = nxt + (int32_t)((k % hw) / gw);
            pos3[2 * n + i + k] = nxt + (int32_t)((k % hw) % gw);
            map[i + k] = feat_row++;
        }
        nxt += std::max(gt, std::max(gh, gw));
        i += span; ++img;
    }
    if (feat_row != rows) throw CmError(CM_ERR_INVALID, "image placeholder count != number of image embeddings");
    if (!dMap) { dMap = dalloc<int>(chunk); dPos3 = dalloc<int>((size_t)3 * chunk); }
    CM_HIP(hipMemcpyAsync(dMap, map.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    CM_HIP(hipMemcpyAsync(dPos3, pos3.data(), 3 * n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    CM_HIP(hipStreamSynchronize(stream));
    ensure_pages(sidx, (int64_t)(start_pos + n));
    activate(sidx);
    splice_map_dev = dMap; pos3_dev = dPos3; pos3_stride = (int)n;
    deep_layers = n_img > 0 ? (int)vcfg.deepstack.size() : 0;     // DeepStack injection after the first decoder layers (qwen3_vl/text.rs:262-278)
    if (deep_layers > cfg.L) throw CmError(CM_ERR_INVALID, "more DeepStack feature maps than decoder layers");
    try {
        prefill(ids, n, start_pos);
    } catch (...) {
        splice_map_dev = nullptr; pos3_dev = nullptr; deep_layers = 0;
        throw;
    }
    splice_map_dev = nullptr; pos3_dev = nullptr; deep_layers = 0;
    q.len = (int64_t)(start_pos + n);
    q.rope_delta = nxt - (int32_t)q.len;
    if (greedy_out) {
        CM_HIP(hipMemcpyAsync(h_st, st, sizeof(StepState), hipMemcpyDeviceToHost, stream));
        CM_HIP(hipStreamSynchronize(stream));
        *greedy_out = h_st->next;
    }
    if (logits_out) fetch_logits(logits_out);
    if (!greedy_out && !logits_out) CM_HIP(hipStreamSynchronize(stream));
}

// Qwen3_5TextModel::embed_only (qwen3_5/model.rs:368-370): rows of the embedding table as f32 (exact: bf16 -> f32)
void Model::embed_tokens(const uint32_t* ids, size_t n, float* out) {
    if (!ids || !out || n == 0) throw CmError(CM_ERR_INVALID, "empty input");
    for (size_t i = 0; i < n; ++i) if (ids[i] >= (uint32_t)cfg.V) throw CmError(CM_ERR_RANGE, "token id >= vocab_size");
    ensure_prefill_buffers();
    if (!prefill_ok) throw CmError(CM_ERR_UNSUPPORTED, "prefill buffers unavailable for this model");
    const int H = cfg.H;
    for (size_t off = 0; off < n; off += (size_t)chunk) {
        const int S = (int)std::min<size_t>((size_t)chunk, n - off);
        CM_HIP(hipStreamSynchronize(stream));
        memcpy(h_ids, ids + off, (size_t)S * sizeof(uint32_t));
        CM_HIP(hipMemcpyAsync(d_ids, h_ids, (size_t)S * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        if (quantized && q_embed.fmt != QFMT_NONE) launch_embed_rows_q(q_embed, d_ids, pX, S, H, cfg.V, stream);
        else launch_embed_rows(embed, d_ids, pX, S, H, cfg.V, stream);
        CM_HIP(hipMemcpyAsync(out + off * (size_t)H, pX, (size_t)S * H * sizeof(float), hipMemcpyDeviceToHost, stream));
        CM_HIP(hipStreamSynchronize(stream));
    }
}

// Qwen3_5TextModel::forward_embeds (qwen3_5/model.rs:430-510): run the decoder over hidden rows the caller built (text
// embeddings with image features spliced in) at explicit 3-axis MRoPE positions; logits of the last position only.  Causal
// over the cached prefix + these rows (the reference's default mask, prefill.rs:56-136).  pos3 == null: positions
// start_pos + i + rope_delta on all three axes (a text-only continuation).  The sequence's MRoPE counter for later decode
// steps becomes max(pos3) + 1 (vlm.rs:294-301: rope_delta = counter - cached length).
void Model::forward_embeds(int sidx, const float* embeds, size_t n, const int32_t* pos3, size_t start_pos, float* logits_out,
                           uint32_t* greedy_out) {
    if (n == 0 || !embeds) throw CmError(CM_ERR_INVALID, "empty input");
    Seq& q = seq(sidx);
    if ((int64_t)start_pos != q.len) throw CmError(CM_ERR_RANGE, "forward_embeds must continue exactly at the cached length");
    if (start_pos + n > (size_t)max_seq) throw CmError(CM_ERR_RANGE, "start_pos + n exceeds max_seq_len");
    ensure_prefill_buffers();
    if (!prefill_ok) throw CmError(CM_ERR_UNSUPPORTED, "prefill GEMM shapes unsupported for this model");
    int32_t top = -1;
    if (pos3) {
        if ((int)n > chunk) throw CmError(CM_ERR_UNSUPPORTED, "explicit positions for more rows than prefill_chunk are not implemented");
        for (size_t i = 0; i < 3 * n; ++i) {
            if (pos3[i] < 0 || pos3[i] >= max_seq) throw CmError(CM_ERR_RANGE, "position id outside the rotary table");
            top = std::max(top, pos3[i]);
        }
        // later steps of this sequence rotate at cache position + rope_delta (= top + 1 - len, vlm.rs:294-301) and the sequence
        // may grow to max_seq tokens: a POSITIVE delta would walk past the max_seq-row cos / sin table before the cache is
        // full.  Image prompts compress positions (delta <= 0); a pos3 that runs ahead of the cache is refused up front.
        if ((int64_t)top + 1 > (int64_t)(start_pos + n))
            throw CmError(CM_ERR_RANGE, "pos3 runs ahead of the cache (max position + 1 > start_pos + n): later steps would leave the rotary table");
        if (!dMap) { dMap = dalloc<int>(chunk); dPos3 = dalloc<int>((size_t)3 * chunk); }
        CM_HIP(hipMemcpyAsync(dPos3, pos3, 3 * n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        CM_HIP(hipStreamSynchronize(stream));
    }
    ensure_pages(sidx, (int64_t)(start_pos + n));
    activate(sidx);
    embeds_host = embeds;
    pos3_dev = pos3 ? dPos3 : nullptr; pos3_stride = (int)n;
    try {
        prefill(nullptr, n, start_pos);
    } catch (...) {
        embeds_host = nullptr; pos3_dev = nullptr;
        throw;
    }
    embeds_host = nullptr; pos3_dev = nullptr;
    q.len = (int64_t)(start_pos + n);
    if (pos3) q.rope_delta = top + 1 - (int32_t)q.len;
    if (greedy_out) {
        CM_HIP(hipMemcpyAsync(h_st, st, sizeof(StepState), hipMemcpyDeviceToHost, stream));
        CM_HIP(hipStreamSynchronize(stream));
        *greedy_out = h_st->next;
    }
    if (logits_out) fetch_logits(logits_out);
    if (!greedy_out && !logits_out) CM_HIP(hipStreamSynchronize(stream));
}

}  // namespace cm
