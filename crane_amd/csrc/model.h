// Host runtime of the MI355X inference path: model object, paged KV-cache allocator,
// sequences, decode hipGraph, autoregressive loop.  Internal C++; the public surface is
// include/crane_mi355.h.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/crane_mi355.h"
#include "kernels.h"

namespace cm {

struct CmError : std::runtime_error {
    int code;
    CmError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define CM_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            throw cm::CmError(_e == hipErrorOutOfMemory ? CM_ERR_OOM : CM_ERR_DEVICE,         \
                              std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)

// config.json (qwen3/modeling.rs:94-130 and defaults)
struct Config {
    std::string model_type = "qwen3";
    int V = 0, H = 0, I = 0, L = 0, Hq = 0, Hkv = 0, D = 0, max_pos = 0;
    float eps = 1e-6f;
    double theta = 1e6;
    bool tie = true, qk_norm = true, attention_bias = false;
    long long eos = -1;
    // Qwen 3.5 / 3.6 / 3.8 hybrid (qwen3_5/config.rs:47-254)
    bool hybrid = false;            // false: qwen3 dense
    int rot_dim = 0;                // rotary slice of the head (== D for qwen3)
    int interval = 4;               // full_attention_interval
    int NK = 0, NV = 0, Kd = 0, Vd = 0, conv_k = 4;   // NK / NV are THIS RANK's head counts under TP
    int NK_g = 0, NV_g = 0;                           // checkpoint-wide head counts
    bool attn_gate = false;
    float norm_off = 0.f;           // Qwen35RmsNorm: weight = 1 + w (folded at load)
    int mrope_sec[3] = {11, 11, 10};
    int key_dim() const { return NK * Kd; }
    int value_dim() const { return NV * Vd; }
    int conv_dim() const { return 2 * key_dim() + value_dim(); }
    bool layer_full(int i) const { return !hybrid || ((i + 1) % interval) == 0; }
};

struct LayerW {
    // quantised model, prompt-weight cache on: the bf16 GEMM operand of each projection, dequantised ONCE (filled at the first prompt
    // pass that reaches the layer): [0] qkv / in_proj (merged, padded), [1] o / out_proj, [2] gate|up, [3] down
    uint16_t* dq[4] = {nullptr, nullptr, nullptr, nullptr};
    uint16_t* qkv = nullptr;      // [(Hq_l + 2 Hkv_l) D, H]     merged (modeling.rs:187-204)
    uint16_t* o = nullptr;        // [H, Hq_l D]                 K-slice of o_proj under TP
    uint16_t* gate_up = nullptr;  // [2 I_l, H] rows interleaved gate_j, up_j
    uint16_t* down = nullptr;     // [H, I_l]
    float* ln1 = nullptr;         // [H] f32, (1 + w) folded for Qwen3.5
    float* ln2 = nullptr;
    float* qn = nullptr;          // [D] or null
    float* kn = nullptr;
    // Qwen3.5: full-attention layers store qkv as [q (Hq D) | gate (Hq D) | k | v] rows;
    // GDN layers use the fields below instead of qkv / o
    bool full = true;
    int gdn_idx = -1;             // index among the GDN layers
    uint16_t* in_proj = nullptr;  // [conv_dim + VD + 2 NV (padded to 128), H]: qkv | z | b | a
    uint16_t* out_proj = nullptr; // [H, VD]
    float* conv_w = nullptr;      // [conv_dim, 4]
    float* A_log = nullptr;       // [NV]
    float* dt_bias = nullptr;     // [NV]
    float* gnorm = nullptr;       // [Vd] plain RMSNormGated weight
    // quantised linears (GGUF / ISQ; loader_gguf.cpp).  q|k|v are merged into ONE segment when their ggml types
    // agree, otherwise kept as separate GEMV launches into the same qkv buffer; gate/up are row-interleaved
    // like the bf16 layout when their types agree, else two GEMVs + a SiLU*mul launch.
    QWeight q_qkv[3];
    int n_qkv = 0;
    int qkv_row0[3] = {0, 0, 0};
    QWeight q_o, q_gate_up, q_gate, q_up, q_down;
    QWeight q_in_proj, q_in_proj_z, q_out_proj;   // GDN: in_proj rows [qkv | z] quantised; the a / b gate rows stay bf16 (projection.rs:78-83)
    uint16_t* in_proj_ba = nullptr;  // [2 NV, H] bf16: b rows then a rows
    uint16_t* ba_pad = nullptr;      // int8 prompt pass: the same rows zero-padded to one 128-row bf16 GEMM tile
    bool split_gate_up = false;
};

struct Seq {
    bool used = false;
    int64_t len = 0;                 // cached tokens
    int32_t rope_delta = 0;          // MRoPE counter - cache position (non-zero after an image prompt)
    std::vector<int32_t> pages;      // page ids, one per kv_block_size tokens
};

// Qwen 3.5-VL vision tower (qwen3_5/config.rs VisionConfig; qwen3_5/vision.rs)
struct VisionCfg {
    bool present = false;
    int depth = 0, hidden = 0, heads = 0, inter = 0, patch = 16, tpatch = 2, merge = 2, in_ch = 3, out_hidden = 0, num_pos = 0;
    int act = 1;          // block MLP: 1 = gelu_pytorch_tanh, 2 = gelu (erf)
    int merger_act = 1;   // PatchMerger `xs.gelu()` (vision.rs:276): tanh form in the reference, erf in HF (CM_VISION_MERGER_GELU=erf)
    long long image_token = -1;
    std::vector<int> deepstack;   // blocks after which a DeepStack merger taps the hidden states (Qwen3-VL: [5, 11, 17])
    int patch_dim() const { return in_ch * tpatch * patch * patch; }
};
struct VisionBlockW {
    float *n1w = nullptr, *n1b = nullptr, *n2w = nullptr, *n2b = nullptr;
    uint16_t *qkv_w = nullptr, *proj_w = nullptr, *fc1_w = nullptr, *fc2_w = nullptr;
    float *qkv_b = nullptr, *proj_b = nullptr, *fc1_b = nullptr, *fc2_b = nullptr;
};
struct VisionW {
    uint16_t* patch_w = nullptr; float* patch_b = nullptr; uint16_t* pos_table = nullptr;
    std::vector<VisionBlockW> blocks;
    float *mn_w = nullptr, *mn_b = nullptr;
    uint16_t *mfc1_w = nullptr, *mfc2_w = nullptr; float *mfc1_b = nullptr, *mfc2_b = nullptr;
    struct Merger { float *n_w = nullptr, *n_b = nullptr, *fc1_b = nullptr, *fc2_b = nullptr; uint16_t *fc1_w = nullptr, *fc2_w = nullptr; };
    std::vector<Merger> deep;     // deepstack_merger_list.k: LayerNorm over the regrouped 4 x hidden row, fc1 + GELU, fc2
};

struct Rccl;         // dlopen'ed RCCL entry points + communicator, or the peer-store transport of an in-process group (tp.cpp)
struct PeerShared;   // state shared by the ranks of an in-process tensor-parallel group (tp.h)

struct Model {
    Config cfg;
    cm_opts opts{};
    int dev = 0, num_cu = 256;
    int tp = 1, rank = 0;
    // host-only "plan" instance (cm_tp_shard_plan): no device is touched -- dalloc hands out fake addresses and records them,
    // the loader's fetch calls are recorded instead of executed
    bool plan_only = false;
    size_t plan_bump = 0;
    std::vector<std::pair<size_t, size_t>> plan_allocs;      // (fake base address, bytes)
    // local (per-rank) shard geometry
    int Hq_l = 0, Hkv_l = 0, I_l = 0, V_l = 0, v0 = 0, kvh0 = 0, nrep = 1;
    int page = 64, max_seq = 0, max_pages_per_seq = 0, nsplit = 32;
    int64_t n_pages = 0;
    hipStream_t stream = nullptr;

    // weights
    uint16_t* embed = nullptr;     // [V, H] replicated
    uint16_t* lm_head = nullptr;   // [V_l, H] (points into embed when tied)
    bool lm_head_owned = false;    // lm_head is its own allocation of v_eff rows (false: an alias of embed -- tied, or the
                                   // hybrid family's fallback when the checkpoint has no lm_head.weight)
    float* norm = nullptr;
    std::vector<LayerW> layers;
    float* cos = nullptr;
    float* sin = nullptr;
    uint64_t weight_bytes = 0;
    std::vector<void*> allocs;
    std::vector<size_t> alloc_sizes;   // weight bytes of each allocation (0 for scratch)

    // paged KV pool: [L][2][n_pages][Hkv_l][page][D] f16 (default) / bf16 / f32, or int8 / int4 codes + scales: cm_opts.kv_dtype
    uint8_t* kv_pool = nullptr;
    size_t page_elems = 0;
    size_t kv_esize = 2;               // bytes per cached element for the roofline accounting (int4: counted as 1/2 below)
    bool kv_f32 = false;
    int kv_mode = KV_F16;              // internal page element type (dev_common.h KV_*: bf16, f32, int8 / int4 per-token symmetric
                                       // -- qwen3_5/kv_cache.rs:209-342 --, f16), mapped from cm_opts.kv_dtype in init_common
    bool kvq() const { return kv_mode == KV_INT8 || kv_mode == KV_INT4; }
    size_t kv_row_bytes = 0;           // bytes of one token row of one KV head
    size_t page_bytes = 0;             // bytes of one page of one (layer, K|V): rows + (quantised) f32 scales
    std::vector<int32_t> free_pages;
    std::vector<int32_t> page_ref;
    std::vector<Seq> seqs;
    int32_t* d_bt = nullptr;       // active block table on device [max_pages_per_seq]
    int32_t* h_bt = nullptr;       // pinned mirror
    int active_seq = -1;
    size_t active_pages_uploaded = 0;

    // vision tower (Qwen3.5-VL)
    VisionCfg vcfg;
    VisionW vw;
    int v_cap = 0;                 // patches the vision scratch is sized for
    float *vX = nullptr, *vQKV = nullptr, *vFeat = nullptr, *vCos = nullptr, *vSin = nullptr, *vPix = nullptr, *vW4 = nullptr;
    uint16_t *vK = nullptr, *vV = nullptr;   // [2 (bf16 hi, lo)][pages][heads][64][64] K / V scratch of one block
    size_t vkv_lo_off = 0;
    hipEvent_t v_ev0 = nullptr, v_ev1 = nullptr;   // around the tower's kernels of the last vision_encode (cm_debug_read "vision_ms")
    bool v_timed = false;
    float *vPartO = nullptr, *vPartML = nullptr;   // split-KV partials of the frame attention (VIT_KSPLIT runs of key tiles)
    int vit_ksplit = 2;                             // CM_VIT_KSPLIT
    uint16_t *vA_hi = nullptr, *vA_lo = nullptr, *vB_hi = nullptr, *vB_lo = nullptr, *vQ_hi = nullptr, *vQ_lo = nullptr;
    int32_t *vIdx = nullptr, *vBt = nullptr, *dMap = nullptr, *dPos3 = nullptr;
    const int32_t* pos3_dev = nullptr;   // set only while a VLM prefill runs
    const int32_t* splice_map_dev = nullptr;
    float* vDeep = nullptr;              // [n_deepstack][v_cap / merge^2 + 1][out_hidden] DeepStack feature maps of the last encode
    size_t deep_stride = 0;
    int deep_layers = 0;                 // set only while a VLM prefill runs: decoder layers li < deep_layers get vDeep[li] added
    int pos3_stride = 0;
    void ensure_vision_buffers(int n_patches);
    int vision_encode(const float* pix, size_t n_patches, const uint32_t* grid, size_t n_img);   // -> vFeat, returns rows
    void vlm_forward(int s, const uint32_t* ids, size_t n, size_t start_pos, const float* pix, size_t n_patches,
                     const uint32_t* grid, size_t n_img, float* logits_out, uint32_t* greedy_out);
    // Qwen3_5TextModel::embed_only / forward_embeds (qwen3_5/model.rs:368,430): the decoder over caller-supplied hidden rows
    void embed_tokens(const uint32_t* ids, size_t n, float* out);
    void forward_embeds(int s, const float* embeds, size_t n, const int32_t* pos3, size_t start_pos, float* logits_out, uint32_t* greedy_out);
    const float* embeds_host = nullptr;  // set only while forward_embeds runs: prefill() uploads these rows instead of gathering ids
    float* dEmb = nullptr;               // device staging of embed_tokens

    // per-sequence GDN state pools (Qwen3.5): [slots][gdn_layers][...]
    int gdn_layers = 0, in_proj_rows = 0, in_proj_pad = 0;
    float* conv_pool = nullptr;
    float* state_pool = nullptr;
    float* gdn_scratch = nullptr;      // decode step: raw y + partial sums of the 4 workgroups of a value head
    int* gdn_ticket = nullptr;
    size_t conv_slot_elems = 0, state_slot_elems = 0;
    void reset_gdn_state(int slot);

    // decode scratch (f32)
    float* x = nullptr;        // [H] residual stream
    float* y = nullptr;        // [H] TP partial
    float* qkv = nullptr;      // [(Hq_l+2Hkv_l) D]
    float* attn = nullptr;     // [Hq_l D]
    float* hbuf = nullptr;     // [I_l]
    float* logits = nullptr;   // [V_l * tp] (gathered under TP)
    float* part_o = nullptr;
    float* part_ml = nullptr;
    float* pmax = nullptr;
    int* pidx = nullptr;
    int lm_grid = 0;
    StepState* st = nullptr;
    uint32_t* ring = nullptr;          // device token ring
    static constexpr int RING = 4096;
    uint32_t ring_count = 0;           // host mirror of StepState.pad
    StepState* h_st = nullptr;         // pinned
    uint32_t* h_ring = nullptr;        // pinned
    float* h_logits = nullptr;         // pinned [V]

    // prefill scratch (allocated on first use; sized for one chunk)
    int chunk = 2048, chunk_pad = 2048;
    bool prefill_ok = false, prefill_split2 = true;
    int gemm256 = 1;           // prompt-pass / decode-group GEMMs: 0 = never an LDS-DMA kernel, 1 = automatic, 2 = always kernels_gemmw4.hip, 3 = kernels_gemm256.hip only (cm_debug_set("gemm256"))
    float* pX = nullptr;        // [chunk, H] f32 residual stream
    float* pWS = nullptr;       // split-K workspace of the prefill GEMMs: at most 1024 partial tiles of 128 x 128 f32
    static constexpr size_t gemm_ws_floats = (size_t)1024 * 128 * 128;
    float* pY = nullptr;        // [chunk, H] TP partial
    float* pQKV = nullptr;      // [chunk, qkv_rows] f32
    uint16_t *pXN_hi = nullptr, *pXN_lo = nullptr;     // [chunk_pad, H]
    uint16_t *pQ_hi = nullptr, *pQ_lo = nullptr;       // [chunk_pad, Hq_l D]
    uint16_t *pAT_hi = nullptr, *pAT_lo = nullptr;     // [chunk_pad, Hq_l D]
    uint16_t *pHH_hi = nullptr, *pHH_lo = nullptr;     // [chunk_pad, I_l]
    float* pGY = nullptr;       // [chunk, value_dim] f32 GDN output (Qwen3.5)
    uint32_t* d_ids = nullptr;
    uint32_t* h_ids = nullptr;

    // batched decode scratch (<= 8 sequences per step)
    static constexpr int MAXB = 128;           // sequences of one batched step: one 128-row M tile of the MFMA GEMM projections
    static constexpr int GEMV_MAXB = 64;       // ... on the batched GEMVs (2 / 4 / 8 L2-sharing groups of 8)
    int batch_max = 64;                        // CM_BATCH_MAX = 8 | 16 | 32 | 64 (A/B)
    int lm_head_gemm_min = 9;                  // CM_LM_HEAD_GEMM_MIN: groups of this many sequences or more run the head as an MFMA GEMM + row arg-max (0 = never)
    bool attn_outq = true;                     // CM_ATTN_OUTQ / cm_debug_set("attn_outq"): the single-split matrix-core attention of a quantised group also writes the Q8_0 blocks of its rows (A/B)
    int q_gemm_min = 8;                        // CM_Q_GEMM_MIN: quantised weights (Q8_0 layout): groups of this many sequences or more run their projections on the int8 matrix cores (0 = never)
    signed char* qx_codes = nullptr;           //   the group's activation rows as Q8_0 codes [QGEMM_MAXM][Kmax] ...
    float* qx_scales = nullptr;                //   ... and block scales [Kmax / 32][QGEMM_MAXM]
    signed char* qx_codes2 = nullptr;          //   second pair (the current one is always qx_codes / qx_scales: swapped when the
    float* qx_scales2 = nullptr;               //   unsplit gate|up GEMM wrote the next projection's codes into this one)
    // cm_debug_set("q_capture", 1) (tests): every set of Q8_0 activation rows an int8-MFMA projection of a decode group consumes is
    // copied to the host as it is produced -- records {K, rows, codes[rows][K], scales[rows][K / 32]} as floats, in consumption order --
    // so that the parity tests can check the quantiser's roundings and feed the oracle the codes the kernel multiplied
    bool q_capture = false;
    std::vector<float> q_cap;
    void q_capture_rows(int nb, int K, int xs = QGEMM_MAXM);
    int batch_gemm_min = 9;                    // CM_BATCH_GEMM_MIN: batched decode of this many sequences or more runs its projections as MFMA GEMMs (0 = never)
    StepState* stb = nullptr;          // device [MAXB]
    StepState* h_stb = nullptr;        // pinned [MAXB]
    int32_t* d_btb = nullptr;          // device [MAXB][max_pages_per_seq]
    int32_t* h_btb = nullptr;          // pinned
    float *xb = nullptr, *qkvb = nullptr, *attnb = nullptr, *hbb = nullptr, *logitsb = nullptr, *part_ob = nullptr, *part_mlb = nullptr, *pmaxb = nullptr;
    int* pidxb = nullptr;
    float* h_logitsb = nullptr;
    int ldq = 0;
    void ensure_batch_buffers();
    // after_group(g0, nb): called once per <= MAXB group after the stream is idle; logitsb[b * V] (b < nb) is valid
    void decode_batch(const int32_t* sq, const uint32_t* toks, size_t n, float* logits_out, uint32_t* greedy_out,
                      const std::function<void(size_t, int)>* after_group = nullptr);

    // device sampler scratch (model_sample.cpp; allocated on first use)
    unsigned long long* tk_cand = nullptr; size_t tk_cand_cap = 0;
    float* tk_in = nullptr; size_t tk_in_cap = 0;
    uint32_t* tk_hist = nullptr;       // [4096] radix-select histogram (kept zeroed between calls) + [4] select record
    uint32_t* tk_idx = nullptr;        // [512]
    float* tk_val = nullptr;           // [512]
    uint32_t* d_pen = nullptr;         // [2][PEN_CAP] distinct ids, counts
    uint32_t* h_pen = nullptr;         // pinned
    uint32_t* d_tok = nullptr;
    struct SampleReq { int slot; cm_sample_params p; const uint32_t* ctx; size_t n_ctx; bool true_div; float* dev_logits; };
    void sample_enqueue_rows(const SampleReq* rq, int n);      // every row in one set of launches (model_sample.cpp)
    struct SampleRowDev* d_stab = nullptr;                      // device / pinned row tables of sample_enqueue_rows
    struct SampleRowDev* h_stab = nullptr;
    bool sample_rows_on = true;                                 // cm_debug_set("sample_rows", 0): the per-row path (A/B, tests)
    uint32_t* h_tk = nullptr;          // pinned [1 + 512 + 512]
    float* sm_pmax = nullptr; int* sm_pidx = nullptr;
    static constexpr int PEN_CAP = 8192;
    void ensure_sampler();
    void gather_logits();              // TP: all-gather the vocab shards into `logits`
    void topk(const float* host_logits, size_t n, uint32_t k, uint32_t* idx_out, float* val_out);
    uint32_t sample(const cm_sample_params& p, const uint32_t* ctx, size_t n_ctx, bool true_div = false, float* dev_logits = nullptr);
    // pipelined form for the engine's decode rounds: enqueue one row per slot (own scratch, no host sync), then ONE sync
    static constexpr int SAMPLE_SLOTS = 128;    // = MAXB: every sampled row of a batched group has its own slot
    void sample_enqueue(int slot, const cm_sample_params& p, const uint32_t* ctx, size_t n_ctx, bool true_div, float* dev_logits);
    void sample_collect(int n_slots, uint32_t* tokens_out);
    unsigned long long* tk_cand_rows = nullptr; size_t tk_cand_row_cap = 0;
    bool logits_gathered = false;

    // quantised weights (GGUF: TP = 1; ISQ: also under TP for the dense family): embedding / lm_head tables + per-layer QWeights in LayerW
    bool quantized = false;
    bool gdn_chunked = false;          // GGUF value-head order (VHeadOrder::Chunked)
    bool quant_act_int = true;         // ggml vec_dot semantics: activations -> Q8_0 / Q8_K + integer dots (CM_QUANT_ACT=f32: exact dequant x f32)
    QWeight q_embed, q_lm_head;
    float *kshadow = nullptr, *vshadow = nullptr;   // int8/int4 KV prefill: dequantised f32 K/V of ONE layer, identity pages
    int32_t* d_ident_bt = nullptr;                  // [max_pages_per_seq] 0, 1, 2, ...
    // prompts over Q8_0-layout weights on the int8 matrix cores (default where every projection of the dense family qualifies;
    // CM_QUANT_PREFILL_INT8=0 / cm_debug_set("prefill_q8", 0) before the first prompt: the dequantised-to-bf16 GEMMs)
    bool q8_prefill_want = true, q8_prefill = false;
    float* pATf = nullptr;             // [chunk][Hq_l D] f32 attention rows of the pass (the o_proj quantiser's input)
    float* pHf = nullptr;              // [chunk][I_l] f32 silu(gate) * up rows of the pass
    int q8_xs = QGEMM_MAXM;            // floats between the scale rows of the code buffers in a prompt pass of more than 128 rows
    int prefill_chunk_rows() const;
    bool q8_prefill_eligible() const;
    uint16_t* wq_scratch = nullptr;    // [2][max N*K] bf16 hi | lo planes: one dequantised matrix at a time for the prefill GEMMs
    size_t wq_scratch_elems = 0;
    float* pGU = nullptr;              // [chunk][2 I_l] f32 gate|up sums of the two-pass (hi + lo operand) GEMM
    bool wq_cache = false;             // keep every dequantised matrix instead (CM_QUANT_PREFILL_CACHE=1, opt-in)
    bool gdn_ck_on = true;                      // cm_debug_set("gdn_chunked"): prompts of >= 64 tokens take the chunk-parallel Gated-Delta-Net scan
    float *gdn_pre_q = nullptr, *gdn_pre_k = nullptr, *gdn_pre_v = nullptr, *gdn_pre_bd = nullptr, *gdn_pre_g = nullptr, *gdn_ck = nullptr;   // GDN prefill scratch
    float* yb = nullptr;               // [MAXB][H] TP partial sums of the batched step
    int lm_gridb = 0;                  // row length of a pmaxb / pidxb slab
    float* gu_tmpb = nullptr;          // [MAXB][2 I] the same for the batched step
    float* gu_tmp = nullptr;           // [2 I] scratch when gate / up have different ggml types
    uint64_t quant_weight_bytes = 0;   // bytes of every quantised matrix read once per decoded token
    void debug_qgemv(int layer, const std::string& which, const float* x, size_t k, float* y, size_t n);
    void debug_qgemm(int layer, const std::string& which, const float* x, size_t m, size_t k, float* y, size_t n);
    void isq_q8_0(int mode = 8);       // in-situ quantisation of the loaded bf16 linears (ops/linear.rs:83-116): 8 Q8_0; 4 / 5 the Q4_0 / Q5_0
                                       // reference quantisers, stored in the Q8_0 layout (exact: q - 8 / q - 16 are int8 codes)
    void dfree(void* p);

    // persistent decode kernel (kernels_engine.hip): the whole token in one launch (attention inside), or per layer ONE
    // launch for o_proj -> gate||up -> down_proj -> next QKV around the separate attention kernels
    bool engine_on = false, engine_full = false;
    bool eng_head = false;                     // whole-token launch also runs the embedding row, the final norm + lm_head and the per-wave arg-max partials (CM_ENG_HEAD=0: off)
    float* eng_pmax = nullptr;                 //   [num_cu * stream waves] partials for argmax_final
    int* eng_pidx = nullptr;
    int eng_tune = (8 << 8) | (8 << 4);        // polling parameters of the persistent kernel (EngArgs::tune)
    int eng_dbg = 0;                           // CM_ENG_DBG (timing experiments only)
    int64_t eng_full_max_ctx = 8192;           // longer contexts: per-layer launches around the MFMA flash-decode kernel
                                               // (measured: whole-token launch 331 vs 322 tok/s at 6000, equal at 8192)
    EngPhase* eng_prog = nullptr;              // device [L][4]: QKV, o_proj, gate||up, down_proj
    EngAttnL* eng_attn = nullptr;              // device [L]
    unsigned long long* eng_gran[ENG_NEDGE] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // multi-sequence prompt pass: device tables of the segments (Model::prefill_multi) so that RoPE / KV append / causal attention
    // of ALL sequences are one launch each per layer instead of one per sequence (cm_debug_set("prefill_seg_batch", 0): per sequence)
    PrefillSegDev* d_segtab = nullptr;
    int32_t* d_rowseg = nullptr;
    int2* d_tiles = nullptr;
    int seg_ntiles = 0;
    bool seg_batch = true, seg_tables_ok = false;
    int eng_chunk = 2048;      // dependency chunk of the persistent kernel (2048; 1024 for widths that are only multiples of 1024)
    bool hybrid_engine = false; // CM_ENGINE_HYBRID=1 (or cm_opts.engine = 1): per-layer persistent chain for the hybrid family -- measured slower, opt-in
    int eng_gpw_res = 0, eng_xf_total = 0;
    bool engine_eligible(std::string* why = nullptr) const;
    bool engine_full_eligible() const;
    void build_engine();
    EngArgs engine_args_common() const;
    EngArgs engine_args(int li) const;
    EngArgs engine_args_full() const;
    void engine_trace(float* out, size_t n);   // debug: microsecond timestamps of one traced launch; runs on a scratch sequence state
    void engine_check();                       // after a host sync: throws if a launch timed out (StepState.rsv[2]); the kernel is
                                               // then already switched off for this handle (engine_failed), later calls work
    bool engine_failed();                      // syncs; true if a launch timed out: error word cleared, persistent kernel disabled,
                                               // graphs dropped -- the caller replays its step(s) on the per-projection launches
    [[noreturn]] void hybrid_engine_abort();   // hybrid family: a timed-out chain launch advanced the recurrent state -- no replay, CM_ERR_DEVICE
    bool engine_capable = false;               // build_engine succeeded (cm_debug_set("engine", 1) may switch it back on)
    bool engine_full_capable = false;
    size_t eng_gsz[ENG_NEDGE] = {0, 0, 0, 0, 0, 0};   // granules per edge buffer (bounds of cm_debug_read("eng_*"))
    uint32_t eng_fail_code = 0;                // code of the last timeout (0: none)
    void drop_graphs();                        // forget every captured decode step (a switch that changes what a step enqueues)

    hipGraph_t graph[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};          // one captured decode step per attention variant
    hipGraphExec_t graph_exec[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // (4 = whole-token persistent launch)
    bool graph_ok[5] = {false, false, false, false, false};
    bool default_prefill_split2() const { return opts.prefill_split != 1; }      // 0 (default) and 2: bf16 hi + lo, the 1e-3 parity mode, on every kind of weights
    bool quant_prefill = true, no_prefill = false, use_mfma_gemv = true;   // CM_QUANT_PREFILL, CM_NO_PREFILL, CM_GEMVM (read at create)
    bool tp_graph = true, rccl_warm = false;   // capture RCCL collectives into the decode graph (CM_TP_GRAPH=0: eager)
    int attn_variant = 0;          // 0: split-KV + combine kernels, 1: per-head blocks + merge fused into o_proj
    int nsplit_mfma = 64;          // token splits of variant 3 (MFMA flash-decode, bf16 KV, head_dim 128, long contexts);
                                   // variant 2 = the same kernel with `nsplit` (32) splits for mid-size contexts
    int64_t attn_mfma_wide_min = 8192;   // measured: 32 splits win up to 4 K (3.26 vs 3.30 ms), 64 at 32 K (3.96 vs 4.08)
int prefill_lo_mask = 0;              // EXPERIMENT (cm_debug_set "prefill_lo_mask"): bit 0 qkv, 1 o_proj, 2 gate||up, 3 down: that GEMM drops the lo plane
    int gdn_defer_max = 4096;             // ... for value dims up to this many elements
    bool gdn_ba_fused = true;             // quantised hybrid decode: the GDN step computes its head's bf16 a / b projections itself (cm_debug_set("gdn_ba_fused"))
    bool gdn_defer_norm = true;           // CM_GDN_DEFER_NORM / cm_debug_set("gdn_defer_norm"): gated RMSNorm in out_proj's prologue
    int attn_batch_ns_min = 1;            // fewest token splits per sequence in the batched decode attention (CM_ATTN_BATCH_NS_MIN)
    int64_t attn_mfma_min_batch = 64;     // matrix-core decode attention from this context on when the group fills the chip (CM_ATTN_MFMA_MIN_BATCH)
        int64_t attn_mfma_min = 768;   // contexts of at least this many tokens use the MFMA kernel (CM_ATTN_MFMA_MIN; 0 = never);
                                   // Qwen3-8B ms/token split vs MFMA(32 splits): 256: 3.056 / 3.068, 1 K: 3.11 / 3.085, 4 K: 3.44 / 3.26
    int attn_ns = 2;               // token splits per head of variant 1
    int attn_splits_force = 0;     // cm_debug_set("attn_splits"): one split count for the single AND the batched VALU attention (tests)
    int64_t attn_heads_max = 0;    // contexts up to this many tokens use variant 1 (CM_ATTN_HEADS_MAX; 0 = never: measured
                                   // slower on MI355X at every context tried)
    bool use_graph = true;

    std::unique_ptr<Rccl> rccl;
    PeerShared* peer_shared = nullptr;   // set by the TpGroup before alloc_runtime: this Model is one rank of an in-process group
    std::string err;

    ~Model();

    // ---- construction ----
    void init_common(const std::string& config_json, const cm_opts* o);
    void alloc_runtime();
    template <typename T> T* dalloc(size_t n, bool count_weight = false);

    // ---- kv / sequences ----
    std::vector<int> kv_index;     // layer -> index among the softmax-attention layers (-1: GDN layer)
    int n_kv_layers = 0;
    void* kpool(int layer) const { return kv_pool + ((size_t)kv_index[(size_t)layer] * 2 + 0) * n_pages * page_bytes; }
    void* vpool(int layer) const { return kv_pool + ((size_t)kv_index[(size_t)layer] * 2 + 1) * n_pages * page_bytes; }
    int seq_alloc();
    void seq_free(int s);
    int seq_fork(int src);
    void seq_truncate(int s, size_t new_len);
    void cow_page(int s, size_t idx);
    void ensure_pages(int s, int64_t upto_len);
    void activate(int s);
    Seq& seq(int s);
    uint64_t kv_bytes() const;

    // ---- forward ----
    void enqueue_decode_step(bool advance);     // one token from st->token at st->pos
    void enqueue_lm_head(bool advance);         // final norm + lm_head + arg-max on x
    void enqueue_quant_layer(int li);           // dense layer over GGUF / ISQ weights
    void ensure_prefill_buffers();
    void ensure_gemm_workspace();
    void prefill(const uint32_t* ids, size_t n, size_t start_pos);   // active sequence, pages ensured
    struct PrefillSeg { int row0, S, start_pos, seq, rope_delta; const int32_t* bt; };   // rows of ONE sequence inside a prompt pass
    void prefill_layers(int S, const PrefillSeg* segs, int nseg, size_t off);
    // whole prompts of several sequences (each from position 0, together <= prefill_chunk tokens) in ONE pass over the weights
    void prefill_multi(const int32_t* sq, const uint32_t* const* ids, const size_t* lens, size_t n_items, uint32_t* greedy_out);
    void lm_head_rows(int nb, bool want_rows);                        // xb rows -> logitsb rows, stb[b].next
    void run_decode_step(bool advance, int64_t ctx_len);   // graph replay or eager; ctx_len = tokens attended (pos + 1)
    void forward(int s, const uint32_t* ids, size_t n, size_t start_pos, float* logits_out, uint32_t* greedy_out);
    void generate(const uint32_t* prompt, size_t n_prompt, const cm_gen_config* g, uint32_t* out, size_t* n_out,
                  cm_token_cb cb, void* user);
    void bench_decode(uint32_t first, size_t k, uint32_t* toks, float* ms);
    void debug_fill_kv(size_t ctx, uint64_t seed);
    void bench_kernel(const std::string& which, size_t iters, float* ms, uint64_t* bytes);
    uint64_t decode_bytes_per_token(size_t ctx) const;
    void fetch_logits(float* out);
};

// loaders (loader.cpp)
void load_from_dir(Model& m, const std::string& dir);
void load_synthetic(Model& m, uint64_t seed);
std::string tp_shard_plan_json(const std::string& config_json, int tp_size, int tp_rank);   // loader.cpp, host only
std::string gguf_config_json(const std::string& path);     // loader_gguf.cpp
void load_from_gguf(Model& m, const std::string& path);

}  // namespace cm
