/*
 * crane_mi355.h -- C ABI of the MI355X-native transformer inference path.
 *
 * Drop-in boundary for lucasjinreal/Crane's model hot path.  Each entry point
 * replaces one method of the two Rust traits the reference routes every token
 * through (citations are into the reference tree):
 *
 *   B1  trait ModelForCausalLM      crane-core/src/generation/based.rs:5-34
 *   B2  trait ModelBackend          crane-serve/src/engine/backend.rs:30-147
 *
 * and the inherent methods of crane-core/src/models/qwen3/model.rs:45-349
 * (qwen3_5/model.rs twin).  The reference exports no extern "C" host API
 * today; INTEGRATION.md shows the `impl ModelBackend for Mi355Backend` shim a
 * maintainer adds on the Rust side (bindgen/`extern "C"` block) -- it is
 * 1:1 with this header.
 *
 * Conventions
 *   - plain pointers and sizes only; caller owns every output buffer;
 *   - every function returns CM_OK (0) or a negative cm_status; the message
 *     for the last failure on a handle is cm_last_error(handle) (maps to
 *     anyhow!(..) on the Rust side); creation failures use
 *     cm_last_global_error();
 *   - a cm_model is NOT thread-safe (matches `&mut self`); it may be moved
 *     between threads (matches `Send`); N handles = N replicas, no globals;
 *   - the library owns device weights, the paged KV-cache pool, the
 *     autoregressive loop and the sampler; tokenizer / chat template stay on
 *     the caller's side (B2 tokenizer()/eos_token_id()).
 */
#ifndef CRANE_MI355_H
#define CRANE_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CM_ABI_VERSION 3

typedef enum cm_status {
    CM_OK = 0,
    CM_ERR_INVALID = -1,      /* bad argument / shape / state                 */
    CM_ERR_IO = -2,           /* model dir / config / safetensors problem     */
    CM_ERR_UNSUPPORTED = -3,  /* architecture or option not implemented       */
    CM_ERR_DEVICE = -4,       /* HIP / RCCL runtime error                     */
    CM_ERR_OOM = -5,          /* KV pool or HBM exhausted                     */
    CM_ERR_RANGE = -6         /* position / token id out of range             */
} cm_status;

/* KV cache element type.
 * F16 (the default, 0): K/V pages hold IEEE binary16.  The reference keeps its cache in the model dtype
 * (modules/kv_cache.rs:38-101) -- F32 on its CPU path, the forward the parity bar is stated against -- and a bf16 page
 * (8 significand bits) moves Qwen3-8B-width logits by 1.06e-3 relative to that forward (measured on the HF fixture
 * tests/golden/qwen3_qwen3-8b-2l.npz).  A binary16 page costs the same 2 bytes per element, keeps 11 significand bits
 * (1.35e-4 on the same fixture) and converts in one instruction (v_cvt_f32_f16) or feeds the f16 matrix-core
 * instructions directly; values saturate at +-65504 (K is bounded by the QK-norm, V is a projection of a normalised row).
 * BF16: the model dtype of a bf16 checkpoint on the reference's GPU path.  F32: 4-byte pages (bit-comparable with the
 * f32 CPU forward; not available inside the persistent decode kernel).
 * INT8 / INT4 = KvCache::Quant (qwen3_5/kv_cache.rs:209-342): per-token symmetric codes
 * (scale = amax/qmax + 1e-8, code = round(x/scale) + 128|8, int4 nibble-packed lo + 16*hi) + one f32 scale per
 * (token, kv head); dequantisation is fused into the attention kernel instead of re-materialising the cache. */
typedef enum cm_kv_dtype { CM_KV_F16 = 0, CM_KV_F32 = 1, CM_KV_INT8 = 2, CM_KV_INT4 = 3, CM_KV_BF16 = 4 } cm_kv_dtype;

/* Options of Model::new / select_device / create_backend
 * (qwen3/model.rs:45-106, crane-serve/src/lib.rs:432-499,
 *  crane-serve/src/engine/model_factory.rs:471-586).  Zero-initialise and set
 * what you need: 0 always means "default". */
typedef struct cm_opts {
    uint32_t abi_version;      /* CM_ABI_VERSION                                     */
    int32_t  device;           /* HIP device ordinal of THIS rank (default 0)        */
    int32_t  tp_rank;          /* tensor-parallel rank   (default 0)                 */
    int32_t  tp_size;          /* tensor-parallel degree (default 1)                 */
    const void* tp_unique_id;  /* 128-byte id from cm_tp_unique_id(), same on all ranks */
    uint32_t max_seq_len;      /* longest sequence (0: min(max_position_embeddings, 32768)) */
    uint32_t max_seqs;         /* concurrently allocated sequences (default 8)       */
    uint32_t kv_block_size;    /* tokens per KV page (default 64)                    */
    uint64_t kv_pool_tokens;   /* pool capacity in tokens (0: max_seqs*max_seq_len)  */
    int32_t  kv_dtype;         /* cm_kv_dtype; default CM_KV_F16 (binary16 pages)    */
    int32_t  use_graph;        /* 0 default(on), 1 on, -1 off: hipGraph decode step  */
    uint32_t prefill_chunk;    /* tokens per prefill chunk (default 2048,            */
                               /*   PREFILL_CHUNK_SIZE engine/mod.rs:65)             */
    int32_t  prefill_split;    /* prompt pass of the MFMA GEMMs: 0 (default) / 2 = parity mode: activations as bf16 hi + lo (two    */
                               /*   products: the 1e-3 bound of a bf16 checkpoint) and, over QUANTISED weights on the dequantised    */
                               /*   path (K-quants, CM_QUANT_ACT=f32, the hybrid family), the dequantised operand as bf16 hi + lo    */
                               /*   too (a second GEMM pass over the lo plane); 1 = plain bf16 activations and a single rounded      */
                               /*   operand plane: the fast approximate mode (3.2-3.7e-3 of the f32 oracle on the dequantised        */
                               /*   weights, tools/probes/quant_prefill_deviation.py).  Q8_0-layout weights (GGUF Q8_0 / Q4_0 /     */
                               /*   Q5_0, every ISQ mode) of the dense family in the integer-dot activation mode do not take this    */
                               /*   path at all: their prompts run on the int8 matrix cores (ops/linear.rs:18-51 semantics).        */
    uint32_t isq;              /* in-situ quantisation of the linears at load: 0 none, CM_ISQ_Q8_0 / _Q4_0 / _Q5_0 */
                               /*   (--quant / CRANE_ISQ, ops/linear.rs:53-116; also read from CRANE_ISQ) */
    int32_t  engine;           /* persistent decode kernel (one launch per token): 0 default (on when the shapes */
                               /*   allow it), 1 require (cm_create fails otherwise), -1 off.  The kernel runs   */
                               /*   one workgroup per CU and its workgroups wait on each other, so it needs the  */
                               /*   device to itself while it runs: no second handle, process or serialising     */
                               /*   profiler on the same GPU.  If that is violated the bounded spins time out:   */
                               /*   the handle then switches the kernel off for good, replays the step on the    */
                               /*   per-projection launches and carries on (cm_engine_active turns 0).           */
    uint32_t debug_flags;      /* CM_DEBUG_* bits; test hooks only, never set in production                      */
    uint32_t tp_mode;          /* how the tp_size ranks are hosted (round 4, CM_ABI_VERSION 3; same struct size):   */
                               /*   CM_TP_SPMD (0): this handle is ONE rank (tp_rank on `device`); one process per  */
                               /*     GPU, every process issues the same cm_* calls, RCCL id in tp_unique_id.       */
                               /*   CM_TP_IN_PROCESS (1): this ONE handle owns ALL tp_size ranks -- one shard per    */
                               /*     device of tp_devices, driven by library worker threads; every cm_* call on    */
                               /*     the handle (and a cm_engine on it) runs on all ranks and returns rank 0's     */
                               /*     results.  What a single ModelBackend object in crane-serve's one engine       */
                               /*     thread (crane-serve/src/lib.rs:1129-1132, engine/backend.rs:30) can host.     */
                               /*     tp_rank / tp_unique_id are ignored.                                           */
    const int32_t* tp_devices; /* CM_TP_IN_PROCESS: tp_size device ordinals (NULL: device, device + 1, ...).        */
                               /*   tp_size DISTINCT devices, or ONE ordinal tp_size times (test mode: every rank   */
                               /*   on one GPU -- the sharding and the exchange steps on a 1-GPU box; exchange      */
                               /*   through the peer-store collective, RCCL refuses two ranks per device; more than */
                               /*   4 ranks on one device need GPU_MAX_HW_QUEUES >= tp_size exported before the HIP */
                               /*   runtime starts -- one hardware queue per rank stream -- else CM_ERR_INVALID).   */
                               /*   A group whose ranks fail asymmetrically (one rank errors, the others do not) is */
                               /*   DEAD: every later call on the handle returns CM_ERR_DEVICE; destroy + re-create. */
    uint32_t tp_collective;    /* CM_TP_IN_PROCESS: 0 default (RCCL over xGMI on distinct devices), CM_TP_COLL_RCCL, */
                               /*   CM_TP_COLL_PEER (one-shot push all-reduce over peer-visible memory, csrc/       */
                               /*   kernels_tp.hip: no RCCL call in the token loop).  EXPERIMENTAL on distinct       */
                               /*   devices: never run on > 1 GPU; needs uncached / fine-grained device memory for   */
                               /*   the inboxes and fails with CM_ERR_UNSUPPORTED where the runtime has none.        */
    uint32_t reserved[1];
} cm_opts;

enum { CM_TP_SPMD = 0u, CM_TP_IN_PROCESS = 1u };
enum { CM_TP_COLL_DEFAULT = 0u, CM_TP_COLL_RCCL = 1u, CM_TP_COLL_PEER = 2u };

/* cm_opts.debug_flags.  TP_LOCAL: no communicator is created and every collective is a local no-op, so ONE rank of a
 * tensor-parallel model can run alone and be compared with the oracle on its shard (the logits are partial sums!).
 * FORCE_RCCL: a tp_size = 1 model routes its reductions through a 1-rank RCCL communicator (exercises the RCCL calls
 * on a single GPU). */
enum { CM_DEBUG_TP_LOCAL = 1u, CM_DEBUG_FORCE_RCCL = 2u };

enum { CM_ISQ_NONE = 0, CM_ISQ_Q4_0 = 2, CM_ISQ_Q5_0 = 6, CM_ISQ_Q8_0 = 8 };   /* ggml type ids */

typedef struct cm_model cm_model;

/* ---- lifecycle ------------------------------------------------------------ */

/* Model::new / from_pretrained (qwen3/model.rs:45-106): reads config.json and
 * every *.safetensors (sharded index honoured, utils/utils.rs:16-57) from
 * `model_dir`, merges QKV and gate||up at load (modeling.rs:187-204,582-588),
 * ties lm_head to the embedding when the config says so (modeling.rs:786-794).
 * A path ending in ".gguf" is loaded as a GGUF checkpoint instead (ModelFormat::Auto,
 * qwen3/model.rs:55-71,108-152): Q8_0 / Q4_0 / Q5_0 (widened exactly into the Q8_0 layout) / Q4_K / Q6_K / Q3_K (widened exactly into Q6_K) matrices stay
 * quantised in HBM (LinearLayer::Quantized, ops/linear.rs:18-51) and are decoded inside the GEMV; decode groups of 25 or more
 * sequences over Q8_0-layout matrices run on the int8 matrix cores.  Other ggml types: CM_ERR_UNSUPPORTED. */
int cm_create(const char* model_dir, const cm_opts* opts, cm_model** out);

/* Same model object from a config.json *string* with deterministic synthetic
 * bf16 weights generated on the device (crane_amd/synth.py documents the
 * generator; used by the benchmarks -- there is no network for checkpoints). */
int cm_create_synthetic(const char* config_json, uint64_t seed, const cm_opts* opts, cm_model** out);

void cm_destroy(cm_model* m);

const char* cm_last_error(const cm_model* m);
const char* cm_last_global_error(void);

/* 128-byte RCCL unique id for opts.tp_unique_id (rank 0 creates, the caller
 * broadcasts it, e.g. over torch.distributed / the engine's own channel). */
int cm_tp_unique_id(void* out128);

/* The config.json the loader derives from a GGUF file's metadata and tensor directory (qwen3/model.rs:138-147,
 * qwen3_5/model.rs:196-287: layer kinds from blk.i.ssm_a, tied head from a missing output.weight, ...).  Host only, no
 * device needed.  *needed = bytes including the terminator; cap = 0 only queries the size.  Errors: cm_last_global_error. */
int cm_gguf_config(const char* path, char* json_out, size_t cap, size_t* needed);

/* Shard discovery + safetensors header parsing of a checkpoint directory (utils/utils.rs:16-57: model.safetensors.index.json,
 * else model.safetensors, else every *.safetensors), host only: {"name": {"dtype", "shape", "nbytes", "fnv1a"}, ...} with
 * an FNV-1a hash of each tensor's bytes.  Same buffer protocol as cm_gguf_config. */
int cm_checkpoint_inspect(const char* model_dir, char* json_out, size_t cap, size_t* needed);

/* Tensor-parallel shard plan of rank `tp_rank` of `tp_size` for the model described by config.json text: every copy the loader
 * makes for that rank -- {"tensor", "rows", "cols", "row0", "nrows", "col0", "ncols", "dst", "dst_off", "dst_stride"}: rows
 * [row0, row0 + nrows) x columns [col0, col0 + ncols) of checkpoint tensor `tensor` ([rows, cols]) land at element offset
 * dst_off of device allocation number dst, dst_stride elements per row -- plus the rank's geometry (Hq_l, Hkv_l, kvh0, I_l, V_l,
 * v0, NK_l, NV_l, weight_bytes).  Host only (the loader's own code path over a recording source; no device is touched): how a
 * multi-process launcher or a test checks the partition of SURVEY 8(e) without GPUs.  Same buffer protocol as cm_gguf_config. */
int cm_tp_shard_plan(const char* config_json, int32_t tp_size, int32_t tp_rank, char* json_out, size_t cap, size_t* needed);

/* ---- introspection (ModelBackend::num_layers/dtype/..., backend.rs:47-60) --- */
size_t cm_num_layers(const cm_model* m);
size_t cm_vocab_size(const cm_model* m);
size_t cm_hidden_size(const cm_model* m);
size_t cm_max_seq_len(const cm_model* m);
/* active_kv_cache_bytes (backend.rs:83-87, qwen3/model.rs:201) */
uint64_t cm_kv_bytes(const cm_model* m);
/* bytes of weights resident in HBM on this rank */
uint64_t cm_weight_bytes(const cm_model* m);
/* algorithmic HBM bytes one decode step reads at context `ctx` on this rank
 * (SURVEY.md section 8(d) formula; used by bench.py's roofline object) */
uint64_t cm_decode_bytes_per_token(const cm_model* m, size_t ctx);

/* ranks this handle reduces over: 1 without tensor parallelism, tp_size once the communicator (RCCL, or the peer-store
 * group of CM_TP_IN_PROCESS) is up, 0 when the collectives are local no-ops (CM_DEBUG_TP_LOCAL) -- bench.py asserts it
 * equals --gpus */
int cm_tp_ranks(const cm_model* m);
/* decode path of this handle: 0 per-projection launches; 1 persistent kernel, one launch per layer around the separate attention
 * kernels; 2 persistent kernel, the whole token (every projection and the attention of every layer) in one launch */
int cm_engine_active(const cm_model* m);

/* ---- single-sequence path (the implicit sequence of B1/B2) ----------------- */

/* ModelBackend::forward_step / Model::forward_step (backend.rs:41,
 * qwen3/model.rs:177-184).  Processes `n` tokens at KV position `start_pos`
 * (n > 1: chunked MFMA prefill; n == 1: decode step) and writes the logits of
 * the LAST position only -- that is all the reference ever produces
 * (modeling.rs:1032-1035) -- as `vocab` f32 values to host buffer `logits_out`. */
int cm_forward_step(cm_model* m, const uint32_t* ids, size_t n, size_t start_pos, float* logits_out);

/* Engine greedy fast path (engine/sampling.rs:191-210 + gpu_argmax,
 * kernels/cuda/fused_ops.cu:251-382): same forward, device arg-max
 * (lowest index wins ties), 4-byte result. */
int cm_forward_step_greedy(cm_model* m, const uint32_t* ids, size_t n, size_t start_pos, uint32_t* token_out);

/* ModelBackend::clear_kv_cache (backend.rs:44) */
void cm_clear_kv(cm_model* m);

/* ModelBackend::warmup (backend.rs:60; qwen3/model.rs:261-267:
 * generate(&[45,546,456], 5 tokens) then clear). */
int cm_warmup(cm_model* m);

/* ---- generation loop (B1) --------------------------------------------------- */

/* GenerationConfig (generation/mod.rs:62-99).  temperature < 0 means None
 * (greedy arg-max: qwen3/model.rs:284); temperature >= 0 samples on the device
 * (top-k / top-p / Gumbel-max and the penalties of sampling.rs:169-478, see the
 * device-side sampler section below); the draw uses its own counter-based RNG
 * stream (the reference pins none). */
typedef struct cm_gen_config {
    uint32_t max_new_tokens;
    float    temperature;        /* < 0 : None */
    float    top_p;              /* < 0 : None */
    float    repetition_penalty; /* 1.0 : off (model.rs:306-315) */
    uint32_t repeat_last_n;
    int64_t  eos_token_id[4];    /* -1 : unused slot (multi-id EOS: qwen3_5/model.rs:871-877) */
    uint32_t sync_every;         /* greedy w/o penalty: tokens enqueued per host sync (0: 1) */
    uint32_t top_k;              /* 0 : unset (64 when top_p is active; sampling.rs:263-268) */
    uint32_t seed_lo, seed_hi;   /* sampling seed (based.rs GenerationConfig has none: candle seeds 299792458) */
    float    frequency_penalty;  /* 0 : off (sampling.rs:456-470) */
    float    presence_penalty;   /* 0 : off */
    uint32_t reserved[2];
} cm_gen_config;

/* TokenStreamer::append (generation/streamer.rs:7-10); return non-zero to stop. */
typedef int (*cm_token_cb)(void* user, uint32_t token);

/* ModelForCausalLM::generate (based.rs:7-31, qwen3/model.rs:275-349): clears
 * the KV cache, feeds the whole prompt at start_pos 0, then one token per
 * step; returns prompt ++ generated in `tokens_out` (capacity >= n_prompt +
 * max_new_tokens); `*n_out` = total length. */
int cm_generate(cm_model* m, const uint32_t* prompt, size_t n_prompt, const cm_gen_config* cfg,
                uint32_t* tokens_out, size_t* n_out, cm_token_cb cb, void* user);

/* ---- device-side sampler (replaces crane-serve/src/engine/sampling.rs:169-373 and
 *      crane_core::ops::topk_indices, crane-core/src/ops/mod.rs + kernels/cuda/topk.cu) ---- */

typedef struct cm_sample_params {
    float    temperature;        /* <= 0 : greedy (sampling.rs:214-217) */
    float    top_p;              /* <= 0 or >= 1 : off */
    uint32_t top_k;              /* 0 : off; clamped to min(64, vocab) like sampling.rs:268 */
    float    repetition_penalty; /* 1 : off */
    float    frequency_penalty;  /* 0 : off */
    float    presence_penalty;   /* 0 : off */
    uint32_t repeat_last_n;      /* penalty window over the tail of `context` (0: whole context) */
    uint32_t draw;               /* counter of the uniform stream; callers pass the step index */
    uint64_t seed;
    uint32_t reserved[6];
} cm_sample_params;

/* Sequence::sample (sampling.rs:169): sample the next token from the logits of the LAST forward call, which
 * are still resident in HBM (no [V] D2H copy).  Penalties are applied IN PLACE to those device logits over
 * the last `repeat_last_n` tokens of `context` (apply_penalties_inplace, sampling.rs:422-478). */
int cm_sample(cm_model* m, const cm_sample_params* p, const uint32_t* context, size_t n_context, uint32_t* token_out);

/* crane_core::ops::topk_indices: exact top-k, total order value-descending / index-ascending, -0.0 == +0.0
 * (rocm_kernels.rs:86-200).  `logits` = host vector of length n, or NULL for the last forward's logits
 * (n ignored).  1 <= k <= min(512, n).  `val_out` may be NULL. */
int cm_topk(cm_model* m, const float* logits, size_t n, uint32_t k, uint32_t* idx_out, float* val_out);

/* copy the (possibly penalised) device logits of the last forward call back to the host: [vocab] */
int cm_read_logits(cm_model* m, float* logits_out);

/* ---- paged-KV sequences (replaces get/set_kv_caches + pad/stack/extract:
 *      backend.rs:66-147, qwen3/modeling.rs:1094-1378) ------------------------- */

/* Sequence 0 always exists: it is the implicit sequence of cm_forward_step. */
int cm_seq_alloc(cm_model* m, int32_t* seq_out);
int cm_seq_free(cm_model* m, int32_t seq);
/* share the prefix pages of `src` (copy-on-write of the last partial page) */
int cm_seq_fork(cm_model* m, int32_t src, int32_t* seq_out);
/* tokens currently cached for `seq` (cache_seq_len, modeling.rs:1121-1127) */
int64_t cm_seq_len(const cm_model* m, int32_t seq);
/* drop cached tokens beyond `new_len` (preemption / re-prefill, engine/mod.rs:430-504) */
int cm_seq_truncate(cm_model* m, int32_t seq, size_t new_len);

/* forward_step on an explicit sequence */
int cm_seq_forward(cm_model* m, int32_t seq, const uint32_t* ids, size_t n, size_t start_pos,
                   float* logits_out /* [vocab] or NULL */, uint32_t* greedy_out /* or NULL */);

/* step_batch_decode (backend.rs:107-121, modeling.rs:1202-1234): one token
 * for each of `n` sequences at their own positions, no padding, no mask.
 * logits_out [n, vocab] (or NULL), greedy_out [n] (or NULL).  Up to 128 sequences
 * share one pass over bf16 weights on one rank (17 or more, also under tensor
 * parallelism: the projections run as MFMA GEMMs over the batch rows), up to 64
 * elsewhere (quantised weights in the default integer-dot mode, tensor parallelism
 * when tp_size divides vocab_size), up to 8 for the hybrid family's quantised layers;
 * larger n: in groups;
 * the remaining combinations decode one sequence at a time. */
int cm_decode_batch(cm_model* m, const int32_t* seqs, const uint32_t* last_tokens, size_t n,
                    float* logits_out, uint32_t* greedy_out);

/* Whole prompts of n sequences in ONE pass over the weights: sequence seqs[i] is cleared and prefilled with ids[i][0 .. lens[i])
 * from position 0 (what cm_seq_forward(seq, ids, n, 0, ...) does for one), together at most prefill_chunk tokens and 128
 * sequences.  The row-wise work (norms, every GEMM) runs once over all rows; RoPE / KV append / causal attention / the
 * Gated-Delta-Net scan run per sequence on its own pages and state.  greedy_out[i] = arg-max of sequence i's last position;
 * logits_out (may be NULL) = [n, vocab] f32 rows.  The continuous-batching engine's prefill step (cm_engine_opts.batch_prefill):
 * a 128-token prompt alone occupies one m-tile of every GEMM and costs what 1024 rows cost.  Not available over int8 / int4 KV
 * pages (CM_ERR_UNSUPPORTED). */
int cm_prefill_batch(cm_model* m, const int32_t* seqs, const uint32_t* const* ids, const size_t* lens, size_t n,
                     float* logits_out, uint32_t* greedy_out);

/* ---- vision-language path (Qwen 3.5-VL; reference crane-core/src/models/qwen3_5/{vision,vlm}.rs) -------- */

/* image placeholder token of the checkpoint (config.json image_token_id), -1 when the model has no vision tower */
int64_t cm_image_token_id(const cm_model* m);

/* Qwen3_5VisionModel::forward (vision.rs:558-584).  pixel_values: host f32 [n_patches, C*T*P*P] exactly as
 * PreprocessorConfig::process emits them (processor.rs:114-210: merge-block-major rows, (channel, temporal, y, x)
 * inside a row); grid_thw: host u32 [n_images, 3].  Writes the merged image tokens
 * [n_patches / merge^2, out_hidden] (f32) to features_out (may be NULL) and their count to rows_out. */
int cm_vision_encode(cm_model* m, const float* pixel_values, size_t n_patches, const uint32_t* grid_thw, size_t n_images,
                     float* features_out, size_t* rows_out);

/* Qwen3_5VLModel::forward (vlm.rs:250-285) on sequence `seq`: encodes the images, splices their rows over the
 * image placeholder tokens of `ids` (splice_image_features, vlm.rs:433-468), builds the 3-axis MRoPE positions
 * (build_position_ids, vlm.rs:190-241) and prefills.  Later cm_seq_forward / cm_forward_step / cm_generate-style
 * decode calls on the same sequence continue with the MRoPE counter (decode_step, vlm.rs:294-301). */
int cm_vlm_forward(cm_model* m, int32_t seq, const uint32_t* ids, size_t n, size_t start_pos, const float* pixel_values,
                   size_t n_patches, const uint32_t* grid_thw, size_t n_images, float* logits_out, uint32_t* greedy_out);

/* Qwen3_5TextModel::embed_only (qwen3_5/model.rs:368-370): the embedding rows of `ids` as host f32 [n, hidden]
 * (what the multimodal wrapper splices image features into, vlm.rs:250-285). */
int cm_embed_tokens(cm_model* m, const uint32_t* ids, size_t n, float* embeds_out);

/* Qwen3_5TextModel::forward_embeds (qwen3_5/model.rs:430-510): every decoder layer over caller-built hidden rows
 * embeds [n, hidden] (host f32) at explicit MRoPE positions pos3 [3, n] (host i32; axes T, H, W as build_position_ids
 * emits them, vlm.rs:190-241), appended to sequence `seq` at start_pos == cm_seq_len(seq); causal over the cached prefix
 * and these rows; logits of the LAST row only (and / or its arg-max).  pos3 == NULL: positions continue the sequence's
 * own counter on all three axes (text-only rows: identical to cm_seq_forward on the same tokens).  Afterwards the
 * sequence's MRoPE counter is max(pos3) + 1, so cm_seq_forward / cm_forward_step decode steps continue like
 * Qwen3_5VLModel::decode_step (vlm.rs:294-301).  No DeepStack injection (the reference's forward_embeds has none; the
 * Qwen3-VL path with DeepStack is cm_vlm_forward). */
int cm_forward_embeds(cm_model* m, int32_t seq, const float* embeds, size_t n, const int32_t* pos3, size_t start_pos,
                      float* logits_out, uint32_t* greedy_out);

/* ---- image preprocessor (host only; PreprocessorConfig::process, qwen3_5/processor.rs:114-210) ----------------------- */

/* preprocessor_config.json (processor.rs:20-35): size.shortest_edge / longest_edge are the min / max PIXEL counts */
typedef struct cm_preproc_config {
    uint32_t patch_size, temporal_patch_size, merge_size, reserved;
    uint64_t min_pixels;         /* size.shortest_edge */
    uint64_t max_pixels;         /* size.longest_edge  */
    float    image_mean[3];
    float    image_std[3];
} cm_preproc_config;

/* smart_resize (processor.rs:64-88): both sides to the NEAREST multiple of patch_size * merge_size, then scaled into
 * [min_pixels, max_pixels] */
int cm_image_smart_resize(const cm_preproc_config* cfg, uint32_t height, uint32_t width, uint32_t* h_out, uint32_t* w_out);

/* One RGB8 image [height][width][3] -> pixel_values [n_patches, 3 * temporal_patch_size * patch_size^2] f32 exactly as
 * cm_vision_encode / cm_vlm_forward expect them (rows merge-block-major, a row = (channel, temporal copy, y, x), the
 * still image duplicated over the temporal patch) + grid_thw = (1, h / patch, w / patch).  Bicubic (Catmull-Rom,
 * PIL / HF `resample: 3`) resize when smart_resize changes the size.  pixel_values_out == NULL: only *n_patches_out
 * and grid_thw_out are written.  Errors: cm_preprocess_last_error(). */
int cm_image_preprocess(const cm_preproc_config* cfg, const uint8_t* rgb, uint32_t height, uint32_t width, float* pixel_values_out,
                        size_t cap_floats, uint32_t grid_thw_out[3], size_t* n_patches_out);
const char* cm_preprocess_last_error(void);

/* ---- continuous-batching engine on the paged KV pool ------------------------------
 * Replaces InferenceEngine's scheduling core (crane-serve/src/engine/mod.rs:622-1057 execute_step /
 * step_prefill / step_decode_batch / evict_if_needed) and Scheduler (scheduler.rs:67-98): FIFO, prefill-priority
 * (one whole prompt per step while running < max_running), otherwise ONE batched decode round over every running
 * sequence.  No pad/stack/extract KV copies: sequences own pages.  Tokenizer, HTTP and channels stay with the
 * caller; the engine speaks token ids and events.  Not thread-safe (like `&mut self`, backend.rs:30). */
typedef struct cm_engine cm_engine;

typedef struct cm_engine_opts {
    uint32_t max_running;        /* Scheduler::max_running (scheduler.rs:31); 0: max_seqs - 1 */
    uint32_t repeat_last_n;      /* penalty window; 0: 64 (engine/mod.rs:588) */
    uint64_t seed;               /* base of the per-request sampling seeds (sampling.rs:480-491 uses the clock) */
    int32_t  batch_prefill;      /* 0 default (on), 1 on, -1 off: several waiting prompts share one pass over the weights     */
                                 /*   (cm_prefill_batch) instead of one prompt per step -- same tokens, same event order       */
    uint32_t reserved[7];
} cm_engine_opts;

/* EngineRequest (engine/types.rs:11-24) */
typedef struct cm_request {
    const uint32_t* tokens;      /* prompt */
    size_t   n_tokens;
    uint32_t max_tokens;
    float    temperature;        /* < 0 : None (== 1.0, sampled); 0 : greedy (sampling.rs:180-183) */
    float    top_p;              /* < 0 : None */
    uint32_t top_k;              /* 0 : None */
    float    repetition_penalty; /* 1 : off */
    float    frequency_penalty;
    float    presence_penalty;
    int64_t  eos_token_id[4];    /* -1 : unused */
    uint64_t seed;               /* 0 : derived from cm_engine_opts.seed and the request id */
    uint32_t reserved[4];
} cm_request;

enum { CM_EV_TOKEN = 0, CM_EV_FINISHED = 1, CM_EV_ERROR = 2 };
enum { CM_FINISH_NONE = 0, CM_FINISH_STOP = 1, CM_FINISH_LENGTH = 2, CM_FINISH_CANCELLED = 3 };

/* EngineResponse (engine/types.rs:53-66) without the detokenised text */
typedef struct cm_engine_event {
    uint64_t req_id;
    uint32_t kind;               /* CM_EV_* */
    uint32_t token;              /* CM_EV_TOKEN */
    uint32_t finish_reason;      /* CM_EV_FINISHED: CM_FINISH_* (Sequence::finish_reason, sequence.rs:117-125) */
    uint32_t prompt_tokens;      /* CM_EV_FINISHED */
    uint32_t completion_tokens;  /* CM_EV_FINISHED */
    int32_t  error;              /* CM_EV_ERROR: cm_status */
} cm_engine_event;

typedef struct cm_engine_stats {
    uint64_t waiting, running, completed, failed, preemptions;
    uint64_t prompt_tokens, completion_tokens, prefill_steps, decode_rounds;
    uint64_t free_pages, total_pages;
    uint64_t reserved[5];
} cm_engine_stats;

int  cm_engine_create(cm_model* m, const cm_engine_opts* opts, cm_engine** out);
void cm_engine_destroy(cm_engine* e);
/* accept_request (engine/mod.rs:525-599): rejects empty prompts and prompts longer than max_seq_len,
 * clamps max_tokens to max_seq_len - prompt_len (effective_max_tokens :507-513). */
int  cm_engine_submit(cm_engine* e, const cm_request* r, uint64_t* req_id_out);
int  cm_engine_cancel(cm_engine* e, uint64_t req_id);
/* One scheduling decision + its execution (Scheduler::schedule + execute_step).  Writes up to `cap` events and
 * keeps the rest queued for the next call; *n_events = 0 with no work left means idle. */
int  cm_engine_step(cm_engine* e, cm_engine_event* events, size_t cap, size_t* n_events);
/* Up to `max_steps` scheduling decisions in one call (stops early when the engine is idle or the next step's events
 * might not fit in `cap`): amortises the caller's per-step overhead; same event stream as repeated cm_engine_step. */
int  cm_engine_step_many(cm_engine* e, size_t max_steps, cm_engine_event* events, size_t cap, size_t* n_events);
int  cm_engine_has_work(const cm_engine* e);
int  cm_engine_get_stats(const cm_engine* e, cm_engine_stats* out);
const char* cm_engine_last_error(const cm_engine* e);

/* ---- measurement hooks (bench.py / tests) ----------------------------------- */

/* Enqueue `k` greedy decode steps for sequence 0 starting from its current
 * last token, timing the region with HIP events on the model's own stream.
 * Returns tokens in `tokens_out[k]`, total milliseconds in *ms_out. */
int cm_bench_decode(cm_model* m, uint32_t first_token, size_t k, uint32_t* tokens_out, float* ms_out);

/* Launch one hot-path kernel `iters` times back to back, cycling over the layers so every
 * launch streams different weights from HBM (no L2 / Infinity-Cache reuse), timed with HIP
 * events on the model's stream.  which: "qkv" | "o" | "gate_up" | "down" | "lm_head".
 * *ms_out = average milliseconds per launch, *bytes_out = algorithmic bytes per launch. */
int cm_bench_kernel(cm_model* m, const char* which, size_t iters, float* ms_out, uint64_t* bytes_out);

/* Fill sequence 0's KV cache with deterministic synthetic K/V up to `ctx`
 * tokens without running prefill (bench set-up only). */
int cm_debug_fill_kv(cm_model* m, size_t ctx, uint64_t seed);

/* Copy an internal device buffer to host for kernel-level parity tests.
 * what: "hidden" (f32 [H] residual after the last forward), "logits". */
int cm_debug_read(cm_model* m, const char* what, float* out, size_t n);

/* Run ONE quantised projection of `layer` on a host vector: y = W_q . x (plain prologue, store epilogue), in the
 * model's activation mode (integer dot or f32).  which: "qkv0".."qkv2" (segments), "o", "gate_up" (interleaved rows
 * 2j = gate_j, 2j+1 = up_j) | "gate" | "up", "down", "lm_head".  Kernel-level parity hook for tests. */
int cm_debug_qgemv(cm_model* m, int32_t layer, const char* which, const float* x, size_t k, float* y, size_t n);

/* The same for `rows` activation rows x [rows][k] through the int8-MFMA GEMM that decode groups and prompt passes use
 * (csrc/kernels_quant_gemm.hip: row quantiser of the weight's vec-dot type -- Q8_0 blocks for Q8_0-layout weights, Q8_K blocks for
 * Q4_K -- then the GEMM, plain input, store epilogue): y [rows][n].  The batched counterpart of `QMatMul::forward` on a [rows, k]
 * input (candle quantized matmul behind ops/linear.rs:18-51).  CM_ERR_UNSUPPORTED for tensors that are not on that path (Q6_K). */
int cm_debug_qgemm(cm_model* m, int32_t layer, const char* which, const float* x, size_t rows, size_t k, float* y, size_t n);

/* Test hook: the peer-store all-reduce / all-gather of an in-process group alone (csrc/kernels_tp.hip): n_ranks threads on
 * `device`, |iters| rounds over `count` elements, every sum and gather checked on the host; iters < 0 starts the ranks' epoch
 * counter at 0xFFFFFFFD so that the rounds cross its 32-bit wrap.  Returns the number of wrong elements (0 = pass), < 0 on error
 * (cm_last_global_error).  More than 4 ranks on one device need GPU_MAX_HW_QUEUES >= n_ranks exported before the HIP runtime starts. */
long cm_debug_peer_selftest(int32_t n_ranks, int32_t device, int32_t iters, int32_t count);

/* Test hook (host logic only, no device): what the int8-MFMA GEMM of a quantised decode group (csrc/kernels_quant_gemm.hip; it replaces the
 * per-sequence `QMatMul::forward` calls of a batched step, candle quantized matmul behind ops/linear.rs:53-116) would do for `m` activation
 * rows over an [n][k] Q8_0-layout matrix on `num_cu` CUs with a split-K workspace of `ws_floats` f32 (0: none): epi 0 store, 1 residual
 * add, 2 SiLU(gate) * up (bits 8 .. 15 of `epi`: the ggml type of the weights -- 0 = the Q8_0 layout, 12 = Q4_K, 14 = Q6_K, round 6).  out[0..7] = { ok (0: the batched GEMV takes the projection), unsplit store, waves per workgroup, activation
 * rows per workgroup, 32-blocks per K group, K groups, K split, workgroups }.  Slice i of the split covers groups [i G / ks, (i + 1) G / ks). */
int cm_debug_qgemm_plan(int32_t m, int32_t n, int32_t k, int32_t epi, uint64_t ws_floats, int32_t num_cu, int64_t out[8]);

/* Test hook: flips a path switch of a live model (the environment switches of the same names are read once, at
 * cm_create).  "no_prefill" = 1: prompts run token by token through the decode kernels; "quant_prefill" = 0: prompts over
 * quantised weights run through the integer-dot decode kernels instead of the dequantised MFMA GEMMs; "attn_outq" = 0 / 1: the single-split matrix-core
 * decode attention of a quantised group writes the Q8_0 blocks of its output rows itself (1, default) or leaves them to a quantiser launch; "prefill_split" = 1 / 0 / 2:
 * cm_opts.prefill_split of the live model (plain bf16 / bf16 hi + lo prompt activations and dequantised operands); "prefill_q8" = 0 (before the
 * first prompt pass): prompts over Q8_0-layout weights on the dequantised GEMMs instead of the int8 matrix cores; "batch_gemm_min" = n: cm_decode_batch
 * runs the projections of n or more sequences as MFMA GEMMs (0 = never: batched GEMVs, rows bit-equal to cm_forward_step);
 * "attn_splits" = n > 0:
 * the VALU decode attention uses n token splits per kv head in the single AND the batched step (their automatic counts differ,
 * which changes the order of the split merge), 0 = automatic; "prefill_split" = -1 restores cm_opts.prefill_split;
 * "attn_mfma_min" / "attn_mfma_wide_min" / "attn_heads_max" / "attn_ns": the context thresholds and split counts that pick the
 * decode-attention kernel (CM_ATTN_* read at cm_create); "engine" = 0 / 1 and "engine_full" = 0 / 1: the persistent decode
 * kernel off / on and its per-layer / whole-token mode (only where cm_create found the shapes eligible); "quant_act_int" = 1 / 0
 * and "vision_merger_gelu" = 1 (tanh) / 2 (erf): CM_QUANT_ACT and CM_VISION_MERGER_GELU of the live model; "gemm256" = 0 / 1: the
 * LDS-DMA GEMM of the prompt pass and of large decode groups; "lm_head_gemm_min" = n: decode groups of n or more sequences run
 * lm_head as one GEMM + row arg-max (0 = never); "sample_rows" = 0 / 1: the engine's sampled rows through the per-row sampler /
 * one set of launches for all rows (same tokens); "tp_graph" = 0 / 1: RCCL collectives launched eagerly / captured into the
 * decode hipGraph (CM_TP_GRAPH); "prefill_seg_batch" = 0 / 1: cm_prefill_batch launches RoPE / KV append / attention once per
 * sequence / once for all sequences of the pass; "gdn_defer_norm" = 0 / 1: the Gated-Delta-Net step normalises itself / leaves the
 * gated RMSNorm to out_proj's prologue; "prefill_lo_mask" (measurement only, tools/lo_mask_probe.py): bit 0 QKV, 1 o_proj, 2 gate||up,
 * 3 down_proj -- that GEMM of the prompt pass drops the lo plane of its activations.
 * cm_debug_read("engine_trace") launches the persistent kernel several times on the live state of sequence 0: the K/V rows at
 * the current position and the residual stream are overwritten -- clear the sequence afterwards. */
int cm_debug_set(cm_model* m, const char* key, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* CRANE_MI355_H */
