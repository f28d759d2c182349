// Host-visible launch interface of the gfx950 kernels (internal; the public ABI is
// include/crane_mi355.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_common.h"

#include <atomic>
#include <mutex>

namespace cm {

// "first launch on this device" latch for hipFuncSetAttribute: function attributes are per device, and since round 4 one process
// may drive several devices from several threads (in-process tensor-parallel groups).  run(f) calls f once per device; a thread
// that finds the device's bit set knows the attribute call has completed.
struct DevOnce {
    std::atomic<unsigned long long> done{0};
    std::mutex mu;
    template <typename F> void run(F&& f) {
        int d = 0;
        (void)hipGetDevice(&d);
        const unsigned long long bit = 1ull << (d & 63);
        if (done.load(std::memory_order_acquire) & bit) return;
        std::lock_guard<std::mutex> g(mu);
        if (done.load(std::memory_order_relaxed) & bit) return;
        f();
        done.fetch_or(bit, std::memory_order_release);
    }
};

enum { PRO_PLAIN = 0, PRO_RMSNORM = 1, PRO_ATTNCOMB = 2, PRO_GDNNORM = 3 };
enum { EPI_STORE = 0, EPI_RESADD = 1, EPI_SILUMUL = 2, EPI_ARGMAX = 3 };

struct GemvArgs {
    const uint16_t* W;     // [N, ldw] bf16, K contiguous
    const float* x;        // [K] f32
    const float* nw;       // [K] f32 RMSNorm weight, (1 + w) already folded for Qwen3.5 (PRO_RMSNORM)
    float* y;              // output (see epilogues)
    const float* res;      // [N] residual (EPI_RESADD); may alias y
    float* pmax;           // [grid] (EPI_ARGMAX)
    int* pidx;             // [grid]
    int N, K, ldw;
    int idx_base;          // added to row index for arg-max (vocab shard offset)
    float eps;
    // PRO_ATTNCOMB: x = part_o [Hq][ns][D] of attn_decode_head_kernel, merged while staging
    const float* part_ml = nullptr;   // [Hq][ns][2]
    const float* gate = nullptr;      // [Hq * D] or null
    int ns = 0, dshift = 0;           // D == 1 << dshift
    // PRO_GDNNORM: x = RAW y of the Gated-Delta-Net step [NV * 128]; the gated RMSNorm of each 128-wide value head
    // (y * rsqrt(mean(y^2) + eps) * w * silu(z), ops/gdn/layer.rs:226-238) is applied while staging
    const float* gdn_z = nullptr;     // [NV * 128] z gate of this token (inside the in_proj output)
    const float* gdn_w = nullptr;     // [128] RMSNormGated weight
};

struct AttnDecArgs {
    const float* qkv;            // projection output of this token (f32)
    const float* qnw;            // [D] f32 (1 + w folded for Qwen3.5) or null
    const float* knw;
    const float* cos;            // [max_pos, rot/2] f32
    const float* sin;
    const float* gate;           // [Hq * D] f32 output gate (Qwen3.5: y * sigmoid(gate)) or null
    const StepState* st;
    const int32_t* block_table;  // [max_pages_per_seq]
    void* kpool;                 // this layer: [pages][Hkv][PAGE][D] bf16 (or f32)
    void* vpool;
    float* part_o;               // [Hq][nsplit][D]
    float* part_ml;              // [Hq][nsplit][2]
    float* out1 = nullptr;       // ONE token split per sequence (launch_attn_decode*: nsplit == 1): the kernel normalises (and gates) itself and
    int out1_stride = 0;         //   writes out1[seq * out1_stride + h * D + d] -- no partials, no combine launch
    uint16_t* out1_hi = nullptr; //   ... or, when set (the caller asked attn_decode_single_split()), as bf16 hi + lo planes [seq][out1_cols]: the A
    uint16_t* out1_lo = nullptr; //   operand of the o_proj GEMM of a large decode group (split_rows2d's arithmetic), instead of the f32 rows
    int out1_cols = 0;
    int qkv_ns = 1;              // qkv rows as qkv_ns f32 slices qkv_slice floats apart (the K-split partials of the int8 qkv GEMM, added in order by the
    size_t qkv_slice = 0;        //   matrix-core kernel's prologue: no reduction launch); 1 = plain rows
    signed char* out1_q = nullptr;  // ... and ALSO (matrix-core kernel, single split) as Q8_0 blocks of the rows: codes [seq][out1_cols] + block scales
    float* out1_qd = nullptr;       //   [out1_cols / 32][QGEMM_MAXM] -- quant_rows_q8_kernel's arithmetic: the int8 o_proj GEMM of a quantised group reads them
    int q_off, k_off, v_off;     // element offsets of q / k / v inside qkv
    int qkv_stride, bt_stride;   // batched step: per-sequence strides of qkv rows and block tables
    int Hkv, page, max_pages, rot_dim;
    size_t page_bytes = 0;       // int8 / int4 KV: bytes of one page (codes [Hkv][PAGE][row] + f32 scales [Hkv][PAGE])
    float eps, scale;
};

// Gated Delta Net layer state machine (kernels_gdn.hip)
constexpr int GDN_CK = 64;                                        // tokens per chunk of the chunk-parallel Gated-Delta-Net prompt scan
constexpr int GDN_CK_FLOATS = 2 * GDN_CK * 128 + GDN_CK * GDN_CK + GDN_CK;
struct GdnArgs {
    const float* proj;           // [S, proj_stride] f32: qkv (conv_dim) | z (NV*V) | b (NV) | a (NV)
    const float* conv_w;         // [conv_dim, 4] f32
    float* conv_pool;            // [slots][gdn_layers][2 (parity)][conv_dim][3] f32
    float* state_pool;           // [slots][gdn_layers][NV][K][V] f32
    const float* A_log;          // [NV]
    const float* dt_bias;        // [NV]
    const float* gnorm_w;        // [V] plain RMSNormGated weight
    float* out;                  // [S, out_stride] f32 gated-normed y
    const StepState* st;         // decode: start_pos and slot are read from device state
    int proj_stride, out_stride;
    int start_pos, slot;         // used when st == nullptr (prefill)
    int S, NV, vpg, key_dim, layer_idx, gdn_layers;
    float* pre_q = nullptr;      // prefill scratch (three-pass path): q^ [S][key_dim], k^ [S][key_dim], v [S][NV][V],
    float* pre_k = nullptr;      //   {beta, decay} [S][NV][2]; null: always the fused kernel
    float* pre_v = nullptr;
    float* pre_bd = nullptr;
    float* pre_g = nullptr;      //   g = log(decay) [S][NV] (the chunk-parallel scan works in log space); null: not written
    float* ck = nullptr;         // chunk-parallel scan scratch: [chunks][NV][GDN_CK_FLOATS] (W | U0 | P | cumulative g); null: sequential scan
    float* gdn_scratch = nullptr;   // decode step on 4 workgroups per head: [n_seq][NV][V + 4] raw y + partial sums of squares
    int* gdn_ticket = nullptr;      //   and [n_seq][NV] arrival tickets (zero between launches); null: one workgroup per head
    int defer_norm = 0;         // decode step: out = RAW y, the gated RMSNorm is the out_proj GEMV's prologue (PRO_GDNNORM) -- no
                                //   cross-workgroup hand-off inside the step
    int chunked = 0;            // value-head order: 0 Interleaved (HF: key head = v / vpg), 1 Chunked (llama.cpp GGUF: v % NK; ops/gdn/config.rs:13-22)
    int n_seq, batch_proj_stride, batch_out_stride;   // batched decode step (grid.y)
    float eps;
    // decode step over QUANTISED weights (round 6): the a / b gate projections stay bf16 (ops/gdn/projection.rs:78-83) and were a GEMV
    // launch of their own per layer; with ba_w set the step computes ITS head's two dot products itself -- b rows then a rows [2 NV][ba_H]
    // bf16, against RMSNorm(ba_x row of the sequence) * ba_nw -- and ignores the b / a columns of proj
    const uint16_t* ba_w = nullptr;
    const float* ba_x = nullptr;     // [n_seq][ba_H] f32 residual stream
    const float* ba_nw = nullptr;    // [ba_H] input-norm weight
    int ba_H = 0;
};

// ---- prefill (S > 1) ----
enum { GEPI_STORE = 0, GEPI_RESADD = 1, GEPI_SILUMUL = 2, GEPI_ACT_SPLIT = 3, GEPI_PARTIAL = 4 /* internal: split-K partial tile */ };

struct GemmArgs {
    const uint16_t* A_hi;   // [Mpad, K] bf16 activations (hi term)
    const uint16_t* A_lo;   // [Mpad, K] lo term or null (plain bf16 activations)
    const uint16_t* W;      // [N, K] bf16
    const uint16_t* W_lo = nullptr;   // [N, K] bf16 lo plane of the operand (W + W_lo = a dequantised ggml matrix to 2^-17): launch_gemm adds A_hi . W_lo^T
                                      // in a second pass (GEPI_SILUMUL then needs gu_tmp)
    float* gu_tmp = nullptr;          // [M][N] f32 scratch of a GEPI_SILUMUL launch with W_lo
    float* C;               // GEPI_STORE: C[m*ldc+n] = acc; GEPI_RESADD: C[m*ldc+n] += acc
    uint16_t* H_hi;         // GEPI_SILUMUL: [M, N/2] bf16 hi (+lo) of silu(gate)*up
    uint16_t* H_lo;
    const float* bias;      // [N] f32 added before the epilogue, or null
    int act;                // GEPI_ACT_SPLIT: 0 identity, 1 gelu (tanh form), 2 gelu (erf form)
    int M, N, K, ldc;
    float* ws;              // split-K workspace (or null: never split) of ws_floats f32; launch_gemm decides the split
    size_t ws_floats;
    int ksplit_cap = 0;        // split-K of the 128-wide kernel: at most this many blocks (0: the default 768; the vision tower passes 512)
    int ksplit;             // set by launch_gemm
    int dbg = 0;            // kernels_gemmw4.hip timing experiments (CM_GEMMW4_DBG); 0 in production
    int wide256 = 1;        // 0: never the LDS-DMA kernels (kernels_gemm256.hip / kernels_gemmw4.hip), 1: automatic, 2: always kernels_gemmw4.hip,
                            // 3: kernels_gemm256.hip only -- A/B switch, cm_debug_set("gemm256")
    // GEPI_RESADD with ldc == N only: ALSO write RMSNorm(C row) * norm_w as bf16 hi + lo planes [M][N] -- the A operand of the NEXT
    // GEMM (rmsnorm_rows_kernel's arithmetic).  Fused into the split-K reduction launch when the GEMM splits K (a large decode
    // group: one launch less per projection), a separate rmsnorm_rows launch otherwise -- launch_gemm does either.
    const float* norm_w = nullptr;
    const float* norm_b = nullptr;   // non-null: LayerNorm with bias (layernorm_rows_kernel's arithmetic, the vision tower) instead of RMSNorm
    uint16_t* norm_hi = nullptr;
    uint16_t* norm_lo = nullptr;
    float norm_eps = 0.f;
};

// one sequence of a multi-sequence prompt pass (Model::prefill_multi): rows [row0, row0 + S) of the pass at positions start_pos...,
// page table at block_table + bt_off
struct PrefillSegDev { int row0, S, start_pos, rope_delta, bt_off, pad0, pad1, pad2; };

struct QkRopeArgs {
    const float* qkv;            // [S, (Hq + 2 Hkv) D] f32
    const float* qnw;
    const float* knw;
    const float* cos;
    const float* sin;
    const int32_t* block_table;
    void* kpool;
    void* vpool;
    uint16_t* q_hi;              // [Spad, Hq, D] bf16, scaled by 1/sqrt(D)
    uint16_t* q_lo;
    int Hq, Hkv, page, start_pos;
    int row_stride, q_off, k_off, v_off, rot_dim;   // layout of one token's projection row
    const int32_t* pos3;         // MRoPE positions (T, H, W): pos3[axis * pos3_stride + s], or null (position = start_pos + s + rope_delta)
    int rope_delta = 0;          // rotary position - cache position of a sequence that holds an image prompt (vlm.rs:294-301)
    int pos3_stride, sec_h, sec_w;   // mrope_section[1], [2]
    float eps, scale;
    // int8 / int4 KV: codes + scales go to kpool / vpool (page_bytes layout), the DEQUANTISED row goes to the f32 shadow
    // (identity page table) that the prefill attention reads
    size_t page_bytes = 0;
    float* kshadow = nullptr;
    float* vshadow = nullptr;
    // several sequences in one launch: row s of the pass belongs to segs[rowseg[s]] (null: one sequence, start_pos / block_table above)
    const PrefillSegDev* segs = nullptr;
    const int32_t* rowseg = nullptr;
};

struct AttnPreArgs {
    const uint16_t* q_hi;
    const uint16_t* q_lo;
    const int32_t* block_table;
    const void* kpool;
    const void* vpool;
    size_t kv_lo_off = 0;        // KV_BF16X2: element offset of the lo halves behind the hi halves in kpool / vpool
    uint16_t* out_hi;            // [Spad, Hq * D] bf16 hi (+lo) -> A operand of the o_proj GEMM
    uint16_t* out_lo;
    float* out_f32 = nullptr;    // causal kernels: write f32 rows [S, Hq * D] here INSTEAD of the bf16 planes (int8 prompt pass: the o_proj quantiser's input)
    const float* gate;           // [S, gate_stride] f32 (Qwen3.5 output gate) or null
    int gate_stride;
    int S, Hq, Hkv, nrep, page, start_pos;
    int causal, kv_lo, kv_hi;    // causal = 0: bidirectional over tokens [kv_lo, kv_hi) (ViT frame)
    // several sequences in one causal launch: tiles[i] = {segment, query tile} ordered by ASCENDING key-tile count (walked from the
    // back: longest first); q / out / gate / block_table above are those of the whole pass
    const PrefillSegDev* segs = nullptr;
    const int2* tiles = nullptr;
    int ntiles = 0;
    int ksplit = 1;              // bidirectional frames only: runs of key tiles per query tile (partials merged by a second kernel)
    float* part_o = nullptr;     // [ksplit][S][Hq][D] un-normalised partial outputs
    float* part_ml = nullptr;    // [ksplit][S][Hq][2] running max, running sum
};

void launch_embed_rows(const uint16_t* emb, const uint32_t* ids, float* x, int S, int H, int V, hipStream_t s);
void launch_rmsnorm_rows(const float* x, const float* w, uint16_t* hi, uint16_t* lo, int S, int H, float eps,
                         hipStream_t s);
void launch_qknorm_rope_kv(const QkRopeArgs& a, int D, int S, int kv_mode, hipStream_t s);
void launch_kvq_dequant_prefix(const void* kpool, const void* vpool, const int32_t* block_table, float* kshadow, float* vshadow,
                               int tokens, int Hkv, int page, int D, int kv_mode, size_t page_bytes, hipStream_t s);
void launch_split_rows(const float* x, uint16_t* hi, uint16_t* lo, size_t n, hipStream_t s);
void launch_split_rows2d(const float* x, int ldx, uint16_t* hi, uint16_t* lo, int rows, int cols, hipStream_t s);   // cols % 4 == 0, ldx % 4 == 0
void launch_add_rows(float* x, const float* y, size_t n, hipStream_t s);
bool launch_gemm(const GemmArgs& a, int epi, hipStream_t s);
int gemm256_rows(bool split, int M);       // rows of that kernel's tile for an M-row launch (parity mode: 64 up to 64 rows, else 128; plain bf16: 256)
bool launch_gemm256(const GemmArgs& a, int epi, int bn, hipStream_t s);   // kernels_gemm256.hip; a.ksplit set by the caller (launch_gemm)
int gemmw4_rows(bool split);               // rows of the one-wave-per-SIMD kernel's tile (kernels_gemmw4.hip): 128 with hi + lo activations, 256 plain
bool launch_gemmw4(const GemmArgs& a, int epi, int bn, hipStream_t s);    // same contract as launch_gemm256
void launch_attn_prefill(const AttnPreArgs& a, int D, int kvt, hipStream_t s);   // kvt: KV_BF16 | KV_F16 | KV_F32 (what the kernel reads)

// ---- vision tower (kernels_vision.hip) ----
void launch_layernorm_rows(const float* x, const float* w, const float* b, uint16_t* hi, uint16_t* lo, int N, int H, float eps,
                           hipStream_t s);
void launch_pos_embed_add(float* x, const uint16_t* table, const int32_t* idx, const float* wts, int N, int H, hipStream_t s);
void launch_vit_rope_kv(const float* qkv, const float* cs, const float* sn, uint16_t* q_hi, uint16_t* q_lo, uint16_t* kpool,
                        uint16_t* vpool, size_t lo_off, int N, int heads, float scale, hipStream_t s);
void launch_splice_rows(float* dst, const float* src, const int32_t* map, int S, int H, hipStream_t s);
void launch_add_rows_map(float* dst, const float* src, const int32_t* map, int S, int H, hipStream_t s);

// ---- batched decode (kernels_decode_batch.hip) ----
struct GemvBArgs {
    const uint16_t* W;     // [N, ldw] bf16
    const float* x;        // [MB, ldx] f32
    const float* nw;       // [K] f32 RMSNorm weight (PRO_RMSNORM)
    float* y;              // [MB, ldy]
    const float* res;      // [MB, ldy] (EPI_RESADD)
    float* pmax;           // [MB, grid] (EPI_ARGMAX)
    int* pidx;
    int N, K, ldw, ldx, ldy, n_seq, idx_base;
    float eps;
    // PRO_ATTNCOMB (see GemvArgs): x = part_o [MB][Hq][ns][D], ldx = Hq * ns * D
    const float* part_ml = nullptr;   // [MB][Hq][ns][2]
    const float* gate = nullptr;      // [MB][gate_stride]
    int ns = 0, dshift = 0, gate_stride = 0;
};
int gemvb_grid(int N, int K, int num_cu);
// matrix-core variant (kernels_decode_mfma.hip): <= 64 sequences (2 / 4 / 8 groups of <= 8 above 8)
bool gemvm_ok(int epi, int n_seq, int K);
int gemvm_nkt(int K);
int gemvm_grid(int N, int K, int num_cu, int n_seq = 1);
void launch_gemvm(int pro, int epi, const GemvBArgs& a, int grid, hipStream_t s);
void launch_gemvb(int pro, int epi, const GemvBArgs& a, int grid, hipStream_t s);
// per row of logits [n_rows][ld] (n valid columns): nblk partial (max, lowest index of the max) pairs -> pmax / pidx [row][nblk]
void launch_argmax_rows(const float* logits, int ld, int n, int idx_base, float* pmax, int* pidx, int nblk, int n_rows, hipStream_t s);

// ---- decode ----
int gemv_rows_per_group(int K);
int gemv_grid(int N, int K, int num_cu, bool allow_nw5 = true);      // < 0: -blocks of five-wave workgroups (launch_gemv decodes it)
void launch_gemv(int pro, int epi, const GemvArgs& a, int grid, hipStream_t s);
void launch_embed_row(const uint16_t* emb, const StepState* st, float* x, int H, int V, int n_seq, hipStream_t s);
void launch_set_state(StepState* st, uint32_t token, int32_t pos, int32_t slot, int32_t rope_delta, hipStream_t s);
void launch_argmax_final(const float* pmax, const int* pidx, int n, StepState* st, uint32_t* ring,
                         int ring_mask, int advance, int n_seq, hipStream_t s, int slabs = 1, size_t slab_stride = 0);
bool attn_decode_single_split(int nsplit, int D);     // launch_attn_decode(_mfma) will write the final output itself (AttnDecArgs::out1)
bool launch_attn_decode_mfma(const AttnDecArgs& a, int D, int nrep, int nsplit, int kv_mode, float* out, int out_stride, int n_seq, hipStream_t s);
bool launch_attn_decode_heads(const AttnDecArgs& a, int D, int nrep, int ns, bool kv_f32, int n_seq, hipStream_t s);
bool launch_attn_decode(const AttnDecArgs& a, int D, int nrep, int nsplit, int kv_mode, float* out, int out_stride, int n_seq,
                        hipStream_t s);
void launch_gdn(const GdnArgs& a, hipStream_t s);
void launch_bf16_to_f32(const uint16_t* src, float* dst, size_t n, float add, hipStream_t s);

// ---- persistent decode kernel (kernels_engine.hip): the row-streaming projections of MANY layers in one launch ----
enum { ENG_STORE = 0, ENG_RESADD = 1, ENG_SILUMUL = 2 };
constexpr int ENG_NCW = 4;                 // comm waves per workgroup
constexpr int ENG_TRACE_PH = 8;            // phases the debug instantiation records
constexpr int ENG_TRACE_EV = 8;            // event slots per (wave, phase)
// granule buffers: one per kind of vector handed between workgroups inside a launch
enum { ENG_E_X0 = 0,      // [H]   residual after down_proj  -> input of the next layer's QKV
       ENG_E_QKV = 1,     // [qkv rows] merged q | k | v of the token -> attention
       ENG_E_PART = 2,    // [Hkv][nsplit][nrep][D + 2] split-KV partials (o, m, l)
       ENG_E_ATTN = 3,    // [Hq * D] attention output -> input of o_proj
       ENG_E_X1 = 4,      // [H]   residual after o_proj -> input of gate||up
       ENG_E_H = 5,       // [I]   silu(gate) * up -> input of down_proj
       ENG_NEDGE = 6 };
struct EngPhase {                 // one row-streaming projection of the program (table in HBM, read as constant memory)
    const uint16_t* W;            // [N, K] bf16, K contiguous
    const float* nw;              // RMSNorm weight applied to the INPUT vector, or null (plain input)
    int N, K;
    int kind;                     // ENG_STORE | ENG_RESADD | ENG_SILUMUL (rows interleaved gate_j, up_j)
    int gpw;                      // row groups (of 2 rows) per stream wave, rounded up
    int nb;                       // 2048-element chunks of K (= batches per row group)
    int gblk;                     // row groups a wave keeps open at once (chunk-major order inside such a block)
    int xoff, xbuf;               // LDS input buffer: float offset; readiness-counter row (0..3; bit 0 = sum-of-squares row)
    int in_edge, out_edge;        // granule buffers (ENG_E_*), -1: none
    int in_tag, out_tag;          // tags of those edges relative to the launch's epoch base
    int layer;                    // decoder layer (attention operands)
    int useq;                     // how many earlier phases of the table share this counter row
    int pre_attn;                 // != 0: the comm waves run this layer's attention before staging this phase's input; the value = batches per
                                  //   stream wave of the previous (QKV) phase after which every q row is complete
};
struct EngAttnL {                 // attention operands of one layer
    void* kpool;                  // [pages][Hkv][PAGE][D] bf16 or f16 (EngArgs::kv_f16)
    void* vpool;
    const float* qnw;             // [D] f32 or null
    const float* knw;
};
struct EngArgs {
    const EngPhase* prog;         // phase table; this launch runs phases [p0, p1)
    const EngAttnL* attn;         // per layer, or null when no phase of the launch has pre_attn
    unsigned long long* gran[ENG_NEDGE];   // 8-byte {f32 value, u32 tag} granules
    const float* vin;             // input vector of phase p0 (written by an earlier kernel)
    const uint16_t* embed;        // whole token incl. the embedding (round 5): [V, H] bf16 table -- row st->token is the input of phase p0
                                  //   AND the residual stream at entry (vin / the entry value of xres are then unused); null: vin / xres
    float* pmax;                  // head phase (plain_last, kind STORE, the last phase = final norm + lm_head): per stream wave the largest
    int* pidx;                    //   logit it produced and its row (strict > / lowest index), [grid * stream waves]; null: no arg-max
    int embed_V;
    int idx_base;                 // head phase: global index of row 0 (vocabulary shard of a tensor-parallel rank)
                                  // (in-kernel attention: min(32, grid / Hkv) token splits per kv head, workgroups [0, Hkv * splits) run it -- computed
                                  // in the kernel: as an argument it doubled the SGPR spills of the headline instantiation and cost it 17 %)
    float* vout;                  // plain_last: output vector of phase p1 - 1 (read by a later kernel)
    float* xres;                  // [H] residual stream (read at entry, written back at exit)
    uint32_t* ctl;                // [0] epoch base (advanced by every launch), [1] error code (0 = none)
    unsigned long long* trace;    // debug instantiation only: [grid][waves][ENG_TRACE_PH][ENG_TRACE_EV] 100 MHz timestamps
    const StepState* st;          // attention: position of the token
    const int32_t* block_table;
    const float* cos;             // [max_pos, D/2]
    const float* sin;
    int p0, p1, plain_last, epoch_step;
    int ub0, ub1, ub2, ub3;       // useq of the first phase of THIS launch on each counter row
    int H, gpw_res, xf_total;     // hidden size; row groups per wave of the residual phases; LDS floats of the input buffers
    int Hkv, page, max_pages, q_off, k_off, v_off;
    int kv_f16;                   // K/V pages hold IEEE binary16 (CM_KV_F16) instead of bf16
    int nrep;                     // GQA group size of the in-kernel attention: 4 (Qwen3-8B) or 2 (Qwen3-VL-2B text, Qwen3-1.7B)
    int chunk;                    // input elements per dependency chunk / weight batch: 2048, 1024 (Qwen3-0.6B widths) or 512 (tensor-parallel shards)
    float eps, scale;
    int tune;                     // polling parameters (CM_ENG_TUNE while tuning), see kernels_engine.hip
    int dbg;                      // timing experiments (CM_ENG_DBG), see kernels_engine.hip; 0 in production
};
struct EngCfg { int nsw, ncw, pf; };
EngCfg engine_config();           // the instantiation the launcher uses (default, or CM_ENG_CFG while tuning)
size_t engine_lds_bytes(const EngArgs& a, int nsw, int ncw);
bool engine_prepare(size_t lds_bytes, int nrep, int chunk);   // raises the dynamic-LDS limit of the kernel; call once outside any stream capture
bool engine_has_chunk(int chunk);        // dependency chunk size the launcher has an instantiation for (2048; 1024 in the default configuration)
bool engine_has_nrep(int nrep);          // the in-kernel attention is instantiated for this GQA group size
bool launch_engine(const EngArgs& a, int grid, hipStream_t s, bool trace = false);

// ---- synthetic weights / utility ----
// dst[(r * dst_row_stride) + c] = bf16(synth(idx = (row0 + r) * full_cols + col0 + c))
void launch_synth_fill(uint16_t* dst, size_t dst_row_stride, int nrows, int ncols, int row0, int col0,
                       int full_cols, uint32_t tseed, float mul, float off, hipStream_t s);
void launch_kv_fill_quant(void* pool, const int32_t* pages, int npages, size_t page_bytes, size_t code_bytes, uint32_t tseed, hipStream_t s);
void launch_kv_fill(void* pool, int kvt, const int32_t* pages, int npages, size_t page_elems, uint32_t tseed,
                    size_t head_elems, int hkv_all, int kvh0, hipStream_t s);

// ---- quantised weights (kernels_quant.hip) ----
enum { QFMT_NONE = 0, QFMT_Q8_0 = 8, QFMT_Q4_K = 12, QFMT_Q6_K = 14 };    // ggml type ids
struct QWeight {
    int fmt = QFMT_NONE;
    int N = 0, K = 0;
    const uint8_t* p0 = nullptr;   // Q8_0 codes | Q4_K qs | Q6_K ql
    const uint8_t* p1 = nullptr;   // Q8_0 d     | Q4_K hdr | Q6_K qh
    const uint8_t* p2 = nullptr;   //                        Q6_K scales
    const uint8_t* p3 = nullptr;   //                        Q6_K d
    QWeight rows(int row0, int n) const;      // a view of rows [row0, row0 + n)
    uint64_t bytes() const;
};
struct GemvQArgs {
    QWeight w;
    const float* x;        // [K] f32
    const float* nw;       // RMSNorm weight (PRO_RMSNORM)
    float* y;
    const float* res;
    float* pmax;
    int* pidx;
    int idx_base;
    float eps;
    int act_int = 0;       // 1: activations quantised to Q8_0 / Q8_K + integer dot products (ggml vec_dot semantics)
    const float* gdn_z = nullptr;   // PRO_GDNNORM (Q8_0 layout, integer-dot mode): x = RAW y of the Gated-Delta-Net step; the gated RMSNorm of every
    const float* gdn_w = nullptr;   //   128-wide value head (x * rms * gdn_w * silu(gdn_z)) runs in the prologue, in front of the row quantiser
};
int gemvq_grid(int N, int num_cu, int fmt = QFMT_NONE);
// batched integer-dot GEMV: <= 8 sequences per pass over the quantised weights (activations always quantised)
struct GemvQBArgs {
    QWeight w;
    const float* x;        // [n_seq, ldx] f32
    const float* nw;
    float* y;              // [n_seq, ldy]
    const float* res;      // [n_seq, ldy] (EPI_RESADD)
    float* pmax;           // [n_seq, grid] (EPI_ARGMAX)
    int* pidx;
    int n_seq, ldx, ldy, idx_base;
    float eps;
};
// decode groups on the int8 matrix cores (kernels_quant_gemm.hip): QFMT_Q8_0-layout weights x Q8_0-quantised activation rows
constexpr int QGEMM_MAXM = 128;            // rows of a decode group (and the scale stride of its buffers)
constexpr int QGEMM_BIGM = 1 << 16;        // rows of one launch (a prompt pass: m-panels of 256 rows)
struct QGemmArgs {
    QWeight w;
    const signed char* xq;     // [M][K] activation codes (launch_quant_rows_q8)
    const float* xd;           // [K / 32][xs] their block scales, transposed
    int xs = 0;                // floats between the scale rows of consecutive blocks (0: QGEMM_MAXM); also the stride of nxd and of the next projection's scales
    int mpan = 1;              // set by launch_gemm_q8: m-panels the grid walks
    float* ws;                 // set by launch_gemm_q8: partial slices [ksplit][M][N], or the output itself (unsplit store)
    size_t slice;
    int M, ksplit, ldp;
    int silu;                  // set by launch_gemm_q8 (unsplit EPI_SILUMUL): store silu(gate) * up of the interleaved column pairs, row stride ldp
    signed char* nxq;          // ... and, when set, quantise those rows (N / 2 columns) for the next projection: codes [M][N / 2] ...
    float* nxd;                //     ... and block scales [N / 64][QGEMM_MAXM] (NOT the buffers this launch reads)
};
void launch_quant_rows_q8(const float* x, int ldx, const float* nw, float eps, signed char* xq, float* xd, int M, int K, hipStream_t s, int xs = QGEMM_MAXM);
bool gemm_q8_ok(const QWeight& w, int M);
// `next` (EPI_RESADD / EPI_SILUMUL): the rows this GEMM writes are the next projection's input -- the reduction launch also quantises
// them (RMSNorm with next->nw first when set), exactly as launch_quant_rows_q8 would; *fused: 0 = not quantised, 1 = into next->xq / xd
// (by the reduction launch), 2 = into next->xq2 / xd2 (by the unsplit gate|up GEMM itself, which cannot overwrite the codes it reads)
struct QNext { const float* nw; float eps; signed char* xq; float* xd; signed char* xq2; float* xd2; };      // (xq2 / xd2: a second pair, or null)
// EPI_STORE with a consumer that adds the K-split slices itself (the decode attention's prologue): when `defer` is given and the launch
// splits K into 2 .. 4 slices, no reduction is launched -- ks slices of `slice` floats at ws, rows N floats apart; ks = 1: y holds the rows as usual
struct QDefer { int ks; size_t slice; const float* ws; };
// what launch_gemm_q8 will do for a shape (host logic only): ok = false -> the caller's GEMV fallback
struct QGemmPlan { bool ok, direct; int geo, mh, mt, qg, groups, ks, grid; size_t lds; int mpan; };
QGemmPlan plan_gemm_q8(int M, int N, int K, int epi, bool have_ws, size_t ws_floats, int num_cu, int fmt = QFMT_Q8_0);
// Q4_K weights on the same GEMM (round 6): the activation rows as Q8_K blocks in groups of 160 bytes [128 codes | the virtual block: 8
// base-128 digits of the 32-code sums + 24 zeros], 5 f32 scales per group ([K / 128 * 5][xs]); launch_gemm_q8 takes them as xq / xd when
// QGemmArgs::w is a Q4_K tensor (no fused next-quantiser: `next` is ignored)
void launch_quant_rows_q8k(const float* x, int ldx, const float* nw, float eps, signed char* xq, float* xd, int M, int K, hipStream_t s, int xs = QGEMM_MAXM);
bool launch_gemm_q8(const QGemmArgs& a, int epi, float* y, int ldy, float* ws, size_t ws_floats, int num_cu, hipStream_t s,
                    const QNext* next = nullptr, int* fused = nullptr, QDefer* defer = nullptr);
int gemvqb_max_seqs(int fmt, int K);
int gemvqb_grid(int fmt, int N, int K, int n_seq, int num_cu);
bool launch_gemvqb(int pro, int epi, const GemvQBArgs& a, int grid, hipStream_t s);
bool launch_gemvq(int pro, int epi, const GemvQArgs& a, int grid, hipStream_t s);
void launch_embed_row_q(const QWeight& w, const StepState* st, float* x, int H, int V, hipStream_t s, int n_seq = 1);
void launch_dequant_rows(const QWeight& w, float* out, int row0, int nrows, hipStream_t s);
void launch_dequant_bf16(const QWeight& w, uint16_t* out, int row_mul, int row_off, hipStream_t s, uint16_t* out_lo = nullptr);   // out_lo: bf16(v - bf16(v)), same layout
void launch_embed_rows_q(const QWeight& w, const uint32_t* ids, float* x, int S, int H, int V, hipStream_t s);
void launch_isq_q8_0(const uint16_t* src, size_t src_stride, int N, int K, void* codes, void* d, hipStream_t s, int mode = 8);   // mode 4 / 5: Q4_0 / Q5_0 quantiser, Q8_0 layout
void launch_silu_mul(const float* gate, const float* up, float* out, int n, hipStream_t s, int n_seq = 1, int in_stride = 0, int out_stride = 0);

// sampler (kernels_sample.hip)
// one row of a batched sampler call (launch_sample_rows): the row's logits, its private scratch, its parameters
struct SampleRowDev {
    float* logits;
    uint32_t* hist;              // [4096] zeroed histogram
    uint32_t* sel;               // [4] select record
    unsigned long long* cand;
    uint32_t* idx_out;           // top-k ids: the row's idx scratch (sampled) or its token slot (greedy: k = 1)
    float* val;
    uint32_t* tok;
    const uint32_t* pen_ids;
    const uint32_t* pen_counts;
    int pen_n, k, kp, sample, true_div;
    float temperature, top_p, rp, rp_inv, fp, pp;
    uint32_t seed_lo, seed_hi, draw;
};
void launch_sample_rows(const SampleRowDev* tab, int nrows, int V, int max_pen, hipStream_t s);
int topk_pad(int k);
int topk_blocks(int n);
void launch_topk(const float* logits, int n, int k, unsigned long long* cand, uint32_t* hist, uint32_t* sel, uint32_t* idx_out,
                 float* val_out, hipStream_t s);
void launch_penalties(float* logits, const uint32_t* ids, const uint32_t* counts, int n, float rp, bool true_div, float fp, float pp,
                      int V, hipStream_t s);
void launch_sample_topk(const uint32_t* idx, const float* val, int k, float temperature, float top_p, uint64_t seed, uint32_t draw,
                        uint32_t* token_out, hipStream_t s);
void launch_gumbel_full(const float* logits, int V, float temperature, uint64_t seed, uint32_t draw, float* pmax, int* pidx, int blocks,
                        uint32_t* token_out, hipStream_t s);

// peer-store collectives of an in-process tensor-parallel group (kernels_tp.hip)
constexpr int TP_MAX_RANKS = 8;
struct PeerCollArgs {
    unsigned long long* inbox[TP_MAX_RANKS];   // inbox of rank d, peer-visible device memory on d's device: [2 parities][n][cap] granules
    const uint32_t* send;                      // this rank's contribution: f32 bit patterns (sum) / 32-bit words (gather)
    uint32_t* recv;                            // sum: [count] (may alias send); gather: [n][count] (send may alias slot `me`)
    uint32_t* ctl;                             // this rank's control words in device memory: [0] epoch, [1] finish ticket, [2] gave up waiting
    uint32_t* err;                             // host-visible (pinned, mapped) error word of this rank
    long max_spin;                             // bound of the wait for one granule (iterations of ~0.1 us)
    int n, me, count;
    int recv_stride;                           // gather: elements between two ranks' slabs in recv
    size_t cap;                                // granules per (parity, source rank) slot
};
void launch_peer_coll(int mode /*0 sum f32, 1 gather words*/, const PeerCollArgs& a, int blocks, hipStream_t s);

}  // namespace cm
