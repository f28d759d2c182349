// Persistent decode kernel: ONE launch on one workgroup per CU runs the row-streaming projections of a whole token --
// per layer RMSNorm + merged QKV, o_proj (+residual), RMSNorm + gate||up (+SiLU*mul), down_proj (+residual) -- and, between
// QKV and o_proj, the attention of that layer (per-head RMSNorm, RoPE, KV append, split-KV online softmax, merge).
// Reference: Qwen3Model::decode / DecoderLayer::forward / Attention::forward / Mlp::forward,
// crane-core/src/models/qwen3/modeling.rs:307-533, 608-642, 698-716, 984-1036.  The arithmetic per row is the fused
// GEMV's of kernels_decode.hip and the attention's is attn_decode_split_kernel's (kernels_attn_decode.hip).
//
// Why: at batch 1 every projection is a pure weight stream; as separate launches a Qwen3-8B layer costs 81 us for
// 62 us of streaming (six boundaries, each with the ramp and drain of the HBM pipe, and two latency-bound attention
// launches).  Here the weight stream does not stop at a dependency:
//   * STREAM waves own fixed row groups of every phase and keep PF register sets of 16-byte non-temporal weight loads
//     in flight (PF x 8 KiB per wave).  Weight addresses do not depend on activations, so while a wave waits for an
//     input vector the next phases' weights are already landing: the register file is the prefetch ring.  Loads are
//     unconditional and the register sets statically named, so every wait is a counted vmcnt.
//   * Dependencies are per 2048-element CHUNK of the input vector, not per vector: a wave keeps `gblk` row groups
//     open and walks them chunk-major, so the last chunk of an input (the one that needs the slowest producer) is
//     needed only after (nb - 1) / nb of the wave's work on that block.
//   * COMM waves have nothing in their own memory queue, so their polls are not stuck behind a prefetch burst
//     (s_waitcnt vmcnt counts in issue order per wave).  A producer writes each f32 of an output vector as ONE 8-byte
//     {value, tag} granule with a write-through (sc1) agent-scope store; a comm wave sweeps 1024 granules per pass with
//     sc1 loads until every tag equals the epoch of the edge, stages the values into LDS in the GEMV's conflict-free
//     permutation (folding the RMSNorm weight and the sum of squares) and bumps the chunk's LDS counter.  The data is
//     its own flag: no fences, placement-independent (MI355X_MICROARCH.md, Guideline 16 R2).
//   * The comm waves of workgroup (kv head, split) also run that split of the layer's attention -- the block of
//     attn_decode_split_kernel on 4 waves -- exchange partials as granules, merge a 16-value slice each and publish it.
//     Round 6: in two parts along the order the QKV rows finish.  The q rows are the phase's first block of row groups; as soon
//     as they exist the workgroups score the OLD tokens (K / V rows prefetched before the poll), publish and gather partials --
//     under the tail of the QKV stream, which is the k / v rows; the token of the step itself is folded in by the merging wave
//     (q . k_new, one more term of the online softmax), so that only "k / v granules -> merge -> publish -> sweep" follows the
//     end of the QKV stream (Qwen3-8B: 2690 -> 2625 us per launch; polling q a batch earlier, or publishing through the
//     scalar cache -- s_store glc + s_dcache_wb, 0.4 us per idle hop in tools/probes/hop_probe.hip -- both measured slower).
//   * The residual stream row r is owned by the same lane of the same wave in o_proj and down_proj and lives in LDS.
// Every spin is bounded; a timeout raises ctl[1], later launches return at once, and the host reports the code.
#include <cstdio>
#include <cstdlib>

#include "dev_common.h"
#include "kernels.h"

namespace cm {

namespace {

constexpr int R = 2;            // rows per group (one wave reduces R rows at a time)
// U (template parameter UC of the kernel) = 512-element pieces per batch: a batch = R x U 16-byte loads per lane = 8 KiB per wave at
// U = 4 (CHUNK = 2048 input elements, the dependency granule of widths that are multiples of 2048) or 4 KiB at U = 2 (CHUNK =
// 1024: hidden 1024 / intermediate 3072 of Qwen3-0.6B)
constexpr int MAXCH = 20;       // chunks per input vector (K <= 20 * chunk: 17 408 = 17 chunks of 1024, Qwen3.8-27B's intermediate size)
constexpr int MAXGB = 6;        // row groups a wave keeps open
constexpr int MAXRES = 4;       // residual row groups per wave
constexpr int AD = 128;         // attention head_dim
constexpr int ACT = 16;         // attention: tokens per workgroup step (4 waves x 4 rows of 16 lanes)
constexpr uint32_t SPIN_LDS = 400000u;     // x ~0.1 us
constexpr uint32_t SPIN_GLOBAL = 60000u;   // x ~0.5 us

typedef unsigned long long u64;
// every pointer taken from the tables is a global-memory pointer: say so, or the accesses become flat_* (which count on
// lgkmcnt as well and force full waits)
#define CM_GLOBAL __attribute__((address_space(1)))
#define CM_CONST __attribute__((address_space(4)))
typedef const CM_CONST EngPhase* ph_ptr;
typedef const CM_GLOBAL u32x4* gw_ptr;
typedef const CM_GLOBAL float* gf_cptr;
typedef CM_GLOBAL float* gf_ptr;
typedef CM_GLOBAL u64* gu_ptr;

__device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add(uint32_t* p, uint32_t v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ u64 gran_ld(const u64* p) {
    return __hip_atomic_load((gu_ptr)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gran_st(u64* p, uint32_t tag, float v) {
    __hip_atomic_store((gu_ptr)p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a.gran[e] with a run-time e: a chain of selects (indexing the by-value argument struct dynamically would copy it to scratch)
__device__ __forceinline__ u64* gran_sel(const EngArgs& a, int e) {
    return e == 0 ? a.gran[0] : e == 1 ? a.gran[1] : e == 2 ? a.gran[2] : e == 3 ? a.gran[3] : e == 4 ? a.gran[4] : a.gran[5];
}

// float index inside an LDS input buffer: the permutation gemv_bf16_kernel stages x with (two conflict-free
// ds_read_b128 per lane and chunk)
__device__ __forceinline__ int xperm(int k) {
    const int c = k >> 9, j = k & 511;
    return ((c * 128 + ((j >> 2) & 1) * 64 + (j >> 3)) << 2) + (j & 3);
}

// LDS control words (uint32 index)
enum { C_PROG = 0,      // batches finished by this workgroup's stream waves (monotonic over the launch)
       C_ABORT = 1,
       C_CBAR = 2,      // barrier counter of the comm waves
       C_CNT = 4,       // [4 counter rows][MAXCH]: passes staged (monotonic; PPC = 2 or 1 passes per chunk)
       C_WORDS = 4 + 4 * MAXCH + 12 };

}  // namespace

template <int NSW, int NCW, int PF, int NREP, bool TRACE, int UC>
__global__ __launch_bounds__((NSW + NCW) * 64, (NSW + NCW + 3) / 4) void engine_kernel(EngArgs a) {
    // PPC: 1024-granule staging passes per chunk (chunks of >= 1024 elements); CPP: chunks per staging pass (512-element chunks:
    // the shard widths of a tensor-parallel rank -- Hq_l * D = 512, I_l = 1536 at TP = 8 of Qwen3-8B)
    constexpr int U = UC, CHUNK = UC * 512, PPC = CHUNK >= 1024 ? CHUNK / 1024 : 1, CPP = CHUNK >= 1024 ? 1 : 1024 / CHUNK;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // ---- LDS carve (floats) ----
    float* ssq = lds + a.xf_total;                       // [2][16] sum of squares per 1024-element pass of a normed input
    float* accp = ssq + 32;                              // [NSW][MAXGB][R] running row sums of the open row groups
    float* xres_l = accp + NSW * MAXGB * R;              // [NSW][MAXRES][R] residual rows owned by the stream waves
    uint32_t* ctrl = (uint32_t*)(xres_l + NSW * MAXRES * R);
    float* asc = (float*)(ctrl + C_WORDS);               // attention scratch of the comm waves

    auto stamp = [&](int ph, int k, u64 val = ~0ull) __attribute__((always_inline)) {
        if constexpr (TRACE) {
            if (lane == 0 && ph < ENG_TRACE_PH)
                a.trace[(((size_t)blockIdx.x * (NSW + NCW) + wave) * ENG_TRACE_PH + ph) * ENG_TRACE_EV + k] = val == ~0ull ? __builtin_amdgcn_s_memrealtime() : val;
        }
    };
    if (__builtin_nontemporal_load(&a.ctl[1]) != 0u) return;       // an earlier launch timed out: do nothing
    const uint32_t base = __builtin_nontemporal_load(&a.ctl[0]);   // epoch base of this launch
    if (threadIdx.x < C_WORDS) ctrl[threadIdx.x] = 0u;
    __syncthreads();

    const int p0 = a.p0, p1 = a.p1;
    // timing experiments only (CM_ENG_DBG; results are garbage): 1 = never wait for an input, 2 = weight loads hit one cache
    // line (no HBM traffic), 64 = no comm waves, 128 = comm waves skip the staging sweeps, 256 = ... skip the attention
    const bool dbg_nowait = (a.dbg & 1) != 0, dbg_noload = (a.dbg & 2) != 0, dbg_nocomm = (a.dbg & 64) != 0,
               dbg_nostage = (a.dbg & 128) != 0, dbg_noattn = (a.dbg & 256) != 0;
    // tuning word (CM_ENG_TUNE): bit 0 = attention-output staging sweeps without a probe; bits 4-7 = margin (batches of the
    // CU's own stream) before an input is probed; bits 8-12 = s_sleep between probes of a gated input
    const bool t_noprobe_attn = (a.tune & 1) != 0;
    const int t_margin = (a.tune >> 4) & 15, t_psleep = (a.tune >> 8) & 31;
    auto fail = [&](uint32_t code) __attribute__((always_inline)) {
        if (lane == 0) { atomicExch(&a.ctl[1], code); ctrl[C_ABORT] = 1u; }
    };

    if (wave >= NSW) {
        if (dbg_nocomm) return;
        // =====================================================================================================
        // COMM waves: stage the input vector of every phase into LDS (chunk by chunk); run the attention
        // =====================================================================================================
        const int cw = wave - NSW;
        const int tid = cw * 64 + lane;
        uint32_t prog_before = 0;                         // batches all phases before the previous one added to C_PROG
        uint32_t nbar = 0;                                // comm-wave barriers passed
        int prev_nbt = 0;                                 // batches per stream wave of the previous phase
        auto own_progress = [&](uint32_t want, uint32_t code) __attribute__((always_inline)) {
            uint32_t spins = 0;
            while (!dbg_nowait && lds_ld(&ctrl[C_PROG]) < want && lds_ld(&ctrl[C_ABORT]) == 0u) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > SPIN_LDS) { fail(code); break; }
            }
        };
        auto cbar = [&]() __attribute__((always_inline)) {
            ++nbar;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_add(&ctrl[C_CBAR], 1u);
            uint32_t spins = 0;
            while (lds_ld(&ctrl[C_CBAR]) < nbar * NCW && lds_ld(&ctrl[C_ABORT]) == 0u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LDS) { fail(0x500u); break; }
            }
            asm volatile("" ::: "memory");
        };

        // ---- attention: everything that depends only on the token (one token per launch) is fetched ONCE, ahead of the
        // first layer: position, RoPE row, the page-table entries of this workgroup's token chunks.  Between "QKV of the
        // layer has arrived" and "partials published" a comm wave then issues no global load at all -- such a load would
        // queue behind this CU's own weight-prefetch burst (~3-5 us) on the critical path of every layer.
        constexpr int NPRE = 4;                           // K/V chunks requested ahead per layer (contexts <= NPRE * nsplit * ACT)
        int a_pos = 0;
        float a_cos = 0.f, a_sin = 0.f;
        size_t a_koff[NPRE], a_eoff = 0;
        int a_tt[NPRE];
        bool a_owner = false;
        int a_kvh = 0, a_split = 0, a_nsplit = 1;        // this workgroup's (kv head, token split) and the splits per kv head, once per launch
        bool a_attwg = false;                             //   (integer divisions: not on the per-layer critical path)
        int append_layer = -1;                            // owner of the step's token: layer whose rounded k / v rows (LDS) still go to the pages
#pragma unroll
        for (int u = 0; u < NPRE; ++u) { a_koff[u] = 0; a_tt[u] = 0; }
        if (a.attn != nullptr) {
            // (at most 32 token splits per kv head; workgroups past Hkv * nsplit -- one kv head per rank under tensor parallelism
            // leaves 224 of 256 -- take no part in the attention: their loads below stay in bounds and are never consumed)
            const int Hkv = a.Hkv, kvh = blockIdx.x % Hkv, nsplit = min(32, (int)gridDim.x / Hkv), split = (blockIdx.x / Hkv) % nsplit;
            a_kvh = kvh; a_split = split; a_nsplit = nsplit; a_attwg = (int)blockIdx.x < Hkv * nsplit;
            const int tok_in_chunk = cw * 4 + (lane >> 4), dimbase = (lane & 15) * 8;
            const CM_GLOBAL int32_t* bt = (const CM_GLOBAL int32_t*)a.block_table;
            a_pos = ((const CM_GLOBAL StepState*)a.st)->pos;
            const int rpos = a_pos + ((const CM_GLOBAL StepState*)a.st)->rsv[0];
            a_cos = ((gf_cptr)a.cos)[(size_t)rpos * (AD >> 1) + lane];
            a_sin = ((gf_cptr)a.sin)[(size_t)rpos * (AD >> 1) + lane];
            a_owner = ((a_pos / ACT) % nsplit) == split;
            int bte[NPRE];
#pragma unroll
            for (int u = 0; u < NPRE; ++u) {
                // a chunk past the context reads chunk 0's rows again (cache hits, never consumed)
                const bool cv = ACT * (split + nsplit * u) < a_pos;         // (old tokens only: the step's own token never comes from the pages)
                a_tt[u] = ACT * (split + nsplit * (cv ? u : 0)) + tok_in_chunk;
                int pi = a_tt[u] / a.page;
                pi = pi < a.max_pages ? pi : a.max_pages - 1;         // speculative loads stay inside the table
                bte[u] = bt[pi];
            }
            const int bto = bt[a_pos / a.page];
#pragma unroll
            for (int u = 0; u < NPRE; ++u) a_koff[u] = ((size_t)(bte[u] * Hkv + kvh) * a.page + (a_tt[u] % a.page)) * AD + dimbase;
            a_eoff = ((size_t)(bto * Hkv + kvh) * a.page + (a_pos % a.page)) * AD;
        }

        for (int p = p0; p < p1; ++p) {
            ph_ptr P = (ph_ptr)a.prog + p;
            const int K = P->K, xbuf = P->xbuf, nbt_p = P->gpw * P->nb;
            if (p == p0) {                                // staged by the stream waves from a.vin
                prev_nbt = nbt_p;
                continue;
            }
            stamp(p - p0, 0);
            if (append_layer >= 0) {
                // KV append of the previous phase's attention, off its critical path: a vector store queues behind this CU's weight
                // loads, and every later poll of the same wave (s_waitcnt vmcnt) would wait for it -- the owner workgroup published its
                // slice ~2 us after the others when the store sat in front of the gather.  The rounded rows are still in LDS.
                if (cw == NCW - 1) {
                    const CM_CONST EngAttnL* AL = (const CM_CONST EngAttnL*)a.attn + append_layer;
                    CM_GLOBAL uint16_t* kpl = (CM_GLOBAL uint16_t*)AL->kpool;
                    CM_GLOBAL uint16_t* vpl = (CM_GLOBAL uint16_t*)AL->vpool;
                    const float* knew = asc + NREP * AD;
                    const float* vnew = knew + AD;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float kx = knew[lane + 64 * j], vx = vnew[lane + 64 * j];
                        kpl[a_eoff + lane + 64 * j] = a.kv_f16 ? f32_to_f16(kx) : f32_to_bf16(kx);
                        vpl[a_eoff + lane + 64 * j] = a.kv_f16 ? f32_to_f16(vx) : f32_to_bf16(vx);
                    }
                }
                append_layer = -1;
            }
            if (P->pre_attn && !dbg_noattn && a_attwg) {
                // ================= attention of this layer (split `blockIdx / Hkv` of kv head `blockIdx % Hkv`) =================
                const CM_CONST EngAttnL* AL = (const CM_CONST EngAttnL*)a.attn + P->layer;
                const int Hkv = a.Hkv, kvh = a_kvh, split = a_split, nsplit = a_nsplit;
                const int r = lane >> 4, sub = lane & 15, dimbase = sub * 8, tok_in_chunk = cw * 4 + r;
                const uint32_t tag = base + (uint32_t)P->in_tag;     // QKV, partials and merged output of this layer share it
                float* qs = asc;                          // [NREP][AD]
                float* knew = qs + NREP * AD;             // [AD]
                float* vnew = knew + AD;                  // [AD]
                float* red_m = vnew + AD;                 // [ACT][NREP]
                float* red_l = red_m + ACT * NREP;        // [ACT][NREP]
                float* red_o = red_l + ACT * NREP;        // [ACT][NREP][AD]
                float* mo = red_o + ACT * NREP * AD;      // [nsplit <= 32][18] gathered partial slices
                const CM_GLOBAL uint16_t* kp = (const CM_GLOBAL uint16_t*)AL->kpool;
                const CM_GLOBAL uint16_t* vp = (const CM_GLOBAL uint16_t*)AL->vpool;
                const CM_GLOBAL int32_t* bt = (const CM_GLOBAL int32_t*)a.block_table;
                auto kv_off = [&](int t) -> size_t {
                    int pi = t / a.page;
                    pi = pi < a.max_pages ? pi : a.max_pages - 1;     // speculative loads stay inside the table
                    return ((size_t)(bt[pi] * Hkv + kvh) * a.page + (t % a.page)) * AD + dimbase;
                };
                // K/V rows of this block's first NPRE chunks and the QK-norm weights: requested before anything else
                // (independent of this layer's QKV)
                u32x4 kq[NPRE], vq[NPRE];
#pragma unroll
                for (int u = 0; u < NPRE; ++u) {
                    kq[u] = *(gw_ptr)(kp + a_koff[u]);
                    vq[u] = *(gw_ptr)(vp + a_koff[u]);
                }
                // Items of the layer's QKV vector: the NREP q heads of the group (wave w < NREP takes head w), the new k and the new v
                // (waves NREP % NCW and (NREP + 1) % NCW).  Round 6: the attention is split in two along the order the rows of the QKV
                // phase finish -- q first (P->pre_attn = batches per stream wave after which every q row is complete), k / v last:
                //   A (under the tail of the QKV stream): q -> norm / RoPE -> scores and weighted values of the OLD tokens t < pos
                //     (K / V rows prefetched above) -> partials published -> this workgroup's output slice gathered from every split;
                //   B (after the k / v rows): k -> norm / RoPE, KV append by the owner, and the merging wave folds the NEW token into
                //     its slice itself (score = q . k_new, one more term of the online softmax) -> publish.
                // Between "QKV streamed" and "o_proj may start" there are then two dependent cross-CU hops (k / v granules, merged
                // output) instead of four (QKV, partials, merged output + the score pass between them).
                static_assert(NREP <= NCW && NCW >= 2, "one q item and at most one of k / v per comm wave");
                const bool has_q = cw < NREP;
                const int kvj = (cw - NREP + 2 * NCW) % NCW;         // 0: this wave takes k, 1: v, else neither
                float nq[2], nk[2];
                {
                    gf_cptr nwq = (gf_cptr)AL->qnw, nwk = (gf_cptr)AL->knw;
                    const bool hq = has_q && nwq != nullptr, hk = kvj == 0 && nwk != nullptr;
                    if (!hq) nwq = (gf_cptr)a.cos;                        // any readable address: the loads stay unconditional
                    if (!hk) nwk = (gf_cptr)a.cos;
                    nq[0] = nwq[lane]; nq[1] = nwq[lane + 64];
                    nk[0] = nwk[lane]; nk[1] = nwk[lane + 64];
                    if (!hq) { nq[0] = 1.f; nq[1] = 1.f; }
                    if (!hk) { nk[0] = 1.f; nk[1] = 1.f; }
                }
                const int pos = a_pos;
                const bool owner = a_owner;
                const u64* GQ = a.gran[ENG_E_QKV];
                // one 128-value item per wave and part, ALL its granules requested in one round trip per poll (a wave without an item
                // polls the group's first q head / the k head: result unused)
                const int gA = a.q_off + (kvh * NREP + (has_q ? cw : 0)) * AD;
                const int gB = kvj == 1 ? a.v_off + kvh * AD : a.k_off + kvh * AD;
                uint32_t tr_spins = 0;
                auto poll_item = [&](int g, float (&xv)[2], uint32_t code) __attribute__((always_inline)) -> uint32_t {
                    uint32_t spins = 0;
                    for (;;) {
                        const u64 x0 = gran_ld(GQ + g + lane), x1 = gran_ld(GQ + g + lane + 64);
                        xv[0] = __uint_as_float((uint32_t)x0); xv[1] = __uint_as_float((uint32_t)x1);
                        const bool ok = (uint32_t)(x0 >> 32) == tag && (uint32_t)(x1 >> 32) == tag;
                        if (__all(ok) || dbg_nowait) break;
                        if (lds_ld(&ctrl[C_ABORT]) != 0u) break;
                        if (++spins > SPIN_GLOBAL) { fail(code); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    return spins;
                };
                // per-head RMSNorm (optional) + rotate-half RoPE over the whole head: the partner of d is d +/- D/2 = the other element of this lane
                auto norm_rope = [&](float (&xv)[2], const float (&nw)[2], bool normed) __attribute__((always_inline)) {
                    if (normed) {
                        const float ss = wave_sum(xv[0] * xv[0] + xv[1] * xv[1]);
                        const float rr = 1.0f / sqrtf(ss / (float)AD + a.eps);
                        xv[0] = xv[0] * rr * nw[0]; xv[1] = xv[1] * rr * nw[1];
                    }
                    const float lo = xv[0], hi = xv[1];
                    xv[0] = lo * a_cos - hi * a_sin;
                    xv[1] = lo * a_sin + hi * a_cos;
                };
                // ================= part A: q and the old tokens =================
                // this workgroup's own stream waves must be through the q rows of the QKV phase before its comm waves poll them
                own_progress(prog_before + (uint32_t)(NSW * min(P->pre_attn, prev_nbt)) - (uint32_t)(NSW / 2), 0x600u + (uint32_t)(p - p0));   // (nearly: the polls are tiny)
                stamp(p - p0, 1);
                {
                    float xv[2];
                    tr_spins = poll_item(gA, xv, 0x700u + (uint32_t)(p - p0));
                    if (has_q) {
                        norm_rope(xv, nq, AL->qnw != nullptr);
                        qs[cw * AD + lane] = xv[0] * a.scale;
                        qs[cw * AD + lane + 64] = xv[1] * a.scale;
                    }
                }
                cbar();
                float qr[NREP][8];
#pragma unroll
                for (int h = 0; h < NREP; ++h) {
                    const f32x4 q0 = *(const f32x4*)&qs[h * AD + dimbase];
                    const f32x4 q1 = *(const f32x4*)&qs[h * AD + dimbase + 4];
                    qr[h][0] = q0[0]; qr[h][1] = q0[1]; qr[h][2] = q0[2]; qr[h][3] = q0[3];
                    qr[h][4] = q1[0]; qr[h][5] = q1[1]; qr[h][6] = q1[2]; qr[h][7] = q1[3];
                }
                float m[NREP], l[NREP], acc[NREP][8];
#pragma unroll
                for (int h = 0; h < NREP; ++h) {
                    m[h] = -INFINITY; l[h] = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[h][e] = 0.f;
                }
                auto consume = [&](const u32x4& kqv, const u32x4& vqv, int t) __attribute__((always_inline)) {
                    const bool valid = t < pos;               // (the token of this step is folded in by the merge, part B)
                    float kf[8], vf[8];
                    if (a.kv_f16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            kf[2 * e] = f16_lo(kqv[e]); kf[2 * e + 1] = f16_hi(kqv[e]);
                            vf[2 * e] = f16_lo(vqv[e]); vf[2 * e + 1] = f16_hi(vqv[e]);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            kf[2 * e] = bf16_lo(kqv[e]); kf[2 * e + 1] = bf16_hi(kqv[e]);
                            vf[2 * e] = bf16_lo(vqv[e]); vf[2 * e + 1] = bf16_hi(vqv[e]);
                        }
                    }
#pragma unroll
                    for (int h = 0; h < NREP; ++h) {
                        float s = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) s += qr[h][e] * kf[e];
                        s = row16_sum(s);
                        if (valid) {
                            const float mn = fmaxf(m[h], s);
                            const float alpha = expf(m[h] - mn);
                            const float pw = expf(s - mn);
                            l[h] = l[h] * alpha + pw;
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[h][e] = acc[h][e] * alpha + pw * vf[e];
                            m[h] = mn;
                        }
                    }
                };
                stamp(p - p0, 4);
#pragma unroll
                for (int u = 0; u < NPRE; ++u)
                    if (ACT * (split + nsplit * u) < pos) consume(kq[u], vq[u], a_tt[u]);
                for (int j = NPRE; ACT * (split + nsplit * j) < pos; j += 2) {    // longer contexts: two more chunks per iteration
                    const bool second = ACT * (split + nsplit * (j + 1)) < pos;
                    int tt[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        tt[u] = ACT * (split + nsplit * (j + u)) + tok_in_chunk;
                        const size_t off = kv_off(tt[u]);          // clamped: always a valid address
                        kq[u] = *(gw_ptr)(kp + off);
                        vq[u] = *(gw_ptr)(vp + off);
                    }
                    consume(kq[0], vq[0], tt[0]);
                    if (second) consume(kq[1], vq[1], tt[1]);
                }
                stamp(p - p0, 5);
                // ---- combine the ACT (wave, row) streams of this workgroup -> partial (m, l, o) per head -> granules ----
                const int slot = cw * 4 + r;
#pragma unroll
                for (int h = 0; h < NREP; ++h) {
                    if (sub == 0) { red_m[slot * NREP + h] = m[h]; red_l[slot * NREP + h] = l[h]; }
                    *(f32x4*)&red_o[(slot * NREP + h) * AD + dimbase] = (f32x4){acc[h][0], acc[h][1], acc[h][2], acc[h][3]};
                    *(f32x4*)&red_o[(slot * NREP + h) * AD + dimbase + 4] = (f32x4){acc[h][4], acc[h][5], acc[h][6], acc[h][7]};
                }
                cbar();
                u64* GP = a.gran[ENG_E_PART];
                for (int it = tid; it < NREP * AD; it += NCW * 64) {
                    const int h = it / AD, d = it % AD;
                    float M = -INFINITY;
#pragma unroll
                    for (int i = 0; i < ACT; ++i) M = fmaxf(M, red_m[i * NREP + h]);
                    float O = 0.f, Ls = 0.f;
                    if (M > -INFINITY) {
#pragma unroll
                        for (int i = 0; i < ACT; ++i) {
                            const float w = expf(red_m[i * NREP + h] - M);
                            O += w * red_o[(i * NREP + h) * AD + d];
                            Ls += w * red_l[i * NREP + h];
                        }
                    }
                    const size_t pb = ((size_t)(kvh * nsplit + split) * NREP + h) * (AD + 2);
                    gran_st(GP + pb + d, tag, O);
                    if (d == 0) { gran_st(GP + pb + AD, tag, M); gran_st(GP + pb + AD + 1, tag, Ls); }
                }
                // ================= part B: the token of this step (its k / v rows have usually arrived by now; polled BEFORE the gather so that
                // it runs in the shadow of the other splits' partials) =================
                own_progress(prog_before + (uint32_t)(NSW * prev_nbt) - (uint32_t)(NSW / 2), 0x680u + (uint32_t)(p - p0));
                {
                    float xv[2];
                    (void)poll_item(gB, xv, 0x780u + (uint32_t)(p - p0));
                    if (kvj < 2) {
                        if (kvj == 0) norm_rope(xv, nk, AL->knw != nullptr);
                        float* dst = kvj == 0 ? knew : vnew;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {       // (kv_f16 is launch-uniform: a scalar branch)
                            const uint16_t b16 = a.kv_f16 ? f32_to_f16(xv[j]) : f32_to_bf16(xv[j]);
                            dst[lane + 64 * j] = a.kv_f16 ? f16_to_f32(b16) : bf16_to_f32(b16);
                        }
                    }
                }
                // ---- this workgroup owns OPB consecutive outputs of the group's NREP * D; gather them from every split ----
                const int OPB = NREP * AD / nsplit;                    // 16 on Qwen3-8B
                const int o0 = split * OPB, hm = o0 / AD, d0 = o0 % AD;
                {
                    const int s_src = tid >> 3, e = tid & 7;           // 8 threads per source split (nsplit <= 32)
                    const size_t pb = ((size_t)(kvh * nsplit + (s_src < nsplit ? s_src : 0)) * NREP + hm) * (AD + 2);
                    const bool act = s_src < nsplit && 2 * e < OPB;
                    const bool act_ml = s_src < nsplit && e < 2;
                    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
                    uint32_t spins = 0;
                    for (;;) {
                        const u64 x0 = gran_ld(GP + pb + d0 + (act ? 2 * e : 0));
                        const u64 x1 = gran_ld(GP + pb + d0 + (act ? 2 * e + 1 : 0));
                        const u64 x2 = gran_ld(GP + pb + AD + (act_ml ? e : 0));
                        v0 = __uint_as_float((uint32_t)x0); v1 = __uint_as_float((uint32_t)x1); v2 = __uint_as_float((uint32_t)x2);
                        const bool ok = (!act || ((uint32_t)(x0 >> 32) == tag && (uint32_t)(x1 >> 32) == tag)) &&
                                        (!act_ml || (uint32_t)(x2 >> 32) == tag);
                        if (__all(ok) || dbg_nowait) break;
                        if (lds_ld(&ctrl[C_ABORT]) != 0u) break;
                        if (++spins > SPIN_GLOBAL) { fail(0x800u + (uint32_t)(p - p0)); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (act) { mo[s_src * 18 + 2 * e] = v0; mo[s_src * 18 + 2 * e + 1] = v1; }
                    if (act_ml) mo[s_src * 18 + 16 + e] = v2;
                    tr_spins |= spins << 16;
                }
                stamp(p - p0, 6);
                stamp(p - p0, 7, (u64)tr_spins);
                cbar();
                if (cw == 0) {
                    // one wave on the critical path of the layer (it was 1.8 - 2.3 us: 32 + 24 LDS reads and 9 exps per lane): lane s < nsplit
                    // owns split s -- its weight is ONE exp, the maximum and the denominator are wave reductions --, then lane (output d,
                    // group of 8 splits) gathers the weights by lane permutes and its 8 values from LDS
                    const int d = lane & 15, sg = lane >> 4;
                    const float s_new = wave_sum(qs[hm * AD + lane] * knew[lane] + qs[hm * AD + lane + 64] * knew[lane + 64]);
                    const int sl = lane < nsplit ? lane : 0;
                    const float ms = lane < nsplit ? mo[sl * 18 + 16] : -INFINITY;
                    const float ls = mo[sl * 18 + 17];
                    const float M = fmaxf(wave_max(ms), s_new);
                    const float ws = ms > -INFINITY ? expf(ms - M) : 0.f;
                    const float w_new = expf(s_new - M);
                    const float Ls = wave_sum(ws * ls) + w_new;
                    float O = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int s2 = sg * 8 + i;
                        O += __shfl(ws, s2) * mo[(s2 < nsplit ? s2 : 0) * 18 + d];       // (ws = 0 past the last split)
                    }
                    if (sg == 0) O += w_new * vnew[(d0 + d) & (AD - 1)];
                    O += __shfl_xor(O, 16);
                    O += __shfl_xor(O, 32);
                    if (lane < OPB) gran_st(a.gran[ENG_E_ATTN] + (size_t)(kvh * NREP + hm) * AD + d0 + lane, tag, O * (1.0f / Ls));
                }
                stamp(p - p0, 2);          // (`mo`, qs, red_* are next written a whole layer later)
                if (owner) append_layer = P->layer;
            } else {
                // (a workgroup outside the attention -- or a launch without it -- still waits for its own stream waves to be through
                // the producing phase before it polls the attention output: 224 workgroups polling through QKV + attention would
                // travel the fabric for nothing)
                if (P->pre_attn && a.attn != nullptr && !dbg_noattn)
                    own_progress(prog_before + (uint32_t)(NSW * prev_nbt) - (uint32_t)(NSW / 2), 0x600u + (uint32_t)(p - p0));
                stamp(p - p0, 1);
            }
            // ---- stage this phase's input: 1024 granules per pass, passes cw, cw + NCW, ...; chunk c = passes PPC * c ... ----
            {
                gf_cptr nw = (gf_cptr)P->nw;
                float* xs = lds + P->xoff;
                const u64* G = gran_sel(a, P->in_edge);
                const uint32_t tag = base + (uint32_t)P->in_tag;
                uint32_t total_spins = 0;
                for (int pass = cw; pass * 1024 < K && !dbg_nostage; pass += NCW) {
                    const int kb = pass * 1024 + lane;
                    float v[16], wv[16];
                    if (!P->pre_attn) {
                        // Polling costs bandwidth next to the weight stream, so a pass is swept only when the chip has
                        // probably finished it: every CU runs the same schedule at nearly the same pace, so "my own stream
                        // waves closed the row groups that produce this pass, plus a margin" is a good predictor.  The
                        // tags stay the truth: a sweep that finds stale granules simply repeats.
                        ph_ptr Q = P - 1;                                  // the producing phase
                        const int per_round = gridDim.x * NSW * (Q->kind == ENG_SILUMUL ? 1 : R);      // outputs one round of row groups covers
                        int gi_last = ((pass + 1) * 1024 + per_round - 1) / per_round - 1;             // last round that writes into this pass
                        gi_last = gi_last < Q->gpw ? gi_last : Q->gpw - 1;
                        const int blk = gi_last / Q->gblk, left = Q->gpw - blk * Q->gblk, cnt = left < Q->gblk ? left : Q->gblk;
                        int b_done = blk * Q->gblk * Q->nb + (Q->nb - 1) * cnt + (gi_last - blk * Q->gblk) + 1 + t_margin;   // + margin
                        b_done = b_done < prev_nbt ? b_done : prev_nbt;
                        own_progress(prog_before + (uint32_t)(NSW * b_done), 0x100u + (uint32_t)(p - p0));
                    }
                    // a pass past the end of a vector whose length is a multiple of 512 only (K = 512, 1536): the upper half is padding
                    // -- never tagged (excluded from `ok`), staged as zeros; buffers are sized in whole passes
                    // (only in the 512-element-chunk instantiation: the selects below cost the 2048-chunk kernel of the headline 17 %
                    // -- more live scalars in the comm path, twice the SGPR spills, a slower stream loop -- when they were unconditional)
                    const bool tail = UC == 1 && (pass + 1) * 1024 > K;
                    if (nw != nullptr) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) wv[i] = nw[(tail && i >= 8) ? lane : kb + i * 64];
                    }
                    uint32_t spins = 0;
                    // A sweep of the pass costs 64 cache-line requests per CU whether it finds the pass complete or not, and
                    // the sweeps of all CUs travel the same fabric as the weight stream.  So the pass is first PROBED: 8 lines
                    // spread over it (the granules of 64 producer waves on 16 CUs), an eighth of a sweep; only a clean probe
                    // is followed by the sweep.
                    if (!(P->pre_attn && t_noprobe_attn)) {
                        const u64* GP0 = G + pass * 1024 + (tail ? (lane >> 3) * 64 : (lane >> 3) * 128) + (lane & 7) * 2 + 1;
                        for (;;) {
                            const u64 x = gran_ld(GP0);
                            if (__all((uint32_t)(x >> 32) == tag) || dbg_nowait) break;
                            if (lds_ld(&ctrl[C_ABORT]) != 0u) break;
                            if (++spins > SPIN_GLOBAL) { fail(0x200u + (uint32_t)(p - p0)); break; }
                            if (P->pre_attn) __builtin_amdgcn_s_sleep(1); else for (int z = 0; z < t_psleep; ++z) __builtin_amdgcn_s_sleep(1);
                        }
                        spins = 0;
                    }
                    for (;;) {
                        bool ok = true;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const u64 x = gran_ld(G + kb + i * 64);
                            v[i] = __uint_as_float((uint32_t)x);
                            ok = ok && ((uint32_t)(x >> 32) == tag || (tail && i >= 8));
                        }
                        if (__all(ok) || dbg_nowait) break;
                        if (lds_ld(&ctrl[C_ABORT]) != 0u) break;
                        if (++spins > SPIN_GLOBAL) { fail(0x300u + (uint32_t)(p - p0)); break; }
                        if (P->pre_attn) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(12);
                    }
                    total_spins += spins;
                    float ss = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float val = (tail && i >= 8) ? 0.f : v[i];
                        if (nw != nullptr) { ss += val * val; val *= wv[i]; }
                        xs[xperm(kb + i * 64)] = val;
                    }
                    if (nw != nullptr) {
                        ss = wave_sum(ss);
                        if (lane == 0) ssq[(xbuf & 1) * 16 + pass] = ss;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) {
#pragma unroll
                        for (int c2 = 0; c2 < CPP; ++c2) lds_add(&ctrl[C_CNT + xbuf * MAXCH + (CPP > 1 ? pass * CPP + c2 : pass / PPC)], 1u);
                    }
                }
                stamp(p - p0, 3, (u64)total_spins);
            }
            prog_before += (uint32_t)(NSW * prev_nbt);
            prev_nbt = nbt_p;
        }
        return;
    }

    // =========================================================================================================
    // STREAM waves: rows x weights, never waiting on HBM
    // =========================================================================================================
    const int gwid = blockIdx.x * NSW + wave, TW = gridDim.x * NSW;
    stamp(0, 3);

    // residual rows owned by this wave (same mapping in o_proj and down_proj): lane i < R of group gi.  Requested first and
    // written to LDS only after the weight prefetch has been issued, so the wait is a counted one.
    // (a.embed: the token's embedding row IS the residual stream and the first phase's input -- no embedding launch in front)
    const CM_GLOBAL uint16_t* erow = nullptr;
    if (a.embed != nullptr) {
        uint32_t tok = ((const CM_GLOBAL StepState*)a.st)->token;
        if (tok >= (uint32_t)a.embed_V) tok = 0;          // (host validates ids; the device stays in bounds regardless)
        erow = (const CM_GLOBAL uint16_t*)a.embed + (size_t)tok * (size_t)a.H;
    }
    float xv0[MAXRES];
#pragma unroll
    for (int gi = 0; gi < MAXRES; ++gi) {
        int row = (gwid + gi * TW) * R + (lane < R ? lane : 0);
        row = row < a.H ? row : a.H - 1;
        xv0[gi] = erow != nullptr ? bf16_to_f32(erow[row]) : a.xres[row];
    }
    // The first phase's input vector was written by an earlier kernel: stream wave w requests pass w of it (1024 floats)
    // BEFORE the weight prefetch (a load issued behind this CU's prefetch burst returns ~5 us later) and stages it once
    // the prefetch is on its way.  Waves past the last pass load a clamped pass and write nothing.
    ph_ptr PF0 = (ph_ptr)a.prog + p0;
    const int npass0 = PF0->K >> 10;          // <= 2 * NSW (the host checks): wave w takes passes w and w + NSW
    f32x4 xin[2][4], win[2][4];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int pw = wave + pp * NSW;
        const int pass = pw < npass0 ? pw : npass0 - 1;
        const CM_GLOBAL f32x4* w4 = (const CM_GLOBAL f32x4*)(PF0->nw != nullptr ? PF0->nw : (erow != nullptr ? (const float*)a.cos : a.vin)) + pass * 256;
        if (erow != nullptr) {           // (launch-uniform branch; 4 bf16 per lane and load)
            const CM_GLOBAL u32x2* e2 = (const CM_GLOBAL u32x2*)erow + pass * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x2 pk = e2[i * 64 + lane];
                xin[pp][i] = (f32x4){bf16_lo(pk[0]), bf16_hi(pk[0]), bf16_lo(pk[1]), bf16_hi(pk[1])};
                win[pp][i] = w4[i * 64 + lane];
            }
        } else {
            const CM_GLOBAL f32x4* v4 = (const CM_GLOBAL f32x4*)a.vin + pass * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i) { xin[pp][i] = v4[i * 64 + lane]; win[pp][i] = w4[i * 64 + lane]; }
        }
    }

    // ---- cursors over the batch sequence: phase -> block of `gblk` row groups -> chunk kb -> group gg of the block ----
    struct Cur {
        ph_ptr P;
        int ph, gb, kb, gg;
        int N, K, gpw, nb, gblk, cnt;       // cnt = groups in the current block
    };
    auto cur_open = [&](Cur& c) __attribute__((always_inline)) {
        c.N = c.P->N; c.K = c.P->K; c.gpw = c.P->gpw; c.nb = c.P->nb; c.gblk = c.P->gblk;
        c.gb = 0; c.kb = 0; c.gg = 0;
        c.cnt = c.gpw < c.gblk ? c.gpw : c.gblk;
    };
    // advance to the next batch; returns true when the phase is finished
    auto cur_next = [&](Cur& c) __attribute__((always_inline)) -> bool {
        if (++c.gg < c.cnt) return false;
        c.gg = 0;
        if (++c.kb < c.nb) return false;
        c.kb = 0;
        ++c.gb;
        const int left = c.gpw - c.gb * c.gblk;
        if (left > 0) { c.cnt = left < c.gblk ? left : c.gblk; return false; }
        return true;
    };

    Cur lc;                                   // load side: runs PF batches ahead
    lc.P = (ph_ptr)a.prog + p0; lc.ph = p0;
    cur_open(lc);
    const CM_GLOBAL uint16_t* lW = (const CM_GLOBAL uint16_t*)lc.P->W;
    bool lvalid = true;
    auto load_batch = [&](u32x4 (&q)[R][U]) __attribute__((always_inline)) {
        const int g = gwid + (lc.gb * lc.gblk + lc.gg) * TW;
        // a row group past the last one, or a batch past the end of the program: every lane reads the same 16 bytes of the
        // matrix (no HBM traffic, result never used) -- the loads stay unconditional
        const bool real = lvalid && g < lc.N / R && !dbg_noload;
        const size_t roff = real ? (size_t)g * R * (size_t)lc.K + (size_t)lc.kb * CHUNK : 0;
        const int loff = real ? lane * 8 : 0;
        const size_t sK = real ? (size_t)lc.K : 0;
        const int sU = real ? 512 : 0;
        const CM_GLOBAL uint16_t* wp = lW + roff + loff;
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int u = 0; u < U; ++u) q[i][u] = __builtin_nontemporal_load((gw_ptr)(wp + i * sK + u * sU));
        if (lvalid && cur_next(lc)) {
            if (lc.ph + 1 < p1) { ++lc.ph; ++lc.P; cur_open(lc); lW = (const CM_GLOBAL uint16_t*)lc.P->W; }
            else lvalid = false;
        }
    };

    Cur cc;                                   // compute side
    cc.P = (ph_ptr)a.prog + p0; cc.ph = p0;
    cur_open(cc);
    int nready = 0;                           // chunks of the current phase's input known to be staged
    bool fresh = true;                        // first batch of a phase
    const f32x4* xs4 = (const f32x4*)lds;
    int ckind = 0, cout = -1, cxbuf = 0;
    uint32_t cneed = 0, ctag = 0;
    bool cnorm = false, cplain = false;
    gf_ptr cvout = (gf_ptr)a.vout;
    float* my_accp = accp + wave * MAXGB * R;
    float* my_xres = xres_l + wave * MAXRES * R;
    float best_v = -INFINITY;                 // head phase: the largest logit this lane (row parity) has produced, and its row
    int best_i = 0x7FFFFFFF;

    auto compute = [&](u32x4 (&q)[R][U]) __attribute__((always_inline)) {
        if (fresh) {
            fresh = false;
            ckind = cc.P->kind; cout = cc.P->out_edge; cxbuf = cc.P->xbuf; cnorm = cc.P->nw != nullptr;
            ctag = base + (uint32_t)cc.P->out_tag;
            cplain = a.plain_last != 0 && cc.ph == p1 - 1;
            xs4 = (const f32x4*)(lds + cc.P->xoff);
            // PPC staged passes per chunk and per phase of this launch that has used the counter row (this one included)
            const int ub = cxbuf == 0 ? a.ub0 : cxbuf == 1 ? a.ub1 : cxbuf == 2 ? a.ub2 : a.ub3;
            cneed = (uint32_t)PPC * (uint32_t)(cc.P->useq - ub + 1);
            nready = 0;
            stamp(cc.ph - p0, 0);
        }
        if (cc.kb >= nready) {                // first touch of this chunk: wait until its passes are staged
            uint32_t spins = 0;
            const uint32_t* w = &ctrl[C_CNT + cxbuf * MAXCH + cc.kb];
            while (!dbg_nowait && lds_ld(w) < cneed && lds_ld(&ctrl[C_ABORT]) == 0u) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_LDS) { fail(0x400u + (uint32_t)(cc.ph - p0)); break; }
            }
            asm volatile("" ::: "memory");
            nready = cc.kb + 1;
            if (cc.kb == 0) stamp(cc.ph - p0, 1);
        }
        float acc[R];
#pragma unroll
        for (int i = 0; i < R; ++i) acc[i] = 0.f;
        const int cb = cc.kb * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4 xa = xs4[(cb + u) * 128 + lane];
            const f32x4 xb = xs4[(cb + u) * 128 + 64 + lane];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                acc[i] += bf16_lo(q[i][u][0]) * xa[0] + bf16_hi(q[i][u][0]) * xa[1] +
                          bf16_lo(q[i][u][1]) * xa[2] + bf16_hi(q[i][u][1]) * xa[3] +
                          bf16_lo(q[i][u][2]) * xb[0] + bf16_hi(q[i][u][2]) * xb[1] +
                          bf16_lo(q[i][u][3]) * xb[2] + bf16_hi(q[i][u][3]) * xb[3];
            }
        }
        // per-batch reduction; lane i < R carries row i's running sum over the chunks in LDS between visits of the group
        const float s0 = wave_sum(acc[0]), s1 = wave_sum(acc[1]);
        float run = lane == 1 ? s1 : s0;
        float* park = my_accp + cc.gg * R;
        if (cc.kb > 0 && lane < R) run += park[lane];
        const bool last_chunk = cc.kb == cc.nb - 1;
        if (!last_chunk) {
            if (lane < R) park[lane] = run;
        } else {
            // ---- row group finished: epilogue ----
            float scale = 1.f;
            if (cnorm) {
                const float* sp = ssq + (cxbuf & 1) * 16;
                const int npass = cc.K >> 10;
                float tot = 0.f;
                for (int c2 = 0; c2 < npass; ++c2) tot += sp[c2];
                scale = 1.0f / sqrtf(tot / (float)cc.K + a.eps);
            }
            const int gi = cc.gb * cc.gblk + cc.gg;
            const int g = gwid + gi * TW;
            const int r0 = g * R;
            const float mine = run * scale;
            if (r0 < cc.N) {
                u64* G = gran_sel(a, cout >= 0 ? cout : 0);
                if (ckind == ENG_RESADD) {
                    if (lane < R) {
                        const float nx = my_xres[gi * R + lane] + mine;
                        my_xres[gi * R + lane] = nx;
                        if (cout >= 0) gran_st(G + r0 + lane, ctag, nx);
                    }
                } else if (ckind == ENG_SILUMUL) {
                    const float up = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), 1));
                    if (lane == 0) {
                        const float h = (mine / (1.0f + expf(-mine))) * up;
                        if (cplain) cvout[g] = h;
                        else if (cout >= 0) gran_st(G + g, ctag, h);
                    }
                } else {
                    if (lane < R) {
                        if (cplain) {
                            cvout[r0 + lane] = mine;
                            // (rows ascend along a wave's groups, so strict > keeps the lowest index of equal logits)
                            if (mine > best_v) { best_v = mine; best_i = r0 + lane; }
                        } else if (cout >= 0) gran_st(G + r0 + lane, ctag, mine);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) lds_add(&ctrl[C_PROG], 1u);                // one more batch of this workgroup finished
        const int phs = cc.ph - p0;
        if (cur_next(cc)) {
            stamp(phs, 2);
            ++cc.ph; ++cc.P;
            fresh = true;
            if (cc.ph < p1) cur_open(cc);
        }
    };

    // ---- the flat batch loop: PF statically named register sets, loads never under a branch ----
    u32x4 q[PF][R][U];
#pragma unroll
    for (int j = 0; j < PF; ++j) load_batch(q[j]);
#pragma unroll
    for (int gi = 0; gi < MAXRES; ++gi)
        if (gi < a.gpw_res && lane < R) my_xres[gi * R + lane] = xv0[gi];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int pw = wave + pp * NSW;
        if (pw < npass0) {                    // stage pass pw of the first phase's input (RMSNorm weight + sum of squares folded)
            float* xs0 = lds + PF0->xoff;
            const bool nrm = PF0->nw != nullptr;
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = xin[pp][i];
                if (nrm) {
                    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                    v[0] *= win[pp][i][0]; v[1] *= win[pp][i][1]; v[2] *= win[pp][i][2]; v[3] *= win[pp][i][3];
                }
                *(f32x4*)(xs0 + xperm((pw * 256 + i * 64 + lane) << 2)) = v;
            }
            if (nrm) {
                ss = wave_sum(ss);
                if (lane == 0) ssq[(PF0->xbuf & 1) * 16 + pw] = ss;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) {
#pragma unroll
                for (int c2 = 0; c2 < CPP; ++c2) lds_add(&ctrl[C_CNT + PF0->xbuf * MAXCH + (CPP > 1 ? pw * CPP + c2 : pw / PPC)], 1u);
            }
        }
    }
    while (cc.ph < p1) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            if (cc.ph < p1) compute(q[j]);
            load_batch(q[j]);
        }
    }

    // ---- exit: residual rows back to HBM for the kernels that follow; block 0 advances the epoch base ----
    for (int gi = 0; gi < a.gpw_res; ++gi) {
        const int row = (gwid + gi * TW) * R + lane;
        if (lane < R && row < a.H) a.xres[row] = my_xres[gi * R + lane];
    }
    if (a.pmax != nullptr) {                  // head phase: this wave's arg-max partial (lanes 0 / 1 = even / odd rows) for argmax_final
        const float ov = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best_v), 1));
        const int oi = __builtin_amdgcn_readlane(best_i, 1);
        if (lane == 0) {
            const bool take = ov > best_v || (ov == best_v && oi < best_i);
            a.pmax[gwid] = take ? ov : best_v;
            a.pidx[gwid] = (take ? oi : best_i) + a.idx_base;      // (global row index: vocabulary shard of a tensor-parallel rank)
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl[0] = base + (uint32_t)a.epoch_step;
}

size_t engine_lds_bytes(const EngArgs& a, int nsw, int ncw) {
    (void)ncw;
    const size_t attn = a.attn != nullptr ? (size_t)(4 * AD + 2 * AD + 2 * ACT * 4 + ACT * 4 * AD + 32 * 18) : 0;
    return ((size_t)a.xf_total + 32 + (size_t)nsw * MAXGB * R + (size_t)nsw * MAXRES * R + C_WORDS + attn) * 4 + 64;
}

// (stream waves per workgroup, register sets in flight per stream wave): the default, or CM_ENG_CFG="nsw,pf" (tuning only)
EngCfg engine_config() {
    static EngCfg c{0, 0, 0};
    if (c.nsw == 0) {
        c = EngCfg{4, ENG_NCW, 4};
        if (const char* e = getenv("CM_ENG_CFG")) {
            int n = 0, p = 0;
            if (sscanf(e, "%d,%d", &n, &p) == 2 && ((n == 4 && p >= 3 && p <= 6) || (n == 8 && (p == 2 || p == 3)))) { c.nsw = n; c.pf = p; }
        }
    }
    return c;
}

template <int NSW, int PF, int NREP = 4, int UC = 4>
static bool prepare_v(size_t lds_bytes) {
    auto k = engine_kernel<NSW, ENG_NCW, PF, NREP, false, UC>;
    auto kt = engine_kernel<NSW, ENG_NCW, PF, NREP, true, UC>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    // the workgroups spin on each other: at least one must fit a CU with this much dynamic LDS (the launch then covers the
    // device with one workgroup per CU; co-residency with OTHER work is the caller's contract, see crane_mi355.h)
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), (NSW + ENG_NCW) * 64, lds_bytes) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        return false;
    }
    return true;
}

template <int NSW, int PF, int NREP = 4, int UC = 4>
static void launch_v(const EngArgs& a, int grid, size_t lds, hipStream_t s, bool trace) {
    if (trace) hipLaunchKernelGGL((engine_kernel<NSW, ENG_NCW, PF, NREP, true, UC>), dim3(grid), dim3((NSW + ENG_NCW) * 64), lds, s, a);
    else hipLaunchKernelGGL((engine_kernel<NSW, ENG_NCW, PF, NREP, false, UC>), dim3(grid), dim3((NSW + ENG_NCW) * 64), lds, s, a);
}

// dependency chunk of 1024 input elements (widths that are multiples of 1024 but not of 2048) or of 512 (the shard widths of a
// tensor-parallel rank): the default configuration only
bool engine_has_chunk(int chunk) {
    const EngCfg c = engine_config();
    return chunk == 2048 || ((chunk == 1024 || chunk == 512) && c.nsw == 4 && c.pf == 4);
}

// GQA group sizes of the in-kernel attention: 4 in every tuning configuration, 2 in the default one
bool engine_has_nrep(int nrep) {
    const EngCfg c = engine_config();
    return nrep == 4 || (nrep == 2 && c.nsw == 4 && c.pf == 4);
}

#define CM_ENG_DISPATCH(CALL)                                            \
    if (c.nsw == 8) { if (c.pf == 2) { CALL(8, 2) } else { CALL(8, 3) } } \
    else if (c.pf == 3) { CALL(4, 3) }                                     \
    else if (c.pf == 5) { CALL(4, 5) }                                     \
    else if (c.pf == 6) { CALL(4, 6) }                                     \
    else { CALL(4, 4) }

// register sets in flight per stream wave at 1024-element chunks (4 KiB batches): CM_ENG_PF1024 = 4 | 6 | 8
static int pf1024() {
    static int v = -1;
    if (v < 0) { v = getenv("CM_ENG_PF1024") ? atoi(getenv("CM_ENG_PF1024")) : 6; if (v != 4 && v != 6 && v != 8) v = 6; }
    return v;
}
#define CM_ENG_1024(CALL, NR) { const int pf = pf1024(); if (pf == 4) { CALL(4, 4, NR, 2) } else if (pf == 6) { CALL(4, 6, NR, 2) } else { CALL(4, 8, NR, 2) } }
// register sets in flight per stream wave at 512-element chunks (2 KiB batches): CM_ENG_PF512 = 8 | 12 | 16
static int pf512() {
    static int v = -1;
    if (v < 0) { v = getenv("CM_ENG_PF512") ? atoi(getenv("CM_ENG_PF512")) : 16; if (v != 8 && v != 12 && v != 16) v = 16; }
    return v;
}
#define CM_ENG_512(CALL, NR) { const int pf = pf512(); if (pf == 8) { CALL(4, 8, NR, 1) } else if (pf == 12) { CALL(4, 12, NR, 1) } else { CALL(4, 16, NR, 1) } }

bool engine_prepare(size_t lds_bytes, int nrep, int chunk) {
    const EngCfg c = engine_config();
    if (chunk == 512) {
        if (!engine_has_chunk(512)) return false;
#define CM_P(N, P, NR, UU) ok = ok && prepare_v<N, P, NR, UU>(lds_bytes);
        bool ok = true;
        if (nrep == 2) CM_ENG_512(CM_P, 2)
        CM_ENG_512(CM_P, 4)
#undef CM_P
        return ok;
    }
    if (chunk == 1024) {
        if (!engine_has_chunk(1024)) return false;
        // both instantiations a handle can launch (whole token with the in-kernel attention of its GQA group; per layer without):
        // a kernel first launched inside a stream capture, unprepared, has been seen to time out on its first replay
#define CM_P(N, P, NR, UU) ok = ok && prepare_v<N, P, NR, UU>(lds_bytes);
        bool ok = true;
        if (nrep == 2) CM_ENG_1024(CM_P, 2)
        CM_ENG_1024(CM_P, 4)
#undef CM_P
        return ok;
    }
    if (nrep == 2 && engine_has_nrep(2)) return prepare_v<4, 4, 2>(lds_bytes) && prepare_v<4, 4, 4>(lds_bytes);
#define CM_ENG_PREP(N, P) return prepare_v<N, P>(lds_bytes);
    CM_ENG_DISPATCH(CM_ENG_PREP)
#undef CM_ENG_PREP
}

bool launch_engine(const EngArgs& a, int grid, hipStream_t s, bool trace) {
    const EngCfg c = engine_config();
    const size_t lds = engine_lds_bytes(a, c.nsw, c.ncw);
    if (lds > 160 * 1024 - 256 || a.gpw_res > MAXRES || a.p1 <= a.p0) return false;
    const bool tr = trace && a.trace != nullptr;
    if (a.chunk == 512) {
        if (!engine_has_chunk(512) || (a.attn != nullptr && a.nrep != 2 && a.nrep != 4)) return false;
#define CM_L(N, P, NR, UU) launch_v<N, P, NR, UU>(a, grid, lds, s, tr);
        if (a.attn != nullptr && a.nrep == 2) CM_ENG_512(CM_L, 2)
        else CM_ENG_512(CM_L, 4)
#undef CM_L
        return true;
    }
    if (a.chunk == 1024) {
        if (!engine_has_chunk(1024) || (a.attn != nullptr && a.nrep != 2 && a.nrep != 4)) return false;
#define CM_L(N, P, NR, UU) launch_v<N, P, NR, UU>(a, grid, lds, s, tr);
        if (a.attn != nullptr && a.nrep == 2) CM_ENG_1024(CM_L, 2)
        else CM_ENG_1024(CM_L, 4)
#undef CM_L
        return true;
    }
    if (a.attn != nullptr && a.nrep == 2) {
        if (!engine_has_nrep(2)) return false;
        launch_v<4, 4, 2>(a, grid, lds, s, tr);
        return true;
    }
#define CM_ENG_LAUNCH(N, P) launch_v<N, P>(a, grid, lds, s, tr);
    CM_ENG_DISPATCH(CM_ENG_LAUNCH)
#undef CM_ENG_LAUNCH
    return true;
}

}  // namespace cm
