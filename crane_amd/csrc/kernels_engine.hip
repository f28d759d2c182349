// Persistent "chain" kernel for the decode step: the row-streaming projections of one decoder layer that follow the
// attention -- o_proj (+residual) -> RMSNorm -> gate||up (+SiLU*mul) -> down_proj (+residual) -> RMSNorm -> the NEXT
// layer's merged QKV projection -- run as ONE launch on one workgroup per CU instead of four dependent launches
// (reference: DecoderLayer::forward / Mlp::forward / Attention::forward, crane-core/src/models/qwen3/modeling.rs:
// 307-363, 608-642, 698-716).  The arithmetic is the fused GEMV's of kernels_decode.hip, operation for operation.
//
// Why: at batch 1 every projection is a pure weight stream, and a dependent launch costs its boundary plus the ramp
// and drain of the HBM pipe (~20 us of the 81 us a Qwen3-8B layer takes as six launches).  Here the weight stream never
// stops at a dependency edge:
//   * STREAM waves (4 per CU) own fixed row groups of every phase and keep PF register sets of 16-byte non-temporal
//     weight loads in flight (PF x 8 KiB per wave).  Weight addresses do not depend on activations, so while a wave
//     waits for the next phase's input vector the loads of that phase are already landing (the register file is the
//     prefetch ring: 128 KiB per CU).  Loads are unconditional and the register sets statically named, so every wait is
//     a counted vmcnt.
//   * COMM waves (4 per CU) have nothing in their own memory queue, so their polls are not stuck behind a prefetch
//     burst (s_waitcnt vmcnt counts in issue order per wave).  They move each phase's output vector from the 1024
//     producing waves to every CU: a producer writes each f32 as ONE 8-byte {value, tag} granule with a write-through
//     (sc1) agent-scope store; a comm wave sweeps the granules with sc1 loads until every tag equals the epoch of the
//     edge, stages the values into LDS in the GEMV's conflict-free permutation (folding the RMSNorm weight and the
//     sum of squares), and bumps an LDS counter the stream waves poll (LDS traffic uses lgkmcnt, never vmcnt).
//     The data is its own flag: no fences, no release/acquire, placement-independent (MI355X_MICROARCH.md, Guideline 16 R2).
//   * the residual stream row r is owned by the same lane of the same wave in o_proj and down_proj and lives in LDS
//     between phases.
// Every spin is bounded; a timeout raises ctl[1], later launches return at once, and the host reports the code.
#include <cstdio>
#include <cstdlib>

#include "dev_common.h"
#include "kernels.h"

namespace cm {

namespace {

constexpr int R = 2;            // rows per group (one wave reduces R rows at a time)
constexpr int U = 4;            // 512-element chunks per batch: a batch = R x U 16-byte loads per lane = 8 KiB per wave
constexpr uint32_t SPIN_LDS = 400000u;     // x ~0.1 us
constexpr uint32_t SPIN_GLOBAL = 60000u;   // x ~0.5 us

typedef unsigned long long u64;
// every pointer taken from the phase table is a global-memory pointer: say so, or the loads become flat_* (which count on
// lgkmcnt as well and force full waits)
#define CM_GLOBAL __attribute__((address_space(1)))
#define CM_CONST __attribute__((address_space(4)))
typedef const CM_CONST EngPhase* ph_ptr;
typedef const CM_GLOBAL u32x4* gw_ptr;
typedef const CM_GLOBAL float* gf_cptr;
typedef CM_GLOBAL float* gf_ptr;

__device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add(uint32_t* p, uint32_t v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// float index inside an LDS input buffer: the permutation gemv_bf16_kernel stages x with (two conflict-free
// ds_read_b128 per lane and chunk)
__device__ __forceinline__ int xperm(int k) {
    const int c = k >> 9, j = k & 511;
    return ((c * 128 + ((j >> 2) & 1) * 64 + (j >> 3)) << 2) + (j & 3);
}

}  // namespace

template <int NSW, int NCW, int PF, bool TRACE>
__global__ __launch_bounds__((NSW + NCW) * 64, (NSW + NCW + 3) / 4) void engine_chain_kernel(EngArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* ssq = lds + a.xf_total;                       // [2][NCW] partial sums of squares
    float* xres_l = ssq + 2 * NCW;                       // [NSW][gpw_res][R] residual rows of this block's stream waves
    uint32_t* ctrl = (uint32_t*)(xres_l + NSW * a.gpw_res * R);   // [0] inputs staged (x NCW), [1] stream waves done (x NSW), [2] abort

    // TRACE (debug instantiation, cm_debug_read "engine_trace"): 100 MHz timestamps per (wave, phase, event)
    auto stamp = [&](int ph, int k, u64 val = ~0ull) __attribute__((always_inline)) {
        if constexpr (TRACE) {
            if (lane == 0) a.trace[(((size_t)blockIdx.x * (NSW + NCW) + wave) * ENG_MAXPH + ph) * 4 + k] = val == ~0ull ? __builtin_amdgcn_s_memrealtime() : val;
        }
    };
    if (__builtin_nontemporal_load(&a.ctl[1]) != 0u) return;       // an earlier launch timed out: do nothing
    const uint32_t base = __builtin_nontemporal_load(&a.ctl[0]);   // epoch base of this launch (tags base+1 ... base+nph)
    if (threadIdx.x < 4) ctrl[threadIdx.x] = 0u;
    __syncthreads();

    const int nph = a.nph;

    if (wave >= NSW) {
        // =========================== COMM waves: stage every phase's input vector into LDS ===========================
        const int cw = wave - NSW;
        for (int p = 1; p < nph; ++p) {                  // phase 0's input is staged by the stream waves (below)
            ph_ptr P = (ph_ptr)a.prog + p;
            const int K = P->K, in_edge = P->in_edge;
            gf_cptr nw = (gf_cptr)P->nw;
            gf_cptr vin = (gf_cptr)P->vin;
            float* xs = lds + P->xoff;
            stamp(p, 0);
            uint32_t total_spins = 0;
            {                                             // do not poll HBM while this CU's own waves are mid-phase
                uint32_t spins = 0;
                while (lds_ld(&ctrl[1]) < (uint32_t)(NSW * p) && lds_ld(&ctrl[2]) == 0u) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > SPIN_LDS) { if (lane == 0) { atomicExch(&a.ctl[1], 0x100u + (uint32_t)p); ctrl[2] = 1u; } break; }
                }
            }
            stamp(p, 1);
            const u64* G = in_edge == 0 ? a.gran0 : (in_edge == 1 ? a.gran1 : a.gran2);
            const uint32_t tag = base + (uint32_t)p;      // written by phase p - 1
            float ss = 0.f;
            for (int pass = cw; pass * 1024 < K; pass += NCW) {
                const int kb = pass * 1024 + lane;
                float v[16], wv[16];
                if (nw != nullptr) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) wv[i] = nw[kb + i * 64];
                }
                if (in_edge < 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = vin[kb + i * 64];
                } else {
                    uint32_t spins = 0;
                    for (;;) {
                        bool ok = true;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const u64 x = __hip_atomic_load(G + kb + i * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            v[i] = __uint_as_float((uint32_t)x);
                            ok = ok && ((uint32_t)(x >> 32) == tag);
                        }
                        if (__all(ok)) break;
                        if (lds_ld(&ctrl[2]) != 0u) break;
                        if (++spins > SPIN_GLOBAL) { if (lane == 0) { atomicExch(&a.ctl[1], 0x200u + (uint32_t)p); ctrl[2] = 1u; } break; }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    total_spins += spins;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float val = v[i];
                    if (nw != nullptr) { ss += val * val; val *= wv[i]; }
                    xs[xperm(kb + i * 64)] = val;
                }
            }
            if (nw != nullptr) {
                ss = wave_sum(ss);
                if (lane == 0) ssq[(p & 1) * NCW + cw] = ss;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_add(&ctrl[0], (uint32_t)NSW);      // NSW * NCW units per staged phase
            stamp(p, 2);
            stamp(p, 3, (u64)total_spins);
        }
        return;
    }

    // =============================== STREAM waves: rows x weights, never waiting on HBM ===============================
    const int gwid = blockIdx.x * NSW + wave, TW = gridDim.x * NSW;
    stamp(0, 3);

    // residual rows owned by this wave (same mapping in o_proj and down_proj): lane i < R of group gi.  Requested first
    // and written to LDS only after the weight prefetch has been issued, so the wait is a counted one.
    constexpr int MAXRES = 4;
    float xv0[MAXRES];
#pragma unroll
    for (int gi = 0; gi < MAXRES; ++gi) {
        int row = (gwid + gi * TW) * R + (lane < R ? lane : 0);
        row = row < a.H ? row : a.H - 1;
        xv0[gi] = a.xres[row];
    }
    // Phase 0's input vector was written by an earlier kernel: the stream waves request their 1/NSW of it BEFORE the weight
    // prefetch (a load issued behind this CU's 128 KiB prefetch burst returns ~5 us later: measured 7.7 us until the
    // o_proj rows could start when the comm waves did this), and stage it once the prefetch is on its way.
    constexpr int MAXV = 8;
    ph_ptr P0 = (ph_ptr)a.prog;
    const int k4n = P0->K >> 2, per = k4n / NSW;                   // f32x4 groups of the vector, per stream wave
    f32x4 xin[MAXV];
    {
        const CM_GLOBAL f32x4* v4 = (const CM_GLOBAL f32x4*)P0->vin;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int idx = wave * per + i * 64 + lane;
            idx = idx < k4n ? idx : k4n - 1;
            xin[i] = v4[idx];
        }
    }

    // ---- load-side cursor (runs PF batches ahead of the compute-side cursor) ----
    ph_ptr LP = (ph_ptr)a.prog;
    int lph = 0, lgi = 0, lkb = 0;
    const CM_GLOBAL uint16_t* lW = (const CM_GLOBAL uint16_t*)LP->W;
    int lN = LP->N, lK = LP->K, lgpw = LP->gpw, lnb = LP->nbpg;
    bool lvalid = true;
    auto load_batch = [&](u32x4 (&q)[R][U]) __attribute__((always_inline)) {
        const int G = lN / R;
        const int g = gwid + lgi * TW;
        // a row group past the last one, or a batch past the end of the program: every lane reads the same 16 bytes of the
        // matrix (no HBM traffic, result never used) -- the loads stay unconditional
        const bool real = lvalid && g < G;
        const size_t roff = real ? (size_t)g * R * (size_t)lK + (size_t)lkb * (U * 512) : 0;
        const int loff = real ? lane * 8 : 0;
        const size_t sK = real ? (size_t)lK : 0;
        const int sU = real ? 512 : 0;
        const CM_GLOBAL uint16_t* wp = lW + roff + loff;
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int u = 0; u < U; ++u) q[i][u] = __builtin_nontemporal_load((gw_ptr)(wp + i * sK + u * sU));
        if (++lkb == lnb) {
            lkb = 0;
            if (++lgi == lgpw) {
                lgi = 0;
                if (lph + 1 < nph) {
                    ++lph; ++LP;
                    lW = (const CM_GLOBAL uint16_t*)LP->W; lN = LP->N; lK = LP->K; lgpw = LP->gpw; lnb = LP->nbpg;
                } else {
                    lvalid = false;
                }
            }
        }
    };

    // ---- compute-side cursor ----
    ph_ptr CP = (ph_ptr)a.prog;
    int cph = 0, cgi = 0, ckb = 0;
    float acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.f;
    float scale = 1.f;
    const f32x4* xs4 = (const f32x4*)lds;
    int cN = 0, cK = 0, cgpw = 1, cnb = 1, ckind = 0, cout = -1;
    gf_ptr cvout = nullptr;

    auto compute = [&](u32x4 (&q)[R][U]) __attribute__((always_inline)) {
        if (cgi == 0 && ckb == 0) {
            // ---- phase start: parameters, then wait until the comm waves have staged this phase's input ----
            cN = CP->N; cK = CP->K; cgpw = CP->gpw; cnb = CP->nbpg; ckind = CP->kind; cout = CP->out_edge; cvout = (gf_ptr)CP->vout;
            xs4 = (const f32x4*)(lds + CP->xoff);
            const uint32_t want = (uint32_t)(NSW * NCW * (cph + 1));
            stamp(cph, 0);
            uint32_t spins = 0;
            while (lds_ld(&ctrl[0]) < want && lds_ld(&ctrl[2]) == 0u) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_LDS) { if (lane == 0) { atomicExch(&a.ctl[1], 0x300u + (uint32_t)cph); ctrl[2] = 1u; } break; }
            }
            asm volatile("" ::: "memory");
            stamp(cph, 1);
            scale = 1.f;
            if (CP->nw != nullptr) {
                const float* sp = ssq + (cph & 1) * NCW;
                float tot = 0.f;
                if (NCW == 4) tot = (sp[0] + sp[1]) + (sp[2] + sp[3]);
                else for (int c = 0; c < NCW; ++c) tot += sp[c];
                scale = 1.0f / sqrtf(tot / (float)cK + a.eps);
            }
        }
        const int cb = ckb * U;
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
        if (++ckb < cnb) return;
        ckb = 0;
        // ---- row group finished: reduce + epilogue ----
        const int g = gwid + cgi * TW;
        const int r0 = g * R;
        float v[R];
#pragma unroll
        for (int i = 0; i < R; ++i) { v[i] = wave_sum(acc[i]) * scale; acc[i] = 0.f; }
        if (r0 < cN) {
            u64* G = cout == 0 ? a.gran0 : (cout == 1 ? a.gran1 : a.gran2);
            const u64 tagw = (u64)(base + 1u + (uint32_t)cph) << 32;
            const float mine = lane == 1 ? v[1] : v[0];
            if (ckind == ENG_RESADD) {
                float* xr = xres_l + (wave * a.gpw_res + cgi) * R;
                if (lane < R) {
                    const float nx = xr[lane] + mine;
                    xr[lane] = nx;
                    if (cout >= 0) __hip_atomic_store(G + r0 + lane, tagw | (u64)__float_as_uint(nx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if (ckind == ENG_SILUMUL) {
                if (lane == 0) {
                    const float h = (v[0] / (1.0f + expf(-v[0]))) * v[1];
                    if (cout >= 0) __hip_atomic_store(G + g, tagw | (u64)__float_as_uint(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else cvout[g] = h;
                }
            } else {
                if (lane < R) {
                    if (cout >= 0) __hip_atomic_store(G + r0 + lane, tagw | (u64)__float_as_uint(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else cvout[r0 + lane] = mine;
                }
            }
        }
        if (++cgi == cgpw) {
            cgi = 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_add(&ctrl[1], 1u);               // this wave is done with phase cph
            stamp(cph, 2);
            ++cph; ++CP;
        }
    };

    // ---- the flat batch loop: PF statically named register sets, loads never under a branch ----
    u32x4 q[PF][R][U];
#pragma unroll
    for (int j = 0; j < PF; ++j) load_batch(q[j]);
#pragma unroll
    for (int gi = 0; gi < MAXRES; ++gi)
        if (gi < a.gpw_res && lane < R) xres_l[(wave * a.gpw_res + gi) * R + lane] = xv0[gi];
    {
        float* xs0 = lds + P0->xoff;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (i * 64 < per) *(f32x4*)(xs0 + xperm((wave * per + i * 64 + lane) << 2)) = xin[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) lds_add(&ctrl[0], (uint32_t)NCW);
    }
    while (cph < nph) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            if (cph < nph) compute(q[j]);
            load_batch(q[j]);
        }
    }

    // ---- exit: residual rows back to HBM for the kernels that follow; block 0 advances the epoch base ----
    for (int gi = 0; gi < a.gpw_res; ++gi) {
        const int row = (gwid + gi * TW) * R + lane;
        if (lane < R && row < a.H) a.xres[row] = xres_l[(wave * a.gpw_res + gi) * R + lane];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl[0] = base + (uint32_t)nph;
}

size_t engine_lds_bytes(const EngArgs& a, int nsw, int ncw) {
    return ((size_t)a.xf_total + 2 * (size_t)ncw + (size_t)nsw * a.gpw_res * R) * 4 + 64;
}

// (stream waves per workgroup, register sets in flight per stream wave): the default, or CM_ENG_CFG="nsw,pf" (tuning only)
EngCfg engine_config() {
    static EngCfg c{0, 0, 0};
    if (c.nsw == 0) {
        c = EngCfg{ENG_NSW, ENG_NCW, ENG_PF};
        if (const char* e = getenv("CM_ENG_CFG")) {
            int n = 0, p = 0;
            if (sscanf(e, "%d,%d", &n, &p) == 2 && ((n == 4 && (p == 3 || p == 4)) || (n == 8 && (p == 2 || p == 3)))) { c.nsw = n; c.pf = p; }
        }
    }
    return c;
}

template <int NSW, int PF>
static bool prepare_v(size_t lds_bytes) {
    auto k = engine_chain_kernel<NSW, ENG_NCW, PF, false>;
    auto kt = engine_chain_kernel<NSW, ENG_NCW, PF, true>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return true;
}

bool engine_prepare(size_t lds_bytes) {
    const EngCfg c = engine_config();
    if (c.nsw == 8) return c.pf == 2 ? prepare_v<8, 2>(lds_bytes) : prepare_v<8, 3>(lds_bytes);
    return c.pf == 3 ? prepare_v<4, 3>(lds_bytes) : prepare_v<4, 4>(lds_bytes);
}

template <int NSW, int PF>
static void launch_v(const EngArgs& a, int grid, size_t lds, hipStream_t s, bool trace) {
    if (trace) hipLaunchKernelGGL((engine_chain_kernel<NSW, ENG_NCW, PF, true>), dim3(grid), dim3((NSW + ENG_NCW) * 64), lds, s, a);
    else hipLaunchKernelGGL((engine_chain_kernel<NSW, ENG_NCW, PF, false>), dim3(grid), dim3((NSW + ENG_NCW) * 64), lds, s, a);
}

bool launch_engine_chain(const EngArgs& a, int grid, hipStream_t s, bool trace) {
    const EngCfg c = engine_config();
    const size_t lds = engine_lds_bytes(a, c.nsw, c.ncw);
    if (lds > 160 * 1024 - 256 || a.gpw_res > 4) return false;
    const bool tr = trace && a.trace != nullptr;
    if (c.nsw == 8) { if (c.pf == 2) launch_v<8, 2>(a, grid, lds, s, tr); else launch_v<8, 3>(a, grid, lds, s, tr); }
    else { if (c.pf == 3) launch_v<4, 3>(a, grid, lds, s, tr); else launch_v<4, 4>(a, grid, lds, s, tr); }
    return true;
}

}  // namespace cm
