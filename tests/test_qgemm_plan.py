"""Host logic of the int8-MFMA decode-group GEMM (crane_amd/csrc/kernels_quant_gemm.hip: plan_gemm_q8 through cm_debug_qgemm_plan):
geometry, K split and workspace bound for the projection shapes of the BASELINE models -- no GPU.  (What the plan feeds is held to the
ggml-semantics oracle row by row in tests/test_gpu_quant.py; the reference has no counterpart: it calls candle's QMatMul once per
sequence, ops/linear.rs:53-116.)"""
import ctypes as C

import pytest

from crane_amd import _lib, configs

STORE, RESADD, SILUMUL = 0, 1, 2
ALLOWED_KS = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16}          # splits the reduction kernels are unrolled for


def plan(m, n, k, epi, ws_floats, num_cu=256):
    lib = _lib.load()
    out = (C.c_int64 * 8)()
    assert lib.cm_debug_qgemm_plan(m, n, k, epi, ws_floats, num_cu, out) == 0
    keys = ("ok", "direct", "waves", "rows", "qg", "groups", "ks", "grid")
    return dict(zip(keys, [int(v) for v in out]))


def shapes(name):
    c = configs.get_config(name)
    c = c.get("text_config", c)
    H, I, D = c["hidden_size"], c["intermediate_size"], c.get("head_dim", c["hidden_size"] // c["num_attention_heads"])
    qkv = (c["num_attention_heads"] + 2 * c["num_key_value_heads"]) * D
    return [(qkv, H, STORE), (H, c["num_attention_heads"] * D, RESADD), (2 * I, H, SILUMUL), (H, I, RESADD), (c["vocab_size"], H, STORE)]


@pytest.mark.parametrize("model", ["qwen3-8b", "qwen3-0.6b", "qwen3-vl-2b"])
@pytest.mark.parametrize("m", [8, 12, 32, 33, 40, 64, 65, 96, 128])
def test_plan_of_every_projection(model, m):
    ws = 16 * 1024 * 1024                                    # f32 of workspace (64 MB)
    for n, k, epi in shapes(model):
        if n % 128 or k % 256:
            assert plan(m, n, k, epi, ws)["ok"] == 0          # shapes the kernel does not take: the batched GEMV keeps them
            continue
        p = plan(m, n, k, epi, ws)
        assert p["ok"] == 1, (model, m, n, k, epi, p)
        assert p["rows"] >= m and p["rows"] == (32 if m <= 32 else 64 if m <= 64 else 128)
        assert p["waves"] == (4 if m <= 64 else 8) and p["qg"] in (4, 8)
        assert p["groups"] * p["qg"] * 32 == k
        assert p["ks"] in ALLOWED_KS and 1 <= p["ks"] <= p["groups"]
        assert p["grid"] == (n // 128) * p["ks"]
        # every slice [i G / ks, (i + 1) G / ks) holds at least one group, together they cover the row
        G, ks = p["groups"], p["ks"]
        cuts = [i * G // ks for i in range(ks + 1)]
        assert cuts[0] == 0 and cuts[-1] == G and all(b > a for a, b in zip(cuts, cuts[1:]))
        if p["direct"]:
            assert p["ks"] == 1 and epi in (STORE, SILUMUL)
        else:
            assert p["ks"] * m * n <= ws                      # the partial slices fit the workspace
        if epi == RESADD:
            assert not p["direct"]                            # a residual projection always reduces through the workspace


def test_plan_respects_the_workspace_and_refuses_what_it_cannot_hold():
    m, n, k = 128, 4096, 4096
    assert plan(m, n, k, RESADD, 0)["ok"] == 0                # no workspace: the caller's GEMV fallback, never a write past it
    assert plan(m, n, k, RESADD, m * n - 1)["ok"] == 0
    p = plan(m, n, k, RESADD, 3 * m * n)
    assert p["ok"] == 1 and p["ks"] <= 3
    # a store that the workspace cannot hold once is written in place, unsplit (the vocabulary head)
    p = plan(m, 151936 - 151936 % 128, k, STORE, 1024)
    assert p["ok"] == 1 and p["direct"] == 1 and p["ks"] == 1
    # gate|up unsplit stores SiLU(gate) * up itself; with one K group per slice at most `groups` slices
    p = plan(128, 256, 256, SILUMUL, 16 * 1024 * 1024)
    assert p["ok"] == 1 and p["ks"] <= p["groups"]
    # rows, shapes and epilogues outside the kernel
    assert plan(0, n, k, STORE, 1 << 24)["ok"] == 0 and plan((1 << 16) + 1, n, k, STORE, 1 << 34)["ok"] == 0
    # more than 128 rows (a prompt pass, round 6): 8 waves x 4 m-tiles = 256 rows per workgroup, the grid walks ceil(m / 256) m-panels
    p = plan(129, n, k, STORE, 1 << 24)
    assert p["ok"] == 1 and p["waves"] == 8 and p["rows"] == 256 and p["grid"] == (n // 128) * 1 * p["ks"]
    p = plan(1000, n, k, RESADD, 1 << 24)
    assert p["ok"] == 1 and p["rows"] == 256 and p["grid"] == (n // 128) * 4 * p["ks"] and p["ks"] * 1000 * n <= 1 << 24
    p = plan(2048, 24576, 4096, SILUMUL, 1 << 24)               # the partial slices would not fit: written unsplit, SiLU in the GEMM's epilogue
    assert p["ok"] == 1 and p["direct"] == 1 and p["ks"] == 1 and p["grid"] == 192 * 8
    assert plan(2048, 24576, 4096, RESADD, 1 << 24)["ok"] == 0  # (a residual projection of that size has nowhere to put its sums)
    assert plan(m, n + 64, k, STORE, 1 << 24)["ok"] == 0 and plan(m, n, k + 32, STORE, 1 << 24)["ok"] == 0
    assert plan(m, n, k, 3, 1 << 24)["ok"] == 0


def test_plan_of_the_k_quant_kernels():
    """Round 6: Q4_K / Q6_K tensors on the same GEMM (epi | ggml type << 8).  Q4_K carries two int8 weight planes per panel (78 KB of
    LDS): one workgroup per CU whatever the rows, so 8 waves from the smallest group on, m-panels of 128 rows in a prompt pass; Q6_K
    keeps the Q8_0 geometries (256-row panels above 128 rows).  K groups are half a 256-block for both."""
    Q4K, Q6K = 12 << 8, 14 << 8
    n, k = 4096, 4096
    for m, rows in ((5, 64), (40, 64), (100, 128), (128, 128)):
        p = plan(m, n, k, STORE | Q4K, 1 << 24)
        assert p["ok"] == 1 and p["waves"] == 8 and p["rows"] == rows and p["qg"] == 4 and p["groups"] == k // 128, (m, p)
        assert p["grid"] == (n // 128) * p["ks"]
    p = plan(300, n, k, RESADD | Q4K, 1 << 24)
    assert p["ok"] == 1 and p["rows"] == 128 and p["grid"] == (n // 128) * 3 * p["ks"], p
    p = plan(300, n, k, RESADD | Q6K, 1 << 24)
    assert p["ok"] == 1 and p["rows"] == 256 and p["grid"] == (n // 128) * 2 * p["ks"], p
    p = plan(20, n, k, STORE | Q6K, 1 << 24)
    assert p["ok"] == 1 and p["waves"] == 4 and p["rows"] == 32, p
