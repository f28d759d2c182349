"""bench.py's launcher contract, the part that can be checked without a GPU: `python bench.py --gpus N` on a node with fewer
than N devices must exit non-zero instead of printing a line that claims N GPUs (round-1 review: it used to report TP=1 as
N GPUs)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_more_gpus_than_the_node_has():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have >= 8:
        pytest.skip("node has 8 GPUs")
    want = 8 if have >= 2 else 2
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0
    assert "refusing" in p.stderr and '"metric"' not in p.stdout


def test_committed_bench_line_carries_the_contract_fields():
    """The last bench line of the round (profiles/r04_bench_default.json, written by `python bench.py` on the MI355X) has every
    field the driver and the judge read, and its own numbers are consistent with each other."""
    import json
    line = None
    for l in open(os.path.join(ROOT, "profiles", "r04_bench_default.json")):
        if l.startswith("{"):
            line = json.loads(l)
    assert line is not None
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None and line["dtype"] == "bf16"
    assert abs(line["value"] - 1000.0 / line["ms_per_step"]) < 0.01 * line["value"]          # tokens/s == 1 / step time
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["us_per_launch"] * 1e-6) / 1e9) < 0.01 * r["achieved"]
    assert r["traffic"] is None or 0.9 * r["bytes_per_launch"] < r["traffic"] < 1.5 * r["bytes_per_launch"]
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    p = line["parity"]
    assert p["ok"] is True and p["kv_pages"] == "f16"
    w, t = p["model_written_cache"], p["timed_configuration"]          # a cache the model wrote itself / the timed synthetic cache
    assert w["greedy_equal"] == w["greedy_checked"] and max(w["prefill_logit_rel"], w["decode_logit_rel"]) < 1e-3
    assert t["tokens_equal"] == t["tokens_checked"] and t["logit_rel"] < 1e-3
    lw = p["model_written_cache_ctx1024"]                              # round 4: the headline configuration itself -- every layer, a 1024-token
    assert lw["ok"] is True and lw["prompt_tokens"] == 1024 and lw["layers"] == 36            # cache the model wrote, against oracle/c
    assert lw["greedy_equal"] == lw["greedy_checked"] and max(lw["prefill_logit_rel"], lw["decode_logit_rel"]) < 1e-3
    assert p["logit_rel"] >= max(lw["prefill_logit_rel"], lw["decode_logit_rel"])             # the headline figure includes it
    assert line["roofline_step"]["frac"] >= 0.69
