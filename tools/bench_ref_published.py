"""The reference's own published configuration for this path, on one MI355X: Qwen3.5-2B with Q8_0 weights (README.md:86-87: decode
63.0 / 55.6 / 49.2 tok/s at depth 0 / 2048 / 4096 and 2726 tok/s prefill on an RX 7800 XT; README.md:496-503: 4 165- .. 16 429-token
prompts in chunks of 512).  Synthetic weights of that geometry (crane_amd/configs.py "qwen3.5-2b"), in-situ Q8_0 (CRANE_ISQ semantics,
ops/linear.rs:83-116), integer-dot activations (candle CPU QMatMul semantics) -- f16 KV pages.  One JSON line.
usage: python tools/bench_ref_published.py [model] [isq]"""
import json, sys, time
sys.path.insert(0, ".")
from crane_amd import configs
from crane_amd.backend import Model

model = sys.argv[1] if len(sys.argv) > 1 else "qwen3.5-2b"
isq = sys.argv[2] if len(sys.argv) > 2 else "q8_0"
cfg = configs.get_config(model)
V = cfg.get("text_config", cfg)["vocab_size"]
out = {"model": model, "weights": f"synthetic, ISQ {isq}", "kv": "f16 pages", "reference": {
    "hardware": "RX 7800 XT (README.md:86-87)", "decode_tok_s": {"0": 63.0, "2048": 55.6, "4096": 49.2}, "prefill_tok_s": 2726}}
m = Model.synthetic(cfg, seed=0, max_seq_len=16640, max_seqs=2, isq=isq)
dec = {}
for depth in (0, 2048, 4096):
    m.clear_kv_cache()
    m.debug_fill_kv(max(depth, 1), seed=1)
    toks, _ = m.bench_decode(3, 16)
    t0 = time.perf_counter(); toks, ev_ms = m.bench_decode(int(toks[-1]), 128); dt = time.perf_counter() - t0
    dec[str(depth)] = {"tok_s": round(128 / dt, 1), "ms_per_token": round(dt / 128 * 1e3, 4),
                       "bytes_per_token": m.decode_bytes_per_token(depth + 80), "hbm_frac": round(m.decode_bytes_per_token(depth + 80) / (dt / 128) / 8e12, 4)}
out["decode"] = dec
pre = {}
for n in (4165, 8253, 16429):
    ids = [(7 * i + 3) % V for i in range(n)]
    m.clear_kv_cache(); m.forward_step_greedy(ids, 0); m.clear_kv_cache()
    t0 = time.perf_counter(); m.forward_step_greedy(ids, 0); dt = time.perf_counter() - t0
    pre[str(n)] = {"ms": round(dt * 1e3, 2), "tok_s": round(n / dt, 0)}
out["prefill_chunk_2048"] = pre
out["weight_bytes"] = m.weight_bytes() if hasattr(m, "weight_bytes") else None
m.close()
print(json.dumps(out))
