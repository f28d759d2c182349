"""One decode round of N sequences at a short context under rocprof: python tools/probes/round_profile.py [nseq] [ctx] [isq]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from crane_amd import configs
from crane_amd.backend import Model
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 192
isq = sys.argv[3] if len(sys.argv) > 3 else None
cfg = configs.get_config("qwen3-8b")
m = Model.synthetic(cfg, seed=0, max_seq_len=ctx + 256, max_seqs=nseq + 1, isq=isq)
ids = configs.synthetic_prompt(ctx, cfg["vocab_size"])
seqs = []
for i in range(nseq):
    s = m.seq_alloc(); m.seq_forward(s, ids, 0, want_logits=False); seqs.append(s)
toks = [5 + i for i in range(nseq)]
for _ in range(4):
    _, g = m.step_batch_decode(seqs, toks, want_logits=False); toks = [int(t) for t in g]
best = 1e9
for rep in range(int(os.environ.get("ROUND_REPS", "1"))):
    t0 = time.perf_counter(); K = 16
    for _ in range(K):
        _, g = m.step_batch_decode(seqs, toks, want_logits=False); toks = [int(t) for t in g]
    best = min(best, (time.perf_counter() - t0) / K)
dt = best
print(f"round of {nseq} at ctx {ctx} isq {isq}: {dt * 1e3:.3f} ms = {dt * 1e6 / 36:.1f} us per layer, {nseq / dt:.0f} tok/s", flush=True)
m.close()
