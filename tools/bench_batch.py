#!/usr/bin/env python3
"""Aggregate decode throughput with N concurrent sequences (cm_decode_batch), Qwen3-8B, context ~1024."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crane_amd import configs
from crane_amd.backend import Model
model = sys.argv[1] if len(sys.argv) > 1 else "qwen3-8b"
isq = sys.argv[2] if len(sys.argv) > 2 else None            # e.g. q8_0: batched decode over in-situ quantised weights
cfg = configs.get_config(model)
m = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=17, isq=isq)
ids = configs.synthetic_prompt(1024, cfg["vocab_size"])
for nseq in (1, 2, 4, 8, 16):
    seqs = []
    for i in range(nseq):
        s = m.seq_alloc(); m.seq_forward(s, ids, 0, want_logits=False); seqs.append(s)
    toks = [5 + i for i in range(nseq)]
    for _ in range(3):
        _, g = m.step_batch_decode(seqs, toks, want_logits=False); toks = [int(t) for t in g]
    t0 = time.perf_counter(); K = 24
    for _ in range(K):
        _, g = m.step_batch_decode(seqs, toks, want_logits=False); toks = [int(t) for t in g]
    dt = time.perf_counter() - t0
    tag = model + (" isq " + isq if isq else "")
    print(f"{tag} batch {nseq:2d}: {dt/K*1e3:7.3f} ms/step  {nseq*K/dt:8.1f} tok/s aggregate", flush=True)
    for s in seqs: m.seq_free(s)
