import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crane_amd import configs
from crane_amd.backend import Model
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 8
isq = sys.argv[2] if len(sys.argv) > 2 else None
cfg = configs.get_config("qwen3-8b")
m = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=nseq + 1, isq=isq)
ids = configs.synthetic_prompt(1024, cfg["vocab_size"])
seqs = []
for i in range(nseq):
    s = m.seq_alloc(); m.seq_forward(s, ids, 0, want_logits=False); seqs.append(s)
toks = [5 + i for i in range(nseq)]
for _ in range(6):
    _, g = m.step_batch_decode(seqs, toks, want_logits=False); toks = [int(t) for t in g]
