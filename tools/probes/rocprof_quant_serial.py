"""Probe: token-serial prompts over quantised weights (quant_prefill=False) under rocprofv3 --kernel-trace crashed with SIGSEGV in round 5
(tools/bench_qgroup.py before QG_QP=1); prints a marker before every API call so that the log shows which one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from crane_amd import configs
from crane_amd.backend import Model
def say(x): print(x, flush=True)
name = sys.argv[1] if len(sys.argv) > 1 else "tiny-qwen3"
cfg = configs.get_config(name)
if len(sys.argv) > 3: cfg = dict(cfg, num_hidden_layers=int(sys.argv[3]))
V = cfg["vocab_size"]
say("create")
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = Model.synthetic(cfg, seed=0, max_seq_len=128, isq="q8_0", max_seqs=NS + 3, quant_prefill=False)
say("seq_alloc")
s = m.seq_alloc()
say("seq_forward 1 token")
m.seq_forward(s, [3], 0, want_logits=False)
say("seq_forward 3 tokens")
m.seq_forward(s, [4, 5, 6], 1, want_logits=False)
for b in range(NS):
    say(f"seq {b}: alloc + 8 tokens")
    t = m.seq_alloc(); m.seq_forward(t, [(7 * i + 3 + 11 * b) % V for i in range(8)], 0, want_logits=False)
say("batch decode")
m.step_batch_decode([s], [7], want_logits=False)
say("close")
m.close()
say("done")
