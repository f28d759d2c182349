"""One 128-sequence decode round over a 4-layer Q4_K (or q8_0 / q4_k_m) GGUF at the 8B widths, for rocprofv3 --kernel-trace."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import gguf_oracle as G
from crane_amd.backend import Model
kind = sys.argv[1] if len(sys.argv) > 1 else "q4_k"
L = 4
cfg = dict(model_type="qwen3", hidden_size=4096, intermediate_size=12288, num_attention_heads=32, num_key_value_heads=8,
           head_dim=128, num_hidden_layers=L, vocab_size=8192, tie_word_embeddings=True, rms_norm_eps=1e-6, rope_theta=1e6, max_position_embeddings=4096)
gt = G.TYPE_NAMES["q4_k" if kind == "q4_k_m" else kind]
rng = np.random.default_rng(0)
tensors = []
for hf, gg in G.qwen3_gguf_names(cfg).items():
    if "norm" in gg:
        n = cfg["head_dim"] if ("q_norm" in gg or "k_norm" in gg) else cfg["hidden_size"]
        tensors.append((gg, np.ones(n, np.float32), G.GGML_F32)); continue
    H, I, D = cfg["hidden_size"], cfg["intermediate_size"], cfg["head_dim"]
    shape = {"token_embd": (cfg["vocab_size"], H), "attn_q": (32 * D, H), "attn_k": (8 * D, H), "attn_v": (8 * D, H), "attn_output": (H, 32 * D),
             "ffn_gate": (I, H), "ffn_up": (I, H), "ffn_down": (H, I)}[gg.split(".")[-2]]
    t = G.GGML_Q6_K if (kind == "q4_k_m" and ("attn_v" in gg or "ffn_down" in gg or gg == "token_embd.weight")) else gt
    tensors.append((gg, (rng.standard_normal(shape, dtype=np.float32) / np.sqrt(shape[1])).astype(np.float32), t))
path = f"/tmp/trace_{kind}.gguf"
G.write_gguf(path, G.qwen3_metadata(cfg), tensors)
nseq = 128
m = Model.from_pretrained(path, max_seq_len=512, max_seqs=nseq + 2)
ids = [(7 * i + 3) % cfg["vocab_size"] for i in range(1024)]
seqs = []
for i in range(nseq):
    s = m.seq_alloc(); m.seq_forward(s, ids[:64 + (i % 16)], 0, want_logits=False); seqs.append(s)
toks = [5 + i for i in range(nseq)]
for _ in range(20):
    _, g = m.step_batch_decode(seqs, toks, want_logits=False); toks = [int(t) for t in g]
p = ids[:1024]
for _ in range(3):
    m.seq_forward(seqs[0], p, 0, want_logits=False)
m.close(); os.remove(path)
