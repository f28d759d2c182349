"""gemm_q4k_i8_kernel at 128 rows on the 8B gate|up / down shapes, for rocprofv3 --pmc: cm_debug_qgemm in a loop."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import gguf_oracle as G
from crane_amd.backend import Model
kind = sys.argv[1] if len(sys.argv) > 1 else "q4_k"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cfg = dict(model_type="qwen3", hidden_size=4096, intermediate_size=12288, num_attention_heads=32, num_key_value_heads=8,
           head_dim=128, num_hidden_layers=1, vocab_size=1024, tie_word_embeddings=True, rms_norm_eps=1e-6, rope_theta=1e6, max_position_embeddings=4096)
gt = G.TYPE_NAMES[kind]
rng = np.random.default_rng(0)
tensors = []
for hf, gg in G.qwen3_gguf_names(cfg).items():
    if "norm" in gg:
        n = cfg["head_dim"] if ("q_norm" in gg or "k_norm" in gg) else cfg["hidden_size"]
        tensors.append((gg, np.ones(n, np.float32), G.GGML_F32)); continue
    H, I, D = cfg["hidden_size"], cfg["intermediate_size"], cfg["head_dim"]
    shape = {"token_embd": (cfg["vocab_size"], H), "attn_q": (32 * D, H), "attn_k": (8 * D, H), "attn_v": (8 * D, H), "attn_output": (H, 32 * D),
             "ffn_gate": (I, H), "ffn_up": (I, H), "ffn_down": (H, I)}[gg.split(".")[-2]]
    tensors.append((gg, (rng.standard_normal(shape, dtype=np.float32) / np.sqrt(shape[1])).astype(np.float32), gt))
path = f"/tmp/pmc_{kind}.gguf"
G.write_gguf(path, G.qwen3_metadata(cfg), tensors)
m = Model.from_pretrained(path, max_seq_len=256, max_seqs=2)
x = rng.standard_normal((M, 4096)).astype(np.float32)
xi = rng.standard_normal((M, 12288)).astype(np.float32)
for _ in range(10):
    m.debug_qgemm(0, "gate_up", x, 2 * 12288)
    m.debug_qgemm(0, "down", xi, 4096)
m.close(); os.remove(path)
