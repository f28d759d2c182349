"""Dequant-GEMV bandwidth per ggml type on Qwen3-8B layer shapes: writes a 4-layer GGUF (random weights, quantised by
oracle/gguf_oracle.py), loads it through cm_create and times each projection with cm_bench_kernel.
usage: python tools/bench_gguf.py q4_k [layers]"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import gguf_oracle as G
from crane_amd.backend import Model

kind = sys.argv[1] if len(sys.argv) > 1 else "q4_k"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = dict(model_type="qwen3", hidden_size=4096, intermediate_size=12288, num_attention_heads=32, num_key_value_heads=8,
           head_dim=128, num_hidden_layers=L, vocab_size=8192, tie_word_embeddings=True, rms_norm_eps=1e-6, rope_theta=1e6,
           max_position_embeddings=4096)
gt = G.TYPE_NAMES["q4_k" if kind == "q4_k_m" else kind]
def type_of(gg):          # q4_k_m: Q6_K for attn_v / ffn_down / the embedding (llama.cpp's recipe in outline), Q4_K elsewhere
    if kind == "q4_k_m" and ("attn_v" in gg or "ffn_down" in gg or gg == "token_embd.weight"):
        return G.GGML_Q6_K
    return gt
rng = np.random.default_rng(0)
t0 = time.time()
tensors = []
for hf, gg in G.qwen3_gguf_names(cfg).items():
    if "norm" in gg:
        n = cfg["head_dim"] if ("q_norm" in gg or "k_norm" in gg) else cfg["hidden_size"]
        tensors.append((gg, np.ones(n, np.float32), G.GGML_F32)); continue
    H, I, D = cfg["hidden_size"], cfg["intermediate_size"], cfg["head_dim"]
    shape = {"token_embd": (cfg["vocab_size"], H), "attn_q": (32 * D, H), "attn_k": (8 * D, H), "attn_v": (8 * D, H), "attn_output": (H, 32 * D),
             "ffn_gate": (I, H), "ffn_up": (I, H), "ffn_down": (H, I)}[gg.split(".")[-2]]
    tensors.append((gg, (rng.standard_normal(shape, dtype=np.float32) / np.sqrt(shape[1])).astype(np.float32), type_of(gg)))
path = f"/tmp/bench_{kind}.gguf"
G.write_gguf(path, G.qwen3_metadata(cfg), tensors)
del tensors
print(f"wrote {path} ({os.path.getsize(path) / 1e6:.0f} MB) in {time.time() - t0:.1f}s", flush=True)
m = Model.from_pretrained(path, max_seq_len=2048, max_seqs=9)
for which in ["qkv", "o", "gate_up", "down"]:
    r = m.bench_kernel(which, 360)
    print(f"{kind} {which:8s} {r['ms'] * 1e3:8.2f} us  {r['bytes'] / r['ms'] / 1e9:6.2f} TB/s  ({r['bytes'] / 1e6:.1f} MB)")
m.debug_fill_kv(1024, seed=1)
toks, ms = m.bench_decode(3, 64)
print(f"{kind} {L}-layer decode step {ms / 64 * 1e3:.1f} us")
# batched decode (gemvqb): 8 sequences per pass over the codes
ids = [(7 * i + 3) % cfg["vocab_size"] for i in range(1024)]
for nseq in (2, 4, 8):
    seqs = []
    for i in range(nseq):
        s = m.seq_alloc(); m.seq_forward(s, ids, 0, want_logits=False); seqs.append(s)
    toks = [5 + i for i in range(nseq)]
    for _ in range(3):
        _, g = m.step_batch_decode(seqs, toks, want_logits=False); toks = [int(t) for t in g]
    t0 = time.perf_counter(); K = 32
    for _ in range(K):
        _, g = m.step_batch_decode(seqs, toks, want_logits=False); toks = [int(t) for t in g]
    dt = time.perf_counter() - t0
    print(f"{kind} {L}-layer batch {nseq}: {dt / K * 1e6:8.1f} us/step", flush=True)
    for s in seqs: m.seq_free(s)
# large decode groups and a prompt pass: Q8_0-layout and Q4_K tensors take the int8 matrix cores (kernels_quant_gemm.hip; round 6: Q4_K),
# everything else the batched GEMV in steps of 8 .. 64 rows / the dequantised bf16 hi + lo GEMMs.  CM_Q_GEMM_MIN=0 python ... : A/B
for nseq in (16, 64, 128):
    m2 = Model.from_pretrained(path, max_seq_len=512, max_seqs=nseq + 2)
    seqs = []
    for i in range(nseq):
        s = m2.seq_alloc(); m2.seq_forward(s, ids[:64 + (i % 16)], 0, want_logits=False); seqs.append(s)
    toks = [5 + i for i in range(nseq)]
    for _ in range(3):
        _, g = m2.step_batch_decode(seqs, toks, want_logits=False); toks = [int(t) for t in g]
    t0 = time.perf_counter(); K = 16
    for _ in range(K):
        _, g = m2.step_batch_decode(seqs, toks, want_logits=False); toks = [int(t) for t in g]
    dt = time.perf_counter() - t0
    print(f"{kind} {L}-layer group of {nseq}: {dt / K * 1e6:8.1f} us/round = {dt / K / L * 1e6:7.1f} us per layer", flush=True)
    m2.close()
m3 = Model.from_pretrained(path, max_seq_len=2112, max_seqs=2)
for n in (128, 1024, 2048):
    p = [(7 * i + 3) % cfg["vocab_size"] for i in range(n)]
    m3.clear_kv_cache(); m3.forward_step_greedy(p, 0); m3.clear_kv_cache()
    t0 = time.perf_counter(); m3.forward_step_greedy(p, 0); dt = time.perf_counter() - t0
    print(f"{kind} {L}-layer prompt {n}: {dt * 1e3:7.2f} ms = {dt / L * 1e6:7.1f} us per layer", flush=True)
m3.close()
m.close()
os.remove(path)
