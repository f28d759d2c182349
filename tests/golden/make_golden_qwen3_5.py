#!/usr/bin/env python3
"""Generate tests/golden/qwen3_5_*.npz with HuggingFace transformers' Qwen3_5ForCausalLM (CPU, f32, eager).
See make_golden_qwen3.py for why HF is the source of golden vectors.  Run from the repo root."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from crane_amd import configs, synth  # noqa: E402
from transformers import Qwen3_5ForCausalLM, Qwen3_5TextConfig  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
PROMPT_LEN, NEW = 19, 10


def main():
    torch.set_num_threads(8)
    # "qwen3.8-27b-geom4": the real Qwen3.8-27B layer geometry (H 5120, 24 q / 4 kv heads x 256, 16 key / 48 value GDN heads,
    # I 17408) with 4 layers and a 4096-entry vocabulary -- the configuration of test_hip_qwen38_27b_geometry
    for name in ("tiny-qwen3.5", "qwen3.8-27b-geom4"):
        if name == "qwen3.8-27b-geom4":
            cfg = dict(configs.get_config("qwen3.8-27b"), num_hidden_layers=4, vocab_size=4096, max_position_embeddings=4096)
        else:
            cfg = configs.get_config(name)
        w = synth.synth_weights_f32(cfg, seed=0)
        hc = Qwen3_5TextConfig(**{k: v for k, v in cfg.items()
                                  if k not in ("model_type", "torch_dtype", "full_attention_interval", "attn_output_gate")})
        hc._attn_implementation = "eager"
        m = Qwen3_5ForCausalLM(hc).float().eval()
        res = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=False)
        assert not res.unexpected_keys and not res.missing_keys, res
        ids = configs.synthetic_prompt(PROMPT_LEN, cfg["vocab_size"])
        with torch.no_grad():
            out = m(torch.tensor([ids]), use_cache=True)
            prefill_logits = out.logits[0, -1].numpy().astype(np.float32)
            step = m(torch.tensor([[5]]), past_key_values=out.past_key_values)
            decode_logits = step.logits[0, -1].numpy().astype(np.float32)
            toks = m.generate(torch.tensor([ids]), max_new_tokens=NEW, do_sample=False)[0].tolist()
        np.savez_compressed(os.path.join(OUT, f"qwen3_5_{name}.npz"), prompt=np.array(ids, dtype=np.int64),
                            prefill_logits=prefill_logits, decode_token=np.array([5]), decode_logits=decode_logits,
                            greedy_tokens=np.array(toks, dtype=np.int64), seed=np.array([0]))
        print(name, "max|logit|", float(np.abs(prefill_logits).max()), "tokens", toks[PROMPT_LEN:])


if __name__ == "__main__":
    main()
