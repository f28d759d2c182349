#!/usr/bin/env python3
"""Generate tests/golden/qwen3_*.npz with HuggingFace transformers (CPU, f32, eager attention).

The reference (lucasjinreal/Crane) cannot be built in this image (no Rust, candle not vendored)
and stores no logits/token fixtures of its own; it claims bit-exact arg-max
parity with HF transformers (reference README.md:402-404).  So the golden vectors come from HF
run on the deterministic synthetic checkpoints of crane_amd/synth.py.  Run from the repo root:

    python tests/golden/make_golden_qwen3.py

Needs: torch, transformers (present in the build container; not needed to *run* the tests).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from crane_amd import configs, synth  # noqa: E402
from transformers import Qwen3Config, Qwen3ForCausalLM  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
PROMPT_LEN, NEW = 21, 12


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    # "qwen3-8b-2l": the headline geometry (every projection shape of Qwen3-8B, GQA 4, the 151 936-row untied lm_head) at
    # 2 layers -- the configuration tests/test_gpu_parity_headline.py compares the HIP path with oracle/c on
    # "qwen3-0.6b-2l": the same for BASELINE configs[0] (Qwen3-0.6B widths, 16 q / 8 kv heads, TIED 151 936-row table)
    for name in ("tiny-qwen3", "tiny-qwen3-untied", "qwen3-8b-2l", "qwen3-0.6b-2l"):
        cfg = configs.get_config(name)
        w = synth.synth_weights_f32(cfg, seed=0)
        hc = Qwen3Config(**{k: v for k, v in cfg.items() if k not in ("model_type", "torch_dtype", "use_qk_norm")})
        hc._attn_implementation = "eager"
        m = Qwen3ForCausalLM(hc).float().eval()
        missing = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=False)
        assert not missing.unexpected_keys, missing
        assert all(k == "lm_head.weight" for k in missing.missing_keys), missing      # tied head
        ids = configs.synthetic_prompt(PROMPT_LEN, cfg["vocab_size"])
        with torch.no_grad():
            out = m(torch.tensor([ids]), use_cache=True)
            prefill_logits = out.logits[0, -1].numpy().astype(np.float32)
            step = m(torch.tensor([[5]]), past_key_values=out.past_key_values)
            decode_logits = step.logits[0, -1].numpy().astype(np.float32)
            toks = m.generate(torch.tensor([ids]), max_new_tokens=NEW, do_sample=False)[0].tolist()
        np.savez_compressed(os.path.join(OUT, f"qwen3_{name}.npz"),
                            prompt=np.array(ids, dtype=np.int64), prefill_logits=prefill_logits,
                            decode_token=np.array([5]), decode_logits=decode_logits,
                            greedy_tokens=np.array(toks, dtype=np.int64), seed=np.array([0]))
        print(name, "prefill max|logit|", float(np.abs(prefill_logits).max()), "tokens", toks[PROMPT_LEN:])


if __name__ == "__main__":
    main()
