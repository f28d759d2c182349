import sys; sys.path.insert(0,'.')
import numpy as np
from crane_amd import configs, synth
from crane_amd.backend import Model
from tests.test_gpu_kv_quant import _oracle, rel
for name in ["tiny-qwen3-untied","tiny-qwen3.5"]:
    for kv in ["int8","int4"]:
        cfg=configs.get_config(name); w=synth.synth_weights_f32(cfg,seed=0); o=_oracle(name,cfg,w,kv)
        m=Model.synthetic(cfg,seed=0,max_seq_len=256,kv_dtype=kv)
        ids=configs.synthetic_prompt(70,cfg["vocab_size"]); ref=o.forward(ids,0); got=m.forward_step(ids,0).reshape(-1)
        worst=rel(got,ref); tok=int(ref.argmax())
        for s in range(10):
            ref=o.forward([tok],70+s); got=m.forward_step([tok],70+s).reshape(-1); worst=max(worst,rel(got,ref)); tok=int(ref.argmax())
        print(name,kv,worst); m.close()
