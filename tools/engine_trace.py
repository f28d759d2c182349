"""Timeline of ONE persistent chain launch (cm_debug_read "engine_trace"): where the ~70 us of a Qwen3-8B layer go.
Prints, per phase, when the stream waves arrive at / pass the input wait and finish their rows, and when the comm waves
start polling and finish staging (microseconds since the earliest stamp of the launch; min / median / max over the chip)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crane_amd import configs
from crane_amd.backend import Model

name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-8b"
cfg = configs.get_config(name)
if len(sys.argv) > 2:
    cfg["num_hidden_layers"] = int(sys.argv[2])
TPL = int(os.environ.get("TRACE_TP_LOCAL", "0"))       # ONE rank's shard of a TP = N model (collectives = local no-ops)
kw = dict(tp_size=TPL, tp_rank=0, debug_tp_local=True) if TPL else {}
m = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=1, engine=1, **kw)
m.debug_fill_kv(1024, seed=1)
m.bench_decode(3, 8)
NSW = int(os.environ.get('CM_ENG_CFG', '4,4').split(',')[0])
NB, NW, MAXPH, NEV = 256, NSW + 4, 8, 8
for rep in range(2):
    t = m.debug_read("engine_trace", NB * NW * MAXPH * NEV).reshape(NB, NW, MAXPH, NEV)
    print(f"--- traced launch {rep} (CM_ENG_CFG={os.environ.get('CM_ENG_CFG', 'default')}) ---")
    ent = t[:, :NSW, 0, 3]
    print(f"kernel entry (stream waves): min {ent.min():.2f} med {np.median(ent):.2f} max {ent.max():.2f}")
    full = os.environ.get("CM_ENGINE_FULL", "1") != "0"
    names = ["qkv", "o_proj", "gate_up", "down"] * 2 if full else ["o_proj", "gate_up", "down", "qkv_next", "-", "-", "-", "-"]
    def st(a):
        a = a[a > 0]
        return f"{a.min():7.2f} {np.median(a):7.2f} {a.max():7.2f}" if a.size else "   -"
    for p in range(MAXPH):
        s, c = t[:, :NSW, p, :], t[:, NSW:, p, :]
        print(f"phase {p} {names[p]:9s} stream: arrive[{st(s[..., 0])}] go[{st(s[..., 1])}] done[{st(s[..., 2])}]")
        print(f"                  comm  : begin [{st(c[..., 0])}] poll[{st(c[..., 1])}] attn[{st(c[..., 2])}] stale sweeps avg {c[..., 3].mean():.1f} max {c[..., 3].max():.0f}")
        if (c[..., 4] > 0).any():
            sp = c[..., 7].astype(np.int64)
            print(f"                  attn  : q+rope[{st(c[..., 4])}] old-token scores[{st(c[..., 5])}] k/v + gathered[{st(c[..., 6])}]  failed polls: q avg {(sp & 0xFFFF).mean():.2f} max {(sp & 0xFFFF).max()}, gather avg {(sp >> 16).mean():.2f} max {(sp >> 16).max()}")
    if os.environ.get("TRACE_XCD"):
        # stream-wave finish time of every phase by XCD (workgroup b runs on XCD b % 8) and by position inside the XCD
        for p in range(MAXPH):
            d = t[:, :NSW, p, 2]
            byx = [np.median(d[x::8][d[x::8] > 0]) for x in range(8)]
            print(f"phase {p} done by XCD: " + " ".join(f"{v:7.2f}" for v in byx) + f"   spread of CU medians {np.ptp(np.median(d, axis=1)):.2f}")
            c0 = t[:, NSW, p, :]                     # comm wave 0 of every workgroup: the attention stamps
            if (c0[:, 2] > 0).any() and (c0[:, 4] > 0).any():
                for ev, nm in ((4, "qkv+rope"), (5, "scores"), (6, "gathered"), (2, "published")):
                    v = c0[:, ev]
                    print(f"    attn {nm:9s} by XCD (= kv head): " + " ".join(f"{np.median(v[x::8]):7.2f}" for x in range(8)) +
                          "   by split 0-9 / 10-31: " + f"{np.median(v[:80]):7.2f} {np.median(v[80:]):7.2f}   max {v.max():7.2f} at wg {int(v.argmax())}")
    print(f"end of launch: {t[..., :3].max():.2f} us")
m.close()
