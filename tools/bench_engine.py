"""Serving-loop throughput of the continuous-batching engine (cm_engine_*): N requests, prompt P, G generated tokens,
on synthetic Qwen3-8B.  Prints generated tokens/s over the whole run (prefills included)."""
import os, sys, time
sys.path.insert(0, ".")
from crane_amd import configs
from crane_amd.backend import Model
from crane_amd.engine import GenerationParams, InferenceEngine

model = sys.argv[1] if len(sys.argv) > 1 else "qwen3-8b"
N, P, G = (int(x) for x in (sys.argv[2:5] + ["32", "128", "128"][len(sys.argv[2:5]):]))
SPC = int(sys.argv[5]) if len(sys.argv) > 5 else 1          # scheduler steps per native call (cm_engine_step_many)
MAXR = [int(x) for x in sys.argv[6].split(',')] if len(sys.argv) > 6 else [1, 8]      # max_running values to run
ISQ = sys.argv[7] if len(sys.argv) > 7 else None                                        # in-situ quantisation ("q8_0", "q4_k")
cfg = configs.get_config(model)
m = Model.synthetic(cfg, seed=0, max_seq_len=P + G + 64, max_seqs=max(MAXR) + 1, **({"isq": ISQ} if ISQ else {}))
V = cfg["vocab_size"]
MODES = [("greedy", GenerationParams.greedy(G)),
         ("server defaults (T 0.8, top_p 0.95, top_k 40, rep 1.05)", GenerationParams(max_tokens=G))]
for label, params in (MODES[:1] if ISQ or os.environ.get("BENCH_GREEDY") else MODES):     # BENCH_GREEDY=1: greedy runs only
    for max_running in MAXR:
        eng = InferenceEngine(m, max_running=max_running, seed=1, batch_prefill=os.environ.get("BENCH_BATCH_PREFILL", "1") != "0")
        for j in range(N):
            eng.submit([(7 * i + 3 + 11 * j) % V for i in range(P)], params)
        t0 = time.perf_counter()
        toks, done = eng.run_until_idle(steps_per_call=SPC)
        dt = time.perf_counter() - t0
        st = eng.stats()
        n = sum(len(v) for v in toks.values())
        print(f"{model} {label:55s} max_running={max_running} steps/call={SPC}: {n} tokens in {dt:.2f}s = {n / dt:7.1f} tok/s "
              f"(prefill steps {st['prefill_steps']}, decode rounds {st['decode_rounds']}, preemptions {st['preemptions']})", flush=True)
        eng.close()
m.close()
