#!/usr/bin/env python3
"""Headline benchmark: greedy decode tokens/s, Qwen3-8B bf16, context 1024 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one decode token through the whole model (36 layers + lm_head + arg-max) with
weights and KV cache resident in HBM.  N = 1: TP=1 on one MI355X.  N > 1 (launched by
`python -m torch.distributed.run --nproc-per-node N ...`): tensor parallel TP=N over RCCL/xGMI
(attention heads + MLP columns sharded, 2 all-reduces per layer) -- total work is fixed, so
`scaling` is "strong".  torch is used only for the rendezvous (gloo) around the timed region.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel,
HIP-event timed on the model's own stream) and `cpu_baseline` (oracle/c port on host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def cpu_baseline(model_name: str, ctx: int, budget_s: float = 20.0):
    """Time the C port of the reference's CPU decode path (oracle/c) on the host cores."""
    so = os.path.join(ROOT, "oracle", "c", "libqwen3_cpu.so")
    if not os.path.exists(so):
        return None
    try:
        from oracle.c_oracle import time_decode
        return time_decode(model_name, ctx, budget_s)
    except Exception as e:  # baseline must never break the headline number
        return {"error": str(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--model", default="qwen3-8b")
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--kv", default="bf16", choices=["bf16", "f32", "int8", "int4"], help="KV cache element type (bf16 = headline)")
    ap.add_argument("--isq", default=None, help="in-situ weight quantisation (q8_0): a DIFFERENT workload than the bf16 headline")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.gpus
    if world != n and world != 1:
        raise SystemExit(f"--gpus {n} but WORLD_SIZE={world}")

    from crane_amd import configs
    from crane_amd.backend import Model

    dist = None
    uid = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from crane_amd import _lib
        lib = _lib.load()
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            rc = lib.cm_tp_unique_id(buf)
            if rc != 0:
                raise SystemExit("cm_tp_unique_id failed: " + lib.cm_last_global_error().decode())
        t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        dist.broadcast(t, src=0)
        uid = bytes(t.numpy().tobytes())

    cfg = configs.get_config(args.model)
    K, W, ctx = args.steps, args.warmup, args.ctx
    import torch
    ndev = max(1, torch.cuda.device_count())
    m = Model.synthetic(cfg, seed=0, device=(local_rank % ndev) if world > 1 else 0,
                        max_seq_len=max(2048, ctx + K + W + 64), max_seqs=1,
                        use_graph=-1 if args.no_graph else 0,
                        tp_rank=rank if world > 1 else 0, tp_size=world, tp_unique_id=uid, isq=args.isq, kv_dtype=args.kv)
    m.debug_fill_kv(ctx, seed=1)            # synthetic KV for positions [0, ctx): inputs resident in HBM
    first = 3
    if W > 0:
        toks, _ = m.bench_decode(first, W)
        first = int(toks[-1])

    def barrier():
        if dist is not None:
            dist.barrier()

    import torch
    barrier()
    torch.cuda.synchronize() if torch.cuda.is_available() else None
    t0 = time.perf_counter()
    toks, ev_ms = m.bench_decode(first, K)   # enqueues K steps, synchronises on the model's stream
    torch.cuda.synchronize() if torch.cuda.is_available() else None
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    ms_per_step = dt * 1e3 / K
    value = K / dt
    avg_ctx = ctx + W + K / 2.0
    bytes_tok_rank = m.decode_bytes_per_token(int(avg_ctx))

    # dominant kernel: RMSNorm + gate||up GEMV + SiLU*mul (2/3 of the weight bytes), HIP-event timed
    roof = None
    try:
        kb = m.bench_kernel("gate_up", 360)
        roof = {"bound": "hbm", "kernel": kb["kernel"], "achieved": round(kb["bytes"] / (kb["ms"] * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                "bytes_per_launch": kb["bytes"], "us_per_launch": round(kb["ms"] * 1e3, 3)}
        roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBS, 4)
        # PMC traffic cannot be collected inside this process: it comes from the committed rocprofv3 --pmc passes
        # (profiles/r01_pmc_traffic_decode.json; FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE), per launch
        try:
            if args.model == "qwen3-8b" and not args.isq:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_decode.json")))
                for r in pm["kernels"]:
                    if "gemv_bf16_kernel<1, 2," in r["kernel"]:
                        roof["traffic"] = r["hbm_read_bytes_corrected"] + r["hbm_write_bytes"]
                        roof["traffic_source"] = "profiles/r01_pmc_traffic_decode.json"
        except Exception:
            pass
    except Exception as e:
        roof = {"bound": "hbm", "error": str(e)}
    step_gbs = bytes_tok_rank / (ms_per_step * 1e-3) / 1e9
    roof_step = {"achieved": round(step_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(step_gbs / HBM_PEAK_GBS, 4), "bytes_per_token_per_gpu": bytes_tok_rank,
                 "event_ms_per_step": round(ev_ms / K, 4)}

    # prefill throughput (second half of BASELINE.json's metric): one 1024-token prompt, MFMA path
    prefill = None
    try:
        import numpy as np
        if args.isq:
            raise RuntimeError("quantised weights prefill token-serially; not measured here")
        ids = configs.synthetic_prompt(1024, cfg["vocab_size"])
        m.clear_kv_cache(); m.forward_step_greedy(ids, 0)            # warm-up (allocates chunk buffers)
        m.clear_kv_cache()
        tp0 = time.perf_counter(); m.forward_step_greedy(ids, 0); tp1 = time.perf_counter()
        prefill = {"tokens": 1024, "ms": round((tp1 - tp0) * 1e3, 3), "tokens_per_s": round(1024 / (tp1 - tp0), 1),
                   "activations": "bf16x2 split (parity mode)"}
    except Exception as e:
        prefill = {"error": str(e)}

    cpu = None
    if rank == 0 and n == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.model, ctx)

    wdt = args.isq or "bf16"
    if rank == 0:
        line = {
            "metric": "decode tokens/s Qwen3-8B bf16 greedy, ctx 1024" if (args.model == "qwen3-8b" and not args.isq and args.kv == "bf16" and ctx == 1024)
                      else f"decode tokens/s {args.model} {wdt} weights / {args.kv} KV greedy, ctx {ctx}",
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": n, "steps": K, "warmup": W,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if n > 1 else "weak", "vs_baseline": None,
            "dtype": "bf16" if not args.isq else f"{args.isq} weights, f32 accumulate", "data": "synthetic",
            "config": {"workload": f"{args.model} greedy decode, batch 1, context {ctx} (+{W}+{K} generated), "
                                   f"{wdt} weights + {args.kv} paged KV, f32 activations",
                       "parallelism": f"tp{n}", "graph": not args.no_graph},
            "roofline": roof, "roofline_step": roof_step, "prefill": prefill, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    m.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
