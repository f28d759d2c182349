#!/usr/bin/env python3
"""Headline benchmark: greedy decode tokens/s, Qwen3-8B bf16, context 1024 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--model qwen3-8b|qwen3-0.6b|qwen3.5-0.8b|qwen3.8-27b]

One "step" = one decode token through the whole model (all layers + lm_head + arg-max) with weights and KV cache
resident in HBM.  N = 1: TP=1 on one MI355X.  N > 1: tensor parallel TP=N over RCCL/xGMI (attention heads + MLP columns
sharded, 2 all-reduces per layer); total work is fixed, so `scaling` is "strong".  The N ranks are one process per GPU:
when this script is started WITHOUT a torch.distributed environment (`python bench.py --gpus N`) it re-launches itself
under `python -m torch.distributed.run --nproc-per-node N`; started by torchrun it uses the ranks it was given.  Every
rank creates its shard with tp_size = N, rank 0 creates the RCCL id, and the run aborts unless the library reports an
N-rank communicator -- a TP=1 number can never be labelled as N GPUs.  torch is used only for the rendezvous (gloo).

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel, HIP-event timed on the model's
own stream), `roofline_step` (whole step), `parity` (first tokens / logits of this very configuration against the CPU
oracle) and `cpu_baseline` (oracle/c port on the host cores; rank 0, N = 1).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def cpu_leg(model_name: str, ctx: int, budget_s: float, long_ctx: int = 0, long_budget_s: float = 90.0):
    """oracle/c on the host cores: greedy decode of the same workload (KV filled with the device's synthetic values).
    Returns (cpu_baseline dict, reference tokens, logits of the first step, model-written-cache reference) -- the checker
    side of `parity`."""
    so = os.path.join(ROOT, "oracle", "c", "libqwen3_cpu.so")
    if not os.path.exists(so):
        return None, None, None, None
    from oracle.c_oracle import time_decode
    return time_decode(model_name, ctx, budget_s, long_ctx=long_ctx, long_budget_s=long_budget_s)


MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak of the MI355X (MI355X_MICROARCH.md; the 2:1-sparsity figure is not used)


def vision_leg(m, cfg, model_name: str, cpu: bool):
    """BASELINE configs[3] (image + text): one 448 x 448 RGB image -> cm_image_preprocess (784 patches) -> the vision tower
    (+ DeepStack mergers) -> 196 merged tokens -> image+text prefill -> decode.  The tower is the MFMA-bound stage of the path
    (`roofline_tower`: USEFUL flops of the tower's GEMMs and attention / time; the parity mode multiplies bf16 hi + lo
    activation planes, i.e. issues twice that on the matrix cores); decode after the image is the HBM-bound text model the
    main line measures.  cpu_baseline: the numpy tower oracle (oracle/qwen3_5_vision_oracle.py, f32) timed on a SMALL image
    (64 x 64 = 16 patches, labelled) on the host cores, and compared with the HIP tower on that image in the same run."""
    import time
    import numpy as np
    from crane_amd.processor import PreprocessorConfig
    vc = cfg["vision_config"]
    rng = np.random.default_rng(0)
    image = rng.integers(0, 256, size=(448, 448, 3), dtype=np.uint8)
    t0 = time.perf_counter()
    pix, g = PreprocessorConfig().process(image)
    t_pp = time.perf_counter() - t0
    grid = [list(g)]
    npatch = int(pix.shape[0])
    feat = m.encode_images(pix, grid)                                   # warm-up (allocates the tower scratch)
    reps = 8
    t0 = time.perf_counter()
    dev_ms = []
    for _ in range(reps):
        feat = m.encode_images(pix, grid)                               # synchronous: pixels go up (4.8 MB), features come back (1.6 MB)
        dev_ms.append(float(m.debug_read("vision_ms", 1)[0]))           # HIP events around the tower's kernels (inputs resident in HBM)
    t_wall = (time.perf_counter() - t0) / reps
    t_enc = sum(dev_ms) / len(dev_ms) * 1e-3
    Hv, Iv, L = vc["hidden_size"], vc["intermediate_size"], vc["depth"]
    flops = npatch * L * 2 * (4 * Hv * Hv + 2 * Hv * Iv) + L * 4 * npatch * npatch * Hv
    tf = flops / t_enc / 1e12
    img = cfg["image_token_id"]
    ids = [5, 6, 7, cfg["vision_start_token_id"]] + [img] * int(feat.shape[0]) + [cfg["vision_end_token_id"], 8, 9, 10]
    m.clear_kv_cache(); m.vlm_forward(ids, pix, grid); m.clear_kv_cache()
    t0 = time.perf_counter()
    _, nxt = m.vlm_forward(ids, pix, grid)
    t_pre = time.perf_counter() - t0
    toks, pos = [nxt], len(ids)
    t0 = time.perf_counter()
    for _ in range(16):
        toks.append(m.forward_step_greedy([toks[-1]], pos)); pos += 1
    t_dec = (time.perf_counter() - t0) / 16
    out = {"image": "448x448 RGB, synthetic", "patches": npatch, "merged_tokens": int(feat.shape[0]),
           "preprocess_ms": round(t_pp * 1e3, 3), "tower_ms": round(t_enc * 1e3, 3),
           "tower_ms_host_buffers": round(t_wall * 1e3, 3),              # cm_vision_encode as called: + PCIe both ways + host index tables
           "roofline_tower": {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "flops": int(flops),
                              "note": "useful flops (GEMMs + attention of the tower); bf16 hi + lo activations issue 2 MFMAs per product"},
           "image_text_prefill_ms": round(t_pre * 1e3, 3), "prompt_tokens": len(ids),
           "decode_after_image_ms_per_token": round(t_dec * 1e3, 3)}
    if cpu:
        try:
            from crane_amd import synth
            from oracle.qwen3_vl_oracle import DeepstackVisionOracle
            t0 = time.perf_counter()
            w = {}
            for name, shape, std, off in synth.specs_for(cfg):
                if name.startswith("model.visual."):
                    w[name] = synth.bf16_bits_to_f32(synth.synth_bf16_bits(name, int(np.prod(shape)), 0, std, off)).reshape(shape)
            t_build = time.perf_counter() - t0
            o = DeepstackVisionOracle(vc, w)
            small = rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)
            spix, sg = PreprocessorConfig().process(small)
            t0 = time.perf_counter()
            ref, _ = o.forward_with_deepstack(spix, [list(sg)])
            t_cpu = time.perf_counter() - t0
            got = m.encode_images(spix, [list(sg)])
            rel = float(np.abs(got - ref).max() / np.abs(ref).max())
            try:
                from threadpoolctl import threadpool_info
                cores = max([int(i.get("num_threads", 1)) for i in threadpool_info()] or [1])       # numpy's BLAS pool
            except Exception:
                cores = 1
            out["cpu_baseline"] = {"value": round(spix.shape[0] / t_cpu, 2), "unit": "patches/s", "cores": cores,
                                   "kind": "port", "sample": f"numpy f32 tower oracle on a 64 x 64 image (smart_resize -> {spix.shape[0]} patches, "
                                   f"{t_cpu:.2f}s; numpy BLAS threads; weight synthesis {t_build:.1f}s excluded); the HIP tower does "
                                   f"{npatch / t_enc:.0f} patches/s at 784 patches"}
            out["parity"] = {"reference": "oracle/qwen3_vl_oracle.py DeepstackVisionOracle (f32) on the same synthetic weights, 64 x 64 image",
                             "feature_rel": float(f"{rel:.3e}"), "ok": bool(rel < 1e-3)}
        except Exception as e:
            out["cpu_baseline"] = {"error": str(e)}
    return out


def relaunch_under_torchrun(n: int) -> int:
    """`python bench.py --gpus N` (no torchrun): start the N ranks ourselves, one process per GPU."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        print(f"bench.py: --gpus {n} but this node exposes {have} GPU(s); refusing to report a {n}-GPU number",
              file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--model", default="qwen3-8b")
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU decode steps for cpu_baseline / parity")
    ap.add_argument("--parity-ctx", type=int, default=-1, help="parity on a model-written cache of this many tokens (default: the "
                    "benchmark context; 0 = only the 48-token case)")
    ap.add_argument("--parity-budget", type=float, default=90.0, help="seconds of host time the long-context parity leg may take "
                    "(the token-serial hybrid CPU port skips it beyond this)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--engine", type=int, default=0, choices=[-1, 0, 1], help="persistent chain kernel: 1 require, -1 off, 0 library default")
    ap.add_argument("--kv", default="f16", choices=["f16", "bf16", "f32", "int8", "int4"],
                    help="KV page element type (f16 = library default and headline: 2 bytes per element like bf16, inside the 1e-3 parity bar)")
    ap.add_argument("--tp-local", type=int, default=0, metavar="N", help="time ONE rank's shard of a TP=N model on one GPU: the C++ loader "
                    "slices rank --rank of N, every collective is a local no-op (CM_DEBUG_TP_LOCAL).  The line is labelled "
                    "emulated_shard r/N, rccl_ranks 1: the compute half of the 1->N scaling curve, NOT an N-GPU number")
    ap.add_argument("--rank", type=int, default=0, help="which rank's shard --tp-local times")
    ap.add_argument("--tp-inprocess", type=int, default=0, metavar="N", help="ONE handle that owns all N ranks (cm_opts.tp_mode = "
                    "CM_TP_IN_PROCESS: library worker threads, what a single crane-serve ModelBackend can host).  With N visible "
                    "GPUs: one rank per GPU (a real N-GPU number, n_gpus = N).  With fewer: every rank on GPU 0 (emulated_group: "
                    "functional -- parity of the sharded model against the CPU oracle -- the time is NOT an N-GPU number)")
    ap.add_argument("--tp-collective", default=None, choices=["rccl", "peer"], help="exchange steps of --tp-inprocess on distinct "
                    "devices: RCCL (default) or the peer-store kernels")
    ap.add_argument("--force-rccl", action="store_true", help="TP=1 with every reduction routed through a 1-rank RCCL communicator "
                    "(CM_DEBUG_FORCE_RCCL): per-call enqueue cost of the collectives, one GPU")
    ap.add_argument("--isq", default=None, help="in-situ weight quantisation (q8_0): a DIFFERENT workload than the bf16 headline")
    args = ap.parse_args()

    if args.tp_inprocess:      # ranks that share a device need one hardware queue per rank stream: before the HIP runtime starts
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    n = args.gpus
    if n < 1:
        raise SystemExit("--gpus must be >= 1")
    if n > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(n))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n:
        raise SystemExit(f"--gpus {n} but WORLD_SIZE={world}: one rank per GPU is required")

    import numpy as np
    import torch
    from crane_amd import configs
    from crane_amd.backend import Model

    dist = None
    uid = None
    if world > 1:
        import torch.distributed as dist
        if torch.cuda.device_count() < world:
            raise SystemExit(f"--gpus {n} but only {torch.cuda.device_count()} visible device(s)")
        torch.cuda.set_device(local_rank)       # torch.cuda.synchronize() below then waits on THIS rank's device
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
    tpl = args.tp_local
    if tpl and (n != 1 or not (0 <= args.rank < tpl)):
        raise SystemExit("--tp-local N needs --gpus 1 and 0 <= --rank < N")
    tpg, emulated = args.tp_inprocess, False
    if tpg:
        if world != 1 or tpl or args.force_rccl:
            raise SystemExit("--tp-inprocess is a single-process mode (no torchrun, no --tp-local / --force-rccl)")
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        emulated = have < tpg
        devs = [0] * tpg if emulated else list(range(tpg))
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        mk = lambda coll: Model.synthetic(cfg, seed=0, max_seq_len=max(2048, ctx + K + W + 64), max_seqs=1, use_graph=-1 if args.no_graph else 0,
                                          tp_size=tpg, tp_in_process=True, tp_devices=devs, tp_collective=coll, kv_dtype=args.kv)
        tp_fallback = None
        try:
            m = mk(args.tp_collective)
        except Exception as e:
            # RCCL inside ONE process (a communicator rank per library thread, collectives captured into one hipGraph per rank) has
            # never run on more than one rank in this repository: if the group cannot be created on distinct devices, say why and
            # take the library's own peer-store exchange (kernels_tp.hip) instead of failing the run
            if emulated or args.tp_collective == "peer":
                raise
            tp_fallback = f"RCCL in-process group failed ({type(e).__name__}: {str(e)[:200]}); fell back to --tp-collective peer"
            print("bench.py: " + tp_fallback, file=sys.stderr, flush=True)
            m = mk("peer")
        if m.tp_ranks() != tpg:
            raise SystemExit(f"library reports {m.tp_ranks()} rank(s), expected {tpg}")
        n = tpg if not emulated else 1
    m = m if tpg else Model.synthetic(cfg, seed=0, device=local_rank if world > 1 else 0,
                        max_seq_len=max(2048, ctx + K + W + 64), max_seqs=1,
                        use_graph=-1 if args.no_graph else 0, engine=args.engine,
                        tp_rank=args.rank if tpl else rank, tp_size=tpl if tpl else world, tp_unique_id=uid, isq=args.isq,
                        kv_dtype=args.kv, debug_tp_local=bool(tpl), debug_force_rccl=args.force_rccl)
    ranks = m.tp_ranks()
    if tpg:
        pass
    elif tpl:
        if ranks != 0:
            raise SystemExit(f"--tp-local: expected no communicator, library reports {ranks} rank(s)")
        args.no_cpu_baseline = True          # a single rank's partial sums have no CPU counterpart (tests/test_gpu_tp_shards.py checks them)
    elif ranks != n:
        raise SystemExit(f"library reports an RCCL communicator of {ranks} rank(s), expected {n}")
    m.debug_fill_kv(ctx, seed=1)            # synthetic KV for positions [0, ctx): inputs resident in HBM
    first = 3
    if W > 0:
        toks, _ = m.bench_decode(first, W)
        first = int(toks[-1])

    def barrier():
        if dist is not None:
            dist.barrier()

    def dev_sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    barrier()
    dev_sync()
    t0 = time.perf_counter()
    toks, ev_ms = m.bench_decode(first, K)   # enqueues K steps, synchronises on the model's stream
    dev_sync()
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

    # dominant kernel, HIP-event timed on the model's stream over launches that cycle through the layers' weights:
    # per-projection launches -> RMSNorm + gate||up GEMV + SiLU*mul (2/3 of the weight bytes); persistent path -> the
    # chain launch itself (o_proj + gate||up + down_proj + next QKV: every weight byte of a layer)
    roof = None
    dom = "chain" if m.engine_active() else "gate_up"
    if tpl or args.force_rccl or tpg:
        dom = None                              # the shard line reports the whole step only
    pmc_key = "engine_kernel" if m.engine_active() else "gemv_bf16_kernel<1, 2,"
    try:
        if dom is None:
            raise RuntimeError("not collected for --tp-local / --force-rccl (roofline_step is the figure)")
        kb = m.bench_kernel(dom, 360)
        roof = {"bound": "hbm", "kernel": kb["kernel"], "achieved": round(kb["bytes"] / (kb["ms"] * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                "bytes_per_launch": kb["bytes"], "us_per_launch": round(kb["ms"] * 1e3, 3)}
        roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBS, 4)
        # PMC traffic cannot be collected inside this process: it comes from the committed rocprofv3 --pmc passes
        # (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE), per launch
        try:
            if not args.isq and n == 1:
                san = args.model.replace("-", "_").replace(".", "_")
                # newest first; re-collected whenever kernels_engine.hip / the decode GEMVs change (tools/gpu_round.sh traffic)
                # (round 5: the whole-token kernel also streams the embedding row, the final norm and lm_head -- its launch is r05's)
                srcs = (f"r06_pmc_traffic_decode_{san}.json", f"r05_pmc_traffic_decode_{san}.json", f"r04_pmc_traffic_decode_{san}.json",
                        f"r03_pmc_traffic_decode_{san}.json")
                for src in srcs:
                    path = os.path.join(ROOT, "profiles", src)
                    if not os.path.exists(path):
                        continue
                    pm = json.load(open(path))
                    for r in pm["kernels"]:
                        if pmc_key in r["kernel"]:
                            roof["traffic"] = r["hbm_read_bytes_corrected"] + r["hbm_write_bytes"]
                            roof["traffic_source"] = "profiles/" + src + " -- " + pm.get("source", "")       # (names the tree it was measured on)
                    if roof["traffic"] is not None:
                        break
        except Exception:
            pass
    except Exception as e:
        roof = {"bound": "hbm", "error": str(e)}
    step_gbs = bytes_tok_rank / (ms_per_step * 1e-3) / 1e9
    roof_step = {"achieved": round(step_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(step_gbs / HBM_PEAK_GBS, 4), "bytes_per_token_per_gpu": bytes_tok_rank,
                 "event_ms_per_step": round(ev_ms / K, 4)}
    if dist is not None:                       # every rank's own step figure (the shards differ by the vocabulary tail)
        per = [None] * world
        dist.all_gather_object(per, roof_step)
        roof_step = {"rank0": roof_step, "per_rank": per}

    # prefill throughput (second half of BASELINE.json's metric): one 1024-token prompt, MFMA path
    prefill = None
    try:
        if args.isq:
            raise RuntimeError("quantised weights: not part of the bf16 headline")
        if tpl or args.force_rccl or (tpg and emulated):
            raise RuntimeError("not timed for --tp-local / --force-rccl / an emulated group")
        ids = configs.synthetic_prompt(1024, cfg.get("text_config", cfg)["vocab_size"])
        m.clear_kv_cache(); m.forward_step_greedy(ids, 0)            # warm-up (allocates chunk buffers)
        m.clear_kv_cache()
        barrier()
        tp0 = time.perf_counter(); m.forward_step_greedy(ids, 0); tp1 = time.perf_counter()
        prefill = {"tokens": 1024, "ms": round((tp1 - tp0) * 1e3, 3), "tokens_per_s": round(1024 / (tp1 - tp0), 1),
                   "activations": "bf16x2 split (parity mode)"}
        # the same prompt with plain bf16 activations (one MFMA per product: the arithmetic of a bf16 GPU forward of the
        # reference; logits move by ~5e-3, outside the 1e-3 bar, so it is an option, not the default)
        m.debug_set("prefill_split", 1)
        try:
            m.clear_kv_cache(); m.forward_step_greedy(ids, 0); m.clear_kv_cache()
            barrier()
            tp0 = time.perf_counter(); m.forward_step_greedy(ids, 0); tp1 = time.perf_counter()
        finally:
            m.debug_set("prefill_split", -1)                         # back to cm_opts.prefill_split
        prefill["plain_bf16"] = {"ms": round((tp1 - tp0) * 1e3, 3), "tokens_per_s": round(1024 / (tp1 - tp0), 1)}
    except Exception as e:
        prefill = {"error": str(e)}

    # CPU leg (rank 0, one GPU): baseline + parity of THIS configuration (same synthetic weights, same synthetic KV)
    cpu, parity = None, None
    if rank == 0 and (n == 1 or tpg) and not args.no_cpu_baseline:
        # oracle/c holds the model as bf16 on the host: a model that would not leave half of the host's RAM free is not timed
        import psutil
        need = m.weight_bytes() * 1.05
        if need > 0.5 * psutil.virtual_memory().total:
            cpu = {"skipped": f"the CPU port needs {need / 2**30:.0f} GiB of host memory for this model (host: "
                              f"{psutil.virtual_memory().total / 2**30:.0f} GiB); BASELINE configs[0] / [2] are the CPU-sized cases"}
        else:
            try:
                pctx = ctx if args.parity_ctx < 0 else args.parity_ctx
                cpu, ref_toks, ref_logits, wr = cpu_leg(args.model, ctx, args.cpu_budget, pctx, args.parity_budget)
                if ref_toks:
                    def relerr(a, b):
                        return float(np.abs(a - b).max() / np.abs(b).max())
                    # (1) the parity figure: a cache the MODEL wrote.  Prompt through the MFMA prefill, one decode step and the
                    # greedy continuation through the decode kernels over the pages that prefill wrote, in the benchmarked KV mode
                    m.clear_kv_cache()
                    pl = m.forward_step(wr["prompt"], 0)[0, 0]
                    dl = m.forward_step([wr["greedy"][0]], len(wr["prompt"]))[0, 0]
                    from crane_amd.backend import GenerationConfig
                    gen = m.generate(wr["prompt"], GenerationConfig.greedy(len(wr["greedy"])))[len(wr["prompt"]):]
                    eqw = 0
                    while eqw < len(gen) and gen[eqw] == wr["greedy"][eqw]:
                        eqw += 1
                    rp, rd = relerr(pl, wr["prefill_logits"]), relerr(dl, wr["decode_logits"])
                    # (1b) the same at the benchmark's own context: every layer, a `pctx`-token prompt through the HIP prefill in
                    # the benchmarked KV mode, one decode step + greedy ids over the pages it wrote (f16 K/V rounding grows with
                    # depth and context: this is the headline configuration itself, not an extrapolation from 48 tokens)
                    long_par, rl = None, 0.0
                    wl = wr.get("long")
                    if wl and "skipped" in wl:
                        long_par = wl
                    elif wl:
                        m.clear_kv_cache()
                        pl2 = m.forward_step(wl["prompt"], 0)[0, 0]
                        dl2 = m.forward_step([wl["greedy"][0]], len(wl["prompt"]))[0, 0]
                        gen2 = m.generate(wl["prompt"], GenerationConfig.greedy(len(wl["greedy"])))[len(wl["prompt"]):]
                        eql = 0
                        while eql < len(gen2) and gen2[eql] == wl["greedy"][eql]:
                            eql += 1
                        rp2, rd2 = relerr(pl2, wl["prefill_logits"]), relerr(dl2, wl["decode_logits"])
                        rl = max(rp2, rd2)
                        long_par = {"prompt_tokens": len(wl["prompt"]), "layers": int(cfg.get("text_config", cfg)["num_hidden_layers"]),
                                    "prefill_logit_rel": float(f"{rp2:.3e}"), "decode_logit_rel": float(f"{rd2:.3e}"),
                                    "greedy_checked": len(wl["greedy"]), "greedy_equal": eql, "cpu_seconds": wl["cpu_seconds"],
                                    "ok": bool(eql == len(wl["greedy"]) and rl < 1e-3)}
                    # (2) the timed configuration itself (synthetic KV at the benchmark context: bf16-exact fill values, so this
                    # checks the kernels at the timed shapes, not the KV rounding)
                    m.debug_fill_kv(ctx, seed=1)
                    got0 = m.forward_step([3], ctx)[0, 0]
                    lrel = relerr(got0, ref_logits)
                    m.debug_fill_kv(ctx, seed=1)
                    gt, _ = m.bench_decode(3, len(ref_toks))
                    gt = [int(t) for t in gt]
                    eq = 0
                    while eq < len(ref_toks) and gt[eq] == ref_toks[eq]:
                        eq += 1
                    parity = {"reference": "oracle/c: the f32 CPU forward (K/V appends unrounded) on identical synthetic weights",
                              "kv_pages": args.kv,
                              "model_written_cache": {"prompt_tokens": len(wr["prompt"]), "prefill_logit_rel": float(f"{rp:.3e}"),
                                                      "decode_logit_rel": float(f"{rd:.3e}"), "greedy_checked": len(wr["greedy"]),
                                                      "greedy_equal": eqw},
                              "timed_configuration": {"note": "synthetic KV of the benchmark context (bf16-exact fill values)",
                                                      "tokens_checked": len(ref_toks), "tokens_equal": eq,
                                                      "logit_rel": float(f"{lrel:.3e}")},
                              "logit_rel": float(f"{max(rp, rd, rl):.3e}"),
                              "ok": bool(eqw == len(wr["greedy"]) and eq == len(ref_toks) and max(rp, rd, lrel, rl) < 1e-3
                                         and (long_par is None or long_par.get("ok", True)))}
                    if long_par is not None:
                        parity[f"model_written_cache_ctx{pctx}"] = long_par
            except Exception as e:  # the baseline must never break the headline number
                cpu = {"error": str(e)}

    vision = None
    if rank == 0 and n == 1 and "vision_config" in cfg:
        try:
            vision = vision_leg(m, cfg, args.model, not args.no_cpu_baseline)
        except Exception as e:
            vision = {"error": str(e)}

    wdt = args.isq or "bf16"
    if rank == 0:
        headline = args.model == "qwen3-8b" and not args.isq and args.kv == "f16" and ctx == 1024
        line = {
            "metric": "decode tokens/s Qwen3-8B bf16 greedy, ctx 1024" if headline
                      else f"decode tokens/s {args.model} {wdt} weights / {args.kv} KV greedy, ctx {ctx}",
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": n, "steps": K, "warmup": W,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if n > 1 else "weak", "vs_baseline": None,
            "dtype": "bf16" if not args.isq else f"{args.isq} weights, f32 accumulate", "data": "synthetic",
            "config": {"workload": f"{args.model} greedy decode, batch 1, context {ctx} (+{W}+{K} generated), "
                                   f"{wdt} weights + {args.kv} paged KV, f32 activations",
                       "parallelism": f"tp{tpl} (ONE rank's shard, collectives = local no-ops)" if tpl else f"tp{n}",
                       "rccl_ranks": 1 if (tpl or args.force_rccl) else ranks, "graph": not args.no_graph,
                       "decode_path": {0: "per-projection launches", 1: "persistent kernel per layer + attention launches",
                                       2: "persistent decode kernel: one launch per token (cm_opts.engine)"}[m.engine_active()]},
            "roofline": roof, "roofline_step": roof_step, "prefill": prefill, "parity": parity, "cpu_baseline": cpu,
        }
        if vision is not None:
            line["vision"] = vision
        if tpl:
            L, H = cfg.get("text_config", cfg)["num_hidden_layers"], cfg.get("text_config", cfg)["hidden_size"]
            line["metric"] += f" -- rank {args.rank} of TP={tpl} ALONE (emulated shard, no collectives)"
            line["config"]["emulated_shard"] = f"{args.rank}/{tpl}"
            line["config"]["not_an_n_gpu_number"] = ("per-rank compute time of the shard; the 2 x L all-reduces of 4 H bytes and the "
                                                     "arg-max gather of a real TP step are NOT in it")
            line["collectives_skipped"] = {"all_reduce_per_token": 2 * L, "bytes_per_all_reduce": 4 * H, "all_gather_per_token": 1}
        if args.force_rccl:
            line["metric"] += " -- reductions through a 1-rank RCCL communicator (CM_DEBUG_FORCE_RCCL)"
            line["config"]["force_rccl"] = True
        if tpg:
            line["config"]["tp_host"] = "ONE handle, cm_opts.tp_mode = CM_TP_IN_PROCESS (library worker threads)"
            line["config"]["tp_collective"] = ("peer-store kernels (kernels_tp.hip)" if emulated or args.tp_collective == "peer" or tp_fallback else "RCCL")
            if tp_fallback:
                line["config"]["tp_collective_fallback"] = tp_fallback
            line["config"]["parallelism"] = f"tp{tpg}"
            if emulated:
                line["metric"] += f" -- TP={tpg} group with EVERY rank on one GPU (emulated: functional check, not an N-GPU time)"
                line["config"]["emulated_group"] = f"{tpg} ranks on 1 GPU"
                line["scaling"] = "weak"
        if n > 1:
            # the exchange steps of one token under TP: one f32 [H] all-reduce behind each row-parallel projection
            # (o_proj / GDN out_proj, down_proj), one all-gather of the ranks' (max, index) arg-max partials behind lm_head
            L, H = cfg.get("text_config", cfg)["num_hidden_layers"], cfg.get("text_config", cfg)["hidden_size"]
            line["collectives"] = {"library": "RCCL", "all_reduce_per_token": 2 * L, "bytes_per_all_reduce": 4 * H,
                                   "all_reduce_bytes_per_token": 8 * L * H, "all_gather_per_token": 1,
                                   "bytes_per_all_gather": 8 * n, "captured_in_hipgraph": not args.no_graph,
                                   "measured": "per-rank roofline_step above; no multi-GPU box was available to the build, "
                                               "the first real > 1-rank run is the driver's"}
        print(json.dumps(line), flush=True)
    m.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
