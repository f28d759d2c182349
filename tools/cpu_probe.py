"""Host-side probe for the GPU box: what the CPU looks like and how oracle/c scales with OMP threads (sizes the
cpu_baseline leg of bench.py).  Test infrastructure; not part of the product path."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from crane_amd import configs
    from oracle import c_oracle
    name = sys.argv[2]
    cfg = configs.get_config(name)
    t0 = time.time()
    c = c_oracle.CQwen3(cfg, seed=0, max_seq=1100)
    tb = time.time() - t0
    c.fill_kv_paged(1024, 1, 64)
    c.forward([3], 1024)
    t0 = time.time()
    for i in range(4):
        c.forward([3], 1025 + i)
    td = (time.time() - t0) / 4
    ids = configs.synthetic_prompt(32, cfg["vocab_size"])
    t0 = time.time()
    c.forward(ids, 0)
    tp = (time.time() - t0) / 32
    print(f"{name} threads={c.threads()} build={tb:.2f}s decode_step={td*1e3:.1f}ms prefill_tok={tp*1e3:.1f}ms", flush=True)
    sys.exit(0)

print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Core|Socket|NUMA'", shell=True, capture_output=True, text=True).stdout)
print(subprocess.run("free -g | head -2", shell=True, capture_output=True, text=True).stdout)
for thr in (8, 16, 32, 64, 128):
    for extra in ({}, {"OMP_PROC_BIND": "spread", "OMP_WAIT_POLICY": "passive"}):
        env = dict(os.environ, OMP_NUM_THREADS=str(thr), **extra)
        t0 = time.time()
        r = subprocess.run([sys.executable, __file__, "child", "qwen3-8b-2l"], env=env, capture_output=True, text=True, timeout=300)
        print(thr, extra, r.stdout.strip(), r.stderr.strip()[-200:], f"wall={time.time()-t0:.1f}s", flush=True)
