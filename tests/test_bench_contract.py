"""bench.py's launcher contract, the part that can be checked without a GPU: `python bench.py --gpus N` on a node with fewer
than N devices must exit non-zero instead of printing a line that claims N GPUs (round-1 review: it used to report TP=1 as
N GPUs)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_more_gpus_than_the_node_has():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have >= 8:
        pytest.skip("node has 8 GPUs")
    want = 8 if have >= 2 else 2
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0
    assert "refusing" in p.stderr and '"metric"' not in p.stdout
