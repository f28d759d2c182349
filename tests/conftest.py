import os
import sys

import pytest

# in-process tensor-parallel groups with every rank on ONE device (tests/test_gpu_tp_group.py): one hardware queue per rank
# stream, so that a rank's waiting exchange kernel never sits in front of another rank's kernels (ROCm default: 4 queues per
# process and device).  Must be in the environment before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
