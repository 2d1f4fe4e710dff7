import os
import sys

# Idle OpenMP workers SLEEP instead of spinning (must be set before libgomp is loaded, i.e. before the first `import torch`): by the time the
# oracle tests run, the process holds two OpenMP runtimes and two OpenBLAS pools (torch, scipy, transformers), and spinning workers of one
# starve the other on the build container's 8 cores -- the tiny-model oracle tests took 60-100 s each instead of 3 s (CPU suite 6 min vs 3).
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # The oracle side of the parity tests is eager PyTorch-CPU.  On the GPU box's host (256 hardware threads) the default
    # thread count is pathological for it (one UNet forward: 4.3 s at 32 threads, 275 s at 256 -- DESIGN.md section 6), and the
    # tiny-model tests alone went from 45 s to 8 min on a busy host: cap it.
    import torch
    torch.set_num_threads(min(32, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
