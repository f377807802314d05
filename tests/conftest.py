"""pytest config: registers the `gpu` marker and puts the package dir on sys.path.

`-m "not gpu"` tests: oracle vs golden fixtures, host-side mirrors, C-ABI symbol check,
2-rank gloo tests.  `-m gpu` tests: the HIP path (through the C-ABI) vs the oracle.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flow-factory_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _seed_the_global_generators():
    """This torch build seeds its default generators from entropy at start-up (`torch.initial_seed()` differs per process), and the reference's
    rollout draws its noise on the GLOBAL generator (`generator=None`): a test that does not seed it itself ran on different noise every time --
    the GRPO-Guard trainer epoch failed one run in ~13 on `kl_div > 0` (an update too small for bf16 to see).  Every test now starts from the
    same state of torch's, numpy's and Python's global generators."""
    import random
    import numpy as np
    import torch
    random.seed(1234)
    np.random.seed(1234)
    torch.manual_seed(1234)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(1234)
    yield
