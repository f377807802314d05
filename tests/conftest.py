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
    config.addinivalue_line("markers", "gpu_next: GPU check written without GPU time at hand; collected only with MI355_NEXT=1 "
                                       "(tests/test_gpu_next_round.py)")


def pytest_collection_modifyitems(config, items):
    # checks that have never run on an MI355X stay out of BOTH the `-m gpu` and the `-m "not gpu"` runs (deselected, not skipped: a skip
    # would read as hidden coverage) until MI355_NEXT=1 asks for them
    if os.environ.get("MI355_NEXT") == "1":
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("gpu_next") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep
