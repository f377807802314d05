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
