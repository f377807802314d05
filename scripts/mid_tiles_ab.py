"""A/B of the six-wave mid-size GEMM tiles (mi355_tune_set key 30; csrc/gemm.hip: launch_epi): back-to-back launches of the gated-residual operator at
the shapes of the reference's 512^2 examples (4096 image rows; N = 1536; K = 1536 / 6144), bit identity asserted, TFLOP/s per setting.

    python scripts/mid_tiles_ab.py"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "flow-factory_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from mi355_flow import _lib, engine  # noqa: E402

lib = _lib.load()


def timeit(fn, n=30):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


print(f"{'shape (M, N, K)':24s} {'128x128':>10s} {'128x192':>10s}   us / launch and TFLOP/s (median of 3 interleaved rounds)")
for (M, N, K) in [(4096, 1536, 1536), (4096, 1536, 6144), (4096, 3072, 3072), (2048, 3072, 3072), (3072, 1536, 6144)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda")
    gate = torch.randn(4, N, device="cuda").bfloat16() * 0.01
    xres = torch.randn(M, N, device="cuda").bfloat16()
    lib.mi355_tune_set(30, 0); r0 = engine.op_linear_gate_res(xres.clone(), x, w, b, gate, (M + 3) // 4)
    lib.mi355_tune_set(30, 1); r1 = engine.op_linear_gate_res(xres.clone(), x, w, b, gate, (M + 3) // 4)
    same = torch.equal(r0, r1)
    fl = 2.0 * M * N * K
    t = {0: [], 1: []}
    for _ in range(3):
        for v in (0, 1):
            lib.mi355_tune_set(30, v)
            t[v].append(timeit(lambda: engine.op_linear_gate_res(xres, x, w, b, gate, (M + 3) // 4)))
    m0, m1 = sorted(t[0])[1], sorted(t[1])[1]
    print(f"{str((M, N, K)):24s} {m0 * 1e6:6.1f} us {fl / m0 / 1e12:6.0f} TF   {m1 * 1e6:6.1f} us {fl / m1 / 1e12:6.0f} TF   bit-identical {same}", flush=True)
lib.mi355_tune_set(30, 0)
