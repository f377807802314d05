"""A/B of the 256x256 GEMM schedules: bitwise agreement (same k-order => identical results; a
staging race shows up as a mismatch) over repeated runs, then timing."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import engine, _lib
lib = _lib.load()

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

bad = 0
for (M, N, K) in [(32768, 1536, 1536), (32768, 6144, 1536), (32768, 1536, 6144), (32768, 3072, 1536), (8192, 8192, 8192), (4096 * 4, 1536, 64),
                  (16384 + 100, 1536 + 64, 1536)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda")
    lib.mi355_tune_set(0, 0); ref = engine.op_linear(x, w, b, 0); t0 = timeit(lambda: engine.op_linear(x, w, b, 0))
    lib.mi355_tune_set(0, 1)
    t1 = timeit(lambda: engine.op_linear(x, w, b, 0))
    lib.mi355_tune_set(0, 1)
    mism = 0
    y0 = engine.op_linear(x, w, b, 0)
    maxd = float((y0.float() - ref.float()).abs().max())
    for rep in range(10):
        y = engine.op_linear(x, w, b, 0)
        if not torch.equal(y, y0): mism += 1
    t2 = timeit(lambda: engine.op_linear(x, w, b, 0))
    tt = timeit(lambda: torch.nn.functional.linear(x, w))
    fl = 2.0 * M * N * K
    bad += mism
    print(f"M={M} N={N} K={K}: simple {fl/t0/1e12:7.1f} | pp 4-phase {fl/t1/1e12:7.1f} | pp 2-phase {fl/t2/1e12:7.1f} | hipBLASLt {fl/tt/1e12:7.1f} TF | pp32 nondeterministic runs {mism}/10, max|pp32-simple| {maxd:.3g}", flush=True)
print("RACE-SCREEN", "FAIL" if bad else "OK")
