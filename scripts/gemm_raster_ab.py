"""A/B of the GEMM tile raster (mi355_tune_set(7, gm)): tile rows per band; 0 = row-major (round 1).  Bit-identical results expected."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import engine, _lib
lib = _lib.load()

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

shapes = [(32768, 1536, 1536), (32768, 6144, 1536), (32768, 1536, 6144), (32768, 3072, 1536), (1536, 32768, 1536), (8192, 8192, 8192),
          (8192, 6144, 1536), (8192, 1536, 6144), (2664, 3072, 1536), (2664, 6144, 1536)]
gms = [0, 2, 4, 6, 8, 12]
print("shape (M,N,K)".ljust(22) + "".join(f"gm={g}".rjust(9) for g in gms) + "   hipBLASLt")
for (M, N, K) in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda")
    lib.mi355_tune_set(7, 0)
    ref = engine.op_linear(x, w, b, 0)
    row = f"{M},{N},{K}".ljust(22)
    for g in gms:
        lib.mi355_tune_set(7, g)
        y = engine.op_linear(x, w, b, 0)
        assert torch.equal(y, ref), ("raster changed the result", M, N, K, g)
        t = timeit(lambda: engine.op_linear(x, w, b, 0))
        row += f"{2.0 * M * N * K / t / 1e12:9.1f}"
    tt = timeit(lambda: torch.nn.functional.linear(x, w))
    print(row + f"{2.0 * M * N * K / tt / 1e12:12.1f}", flush=True)
lib.mi355_tune_set(7, 6)
