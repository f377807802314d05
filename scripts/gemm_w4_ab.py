"""A/B of the experimental 4-wave GEMM main loop (csrc/gemm_w4.hip, mi355_op_linear_w4) against the shipped ping-pong kernel and
hipBLASLt (torch): parity vs fp32 first, then TFLOP/s on the model's shapes.  `python scripts/gemm_w4_ab.py [out_file]`"""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import engine, _lib
lib = _lib.load()
out_f = open(sys.argv[1], "w") if len(sys.argv) > 1 else None


def say(s):
    print(s, flush=True)
    if out_f:
        out_f.write(s + "\n"); out_f.flush()


def timeit(fn, iters=12, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


say("# gemm_w4 (4 waves x 128x128, 32x32x16 MFMA, 1 barrier / K-tile) vs pp16 (shipped) vs hipBLASLt; bias epilogue; TFLOP/s")
g = torch.Generator(device="cuda").manual_seed(0)
ok = True
for (M, N, K) in [(256, 256, 64), (256, 256, 128), (512, 768, 192), (1024, 512, 1536)]:
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    y = engine.op_linear_w4(x, w, b)
    ref = x.float() @ w.float().t() + b
    rel = float((y.float() - ref).norm() / ref.norm())
    same = bool(torch.equal(y, engine.op_linear_w4(x, w, b)))
    say(f"parity {M}x{N}x{K}: rel-L2 {rel:.3e}  run-to-run identical {same}")
    ok = ok and rel < 4e-3 and same
say(f"parity {'OK' if ok else 'FAILED'}")
if ok:
    say("shape (M,N,K)".ljust(22) + "w4".rjust(9) + "pp16".rjust(9) + "hipBLASLt".rjust(11))
    for (M, N, K) in [(32768, 1536, 6144), (8192, 8192, 8192), (32768, 6144, 1536), (32768, 3072, 1536), (32768, 1536, 1536)]:
        x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
        b = torch.randn(N, device="cuda", generator=g)
        fl = 2.0 * M * N * K / 1e12
        t4 = timeit(lambda: engine.op_linear_w4(x, w, b))
        tp = timeit(lambda: engine.op_linear(x, w, b, 0))
        tb = timeit(lambda: torch.nn.functional.linear(x, w))
        say(f"{M},{N},{K}".ljust(22) + f"{fl / t4:9.1f}{fl / tp:9.1f}{fl / tb:11.1f}")
