"""A/B of the 4-wave GEMM with the hand-scheduled main loop (csrc/gemm_w4_asm.inc, tune key 0 = 2) against the shipped 8-wave ping-pong kernel
(key 0 = 1) and hipBLASLt (torch) on the model's shapes: parity vs fp32 torch on the same bf16 operands, run-to-run bit equality (a staging
race shows up as a mismatch), bit equality with the ping-pong kernel (same k order inside a K-tile? NOT promised: reported), then timing in
interleaved rounds, and the loop's ablation builds (no LDS-DMA loads / no fragment reads / MFMA only: garbage results, timing only).
usage: python scripts/gemm_w4_ab.py [--quick]"""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import engine, _lib
import ctypes as C
lib = _lib.load()


def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def linear_dbg(x, w, b, mask):
    """mi355_op_linear_trace without a trace buffer = the plain bias GEMM with MI355_DBG_MASK ablation knobs"""
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    os.environ["MI355_DBG_MASK"] = str(mask)
    _lib.check(lib.mi355_op_linear_trace(torch.cuda.current_stream().cuda_stream, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, None), "trace")
    return out


ok = True
# ---- parity on small shapes first (a wrong address would otherwise be found in a 32768-row haystack)
for (M, N, K) in [(256, 256, 128), (256, 512, 256), (512, 256, 384), (1024, 768, 1536), (2048, 1536, 6144)]:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    ref = x.float() @ w.float().t() + b
    lib.mi355_tune_set(3, 1)                  # ping-pong / w4 from 1 tile up (default 128)
    lib.mi355_tune_set(0, 3); y1 = engine.op_linear(x, w, b, 0)
    lib.mi355_tune_set(0, 2); y2 = engine.op_linear(x, w, b, 0)
    rel1 = float((y1.float() - ref).norm() / ref.norm()); rel2 = float((y2.float() - ref).norm() / ref.norm())
    same = all(torch.equal(y2, engine.op_linear(x, w, b, 0)) for _ in range(5))
    # where is it wrong, if it is: per 64x64 block error map
    line = f"parity M={M} N={N} K={K}: pp rel-L2 {rel1:.3e}  w4 rel-L2 {rel2:.3e}  w4 run-to-run identical {same}  w4 == pp bitwise {torch.equal(y1, y2)}"
    if not (rel2 < 4e-3 and same):
        ok = False
        err = (y2.float() - ref).abs().reshape(M // 64, 64, N // 64, 64).amax(dim=(1, 3))
        line += "\n   max-abs error per 64x64 block (rows = m blocks):\n" + "\n".join("   " + " ".join(f"{float(v):8.2e}" for v in r) for r in err[:8])
    print(line, flush=True)
    for act, name in ((2, "gelu"), (1, "silu")):
        lib.mi355_tune_set(0, 3); z1 = engine.op_linear(x, w, b, act)
        lib.mi355_tune_set(0, 2); z2 = engine.op_linear(x, w, b, act)
        print(f"   act {name}: w4 == pp bitwise {torch.equal(z1, z2)}, max diff {float((z1.float() - z2.float()).abs().max()):.3e}", flush=True)
    # gated residual in place (x += gate * (a @ w.T + b)), two samples; and the schedule variants s1 / s3 of the loop
    gate = torch.randn(2, N, device="cuda", generator=g).bfloat16()
    x0 = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    refg = x0.float() + gate.float().repeat_interleave(M // 2, 0) * (x.float() @ w.float().t() + b).bfloat16().float()
    lib.mi355_tune_set(0, 3); g1 = engine.op_linear_gate_res(x0.clone(), x, w, b, gate, M // 2)
    lib.mi355_tune_set(0, 2); g2 = engine.op_linear_gate_res(x0.clone(), x, w, b, gate, M // 2)
    relg = float((g2.float() - refg).norm() / refg.norm())
    print(f"   gate_res: w4 rel-L2 {relg:.3e}, w4 == pp bitwise {torch.equal(g1, g2)}", flush=True)
    ok = ok and relg < 6e-3 and torch.equal(g1, g2)
    lib.mi355_tune_set(0, 2)
    for mask, name in ((36, "s1"), (37, "s3")):
        ys = linear_dbg(x, w, b, mask)
        print(f"   schedule {name}: == default schedule bitwise {torch.equal(ys, y2)}", flush=True)
        ok = ok and torch.equal(ys, y2)
lib.mi355_tune_set(3, 128)
lib.mi355_tune_set(0, 1)
print("PARITY", "OK" if ok else "FAIL", flush=True)

if "--quick" not in sys.argv:
    print(f"{'shape (M,N,K)':26s} {'pp':>8s} {'w4':>8s} {'w4 s1':>8s} {'w4 s3':>8s} {'hipBLASLt':>10s} | gate_res: {'pp':>8s} {'w4':>8s} | w4 ablations (no epilogue): {'no-load':>8s} {'no-read':>8s} {'mfma':>8s}   (TFLOP/s; median of 3 interleaved rounds)")
    for (M, N, K) in [(32768, 1536, 1536), (32768, 6144, 1536), (32768, 1536, 6144), (32768, 3072, 1536), (8192, 8192, 8192)]:
        x = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
        b = torch.randn(N, device="cuda")
        fl = 2.0 * M * N * K
        r = {k: [] for k in ("pp", "w4", "s1", "s3", "blt", "gpp", "gw4", "nl", "nr", "mf")}
        gate = torch.randn(8, N, device="cuda").bfloat16() * 0.01
        xres = torch.randn(M, N, device="cuda").bfloat16()
        for _ in range(3):
            lib.mi355_tune_set(0, 3); r["pp"].append(fl / timeit(lambda: engine.op_linear(x, w, b, 0)) / 1e12)
            r["gpp"].append(fl / timeit(lambda: engine.op_linear_gate_res(xres, x, w, b, gate, M // 8)) / 1e12)
            lib.mi355_tune_set(0, 2); r["w4"].append(fl / timeit(lambda: engine.op_linear(x, w, b, 0)) / 1e12)
            r["gw4"].append(fl / timeit(lambda: engine.op_linear_gate_res(xres, x, w, b, gate, M // 8)) / 1e12)
            r["s1"].append(fl / timeit(lambda: linear_dbg(x, w, b, 36)) / 1e12)
            r["s3"].append(fl / timeit(lambda: linear_dbg(x, w, b, 37)) / 1e12)
            r["blt"].append(fl / timeit(lambda: torch.nn.functional.linear(x, w)) / 1e12)
            r["nl"].append(fl / timeit(lambda: linear_dbg(x, w, b, 33)) / 1e12)
            r["nr"].append(fl / timeit(lambda: linear_dbg(x, w, b, 34)) / 1e12)
            r["mf"].append(fl / timeit(lambda: linear_dbg(x, w, b, 35)) / 1e12)
        med = {k: sorted(v)[1] for k, v in r.items()}
        print(f"{str((M, N, K)):26s} {med['pp']:8.1f} {med['w4']:8.1f} {med['s1']:8.1f} {med['s3']:8.1f} {med['blt']:10.1f} |           {med['gpp']:8.1f} {med['gw4']:8.1f} |                             {med['nl']:8.1f} {med['nr']:8.1f} {med['mf']:8.1f}", flush=True)
    lib.mi355_tune_set(0, 1)
