"""A/B of the 256x192-tile GEMM kernel (gemm_w6_kernel, mi355_tune_set key 40) against what the dispatch runs without it:
(1) operator level -- bias / GELU / in-place gated residual at the shapes of an 8192- / 16384-row image stream (SD3.5 at B = 2 / 4, 1024^2),
    FLUX.1's 4608 rows, and shapes the rule must leave alone: results must be BIT-IDENTICAL (same MFMA, same k order per output element, shared
    epilogues), then timing;
(2) the whole SD3.5-medium forward at 1024^2 (B = 1, 2, 4) and 512^2 (B = 8): bit-identical under key 40 = 0 / 1 / 2 (covers the q/k RMSNorm
    epilogue, which has no operator-level entry point)."""
import math, os, sys, json, torch
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


def mode(v):
    _lib.check(lib.mi355_tune_set(40, v))


bad = 0
shapes = [(8192, 1536, 1536), (8192, 1536, 6144), (8192, 3072, 1536), (8192, 4608, 1536), (8192, 6144, 1536), (16384, 1536, 1536), (16384, 1536, 6144),
          (16384, 3072, 1536), (4608, 3072, 3072), (4608, 3072, 12288), (32768, 1536, 1536), (4096, 1536, 1536), (2048, 1536, 128), (256, 192, 128),
          (8192 + 256, 1536 + 192, 256), (8192, 1536 + 64, 1536), (8192 + 8, 1536, 1536)]
for (M, N, K) in shapes:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    rps = 1024 if M % 1024 == 0 else 333
    gate = torch.randn((M + rps - 1) // rps, N, device="cuda", generator=g).bfloat16()
    res0 = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    fl = 2.0 * M * N * K
    row = {"M": M, "N": N, "K": K}
    for name, fn in (("bias", lambda: engine.op_linear(x, w, b, 0)), ("gelu", lambda: engine.op_linear(x, w, b, 2)),
                     ("gate_res", lambda: engine.op_linear_gate_res(res0.clone(), x, w, b, gate, rps))):
        mode(0); ref = fn()
        mode(2); got = fn()
        same = bool(torch.equal(ref, got))
        rep = sum(0 if torch.equal(fn(), got) else 1 for _ in range(5))
        if not same or rep:
            bad += 1
        if name == "gate_res":      # (time the GEMM, not the clone: in place on a scratch copy)
            scratch = res0.clone()
            tf = lambda: engine.op_linear_gate_res(scratch, x, w, b, gate, rps)
        else:
            tf = fn
        mode(0); t0 = timeit(tf)
        mode(2); t2 = timeit(tf)
        mode(1); t1 = timeit(tf)
        row[name] = {"bit_identical": same, "nondeterministic_reruns": rep, "us_off": round(t0 * 1e6, 1), "us_forced": round(t2 * 1e6, 1),
                     "us_rule": round(t1 * 1e6, 1), "tflops_off": round(fl / t0 / 1e12, 1), "tflops_forced": round(fl / t2 / 1e12, 1)}
    print(json.dumps(row), flush=True)
mode(1)

# ---- whole forward, full width
from mi355_flow.weights import synthetic_state_dict
cfg = engine.TransformerConfig()
e = engine.Engine(cfg)
e.bind_state_dict(synthetic_state_dict(cfg, device="cuda", seed=1234, dtype=torch.bfloat16))
e.ready()
for (B, hw) in ((2, 128), (4, 128), (1, 128), (8, 64)):
    g = torch.Generator(device="cuda").manual_seed(B * hw)
    xl = torch.randn(B, 16, hw, hw, device="cuda", generator=g).half()
    pe = torch.randn(B, 333, 4096, device="cuda", generator=g).bfloat16()
    pp = torch.randn(B, 2048, device="cuda", generator=g).bfloat16()
    t = torch.full((B,), 700.0, device="cuda")
    plan = e.plan(B, 1, hw, hw, 333, 1)
    outs, times = {}, {}
    for m in (0, 1, 2):
        mode(m)
        outs[m] = plan.transformer_forward(xl, t, pe, pp).clone()
        times[m] = timeit(lambda: plan.transformer_forward(xl, t, pe, pp), iters=10, warm=2)
    same = bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]))
    bad += 0 if same else 1
    print(json.dumps({"forward": f"B'={B} latent {hw}x{hw}", "bit_identical_0_1_2": same, "ms_off": round(times[0] * 1e3, 3), "ms_rule": round(times[1] * 1e3, 3),
                      "ms_forced": round(times[2] * 1e3, 3)}), flush=True)
mode(1)
e.close()
print("W6-KERNEL-AB", "FAIL" if bad else "OK")
