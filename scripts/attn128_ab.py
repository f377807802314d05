"""A/B of the head_dim-128 attention kernels (FLUX.1 / Wan / Qwen-Image): the shipped 8-wave kernels (running max; static softmax) against the
4-wave kernel with the hand-scheduled key loop (csrc/attn128_w4_asm.inc, the default where it applies; tune key 5 = 5 switches it off; static softmax only).  Parity vs fp32 SDPA on small and
ragged shapes first (one lane-layout slip would otherwise hide in a 4608-token haystack), then TFLOP/s at the model shapes.
usage: python scripts/attn128_ab.py [--quick]"""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import flux as fx, _lib
lib = _lib.load()


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def make(B, H, S, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    S_pad = (S + 63) // 64 * 64
    nrm = lambda t: t / t.pow(2).mean(-1, keepdim=True).sqrt()
    q = torch.zeros(B, H, S_pad, 128, device="cuda", dtype=torch.bfloat16); k = torch.zeros_like(q); v = torch.zeros_like(q)
    q[:, :, :S] = (nrm(torch.randn(B, H, S, 128, device="cuda", generator=g)) * scale).bfloat16()
    k[:, :, :S] = nrm(torch.randn(B, H, S, 128, device="cuda", generator=g)).bfloat16()
    v[:, :, :S] = torch.randn(B, H, S, 128, device="cuda", generator=g).bfloat16()
    return q, k, v.transpose(2, 3).contiguous(), v


def run(var, bound, q, k, vT, S, n_first):
    lib.mi355_tune_set(5, var); lib.mi355_tune_set(21, bound)
    try:
        o1, o2 = fx.op_attention128(q, k, vT, S, n_first)
    finally:
        lib.mi355_tune_set(5, 0); lib.mi355_tune_set(21, 0)
    B = q.shape[0]
    return torch.cat([o1.view(B, n_first, -1)] + ([o2.view(B, S - n_first, -1)] if o2 is not None else []), 1)


ok = True
for (B, H, S, n_first) in [(1, 1, 64, 64), (1, 2, 128, 128), (2, 3, 256 + 77, 256), (1, 2, 1000, 512), (2, 2, 4608, 512), (1, 2, 20280, 20280)]:
    q, k, vT, v = make(B, H, S, S + H)
    ref = torch.nn.functional.scaled_dot_product_attention(q[:, :, :S].float(), k[:, :, :S].float(), v[:, :, :S].float()).transpose(1, 2).reshape(B, S, H * 128)
    out = {name: run(var, bound, q, k, vT, S, n_first) for name, var, bound in (("dyn", 5, 0), ("static8", 5, 40), ("w4", 0, 40))}

    rel = {n: float((o.float() - ref).norm() / ref.norm()) for n, o in out.items()}
    same = torch.equal(out["w4"], run(2, 40, q, k, vT, S, n_first))
    d = float((out["w4"].float() - out["static8"].float()).abs().max())
    line = f"parity B={B} H={H} S={S}: rel-L2 vs fp32 SDPA dyn {rel['dyn']:.2e} static8 {rel['static8']:.2e} w4 {rel['w4']:.2e}; w4 run-to-run identical {same}; max|w4 - static8| {d:.2e}"
    if not (rel["w4"] < 6e-3 and same):
        ok = False
        err = (out["w4"].float() - ref).abs()
        line += f"\n   worst query rows: {err.amax(dim=(0, 2)).topk(min(8, S)).indices.tolist()}  per-head max: {err.reshape(B, S, H, 128).amax(dim=(0, 1, 3)).tolist()}"
        line += f"\n   per 32-query block max err (first 16 blocks): {[round(float(x), 3) for x in err.amax(dim=(0, 2)).reshape(-1)[: (S // 32) * 32].reshape(-1, 32).amax(1)[:16]]}"
    print(line, flush=True)
print("PARITY", "OK" if ok else "FAIL", flush=True)
if "--quick" not in sys.argv:
    for (B, H, S, what) in [(8, 24, 4608, "FLUX.1 joint, B = 8 at 1024^2"), (2, 24, 4608, "FLUX.1 B = 2"), (4, 12, 20280, "Wan2.1 480x832x49, forward batch 4"),
                            (4, 24, 4160, "Qwen-Image 1024^2 + 64 text tokens, forward batch 4")]:
        q, k, vT, v = make(B, H, S, 7)
        fl = 4.0 * B * H * S * S * 128
        r = {}
        for rep in range(3):
            for name, var, bound in (("dyn", 5, 0), ("static8", 5, 40), ("w4", 0, 40), ("novalu", 101, 40), ("noread", 102, 40), ("mfma", 103, 40), ("nosync", 104, 40)):
                lib.mi355_tune_set(5, var); lib.mi355_tune_set(21, bound)
                r.setdefault(name, []).append(fl / timeit(lambda: fx.op_attention128(q, k, vT, S, S)) / 1e12)
        lib.mi355_tune_set(5, 0); lib.mi355_tune_set(21, 0)
        med = {n: sorted(x)[1] for n, x in r.items()}
        print(f"{what:52s} S={S:6d}: running-max {med['dyn']:7.1f}  static 8-wave {med['static8']:7.1f}  w4 {med['w4']:7.1f} | ablations (garbage results): no softmax VALU {med['novalu']:7.1f}  no fragment reads {med['noread']:7.1f}  neither {med['mfma']:7.1f}  MFMAs alone (no loads, no barrier) {med['nosync']:7.1f} TFLOP/s", flush=True)
