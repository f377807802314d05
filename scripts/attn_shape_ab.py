"""head_dim-64 attention at small grids: 8-wave / 4-wave / 4-wave-at-five-per-SIMD workgroups (mi355_tune_set(23, .)), static and running-max
softmax, timed per launch and compared bit for bit.  Shapes = the joint attention of SD3.5 at the reference's 512^2 examples (S = 1357) for
forward batches 4 / 8 / 16, the dual attention there (S = 1024), and B' = 1 / 2 at 1024^2."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import engine, _lib
lib = _lib.load()


def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


H = 24
for (B, S, n_img) in [(4, 1357, 1024), (8, 1357, 1024), (16, 1357, 1024), (4, 1024, 1024), (8, 1024, 1024), (1, 4429, 4096), (2, 4429, 4096), (4, 4429, 4096)]:
    S_pad = (S + 63) // 64 * 64
    g = torch.Generator(device="cuda").manual_seed(S + B)
    q = torch.zeros(B, H, S_pad, 64, device="cuda", dtype=torch.bfloat16); k = torch.zeros_like(q); v = torch.zeros_like(q)
    q[:, :, :S] = torch.randn(B, H, S, 64, device="cuda", generator=g).bfloat16(); k[:, :, :S] = torch.randn(B, H, S, 64, device="cuda", generator=g).bfloat16()
    v[:, :, :S] = torch.randn(B, H, S, 64, device="cuda", generator=g).bfloat16()
    vT = v.transpose(2, 3).contiguous()
    fl = 4.0 * B * H * S * S * 64
    for bound in (40, 0):
        lib.mi355_tune_set(6, bound if bound else 1)
        if not bound:
            lib.mi355_tune_set(6, 0)
        ref = None
        line = f"B'={B:2d} S={S} {'static ' if bound else 'dynamic'}:"
        for shape in (1, 2, 3, 0):
            lib.mi355_tune_set(23, shape)
            oi, oc = engine.op_attention(q, k, vT, S, n_img)
            got = torch.cat([oi.reshape(-1), oc.reshape(-1)])
            if ref is None: ref = got.clone()
            same = torch.equal(got, ref)
            t = timeit(lambda: engine.op_attention(q, k, vT, S, n_img))
            line += f"  shape {shape}: {t*1e6:7.1f} us {fl/t/1e12:6.1f} TF{'' if same else ' DIFFERS'}"
        print(line, flush=True)
lib.mi355_tune_set(23, 0); lib.mi355_tune_set(6, 1)
