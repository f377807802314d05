import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import engine, _lib
lib = _lib.load()
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
B, H = 8, 24
for (S, n_img, scale) in [(4429, 4096, 1.0), (4096, 4096, 1.0), (4429, 4096, 3.0)]:
    S_pad = (S + 63) // 64 * 64
    q = torch.zeros(B, H, S_pad, 64, device="cuda", dtype=torch.bfloat16); k = torch.zeros_like(q); v = torch.zeros_like(q)
    q[:, :, :S] = (torch.randn(B, H, S, 64, device="cuda") * scale).bfloat16(); k[:, :, :S] = (torch.randn(B, H, S, 64, device="cuda") * scale).bfloat16()
    v[:, :, :S] = torch.randn(B, H, S, 64, device="cuda").bfloat16()
    vT = v.transpose(2, 3).contiguous()
    ref = torch.nn.functional.scaled_dot_product_attention(q[:1, :4, :S].float(), k[:1, :4, :S].float(), v[:1, :4, :S].float()).transpose(1, 2).reshape(1, S, 256)
    fl = 4.0 * B * H * S * S * 64
    # op-level entry: key 6 >= 2 = the proven |score| bound -> static-softmax kernels; the scaled inputs exceed any such bound and run the
    # running-max kernels.  (variant 3 = row sums on the matrix pipe was measured here in round 3 and deleted: profiles/r03a_attn_ab_variants.txt)
    lib.mi355_tune_set(6, 40 if scale == 1.0 else 0)
    for var in (0, 1, 2):                # 0 plain online softmax, 1 deferred rescale / static (shipped), 2 the same on 4-wave workgroups
        lib.mi355_tune_set(1, var)
        oi, oc = engine.op_attention(q, k, vT, S, n_img)
        got = torch.cat([oi.view(B, n_img, H * 64), oc.view(B, S - n_img, H * 64)], 1)[:1, :, :256].float()
        err = float((got - ref).norm() / ref.norm()); mabs = float((got - ref).abs().max())
        t = timeit(lambda: engine.op_attention(q, k, vT, S, n_img))
        print(f"S={S} scale={scale} variant {var}: {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF  rel-L2 {err:.2e} max-abs {mabs:.2e}", flush=True)
lib.mi355_tune_set(1, 1); lib.mi355_tune_set(6, 1)
