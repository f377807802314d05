"""A/B of the implicit-GEMM 3x3 conv against plain GEMMs of the same (M, N, K) on one MI355X."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
import torch
from mi355_flow import _lib, vae
from mi355_flow.engine import op_linear

lib = _lib.load()


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


B = 2
shapes = [(1024, 128, 128, 0), (1024, 256, 128, 0), (512, 256, 256, 0), (512, 512, 256, 0), (256, 512, 512, 0), (128, 512, 512, 0),
          (1024, 256, 256, 1), (512, 512, 512, 1)]
for (H, ci, co, up) in shapes:
    hin = H // 2 if up else H
    x = torch.randn(B, hin, hin, ci, device="cuda").bfloat16()
    w = torch.randn(co, 9, ci, device="cuda").bfloat16() * 0.02
    b = torch.zeros(co, device="cuda")
    M, N, K = B * H * H, co, 9 * ci
    fl = 2.0 * M * N * K
    row = {"HxW": H, "cin": ci, "cout": co, "up": up, "M": M, "N": N, "K": K}
    for cc in (0, 1, 3, 4):
        if cc == 3 and co < 256: continue
        if cc == 4 and co != 128: continue
        _lib.check(lib.mi355_tune_set(4, cc))
        ms = timeit(lambda: vae.op_conv3x3(x, w, b, None, bool(up)))
        row[f"conv_cfg{cc}_TF"] = round(fl / ms / 1e9, 1)
    _lib.check(lib.mi355_tune_set(4, 0))
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = w.reshape(co, K)
    for gv, name in ((0, "gemm_simple_TF"), (1, "gemm_pp_TF")):
        _lib.check(lib.mi355_tune_set(0, gv))
        ms = timeit(lambda: op_linear(A, W, b))
        row[name] = round(fl / ms / 1e9, 1)
    _lib.check(lib.mi355_tune_set(0, 1))
    del A
    print(json.dumps(row), flush=True)
