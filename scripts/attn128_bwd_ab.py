"""Microbenchmark + numerics of the head_dim-128 flash-attention backward (attention128_bwd.hip) at the FLUX.1 1024^2 shape (B = 1, 24 heads,
S = 4608): run under `rocprofv3 --kernel-trace --stats` for per-kernel durations; prints the gradients' distance to fp32 autograd."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
import torch
from mi355_flow import _lib
from mi355_flow.engine import _ptr, _stream
lib = _lib.load()
B, H, S = 1, int(os.environ.get("H", 24)), int(os.environ.get("S", 4608))
S_pad = (S + 63) // 64 * 64
g = torch.Generator(device="cuda").manual_seed(1)
c = 1.4426950408889634 / math.sqrt(128.0)
q = torch.zeros(B, H, S_pad, 128, device="cuda", dtype=torch.bfloat16); k = torch.zeros_like(q); v = torch.zeros_like(q)
q[:, :, :S] = (torch.randn(B, H, S, 128, device="cuda", generator=g) * c).bfloat16()
k[:, :, :S] = (torch.randn(B, H, S, 128, device="cuda", generator=g) * 1.2).bfloat16()
v[:, :, :S] = torch.randn(B, H, S, 128, device="cuda", generator=g).bfloat16()
vT = v.transpose(2, 3).contiguous()
do = torch.randn(B * S, H * 128, device="cuda", generator=g).bfloat16()
o = torch.empty_like(do); dq, dk, dv = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(q)
_lib.check(lib.mi355_tune_set(21, 60))
for _ in range(int(os.environ.get("REPS", 6))):
    _lib.check(lib.mi355_op_attention128_fwd_bwd(_stream(), _ptr(q), _ptr(k), _ptr(vT), _ptr(do), _ptr(o), _ptr(dq), _ptr(dk), _ptr(dv), B, H, S, S_pad))
# numerics on 2 heads (fp32 autograd of the full 24 heads is 2 GiB of scores per head: slice)
hs = slice(0, 2)
qr = q[:, hs, :S].float().requires_grad_(True); kr = k[:, hs, :S].float().requires_grad_(True); vr = v[:, hs, :S].float().requires_grad_(True)
p = torch.softmax((qr @ kr.transpose(2, 3)) * math.log(2.0), dim=-1)
oref = (p @ vr).transpose(1, 2).reshape(B * S, 2 * 128)
oref.backward(do[:, :256].float())
rel = lambda a, b: float((a.float() - b).norm() / b.norm())
print({"S": S, "H": H, "o": rel(o[:, :256], oref), "dq": rel(dq[:, hs, :S], qr.grad), "dk": rel(dk[:, hs, :S], kr.grad), "dv": rel(dv[:, hs, :S], vr.grad)})
