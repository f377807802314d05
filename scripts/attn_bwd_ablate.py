"""Timing of the head_dim-64 attention-backward passes at the SD3.5 shape (B = 2, H = 24, S = 4429) through the operator entry point, for one
value of mi355_tune_set(43, .) per process (run under rocprofv3 --kernel-trace --stats: the per-kernel averages are the result).
43 = 0 round-3 kernels, 1 software-pipelined (shipped), 2..5 ablation builds of the dK/dV loop (MI355_ALLOW_ABLATION=1; wrong results)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import _lib
from mi355_flow.engine import _ptr, _stream
lib = _lib.load()
v = int(sys.argv[1]) if len(sys.argv) > 1 else 1
_lib.check(lib.mi355_tune_set(43, v))
B, H, S = 2, 24, 4429
S_pad = (S + 63) // 64 * 64
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.zeros(B, H, S_pad, 64, device="cuda", dtype=torch.bfloat16)
k, vv = torch.zeros_like(q), torch.zeros_like(q)
q[:, :, :S] = (torch.randn(B, H, S, 64, device="cuda", generator=g) * 0.18).bfloat16()
k[:, :, :S] = (torch.randn(B, H, S, 64, device="cuda", generator=g) * 1.3).bfloat16()
vv[:, :, :S] = torch.randn(B, H, S, 64, device="cuda", generator=g).bfloat16()
vT = vv.transpose(2, 3).contiguous()
do = torch.randn(B * S, H * 64, device="cuda", generator=g).bfloat16()
o = torch.empty_like(do)
dq, dk, dv = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(q)
for _ in range(12):
    _lib.check(lib.mi355_op_attention_fwd_bwd(_stream(), _ptr(q), _ptr(k), _ptr(vT), _ptr(do), _ptr(o), _ptr(dq), _ptr(dk), _ptr(dv), B, H, S, S_pad), "op")
torch.cuda.synchronize()
print("done", v)
