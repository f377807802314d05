"""Diagnostic: is a sample's result independent of the batch it sits in (ping-pong 256x256 GEMM kernel at B=8 vs 128x128 kernel at B=1)?"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
import torch
from mi355_flow import engine
from mi355_flow.weights import synthetic_state_dict
cfg = engine.TransformerConfig(num_layers=3, dual_layers=(0, 1))
e = engine.Engine(cfg); e.bind_state_dict(synthetic_state_dict(cfg, device="cuda", seed=1, dtype=torch.bfloat16)); e.ready()
g = torch.Generator().manual_seed(3)
h = w = 64
x = torch.randn(1, 16, h, w, generator=g).half().cuda()
pe = torch.randn(1, 333, 4096, generator=g).bfloat16().cuda(); pp = torch.randn(1, 2048, generator=g).bfloat16().cuda()
t = torch.tensor([900.0])
y1 = e.plan(1, 1, h, w, 333, 1).transformer_forward(x, t.cuda(), pe, pp)
for B in (2, 4, 8):
    yb = e.plan(B, 1, h, w, 333, 1).transformer_forward(x.repeat(B, 1, 1, 1), t.repeat(B).cuda(), pe.repeat(B, 1, 1), pp.repeat(B, 1))
    d = (yb.float() - y1.float()).abs()
    print(json.dumps({"B": B, "equal_to_alone": bool(torch.equal(yb, y1.expand_as(yb))), "all_rows_equal": bool(all(torch.equal(yb[0], yb[i]) for i in range(B))),
                      "n_diff": int((d > 0).sum()), "max_abs": float(d.max())}))
