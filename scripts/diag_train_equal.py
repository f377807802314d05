"""Diagnostic: is the training-mode forward (mi355_denoise_step_train) bit-identical to the no-grad replay (mi355_denoise_step)?"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
import torch
from mi355_flow import engine, _lib
from mi355_flow.weights import synthetic_state_dict

lib = _lib.load()
for kv in filter(None, os.environ.get("MI355_TUNE", "").split(",")):
    k_, v_ = kv.split("=")
    _lib.check(lib.mi355_tune_set(int(k_), int(v_)), "tune_set")
L = int(os.environ.get("LAYERS", "3"))
cfg = engine.TransformerConfig(num_layers=L, dual_layers=tuple(range(min(2, L))))
sd = synthetic_state_dict(cfg, device="cuda", seed=1, dtype=torch.bfloat16)
e = engine.Engine(cfg); e.bind_state_dict(sd); e.ready()
g = torch.Generator().manual_seed(3)
for (B, h, w) in ((1, 32, 32), (2, 64, 64)):
    x = torch.randn(B, 16, h, w, generator=g).half().cuda()
    x1 = (x.float() + 0.1 * torch.randn(B, 16, h, w, generator=g).cuda()).half()
    pe = torch.randn(B, 333, 4096, generator=g).bfloat16().cuda(); pp = torch.randn(B, 2048, generator=g).bfloat16().cuda()
    plan = e.plan(B, 1, h, w, 333, 1)
    t = torch.full((B,), 900.0)
    a = plan.denoise_step(x, t, pe, pp, None, None, 1.0, torch.full((B,), 0.9), torch.full((B,), 0.75), torch.full((B,), 0.7), 0.9, "Flow-SDE",
                          next_latents=x1, want=("noise_pred",))
    for scope in (False, True):
        e.set_train_scope(scope)
        b = plan.denoise_step_train(x, t, pe, pp, None, None, 1.0, torch.full((B,), 0.9), torch.full((B,), 0.75), torch.full((B,), 0.7), 0.9,
                                    "Flow-SDE", x1)
        d = (a.noise_pred.float() - b.noise_pred.float()).abs()
        print(json.dumps({"tune": os.environ.get("MI355_TUNE", ""), "layers": L, "B": B, "hw": h, "full_scope": scope,
                          "log_prob_equal": bool(torch.equal(a.log_prob, b.log_prob)), "noise_pred_equal": bool(torch.equal(a.noise_pred, b.noise_pred)),
                          "n_diff": int((d > 0).sum()), "max_abs_diff": float(d.max())}))
