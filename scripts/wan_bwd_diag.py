"""Diagnostic for the Wan native backward (opt-in): per-parameter rel-L2 vs the fp32 oracle autograd for a list of shapes."""
import os, sys
os.environ["MI355_WAN_NATIVE_BACKWARD"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "flow-factory_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import test_gpu_wan_backward as T
from mi355_flow import wan as wn
from oracle import wan_ref as R

cfg_o = R.tiny_config()
for (B, Tt, h, w, Nt, g) in [(1, 2, 6, 10, 64, 5.0), (1, 2, 6, 10, 63, 5.0), (2, 2, 6, 10, 64, 5.0), (1, 2, 6, 10, 64, 1.0), (1, 3, 8, 12, 64, 5.0)]:
    ad, mod = T._build(wn, cfg_o, lambda n: any(k in n for k in T.DEFAULT_TARGETS))
    inp = T._inputs(cfg_o, B, Tt, h, w, Nt, seed=5)
    ad.scheduler.set_timesteps(4)
    ad.scheduler.sigmas = ad.scheduler.sigmas.clone(); ad.scheduler.sigmas[1] = 0.9
    kw = T._kw(inp, B, 900.0, 750.0, 0.7, g)
    out = ad.forward(**kw)
    ((inp["wlp"].cuda() * out.log_prob).sum() + 3.0 * (inp["wnp"].cuda() * out.noise_pred).mean()).backward()
    _, g_ref = T._oracle_loss(mod, cfg_o, inp, g, 900.0, 750.0, 0.7, 0.9, 3.0)
    bad = [(n.replace("blocks.", "b").replace(".weight", ".w").replace(".bias", ".b"), round(T._rel(p.grad, g_ref[n]), 3)) for n, p in mod.named_parameters()
           if p.requires_grad and T._rel(p.grad, g_ref[n]) > 0.15]
    print(f"B{B} T{Tt} {h}x{w} Nt{Nt} g{g}: S={Tt*(h//2)*(w//2)} Mc={B*(2 if g>1 else 1)*Nt}: {len(bad)} bad: {bad}", flush=True)
    ad.engine.close()
