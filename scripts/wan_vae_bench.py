"""Causal 3-D video VAE decode microbenchmark on one MI355X (released Wan2.1 / Qwen-Image VAE geometry, synthetic weights):
a 480 x 832 x 49-frame clip (13 latent frames) and a single 1024^2 image (the Qwen-Image case)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
sys.path.insert(0, ROOT)
import torch
from mi355_flow import vae


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2)
    a = ap.parse_args()
    from oracle import wan_vae_ref as V          # bench-side only: synthetic weights + the algorithmic FLOP count
    sd = {k: v.cuda() for k, v in V.make_synthetic_state_dict(V.WAN21, seed=1).items()}
    dec = vae.WanVAEDecoder(vae.WanVAEConfig())
    dec.bind_state_dict(sd)
    dec.ready()
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, (B, T, h, w) in {"wan_480x832x49": (1, 13, 60, 104), "qwen_1024x1024": (4, 1, 128, 128)}.items():
        lat = torch.randn(B, 16, T, h, w, device="cuda", generator=g).half()
        out = dec.decode(lat, max_batch=B); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            out = dec.decode(lat, max_batch=B)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / a.iters
        fl = V.decode_flops(V.WAN21, T, h, w) * B
        print(json.dumps({"case": name, "batch": B, "latent": [T, h, w], "frames": out.shape[1], "ms_per_decode": round(el * 1e3, 1),
                          "ms_per_sample": round(el * 1e3 / B, 1), "algorithmic_tflops": round(fl / el / 1e12, 1),
                          "tflop_per_sample": round(fl / B / 1e12, 2), "finite": bool(torch.isfinite(out.float()).all()),
                          "workspace_gib": round(dec.workspace_bytes(B, T, h, w) / 2**30, 2)}))


if __name__ == "__main__":
    main()
