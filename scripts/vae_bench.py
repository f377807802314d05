"""VAE decode microbenchmark on one MI355X: SD3 decoder geometry, synthetic weights, 1024^2 images by default."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
import torch
from mi355_flow import _lib, vae
from mi355_flow.weights import synthetic_vae_state_dict, vae_decode_flops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=128)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--conv-cfg", type=int, nargs="*", default=[0])
    a = ap.parse_args()
    cfg = vae.VAEConfig()
    dec = vae.VAEDecoder(cfg)
    dec.bind_state_dict(synthetic_vae_state_dict(cfg)); dec.ready()
    g = torch.Generator(device="cuda").manual_seed(0)
    lat = torch.randn(a.batch, 16, a.latent, a.latent, device="cuda", generator=g).half()
    fl = vae_decode_flops(cfg, a.latent, a.latent) * a.batch
    for cc in a.conv_cfg:
        _lib.check(_lib.load().mi355_tune_set(4, cc))
        img = dec.decode(lat, max_batch=a.batch); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): img = dec.decode(lat, max_batch=a.batch)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(json.dumps({"conv_cfg": cc, "batch": a.batch, "image": a.latent * 8, "ms_per_decode": round(ms, 2),
                          "ms_per_image": round(ms / a.batch, 2), "tflops": round(fl / ms / 1e9, 1),
                          "workspace_gib": round(dec.workspace_bytes(a.batch, a.latent, a.latent) / 2**30, 2),
                          "finite": bool(torch.isfinite(img.float()).all())}))


if __name__ == "__main__":
    main()
