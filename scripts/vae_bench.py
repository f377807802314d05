"""VAE decode microbenchmark on one MI355X: SD3 decoder geometry, synthetic weights, 1024^2 images by default."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
import torch
from mi355_flow import _lib, vae


def decode_flops(cfg, h, w):
    rev = list(reversed(cfg.block_out_channels)); top = rev[0]; hw = h * w
    mac = hw * 9 * cfg.latent_channels * top + 4 * hw * 9 * top * top + 4 * hw * top * top + 2 * hw * hw * top
    prev, res = top, hw
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            ci = prev if j == 0 else co
            mac += res * 9 * ci * co + res * 9 * co * co + (res * ci * co if ci != co else 0)
        if i != len(rev) - 1:
            res *= 4; mac += res * 9 * co * co
        prev = co
    return 2.0 * (mac + res * 9 * rev[-1] * cfg.out_channels)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=128)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--conv-cfg", type=int, nargs="*", default=[0])
    a = ap.parse_args()
    cfg = vae.VAEConfig()
    dec = vae.VAEDecoder(cfg)
    g = torch.Generator(device="cuda").manual_seed(0)
    sd = {}
    tmp = vae.VAEDecoder.param_names(dec)
    shapes = {}
    # shapes from the engine's own table are not exposed; rebuild from the architecture
    def conv(n, co, ci, k=3): shapes[n + ".weight"] = (co, ci, k, k); shapes[n + ".bias"] = (co,)
    def lin(n, co, ci): shapes[n + ".weight"] = (co, ci); shapes[n + ".bias"] = (co,)
    def norm(n, c): shapes[n + ".weight"] = (c,); shapes[n + ".bias"] = (c,)
    def resnet(n, ci, co):
        norm(n + ".norm1", ci); conv(n + ".conv1", co, ci); norm(n + ".norm2", co); conv(n + ".conv2", co, co)
        if ci != co: conv(n + ".conv_shortcut", co, ci, 1)
    rev = list(reversed(cfg.block_out_channels)); top = rev[0]
    conv("decoder.conv_in", top, cfg.latent_channels)
    resnet("decoder.mid_block.resnets.0", top, top); resnet("decoder.mid_block.resnets.1", top, top)
    norm("decoder.mid_block.attentions.0.group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"): lin("decoder.mid_block.attentions.0." + n, top, top)
    prev = top
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1): resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        if i != len(rev) - 1: conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co)
        prev = co
    norm("decoder.conv_norm_out", prev); conv("decoder.conv_out", cfg.out_channels, prev)
    assert sorted(shapes) == sorted(tmp), set(tmp) ^ set(shapes)
    for n, s in shapes.items():
        if "norm" in n and n.endswith("weight"): sd[n] = torch.ones(s, device="cuda")
        elif n.endswith("bias"): sd[n] = torch.zeros(s, device="cuda")
        else:
            fan = 1
            for d in s[1:]: fan *= d
            sd[n] = torch.randn(s, device="cuda", generator=g) / fan ** 0.5
    dec.bind_state_dict(sd); dec.ready()
    lat = torch.randn(a.batch, 16, a.latent, a.latent, device="cuda", generator=g).half()
    fl = decode_flops(cfg, a.latent, a.latent) * a.batch
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
