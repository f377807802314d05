"""Qwen-Image rollout microbenchmark on one MI355X: the real geometry (60 layers, 20.4 B parameters = 41 GB bf16 RESIDENT -- the
reference shards this model with FSDP2, config E), 1024^2, true CFG with a ragged negative prompt, synthetic weights."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
import torch
from mi355_flow import qwen


def param_shape(cfg, name):
    """Shape of the diffusers QwenImageTransformer2DModel parameter `name` at this config."""
    D, J, C, T, hd = cfg.dim, cfg.joint_attention_dim, cfg.in_channels, cfg.time_proj_dim, cfg.attention_head_dim
    base = name[:-5] if name.endswith(".bias") else name[:-7]
    if "norm_q" in name or "norm_k" in name or "norm_added" in name:
        return (hd,)
    if name == "txt_norm.weight":
        return (J,)
    out_in = {"img_in": (D, C), "txt_in": (D, J), "time_text_embed.timestep_embedder.linear_1": (D, T),
              "time_text_embed.timestep_embedder.linear_2": (D, D), "norm_out.linear": (2 * D, D), "proj_out": (C, D)}.get(base)
    if out_in is None:
        tail = base.split(".", 2)[2]
        out_in = {"img_mod.1": (6 * D, D), "txt_mod.1": (6 * D, D), "img_mlp.net.0.proj": (4 * D, D), "img_mlp.net.2": (D, 4 * D),
                  "txt_mlp.net.0.proj": (4 * D, D), "txt_mlp.net.2": (D, 4 * D)}.get(tail, (D, D))
    return out_in if name.endswith(".weight") else (out_in[0],)


def synthetic_tensor(cfg, name, device, g, std=0.02):
    shape = param_shape(cfg, name)
    t = torch.randn(shape, device=device, generator=g, dtype=torch.bfloat16) * std
    return t + 1 if len(shape) == 1 and "norm" in name else t


def synthetic_weights(engine, device, seed=7, std=0.02):
    """name -> bf16 tensor, drawn on the GPU tensor by tensor and bound immediately (never 2 x 41 GB alive)."""
    g = torch.Generator(device=device).manual_seed(seed)
    for name in engine.param_names():
        engine.bind_tensor(name, synthetic_tensor(engine.cfg, name, device, g, std))
        if len(engine._keepalive) >= 64:
            engine.finish_binding()
    engine.finish_binding()
    engine.ready()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--n-text", type=int, default=64)
    ap.add_argument("--guidance", type=float, default=4.0)
    ap.add_argument("--denoise-steps", type=int, default=4)
    ap.add_argument("--layers", type=int, default=60)
    ap.add_argument("--iters", type=int, default=1)
    ap.add_argument("--dynamics", default="Flow-SDE", help="Flow-SDE | Dance-SDE | CPS | ODE (DGPO samples with ODE)")
    ap.add_argument("--ab-shapes", default=None, metavar="BxSIZE,...", help="shapes of the --ab-two-stream run (default: --batch x --size)")
    ap.add_argument("--ab-two-stream", action="store_true",
                    help="A/B of the text chain on a side stream (mi355_tune_set key 12) in ONE process: same seeded rollout under 0 / 1, "
                         "bit-identity asserted, seconds per rollout of both (not yet run on the GPU)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    cfg = qwen.QwenConfig(num_layers=a.layers)
    t0 = time.perf_counter()
    ad = qwen.QwenImageNativeAdapter.__new__(qwen.QwenImageNativeAdapter)
    ad.device, ad.transformer_dtype, ad._latent_storage, ad._live_weights = dev, torch.bfloat16, "bf16", None
    ad.vae_decoder, ad.vae_max_batch = None, 4
    ad.scheduler = qwen.FlowMatchEulerDiscreteSDEScheduler(shift=1.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=0.9, max_image_seq_len=8192,
                                                           shift_terminal=0.02, sde_steps=[1, 2, 3], num_sde_steps=1, dynamics_type=a.dynamics)
    ad.engine = qwen.QwenEngine(cfg)
    synthetic_weights(ad.engine, dev)
    torch.cuda.synchronize()
    t_bind = time.perf_counter() - t0
    ad.rollout()
    B, N = a.batch, a.denoise_steps
    g = torch.Generator(device=dev).manual_seed(1)
    J = cfg.joint_attention_dim
    pe = torch.randn(B, a.n_text, J, device=dev, generator=g).bfloat16()
    pm = torch.ones(B, a.n_text, dtype=torch.long, device=dev)
    n_neg = max(1, a.n_text // 8)
    ne = torch.randn(B, n_neg, J, device=dev, generator=g).bfloat16() if a.guidance > 1 else None
    nm = torch.ones(B, n_neg, dtype=torch.long, device=dev) if a.guidance > 1 else None
    run = lambda: ad.inference(prompt=None, height=a.size, width=a.size, num_inference_steps=N, guidance_scale=a.guidance, prompt_embeds=pe,
                               prompt_embeds_mask=pm, negative_prompt_embeds=ne, negative_prompt_embeds_mask=nm, compute_log_prob=a.dynamics != "ODE")
    s = run(); torch.cuda.synchronize()
    if a.ab_two_stream:
        from mi355_flow import _lib
        lib = _lib.load()
        for shape in (a.ab_shapes or f"{B}x{a.size}").split(","):
            B, size = (int(v) for v in shape.split("x"))
            pe = torch.randn(B, a.n_text, J, device=dev, generator=g).bfloat16()
            pm = torch.ones(B, a.n_text, dtype=torch.long, device=dev)
            ne = torch.randn(B, n_neg, J, device=dev, generator=g).bfloat16() if a.guidance > 1 else None
            nm = torch.ones(B, n_neg, dtype=torch.long, device=dev) if a.guidance > 1 else None
            run = lambda: ad.inference(prompt=None, height=size, width=size, num_inference_steps=N, guidance_scale=a.guidance, prompt_embeds=pe,
                                       prompt_embeds_mask=pm, negative_prompt_embeds=ne, negative_prompt_embeds_mask=nm, compute_log_prob=a.dynamics != "ODE")
            res, secs = {}, {}
            modes = ((0, 0), (1, 0), (0, 1), (1, 1))                  # (two-stream: key 12, hipGraph replay of the loop: key 17)
            for mode in modes + modes:
                lib.mi355_tune_set(12, mode[0]); lib.mi355_tune_set(17, mode[1])
                torch.cuda.manual_seed(5)
                o = run(); torch.cuda.synchronize()                 # (creates the side stream / captures the graph of this configuration)
                torch.cuda.manual_seed(5)
                t0 = time.perf_counter()
                for _ in range(a.iters): o = run()
                torch.cuda.synchronize()
                secs.setdefault(mode, []).append((time.perf_counter() - t0) / a.iters)
                lat = torch.stack([x.all_latents for x in o])
                if mode in res:
                    assert torch.equal(res[mode], lat), f"mode {mode}: run-to-run difference"
                res[mode] = lat
            lib.mi355_tune_set(12, 0); lib.mi355_tune_set(17, 0)
            same = all(bool(torch.equal(res[modes[0]], res[m])) for m in modes[1:])
            best = {m: min(v) for m, v in secs.items()}
            print(json.dumps({"ab": "qwen (two-stream key 12, graph key 17)", "batch": B, "image": f"{size}x{size}", "denoise_steps": N, "bit_identical": same,
                              "s_per_rollout": {f"two{m[0]}_graph{m[1]}": round(t, 4) for m, t in best.items()},
                              "gain_pct_vs_single_eager": {f"two{m[0]}_graph{m[1]}": round((best[modes[0]] / t - 1) * 100, 2) for m, t in best.items()}}), flush=True)
            assert same
        return
    t0 = time.perf_counter()
    for _ in range(a.iters): s = run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / a.iters
    Ni = (a.size // 16) ** 2
    n_cfg = 2 if ne is not None else 1
    D = cfg.dim
    def flops(nt):
        S = Ni + nt
        return 2.0 * (cfg.num_layers * (S * 12 * D * D + 2 * S * S * D) + Ni * 64 * D * 2 + nt * J * D)
    F = flops(a.n_text) + (flops(n_neg) if n_cfg == 2 else 0)          # algorithmic: each branch at its own text length
    ok = bool(torch.isfinite(s[0].all_latents.float()).all() and (s[0].log_probs is None or torch.isfinite(s[0].log_probs).all()))
    n_text_plan = (max(a.n_text, n_neg) + 31) // 32 * 32
    print(json.dumps({"model": f"Qwen-Image geometry, {cfg.num_layers} layers", "batch": B, "n_cfg": n_cfg, "dynamics": a.dynamics, "image": f"{a.size}x{a.size}", "tokens": Ni + a.n_text,
                      "denoise_steps": N, "bind_s": round(t_bind, 1), "s_per_rollout": round(el, 3), "denoise_steps_per_s": round(B * N / el, 3),
                      "forward_tflops": round(F * B * N / el / 1e12, 1), "frac_of_2.5PF": round(F * B * N / el / 2.5e15, 4),
                      "flops_per_step_per_sample": F, "finite": ok, "hbm_allocated_gib": round(torch.cuda.mem_get_info()[1] / 2**30 - torch.cuda.mem_get_info()[0] / 2**30, 1),
                      "workspace_gib": round(ad.engine.plan(B, n_cfg, a.size // 8, a.size // 8, n_text_plan, N).workspace_bytes / 2**30, 2)}))


if __name__ == "__main__":
    main()
