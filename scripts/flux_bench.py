"""FLUX.1-dev rollout microbenchmark on one MI355X: real geometry (11.9 B parameters), synthetic weights / prompts."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
import torch
from mi355_flow import flux
from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler
from mi355_flow.weights import synthetic_flux_state_dict, flux_forward_flops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--n-text", type=int, default=512)
    ap.add_argument("--denoise-steps", type=int, default=4)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--attn-only", action="store_true")
    ap.add_argument("--ab-two-stream", default=None, metavar="BxSIZE,...",
                    help="A/B of the double blocks' text chain on a side stream (mi355_tune_set key 14) in ONE process over a list of shapes, "
                         "e.g. 1x384,2x512,1x1024,8x1024 (the reference's examples sample at 1x384 / 2x512); bit-identity asserted (not yet run on the GPU)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    if a.attn_only:
        B, H, S = a.batch, 24, (a.size // 16) ** 2 + a.n_text
        S_pad = (S + 63) // 64 * 64
        q = torch.randn(B, H, S_pad, 128, device=dev).bfloat16(); k = torch.randn_like(q); vT = torch.randn(B, H, 128, S_pad, device=dev).bfloat16()
        from mi355_flow import _lib
        for var in (0, 1, 0, 1):
            _lib.check(_lib.load().mi355_tune_set(5, var))
            flux.op_attention128(q, k, vT, S); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): flux.op_attention128(q, k, vT, S)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(json.dumps({"op": "attention128", "variant": var, "B": B, "H": H, "S": S, "ms": round(ms, 3),
                              "tflops": round(4.0 * B * H * S * S * 128 / ms / 1e9, 1)}))
        _lib.check(_lib.load().mi355_tune_set(5, 0))
        return
    cfg = flux.FluxConfig()
    sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1, seed=42, dynamics_type="Flow-SDE",
                                               shift=3.0, use_dynamic_shifting=True)
    t0 = time.time()
    sd = synthetic_flux_state_dict(cfg, device=dev)
    ad = flux.Flux1NativeAdapter(sd, cfg, sched, latent_storage_dtype="fp16")
    del sd
    torch.cuda.empty_cache()
    print(f"weights bound in {time.time() - t0:.1f} s", file=sys.stderr)
    ad.rollout()
    if a.ab_two_stream:
        from mi355_flow import _lib
        lib = _lib.load()
        N = a.denoise_steps
        for shape in a.ab_two_stream.split(","):
            B, size = (int(v) for v in shape.split("x"))
            g = torch.Generator(device=dev).manual_seed(1)
            pe = torch.randn(B, a.n_text, cfg.joint_attention_dim, device=dev, generator=g).bfloat16()
            pp = torch.randn(B, cfg.pooled_projection_dim, device=dev, generator=g).bfloat16()
            run = lambda: ad.inference(prompt=None, height=size, width=size, num_inference_steps=N, guidance_scale=3.5, prompt_embeds=pe,
                                       pooled_prompt_embeds=pp, compute_log_prob=True, trajectory_indices="all")
            res, secs = {}, {}
            modes = ((0, 0), (1, 0), (0, 1), (1, 1))              # (two-stream double blocks: key 14, hipGraph replay of the loop: key 16)
            for mode in modes + modes:
                lib.mi355_tune_set(14, mode[0]); lib.mi355_tune_set(16, mode[1])
                torch.cuda.manual_seed(5)
                o = run(); torch.cuda.synchronize()               # (first graph-mode run of a configuration captures)
                torch.cuda.manual_seed(5)
                t0 = time.perf_counter()
                for _ in range(a.iters): o = run()
                torch.cuda.synchronize()
                secs.setdefault(mode, []).append((time.perf_counter() - t0) / a.iters)
                lat = torch.stack([x.all_latents for x in o])
                assert mode not in res or torch.equal(res[mode], lat), f"mode {mode}: run-to-run difference"
                res[mode] = lat
            lib.mi355_tune_set(14, 0); lib.mi355_tune_set(16, 0)
            same = all(bool(torch.equal(res[modes[0]], res[m])) for m in modes[1:])
            best = {m: min(v) for m, v in secs.items()}
            print(json.dumps({"ab": "flux (two-stream key 14, graph key 16)", "batch": B, "size": size, "denoise_steps": N, "bit_identical": same,
                              "s_per_rollout": {f"two{m[0]}_graph{m[1]}": round(t, 4) for m, t in best.items()},
                              "denoise_steps_per_s": {f"two{m[0]}_graph{m[1]}": round(B * N / t, 2) for m, t in best.items()},
                              "gain_pct_vs_single_eager": {f"two{m[0]}_graph{m[1]}": round((best[modes[0]] / t - 1) * 100, 2) for m, t in best.items()}}),
                  flush=True)
            assert same
        return
    B, N = a.batch, a.denoise_steps
    g = torch.Generator(device=dev).manual_seed(1)
    pe = torch.randn(B, a.n_text, cfg.joint_attention_dim, device=dev, generator=g).bfloat16()
    pp = torch.randn(B, cfg.pooled_projection_dim, device=dev, generator=g).bfloat16()
    run = lambda: ad.inference(prompt=None, height=a.size, width=a.size, num_inference_steps=N, guidance_scale=3.5, prompt_embeds=pe,
                               pooled_prompt_embeds=pp, compute_log_prob=True, trajectory_indices="all")
    s = run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters): s = run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / a.iters
    Ni = (a.size // 16) ** 2
    F = flux_forward_flops(cfg, Ni, a.n_text)
    ok = bool(torch.isfinite(s[0].all_latents.float()).all() and torch.isfinite(s[0].log_probs).all())
    print(json.dumps({"model": "FLUX.1-dev geometry", "batch": B, "size": a.size, "denoise_steps": N, "s_per_rollout": round(el, 3),
                      "denoise_steps_per_s": round(B * N / el, 2), "forward_tflops": round(F * B * N / el / 1e12, 1),
                      "frac_of_2.5PF": round(F * B * N / el / 2.5e15, 4), "flops_per_forward_per_sample": F, "finite": ok,
                      "workspace_gib": round(ad.engine.plan(B, a.size // 8, a.size // 8, a.n_text, N).workspace_bytes / 2**30, 2)}))


if __name__ == "__main__":
    main()
