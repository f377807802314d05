"""Wan2.1-T2V-1.3B rollout microbenchmark on one MI355X: real geometry (1.42 B parameters), 480p x 49 frames, synthetic weights."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
import torch
from mi355_flow import wan
from mi355_flow.weights import synthetic_wan_state_dict, wan_forward_flops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--frames", type=int, default=49)
    ap.add_argument("--n-text", type=int, default=512)
    ap.add_argument("--guidance", type=float, default=5.0)
    ap.add_argument("--denoise-steps", type=int, default=4)
    ap.add_argument("--iters", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda")
    cfg = wan.WanConfig()
    sched = wan.UniPCMultistepSDEScheduler(flow_shift=3.0, noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1, seed=42, dynamics_type="Flow-SDE")
    ad = wan.Wan2T2VNativeAdapter(synthetic_wan_state_dict(cfg, device=dev), cfg, sched, latent_storage_dtype="fp16")
    ad.rollout()
    B, N = a.batch, a.denoise_steps
    g = torch.Generator(device=dev).manual_seed(1)
    pe = torch.randn(B, a.n_text, cfg.text_dim, device=dev, generator=g).bfloat16()
    ne = torch.randn(B, a.n_text, cfg.text_dim, device=dev, generator=g).bfloat16() if a.guidance > 1 else None
    run = lambda: ad.inference(prompt=None, height=a.height, width=a.width, num_frames=a.frames, num_inference_steps=N, guidance_scale=a.guidance,
                               prompt_embeds=pe, negative_prompt_embeds=ne, compute_log_prob=True, trajectory_indices="all")
    s = run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters): s = run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / a.iters
    T, h, w = (a.frames - 1) // 4 + 1, a.height // 8, a.width // 8
    S = T * (h // 2) * (w // 2)
    n_cfg = 2 if ne is not None else 1
    F = wan_forward_flops(cfg, S, a.n_text)
    ok = bool(torch.isfinite(s[0].all_latents.float()).all() and torch.isfinite(s[0].log_probs).all())
    print(json.dumps({"model": "Wan2.1-T2V-1.3B geometry", "batch": B, "n_cfg": n_cfg, "video": f"{a.height}x{a.width}x{a.frames}", "tokens": S,
                      "denoise_steps": N, "s_per_rollout": round(el, 3), "denoise_steps_per_s": round(B * N / el, 3),
                      "forward_tflops": round(n_cfg * F * B * N / el / 1e12, 1), "frac_of_2.5PF": round(n_cfg * F * B * N / el / 2.5e15, 4),
                      "flops_per_forward_per_sample": F, "finite": ok,
                      "workspace_gib": round(ad.engine.plan(B, n_cfg, T, h, w, a.n_text, N).workspace_bytes / 2**30, 2)}))


if __name__ == "__main__":
    main()
