"""GPU parity of the FLUX.1 rollout path against the CPU oracle (oracle/flux_ref.py) and plain torch fp32
references of the new operators (head_dim-128 attention, RMSNorm + RoPE).  Everything goes through the C ABI."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.bfloat16().float()


@pytest.fixture(scope="module")
def fx():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mi355_flow import flux
    return flux


@pytest.mark.parametrize("B,H,S,n_first,scale", [(1, 2, 300, 300, 1.0), (2, 3, 333, 77, 1.0), (1, 2, 1024, 512, 3.0), (1, 1, 64, 0, 1.0)])
def test_attention128_matches_torch(fx, B, H, S, n_first, scale):
    g = torch.Generator().manual_seed(S + H)
    S_pad = (S + 63) // 64 * 64
    q = torch.zeros(B, H, S_pad, 128); k = torch.zeros_like(q); v = torch.zeros_like(q)
    q[:, :, :S] = _bf(torch.randn(B, H, S, 128, generator=g) * scale)
    k[:, :, :S] = _bf(torch.randn(B, H, S, 128, generator=g) * scale)
    v[:, :, :S] = _bf(torch.randn(B, H, S, 128, generator=g))
    ref = F.scaled_dot_product_attention(q[:, :, :S], k[:, :, :S], v[:, :, :S]).transpose(1, 2).reshape(B, S, H * 128)
    o1, o2 = fx.op_attention128(q.bfloat16().cuda(), k.bfloat16().cuda(), v.transpose(2, 3).contiguous().bfloat16().cuda(), S, n_first)
    parts = [o1.view(B, n_first, H * 128).float().cpu()]
    if o2 is not None:
        parts.append(o2.view(B, S - n_first, H * 128).float().cpu())
    got = torch.cat(parts, dim=1)
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 6e-3, rel


def test_rope_norm_matches_oracle(fx):
    from oracle import flux_ref as R
    g = torch.Generator().manual_seed(5)
    B, H, Nt, hp, wp = 2, 3, 5, 3, 4
    Ni, D = hp * wp, 3 * 128
    S = Nt + Ni
    S_pad = 64
    ids = torch.cat([torch.zeros(Nt, 3), R.prepare_img_ids(hp, wp)], 0)
    cos, sin = R.rope_cos_sin(ids)
    cs = torch.stack([cos[:, 0::2], sin[:, 0::2]], dim=-1).contiguous()       # [S, 64, 2]
    src = _bf(torch.randn(B * Ni, 2 * D, generator=g))
    wq, wk = 1 + 0.1 * torch.randn(128, generator=g), 1 + 0.1 * torch.randn(128, generator=g)
    q, k = fx.op_rope_norm(src.bfloat16().cuda(), 0, D, wq.cuda(), wk.cuda(), cs.cuda(), B, H, Ni, Nt, S_pad, 1e-6, 0.5)
    x = src.view(B, Ni, 2, H, 128).permute(2, 0, 3, 1, 4)                      # (2, B, H, Ni, 128)
    rq = R.apply_rope(R._rms(x[0], wq, 1e-6), cos[Nt:], sin[Nt:]) * 0.5
    rk = R.apply_rope(R._rms(x[1], wk, 1e-6), cos[Nt:], sin[Nt:])
    assert (q[:, :, Nt:S].float().cpu() - rq).abs().max().item() < 2e-2
    assert (k[:, :, Nt:S].float().cpu() - rk).abs().max().item() < 4e-2
    assert float(q[:, :, :Nt].abs().max()) == 0.0 and float(q[:, :, S:].abs().max()) == 0.0      # only the addressed rows are written


def _setup(fx, cfg_o, seed=77):
    from oracle import flux_ref as R
    sd = {k: _bf(v) for k, v in R.make_synthetic_state_dict(cfg_o, seed).items()}
    cfg = fx.FluxConfig(num_layers=cfg_o.num_layers, num_single_layers=cfg_o.num_single_layers, num_attention_heads=cfg_o.num_attention_heads,
                        joint_attention_dim=cfg_o.joint_attention_dim, pooled_projection_dim=cfg_o.pooled_projection_dim,
                        guidance_embeds=cfg_o.guidance_embeds)
    return sd, cfg


@pytest.mark.parametrize("h,w,Nt,B,guidance_embeds", [(8, 8, 16, 2, True), (6, 10, 13, 1, True), (16, 16, 64, 2, False)])
def test_flux_forward_matches_oracle(fx, h, w, Nt, B, guidance_embeds):
    from oracle import flux_ref as R
    cfg_o = R.tiny_config()
    cfg_o.guidance_embeds = guidance_embeds
    sd, cfg = _setup(fx, cfg_o)
    eng = fx.FluxEngine(cfg)
    eng.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    eng.ready()
    g = torch.Generator().manual_seed(h * w + Nt)
    x = R.pack_latents(torch.randn(B, 16, h, w, generator=g)).half()
    enc = _bf(torch.randn(B, Nt, cfg_o.joint_attention_dim, generator=g))
    pool = _bf(torch.randn(B, cfg_o.pooled_projection_dim, generator=g))
    tm = torch.tensor([900.0, 412.5][:B])
    gm = torch.full((B,), 3500.0)
    ref = R.flux_forward(sd, cfg_o, x.float(), tm, gm, pool, enc, R.prepare_img_ids(h // 2, w // 2), premultiplied=True)
    plan = eng.plan(B, h, w, Nt, 1)
    got = plan.transformer_forward(x.cuda(), tm, gm if guidance_embeds else None, enc.cuda(), pool.cuda()).float().cpu()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 2e-2, rel
    eng.close()


def test_flux_rollout_matches_oracle_and_replays(fx):
    """N-step rollout (latents, per-step log-probs) vs the oracle on identical draws; adapter.forward() replay of a stored
    transition reproduces the rollout log-prob bit for bit (ratio == 1)."""
    from oracle import flux_ref as R, scheduler_ref as S
    from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler
    cfg_o = R.tiny_config()
    sd, cfg = _setup(fx, cfg_o, seed=11)
    sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[0, 1, 2, 3], num_sde_steps=2, seed=42, dynamics_type="Flow-SDE",
                                               shift=3.0, use_dynamic_shifting=True)
    ad = fx.Flux1NativeAdapter({k: v.cuda() for k, v in sd.items()}, cfg, sched, latent_storage_dtype="fp16")
    ad.rollout()
    B, Nt, N, H, W = 2, 16, 5, 128, 128
    g = torch.Generator().manual_seed(9)
    pe = torch.randn(B, Nt, cfg_o.joint_attention_dim, generator=g).bfloat16()
    pp = torch.randn(B, cfg_o.pooled_projection_dim, generator=g).bfloat16()
    torch.cuda.manual_seed(77)
    samples = ad.inference(prompt=["a", "b"], height=H, width=W, num_inference_steps=N, guidance_scale=3.5, prompt_embeds=pe.cuda(),
                           pooled_prompt_embeds=pp.cuda(), trajectory_indices="all")
    # identical draws, in the reference's order
    torch.cuda.manual_seed(77)
    h = w = H // 8
    init = R.pack_latents(torch.randn((B, 16, h, w), device="cuda", dtype=torch.bfloat16)).cpu()
    noise = torch.stack([torch.randn((B, (h // 2) * (w // 2), 64), device="cuda", dtype=torch.float32) for _ in range(N)]).cpu()
    ts = samples[0].timesteps.float().cpu()
    sig = ad.scheduler.sigmas.float().cpu()
    nl = ad.scheduler.host_noise_levels()
    assert sum(e > 0 for e in nl) == 2
    ref = R.rollout(sd, cfg_o, pe, pp, 3.5, init, noise, ts, sig, nl, R.prepare_img_ids(h // 2, w // 2), torch.float16)
    assert len(samples) == B and samples[0].all_latents.shape == (N + 1, (h // 2) * (w // 2), 64)
    assert samples[0].all_latents.dtype == torch.float16 and tuple(samples[0].img_ids.shape) == ((h // 2) * (w // 2), 3)
    sde = [i for i in range(N) if nl[i] > 0]
    for b in range(B):
        got = samples[b].all_latents.float().cpu()
        for pos in range(N + 1):
            r = ref["all_latents"][pos, b].float()
            assert ((got[pos] - r).norm() / r.norm()).item() < 2e-2
        lp = samples[b].log_probs.cpu()
        assert lp.shape == (len(sde),)
        torch.testing.assert_close(lp, torch.stack([ref["log_probs"][i, b] for i in sde]), rtol=1e-3, atol=1e-4)
    # replay (grpo.py:229-263): stored (x_i, x_{i+1}) of an SDE step through forward()
    i = sde[0]
    x_i = torch.stack([s.all_latents[i] for s in samples]).cuda()
    x_n = torch.stack([s.all_latents[i + 1] for s in samples]).cuda()
    t = samples[0].timesteps[i].reshape(1).expand(B).cuda()
    t_next = samples[0].timesteps[i + 1].reshape(1).expand(B).cuda()
    out = ad.forward(t=t, latents=x_i, prompt_embeds=pe.cuda(), pooled_prompt_embeds=pp.cuda(), img_ids=samples[0].img_ids, t_next=t_next,
                     next_latents=x_n, guidance_scale=3.5, noise_level=nl[i], compute_log_prob=True, return_kwargs=["log_prob", "next_latents_mean"])
    old = torch.stack([s.log_probs[0] for s in samples]).cuda()
    assert torch.equal(torch.exp(out.log_prob - old), torch.ones_like(old))
    ad.engine.close()


def test_flux_errors(fx):
    with pytest.raises(RuntimeError, match="head_dim must be 128"):
        fx.FluxEngine(fx.FluxConfig(attention_head_dim=64))
    eng = fx.FluxEngine(fx.FluxConfig(num_layers=1, num_single_layers=1, num_attention_heads=1, joint_attention_dim=64, pooled_projection_dim=64))
    plan = eng.plan(1, 4, 4, 8, 1)
    x = torch.zeros(1, 4, 64).cuda()
    with pytest.raises(RuntimeError, match="has not been bound"):
        plan.transformer_forward(x, torch.tensor([500.0]), torch.tensor([3500.0]), torch.zeros(1, 8, 64).cuda(), torch.zeros(1, 64).cuda())
    with pytest.raises(RuntimeError, match="must be even"):
        eng.plan(1, 5, 4, 8, 1)
    eng.close()


def test_flux_full_width_blocks_large_grid_dispatch(fx):
    """FLUX.1-dev WIDTH (D = 3072, 24 heads x 128, T5 width 4096) with 1 double + 2 single blocks at a token count that takes the
    large-grid paths (persistent ping-pong GEMMs incl. the K = 15 360 proj_out, 8-wave attention128, EPI_VT head_dim 128)."""
    from oracle import flux_ref as R
    cfg_o = R.FluxConfig(num_layers=1, num_single_layers=2)
    sd = {k: _bf(v) for k, v in R.make_synthetic_state_dict(cfg_o, seed=5, std=0.02).items()}
    cfg = fx.FluxConfig(num_layers=1, num_single_layers=2)
    eng = fx.FluxEngine(cfg)
    eng.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    eng.ready()
    B, h, w, Nt = 4, 64, 64, 512
    g = torch.Generator().manual_seed(3)
    x = R.pack_latents(torch.randn(B, 16, h, w, generator=g)).half()
    enc = _bf(torch.randn(B, Nt, 4096, generator=g))
    pool = _bf(torch.randn(B, 768, generator=g))
    tm = torch.tensor([875.0, 600.0, 310.5, 48.0])
    gm = torch.full((B,), 3500.0)
    plan = eng.plan(B, h, w, Nt, 1)
    got = plan.transformer_forward(x.cuda(), tm, gm, enc.cuda(), pool.cuda())
    from _gpu_oracle import check_in_band
    check_in_band("FLUX full width 1 + 2 blocks, B = 4, 512^2 (large-grid dispatch)", got, R.flux_forward, sd, cfg_o, x.float(), tm, gm, pool, enc,
                  R.prepare_img_ids(h // 2, w // 2), premultiplied=True)
    eng.close()


def test_flux_stepwise_callbacks_equal_fused_rollout(fx):
    """extra_call_back_kwargs that need per-step tensors switch to per-step engine calls: same kernels, bit-identical trajectory."""
    from oracle import flux_ref as R
    from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler
    cfg_o = R.tiny_config()
    sd, cfg = _setup(fx, cfg_o, seed=3)
    sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[0, 1, 2], num_sde_steps=2, seed=1, dynamics_type="Flow-SDE",
                                               shift=3.0, use_dynamic_shifting=True)
    ad = fx.Flux1NativeAdapter({k: v.cuda() for k, v in sd.items()}, cfg, sched)
    ad.rollout()
    g = torch.Generator().manual_seed(1)
    pe = torch.randn(2, 16, cfg_o.joint_attention_dim, generator=g).bfloat16().cuda()
    pp = torch.randn(2, cfg_o.pooled_projection_dim, generator=g).bfloat16().cuda()
    kw = dict(prompt=["a", "b"], height=64, width=64, num_inference_steps=4, guidance_scale=3.5, prompt_embeds=pe, pooled_prompt_embeds=pp)
    torch.cuda.manual_seed(5)
    a = ad.inference(**kw)
    torch.cuda.manual_seed(5)
    b = ad.inference(**kw, extra_call_back_kwargs=["noise_pred", "noise_level"])
    for sa, sb in zip(a, b):
        assert torch.equal(sa.all_latents, sb.all_latents) and torch.equal(sa.log_probs, sb.log_probs)
        assert sb.extra_kwargs["noise_pred"].shape[0] == 4 and sb.extra_kwargs["noise_pred"].shape[1:] == sa.all_latents.shape[1:]
    ad.engine.close()


def test_flux_full_width_blocks_at_1024_token_count(fx):
    """BASELINE.json configs[2] token count: FLUX.1-dev width, one double-stream + one single-stream block at 1024^2
    (4096 image + 512 text = 4608 joint tokens), B = 1, vs the fp32 oracle (model body unpinned, oracle/flux_ref.py)."""
    from oracle import flux_ref as R
    cfg_o = R.FluxConfig(num_layers=1, num_single_layers=1)
    sd = {k: _bf(v) for k, v in R.make_synthetic_state_dict(cfg_o, seed=11, std=0.02).items()}
    eng = fx.FluxEngine(fx.FluxConfig(num_layers=1, num_single_layers=1))
    eng.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    eng.ready()
    B, h, w, Nt = 1, 128, 128, 512
    g = torch.Generator().manual_seed(17)
    x = R.pack_latents(torch.randn(B, 16, h, w, generator=g)).half()
    enc = _bf(torch.randn(B, Nt, 4096, generator=g))
    pool = _bf(torch.randn(B, 768, generator=g))
    tm, gm = torch.tensor([640.0]), torch.full((B,), 3500.0)
    got = eng.plan(B, h, w, Nt, 1).transformer_forward(x.cuda(), tm, gm, enc.cuda(), pool.cuda())
    from _gpu_oracle import check_in_band
    check_in_band("FLUX full-width 1+1 blocks, S = 4608", got, R.flux_forward, sd, cfg_o, x.float(), tm, gm, pool, enc,
                  R.prepare_img_ids(h // 2, w // 2), premultiplied=True)
    eng.close()
