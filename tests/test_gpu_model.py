"""GPU parity of the denoiser forward, the single denoise step and the whole rollout against the
CPU oracle (oracle/mmditx_ref.py, oracle/rollout_ref.py) on identical weights, prompts and noise.

Stated tolerances: network output rel-L2 <= 2e-2 vs the fp32 oracle on the
bf16-rounded weights (bf16 activations through L blocks); per-step latents differ by a few ulp of
the storage dtype; rollout log-prob rtol 1e-3 (north star); replay ratio == 1 engine-vs-engine.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _setup(cfg_o, std, seed=1234):
    from mi355_flow import engine
    from oracle import mmditx_ref as M
    sd = M.make_synthetic_state_dict(cfg_o, seed=seed, std=std)
    sd = {k: v.bfloat16().float() for k, v in sd.items()}  # both sides see the same bf16-rounded weights
    cfg_e = engine.TransformerConfig(
        num_layers=cfg_o.num_layers, num_heads=cfg_o.num_heads, joint_attention_dim=cfg_o.joint_attention_dim,
        pooled_projection_dim=cfg_o.pooled_projection_dim, pos_embed_max_size=cfg_o.pos_embed_max_size,
        dual_layers=tuple(cfg_o.dual_layers))
    e = engine.Engine(cfg_e)
    e.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    e.ready()
    return engine, e, sd


@pytest.fixture(scope="module")
def tiny():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import mmditx_ref as M
    cfg = M.tiny_config(num_layers=3, num_heads=2, dual_layers=(0, 1), joint_attention_dim=128, pooled_projection_dim=128,
                        pos_embed_max_size=24)
    engine, e, sd = _setup(cfg, std=0.08)
    yield cfg, engine, e, sd
    e.close()


@pytest.mark.parametrize("B,h,w,Nt,lat_dt", [(2, 16, 16, 13, torch.float16), (1, 32, 32, 77, torch.bfloat16),
                                             (3, 8, 24, 5, torch.float32)])
def test_forward_tiny_vs_oracle(tiny, B, h, w, Nt, lat_dt):
    from oracle import mmditx_ref as M
    cfg, engine, e, sd = tiny
    g = torch.Generator().manual_seed(B * 100 + h)
    x = torch.randn(B, 16, h, w, generator=g).to(lat_dt)
    enc = torch.randn(B, Nt, cfg.joint_attention_dim, generator=g).bfloat16()
    pooled = torch.randn(B, cfg.pooled_projection_dim, generator=g).bfloat16()
    t = torch.tensor([873.0] * B)
    plan = e.plan(B, 1, h, w, Nt, 4)
    y = plan.transformer_forward(x.cuda(), t.cuda(), enc.cuda(), pooled.cuda())
    torch.cuda.synchronize()
    t_net = t.to(lat_dt).float()  # the network sees t rounded to the latent dtype (sd3_5.py:394)
    ref = M.mmdit_forward(sd, cfg, x.float(), t_net, enc.float(), pooled.float())
    refq = M.mmdit_forward(sd, cfg, x.float(), t_net, enc.float(), pooled.float(), quant=M.bf16_round)
    assert torch.isfinite(y.float()).all()
    assert _rel(y, ref) < 2e-2, _rel(y, ref)       # vs plain fp32 oracle
    assert _rel(y, refq) < 1.5e-2, _rel(y, refq)   # vs oracle with bf16 round-trips at the autocast points


def test_forward_large_token_count_dispatch(tiny):
    """Forward batch 8 at 1024^2 token counts (M = 32768 rows) on the tiny model: every GEMM of the image stream goes
    through the persistent ping-pong kernel (>= 128 tiles) except proj_out (N = 64, scalar-scatter epilogue)."""
    from oracle import mmditx_ref as M
    cfg, engine, e, sd = tiny
    import copy
    g = torch.Generator().manual_seed(77)
    B, h, w, Nt = 8, 128, 128, 21
    cfg_big = copy.copy(cfg)
    cfg_big.pos_embed_max_size = 24
    x = torch.randn(B, 16, 48, 48, generator=g).half()     # 24x24 patches fit pos_embed_max_size 24: Ni = 576
    enc = torch.randn(B, Nt, cfg.joint_attention_dim, generator=g).bfloat16()
    pooled = torch.randn(B, cfg.pooled_projection_dim, generator=g).bfloat16()
    t = torch.full((B,), 700.0)
    # 64 samples x 576 tokens = 36864 rows >= 128 tiles of 256
    reps = 8
    xb, eb, pb = x.repeat(reps, 1, 1, 1), enc.repeat(reps, 1, 1), pooled.repeat(reps, 1)
    plan = e.plan(B * reps, 1, 48, 48, Nt, 1)
    y = plan.transformer_forward(xb.cuda(), t.repeat(reps).cuda(), eb.cuda(), pb.cuda())
    ref = M.mmdit_forward(sd, cfg, x.float(), t, enc.float(), pooled.float())
    assert _rel(y[:B], ref) < 2e-2
    for r in range(1, reps):
        assert torch.equal(y[r * B:(r + 1) * B], y[:B])   # identical samples -> identical results in every tile position


def test_forward_cfg_batch_order(tiny):
    """n_cfg == 2: forward batch is [negative, positive] on duplicated latents (sd3_5.py:409-413)."""
    from oracle import mmditx_ref as M
    cfg, engine, e, sd = tiny
    g = torch.Generator().manual_seed(11)
    B, h, w, Nt = 2, 16, 16, 9
    x = torch.randn(B, 16, h, w, generator=g).half()
    pe, ne = (torch.randn(B, Nt, 128, generator=g).bfloat16() for _ in range(2))
    pp, npl = (torch.randn(B, 128, generator=g).bfloat16() for _ in range(2))
    t = torch.tensor([500.0])
    plan = e.plan(B, 2, h, w, Nt, 4)
    y = plan.transformer_forward(x.cuda(), t.cuda(), ne.cuda(), npl.cuda(), pe.cuda(), pp.cuda())
    ref = M.mmdit_forward(sd, cfg, torch.cat([x, x]).float(), torch.full((2 * B,), 500.0), torch.cat([ne, pe]).float(),
                          torch.cat([npl, pp]).float())
    assert _rel(y, ref) < 2e-2
    assert _rel(y[:B], ref[:B]) < 2.5e-2 and _rel(y[B:], ref[B:]) < 2.5e-2


def test_denoise_step_and_replay_ratio(tiny):
    """SD3_5Adapter.forward parity + the GRPO invariant: replay on stored (x_i, x_{i+1}) reproduces the
    rollout log-prob exactly (ratio == 1, train_inference_consistency.md:20-29)."""
    from oracle import rollout_ref as R
    cfg, engine, e, sd = tiny
    g = torch.Generator().manual_seed(21)
    B, h, w, Nt = 2, 16, 16, 13
    x = torch.randn(B, 16, h, w, generator=g).half()
    pe, ne = (torch.randn(B, Nt, 128, generator=g).bfloat16() for _ in range(2))
    pp, npl = (torch.randn(B, 128, generator=g).bfloat16() for _ in range(2))
    eps = torch.randn(B, 16, h, w, generator=g)
    t, t_next, eta, smax, gs = torch.tensor(900.0), torch.tensor(750.0), 0.7, 0.9, 4.5
    plan = e.plan(B, 2, h, w, Nt, 4)
    want = ("next_latents", "next_latents_mean", "noise_pred", "std_dev_t", "dt")
    o = plan.denoise_step(x.cuda(), t, ne.cuda(), npl.cuda(), pe.cuda(), pp.cuda(), gs, t / 1000, t_next / 1000, eta, smax,
                          "Flow-SDE", noise=eps.cuda(), want=want)
    ref = R.forward_step(sd, cfg, t, t_next, x, pe, pp, ne, npl, gs, noise_level=eta, sigma_max=smax, variance_noise=eps)
    assert _rel(o.noise_pred, ref["noise_pred"]) < 3e-2
    # feed the ENGINE's noise_pred through the oracle scheduler: isolates the step from network error
    ref2 = R.forward_step(sd, cfg, t, t_next, x, pe, pp, None, None, 1.0, noise_level=eta, sigma_max=smax,
                          variance_noise=eps, denoiser=lambda *a: o.noise_pred.cpu())
    assert torch.equal(o.next_latents.cpu(), ref2["next_latents"])
    np.testing.assert_allclose(o.log_prob.cpu().numpy(), ref2["log_prob"].numpy(), rtol=1e-5)
    np.testing.assert_allclose(o.log_prob.cpu().numpy(), ref["log_prob"].numpy(), rtol=1e-3)  # north-star tolerance
    # replay
    o2 = plan.denoise_step(x.cuda(), torch.full((B,), 900.0), ne.cuda(), npl.cuda(), pe.cuda(), pp.cuda(), gs,
                           torch.full((B,), 0.9), torch.full((B,), 0.75), torch.full((B,), eta), smax, "Flow-SDE",
                           next_latents=o.next_storage, want=want)
    assert torch.equal(o2.log_prob, o.log_prob)  # ratio == exp(0) == 1.0


@pytest.mark.parametrize("storage,guidance", [(torch.float16, 1.0), (torch.bfloat16, 4.5)])
def test_rollout_tiny_vs_oracle(tiny, storage, guidance):
    from oracle import rollout_ref as R, scheduler_ref as S
    cfg, engine, e, sd = tiny
    B, h, w, Nt, N = 2, 16, 16, 13, 6
    g = torch.Generator().manual_seed(31)
    pe, ne = (torch.randn(B, Nt, 128, generator=g).bfloat16() for _ in range(2))
    pp, npl = (torch.randn(B, 128, generator=g).bfloat16() for _ in range(2))
    init, noise = R.draw_rollout_noise(B, 16, h, w, N, torch.bfloat16, torch.Generator().manual_seed(42))
    ts, sig = S.make_schedule(N, shift=3.0)
    nl = S.noise_levels(N, S.current_sde_steps([1, 2, 3], 2, 42, N), 0.7).tolist()
    cfg_on = guidance > 1.0
    ref = R.rollout(sd, cfg, pe, pp, ne if cfg_on else None, npl if cfg_on else None, guidance, init, noise, ts, sig, nl, storage)
    plan = e.plan(B, 2 if cfg_on else 1, h, w, Nt, N)
    lat, lp, fin = plan.rollout(ts.tolist(), sig.tolist(), nl, "Flow-SDE", guidance, init.cuda(), storage, noise.cuda(),
                                pe.cuda(), pp.cuda(), ne.cuda() if cfg_on else None, npl.cuda() if cfg_on else None)
    torch.cuda.synchronize()
    assert lat.shape == (N + 1, B, 16, h, w) and lat.dtype == storage
    assert torch.equal(lat[0].cpu(), S.cast_latents(init, storage))
    assert torch.equal(lat[-1], fin)
    for i in range(1, N + 1):
        r = _rel(lat[i], ref["all_latents"][i])
        assert r < 2e-2, (i, r)
    lp_c, lp_r = lp.cpu(), ref["log_probs"]
    sde = [i for i in range(N) if nl[i] > 0]
    assert len(sde) == 2
    for i in range(N):
        if i in sde:
            np.testing.assert_allclose(lp_c[i].numpy(), lp_r[i].numpy(), rtol=1e-3)
        else:
            assert torch.isnan(lp_c[i]).all()
    # kept positions only (TrajectoryCollector semantics)
    keep = [1, 2, 3]
    lat2, lp2, fin2 = plan.rollout(ts.tolist(), sig.tolist(), nl, "Flow-SDE", guidance, init.cuda(), storage, noise.cuda(),
                                   pe.cuda(), pp.cuda(), ne.cuda() if cfg_on else None, npl.cuda() if cfg_on else None,
                                   keep_positions=keep)
    assert torch.equal(lat2, lat[keep]) and torch.equal(fin2, fin)  # deterministic: bit-identical re-run


def test_full_size_model_256(eng_full=None):
    """SD3.5-medium shapes (24 blocks, 13 dual, D=1536), config A of BASELINE.json (256x256, B=1) vs the fp32
    oracle (on the GPU, fp32), inside 1.5 x its bf16 band."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import mmditx_ref as M
    engine, e, sd = _setup(M.SD35_MEDIUM, std=0.02)
    try:
        g = torch.Generator().manual_seed(4321)
        B, h, w, Nt = 1, 32, 32, 333
        x = torch.randn(B, 16, h, w, generator=g).half()
        enc = torch.randn(B, Nt, 4096, generator=g).bfloat16()
        pooled = torch.randn(B, 2048, generator=g).bfloat16()
        t = torch.tensor([900.0])
        plan = e.plan(B, 1, h, w, Nt, 4)
        y = plan.transformer_forward(x.cuda(), t.cuda(), enc.cuda(), pooled.cuda())
        torch.cuda.synchronize()
        from _gpu_oracle import check_in_band
        check_in_band("SD3.5-medium full depth, 256^2 (config A forward)", y, M.mmdit_forward, sd, M.SD35_MEDIUM, x.float(), t, enc.float(), pooled.float())
    finally:
        e.close()


def test_sd35_large_width_blocks():
    """The same adapter serves SD3.5-LARGE (38 blocks, 38 heads x 64 = 2432 wide, no dual-attention layers, pos_embed_max_size 192):
    here its width with 2 blocks (the last one context_pre_only) at 512x512, B = 2, vs the fp32 oracle."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import mmditx_ref as M
    cfg = M.tiny_config(num_layers=2, num_heads=38, dual_layers=(), joint_attention_dim=4096, pooled_projection_dim=2048,
                        pos_embed_max_size=192)
    engine, e, sd = _setup(cfg, std=0.02, seed=99)
    try:
        g = torch.Generator().manual_seed(8)
        B, h, w, Nt = 2, 64, 64, 77
        x = torch.randn(B, 16, h, w, generator=g).half()
        enc = torch.randn(B, Nt, 4096, generator=g).bfloat16()
        pooled = torch.randn(B, 2048, generator=g).bfloat16()
        t = torch.tensor([650.0, 120.0])
        y = e.plan(B, 1, h, w, Nt, 1).transformer_forward(x.cuda(), t.cuda(), enc.cuda(), pooled.cuda())
        from _gpu_oracle import check_in_band
        check_in_band("SD3.5-large width, 2 blocks, 512^2", y, M.mmdit_forward, sd, cfg, x.float(), t, enc.float(), pooled.float())
    finally:
        e.close()


def test_large_norm_weights_fall_back_to_running_max():
    """q/k RMSNorm weights of ~5 put the proven score bound (11.8 * 25) far above 60: the engine must keep the running-max
    softmax for those layers (scores of +-100 in the log2 domain would overflow the static-bound kernel)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mi355_flow import engine
    from oracle import mmditx_ref as M
    cfg = M.tiny_config(num_layers=2, num_heads=2, dual_layers=(0,), joint_attention_dim=128, pooled_projection_dim=128,
                        pos_embed_max_size=24)
    sd = M.make_synthetic_state_dict(cfg, seed=7, std=0.08)
    for k_ in sd:
        if ".norm_" in k_:
            sd[k_] = sd[k_] * 5.0
    sd = {k_: v.bfloat16().float() for k_, v in sd.items()}
    e = engine.Engine(engine.TransformerConfig(num_layers=2, num_heads=2, joint_attention_dim=128, pooled_projection_dim=128,
                                               pos_embed_max_size=24, dual_layers=(0,)))
    e.bind_state_dict({k_: v.cuda() for k_, v in sd.items()})
    e.ready()
    try:
        g = torch.Generator().manual_seed(2)
        B, h, w, Nt = 2, 16, 16, 13
        x = torch.randn(B, 16, h, w, generator=g).half()
        enc = torch.randn(B, Nt, 128, generator=g).bfloat16()
        pooled = torch.randn(B, 128, generator=g).bfloat16()
        t = torch.tensor([800.0, 300.0])
        y = e.plan(B, 1, h, w, Nt, 1).transformer_forward(x.cuda(), t.cuda(), enc.cuda(), pooled.cuda())
        ref = M.mmdit_forward(sd, cfg, x.float(), t, enc.float(), pooled.float())
        assert torch.isfinite(y.float()).all()
        assert _rel(y, ref) < 3e-2        # very peaked softmaxes: bf16 q/k rounding shows a little more
    finally:
        e.close()


def test_large_norm_weights_at_different_channels_keep_the_static_kernel():
    """The proven bound is 64 * max_d |w_q[d] * w_k[d]| (Cauchy-Schwarz on the weighted heads), not 64 * max|w_q| * max|w_k|: q weights of 4
    at even channels and k weights of 4 at odd channels (0.5 elsewhere) give products of 2 -- every launch stays on the static-softmax
    kernel -- and the forward, run on inputs scaled so that single channels dominate the heads, still matches the oracle."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mi355_flow import engine
    from oracle import mmditx_ref as M
    cfg = M.tiny_config(num_layers=2, num_heads=2, dual_layers=(0,), joint_attention_dim=128, pooled_projection_dim=128,
                        pos_embed_max_size=24)
    sd = M.make_synthetic_state_dict(cfg, seed=7, std=0.08)
    even = (torch.arange(64) % 2 == 0).float()
    for k_ in sd:
        if ".norm_" in k_:
            is_q = "norm_q" in k_ or "norm_added_q" in k_
            big = even if is_q else 1.0 - even
            sd[k_] = (4.0 * big + 0.5 * (1.0 - big)) * torch.sign(sd[k_] + 1e-3)
    sd = {k_: v.bfloat16().float() for k_, v in sd.items()}
    e = engine.Engine(engine.TransformerConfig(num_layers=2, num_heads=2, joint_attention_dim=128, pooled_projection_dim=128,
                                               pos_embed_max_size=24, dual_layers=(0,)))
    e.bind_state_dict({k_: v.cuda() for k_, v in sd.items()})
    e.ready()
    try:
        info = e.attention_info()
        assert info["static"] == info["total"] == 3, info
        assert info["max_bound"] < 60.0 * 0.5, info
        g = torch.Generator().manual_seed(3)
        B, h, w, Nt = 2, 16, 16, 13
        x = (torch.randn(B, 16, h, w, generator=g) * 3.0).half()
        enc = (torch.randn(B, Nt, 128, generator=g) * 3.0).bfloat16()
        pooled = torch.randn(B, 128, generator=g).bfloat16()
        t = torch.tensor([800.0, 300.0])
        y = e.plan(B, 1, h, w, Nt, 1).transformer_forward(x.cuda(), t.cuda(), enc.cuda(), pooled.cuda())
        ref = M.mmdit_forward(sd, cfg, x.float(), t, enc.float(), pooled.float())
        assert torch.isfinite(y.float()).all()
        assert _rel(y, ref) < 3e-2
    finally:
        e.close()


@pytest.mark.parametrize("B,h,w,Nt", [(1, 2, 2, 1), (1, 2, 4, 3), (5, 6, 2, 2)])
def test_forward_degenerate_sizes(tiny, B, h, w, Nt):
    """Smallest legal shapes: one image token (2x2 latent), one text token, odd batch -- every kernel on its ragged / scalar tail path."""
    from oracle import mmditx_ref as M
    cfg, engine, e, sd = tiny
    g = torch.Generator().manual_seed(B * 100 + h * 10 + Nt)
    x = torch.randn(B, 16, h, w, generator=g).half()
    enc = torch.randn(B, Nt, cfg.joint_attention_dim, generator=g).bfloat16()
    pooled = torch.randn(B, cfg.pooled_projection_dim, generator=g).bfloat16()
    t = torch.full((B,), 500.0)
    y = e.plan(B, 1, h, w, Nt, 1).transformer_forward(x.cuda(), t.cuda(), enc.cuda(), pooled.cuda())
    ref = M.mmdit_forward(sd, cfg, x.float(), t, enc.float(), pooled.float())
    assert torch.isfinite(y.float()).all()
    assert _rel(y, ref) < 2e-2
