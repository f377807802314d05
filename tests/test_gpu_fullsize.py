"""GPU parity at the REAL configurations of BASELINE.json (full SD3.5-medium: 24 blocks, 13 dual, D = 1536) against the
fp32 CPU oracle on identical bf16-rounded weights, prompts and noise:

  * config A end to end: 256x256, 4 Euler/SDE steps, B = 1 -- per-step latents, rollout log-prob (rtol 1e-3, north star);
  * replay log-prob ENGINE vs ORACLE on the engine's stored (x_i, x_{i+1}) (SURVEY.md 8(a) item iii, 8(d) tolerance 1e-3):
    the number that sizes the train/inference-consistency hazard when optimize() replays on a different implementation;
  * config B shape: one full-model forward at 1024x1024 (S = 4096 + 333 = 4429 joint tokens), B = 1.

Reference control flow: src/flow_factory/models/stable_diffusion/sd3_5.py:258-304 (loop), :352-448 (forward).
The oracle model body is an unpinned restatement of diffusers' SD3Transformer2DModel (oracle/mmditx_ref.py header).
CPU cost: ~0.9 TFLOP per 256^2 forward (seconds), 11.25 TFLOP for the 1024^2 forward (~1 min on the box's host cores).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_TEXT = 333


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.fixture(scope="module")
def full():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mi355_flow import engine
    from mi355_flow.weights import synthetic_state_dict
    from oracle import mmditx_ref as M
    cfg_e = engine.TransformerConfig()
    # draw on the GPU (the CPU generator needs ~1 min for 2.5 B values); both sides see the same bf16-rounded values
    sd_gpu = synthetic_state_dict(cfg_e, device="cuda", seed=1234, dtype=torch.bfloat16)
    e = engine.Engine(cfg_e)
    e.bind_state_dict(sd_gpu)
    e.ready()
    sd = {k: v.float().cpu() for k, v in sd_gpu.items()}
    del sd_gpu
    torch.cuda.empty_cache()
    torch.set_num_threads(max(torch.get_num_threads(), 1))
    yield e, sd, M.SD35_MEDIUM
    e.close()


def test_config_a_rollout_and_replay_vs_oracle(full):
    """BASELINE.json configs[0]: 256^2, N = 4, B = 1, Flow-SDE eta 0.7, one SDE step of [1,2,3] (seed 42), fp16 storage."""
    from oracle import mmditx_ref as M, rollout_ref as R, scheduler_ref as S
    e, sd, cfg = full
    B, h, w, N = 1, 32, 32, 4
    g = torch.Generator().manual_seed(4321)
    pe = torch.randn(B, N_TEXT, 4096, generator=g).bfloat16()
    pp = torch.randn(B, 2048, generator=g).bfloat16()
    init, noise = R.draw_rollout_noise(B, 16, h, w, N, torch.bfloat16, torch.Generator().manual_seed(42))
    ts, sig = S.make_schedule(N, shift=3.0)
    sde = S.current_sde_steps([1, 2, 3], 1, 42, N)
    nl = S.noise_levels(N, sde, 0.7).tolist()
    ref = R.rollout(sd, cfg, pe, pp, None, None, 1.0, init, noise, ts, sig, nl, torch.float16)
    plan = e.plan(B, 1, h, w, N_TEXT, N)
    lat, lp, fin = plan.rollout(ts.tolist(), sig.tolist(), nl, "Flow-SDE", 1.0, init.cuda(), torch.float16, noise.cuda(),
                                pe.cuda(), pp.cuda())
    torch.cuda.synchronize()
    assert torch.equal(lat[0].cpu(), S.cast_latents(init, torch.float16))
    worst = 0.0
    for i in range(1, N + 1):
        r = _rel(lat[i], ref["all_latents"][i])
        worst = max(worst, r)
        assert r < 2e-2, (i, r)
    steps = [i for i in range(N) if nl[i] > 0]
    assert len(steps) == 1
    i = steps[0]
    np.testing.assert_allclose(lp[i].cpu().numpy(), ref["log_probs"][i].numpy(), rtol=1e-3)   # rollout log-prob, north star
    # how much of that distance is bf16 itself: the same oracle with bf16 round-trips where the reference's bf16 network rounds
    # (what a diffusers bf16 run computes, up to accumulation order) against its own fp32 self -- the engine should sit in that band
    refq = R.rollout(sd, cfg, pe, pp, None, None, 1.0, init, noise, ts, sig, nl, torch.float16, quant=M.bf16_round)
    band = max(_rel(refq["all_latents"][j], ref["all_latents"][j]) for j in range(1, N + 1))
    worst_q = max(_rel(lat[j], refq["all_latents"][j]) for j in range(1, N + 1))
    print(f"config A: bf16-emulating oracle vs fp32 oracle, worst per-step latent rel-L2 {band:.3e}; engine vs bf16-emulating oracle {worst_q:.3e}; "
          f"engine vs fp32 oracle {worst:.3e}")
    assert worst < 3.0 * band + 2e-3, (worst, band)          # the engine is no further from fp32 than a bf16 network is (x3 slack)

    # ---- replay (what optimize() computes, trainers/grpo.py:229-263) on the ENGINE's stored transition, evaluated by the ORACLE
    x_i, x_n = lat[i].cpu(), lat[i + 1].cpu()
    t = ts[i]
    t_next = ts[i + 1] if i + 1 < N else torch.tensor(0.0)
    o = R.forward_step(sd, cfg, t, t_next, x_i, pe, pp, None, None, 1.0, noise_level=nl[i], sigma_max=float(sig[1]),
                       next_latents=x_n.float())
    lp_engine = lp[i].cpu()
    ratio = torch.exp(o["log_prob"] - lp_engine)
    print(f"config A: worst per-step latent rel-L2 {worst:.3e}; replay log-prob oracle {o['log_prob'].tolist()} vs engine rollout "
          f"{lp_engine.tolist()}; |ratio-1| = {float((ratio - 1).abs().max()):.3e}")
    np.testing.assert_allclose(o["log_prob"].numpy(), lp_engine.numpy(), rtol=1e-3)
    assert float((ratio - 1).abs().max()) < 1e-3      # abs(ratio - 1) <= 1e-3 engine-vs-oracle
    # ... and the engine's own replay of the same transition is bit-identical (ratio == 1.0 exactly)
    o2 = plan.denoise_step(lat[i], ts[i].reshape(1).expand(B), pe.cuda(), pp.cuda(), None, None, 1.0,
                           (ts[i].double() / 1000).float().reshape(1).expand(B), (t_next.double() / 1000).float().reshape(1).expand(B),
                           torch.full((B,), nl[i]), float(sig[1]), "Flow-SDE", next_latents=lat[i + 1])
    assert torch.equal(o2.log_prob, lp[i])


def test_config_b_forward_1024_vs_oracle(full):
    """BASELINE.json configs[1] shape: 1024^2 (latent 128x128 -> 4096 image tokens + 333 text tokens), B = 1."""
    from oracle import mmditx_ref as M
    e, sd, cfg = full
    g = torch.Generator().manual_seed(99)
    B, h, w = 1, 128, 128
    x = torch.randn(B, 16, h, w, generator=g).half()
    enc = torch.randn(B, N_TEXT, 4096, generator=g).bfloat16()
    pooled = torch.randn(B, 2048, generator=g).bfloat16()
    t = torch.tensor([750.0])
    plan = e.plan(B, 1, h, w, N_TEXT, 1)
    y = plan.transformer_forward(x.cuda(), t.cuda(), enc.cuda(), pooled.cuda())
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = M.mmdit_forward(sd, cfg, x.float(), t, enc.float(), pooled.float())
    r = _rel(y, ref)
    print(f"config B forward (S = 4429) rel-L2 vs fp32 oracle: {r:.3e}")
    assert torch.isfinite(y.float()).all()
    assert r < 3e-2, r
    # the same sample inside a batch of 2 (different tile positions / workgroup mapping) gives the same result
    plan2 = e.plan(2, 1, h, w, N_TEXT, 1)
    y2 = plan2.transformer_forward(torch.cat([x, x]).cuda(), t.repeat(2).cuda(), torch.cat([enc, enc]).cuda(),
                                   torch.cat([pooled, pooled]).cuda())
    assert _rel(y2[0:1], ref) < 3e-2 and _rel(y2[1:2], ref) < 3e-2


def test_forward_is_bitwise_independent_of_the_batch_a_sample_sits_in(full):
    """optimize() may replay stored transitions in micro-batches of another size than the rollout used (trainers/grpo.py:229-263).  The
    GEMMs then run other kernels (the persistent 256x256 ping-pong kernel from 128 tiles up, the 128x128 kernel below) -- same K order, same
    epilogue arithmetic: a sample's velocity must not depend on its batch, bit for bit, or ratio == 1 would hold only at equal batch sizes."""
    e, sd, cfg = full
    g = torch.Generator().manual_seed(5)
    h = w = 64                                                             # 512^2: 1024 image tokens per sample
    x = torch.randn(1, 16, h, w, generator=g).half().cuda()
    pe, pp = torch.randn(1, N_TEXT, 4096, generator=g).bfloat16().cuda(), torch.randn(1, 2048, generator=g).bfloat16().cuda()
    t = torch.tensor([900.0]).cuda()
    y1 = e.plan(1, 1, h, w, N_TEXT, 1).transformer_forward(x, t, pe, pp)
    for B in (2, 8):
        yb = e.plan(B, 1, h, w, N_TEXT, 1).transformer_forward(x.repeat(B, 1, 1, 1), t.repeat(B), pe.repeat(B, 1, 1), pp.repeat(B, 1))
        assert torch.equal(yb, y1.expand_as(yb)), B


def test_forward_is_bitwise_independent_of_the_batch_at_the_bench_shape(full):
    """The same invariant at BASELINE.json configs[1]'s own geometry and the bench's batch: 1024^2, B = 8 (32 768 image rows: every GEMM on
    the persistent ping-pong kernel, the two-stream forward with the late fork) against B = 1 (4096 rows: early fork, 128x128 tiles for the
    narrow grids).  Eight DIFFERENT samples; each must equal its own single-sample forward bit for bit."""
    e, sd, cfg = full
    g = torch.Generator().manual_seed(6)
    h = w = 128
    B = 8
    x = torch.randn(B, 16, h, w, generator=g).half().cuda()
    pe, pp = torch.randn(B, N_TEXT, 4096, generator=g).bfloat16().cuda(), torch.randn(B, 2048, generator=g).bfloat16().cuda()
    t = torch.tensor([900.0]).cuda()
    y8 = e.plan(B, 1, h, w, N_TEXT, 1).transformer_forward(x, t.repeat(B), pe, pp)
    p1 = e.plan(1, 1, h, w, N_TEXT, 1)
    for b in (0, 3, 7):
        y1 = p1.transformer_forward(x[b:b + 1], t, pe[b:b + 1], pp[b:b + 1])
        assert torch.equal(y8[b:b + 1], y1), b


def test_config_b_forward_sits_in_the_bf16_band(full):
    """How far from the fp32 oracle may a bf16 network be at S = 4429?  The oracle with bf16 round-trips wherever the reference's bf16 module
    materialises a tensor (`quant=M.bf16_round`: what a diffusers bf16 run computes up to accumulation order) against its own fp32 self is
    the band; the engine must sit within 3x that band (+2e-3) of the fp32 oracle -- i.e. its 1.4e-2 at this shape is bf16 rounding through 24
    blocks, not an implementation error (the distance to the bf16-emulating oracle is printed for the record)."""
    from oracle import mmditx_ref as M
    e, sd, cfg = full
    g = torch.Generator().manual_seed(99)                                   # the inputs of test_config_b_forward_1024_vs_oracle
    B, h, w = 1, 128, 128
    x = torch.randn(B, 16, h, w, generator=g).half()
    enc = torch.randn(B, N_TEXT, 4096, generator=g).bfloat16()
    pooled = torch.randn(B, 2048, generator=g).bfloat16()
    t = torch.tensor([750.0])
    y = e.plan(B, 1, h, w, N_TEXT, 1).transformer_forward(x.cuda(), t.cuda(), enc.cuda(), pooled.cuda())
    with torch.no_grad():
        ref = M.mmdit_forward(sd, cfg, x.float(), t, enc.float(), pooled.float())
        refq = M.mmdit_forward(sd, cfg, x.float(), t, enc.float(), pooled.float(), quant=M.bf16_round)
    band, r, rq = _rel(refq, ref), _rel(y, ref), _rel(y, refq)
    print(f"config B forward (S = 4429): bf16-emulating oracle vs fp32 oracle {band:.3e}; engine vs fp32 {r:.3e}; engine vs bf16-emulating {rq:.3e}")
    assert r < 3.0 * band + 2e-3, (r, band)


def test_config_a_replay_gradients_vs_oracle_autograd(full):
    """SURVEY.md 8(f) N1 at the real geometry: the differentiable replay step on full SD3.5-medium (256^2, B = 1) -- grad-mode log-prob
    bit-identical to the no-grad replay (ratio == 1), weight gradients of the attention projections of blocks 0 / 12 / 23 (the
    reference's default target modules, models/abc.py:382-385) vs torch autograd through the fp32 oracle on the host cores."""
    from mi355_flow.adapter import SD3_5NativeAdapter
    from mi355_flow.engine import TransformerConfig
    from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler
    from mi355_flow.weights import module_from_state_dict
    from test_gpu_backward import _cos, _inputs, _oracle_loss
    e, sd, cfg = full
    mod = module_from_state_dict({k: v.clone().cuda() for k, v in sd.items()})          # fp32 master copy of the bf16-rounded values
    picks = ("transformer_blocks.0.", "transformer_blocks.12.", "transformer_blocks.23.")
    for n, p in mod.named_parameters():
        p.requires_grad_(n.startswith(picks) and any(k in n for k in (".to_q.", ".to_k.", ".to_v.", ".to_out.0.")))
    sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1, seed=42, shift=3.0)
    ad = SD3_5NativeAdapter(mod, TransformerConfig(), sched, latent_storage_dtype="fp16")
    ad.rollout()
    try:
        B, h, w = 1, 32, 32
        g = torch.Generator().manual_seed(77)
        inp = dict(x=torch.randn(B, 16, h, w, generator=g).half(), pe=torch.randn(B, N_TEXT, 4096, generator=g).bfloat16(),
                   pp=torch.randn(B, 2048, generator=g).bfloat16(), wlp=torch.ones(B), wnp=torch.zeros(B, 16, h, w))
        t, t_next, eta, smax = 900.0, 750.0, 0.7, 0.9
        sched.set_timesteps(4)
        # a plausible stored transition: x' = mean + noise comes from the engine's own rollout step
        with torch.no_grad():
            o0 = ad.forward(t=torch.full((B,), t), t_next=torch.full((B,), t_next), latents=inp["x"].cuda(), prompt_embeds=inp["pe"].cuda(),
                            pooled_prompt_embeds=inp["pp"].cuda(), guidance_scale=1.0, noise_level=eta, return_kwargs=["next_latents", "log_prob"])
        inp["x1"] = o0.next_latents.half().cpu()
        kw = dict(t=torch.full((B,), t), t_next=torch.full((B,), t_next), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
                  prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(), guidance_scale=1.0, noise_level=eta,
                  compute_log_prob=True, return_kwargs=["log_prob", "dt"])
        with torch.no_grad():
            ref = ad.forward(**kw)
        out = ad.forward(**kw)
        assert torch.equal(out.log_prob.detach(), ref.log_prob) and torch.equal(ref.log_prob, o0.log_prob)      # rollout == replay == grad replay
        out.log_prob.sum().backward()
        lp_ref, g_ref = _oracle_loss(mod, cfg, inp, 1.0, t, t_next, eta, smax, 0.0)
        np.testing.assert_allclose(out.log_prob.detach().cpu().numpy(), lp_ref.numpy(), rtol=1e-3)
        worst, n = 0.0, 0
        for name, prm in mod.named_parameters():
            if not prm.requires_grad:
                continue
            r = _rel(prm.grad, g_ref[name])
            worst, n = max(worst, r), n + 1
            assert r < 6e-2 and _cos(prm.grad, g_ref[name]) > 0.99, (name, r)
        print(f"full SD3.5-medium replay gradients: {n} tensors, worst rel-L2 vs fp32 oracle autograd {worst:.3e}")
        assert n == 8 * 3 + 8 * 2
    finally:
        ad.engine.close()
