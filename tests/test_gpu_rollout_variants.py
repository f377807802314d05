"""More GPU parity cases for the rollout: every dynamics type, eval (ODE) mode, non-square latents,
fp32 / bf16 storage, and the hipGraph replay path against eager launches when the SDE-step
selection (noise levels) changes between epochs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.fixture(scope="module")
def ctx():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mi355_flow import engine
    from oracle import mmditx_ref as M
    cfg = M.tiny_config(num_layers=2, num_heads=2, dual_layers=(0,), joint_attention_dim=128, pooled_projection_dim=128,
                        pos_embed_max_size=24)
    sd = {k: v.bfloat16().float() for k, v in M.make_synthetic_state_dict(cfg, seed=99, std=0.08).items()}
    e = engine.Engine(engine.TransformerConfig(num_layers=2, num_heads=2, joint_attention_dim=128, pooled_projection_dim=128,
                                               pos_embed_max_size=24, dual_layers=(0,)))
    e.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    yield cfg, e, sd
    e.close()


def _inputs(B, h, w, Nt, N, seed):
    from oracle import rollout_ref as R
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16()
    pe, pp, ne, npl = mk(B, Nt, 128), mk(B, 128), mk(B, Nt, 128), mk(B, 128)
    init, noise = R.draw_rollout_noise(B, 16, h, w, N, torch.bfloat16, torch.Generator().manual_seed(seed + 1))
    return pe, pp, ne, npl, init, noise


@pytest.mark.parametrize("dyn,storage,hw", [("Dance-SDE", torch.float16, (16, 16)), ("CPS", torch.bfloat16, (16, 16)),
                                            ("ODE", torch.float16, (8, 24)), ("Flow-SDE", torch.float32, (24, 8))])
def test_rollout_dynamics_vs_oracle(ctx, dyn, storage, hw):
    from oracle import rollout_ref as R, scheduler_ref as S
    cfg, e, sd = ctx
    B, Nt, N = 2, 7, 5
    h, w = hw
    pe, pp, ne, npl, init, noise = _inputs(B, h, w, Nt, N, 100 + len(dyn))
    ts, sig = S.make_schedule(N, shift=3.0)
    nl = [0.0, 0.7, 0.7, 0.0, 0.0] if dyn != "ODE" else [0.0] * N
    ref = R.rollout(sd, cfg, pe, pp, ne, npl, 3.0, init, noise, ts, sig, nl, storage, dynamics_type=dyn)
    plan = e.plan(B, 2, h, w, Nt, N)
    lat, lp, fin = plan.rollout(ts.tolist(), sig.tolist(), nl, dyn, 3.0, init.cuda(), storage, noise.cuda(), pe.cuda(), pp.cuda(),
                                ne.cuda(), npl.cuda())
    for i in range(N + 1):
        assert _rel(lat[i], ref["all_latents"][i]) < 2e-2, (dyn, i)
    for i in range(N):
        if nl[i] > 0:
            np.testing.assert_allclose(lp[i].cpu().numpy(), ref["log_probs"][i].numpy(), rtol=2e-3 if dyn == "CPS" else 1e-3)
        else:
            assert torch.isnan(lp[i]).all()


def test_graph_replay_equals_eager_across_epochs(ctx):
    """The captured hipGraph must honour per-epoch changes that do not change the launch sequence:
    which steps are SDE steps (device-side scalars), new prompts / noise (staging copies), and must be
    rebuilt when guidance or the step count changes."""
    from mi355_flow import _lib
    from oracle import scheduler_ref as S
    cfg, e, sd = ctx
    lib = _lib.load()
    B, h, w, Nt, N = 2, 16, 16, 7, 6
    ts, sig = S.make_schedule(N, shift=3.0)
    plan = e.plan(B, 2, h, w, Nt, N)
    runs = []
    for epoch, (nl, gs) in enumerate([([0, .7, 0, 0, 0, 0], 4.5), ([0, 0, 0, .7, 0, 0], 4.5), ([0, .7, .7, 0, 0, 0], 4.5),
                                      ([0, 0, .7, 0, 0, 0], 2.0)]):
        pe, pp, ne, npl, init, noise = _inputs(B, h, w, Nt, N, 500 + epoch)
        args = (ts.tolist(), sig.tolist(), [float(x) for x in nl], "Flow-SDE", gs, init.cuda(), torch.float16, noise.cuda(),
                pe.cuda(), pp.cuda(), ne.cuda(), npl.cuda())
        lib.mi355_tune_set(2, 1)
        g_out = plan.rollout(*args)          # 1st call warms up eagerly, later calls replay / re-capture
        g_out2 = plan.rollout(*args)
        lib.mi355_tune_set(2, 0)
        e_out = plan.rollout(*args)
        lib.mi355_tune_set(2, 1)
        for a, b, c in zip(g_out, g_out2, e_out):
            assert torch.equal(a, c, ) or (torch.isnan(a) == torch.isnan(c)).all() and torch.equal(torch.nan_to_num(a), torch.nan_to_num(c))
            assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
        runs.append(g_out)
    assert not torch.equal(runs[0][0], runs[1][0])


def _same(a, b):
    return torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)) and bool((torch.isnan(a) == torch.isnan(b)).all())


@pytest.mark.parametrize("n_cfg", [1, 2])
def test_two_stream_forward_is_bit_identical(ctx, n_cfg):
    """mi355_tune_set key 8: the text-stream chain of every block runs on a plan-owned side stream (fork after each joint attention, join
    before the next; graph edges inside the captured rollout).  Same kernels on the same operands: the rollout must be bit-identical to the
    single-stream order -- eagerly and as a replayed hipGraph, from the caller's default stream and from a user stream, repeatedly (a
    missing dependency would show as run-to-run differences)."""
    from mi355_flow import _lib
    from oracle import scheduler_ref as S
    cfg, e, sd = ctx
    lib = _lib.load()
    B, h, w, Nt, N = 3, 16, 24, 9, 5
    ts, sig = S.make_schedule(N, shift=3.0)
    nl = [0.0, 0.7, 0.0, 0.7, 0.0]
    pe, pp, ne, npl, init, noise = _inputs(B, h, w, Nt, N, 900 + n_cfg)
    gs = 4.5 if n_cfg == 2 else 1.0
    args = (ts.tolist(), sig.tolist(), nl, "Flow-SDE", gs, init.cuda(), torch.float16, noise.cuda(), pe.cuda(), pp.cuda()) + \
           ((ne.cuda(), npl.cuda()) if n_cfg == 2 else ())
    plan = e.plan(B, n_cfg, h, w, Nt, N)
    try:
        lib.mi355_tune_set(8, 0)
        lib.mi355_tune_set(2, 0)
        base = plan.rollout(*args)
        torch.cuda.synchronize()
        lib.mi355_tune_set(8, 1)
        # key 10: fork after the joint attention (0) / after the block's last attention (1); key 11: third stream for the image V^T and
        # the dual attention's projections
        for late, third in ((0, 0), (1, 0), (0, 1), (1, 1)):
            lib.mi355_tune_set(10, late)
            lib.mi355_tune_set(11, third)
            lib.mi355_tune_set(2, 0)
            for rep in range(3):                                   # eager, two streams
                out = plan.rollout(*args)
                assert all(_same(a, b) for a, b in zip(out, base)), ("eager", late, third, rep)
            lib.mi355_tune_set(2, 1)
            for rep in range(4):                                   # (re)captured with fork / join edges, then replayed
                out = plan.rollout(*args)
                assert all(_same(a, b) for a, b in zip(out, base)), ("graph", late, third, rep)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                          # caller on a non-default stream
            out = plan.rollout(*args)
            lib.mi355_tune_set(2, 0)
            out_e = plan.rollout(*args)
        side.synchronize()
        assert all(_same(a, b) for a, b in zip(out, base)) and all(_same(a, b) for a, b in zip(out_e, base))
        # auto mode: rows above the threshold keep the single stream, below take two -- same results either way
        lib.mi355_tune_set(8, 2)
        for rows in (1, 1 << 20):
            lib.mi355_tune_set(9, rows)
            lib.mi355_tune_set(2, 1)
            out = plan.rollout(*args)
            out2 = plan.rollout(*args)
            assert all(_same(a, b) for a, b in zip(out, base)) and all(_same(a, b) for a, b in zip(out2, base)), rows
        # single steps (mi355_denoise_step) take the same path
        lib.mi355_tune_set(8, 1)
        x = init.cuda().half()
        kw = dict(noise=noise[0].cuda(), compute_log_prob=True, want=("next_latents_mean", "noise_pred"))
        tt = torch.full((B,), float(ts[0]), device="cuda")
        step_args = (x, tt, pe.cuda(), pp.cuda(), None, None, 1.0) if n_cfg == 1 else (x, tt, ne.cuda(), npl.cuda(), pe.cuda(), pp.cuda(), gs)
        o2 = plan.denoise_step(*step_args, float(sig[0]), float(sig[1]), 0.7, float(sig[1]), "Flow-SDE", **kw)
        lib.mi355_tune_set(8, 0)
        o1 = plan.denoise_step(*step_args, float(sig[0]), float(sig[1]), 0.7, float(sig[1]), "Flow-SDE", **kw)
        assert torch.equal(o1.noise_pred, o2.noise_pred) and torch.equal(o1.log_prob, o2.log_prob)
    finally:
        lib.mi355_tune_set(8, 2)                                   # the shipped defaults
        lib.mi355_tune_set(9, 32768)
        lib.mi355_tune_set(10, 2)
        lib.mi355_tune_set(11, 0)
        lib.mi355_tune_set(2, 1)


def test_adapter_error_paths(ctx):
    from mi355_flow.adapter import SD3_5NativeAdapter
    from mi355_flow.engine import TransformerConfig
    cfg, e, sd = ctx
    ad = SD3_5NativeAdapter({k: v.cuda() for k, v in sd.items()},
                            TransformerConfig(num_layers=2, num_heads=2, joint_attention_dim=128, pooled_projection_dim=128,
                                              pos_embed_max_size=24, dual_layers=(0,)))
    pe, pp = torch.zeros(1, 5, 128).cuda().bfloat16(), torch.zeros(1, 128).cuda().bfloat16()
    # the LoRA `scale` is honoured through the weight binding (a no-op without LoRA layers, as in diffusers); anything else raises
    ad.inference(height=128, width=128, num_inference_steps=2, guidance_scale=1.0, prompt_embeds=pe, pooled_prompt_embeds=pp,
                 joint_attention_kwargs={"scale": 0.5})
    with pytest.raises(NotImplementedError, match="ip_adapter"):
        ad.inference(height=128, width=128, num_inference_steps=2, guidance_scale=1.0, prompt_embeds=pe, pooled_prompt_embeds=pp,
                     joint_attention_kwargs={"ip_adapter_image_embeds": None})
    with pytest.raises(RuntimeError, match="text encoders"):
        ad.inference(prompt=["a cat"], height=128, width=128, num_inference_steps=2, guidance_scale=1.0)
    with pytest.raises((RuntimeError, ValueError), match="GPU"):        # CPU tensors are rejected by the engine wrappers: no CPU fallback
        ad.forward(t=torch.tensor(900.0), t_next=torch.tensor(750.0), latents=torch.zeros(1, 16, 16, 16), prompt_embeds=pe,
                   pooled_prompt_embeds=pp, noise_level=0.0)
    with pytest.raises(RuntimeError, match="exceeds pos_embed_max_size"):
        ad.inference(height=8 * 2 * 30, width=128, num_inference_steps=2, guidance_scale=1.0, prompt_embeds=pe, pooled_prompt_embeds=pp)
    # CFG requested without negatives: warning + CFG disabled (reference behaviour, sd3_5.py:212-214)
    out = ad.inference(height=128, width=128, num_inference_steps=2, guidance_scale=4.5, prompt_embeds=pe, pooled_prompt_embeds=pp,
                       trajectory_indices=None, compute_log_prob=False)
    assert len(out) == 1 and out[0].all_latents is None and out[0].log_probs is None
    with pytest.raises(KeyError):
        ad.engine.bind_state_dict({"proj_out.weight": sd["proj_out.weight"].cuda()})
    ad.engine.close()
