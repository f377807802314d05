"""GPU tests of the differentiable replay step (reference src/flow_factory/trainers/grpo.py:185-342), through the
C ABI (`mi355_denoise_step_train` / `mi355_denoise_step_backward`, `mi355_op_attention_fwd_bwd`):

  * flash-attention backward vs torch autograd of fp32 SDPA on the same bf16 inputs (rel-L2 <= 2e-2: P and dZ are rounded to bf16
    before their MFMAs, like every flash backward);
  * the train / inference consistency invariant IN GRAD MODE: the replay log-prob of `forward()` with autograd enabled is
    bit-identical to the no-grad replay and to the rollout's log-prob (ratio == exp(0) == 1 exactly);
  * weight gradients of a PPO-style loss (log-prob term + a KL-like noise_pred term) vs torch autograd through the fp32 oracle
    (oracle/mmditx_ref.py, differentiable) on identical bf16-rounded weights: per-parameter rel-L2 <= 5e-2 (bf16 activations and
    activation gradients through 3 blocks; the oracle is fp32 end to end), cosine >= 0.995;
  * LoRA: gradients reach lora_A / lora_B through the merged weight, equal to the chain rule applied to the dense gradient;
  * an optimizer step changes the replay log-prob (weights are live) and the next backward still works;
  * full scope (`target_modules: all`): gradients of EVERY parameter (AdaLN modulation linears, norm weights, conditioning MLPs, embedders,
    proj_out) vs the oracle.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import _plugin_fakes as PF

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def _cos(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


# ------------------------------------------------------------------------------------------------- attention backward
@pytest.mark.parametrize("B,H,S", [(1, 2, 64), (1, 1, 100), (1, 2, 192), (2, 1, 256), (1, 2, 300), (1, 1, 449), (2, 3, 333 + 256), (1, 2, 1000), (1, 4, 4429)])
def test_attention_backward_matches_autograd(gpu, B, H, S):
    import ctypes as C
    from mi355_flow import _lib
    from mi355_flow.engine import _ptr, _stream
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(S * 7 + H)
    S_pad = (S + 63) // 64 * 64
    c = 0.125 * 1.4426950408889634
    q = torch.zeros(B, H, S_pad, 64, device="cuda", dtype=torch.bfloat16)
    k, v = torch.zeros_like(q), torch.zeros_like(q)
    q[:, :, :S] = (torch.randn(B, H, S, 64, device="cuda", generator=g) * c).bfloat16()          # stored q carries log2(e)/8
    k[:, :, :S] = (torch.randn(B, H, S, 64, device="cuda", generator=g) * 1.3).bfloat16()
    v[:, :, :S] = torch.randn(B, H, S, 64, device="cuda", generator=g).bfloat16()
    vT = v.transpose(2, 3).contiguous()
    do = torch.randn(B * S, H * 64, device="cuda", generator=g).bfloat16()
    o = torch.empty_like(do)
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(q)
    _lib.check(lib.mi355_op_attention_fwd_bwd(_stream(), _ptr(q), _ptr(k), _ptr(vT), _ptr(do), _ptr(o), _ptr(dq), _ptr(dk), _ptr(dv),
                                              B, H, S, S_pad), "op_attention_fwd_bwd")
    # round 6: the shipped passes are the software-pipelined ones (csrc/gen_attn_bwd64.py; 1 .. 70 tiles here: prologue-only, every ring-slot
    # wrap, ragged last tiles).  Same MFMAs in the same order per output element as the round-3 kernels (mi355_tune_set(43, 0)): the same bits,
    # run after run.
    try:
        _lib.check(lib.mi355_tune_set(43, 0))
        dq0, dk0, dv0 = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(q)
        _lib.check(lib.mi355_op_attention_fwd_bwd(_stream(), _ptr(q), _ptr(k), _ptr(vT), _ptr(do), _ptr(o), _ptr(dq0), _ptr(dk0), _ptr(dv0),
                                                  B, H, S, S_pad), "op_attention_fwd_bwd")
    finally:
        lib.mi355_tune_set(43, 1)
    for name, a, b in (("dq", dq, dq0), ("dk", dk, dk0), ("dv", dv, dv0)):
        assert torch.equal(a, b), (name, float((a.float() - b.float()).abs().max()))
    # the two-row-block form (csrc/gen_attn_bwd64x2.py: 64 keys / queries per wave, one wave per SIMD; mi355_tune_set(43, 6)): the same bits again
    try:
        _lib.check(lib.mi355_tune_set(43, 6))
        dq2, dk2, dv2 = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(q)
        _lib.check(lib.mi355_op_attention_fwd_bwd(_stream(), _ptr(q), _ptr(k), _ptr(vT), _ptr(do), _ptr(o), _ptr(dq2), _ptr(dk2), _ptr(dv2),
                                                  B, H, S, S_pad), "op_attention_fwd_bwd")
    finally:
        lib.mi355_tune_set(43, 1)
    for name, a, b in (("dq", dq2, dq0), ("dk", dk2, dk0), ("dv", dv2, dv0)):
        assert torch.equal(a, b), (name + " (two-row-block form)", float((a.float() - b.float()).abs().max()))
    for _ in range(2):
        dq1, dk1, dv1 = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(q)
        _lib.check(lib.mi355_op_attention_fwd_bwd(_stream(), _ptr(q), _ptr(k), _ptr(vT), _ptr(do), _ptr(o), _ptr(dq1), _ptr(dk1), _ptr(dv1),
                                                  B, H, S, S_pad), "op_attention_fwd_bwd")
        assert torch.equal(dq1, dq) and torch.equal(dk1, dk) and torch.equal(dv1, dv)
    # reference: softmax(ln2 * q~ k^T) v in fp32 with autograd, gradients w.r.t. the STORED (pre-scaled) q
    qr = q[:, :, :S].float().requires_grad_(True)
    kr = k[:, :, :S].float().requires_grad_(True)
    vr = v[:, :, :S].float().requires_grad_(True)
    p = torch.softmax((qr @ kr.transpose(2, 3)) * math.log(2.0), dim=-1)
    oref = (p @ vr).transpose(1, 2).reshape(B * S, H * 64)
    oref.backward(do.float())
    assert _rel(o, oref) < 6e-3
    for name, got, ref in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        r = _rel(got[:, :, :S], ref)
        assert r < 2e-2 and _cos(got[:, :, :S], ref) > 0.9995, (name, r)
        assert float(got[:, :, S:].float().abs().max()) == 0.0 if S_pad > S else True       # padded rows are never written


# ------------------------------------------------------------------------------------------------- model-level gradients
def _tiny():
    from mi355_flow.engine import TransformerConfig
    from oracle import mmditx_ref as M
    cfg_e = TransformerConfig(num_layers=3, num_heads=2, joint_attention_dim=128, pooled_projection_dim=128, pos_embed_max_size=24,
                              dual_layers=(0, 1))
    cfg_o = M.tiny_config(num_layers=3, num_heads=2, dual_layers=(0, 1), joint_attention_dim=128, pooled_projection_dim=128,
                          pos_embed_max_size=24)
    return cfg_e, cfg_o


def _build(train_filter, seed=3, std=0.08):
    """torch module with HF-named parameters (bf16-representable fp32 values) + the standalone adapter bound to it."""
    from mi355_flow.adapter import SD3_5NativeAdapter
    from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler
    from mi355_flow.weights import expected_shapes
    cfg_e, cfg_o = _tiny()
    mod = PF.build_module_tree(expected_shapes(cfg_e), seed=seed, std=std).cuda()
    with torch.no_grad():
        for prm in mod.parameters():
            prm.copy_(prm.bfloat16().float())
        for n, b in mod.named_buffers():
            b.copy_(b.bfloat16().float())
    for n, prm in mod.named_parameters():
        prm.requires_grad_(train_filter(n))
    sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1, seed=42, shift=3.0)
    ad = SD3_5NativeAdapter(mod, cfg_e, sched, latent_storage_dtype="fp16")
    ad.rollout()
    return ad, mod, cfg_o


BLOCK_LINEARS = (".attn.to_q.", ".attn.to_k.", ".attn.to_v.", ".attn.to_out.0.", ".attn.add_q_proj.", ".attn.add_k_proj.", ".attn.add_v_proj.",
                 ".attn.to_add_out.", ".attn2.to_q.", ".attn2.to_k.", ".attn2.to_v.", ".attn2.to_out.0.", ".ff.net.0.proj.", ".ff.net.2.",
                 ".ff_context.net.0.proj.", ".ff_context.net.2.")


def _inputs(B, h, w, Nt, seed=0):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=g)
    return dict(x=mk(B, 16, h, w).half(), x1=mk(B, 16, h, w).half(), pe=mk(B, Nt, 128).bfloat16(), pp=mk(B, 128).bfloat16(),
                ne=mk(B, Nt, 128).bfloat16(), npl=mk(B, 128).bfloat16(), wlp=mk(B), wnp=mk(B, 16, h, w))


def _oracle_loss(mod, cfg_o, inp, guidance, t, t_next, eta, sigma_max, kl_w, quant=None, device="cpu", attn_quant=None):
    """The same loss through the fp32 oracle network + the Flow-SDE step written in differentiable torch (host cores; `device="cuda"`: the
    same plain-PyTorch fp32 oracle on GPU tensors with the MATH attention backend, tests/_gpu_oracle.py -- what makes the full-width model
    affordable).  `quant=M.bf16_round` puts a bf16 round-trip wherever a bf16 module materialises a tensor; autograd rounds the activation
    gradient at the same points."""
    if device != "cpu":
        from _gpu_oracle import cuda, on_gpu
        with on_gpu(grad=True):
            lp, gr = _oracle_loss(mod, cfg_o, cuda(dict(inp)), guidance, t, t_next, eta, sigma_max, kl_w, quant=quant, device="cpu", attn_quant=attn_quant)
        return lp.cpu(), gr
    from oracle import mmditx_ref as M
    dev = inp["x"].device
    sd = {n: p_.detach().to(dev).float().requires_grad_(p_.requires_grad) for n, p_ in list(mod.named_parameters()) + list(mod.named_buffers())}
    x, x1 = inp["x"].float(), inp["x1"].float()
    B = x.shape[0]
    tt = torch.full((B,), float(torch.tensor(t, device="cpu").half()), device=dev)            # the network sees t rounded to the latent dtype (sd3_5.py:394)
    if guidance > 1.0:
        v2 = M.mmdit_forward(sd, cfg_o, torch.cat([x, x]), torch.cat([tt, tt]), torch.cat([inp["ne"], inp["pe"]]).float(),
                             torch.cat([inp["npl"], inp["pp"]]).float(), quant=quant, attn_quant=attn_quant)
        vu, vt = v2.chunk(2)
        v = vu + guidance * (vt - vu)
    else:
        v = M.mmdit_forward(sd, cfg_o, x, tt, inp["pe"].float(), inp["pp"].float(), quant=quant, attn_quant=attn_quant)
    sigma, sigma_n = t / 1000.0, t_next / 1000.0
    dt = sigma_n - sigma
    std = math.sqrt(sigma / (1 - (sigma_max if sigma == 1.0 else sigma))) * eta
    mean = x * (1 + std ** 2 / (2 * sigma) * dt) + v * (1 + std ** 2 * (1 - sigma) / (2 * sigma)) * dt
    sv = std * math.sqrt(-dt)
    if sv > 0:
        lp = (-((x1 - mean) ** 2) / (2 * sv ** 2) - math.log(sv) - math.log(math.sqrt(2 * math.pi))).mean(dim=(1, 2, 3))
    else:                                          # eta = 0 (the matching-loss trainers): no log-prob term
        lp = torch.zeros(B, device=dev)
    loss = (inp["wlp"] * lp).sum() + kl_w * (inp["wnp"] * v).mean()
    loss.backward()
    return lp.detach(), {n: s.grad for n, s in sd.items() if s.requires_grad}


@pytest.mark.parametrize("guidance", [1.0, 4.5])
def test_replay_gradients_match_oracle_autograd_and_ratio_is_one(gpu, guidance):
    ad, mod, cfg_o = _build(lambda n: any(k in n for k in BLOCK_LINEARS))
    B, h, w, Nt = 2, 16, 16, 13
    inp = _inputs(B, h, w, Nt, seed=5)
    t, t_next, eta, smax = 900.0, 750.0, 0.7, 0.9
    ad.scheduler.set_timesteps(4)
    cfg_on = guidance > 1.0
    kw = dict(t=torch.full((B,), t), t_next=torch.full((B,), t_next), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
              prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(),
              negative_prompt_embeds=inp["ne"].cuda() if cfg_on else None, negative_pooled_prompt_embeds=inp["npl"].cuda() if cfg_on else None,
              guidance_scale=guidance, noise_level=eta, compute_log_prob=True, return_kwargs=["log_prob", "noise_pred", "dt"])
    ad.scheduler.sigmas = ad.scheduler.sigmas.clone()
    ad.scheduler.sigmas[1] = smax
    with torch.no_grad():
        ref_out = ad.forward(**kw)                      # the no-grad replay (what round 1 shipped)
    out = ad.forward(**kw)                              # grad mode: the engine's differentiable step
    assert out.log_prob.requires_grad and out.noise_pred.requires_grad
    assert torch.equal(out.log_prob.detach(), ref_out.log_prob)          # ratio == exp(0) == 1.0 EXACTLY in grad mode
    assert torch.equal(out.noise_pred.detach(), ref_out.noise_pred)
    kl_w = 3.0
    loss = (inp["wlp"].cuda() * out.log_prob).sum() + kl_w * (inp["wnp"].cuda() * out.noise_pred).mean()
    loss.backward()
    lp_ref, g_ref = _oracle_loss(mod, cfg_o, inp, guidance, t, t_next, eta, smax, kl_w)
    np.testing.assert_allclose(out.log_prob.detach().cpu().numpy(), lp_ref.numpy(), rtol=5e-3)
    worst, worst_name, n = 0.0, None, 0
    for name, prm in mod.named_parameters():
        if not prm.requires_grad:
            assert prm.grad is None
            continue
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), name
        ref = g_ref[name]
        if float(ref.norm()) < 1e-12:                   # e.g. add_q_proj of the context-pre-only last block: exactly zero both sides
            assert float(prm.grad.float().norm()) < 1e-6, name
            continue
        r = _rel(prm.grad, ref)
        n += 1
        if r > worst:
            worst, worst_name = r, name
        assert r < 5e-2 and _cos(prm.grad, ref) > 0.995, (name, r, _cos(prm.grad, ref))
    print(f"guidance {guidance}: {n} parameter gradients vs fp32 oracle autograd, worst rel-L2 {worst:.3e} ({worst_name})")
    assert n >= 60
    ad.engine.close()


@pytest.mark.parametrize("guidance", [1.0, 4.5])
def test_matching_loss_forward_without_a_stored_transition(gpu, guidance):
    """AWM / NFT / DPO / DGPO / CRD (reference trainers/awm.py:357-370, nft.py:297-304, dpo.py:468, dgpo.py:352-364, crd.py:497-509) call
    `forward()` WITH autograd at an arbitrary t, `t_next = 0`, `noise_level = 0`, `compute_log_prob=False`, no `next_latents`, and
    train on `noise_pred`: same value as the no-grad forward bit for bit, gradients vs the fp32 oracle's autograd."""
    ad, mod, cfg_o = _build(lambda n: any(k in n for k in BLOCK_LINEARS), seed=21)
    B, h, w, Nt = 2, 16, 16, 13
    inp = _inputs(B, h, w, Nt, seed=23)
    t = 437.5                                       # off the scheduler grid; representable in the fp16 latent dtype
    ad.scheduler.set_timesteps(4)
    cfg_on = guidance > 1.0
    kw = dict(t=torch.full((B,), t), t_next=torch.zeros(B), latents=inp["x"].cuda(), prompt_embeds=inp["pe"].cuda(),
              pooled_prompt_embeds=inp["pp"].cuda(), negative_prompt_embeds=inp["ne"].cuda() if cfg_on else None,
              negative_pooled_prompt_embeds=inp["npl"].cuda() if cfg_on else None, guidance_scale=guidance, noise_level=0.0,
              compute_log_prob=False, return_kwargs=["noise_pred"])
    with torch.no_grad():
        ref_out = ad.forward(**kw)
    out = ad.forward(**kw)
    assert out.noise_pred.requires_grad and out.log_prob is None and out.next_latents is None
    assert torch.equal(out.noise_pred.detach(), ref_out.noise_pred)
    kl_w = 3.0
    (kl_w * (inp["wnp"].cuda() * out.noise_pred).mean()).backward()
    _, g_ref = _oracle_loss(mod, cfg_o, inp, guidance, t, 0.0, 0.0, 0.9, kl_w)
    n, worst = 0, 0.0
    for name, prm in mod.named_parameters():
        if not prm.requires_grad:
            continue
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), name
        ref = g_ref[name]
        if float(ref.norm()) < 1e-12:
            assert float(prm.grad.float().norm()) < 1e-6, name
            continue
        if name.endswith("k.bias") or name.endswith("k_proj.bias"):
            # a key bias shifts every score of a query equally: its gradient is zero up to rounding (softmax shift invariance);
            # judge the error against the scale of the matching weight gradient instead of its own near-zero norm
            wref = g_ref[name[:-4] + "weight"]
            err = float((prm.grad.float().cpu() - ref).norm())
            assert err <= max(1.5e-1 * float(ref.norm()), 2e-3 * float(wref.norm())), (name, err, float(ref.norm()), float(wref.norm()))
            continue
        r = _rel(prm.grad, ref)
        worst, n = max(worst, r), n + 1
        assert r < 5e-2 and _cos(prm.grad, ref) > 0.995, (name, r)
    print(f"matching-loss forward, guidance {guidance}: {n} parameter gradients, worst rel-L2 {worst:.3e}")
    assert n >= 50
    # a sampled next state with autograd is not a native output: standalone there is no reference path to fall back to
    with pytest.raises(NotImplementedError, match="sampled next state"):
        ad.forward(**{**kw, "return_kwargs": ["noise_pred", "next_latents"]})
    ad.engine.close()


def test_gradients_ragged_shapes_and_partial_trainable_set(gpu):
    """Odd batch, non-square grid, text length not a multiple of anything; only the attention projections of the image stream are
    trainable (the reference's default target modules, models/abc.py:382-385): frozen parameters get no gradient."""
    targets = (".to_q.", ".to_k.", ".to_v.", ".to_out.0.")
    ad, mod, cfg_o = _build(lambda n: any(k in n for k in targets), seed=9)
    B, h, w, Nt = 3, 8, 24, 5
    inp = _inputs(B, h, w, Nt, seed=11)
    t, t_next, eta, smax = 750.0, 500.0, 0.7, 0.9
    ad.scheduler.set_timesteps(4)
    ad.scheduler.sigmas = ad.scheduler.sigmas.clone()
    ad.scheduler.sigmas[1] = smax
    kw = dict(t=torch.full((B,), t), t_next=torch.full((B,), t_next), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
              prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(), guidance_scale=1.0, noise_level=eta,
              compute_log_prob=True, return_kwargs=["log_prob", "dt"])
    out = ad.forward(**kw)
    (inp["wlp"].cuda() * out.log_prob).sum().backward()
    _, g_ref = _oracle_loss(mod, cfg_o, inp, 1.0, t, t_next, eta, smax, 0.0)
    n = 0
    for name, prm in mod.named_parameters():
        if prm.requires_grad:
            assert _rel(prm.grad, g_ref[name]) < 5e-2, (name, _rel(prm.grad, g_ref[name]))
            n += 1
        else:
            assert prm.grad is None
    assert n == 2 * 4 * (3 + 2)          # weight + bias of 4 projections in attn (3 blocks) and attn2 (2 dual blocks)
    # a weight update is seen by the next replay (weights are live) and the backward keeps working
    lp0 = out.log_prob.detach().clone()
    opt = torch.optim.SGD([p_ for p_ in mod.parameters() if p_.requires_grad], lr=5e-2)
    opt.step()
    opt.zero_grad()
    out2 = ad.forward(**kw)
    assert not torch.equal(out2.log_prob.detach(), lp0)
    out2.log_prob.sum().backward()
    assert all(torch.isfinite(p_.grad).all() for p_ in mod.parameters() if p_.requires_grad)
    ad.engine.close()


@pytest.mark.parametrize("targets", [(".to_q.", ".to_k.", ".to_v.", ".to_out.0."), None], ids=["default_targets", "all_block_linears"])
def test_side_stream_training_schedule_is_bit_identical_to_the_serial_one(gpu, targets):
    """mi355_tune_set(22, .): the context-stream chain of the training forward / backward on a side stream (default) vs in line.  Same
    kernels on the same operands, all deterministic in the default scope: log-prob and every gradient must be torch.equal -- any
    missing join / fork edge shows up here as a difference (repeated: a race need not fire every time).  With every block linear
    trainable the context stream's weight gradients run on the side stream too (their own transposes / split-K partials)."""
    from mi355_flow import _lib
    lib = _lib.load()
    targets = targets or BLOCK_LINEARS
    ad, mod, _ = _build(lambda n: n.startswith("transformer_blocks.") and any(k in n for k in targets), seed=21)
    B, h, w, Nt = 4, 16, 16, 77
    inp = _inputs(B, h, w, Nt, seed=5)
    ad.scheduler.set_timesteps(4)
    kw = dict(t=torch.full((B,), 750.0), t_next=torch.full((B,), 500.0), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
              prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(), guidance_scale=1.0, noise_level=0.7,
              compute_log_prob=True, return_kwargs=["log_prob", "dt"])

    def step():
        for prm in mod.parameters():
            prm.grad = None
        out = ad.forward(**kw)
        (inp["wlp"].cuda() * out.log_prob).sum().backward()
        torch.cuda.synchronize()
        return out.log_prob.detach().clone(), {n: prm.grad.clone() for n, prm in mod.named_parameters() if prm.requires_grad}

    try:
        _lib.check(lib.mi355_tune_set(22, 0), "tune_set")
        lp0, g0 = step()
        _lib.check(lib.mi355_tune_set(22, 1), "tune_set")
        for _ in range(4):
            lp1, g1 = step()
            assert torch.equal(lp0, lp1)
            for n in g0:
                assert torch.equal(g0[n], g1[n]), n
    finally:
        _lib.check(lib.mi355_tune_set(22, 1), "tune_set")
    ad.engine.close()


def test_wide_access_backward_helper_kernels_match_the_general_ones(gpu):
    """mi355_tune_set(25, 1): the 16-byte-access forms of the attention-backward prep kernel (dO scatter + Delta) and of the default-scope
    RMSNorm-backward gather.  Same arithmetic per element, per-head sums formed in another order: the last block's gradients (one gather
    deep) agree to summation-order accuracy, everything else to what three blocks of bf16 backward amplify that to."""
    from mi355_flow import _lib
    lib = _lib.load()
    targets = (".to_q.", ".to_k.", ".to_v.", ".to_out.0.")
    ad, mod, _ = _build(lambda n: any(k in n for k in targets), seed=23)
    B, h, w, Nt = 3, 12, 16, 21
    inp = _inputs(B, h, w, Nt, seed=6)
    ad.scheduler.set_timesteps(4)
    kw = dict(t=torch.full((B,), 750.0), t_next=torch.full((B,), 500.0), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
              prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(), guidance_scale=1.0, noise_level=0.7,
              compute_log_prob=True, return_kwargs=["log_prob", "dt"])

    def grads():
        for prm in mod.parameters():
            prm.grad = None
        out = ad.forward(**kw)
        (inp["wlp"].cuda() * out.log_prob).sum().backward()
        torch.cuda.synchronize()
        return {n: prm.grad.clone() for n, prm in mod.named_parameters() if prm.requires_grad}

    try:
        _lib.check(lib.mi355_tune_set(25, 0), "tune_set")
        g0 = grads()
        _lib.check(lib.mi355_tune_set(25, 1), "tune_set")
        g1 = grads()
    finally:
        _lib.check(lib.mi355_tune_set(25, 0), "tune_set")
    for n in g0:
        if n.endswith("to_k.bias"):       # (zero in exact arithmetic by the softmax's shift invariance: both values are rounding noise)
            continue
        r = _rel(g1[n], g0[n])
        assert r < (5e-3 if n.startswith("transformer_blocks.2.") else 5e-2), (n, r)
    ad.engine.close()


def test_lora_gradients_flow_through_the_merged_weight(gpu):
    ad, mod, cfg_o = _build(lambda n: False, seed=21)
    PF.wrap_lora(mod)
    mod.cuda()
    for n, prm in mod.named_parameters():
        prm.requires_grad_("lora_" in n)
    ad._live_weights.reset()                    # the module tree changed in place: re-resolve the sources
    B, h, w, Nt = 2, 16, 16, 13
    inp = _inputs(B, h, w, Nt, seed=2)
    ad.scheduler.set_timesteps(4)
    kw = dict(t=torch.full((B,), 900.0), t_next=torch.full((B,), 750.0), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
              prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(), guidance_scale=1.0, noise_level=0.7,
              compute_log_prob=True, return_kwargs=["log_prob", "dt"])
    out = ad.forward(**kw)
    (inp["wlp"].cuda() * out.log_prob).sum().backward()
    lay = mod.get_submodule("transformer_blocks.1.attn.to_q")
    A, Bm = lay.lora_A["default"].weight, lay.lora_B["default"].weight
    assert A.grad is not None and Bm.grad is not None and lay.base_layer.weight.grad is None
    assert float(A.grad.norm()) > 0 and float(Bm.grad.norm()) > 0
    # dense check: make the merged weight itself a leaf, take its gradient, apply the chain rule by hand
    from mi355_flow import autograd as AG
    pairs = dict(AG.trainable_sources(ad._live_weights))
    src = pairs["transformer_blocks.1.attn.to_q.weight"]
    Wm = AG._materialise_with_grad(src, 1.0).detach().requires_grad_(True)
    names = list(pairs)
    ws = [AG._materialise_with_grad(s, 1.0).detach().requires_grad_(n_ == "transformer_blocks.1.attn.to_q.weight") for n_, s in pairs.items()]
    ws[names.index("transformer_blocks.1.attn.to_q.weight")] = Wm
    plan = ad.engine.plan(B, 1, h, w, Nt, 1)
    t32 = torch.full((B,), 900.0)
    call = dict(latents=inp["x"].cuda(), timestep=t32, enc_a=inp["pe"].cuda(), pooled_a=inp["pp"].cuda(), enc_b=None, pooled_b=None, guidance=1.0,
                sigma=torch.full((B,), 0.9), sigma_next=torch.full((B,), 0.75), eta=torch.full((B,), 0.7), sigma_max=float(ad.scheduler.sigmas[1]),
                dynamics="Flow-SDE", next_latents=inp["x1"].cuda(), compute_log_prob=True)
    lp = AG._DenoiseReplayFn.apply(ad, plan, names, call, *ws)[0]
    (inp["wlp"].cuda() * lp).sum().backward()
    s = lay.scaling["default"]
    assert _rel(Bm.grad, s * (Wm.grad @ A.detach().t())) < 2e-2
    assert _rel(A.grad, s * (Bm.detach().t() @ Wm.grad)) < 2e-2
    ad.engine.close()


def test_full_scope_gradients_of_every_parameter(gpu):
    """`target_modules: all`: every parameter of the transformer trainable -- AdaLN modulation linears (per-sample token sums of d shift /
    d scale / d gate), q/k RMSNorm weights, timestep / pooled-text MLPs, context_embedder, patch embedding, proj_out, on top of the blocks'
    linear layers -- vs torch autograd through the fp32 oracle.  CFG on (n_cfg = 2: the conditioning path sees both prompt halves)."""
    ad, mod, cfg_o = _build(lambda n: True, seed=13)
    B, h, w, Nt = 2, 16, 16, 13
    inp = _inputs(B, h, w, Nt, seed=17)
    t, t_next, eta, smax, guidance = 900.0, 750.0, 0.7, 0.9, 3.0
    ad.scheduler.set_timesteps(4)
    kw = dict(t=torch.full((B,), t), t_next=torch.full((B,), t_next), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
              prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(), negative_prompt_embeds=inp["ne"].cuda(),
              negative_pooled_prompt_embeds=inp["npl"].cuda(), guidance_scale=guidance, noise_level=eta, compute_log_prob=True,
              return_kwargs=["log_prob", "noise_pred", "dt"])
    with torch.no_grad():
        ref_out = ad.forward(**kw)
    out = ad.forward(**kw)
    assert torch.equal(out.log_prob.detach(), ref_out.log_prob)          # the extra stashes do not touch the forward results
    kl_w = 2.0
    ((inp["wlp"].cuda() * out.log_prob).sum() + kl_w * (inp["wnp"].cuda() * out.noise_pred).mean()).backward()
    _, g_ref = _oracle_loss(mod, cfg_o, inp, guidance, t, t_next, eta, smax, kl_w)
    worst = {}
    n = 0
    for name, prm in mod.named_parameters():
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), name
        ref = g_ref[name]
        if float(ref.norm()) < 1e-12:
            assert float(prm.grad.float().norm()) < 1e-6, name
            continue
        r, c = _rel(prm.grad, ref), _cos(prm.grad, ref)
        kind = ("mod" if ".linear." in name and "norm" in name else "norm_w" if ".norm_" in name else
                "cond" if name.startswith("time_text_embed") else "embed" if name.startswith(("pos_embed", "context_embedder", "proj_out")) else "block")
        if r > worst.get(kind, (0.0, ""))[0]:
            worst[kind] = (r, name)
        assert r < 6e-2 and c > 0.99, (name, r, c)
        n += 1
    print("full scope:", n, "parameter gradients; worst rel-L2 per kind:", {k: f"{v[0]:.3e} ({v[1]})" for k, v in worst.items()})
    assert n >= 80 + 3 * 4 + 16 + 8 + 6 - 4
    ad.engine.close()


@pytest.mark.parametrize("tune", [(), ((6, 0),), ((1, 0),), ((1, 2),)], ids=["static", "dynamic", "plain", "4wave"])
def test_train_forward_is_bit_identical_at_full_width(tune):
    """The rollout / no-grad replay and the training-mode forward run different template instantiations of the fused q/k RMSNorm epilogue
    (EPI_QK_NORM vs EPI_QK_NORM_RSTD) and of the attention kernel (LSE output): hipcc may contract or pack floating-point operations
    differently per instantiation (round 2: v_pk_fma in one, v_pk_mul + v_pk_add in the other -> 1-ulp bf16 differences, ratio != 1).
    SD3.5 width (24 heads x 64), 3 blocks incl. two dual-attention ones, every attention kernel variant: outputs must be torch.equal."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mi355_flow import _lib, engine
    from mi355_flow.weights import synthetic_state_dict
    lib = _lib.load()
    cfg = engine.TransformerConfig(num_layers=3, dual_layers=(0, 1))
    e = engine.Engine(cfg)
    e.bind_state_dict(synthetic_state_dict(cfg, device="cuda", seed=1, dtype=torch.bfloat16))
    e.ready()
    try:
        for k, v in tune:
            _lib.check(lib.mi355_tune_set(k, v), "tune_set")
        g = torch.Generator().manual_seed(3)
        for B, h, w, n_cfg in ((1, 32, 32, 1), (2, 48, 32, 2)):
            x = torch.randn(B, 16, h, w, generator=g).half().cuda()
            x1 = (x.float() + 0.1 * torch.randn(B, 16, h, w, generator=g).cuda()).half()
            pe, pp = torch.randn(B, 333, 4096, generator=g).bfloat16().cuda(), torch.randn(B, 2048, generator=g).bfloat16().cuda()
            ne, npl = (torch.randn(B, 333, 4096, generator=g).bfloat16().cuda(), torch.randn(B, 2048, generator=g).bfloat16().cuda()) if n_cfg == 2 else (None, None)
            plan = e.plan(B, n_cfg, h, w, 333, 1)
            sc = (torch.full((B,), 0.9), torch.full((B,), 0.75), torch.full((B,), 0.7))
            a = plan.denoise_step(x, torch.full((B,), 900.0), pe, pp, ne, npl, 4.5, *sc, 0.9, "Flow-SDE", next_latents=x1, want=("noise_pred",))
            for full in (False, True):
                e.set_train_scope(full)
                b = plan.denoise_step_train(x, torch.full((B,), 900.0), pe, pp, ne, npl, 4.5, *sc, 0.9, "Flow-SDE", x1)
                assert torch.equal(a.noise_pred, b.noise_pred) and torch.equal(a.log_prob, b.log_prob), (tune, B, full)
    finally:
        _lib.check(lib.mi355_tune_set(6, 1), "tune_set")
        _lib.check(lib.mi355_tune_set(1, 1), "tune_set")
        e.set_train_scope(False)
        e.close()


def test_two_grad_forwards_on_one_plan_before_a_single_backward(gpu):
    """DPO (reference trainers/dpo.py:587-588) runs the chosen and the rejected forward -- same shape, hence the same plan and the same
    activation stash -- BEFORE one `backward()` of the summed loss.  Each forward's backward node must differentiate ITS OWN forward: the
    plan stamps every training forward with a serial and re-runs a forward whose stash has been overwritten.  Compared against the two
    losses back-propagated one at a time (each right after its forward, the path every other test takes): bit-identical is not promised
    (accumulation order into `.grad`), equal to fp32 rounding is."""
    targets = (".to_q.", ".to_k.", ".to_v.", ".to_out.0.", ".ff.net.0.proj.")
    ad, mod, cfg_o = _build(lambda n: any(k in n for k in targets), seed=31)
    B, h, w, Nt = 2, 16, 16, 13
    ad.scheduler.set_timesteps(4)

    def kwargs(seed, t):
        inp = _inputs(B, h, w, Nt, seed=seed)
        return inp, dict(t=torch.full((B,), t), t_next=torch.zeros(B), latents=inp["x"].cuda(), prompt_embeds=inp["pe"].cuda(),
                         pooled_prompt_embeds=inp["pp"].cuda(), guidance_scale=1.0, noise_level=0.0, compute_log_prob=False,
                         return_kwargs=["noise_pred"])

    (inp_w, kw_w), (inp_l, kw_l) = kwargs(41, 437.5), kwargs(43, 812.5)
    loss_of = lambda inp, out: (inp["wnp"].cuda() * out.noise_pred).mean()

    def grads():
        g = {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}
        for p in mod.parameters():
            p.grad = None
        return g

    # reference: one forward, one backward, twice (gradients accumulate in .grad)
    loss_of(inp_w, ad.forward(**kw_w)).backward()
    (-0.5 * loss_of(inp_l, ad.forward(**kw_l))).backward()
    g_seq = grads()
    plan = next(iter(ad.engine._plans.values()))
    assert getattr(plan, "recomputed_forwards", 0) == 0
    # DPO order: both forwards first, then ONE backward
    out_w, out_l = ad.forward(**kw_w), ad.forward(**kw_l)
    (loss_of(inp_w, out_w) - 0.5 * loss_of(inp_l, out_l)).backward()
    g_joint = grads()
    assert plan.recomputed_forwards == 1                 # the first forward's stash had been overwritten and was rebuilt
    assert g_seq.keys() == g_joint.keys() and len(g_seq) >= 20
    for n in g_seq:
        assert _rel(g_joint[n], g_seq[n]) < 1e-5, (n, _rel(g_joint[n], g_seq[n]))
    # and the stale-stash failure this guards against is real: the two forwards' gradients differ by far more than that
    loss_of(inp_w, ad.forward(**kw_w)).backward()
    g_w = grads()
    assert max(_rel(g_w[n], g_seq[n]) for n in g_seq) > 1e-2
    ad.engine.close()


# ------------------------------------------------------------------------------------------------- weight gradients on row-major operands
@pytest.mark.parametrize("M,N,K,split,strided", [(64, 128, 128, 1, False), (128, 128, 256, 2, False), (4096, 1536, 1536, 4, False), (8192, 1536, 1536, 7, True),
                                                 (8192, 1536, 6144, 3, False), (2048, 4608, 1536, 2, True), (320, 256, 128, 5, False),
                                                 # ragged M (the other engines' token counts are not multiples of 64: the last m-tile's missing rows read zeros)
                                                 (666, 1536, 1536, 2, False), (333, 256, 256, 1, True), (40560, 1536, 1536, 7, False), (13978, 3072, 3072, 3, True), (1, 128, 128, 1, False)])
def test_weight_gradient_gemm_on_row_major_operands_is_bit_identical_to_the_transposed_copy_path(gpu, M, N, K, split, strided):
    """Round 6 (csrc/gemm_tn.hip, mi355_op_wgrad): dW = dY^T X with dY [M][N] and X [M][K] read as they lie in HBM -- the MFMA fragments (8
    consecutive m for one column) come out of the row-major LDS tiles through `ds_read_b64_tr_b16` -- against the path of rounds 2-5 (two
    transposed copies + the K-contiguous GEMM): the fp32 partial sums of every split are BIT-IDENTICAL (same MFMA, operand order, m order,
    split boundaries), and both agree with torch's fp32 product of the same bf16 values.  `strided`: dY is a column block of a wider buffer
    (the q | k | v gradients lie side by side, engine_train.inc) and X a view with a larger row stride.  Ragged M: the copies are zero-padded to
    whole 64-row tiles, the row-major kernels read zeros for the missing rows of the last tile -- still the same bits."""
    from mi355_flow import engine
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    if strided:
        dy_full = torch.randn(M, N + 256, device="cuda", generator=g).bfloat16()
        x_full = torch.randn(M, K + 64, device="cuda", generator=g).bfloat16()
        dy, x = dy_full[:, 128:128 + N], x_full[:, :K]
    else:
        dy = torch.randn(M, N, device="cuda", generator=g).bfloat16()
        x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    a = engine.op_wgrad(dy, x, split, variant=1)
    b = engine.op_wgrad(dy, x, split, variant=0)
    torch.cuda.synchronize()
    assert torch.equal(a, b), float((a - b).abs().max())
    for _ in range(3):                                    # (a staging race would show as a run-to-run difference)
        assert torch.equal(engine.op_wgrad(dy, x, split, variant=1), a)
    ref = dy.float().t() @ x.float()
    got = a.sum(0)
    assert _rel(got, ref) < 2e-5, _rel(got, ref)
    # the bias gradient's partials from the same launch: per-slice column sums of dY out of the operand fragments (v_dot2c against (1, 1))
    a2, cs = engine.op_wgrad(dy, x, split, variant=1, want_colsum=True)
    assert torch.equal(a2, a) and cs.shape == (split, N)
    nt = (M + 63) // 64                                   # slices are taken on whole 64-row tiles of M rounded up
    for sidx in range(split):
        lo, hi = nt * sidx // split * 64, min(M, nt * (sidx + 1) // split * 64)
        want = dy[lo:hi].float().sum(0)
        assert float((cs[sidx] - want).abs().max()) <= 1e-4 * float(want.abs().max() + 1.0), (sidx, float((cs[sidx] - want).abs().max()))
    assert torch.equal(engine.op_wgrad(dy, x, split, variant=1, want_colsum=True)[1], cs)       # deterministic
    if N % 256 == 0 and K % 256 == 0:
        # the 256 x 256-tile form (8 waves, one workgroup per CU; what the engine takes for N = K = 1536): the same bits again
        c = engine.op_wgrad(dy, x, split, variant=2)
        assert torch.equal(c, a), float((c - a).abs().max())
        for _ in range(3):
            assert torch.equal(engine.op_wgrad(dy, x, split, variant=2), c)
    with pytest.raises(RuntimeError):
        engine.op_wgrad(dy[:, :N - 8], x, 1, variant=1)               # N not a multiple of 128: refused (ragged M is fine since round 6b; ragged N / K is not)


@pytest.mark.parametrize("bf16_master", [False, True])
def test_replay_gradients_with_row_major_weight_gradient_gemms_equal_the_transposed_copy_path(gpu, bf16_master):
    """The same switch inside the engine (`mi355_tune_set(39, .)`): every WEIGHT gradient of the optimize() replay is bit-identical with and
    without the transposed copies; the bias gradients (column sums, taken by the slab kernel instead of on the way through the transpose) agree
    to fp32 reassociation."""
    from mi355_flow import _lib
    lib = _lib.load()
    ad, mod, cfg_o = _build(lambda n: any(k in n for k in BLOCK_LINEARS))
    B, h, w, Nt = 2, 16, 16, 13                           # 128 image rows: whole 64-row tiles; 26 text rows: the transposed-copy path either way
    inp = _inputs(B, h, w, Nt, seed=11)
    ad.scheduler.set_timesteps(4)
    kw = dict(t=torch.full((B,), 900.0), t_next=torch.full((B,), 750.0), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
              prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(), guidance_scale=1.0, noise_level=0.7, compute_log_prob=True,
              return_kwargs=["log_prob", "noise_pred", "dt"])
    if bf16_master:
        mod.bfloat16()
        live = getattr(ad, "_live_weights", None)
        if live is not None:
            live.reset()
    try:
        grads = {}
        for tn in (1, 2, 0):
            _lib.check(lib.mi355_tune_set(39, tn))
            for p in mod.parameters():
                p.grad = None
            out = ad.forward(**kw)
            ((inp["wlp"].cuda() * out.log_prob).sum() + 3.0 * (inp["wnp"].cuda() * out.noise_pred).mean()).backward()
            grads[tn] = {n: p.grad.detach().clone() for n, p in mod.named_parameters() if p.requires_grad}
        n_w = 0
        for n, g1 in grads[1].items():
            g0 = grads[0][n]
            if n.endswith(".weight"):
                assert torch.equal(g1, g0), n
                # key 39 = 2: the same products on 256 x 256 tiles where the shape gives a one-round grid; its split over m differs from the
                # other paths' there (another association of the fp32 partial sums)
                g2 = grads[2][n]
                assert torch.equal(g2, g0) or _rel(g2, g0) < (1e-2 if bf16_master else 2e-6), (n, _rel(g2, g0))
                n_w += int(float(g1.float().norm()) > 0)
            else:
                assert _rel(g1, g0) < (1e-2 if bf16_master else 1e-5), (n, _rel(g1, g0))
        assert n_w >= 20
    finally:
        lib.mi355_tune_set(39, 1)
        ad.engine.close()
