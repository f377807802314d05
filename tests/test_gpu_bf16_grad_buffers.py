"""bf16 gradient buffers (mi355_{engine,flux,qwen}_set_grad_typed, MI355_BF16): with bf16 master parameters -- what the reference trains
(model_args.py:49, bf16 weights) -- the engine's reduction kernels round the fp32 weight / bias gradient sums to bf16 on the way out instead of
writing an fp32 buffer that torch then converts.  The two routes must agree BIT FOR BIT (same fp32 sums, same round-to-nearest-even), for the
three families with a native backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return True


def _run(ad, mod, kw, wlp, wnp, force_f32):
    import mi355_flow.autograd as AG
    orig, seen = AG._grad_buffers, {}

    def spy(eng, names, w_meta, device):
        bufs = orig(eng, names, [(s, torch.float32) for s, _ in w_meta] if force_f32 else w_meta, device)
        seen.update({n: b.dtype for n, b in bufs.items()})
        return bufs
    AG._grad_buffers = spy
    try:
        for p in mod.parameters():
            p.grad = None
        out = ad.forward(**kw)
        ((wlp * out.log_prob).sum() + 3.0 * (wnp * out.noise_pred).mean()).backward()
    finally:
        AG._grad_buffers = orig
    return {n: p.grad.clone() for n, p in mod.named_parameters() if p.requires_grad}, seen


def _check(ad, mod, kw, wlp, wnp, what):
    mod.bfloat16()                                   # bf16 master weights (the values were bf16-representable already)
    live = getattr(ad, "_live_weights", None)
    if live is not None:
        live.reset()
    direct, seen_d = _run(ad, mod, kw, wlp, wnp, False)
    via_f32, seen_f = _run(ad, mod, kw, wlp, wnp, True)
    assert seen_d and all(dt == torch.bfloat16 for dt in seen_d.values()), {n: d for n, d in seen_d.items() if d != torch.bfloat16}
    assert all(dt == torch.float32 for dt in seen_f.values())
    nz = 0
    for n, g in direct.items():
        assert g.dtype == torch.bfloat16 and torch.isfinite(g.float()).all(), n
        assert torch.equal(g, via_f32[n]), (n, float((g.float() - via_f32[n].float()).abs().max()))
        nz += int(float(g.float().norm()) > 0)
    print(f"{what}: {len(direct)} bf16 gradient buffers written by the engine, bit-identical to fp32 buffer -> .to(bfloat16) ({nz} non-zero)")
    assert nz >= len(direct) * 3 // 4


def test_sd3_bf16_gradient_buffers_are_bit_identical_to_the_fp32_route(gpu):
    import test_gpu_backward as TB
    ad, mod, _ = TB._build(lambda n: any(k in n for k in TB.BLOCK_LINEARS))
    B, h, w, Nt = 2, 16, 16, 13
    inp = TB._inputs(B, h, w, Nt, seed=5)
    ad.scheduler.set_timesteps(4)
    kw = dict(t=torch.full((B,), 900.0), t_next=torch.full((B,), 750.0), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
              prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(), negative_prompt_embeds=inp["ne"].cuda(),
              negative_pooled_prompt_embeds=inp["npl"].cuda(), guidance_scale=4.5, noise_level=0.7, compute_log_prob=True,
              return_kwargs=["log_prob", "noise_pred", "dt"])
    _check(ad, mod, kw, inp["wlp"].cuda(), inp["wnp"].cuda(), "SD3.5 (tiny, CFG)")
    ad.engine.close()


def test_flux_bf16_gradient_buffers_are_bit_identical_to_the_fp32_route(gpu):
    import test_gpu_flux_backward as TF
    ad, mod, cfg_o = TF._build(lambda n: any(k in n for k in TF.BLOCK_LINEARS))
    B, h, w, Nt = 2, 8, 8, 16
    inp = TF._inputs(cfg_o, B, h, w, Nt, seed=5)
    ad.scheduler.set_timesteps(4)
    kw = dict(t=torch.full((B,), 900.0), t_next=torch.full((B,), 750.0), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
              prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(), img_ids=inp["img_ids"].cuda(), guidance_scale=3.5,
              noise_level=0.7, compute_log_prob=True, return_kwargs=["log_prob", "noise_pred", "dt"])
    _check(ad, mod, kw, inp["wlp"].cuda(), inp["wnp"].cuda(), "FLUX.1 (tiny)")
    ad.engine.close()


def test_qwen_bf16_gradient_buffers_are_bit_identical_to_the_fp32_route(gpu):
    import test_gpu_qwen_backward as TQ
    from mi355_flow import qwen as qw
    from oracle import qwen_ref as R
    cfg_o = R.tiny_config()
    ad, mod = TQ._build(qw, cfg_o, lambda n: any(k in n for k in TQ.BLOCK_LINEARS))
    B, h, w, Nt = 2, 8, 12, 19
    inp = TQ._inputs(cfg_o, B, h, w, Nt, 2, True, seed=5)
    ad.scheduler.set_timesteps(4, mu=0.6)
    kw = TQ._kw(inp, B, 900.0, 750.0, 0.7, 4.0)
    _check(ad, mod, kw, inp["wlp"].cuda(), inp["wnp"].cuda(), "Qwen-Image (tiny, true CFG, ragged)")
    ad.engine.close()


def test_bf16_buffers_are_refused_outside_the_block_linears(gpu):
    import test_gpu_backward as TB
    ad, mod, _ = TB._build(lambda n: False)
    name = "pos_embed.proj.weight"
    buf = torch.zeros(mod.get_parameter(name).shape, device="cuda", dtype=torch.bfloat16)
    ad.engine.set_train_scope(True)
    with pytest.raises(RuntimeError, match="bf16 gradient buffers are limited"):
        ad.engine.set_grad(name, buf)
    ad.engine.set_grad(name, buf.float())            # fp32 is fine in the full scope
    ad.engine.clear_grads()
    ad.engine.set_train_scope(False)
    ad.engine.close()


@pytest.mark.parametrize("family,key", [("flux", 26), ("qwen", 26), ("qwen", 28), ("flux", 28)])
def test_weight_gradients_on_the_side_stream_are_bit_identical_to_the_serial_schedule(gpu, family, key):
    """mi355_tune_set(26, .): the split-K weight-gradient GEMMs + reductions on the training state's side stream vs on the backward's own stream;
    mi355_tune_set(28, .): the text chain of the Qwen-Image backward on the plan's side stream vs in line -- the same kernels on the same
    operands, only the streams differ: every gradient bit for bit."""
    from mi355_flow import _lib
    lib = _lib.load()
    grads = {}
    lib.mi355_tune_set(27, 0)          # one split-K rule for both legs (key 27's modelled factor applies to serial launches only: another summation order)
    try:
        for side in (0, 1):
            lib.mi355_tune_set(key, side)                  # (FLUX.1 / Qwen-Image read it when the plan's training state is created: a fresh adapter per setting)
            if family == "sd3":
                import test_gpu_backward as TB
                ad, mod, _ = TB._build(lambda n: any(k in n for k in TB.BLOCK_LINEARS))
                B = 2
                inp = TB._inputs(B, 16, 16, 13, seed=5)
                ad.scheduler.set_timesteps(4)
                kw = dict(t=torch.full((B,), 900.0), t_next=torch.full((B,), 750.0), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
                          prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(), negative_prompt_embeds=inp["ne"].cuda(),
                          negative_pooled_prompt_embeds=inp["npl"].cuda(), guidance_scale=4.5, noise_level=0.7, compute_log_prob=True,
                          return_kwargs=["log_prob", "noise_pred", "dt"])
            elif family == "flux":
                import test_gpu_flux_backward as TF
                ad, mod, cfg_o = TF._build(lambda n: any(k in n for k in TF.BLOCK_LINEARS))
                B = 2
                inp = TF._inputs(cfg_o, B, 8, 8, 16, seed=5)
                ad.scheduler.set_timesteps(4)
                kw = dict(t=torch.full((B,), 900.0), t_next=torch.full((B,), 750.0), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
                          prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(), img_ids=inp["img_ids"].cuda(), guidance_scale=3.5,
                          noise_level=0.7, compute_log_prob=True, return_kwargs=["log_prob", "noise_pred", "dt"])
            else:
                import test_gpu_qwen_backward as TQ
                from mi355_flow import qwen as qw
                from oracle import qwen_ref as R
                cfg_o = R.tiny_config()
                ad, mod = TQ._build(qw, cfg_o, lambda n: any(k in n for k in TQ.BLOCK_LINEARS))
                B = 2
                inp = TQ._inputs(cfg_o, B, 8, 12, 19, 2, True, seed=5)
                ad.scheduler.set_timesteps(4, mu=0.6)
                kw = TQ._kw(inp, B, 900.0, 750.0, 0.7, 4.0)
            for it in range(2):                           # the second step re-uses the operand slots
                for p in mod.parameters():
                    p.grad = None
                out = ad.forward(**kw)
                ((inp["wlp"].cuda() * out.log_prob).sum() + 3.0 * (inp["wnp"].cuda() * out.noise_pred).mean()).backward()
            torch.cuda.synchronize()
            grads[side] = {n: p.grad.clone() for n, p in mod.named_parameters() if p.requires_grad}
            ad.engine.close()
    finally:
        lib.mi355_tune_set(key, 1)
        lib.mi355_tune_set(27, 1)
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) >= 40
    for n in grads[0]:
        assert torch.equal(grads[0][n], grads[1][n]), n
    print(f"{family}, key {key}: {len(grads[0])} gradients, side-stream schedule == serial schedule bit for bit")


@pytest.mark.parametrize("bf16_master", [True, False])
def test_sd3_fused_split_k_reduce_and_column_sum_finish_is_bit_identical_to_the_two_launch_form(gpu, bf16_master):
    """Round 6 (`mi355_tune_set(38, .)`, csrc/backward.hip splitk_reduce_colsum_kernel): the weight gradient's split-K reduction and the
    bias gradient's fixed-order column-sum finish in ONE launch per gradient pair -- same sums in the same order: every gradient equals the
    two-launch form's bit for bit, with bf16 gradient buffers (the default route) and with fp32 ones."""
    import test_gpu_backward as TB
    from mi355_flow import _lib
    lib = _lib.load()
    ad, mod, _ = TB._build(lambda n: any(k in n for k in TB.BLOCK_LINEARS))
    B, h, w, Nt = 2, 16, 16, 13
    inp = TB._inputs(B, h, w, Nt, seed=7)
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
        for fused in (1, 0, 1):
            _lib.check(lib.mi355_tune_set(38, fused))
            grads[fused], _ = _run(ad, mod, kw, inp["wlp"].cuda(), inp["wnp"].cuda(), False)
        n_bias = 0
        for n, g in grads[1].items():
            assert torch.isfinite(g.float()).all() and torch.equal(g, grads[0][n]), n
            n_bias += int(n.endswith(".bias") and float(g.float().norm()) > 0)
        assert n_bias >= 10                                  # bias gradients (the column sums) are in the comparison and non-trivial
    finally:
        lib.mi355_tune_set(38, 1)
        ad.engine.close()
