"""GPU parity of the Qwen-Image rollout path against the CPU oracle (oracle/qwen_ref.py; model body
unpinned, see its header) and plain torch references of the new operators.  Everything goes through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.bfloat16().float()


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.fixture(scope="module")
def qw():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mi355_flow import qwen
    return qwen


def _setup(qw, cfg_o, seed=3, std=0.03):
    from oracle import qwen_ref as R
    sd = {k: _bf(v) for k, v in R.make_synthetic_state_dict(cfg_o, seed=seed, std=std).items()}
    cfg = qw.QwenConfig(num_layers=cfg_o.num_layers, num_attention_heads=cfg_o.num_attention_heads,
                        joint_attention_dim=cfg_o.joint_attention_dim, scale_rope=cfg_o.scale_rope)
    return sd, cfg


def _engine(qw, sd, cfg):
    e = qw.QwenEngine(cfg)
    e.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    e.ready()
    return e


def _text(B, Nt, J, lens, g):
    enc = _bf(torch.randn(B, Nt, J, generator=g))
    for b, n in enumerate(lens):
        enc[b, n:] = 0
    return enc


def test_rms_rows_and_cfg_rescale_match_torch(qw):
    from oracle import qwen_ref as R
    g = torch.Generator().manual_seed(1)
    x = _bf(torch.randn(37, 3584, generator=g) * 2)
    w = 1 + 0.1 * torch.randn(3584, generator=g)
    got = qw.op_rms_rows(x.bfloat16().cuda(), w.cuda()).float().cpu()
    ref = R._rms(x, w, 1e-6)
    assert (got - ref).abs().max().item() < 3e-2 and _rel(got, ref) < 3e-3
    neg = torch.randn(1000, 64, generator=g).bfloat16()
    pos = (neg.float() + 0.3 * torch.randn(1000, 64, generator=g)).bfloat16()
    for gs in (1.5, 4.0):
        got = qw.op_cfg_rescale(neg.cuda(), pos.cuda(), gs).cpu()
        ref = R.cfg_rescale_bf16(neg, pos, gs)
        assert ref.dtype == torch.bfloat16
        # every op of the reference rounds to bf16; the kernel follows the same rounding points (norm accumulation order may differ)
        assert (got.float() - ref.float()).abs().max().item() <= 2 ** -7 * ref.float().abs().max().item()
        assert _rel(got, ref) < 2e-3
        # the rescale keeps the per-token norm of the conditional prediction
        assert _rel(got.float().norm(dim=-1), pos.float().norm(dim=-1)) < 1e-2


@pytest.mark.parametrize("h,w,Nt,B,n_cfg,ragged", [(8, 8, 16, 2, 1, False), (8, 12, 19, 2, 2, True), (16, 8, 40, 1, 2, True), (4, 4, 5, 3, 1, True),
                                                    (10, 6, 9, 1, 2, True)])     # odd packed grid (5 x 3): the centred RoPE rows / columns of 1328^2 (83 x 83)
def test_qwen_forward_matches_oracle(qw, h, w, Nt, B, n_cfg, ragged):
    """Tiny width (2 layers, 2 heads), ragged prompts, with and without the negative branch: raw network outputs of both CFG branches
    and the norm-rescaled combination vs the oracle; a padded sample equals the same sample run alone with its text truncated."""
    from oracle import qwen_ref as R
    cfg_o = R.tiny_config()
    sd, cfg = _setup(qw, cfg_o)
    eng = _engine(qw, sd, cfg)
    g = torch.Generator().manual_seed(h * 100 + Nt)
    J = cfg_o.joint_attention_dim
    hp, wp = h // 2, w // 2
    x = _bf(torch.randn(B, hp * wp, 64, generator=g))
    pos_lens = [max(1, Nt - (3 * b if ragged else 0)) for b in range(B)]
    neg_lens = [max(1, Nt - 4 - b) if ragged else Nt for b in range(B)]
    pe, ne = _text(B, Nt, J, pos_lens, g), _text(B, Nt, J, neg_lens, g)
    t = torch.tensor([875.0, 500.0, 120.0][:B])
    tm = qw.model_timestep(t, torch.bfloat16)
    plan = eng.plan(B, n_cfg, h, w, Nt, 1)
    embeds = torch.cat([ne, pe]) if n_cfg == 2 else pe
    lens = (neg_lens + pos_lens) if n_cfg == 2 else pos_lens
    v, raw = plan.transformer_forward(x.bfloat16().cuda(), tm, embeds.cuda(), lens, guidance_scale=4.0, return_raw=True)
    tq = (t.to(torch.bfloat16) / 1000).float()
    with torch.no_grad():
        ref_pos = R.qwen_forward(sd, cfg_o, x, tq, pe, pos_lens, hp, wp)
        assert _rel(raw[-B:], ref_pos) < 2e-2, _rel(raw[-B:], ref_pos)
        if n_cfg == 2:
            ref_neg = R.qwen_forward(sd, cfg_o, x, tq, ne, neg_lens, hp, wp)
            assert _rel(raw[:B], ref_neg) < 2e-2
            ref = R.cfg_rescale_bf16(ref_neg, ref_pos, 4.0)
            assert _rel(v, ref) < 3e-2, _rel(v, ref)
            assert torch.equal(v.cpu(), qw.op_cfg_rescale(raw[:B], raw[-B:], 4.0).cpu())
        else:
            assert torch.equal(v, raw)
    if ragged and B > 1:
        # the last sample alone, text truncated to its valid length (no padding at all): masking == truncation
        b = B - 1
        n = pos_lens[b]
        p1 = eng.plan(1, 1, h, w, n, 1)
        v1 = p1.transformer_forward(x[b:b + 1].bfloat16().cuda(), tm[b:b + 1], pe[b:b + 1, :n].cuda(), [n])
        assert _rel(raw[-1:], v1) < 4e-3, _rel(raw[-1:], v1)
    eng.close()


def test_qwen_rollout_matches_oracle_and_replays(qw):
    """N-step true-CFG rollout (latents, per-step log-probs) vs the oracle on identical draws; adapter.forward() replay of a stored
    transition reproduces the rollout log-prob bit for bit (ratio == 1); the per-step path equals the fused loop."""
    from oracle import qwen_ref as R
    from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler
    cfg_o = R.tiny_config()
    sd, cfg = _setup(qw, cfg_o, seed=11)
    sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[0, 1, 2, 3], num_sde_steps=2, seed=42, dynamics_type="Flow-SDE",
                                               shift=1.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=0.9, max_image_seq_len=8192,
                                               shift_terminal=0.02)
    ad = qw.QwenImageNativeAdapter({k: v.cuda() for k, v in sd.items()}, cfg, sched, latent_storage_dtype="bf16")
    ad.rollout()
    B, N, H, W = 2, 5, 128, 192
    g = torch.Generator().manual_seed(9)
    J = cfg_o.joint_attention_dim
    pos_lens, neg_lens = [21, 17], [6, 6]
    pe = [_bf(torch.randn(n, J, generator=g)).bfloat16() for n in pos_lens]           # ragged lists, as the trainer's collate hands them over
    pm = [torch.ones(n, dtype=torch.long) for n in pos_lens]
    ne = _text(B, 6, J, neg_lens, g).bfloat16()
    nm = torch.ones(B, 6, dtype=torch.long)
    torch.cuda.manual_seed(77)
    kw = dict(prompt=["a", "b"], negative_prompt=None, height=H, width=W, num_inference_steps=N, guidance_scale=4.0,
              prompt_embeds=[p.cuda() for p in pe], prompt_embeds_mask=[m.cuda() for m in pm], negative_prompt_embeds=ne.cuda(),
              negative_prompt_embeds_mask=nm.cuda(), compute_log_prob=True, trajectory_indices="all")
    samples = ad.inference(**kw)
    torch.cuda.manual_seed(77)
    h, w = H // 8, W // 8
    hp, wp = h // 2, w // 2
    from mi355_flow.flux import pack_latents
    init = pack_latents(torch.randn((B, 1, 16, h, w), device="cuda", dtype=torch.bfloat16).reshape(B, 16, h, w)).cpu()
    noise = torch.stack([torch.randn((B, hp * wp, 64), device="cuda", dtype=torch.float32) for _ in range(N)]).cpu()
    ts = samples[0].timesteps.float().cpu()
    sig = ad.scheduler.sigmas.float().cpu()
    assert abs(float(sig[N - 1]) - 0.02) < 1e-6                       # shift_terminal
    nl = ad.scheduler.host_noise_levels()
    assert sum(e > 0 for e in nl) == 2
    pe_pad = torch.zeros(B, max(pos_lens), J)
    for b, p in enumerate(pe):
        pe_pad[b, :p.shape[0]] = p.float()
    ref = R.rollout(sd, cfg_o, pe_pad, pos_lens, ne.float(), neg_lens, 4.0, init, noise, ts, sig, nl, hp, wp, torch.bfloat16)
    assert len(samples) == B and samples[0].all_latents.shape == (N + 1, hp * wp, 64) and samples[0].all_latents.dtype == torch.bfloat16
    assert samples[0].img_shapes == [(1, hp, wp)] and samples[1].prompt_embeds_mask.shape == (17,)
    sde = [i for i in range(N) if nl[i] > 0]
    for b in range(B):
        got = samples[b].all_latents.float().cpu()
        for pos in range(N + 1):
            r = ref["all_latents"][pos, b].float()
            assert ((got[pos] - r).norm() / r.norm()).item() < 2.5e-2, (b, pos)
        lp = samples[b].log_probs.cpu()
        assert lp.shape == (len(sde),)
        torch.testing.assert_close(lp, torch.stack([ref["log_probs"][i, b] for i in sde]), rtol=1e-3, atol=1e-4)
    # replay (grpo.py:229-263): stored (x_i, x_{i+1}) of an SDE step through forward(), padded-batch inputs this time
    i = sde[0]
    x_i = torch.stack([s.all_latents[i] for s in samples]).cuda()
    x_n = torch.stack([s.all_latents[i + 1] for s in samples]).cuda()
    t = samples[0].timesteps[i].reshape(1).expand(B).cuda()
    t_next = samples[0].timesteps[i + 1].reshape(1).expand(B).cuda()
    mask = (torch.arange(max(pos_lens))[None] < torch.tensor(pos_lens)[:, None]).long()
    out = ad.forward(t=t, latents=x_i, prompt_embeds=pe_pad.bfloat16().cuda(), prompt_embeds_mask=mask.cuda(), img_shapes=[s.img_shapes for s in samples],
                     negative_prompt_embeds=ne.cuda(), negative_prompt_embeds_mask=nm.cuda(), guidance_scale=4.0, t_next=t_next, next_latents=x_n,
                     noise_level=nl[i], compute_log_prob=True, return_kwargs=["log_prob", "next_latents_mean"])
    old = torch.stack([s.log_probs[0] for s in samples]).cuda()
    assert torch.equal(torch.exp(out.log_prob - old), torch.ones_like(old))
    # per-step engine calls (callback tensors requested) == the fused loop
    torch.cuda.manual_seed(77)
    s2 = ad.inference(**{**kw, "extra_call_back_kwargs": ["noise_pred"]})
    for b in range(B):
        assert torch.equal(s2[b].all_latents, samples[b].all_latents) and torch.equal(s2[b].log_probs, samples[b].log_probs)
        assert s2[b].extra_kwargs["noise_pred"].shape[0] == N
    # guidance_scale <= 1 or no negative prompt: single branch (qwen_image.py:499-507)
    torch.cuda.manual_seed(77)
    s3 = ad.inference(**{**kw, "guidance_scale": 1.0})
    assert len(s3) == B and torch.isfinite(s3[0].all_latents.float()).all()
    with pytest.raises(NotImplementedError):
        ad.inference(**{**kw, "attention_kwargs": {"ip_adapter": 1}})
    ad.engine.close()


def test_qwen_module_source_is_live(qw):
    """The adapter bound to an nn.Module (what the plugin / an FSDP2- or LoRA-wrapped transformer is): an in-place optimizer step is
    picked up by the next engine call, and the result equals a fresh state-dict bind of the updated values."""
    from oracle import qwen_ref as R
    from mi355_flow.weights import module_from_state_dict
    cfg_o = R.tiny_config()
    sd, cfg = _setup(qw, cfg_o, seed=5)
    mod = module_from_state_dict({k: v.clone().cuda() for k, v in sd.items()})
    ad = qw.QwenImageNativeAdapter(mod, cfg, latent_storage_dtype="bf16")
    ad.rollout()
    g = torch.Generator().manual_seed(2)
    B, hp, wp, Nt = 1, 4, 4, 8
    x = torch.randn(B, hp * wp, 64, generator=g).bfloat16().cuda()
    pe = torch.randn(B, Nt, cfg_o.joint_attention_dim, generator=g).bfloat16().cuda()
    kw = dict(t=torch.tensor([600.0]), latents=x, prompt_embeds=pe, prompt_embeds_mask=torch.ones(B, Nt, dtype=torch.long).cuda(),
              img_shapes=[[(1, hp, wp)]], guidance_scale=1.0, t_next=torch.tensor([400.0]), next_latents=x, noise_level=0.7,
              return_kwargs=["noise_pred", "log_prob"])
    ad.scheduler.set_timesteps(4, mu=0.6)
    with torch.no_grad():                                  # (grad mode syncs once more while looking for trainable parameters)
        a = ad.forward(**kw).noise_pred
    assert ad._live_weights.last_rebinds == 0              # second sync after the constructor's: nothing changed
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if n.endswith("attn.to_q.weight"):
                p.add_(0.05 * torch.randn(p.shape, generator=g).cuda())
        b = ad.forward(**kw).noise_pred
    assert ad._live_weights.last_rebinds == cfg_o.num_layers and not torch.equal(a, b)
    fresh = qw.QwenImageNativeAdapter({k: v.detach().clone() for k, v in mod.state_dict().items()}, cfg, latent_storage_dtype="bf16")
    fresh.rollout()
    fresh.scheduler.set_timesteps(4, mu=0.6)
    with torch.no_grad():
        assert torch.equal(fresh.forward(**kw).noise_pred, b)
    fresh.engine.close()
    ad.engine.close()


def test_qwen_full_width_blocks_at_1024(qw):
    """Qwen-Image width (24 heads x 128, text dim 3584), two blocks, 1024^2 (4096 image tokens) with a ragged CFG text batch, vs the fp32
    oracle.  The 60-layer / 41 GB model itself is exercised by scripts/qwen_bench.py."""
    from oracle import qwen_ref as R
    cfg_o = R.QwenConfig(num_layers=2)
    sd, cfg = _setup(qw, cfg_o, seed=13, std=0.02)
    eng = _engine(qw, sd, cfg)
    B, h, w, Nt = 1, 128, 128, 96
    g = torch.Generator().manual_seed(17)
    x = _bf(torch.randn(B, 4096, 64, generator=g))
    pos_lens, neg_lens = [83], [7]
    pe, ne = _text(B, Nt, 3584, pos_lens, g), _text(B, Nt, 3584, neg_lens, g)
    t = torch.tensor([640.0])
    v, raw = eng.plan(B, 2, h, w, Nt, 1).transformer_forward(x.bfloat16().cuda(), qw.model_timestep(t, torch.bfloat16), torch.cat([ne, pe]).cuda(),
                                                              neg_lens + pos_lens, guidance_scale=4.0, return_raw=True)
    tq = (t.to(torch.bfloat16) / 1000).float()
    from _gpu_oracle import check_in_band
    rp, rpq, _, _ = check_in_band("Qwen-Image full-width 2 blocks, S = 4096 + 96, conditional", raw[1:], R.qwen_forward, sd, cfg_o, x, tq, pe, pos_lens, 64, 64)
    rn, rnq, _, _ = check_in_band("Qwen-Image full-width 2 blocks, unconditional (7-token prompt)", raw[:1], R.qwen_forward, sd, cfg_o, x, tq, ne, neg_lens, 64, 64)
    band3, r3 = _rel(R.cfg_rescale_bf16(rnq, rpq, 4.0), R.cfg_rescale_bf16(rn, rp, 4.0)), _rel(v, R.cfg_rescale_bf16(rn, rp, 4.0))
    print(f"Qwen-Image full-width 2 blocks: norm-rescaled true CFG 4.0 {r3:.3e} (band {band3:.3e})")
    assert r3 < 1.5 * band3 + 1e-3, (r3, band3)
    eng.close()


def test_qwen_full_width_blocks_at_config_e_1328(qw):
    """BASELINE.json configs[4] at ITS OWN shape (reference qwen_image.py:476-600): 1328^2 = a 166 x 166 latent grid = 83 * 83 = 6889 image
    tokens (odd grid: the centred RoPE rows / columns are asymmetric; 6889 is not a multiple of 64, so the ragged text tail starts
    mid-tile), true CFG with ragged prompts, Qwen-Image width, 2 blocks, vs the fp32 oracle (model body unpinned)."""
    from oracle import qwen_ref as R
    cfg_o = R.QwenConfig(num_layers=2)
    sd, cfg = _setup(qw, cfg_o, seed=14, std=0.02)
    eng = _engine(qw, sd, cfg)
    B, h, w, Nt = 1, 166, 166, 96
    Ni = (h // 2) * (w // 2)
    g = torch.Generator().manual_seed(18)
    x = _bf(torch.randn(B, Ni, 64, generator=g))
    pos_lens, neg_lens = [91], [5]
    pe, ne = _text(B, Nt, 3584, pos_lens, g), _text(B, Nt, 3584, neg_lens, g)
    t = torch.tensor([640.0])
    v, raw = eng.plan(B, 2, h, w, Nt, 1).transformer_forward(x.bfloat16().cuda(), qw.model_timestep(t, torch.bfloat16), torch.cat([ne, pe]).cuda(),
                                                              neg_lens + pos_lens, guidance_scale=4.0, return_raw=True)
    assert torch.isfinite(v.float()).all()
    tq = (t.to(torch.bfloat16) / 1000).float()
    from _gpu_oracle import check_in_band
    rp, rpq, _, band = check_in_band("Qwen-Image full-width 2 blocks, 1328^2 (S = 6889 + 96), conditional", raw[1:], R.qwen_forward, sd, cfg_o, x, tq, pe,
                                     pos_lens, h // 2, w // 2)
    rn, rnq, _, _ = check_in_band("Qwen-Image full-width 2 blocks, 1328^2, unconditional (5-token prompt)", raw[:1], R.qwen_forward, sd, cfg_o, x, tq, ne,
                                  neg_lens, h // 2, w // 2)
    band3, r3 = _rel(R.cfg_rescale_bf16(rnq, rpq, 4.0), R.cfg_rescale_bf16(rn, rp, 4.0)), _rel(v, R.cfg_rescale_bf16(rn, rp, 4.0))
    rows = raw[1].float().cpu().reshape(h // 2, w // 2, 64)
    rrows = rp[0].float().cpu().reshape(h // 2, w // 2, 64)
    worst_row = max(float((rows[i] - rrows[i]).norm() / rrows[i].norm()) for i in range(h // 2))
    print(f"Qwen-Image full-width 2 blocks, 1328^2: true CFG {r3:.3e} (band {band3:.3e}), worst image row {worst_row:.3e}")
    assert r3 < 1.5 * band3 + 1e-3 and worst_row < 2.0 * (1.5 * band + 1e-3), (r3, band3, worst_row, band)
    eng.close()


def test_qwen_errors(qw):
    from oracle import qwen_ref as R
    cfg_o = R.tiny_config()
    sd, cfg = _setup(qw, cfg_o)
    eng = qw.QwenEngine(cfg)
    with pytest.raises(KeyError):
        eng.bind_state_dict({k: v.cuda() for k, v in sd.items() if "txt_norm" not in k})
    eng.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    eng.ready()
    plan = eng.plan(1, 1, 8, 8, 8, 1)
    x = torch.zeros(1, 16, 64, dtype=torch.bfloat16, device="cuda")
    pe = torch.zeros(1, 8, cfg_o.joint_attention_dim, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError, match="txt_lens"):
        plan.transformer_forward(x, torch.tensor([500.0]), pe, [9])
    with pytest.raises(RuntimeError):
        qw.QwenEngine(qw.QwenConfig(attention_head_dim=64))
    eng.close()
